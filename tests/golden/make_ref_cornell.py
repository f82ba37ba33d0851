"""Extracts the DATA of the reference's Cornell-box fixture (vertex coordinates, faces, material
table) into tests/golden/ref_fixtures/cornell_box.npz so that tests on the GPU box -- where /root/reference
does not exist -- can render the reference's own geometry.  Runs only in the dev container.

    python tests/golden/make_ref_cornell.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from renderbaby_amd import scene_io  # noqa: E402

SRC = "/root/reference/included/fixtures/cornell_box"
obj = scene_io.parse_obj(open(os.path.join(SRC, "cornell-box.obj")).read())
mats = scene_io.parse_mtl(open(os.path.join(SRC, "cornell-box.mtl")).read())
mesh = scene_io.obj_to_mesh(obj, mats)
np.savez_compressed(
    os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fixtures", "cornell_box.npz"),
    vertices=mesh.vertices, uvs=mesh.uvs, material_index=mesh.material_index,
    mat_names=np.array([m.name for m in mats]),
    mat_ka=np.array([m.ka for m in mats], np.float64), mat_kd=np.array([m.kd for m in mats], np.float64),
    mat_ks=np.array([m.ks for m in mats], np.float64), mat_ke=np.array([m.ke for m in mats], np.float64),
    mat_d=np.array([m.d for m in mats], np.float64), mat_ns=np.array([m.ns for m in mats], np.float64))
print(len(mesh.material_index), "triangles,", len(mats), "materials:", [m.name for m in mats])
