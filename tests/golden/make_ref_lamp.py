"""Extracts the DATA of the reference's largest fixture scene,
included/fixtures/scenes/final_cornell_with_lamp_and_spheres.rscn (68 768 triangles, 4 spheres,
2056x2056, 512 spp), into tests/golden/ref_fixtures/lamp_scene.npz: the flat arrays the
Scene -> RenderConfig adapter produces from it, minus the BVH (rebuilt on load).  Runs only in the
dev container; the GPU box has no /root/reference.

    python tests/golden/make_ref_lamp.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from renderbaby_amd import scene_io  # noqa: E402

SRC = "/root/reference/included/fixtures/scenes/final_cornell_with_lamp_and_spheres.rscn"
s = scene_io.load_scene(SRC)
t = s.bvh_triangles
verts = np.stack([t["v0"], t["v1"], t["v2"]], axis=1).astype(np.float32)      # [n, 3, 3]
# most vertices are shared: store the unique ones + indices (smaller than 2.5 MB of raw corners)
uniq, inv = np.unique(verts.reshape(-1, 3), axis=0, return_inverse=True)
assert np.array_equal(uniq[inv].reshape(verts.shape), verts)
uv = s.uvs.reshape(-1, 2).astype(np.float32)                                  # one pair per un-indexed corner
assert len(uv) == len(t) * 3 and np.array_equal(t["v0_index"], np.arange(len(t), dtype=np.uint32) * 3)
uq_uv, inv_uv = np.unique(uv, axis=0, return_inverse=True)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fixtures", "lamp_scene.npz")
np.savez_compressed(out, uniforms=s.uniforms, spheres=s.spheres, lights=s.lights, meshes=s.meshes,
                    unique_vertices=uniq, corner_index=inv.astype(np.uint32).reshape(-1, 3),
                    mesh_index=t["mesh_index"].astype(np.uint16), unique_uvs=uq_uv,
                    uv_index=inv_uv.astype(np.uint32).reshape(-1, 3))
print(len(t), "triangles,", len(uniq), "unique vertices,", len(s.meshes), "meshes,", len(s.spheres), "spheres ->",
      os.path.getsize(out), "bytes")
