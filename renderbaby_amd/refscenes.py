"""Scenes built from DATA of the reference's fixtures (tests/golden/ref_fixtures/*.npz: vertices, materials, uniforms
extracted by tests/golden/make_ref_cornell.py / make_ref_lamp.py in the dev container).  Input data for the parity tests
and for `bench.py --workload lamp`; nothing here touches the oracle."""
import os

import numpy as np

from . import abi, scene_io, scenes

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")


def ref_cornell_mesh():
    z = np.load(os.path.join(HERE, "golden", "ref_fixtures", "cornell_box.npz"))
    mats = [scene_io.ObjMaterial(str(n), list(ka), list(kd), list(ks), list(ke), float(d), float(ns), 2)
            for n, ka, kd, ks, ke, d, ns in zip(z["mat_names"], z["mat_ka"], z["mat_kd"], z["mat_ks"], z["mat_ke"],
                                                z["mat_d"], z["mat_ns"])]
    return scene_io.SceneMesh(z["vertices"], z["uvs"], z["material_index"], mats)


def ref_cornell(width=64, height=48, spp=4, max_depth=5, n_spheres=8, bvh_builder=None):
    """The reference's Cornell box (32 triangles: walls, two boxes, area light; MTL materials) through
    the adapter restatement, plus seeded spheres placed like scenes.cornell_spheres."""
    mesh = ref_cornell_mesh()
    sp = scenes.cornell_spheres(n=n_spheres) if n_spheres else np.zeros(0, abi.SPHERE)
    if n_spheres:
        sp = sp.copy()
        sp["center"][:, 1] -= 0.4   # this box's floor is at y = -0.16
    u = scenes.make_uniforms(width, height, spp, max_depth, cam_pos=(-0.25, 2.6, 6.5), cam_dir=(0, 0, -1),
                             ground_enabled=0, sky=(0, 0, 0))
    return scene_io.scene_to_flat([mesh], spheres=sp, uniforms=u, bvh_builder=bvh_builder, name="ref_cornell")


def ref_lamp(width=None, height=None, spp=None, max_depth=None, bvh_builder=None):
    """The reference's largest fixture scene (final_cornell_with_lamp_and_spheres.rscn: Cornell box,
    a 68 736-triangle lamp, 4 spheres; 2056x2056, 512 spp in the file) from the flat arrays extracted
    by tests/golden/make_ref_lamp.py; the BVH is rebuilt here."""
    z = np.load(os.path.join(HERE, "golden", "ref_fixtures", "lamp_scene.npz"))
    tris = z["unique_vertices"][z["corner_index"].astype(np.int64)]          # [n, 3, 3]
    tuv = z["unique_uvs"][z["uv_index"].astype(np.int64)]                    # [n, 3, 2]
    mi = z["mesh_index"].astype(np.int64)
    groups = [(z["meshes"][m]["material"], tris[mi == m]) for m in range(len(z["meshes"]))]
    uv_groups = [tuv[mi == m] for m in range(len(z["meshes"]))]
    u = z["uniforms"].copy()
    if width:
        u["width"] = width
    if height:
        u["height"] = height
    if spp:
        u["total_samples"] = spp
    if max_depth:
        u["max_depth"] = max_depth
    return scenes._finish("ref_lamp", u, z["spheres"], z["lights"], groups, uv_groups, bvh_builder=bvh_builder)
