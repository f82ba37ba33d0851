"""Generates the golden fixtures in this directory from the CPU oracle.

The reference has no golden vectors for this path and cannot be run here
(SURVEY.md section 8(c)), so these are ORACLE outputs: they pin the oracle against
drift and give the HIP path committed vectors to reproduce.  Each .npz is data
only: the flattened scene arrays (inputs) and the expected f32 accumulation,
RGBA8 frame (x mirrored) and work counters (outputs).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from renderbaby_amd import scenes  # noqa: E402
from tests import _oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "cornell_32x32_4spp_d4": lambda: scenes.cornell(32, 32, 4, 4, bvh_builder=_oracle.bvh_build),
    "cornell_40x24_3spp_d8": lambda: scenes.cornell(40, 24, 3, 8, bvh_builder=_oracle.bvh_build),
    "feature_24x16_3spp": lambda: scenes.feature_scene(24, 16, 3, 5, bvh_builder=_oracle.bvh_build),
    "feature_hash_24x16_3spp": lambda: scenes.feature_scene(24, 16, 3, 5, color_hash=1, bvh_builder=_oracle.bvh_build),
    "mesh578_32x20_2spp": lambda: scenes.mesh_scene(12, 12, 32, 20, 2, 5, seed=7, bvh_builder=_oracle.bvh_build),
}


def save(name, scene):
    acc, out, rgba, st = _oracle.render(scene)
    d = dict(uniforms=scene.uniforms, spheres=scene.spheres, lights=scene.lights, meshes=scene.meshes,
             bvh_nodes=scene.bvh_nodes, bvh_indices=scene.bvh_indices, bvh_triangles=scene.bvh_triangles,
             uvs=scene.uvs, n_textures=np.int32(len(scene.textures)), accum=acc, rgba=rgba,
             stats=np.array([st[k] for k in _oracle.STAT_KEYS], dtype=np.uint64))
    for i, (w, h, data) in enumerate(scene.textures):
        d[f"tex{i}_wh"] = np.array([w, h], dtype=np.uint32)
        d[f"tex{i}_data"] = np.asarray(data, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes", st)


if __name__ == "__main__":
    for n, mk in CASES.items():
        save(n, mk())
