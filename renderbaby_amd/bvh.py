"""BVH build through the library's host-side restatement of ``BVH::new``
(crates/engine-bvh/src/bvh.rs:87-150) -- rb_bvh_build in include/rb_abi.h."""
import ctypes as C

import numpy as np

from . import abi
from ._lib import load


def build(tris):
    """tris: abi.GPU_TRIANGLE[] -> (abi.BVH_NODE[], uint32[] indices)."""
    lib = load()
    tris = np.ascontiguousarray(tris, dtype=abi.GPU_TRIANGLE)
    n = len(tris)
    n_nodes = C.c_size_t(0)
    rc = lib.rb_bvh_build(tris.ctypes.data, n, None, 0, C.byref(n_nodes), None)
    if rc:
        raise RuntimeError(f"rb_bvh_build failed: {rc}")
    nodes = np.zeros(n_nodes.value, dtype=abi.BVH_NODE)
    indices = np.zeros(n, dtype=np.uint32)
    rc = lib.rb_bvh_build(tris.ctypes.data, n, nodes.ctypes.data, len(nodes), C.byref(n_nodes), indices.ctypes.data)
    if rc:
        raise RuntimeError(f"rb_bvh_build failed: {rc}")
    return nodes, indices
