"""The chunked walk's tree (rb_bvh.cpp, chunk_tree_build) on the host: the structural invariants k_trace_chunk relies on
without checking, verified by the library itself (rb_debug_chunk_tree -> chunk_tree_check) for the reference builder's trees,
for caller-made ones and for the fuzzer's nasty geometry.  No GPU."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from renderbaby_amd import _lib, abi, scenes
from tests import _refscenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _check(tris, nodes, indices):
    """-> dict(built, nodes, positions, depth, chunks, unbounded); raises with the library's message on a violation."""
    lib = _lib.load()
    tris = np.ascontiguousarray(tris, dtype=abi.GPU_TRIANGLE)
    nodes = np.ascontiguousarray(nodes, dtype=abi.BVH_NODE)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    out = (C.c_uint64 * 6)()
    rc = lib.rb_debug_chunk_tree(tris.ctypes.data, len(tris), nodes.ctypes.data, len(nodes), indices.ctypes.data, len(indices), out)
    assert rc == 0, (rc, (lib.rb_last_error(None) or b"").decode())
    return dict(zip(("built", "nodes", "positions", "depth", "chunks", "unbounded"), (int(x) for x in out)))


@pytest.mark.parametrize("n", [8, 24, 70])
def test_reference_builder_trees(n):
    s = scenes.mesh_scene(n, n, 32, 24, 1, 3, seed=n)
    r = _check(s.bvh_triangles, s.bvh_nodes, s.bvh_indices)
    assert r["built"] == 1 and r["positions"] == len(s.bvh_triangles)
    assert r["chunks"] >= r["positions"] / 16 and r["depth"] <= 31
    assert r["unbounded"] >= 1   # the 4 x 4 light quad: beyond the range the margin is claimed for


def test_c3_and_the_reference_lamp_scene():
    s = scenes.mesh_c3()
    r = _check(s.bvh_triangles, s.bvh_nodes, s.bvh_indices)
    assert r["built"] == 1 and r["positions"] == 50178
    s = _refscenes.ref_lamp(width=8, height=8, spp=1)
    r = _check(s.bvh_triangles, s.bvh_nodes, s.bvh_indices)
    assert r["built"] == 1 and r["positions"] == 68768
    # the wall-sized triangles got subtrees of their own: far fewer unbounded child slots than chunks
    assert 0 < r["unbounded"] < r["chunks"] / 20


def test_caller_made_trees():
    gp = _load("tests/test_gpu_parity.py", "gpu_parity_helpers")
    base = scenes.mesh_scene(20, 20, 32, 24, 1, 3, seed=31)
    tris = base.bvh_triangles
    for max_leaf, lop in ((300, 0), (5, 0), (40, 4)):
        nodes, idx = gp._py_tree(tris, max_leaf, lopsided=lop)
        r = _check(tris, nodes, idx)
        assert r["built"] == 1 and r["positions"] == len(tris), (max_leaf, lop)
    nodes, idx = gp._py_tree(tris, 40, lopsided=8)       # too deep for the stack below the caller's leaves
    assert _check(tris, nodes, idx)["built"] == 0
    # boxes that no longer contain their triangles: every such child slot must have an unbounded margin (the checker
    # refuses a slot that could still be culled).  A finer mesh, so that whole boxes have finite margins.
    fine = scenes.mesh_scene(70, 70, 32, 24, 1, 3, seed=21)
    rng = np.random.default_rng(5)
    nodes = fine.bvh_nodes.copy()
    c = (nodes["aabb_min"] + nodes["aabb_max"]) * np.float32(0.5)
    h = (nodes["aabb_max"] - nodes["aabb_min"]) * np.float32(0.5) * rng.uniform(0.3, 0.9, c.shape).astype(np.float32)
    nodes["aabb_min"], nodes["aabb_max"] = c - h, c + h
    shrunk = _check(fine.bvh_triangles, nodes, fine.bvh_indices)
    whole = _check(fine.bvh_triangles, fine.bvh_nodes, fine.bvh_indices)
    assert shrunk["built"] == 1 and shrunk["unbounded"] > whole["unbounded"] + 200 and whole["unbounded"] < whole["chunks"] / 3


def test_indices_beyond_the_triangle_count_are_left_out():
    s = scenes.mesh_scene(10, 10, 32, 24, 1, 3)
    n = len(s.bvh_triangles)
    r = _check(s.bvh_triangles[: n // 2], s.bvh_nodes, s.bvh_indices)   # shader.wgsl:336: ids >= the count are skipped
    assert r["built"] == 1 and r["positions"] == int((s.bvh_indices < n // 2).sum())


def test_fuzzed_geometry():
    fuzz = _load("tools/fuzz_parity.py", "fuzz_parity_helpers")
    built = 0
    for seed in range(400):
        s = fuzz.random_scene(seed)
        if len(s.bvh_nodes) <= 1:
            continue
        r = _check(s.bvh_triangles, s.bvh_nodes, s.bvh_indices)   # degenerate, huge, tiny, coincident triangles
        built += r["built"]
        assert r["built"] == 0 or r["positions"] == len(s.bvh_triangles), seed
    assert built > 100


def test_a_sliver_beyond_the_claimed_range_is_always_entered():
    # The stored bounds carry |e1| |e2|, the range they are claimed for is about L^2: a 0.5 x 0.001 sliver has
    # |e1| |e2| / 1e-6 = 500 but L^2 / 1e-6 = 2.5e5 > 1.5e5, so every child slot above it must carry +inf as its floor bound.
    base = scenes.mesh_scene(70, 70, 32, 24, 1, 3, seed=70)   # 0.16-unit cells: inside the range
    gp = _load("tests/test_gpu_parity.py", "gpu_parity_helpers2")
    nodes, indices = gp._py_tree(base.bvh_triangles, 128, 0)
    r0 = _check(base.bvh_triangles, nodes, indices)
    tris = base.bvh_triangles.copy()
    v0 = tris["v0"][5][:3].copy()
    tris["v1"][5][:3] = v0 + np.float32([0.5, 0.0, 0.0])
    tris["v2"][5][:3] = v0 + np.float32([0.0, 0.001, 0.0])
    nodes, indices = gp._py_tree(tris, 128, 0)
    r = _check(tris, nodes, indices)
    assert r["built"] == 1 and r["unbounded"] > r0["unbounded"]
