"""Round-4 additions behind the boundary: the sphere tree from either builder (device: level-synchronous median splits,
host: the same splits recursively) under every kernel that walks it, the pooled sphere kernel's tie rule and degenerate
inputs, rb_reserve, the environment's reference-walk override, and the debug hook of the profiling build."""
import ctypes as C
import os

import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, _lib, abi, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu


def _frame(scene, **kw):
    rc = RenderConfig.from_scene(scene)
    kernel = kw.pop("kernel", abi.KERNEL_STREAM)
    e = Engine.new(rc, kernel=kernel, **kw)
    f = e.render(rc)
    acc, st, name, tree = e.read_accumulation(), e.stats(), e.last_kernel_name(), e.sphere_tree_builder()
    e.close()
    return f.pixels, acc, st, name, tree


@pytest.mark.parametrize("n,extent,size", [(65, 3.0, 40), (1023, 10.0, 64), (1025, 10.0, 64), (20_000, 30.0, 112)])
def test_both_sphere_tree_builders_deliver_the_linear_scans_frame(n, extent, size):
    # around the thresholds: 65 = the first count with a tree, 1023 / 1025 = the host / device builder by default; leaves of 16
    # and 4-wide nodes with one, two, three or four children come out of these counts
    s = scenes.spheres_scene(n=n, width=size, height=size, spp=2, max_depth=5, extent=extent)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    for kw, want_kernel, want_tree in ((dict(), "k_trace_sph", "device-median" if n >= 1024 else "host-median"),
                                       (dict(sphere_tree="host"), "k_trace_sph", "host-median"),
                                       (dict(sphere_tree="device"), "k_trace_sph", "device-median"),
                                       (dict(sphere_tree="device", no_leaf_stepping=True), "k_trace", "device-median"),
                                       (dict(sphere_tree="device", kernel=abi.KERNEL_QUEUE), "k_queue", "device-median"),
                                       (dict(sphere_tree="host", kernel=abi.KERNEL_PIXEL), "k_pixel", "host-median")):
        rgba, acc, st, name, tree = _frame(s, **dict(kw))
        assert name == want_kernel and tree[0] == want_tree, (kw, name, tree)
        bad = np.argwhere(acc.view(np.uint32) != o_acc.view(np.uint32))
        assert len(bad) == 0, (kw, len(bad), bad[:4])
        assert np.array_equal(rgba, o_rgba)
        assert st["segments"] == o_st["segments"]


def test_sphere_tree_with_coincident_and_nested_spheres():
    # every centre the same (the builders' sorts see equal keys; the tie rule decides everything), radii nested: the winner is
    # the lowest index among equal t, as in the reference's scan
    s = scenes.spheres_scene(n=300, width=48, height=48, spp=2, max_depth=4, extent=5.0)
    sp = s.spheres.copy()
    sp["center"][:150] = sp["center"][0]
    sp["radius"][:150] = np.float32(0.7)            # 150 identical spheres: t ties among them
    sp["center"][150:] = sp["center"][150]
    sp["radius"][150:] = np.linspace(0.2, 2.0, 150).astype(np.float32)   # nested
    s = scenes.Scene(s.uniforms, sp, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs)
    o_acc, _, o_rgba, _ = _oracle.render(s)
    for kw in (dict(sphere_tree="device"), dict(sphere_tree="host"), dict(no_sphere_bvh=True)):
        rgba, acc, _, _, _ = _frame(s, **kw)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), kw
        assert np.array_equal(rgba, o_rgba)


def test_sphere_tree_next_to_a_multi_node_mesh():
    # more than 64 spheres AND a multi-node mesh: the mesh walks run the per-lane sphere walk inside segment_finish (their
    # SPHTREE instantiation), on the same tree and the same LDS stack columns
    m = scenes.mesh_scene(24, 24, 64, 40, 3, 5, seed=7)
    sp = scenes.spheres_scene(n=400, width=8, height=8, spp=1, max_depth=1, extent=4.0).spheres.copy()
    sp["center"] += np.array([0.0, 2.0, -6.0], np.float32)
    s = scenes.Scene(m.uniforms, sp, m.lights, m.meshes, m.bvh_nodes, m.bvh_indices, m.bvh_triangles, m.uvs)
    o_acc, _, o_rgba, _ = _oracle.render(s)
    for kw, want in ((dict(), "k_trace_chunk"), (dict(sphere_tree="device"), "k_trace_chunk"), (dict(reference_walk=True), "k_trace_bvh"),
                     (dict(reference_walk=True, lds_mode=1), "k_trace_bvh"), (dict(fast_bvh=True), "k_trace_fast")):
        rgba, acc, _, name, tree = _frame(s, **kw)
        assert name.startswith(want) and tree[0] != "", (kw, name, tree)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), kw
        assert np.array_equal(rgba, o_rgba)


def test_reserve_takes_the_allocation_out_of_the_first_dispatch():
    s = scenes.cornell(256, 256, 8, 4)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    e.update(rc)
    e.reserve(8)            # prepared data + the colour buffer, nothing traced
    assert e.stats()["segments"] == 0
    e.dispatch(0, 8)
    e.sync()
    a = e.read_accumulation()
    e.close()
    o_acc, _, _, _ = _oracle.render(s)
    assert np.array_equal(a.view(np.uint32), o_acc.view(np.uint32))


def test_the_environment_can_force_the_reference_walk(monkeypatch):
    s = scenes.mesh_scene(24, 24, 64, 40, 2, 5, seed=3)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    a = e.render(rc).pixels.copy()
    assert e.last_kernel_name() == "k_trace_chunk"
    e.close()
    monkeypatch.setenv("RB_REFERENCE_WALK", "1")
    e = Engine.new(rc, chunk_walk=True)     # the host's own flags lose
    b = e.render(rc).pixels.copy()
    assert e.last_kernel_name().startswith("k_trace_bvh")
    e.close()
    assert np.array_equal(a, b)


def test_the_product_build_counts_no_passes():
    out = (C.c_uint64 * 64)()
    if os.environ.get("RB_LIBRARY_PATH"):
        pytest.skip("a variant library is loaded")
    assert _lib.load().rb_debug_walk_profile(out, 0) != 0   # only a profiling build (tools/walk_profile.sh) has counters
