"""Round-4 additions behind the boundary: the sphere tree from either builder (device: level-synchronous median splits,
host: the same splits recursively) under every kernel that walks it, the pooled sphere kernel's tie rule and degenerate
inputs, rb_reserve, the environment's reference-walk override, the debug hook of the profiling build, and the chunked walk's
tree built on the device (one block per reference leaf) against the host builder's: same invariants, same census, same frames."""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, _lib, abi, scenes
from renderbaby_amd.engine import Change
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


# ---------------------------------------------------------------- the chunked walk's tree from the device builder
def _chunk(scene, **kw):
    rc = RenderConfig.from_scene(scene)
    e = Engine.new(rc, **kw)
    f = e.render(rc)
    acc, name, builder, census = e.read_accumulation(), e.last_kernel_name(), e.chunk_tree_builder()[0], e.debug_chunk_tree()
    e.close()
    return f.pixels, acc, name, builder, census


def _same_census(d, h):
    # same split-off sets and the same split sizes, so the same shape: nodes, positions, depth and chunks agree exactly.  Which of
    # several triangles with EQUAL centroids lands in which chunk is the sort's business (nth_element there, a bitonic network
    # here), so the count of child slots whose margin is unbounded may differ by the few chunks a wall-sized triangle moved between
    for k in ("nodes", "positions", "depth", "chunks"):
        assert d[k] == h[k], (k, d, h)
    assert abs(d["unbounded"] - h["unbounded"]) <= 0.1 * h["unbounded"] + 4, (d, h)


def _with_tree(s, nodes, idx, **params):
    s = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, nodes, idx, s.bvh_triangles, s.uvs)
    return s.with_params(**params) if params else s


@pytest.mark.parametrize("grid,size", [(6, 24), (24, 40), (70, 48)])
def test_device_built_chunk_tree_equals_the_hosts_in_everything_that_matters(grid, size):
    # 72 + ..., 1 152 + ..., 9 800 + ... triangles: forced onto the device builder (its default starts at 16 384 slots); the
    # tree is read back and put through the host's invariant checker, its census compared with the host builder's
    s = scenes.mesh_scene(grid, grid, size, size, 2, 5, seed=grid)
    o_acc, _, o_rgba, _ = _oracle.render(s)
    d_rgba, d_acc, name, builder, d_cen = _chunk(s, chunk_tree="device")
    assert name == "k_trace_chunk" and builder == "device" and d_cen is not None
    h_rgba, h_acc, name, builder, h_cen = _chunk(s, chunk_tree="host")
    assert name == "k_trace_chunk" and builder == "host"
    _same_census(d_cen, h_cen)
    for acc, rgba in ((d_acc, d_rgba), (h_acc, h_rgba)):
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32))
        assert np.array_equal(rgba, o_rgba)


def test_device_chunk_tree_is_the_default_for_c3_and_splits_off_the_lamps_walls():
    from renderbaby_amd import refscenes
    for s, positions in ((scenes.mesh_c3().with_params(width=64, height=48, spp=1), 50178), (refscenes.ref_lamp(width=48, height=48, spp=1), 68768)):
        d_rgba, d_acc, name, builder, d_cen = _chunk(s)
        assert name == "k_trace_chunk" and builder == "device" and d_cen["positions"] == positions
        h_rgba, h_acc, _, builder, h_cen = _chunk(s, chunk_tree="host")
        assert builder == "host"
        _same_census(d_cen, h_cen)
        assert np.array_equal(d_acc.view(np.uint32), h_acc.view(np.uint32)) and np.array_equal(d_rgba, h_rgba)
        assert 0 < d_cen["unbounded"] < d_cen["chunks"] / 3


def test_device_chunk_tree_with_caller_made_trees_and_invalid_slots():
    spec = importlib.util.spec_from_file_location("gpu_parity_helpers", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    gp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp)
    base = scenes.mesh_scene(20, 20, 40, 32, 2, 4, seed=31)
    o_acc = _oracle.render(base)[0]
    for max_leaf, lop, want in ((401, 0, "host"), (300, 0, "device"), (5, 0, "device"), (1, 0, "device"), (40, 4, "device")):   # leaves of 401 / 201 / 4 / 1 / 40 triangles
        nodes, idx = gp._py_tree(base.bvh_triangles, max_leaf, lopsided=lop)
        s = _with_tree(base, nodes, idx)
        fat = int(nodes["primitive_count"].max())
        rgba, acc, name, builder, cen = _chunk(s, chunk_tree="device")
        # leaves beyond the 256 triangles a block holds go to the host builder, whatever was asked for
        assert name == "k_trace_chunk" and builder == ("host" if fat > 256 else "device") and builder == want, (max_leaf, lop, fat, builder)
        assert cen["positions"] == len(base.bvh_triangles)
        assert np.array_equal(acc.view(np.uint32), _oracle.render(s)[0].view(np.uint32)), (max_leaf, lop)
    # indices beyond the triangle count (guard shader.wgsl:336) and a leaf with nothing valid in it
    nodes, idx = gp._py_tree(base.bvh_triangles, 16)
    idx = idx.copy()
    leaf = np.flatnonzero(nodes["primitive_count"] > 0)[3]
    lo, cnt = int(nodes["first_primitive"][leaf]), int(nodes["primitive_count"][leaf])
    idx[lo:lo + cnt] = 0xFFFFFFF0          # a whole leaf of invalid slots
    idx[::7] = len(base.bvh_triangles) + 5  # and every seventh elsewhere
    s = _with_tree(base, nodes, idx)
    o_acc = _oracle.render(s)[0]
    n_valid = int((idx < len(base.bvh_triangles)).sum())
    for kw in (dict(chunk_tree="device"), dict(chunk_tree="host")):
        rgba, acc, name, builder, cen = _chunk(s, **kw)
        assert name == "k_trace_chunk" and builder == kw["chunk_tree"] and cen["positions"] == n_valid
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), kw


def test_device_chunk_tree_of_degenerate_triangles():
    # points, needles and repeated triangles: zero normals, zero bounds, equal sort keys everywhere
    s = scenes.mesh_scene(12, 12, 32, 24, 2, 4, seed=5)
    t = s.bvh_triangles.copy()
    t["v1"][::3] = t["v0"][::3]            # needles (no normal)
    t["v2"][::9] = t["v0"][::9]
    t["v1"][::9] = t["v0"][::9]            # points
    t[100:160] = t[100]                    # sixty copies of one triangle: t ties go to the lower rank
    from renderbaby_amd import bvh
    nodes, idx = bvh.build(t)
    s = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, nodes, idx, t, s.uvs)
    o_acc = _oracle.render(s)[0]
    for kw in (dict(chunk_tree="device"), dict(chunk_tree="host")):
        rgba, acc, name, builder, cen = _chunk(s, **kw)
        assert name == "k_trace_chunk" and builder == kw["chunk_tree"] and cen is not None
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), kw


def test_large_mesh_whose_leaves_the_device_builder_declines_is_fetched_back_for_the_host_builder():
    # from 16 384 elements up rb_update keeps no host copy of the mesh (the device builder needs none); a caller's tree with
    # leaves of more than 256 triangles still goes to the host builder, which then reads the mesh back from the device
    spec = importlib.util.spec_from_file_location("gpu_parity_helpers", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    gp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp)
    s = scenes.mesh_c3().with_params(width=48, height=32, spp=1)
    nodes, idx = gp._py_tree(s.bvh_triangles, 1000)
    assert int(nodes["primitive_count"].max()) > 256
    s = _with_tree(s, nodes, idx)
    rgba, acc, name, builder, cen = _chunk(s)
    assert name == "k_trace_chunk" and builder == "host" and cen["positions"] == 50178
    r_rgba, r_acc, r_name, _, r_cen = _chunk(s, reference_walk=True)
    assert r_name.startswith("k_trace_bvh") and r_cen is None
    assert np.array_equal(acc.view(np.uint32), r_acc.view(np.uint32)) and np.array_equal(rgba, r_rgba)


def test_updates_of_a_large_mesh_rebuild_the_device_tree_from_what_is_on_the_device():
    # 50 178 triangles: rb_update keeps no host copy, the device builder makes the tree.  Triangles, count, tree and indices are then
    # changed one at a time on the SAME engine; after every change the frame must be the one a fresh reference-walk engine
    # renders of the scene as it now stands, and the tree read back must pass the checker
    spec = importlib.util.spec_from_file_location("gpu_parity_helpers", os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_parity.py"))
    gp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp)
    s = scenes.mesh_c3().with_params(width=48, height=32, spp=1)

    def reference(scene):
        rc = RenderConfig.from_scene(scene)
        e = Engine.new(rc, reference_walk=True)
        e.render(rc)
        a = e.read_accumulation()
        e.close()
        return a

    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    e.render(rc)
    assert e.chunk_tree_builder()[0] == "device" and e.debug_chunk_tree()["positions"] == 50178
    assert np.array_equal(e.read_accumulation().view(np.uint32), reference(s).view(np.uint32))
    # (1) new triangles, tree and indices kept (the boxes no longer fit all of them: such children are always entered)
    t = s.bvh_triangles.copy()
    rng = np.random.default_rng(9)
    for k in ("v0", "v1", "v2"):
        t[k] += rng.uniform(-0.05, 0.05, t[k].shape).astype(np.float32)
    s1 = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, t, s.uvs)
    e.render(RenderConfig(uniforms=Change.update(s.uniforms), bvh_triangles=Change.update(t)))   # (a render needs uniforms in its update: gpu_wrapper.rs:313)
    assert e.chunk_tree_builder()[0] == "device" and e.debug_chunk_tree()["positions"] == 50178
    assert np.array_equal(e.read_accumulation().view(np.uint32), reference(s1).view(np.uint32))
    # (2) a caller's tree with leaves the device builder declines: the host builder fetches the NEW triangles back
    nodes, idx = gp._py_tree(t, 1000)
    s2 = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, nodes, idx, t, s.uvs)
    e.render(RenderConfig(uniforms=Change.update(s.uniforms), bvh_nodes=Change.update(nodes), bvh_indices=Change.update(idx)))
    assert e.chunk_tree_builder()[0] == "host" and e.debug_chunk_tree()["positions"] == 50178
    assert np.array_equal(e.read_accumulation().view(np.uint32), reference(s2).view(np.uint32))
    # (3) back to a tree of small leaves with a seventh of the indices invalid: the device builder again
    nodes, idx = gp._py_tree(t, 64)
    idx = idx.copy()
    idx[::7] = len(t) + 3
    s3 = scenes.Scene(s.uniforms, s.spheres, s.lights, s.meshes, nodes, idx, t, s.uvs)
    e.render(RenderConfig(uniforms=Change.update(s.uniforms), bvh_nodes=Change.update(nodes), bvh_indices=Change.update(idx)))
    assert e.chunk_tree_builder()[0] == "device" and e.debug_chunk_tree()["positions"] == int((idx < len(t)).sum())
    assert np.array_equal(e.read_accumulation().view(np.uint32), reference(s3).view(np.uint32))
    e.close()
