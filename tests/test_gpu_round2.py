"""Round-2 behaviour of the boundary: refused updates leave the previous scene live, the caller's triangle
count under Keep (shader.wgsl:336), the progressive iterator running one pass ahead of the read-back, and
several devices behind one handle (rb_create_multi) -- all through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from renderbaby_amd import Change, Engine, RenderConfig, RenderError, abi, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu


def _render_again(e):
    """rb_render without an update: what a host does that keeps rendering after a failed update."""
    return e.render_current().pixels


def test_refused_update_leaves_the_previous_scene_renderable():
    # validate_scene used to run after the buffers had been overwritten, so a refused tree stayed
    # live on an engine that still accepted rb_render / rb_iter_next (a GPU hang for a cyclic tree).
    s = scenes.mesh_scene(12, 12, 24, 16, 2, 3)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, reference_walk=True)
    good = e.render(rc).pixels.copy()
    nodes = s.bvh_nodes.copy()
    nodes[1]["primitive_count"] = 0
    nodes[1]["left"] = 0   # cycle back to the root
    nodes[1]["right"] = 0
    for bad_field, value, code in (("bvh_nodes", nodes, 13), ("bvh_indices", s.bvh_indices[:5].copy(), 13)):
        bad = RenderConfig(uniforms=Change.update(s.uniforms))
        setattr(bad, bad_field, Change.create(value))
        with pytest.raises(RenderError) as ei:
            e.update(bad)
        assert ei.value.code == code
        assert np.array_equal(_render_again(e), good)       # the refused update touched nothing
    tris = s.bvh_triangles.copy()
    tris[3]["mesh_index"] = 99
    with pytest.raises(RenderError) as ei:
        e.update(RenderConfig(uniforms=Change.update(s.uniforms), bvh_triangles=Change.create(tris)))
    assert ei.value.code == 7
    assert np.array_equal(_render_again(e), good)
    # the iterator of the still-live scene works as well
    w, h = e.size()
    out = np.empty((h, w, 4), dtype=np.uint8)
    cfg, keep = RenderConfig.from_scene(s, create=False).to_c()
    e._check(e._lib.rb_iter_begin(e._h, C.byref(cfg)))
    del keep
    last = None
    while e._lib.rb_iter_has_next(e._h):
        e._check(e._lib.rb_iter_next(e._h, out.ctypes.data))
        last = out.copy()
    assert np.array_equal(last, good)
    e.close()


@pytest.mark.parametrize("kw", [dict(reference_walk=True), dict(host_bvh=True), dict(chunk_walk=True)], ids=["reference-walk", "own-tree", "chunk-walk"])
def test_keep_honours_the_callers_triangle_count(kw):
    # shader.wgsl:336 skips triangle ids >= uniforms.bvh_triangle_count, and gpu_wrapper.rs:489-495 leaves the
    # caller's count in force when the triangles are Keep: a uniforms-only update with a smaller count renders
    # fewer triangles.  The oracle applies the same patch-up rule (rbo_scene.counts_kept).
    s = scenes.mesh_scene(10, 10, 32, 20, 2, 3)
    n = len(s.bvh_triangles)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, **kw)
    full = e.render(rc).pixels.copy()
    assert np.array_equal(full, _oracle.render(s)[2])
    for count in (n // 2, 0, n + 1000):
        u = s.uniforms.copy()
        u["bvh_triangle_count"] = count
        got = e.render(RenderConfig(uniforms=Change.update(u))).pixels.copy()
        kept = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs, s.textures)
        want = _oracle.render(kept, counts_kept=_oracle.KEPT_TRIANGLES)[2]
        assert np.array_equal(got, want), count
        assert (count >= n) == np.array_equal(want, full)   # the count really bites
    e.close()


@pytest.mark.parametrize("per_frame", [1, 3])
def test_iterator_running_ahead_delivers_the_same_frames(per_frame):
    # rb_iter_next starts the next pass group on the second frame slot before it copies the current frame out
    s = scenes.feature_scene(40, 24, 7, 4)
    rc = RenderConfig.from_scene(s)
    frames = {}
    for mode, kw in (("ahead", dict()), ("plain", dict(no_run_ahead=True))):
        e = Engine.new(rc, **kw)
        it = e.frame_iterator(rc, passes_per_frame=per_frame)
        frames[mode] = [f.pixels.copy() for f in it]
        acc = e.read_accumulation()
        st = e.stats()
        assert st["paths"] == 40 * 24 * 7
        e.close()
        o_acc, _, o_rgba, _ = _oracle.render(s)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), mode
        assert np.array_equal(frames[mode][-1], o_rgba), mode
    assert len(frames["ahead"]) == len(frames["plain"]) == -(-7 // per_frame)
    for a, b in zip(frames["ahead"], frames["plain"]):
        assert np.array_equal(a, b)
    # frame k is the running average of the first (k + 1) * per_frame samples
    k = 1
    o = _oracle.render(s, 0, min((k + 1) * per_frame, 7))[2]
    assert np.array_equal(frames["ahead"][k], o)


def test_a_pass_started_ahead_is_dropped_when_the_scene_changes():
    # an update between two rb_iter_next calls: the reference would render the next pass with the NEW scene on
    # top of the accumulation so far (lib.rs:200-203 after gpu_wrapper.rs:116-300); the pass already running
    # with the old scene on the other slot must not be delivered
    a = scenes.cornell(32, 20, 4, 3)
    b = scenes.cornell(32, 20, 4, 3, seed=5)
    out = {}
    for mode, kw in (("ahead", dict()), ("plain", dict(no_run_ahead=True))):
        e = Engine.new(RenderConfig.from_scene(a), **kw)
        it = e.frame_iterator(RenderConfig.from_scene(a))
        f0 = it.next().pixels.copy()
        e.update(RenderConfig(uniforms=Change.update(b.uniforms), spheres=Change.update(b.spheres)))
        rest = [f.pixels.copy() for f in it]
        out[mode] = [f0] + rest
        e.close()
    assert len(out["ahead"]) == 4
    for x, y in zip(out["ahead"], out["plain"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,stripe_rows", [(2, 16), (3, 1), (8, 4)])
def test_multi_device_handle_assembles_the_single_device_frame(n, stripe_rows):
    # rb_create_multi with n shards.  One GPU is all a test box has: the shards run on device 0 and the stripes
    # move with the peer-copy transport (RCCL refuses one device twice); the sharding, the gather buffer layout,
    # the de-interleave kernel and the read-back are the code the RCCL transport uses too.
    s = scenes.feature_scene(50, 37, 3, 4)
    rc = RenderConfig.from_scene(s)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    e = Engine.new(rc, devices=[0] * n, stripe_rows=stripe_rows, gather_peer_copy=True, stats=True)
    f = e.render(rc)
    assert f.pixels.shape == (37, 50, 4)
    assert np.array_equal(f.pixels, o_rgba)
    assert np.array_equal(e.read_accumulation().view(np.uint32), o_acc.view(np.uint32))
    st = e.stats()
    assert st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"]
    # the progressive iterator through the same handle
    frames = [fr.pixels.copy() for fr in e.frame_iterator(RenderConfig.from_scene(s, create=False))]
    assert len(frames) == 3 and np.array_equal(frames[-1], o_rgba)
    assert np.array_equal(frames[0], _oracle.render(s, 0, 1)[2])
    # a new resolution through the same handle
    t = s.with_params(width=33, height=50, spp=2)
    assert np.array_equal(e.render(RenderConfig(uniforms=Change.update(t.uniforms))).pixels, _oracle.render(t)[2])
    e.close()


def test_multi_device_handle_with_a_mesh_and_the_stream_kernels():
    s = scenes.mesh_scene(24, 24, 96, 64, 2, 4)
    rc = RenderConfig.from_scene(s)
    single = Engine.new(rc)
    want = single.render(rc).pixels.copy()
    single.close()
    e = Engine.new(rc, devices=[0, 0, 0, 0], gather_peer_copy=True)
    assert np.array_equal(e.render(rc).pixels, want)
    e.close()


def test_rccl_communicator_of_one_rank():
    # one process per device (rb_comm_unique_id / rb_comm_init_rank): with the one GPU a test box has this is a
    # communicator of a single rank -- librccl is found and loaded, the id made, the communicator created, the
    # render goes through the gather path's root side and the communicator is destroyed with the engine
    s = scenes.cornell(24, 16, 2, 3)
    rc = RenderConfig.from_scene(s)
    cid = Engine.comm_unique_id()
    assert len(cid) == abi.COMM_ID_BYTES and any(cid)
    e = Engine.new(rc)
    e.comm_init_rank(cid, 0, 1)
    assert np.array_equal(e.render(rc).pixels, _oracle.render(s)[2])
    info = e.comm_info()   # ncclCommCount / ncclCommUserRank of the live communicator, and the root's share of the exchange
    assert info["rccl_ranks"] == 1 and info["rccl_rank"] == 0 and info["gather_ms"] >= 0.0
    with pytest.raises(RenderError):
        e.comm_init_rank(cid, 1, 2)   # does not match the engine's shard geometry
    e.close()


def test_single_device_group_needs_no_communicator():
    s = scenes.cornell(24, 16, 2, 3)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, devices=[0])
    assert np.array_equal(e.render(rc).pixels, _oracle.render(s)[2])
    assert e.comm_info()["rccl_ranks"] == 0   # nothing goes through RCCL
    e.close()


def _graze_scene(seed, coarse):
    """A mesh and an eye in the plane of one of its triangles (tools/fuzz_parity.py::graze)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(seed)
    if coarse:   # a few hundred large triangles in every orientation: wide normal cones, huge L^2
        tris = []
        for _ in range(400):
            c = rng.uniform(-6, 6, 3)
            tris.append(tuple(tuple(map(float, c + rng.uniform(-4, 4, 3))) for _ in range(3)))
        groups = [(scenes.material(**scenes.KHAKI), tris), (scenes.material(**scenes.LIGHT), scenes._light_quad())]
        u = scenes.make_uniforms(48, 32, 2, 4, cam_pos=(0, 0, 9), cam_dir=(0, 0, -1), sky=(0.5, 0.7, 1.0))
        sc = scenes._finish("coarse", u, np.zeros(0, abi.SPHERE), np.zeros(0, abi.POINT_LIGHT), groups)
    else:        # the smooth terrain + blob of C3 at reduced size: narrow cones, small triangles
        sc = scenes.mesh_scene(40, 40, 48, 32, 2, 4, seed=seed)
    fz.graze(sc, rng)
    return sc


@pytest.mark.parametrize("coarse", [False, True], ids=["smooth-mesh", "coarse-soup"])
@pytest.mark.parametrize("builder", ["chunk", "host-sah", "device-ploc", "device-lbvh"])
def test_own_tree_with_rays_in_the_plane_of_a_triangle(coarse, builder):
    # the culling margins of the chunked walk and of the library's tree must cover hits the reference reports from a
    # near-zero determinant (|a| down to 1e-6): eyes in the plane of a triangle, looking along it, 12 scenes each
    kw = dict(chunk_walk=True) if builder == "chunk" else dict(host_bvh=True) if builder == "host-sah" else \
        dict(device_bvh=True, device_lbvh=(builder == "device-lbvh"))
    for seed in range(12):
        sc = _graze_scene(100 + seed, coarse)
        o_acc, _, o_rgba, o_st = _oracle.render(sc)
        rc = RenderConfig.from_scene(sc)
        e = Engine.new(rc, **kw)
        f = e.render(rc)
        acc, st = e.read_accumulation(), e.stats()
        assert e.last_kernel_name() == ("k_trace_chunk" if builder == "chunk" else "k_trace_fast")
        e.close()
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (seed, coarse, builder)
        assert np.array_equal(f.pixels, o_rgba) and st["segments"] == o_st["segments"], (seed, coarse, builder)


def test_pinned_frame_buffers_receive_the_same_frames():
    from renderbaby_amd.engine import PinnedFrame
    s = scenes.feature_scene(64, 40, 5, 4)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    want = [f.pixels.copy() for f in e.frame_iterator(rc)]
    pf = PinnedFrame(64, 40)
    cfg, keep = RenderConfig.from_scene(s, create=False).to_c()
    e._check(e._lib.rb_iter_begin(e._h, C.byref(cfg)))
    del keep
    for k in range(5):
        e._check(e._lib.rb_iter_next(e._h, pf.array.ctypes.data))
        assert np.array_equal(pf.array, want[k]), k
    e._check(e._lib.rb_render(e._h, pf.array.ctypes.data))
    assert np.array_equal(pf.array, want[-1])
    e.close()
    pf.free()


def test_destroy_with_a_pass_still_running_ahead():
    # rb_destroy must wait for whatever the engine's streams still hold -- here the pass the iterator started
    # ahead of the caller -- before events, streams and buffers go (r01 recorded a host segmentation fault after
    # ~50 000 create / destroy cycles that was never explained; this is the one ordering the churn did not cover)
    s = scenes.cornell(96, 64, 8, 6)
    rc = RenderConfig.from_scene(s)
    want = _oracle.render(s, 0, 2)[2]
    for i in range(150):
        e = Engine.new(rc)
        it = e.frame_iterator(rc)
        it.next()
        f = it.next()       # the third pass is in flight on the second slot now
        if i % 50 == 0:
            assert np.array_equal(f.pixels, want)
        e.close()


def test_getters_from_another_thread_while_rendering():
    # rb_get_size / rb_last_error used to read engine state without the engine's lock (rb_get_size even wrote the
    # error string of a const engine): a GUI thread polling them next to the render thread raced on a std::string
    import threading
    s = scenes.cornell(64, 48, 3, 4)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    e.update(rc)
    stop = threading.Event()
    seen = []

    def poll():
        while not stop.is_set():
            seen.append(e.size())
            e._lib.rb_last_error(e._h)
            e._lib.rb_iter_has_next(e._h)

    t = threading.Thread(target=poll)
    t.start()
    try:
        bad = RenderConfig(uniforms=Change.update(s.uniforms), uvs=Change.update(np.zeros(3, np.float32)))
        for _ in range(60):
            f = e.render(RenderConfig.from_scene(s, create=False))
            with pytest.raises(RenderError):
                e.update(bad)      # keeps rewriting the error string under the poller
    finally:
        stop.set()
        t.join()
    assert np.array_equal(f.pixels, _oracle.render(s)[2]) and set(seen) == {(64, 48)}
    e.close()


def test_multi_device_handle_into_page_locked_memory_and_running_ahead():
    # r03: the exchange and the read-back of a sharded engine run on its second stream; into page-locked memory the frame
    # leaves by DMA on that stream; every part runs a pass ahead.  Frame k must still be the running average of samples 0..k.
    from renderbaby_amd.engine import PinnedFrame
    s = scenes.feature_scene(50, 37, 5, 4)
    rc = RenderConfig.from_scene(s)
    want = [_oracle.render(s, 0, k + 1)[2] for k in range(5)]
    for ahead in (True, False):
        e = Engine.new(rc, devices=[0, 0, 0], stripe_rows=4, gather_peer_copy=True, no_run_ahead=not ahead)
        pf = PinnedFrame(50, 37)
        cfg, keep = rc.to_c()
        e._check(e._lib.rb_iter_begin(e._h, C.byref(cfg)))
        del keep
        k = 0
        while e._lib.rb_iter_has_next(e._h):
            e._check(e._lib.rb_iter_next(e._h, pf.array.ctypes.data))
            assert np.array_equal(pf.array, want[k]), (ahead, k)
            k += 1
        assert k == 5
        # a blocking render into the same page-locked buffer
        e._check(e._lib.rb_render(e._h, pf.array.ctypes.data))
        assert np.array_equal(pf.array, want[-1])
        e.close()
        pf.free()
