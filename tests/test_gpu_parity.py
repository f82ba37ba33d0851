"""HIP path vs the CPU oracle on identical seeded inputs, through the C ABI.

Bar: bit-exact on the f32 accumulation and on RGBA8 (integer/byte work and,
because both sides perform the same IEEE binary32 operations in the same
order, the floating-point work too).  Work counters must agree exactly.
"""
import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, abi, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu

KERNELS = [abi.KERNEL_PIXEL, abi.KERNEL_QUEUE, abi.KERNEL_STREAM]


_WALK_KW = ("reference_walk", "fast_bvh", "host_bvh", "device_bvh", "device_lbvh", "own_tree", "chunk_walk")


def _ref(kw):
    """The tests that compare work counters with the oracle's (or name the reference-walk kernels) pin the
    reference walk; the chunked walk -- the default for multi-node meshes -- and the library's own tree have their
    own tests."""
    return kw if any(k in kw for k in _WALK_KW) else dict(kw, reference_walk=True)


def _hip(scene, kernel, stats=True, **kw):
    kw = _ref(kw)
    rc = RenderConfig.from_scene(scene)
    eng = Engine.new(rc, kernel=kernel, stats=stats, **kw)
    frame = eng.render(rc)
    acc = eng.read_accumulation()
    st = eng.stats()
    eng.close()
    return frame, acc, st


def _assert_same(scene, kernel):
    o_acc, _, o_rgba, o_st = _oracle.render(scene)
    frame, acc, st = _hip(scene, kernel)
    # shader x order for the accumulation, mirrored x for the frame
    assert acc.shape == o_acc.shape
    bad = np.argwhere(acc.view(np.uint32) != o_acc.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} accumulation words differ, first {bad[:5]}: {acc[tuple(bad[0][:2])]} vs {o_acc[tuple(bad[0][:2])]}"
    assert np.array_equal(frame.pixels, o_rgba)
    for k in ("segments", "paths", "nodes_popped", "tris_tested", "spheres_tested", "lights_tested", "mesh_hits"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])


@pytest.mark.parametrize("kernel", KERNELS)
def test_cornell_bit_exact(kernel):
    _assert_same(scenes.cornell(96, 64, 8, 4), kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_cornell_depth8_odd_size(kernel):
    _assert_same(scenes.cornell(61, 37, 5, 8), kernel)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("color_hash", [0, 1])
def test_feature_scene_bit_exact(kernel, color_hash):
    # ground + checkerboard, textured mesh and sphere, metal/mirror, two point lights, sky
    _assert_same(scenes.feature_scene(48, 32, 6, 5, color_hash=color_hash), kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_mesh_bvh_bit_exact(kernel):
    s = scenes.mesh_scene(24, 24, 64, 40, 4, 5, seed=7)  # 2304 triangles, multi-level BVH
    assert len(s.bvh_nodes) > 1
    _assert_same(s, kernel)


@pytest.mark.parametrize("grid,lds_mode,name", [(12, 0, "k_trace_bvh_lds"), (12, 1, "k_trace_bvh"),
                                                (24, 0, "k_trace_bvh_lds"), (40, 0, "k_trace_bvh")])
def test_leaf_stepped_walk_from_lds_and_from_l2(grid, lds_mode, name):
    # meshes that fit in LDS next to the traversal stacks are staged there; larger ones, or
    # lds_mode = 1, walk the same tree through L1/L2.  Same bits either way.
    s = scenes.mesh_scene(grid, grid, 64, 40, 3, 5, seed=grid)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc, stats=True, lds_mode=lds_mode, reference_walk=True)
    frame = eng.render(rc)
    acc, st = eng.read_accumulation(), eng.stats()
    assert eng.last_kernel_name() == name
    eng.close()
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32))
    assert np.array_equal(frame.pixels, o_rgba)
    for k in ("segments", "nodes_popped", "tris_tested", "mesh_hits"):
        assert st[k] == o_st[k], k


@pytest.mark.parametrize("kernel", KERNELS)
def test_sky_only_known_answer(kernel):
    s = scenes.sky_only()
    frame, acc, st = _hip(s, kernel)
    assert np.all(frame.pixels == np.array([147, 164, 181, 255], dtype=np.uint8))
    assert st["segments"] == s.width * s.height


def test_device_math_is_ieee():
    import ctypes as C
    from renderbaby_amd._lib import load
    rng = np.random.default_rng(1)
    n = 1 << 16
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)
    b = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 6, n)).astype(np.float32)
    b[b == 0] = 1.0
    out = np.zeros((8, n), dtype=np.float32)
    assert load().rb_debug_math(a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
    with np.errstate(all="ignore"):
        assert np.array_equal(out[0].view(np.uint32), (a / b).view(np.uint32)), "f32 division is not correctly rounded"
        assert np.array_equal(out[1].view(np.uint32), np.sqrt(np.abs(a)).view(np.uint32)), "f32 sqrt is not correctly rounded"
        z = (a * b).astype(np.float32)
        ln = np.sqrt(((a * a + b * b).astype(np.float32) + z * z).astype(np.float32)).astype(np.float32)
        for k, comp in enumerate((a, b, z)):
            exp = (comp / ln).astype(np.float32)
            ok = (out[2 + k].view(np.uint32) == exp.view(np.uint32)) | (np.isnan(exp) & np.isnan(out[2 + k]))
            assert ok.all(), f"normalize component {k}"
        exp = (a.view(np.uint32).astype(np.float32) / np.float32(4294967296.0)).astype(np.float32)
        assert np.array_equal(out[5].view(np.uint32), exp.view(np.uint32)), "u32->f32"
        exp = ((a * b + b * b).astype(np.float32) + a * a).astype(np.float32)
        assert np.array_equal(out[7].view(np.uint32), exp.view(np.uint32)), "dot: contraction or reassociation"


def test_stream_kernel_chunking_is_bit_identical():
    # the (pixel, sample) stream is cut into launch chunks by passes_per_launch / the
    # colour-buffer budget; any cut must give the same frame
    s = scenes.feature_scene(40, 24, 7, 5)
    o_acc, _, o_rgba, _ = _oracle.render(s)
    for ppl in (0, 1, 3, 7):
        frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=False, passes_per_launch=ppl)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), ppl
        assert np.array_equal(frame.pixels, o_rgba), ppl
    frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=False, color_budget_mib=1)
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32))


@pytest.mark.parametrize("kernel", KERNELS)
def test_progressive_iterator_matches_blocking_render(kernel):
    # frame k of the iterator = running average of samples 0..k (lib.rs:169-228)
    s = scenes.cornell(40, 24, 4, 4)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc, kernel=kernel)
    it = eng.frame_iterator(rc)
    frames = []
    while it.has_next():
        frames.append(it.next().pixels.copy())
    assert len(frames) == 4
    with pytest.raises(Exception) as ei:
        it.next()
    assert "No more frames available" in str(ei.value)
    for k in range(4):
        _, _, o_rgba, _ = _oracle.render(s, 0, k + 1)
        assert np.array_equal(frames[k], o_rgba), k
    # a new iterator restarts at pass 0 with a cleared accumulation (lib.rs:91,181-192)
    it2 = eng.frame_iterator(RenderConfig.from_scene(s, create=False))
    assert np.array_equal(it2.next().pixels, frames[0])
    eng.close()


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("world,stripe_rows", [(2, 16), (3, 4), (8, 1)])
def test_sharded_engines_reassemble_the_single_gpu_frame(kernel, world, stripe_rows):
    import torch
    from renderbaby_amd import dist as rdist
    s = scenes.cornell(48, 37, 3, 4)
    rc = RenderConfig.from_scene(s)
    full = Engine.new(rc, kernel=kernel).render(rc).pixels
    parts = []
    for r in range(world):
        eng = Engine.new(rc, kernel=kernel, shard_rank=r, shard_count=world, stripe_rows=stripe_rows)
        parts.append(torch.from_numpy(eng.render(rc).pixels.copy()))
        eng.close()
    frame = rdist.assemble(parts, s.height, stripe_rows).numpy()
    assert np.array_equal(frame, full)


def _scene_for(which):
    if which == "mesh":      # multi-node tree: k_trace_chunk / k_trace_bvh / k_trace_bvh_lds / k_trace_fast
        return scenes.mesh_scene(24, 24, 48, 36, 5, 5, seed=3)
    return scenes.spheres_scene(n=3000, width=48, height=36, spp=5, max_depth=5, extent=12.0)   # k_trace_sph


@pytest.mark.parametrize("which,kw,kernel_name", [
    ("mesh", dict(), "k_trace_chunk"), ("mesh", dict(chunk_walk=True), "k_trace_chunk"),
    ("mesh", dict(reference_walk=True), "k_trace_bvh_lds"), ("mesh", dict(lds_mode=1, reference_walk=True), "k_trace_bvh"),
    ("mesh", dict(fast_bvh=True), "k_trace_fast"), ("mesh", dict(own_tree=True), "k_trace_fast"), ("mesh", dict(device_bvh=True), "k_trace_fast"),
    ("mesh", dict(device_lbvh=True), "k_trace_fast"),
    ("spheres", dict(), "k_trace_sph"), ("spheres", dict(no_leaf_stepping=True), "k_trace")])
def test_stepped_kernels_under_chunking_sharding_and_the_iterator(which, kw, kernel_name):
    # the persistent, stepped kernels keep per-lane walk state across passes; cutting the work into
    # launch chunks, row stripes or one-sample frames must not change a bit
    import torch
    from renderbaby_amd import dist as rdist
    s = _scene_for(which)
    rc = RenderConfig.from_scene(s)
    o_acc, _, o_rgba, _ = _oracle.render(s)
    eng = Engine.new(rc, **kw)
    full = eng.render(rc)
    assert eng.last_kernel_name() == kernel_name
    assert np.array_equal(eng.read_accumulation().view(np.uint32), o_acc.view(np.uint32))
    assert np.array_equal(full.pixels, o_rgba)
    # progressive: frame k = running average of samples 0..k
    it = eng.frame_iterator(RenderConfig.from_scene(s, create=False))
    last = None
    for k in range(s.total_samples):
        last = it.next().pixels.copy()
        if k == 1:
            assert np.array_equal(last, _oracle.render(s, 0, 2)[2])
    assert np.array_equal(last, o_rgba)
    eng.close()
    # launch chunking
    eng = Engine.new(rc, passes_per_launch=2, **kw)
    assert np.array_equal(eng.render(rc).pixels, o_rgba)
    eng.close()
    # row-stripe sharding
    parts = []
    for r in range(3):
        e = Engine.new(rc, shard_rank=r, shard_count=3, stripe_rows=4, **kw)
        parts.append(torch.from_numpy(e.render(rc).pixels.copy()))
        e.close()
    assert np.array_equal(rdist.assemble(parts, s.height, 4).numpy(), o_rgba)


@pytest.mark.parametrize("n,extent,size", [(200, 6.0, 64), (10_000, 25.0, 192)])
def test_sphere_bvh_matches_the_linear_scan(n, extent, size):
    # BASELINE C4 scaled down: the library's sphere acceleration structure must give
    # exactly the linear scan's winner (oracle = reference algorithm, O(N) per segment)
    s = scenes.spheres_scene(n=n, width=size, height=size, spp=2, max_depth=5, extent=extent)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    # the tree from either builder (device: Morton order + LBVH; host: median splits), walked by the pooled kernel and per
    # segment, and the GPU's own linear scan
    for kw in (dict(), dict(sphere_tree="host"), dict(sphere_tree="device"), dict(sphere_tree="device", no_leaf_stepping=True),
               dict(sphere_tree="host", kernel=abi.KERNEL_QUEUE), dict(no_sphere_bvh=True)):
        kernel = kw.pop("kernel", abi.KERNEL_STREAM)
        frame, acc, st = _hip(s, kernel, stats=False, **kw)
        bad = np.argwhere(acc.view(np.uint32) != o_acc.view(np.uint32))
        assert len(bad) == 0, (kw, len(bad), bad[:4])
        assert np.array_equal(frame.pixels, o_rgba)
        assert st["segments"] == o_st["segments"]
    # the hit rate must be meaningful for the comparison to mean anything
    assert (o_acc[..., :3].sum(-1) > 0).mean() > 0.5


@pytest.mark.parametrize("color_hash", [0, 1])
@pytest.mark.parametrize("grid,w,h,spp", [(24, 64, 40, 4), (112, 96, 54, 2)])
def test_fast_bvh_reproduces_the_reference_walk(grid, w, h, spp, color_hash):
    # RB_FLAG_FAST_BVH: own SAH tree + culling, winner re-validated against the reference tree
    s = scenes.mesh_scene(grid, grid, w, h, spp, 5, seed=7)
    if color_hash:
        u = s.uniforms.copy()
        u["color_hash_enabled"] = 1
        s = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=True, fast_bvh=True)
    bad = np.argwhere(acc.view(np.uint32) != o_acc.view(np.uint32))
    assert len(bad) == 0, (len(bad), bad[:4])
    assert np.array_equal(frame.pixels, o_rgba)
    assert st["segments"] == o_st["segments"]
    assert st["tris_tested"] < o_st["tris_tested"]  # it must actually be the fast walk


@pytest.mark.parametrize("grid,w,h,spp,lbvh,builder", [
    (24, 64, 40, 4, False, "device-ploc"), (112, 96, 54, 2, False, "device-ploc"), (24, 64, 40, 4, True, "device-lbvh"),
    (112, 96, 54, 2, True, "device-lbvh"), (12, 64, 40, 2, False, "host-sah")])
def test_device_built_tree_reproduces_the_reference_walk(grid, w, h, spp, lbvh, builder):
    # RB_FLAG_DEVICE_BVH: the fast walk's tree built on the GPU (Morton order + locally-ordered
    # clustering, or plain LBVH with RB_FLAG_DEVICE_LBVH).  The tree only steers the walk, so the frame
    # is still the reference walk's, bit for bit.  Below 1024 triangles the host builder is used.
    s = scenes.mesh_scene(grid, grid, w, h, spp, 5, seed=7)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc, stats=True, device_bvh=True, device_lbvh=lbvh)
    frame = eng.render(rc)
    acc, st = eng.read_accumulation(), eng.stats()
    name, ms = eng.fast_bvh_builder()
    eng.close()
    assert name == builder and ms > 0.0
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    bad = np.argwhere(acc.view(np.uint32) != o_acc.view(np.uint32))
    assert len(bad) == 0, (len(bad), bad[:4])
    assert np.array_equal(frame.pixels, o_rgba)
    assert st["segments"] == o_st["segments"]
    assert st["tris_tested"] < o_st["tris_tested"]


def test_threaded_host_tree_build_is_deterministic(monkeypatch):
    # the host SAH builder forks big subtrees onto threads; the tree must be the one the sequential
    # build makes: same frame, and the same node / triangle counters in the walk
    s = scenes.mesh_scene(160, 160, 64, 40, 2, 5, seed=5)       # 102 402 triangles: forks at several levels
    out = []
    for seq in ("0", "1"):
        monkeypatch.setenv("RB_HOST_BUILD_SEQUENTIAL", seq)
        frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=True, fast_bvh=True)
        out.append((acc, st))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    for k in ("segments", "nodes_popped", "tris_tested", "mesh_hits"):
        assert out[0][1][k] == out[1][1][k], k


def test_fast_reciprocal_and_sqrt_are_exhaustively_exact():
    # the kernels replace the 12-instruction IEEE divide by rcp + Newton/FMA steps where the
    # operands allow; correctness is a property of the significand, so it is checked for ALL
    # 2^23 significands (both signs) at exponents across the admitted range
    from renderbaby_amd._lib import load
    lib = load()
    for expo in (27, 28, 67, 126, 127, 128, 187, 226):
        out = np.zeros(16, np.uint32)
        assert lib.rb_debug_rcp_exhaustive(expo, out.ctypes.data) == 0
        assert out[0] == 0, (expo, [hex(int(x)) for x in out[1:4]])
    for expo in (67, 68, 126, 127, 128, 186, 0, 1, 254):  # sqrt: even/odd exponents + fallback ranges
        out = np.zeros(16, np.uint32)
        assert lib.rb_debug_rcp_exhaustive(0x100 | expo, out.ctypes.data) == 0
        assert out[0] == 0, ("sqrt", expo, [hex(int(x)) for x in out[1:4]])


def test_fast_division_on_sampled_significand_pairs():
    # the full 2^46-pair walk takes 67 s (profiles/r01_div_exhaustive_2p46.log: 0 mismatches);
    # here: 2^12 denominators x 2^23 numerators for several exponent pairs
    from renderbaby_amd._lib import load
    lib = load()
    for ea, eb, b0 in ((127, 127, 0x123000), (90, 150, 0x7FF000), (160, 100, 0x000000), (127, 68, 0x400000),
                       (30, 127, 0x2AB000)):
        out = np.zeros(16, np.uint64)
        assert lib.rb_debug_div_exhaustive(b0, 1 << 12, ea, eb, 0, 1 << 23, out.ctypes.data) == 0
        assert out[0] == 0, (ea, eb, [hex(int(x)) for x in out[1:5]])


@pytest.mark.parametrize("w,h", [(1, 1), (1, 5), (5, 1), (2, 2), (17, 3)])
def test_degenerate_frame_sizes(w, h):
    # width-1 / height-1 divisors of the primary ray become 0 for one-pixel axes (shader.wgsl:699-700):
    # inf/NaN directions must flow through exactly as in the oracle (every test fails => sky)
    s = scenes.feature_scene(w, h, 3, 4)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    for kernel in KERNELS:
        frame, acc, st = _hip(s, kernel)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (kernel, acc, o_acc)
        assert np.array_equal(frame.pixels, o_rgba)
        assert st["segments"] == o_st["segments"]


@pytest.mark.parametrize("which,kw", [("mesh", dict(reference_walk=True)), ("mesh", dict(lds_mode=1, reference_walk=True)),
                                      ("mesh", dict(chunk_walk=True)), ("mesh", dict(fast_bvh=True)), ("mesh", dict(device_bvh=True)), ("mesh", dict(own_tree=True)),
                                      ("spheres", dict())])
@pytest.mark.parametrize("w,h,spp,depth", [(1, 1, 2, 3), (3, 1, 1, 1), (2, 7, 2, 0), (9, 5, 1, 16)])
def test_stepped_kernels_on_degenerate_frames(which, kw, w, h, spp, depth):
    # one-pixel axes (inf / NaN primary directions), a single sample, depth 0 and a depth beyond any
    # path through the stepped kernels: same bits and same counters as the oracle
    base = _scene_for(which)
    s = base.with_params(width=w, height=h, spp=spp, max_depth=depth)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=True, **kw)
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32))
    assert np.array_equal(frame.pixels, o_rgba)
    assert st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"]


def test_zero_samples_and_zero_depth():
    s = scenes.cornell(8, 4, 0, 4)  # total_samples = 0: no passes; the reference returns the cleared buffers
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc)
    f = eng.render(rc)
    assert np.all(f.pixels == 0)
    it = eng.frame_iterator(RenderConfig.from_scene(s, create=False))
    assert not it.has_next()
    eng.close()
    s = scenes.cornell(8, 4, 2, 0)  # max_depth = 0: every path is black
    frame, acc, st = _hip(s, abi.KERNEL_STREAM)
    assert np.all(frame.pixels[..., :3] == 0) and np.all(frame.pixels[..., 3] == 255)
    assert st["segments"] == 0 and st["paths"] == 8 * 4 * 2


def test_largest_frame_of_the_baseline_configs():
    # 4096 x 4096 (C4's frame): addressing of a 16.8 M-pixel frame, sky only, 1 spp
    s = scenes.sky_only(4096, 4096, 1)
    frame, acc, st = _hip(s, abi.KERNEL_STREAM, stats=False)
    assert frame.pixels.shape == (4096, 4096, 4)
    assert np.all(frame.pixels.reshape(-1, 4) == np.array([147, 164, 181, 255], dtype=np.uint8))
    assert st["paths"] == 4096 * 4096


@pytest.mark.parametrize("n,extent,camscale", [(20000, 100.0, 8.0), (50000, 30.0, 40.0), (5000, 400.0, 1.0)])
def test_sphere_bvh_under_poor_conditioning(n, extent, camscale):
    # far cameras / tiny spheres: the reference's discriminant is mostly rounding noise there, and the
    # sphere tree must still return the linear scan's winner (here the GPU's own linear scan)
    s = scenes.spheres_scene(n=n, width=160, height=160, spp=2, max_depth=5, extent=extent)
    u = s.uniforms.copy()
    u["camera"]["pos"] = np.array([0, 30, 120], np.float32) * np.float32(camscale)
    u["camera"]["dir"] = -u["camera"]["pos"]
    s = scenes.Scene(u, s.spheres, s.lights, s.meshes, s.bvh_nodes, s.bvh_indices, s.bvh_triangles, s.uvs)
    _, a, st = _hip(s, abi.KERNEL_STREAM, stats=False)
    _, b, st2 = _hip(s, abi.KERNEL_STREAM, stats=False, no_sphere_bvh=True)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert st["segments"] == st2["segments"]


@pytest.mark.parametrize("kw,builder", [(dict(), ""), (dict(fast_bvh=True), "host-sah"), (dict(device_bvh=True), "device-ploc"),
                                        (dict(device_lbvh=True), "device-lbvh")])
def test_coincident_triangles(kw, builder):
    # 2 500 copies of one triangle (identical boxes, identical Morton codes, identical t for every ray)
    # in front of a wall: the builders must terminate on the run of duplicates, and the winner among the
    # copies is the first one in the reference's visit order
    tri = ((-1.0, 0.5, -3.0), (1.0, 0.5, -3.0), (0.0, 2.0, -3.2))
    wall = scenes._quad((-3, 0, -5), (3, 0, -5), (3, 4, -5), (-3, 4, -5))
    groups = [(scenes.material(**scenes.RED), [tri] * 2500), (scenes.material(**scenes.KHAKI), wall),
              (scenes.material(**scenes.LIGHT), scenes._quad((-1, 3.9, -4), (1, 3.9, -4), (1, 3.9, -2), (-1, 3.9, -2)))]
    u = scenes.make_uniforms(40, 30, 2, 4, cam_pos=(0, 1.5, 2), cam_dir=(0, 0, -1), ground_enabled=1, ground_height=0.0,
                             sky=(0.3, 0.4, 0.5), color_hash=1)
    s = scenes._finish("coincident", u, np.zeros(0, abi.SPHERE), np.zeros(0, abi.POINT_LIGHT), groups)
    assert len(s.bvh_nodes) > 1
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    rc = RenderConfig.from_scene(s)
    eng = Engine.new(rc, **kw)
    frame = eng.render(rc)
    acc = eng.read_accumulation()
    assert eng.fast_bvh_builder()[0] == builder
    eng.close()
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)) and np.array_equal(frame.pixels, o_rgba)


# ---- trees the reference's builder would never make, but the boundary accepts (RenderConfig.bvh_nodes is the caller's):
# the walk must reproduce what shader.wgsl:282-392 does WITH THAT TREE, whatever its quality
def _py_tree(tris, max_leaf, rng=None, lopsided=0):
    """A median-split tree in the reference's node layout with leaves of up to `max_leaf` triangles; `lopsided` = k cuts
    every range 1 : k - 1 instead of in halves (a deep, unbalanced tree)."""
    cent = (tris["v0"] + tris["v1"] + tris["v2"]) / np.float32(3.0)
    lo = np.minimum(np.minimum(tris["v0"], tris["v1"]), tris["v2"])
    hi = np.maximum(np.maximum(tris["v0"], tris["v1"]), tris["v2"])
    nodes, order = [], []

    def build(ids):
        me = len(nodes)
        nodes.append(None)
        bmin, bmax = lo[ids].min(axis=0), hi[ids].max(axis=0)
        if len(ids) <= max_leaf:
            nodes[me] = (bmin, bmax, 0, 0, len(order), len(ids))
            order.extend(ids.tolist())
            return me
        ext = cent[ids].max(axis=0) - cent[ids].min(axis=0)
        ids = ids[np.argsort(cent[ids, int(np.argmax(ext))], kind="stable")]
        mid = max(1, len(ids) // lopsided) if lopsided else len(ids) // 2
        left, right = build(ids[:mid]), build(ids[mid:])
        nodes[me] = (bmin, bmax, left, right, 0, 0)
        return me
    build(np.arange(len(tris)))
    out = np.zeros(len(nodes), dtype=abi.BVH_NODE)
    for i, (bmin, bmax, l, r, first, count) in enumerate(nodes):
        out[i]["aabb_min"], out[i]["aabb_max"] = bmin, bmax
        out[i]["left"], out[i]["right"], out[i]["first_primitive"], out[i]["primitive_count"] = l, r, first, count
    return out, np.asarray(order, dtype=np.uint32)


def _with_tree(s, nodes, indices):
    u = s.uniforms.copy()
    u["bvh_node_count"] = len(nodes)
    return scenes.Scene(u, s.spheres, s.lights, s.meshes, nodes, indices, s.bvh_triangles, s.uvs, s.textures)


@pytest.mark.parametrize("shape", ["leaves-of-300", "leaves-of-5", "lopsided", "too-deep-for-the-chunk-stack", "boxes-shrunk", "boxes-grown", "boxes-shifted"])
@pytest.mark.parametrize("kw", [dict(), dict(reference_walk=True), dict(fast_bvh=True)], ids=["chunk", "reference", "own-tree"])
def test_caller_made_trees(shape, kw):
    base = scenes.mesh_scene(20, 20, 72, 48, 3, 5, seed=31)   # 1 600 triangles + light quad
    tris = base.bvh_triangles
    rng = np.random.default_rng(5)
    if shape == "leaves-of-300":
        nodes, idx = _py_tree(tris, 300)
    elif shape == "leaves-of-5":
        nodes, idx = _py_tree(tris, 5)
    elif shape == "lopsided":
        nodes, idx = _py_tree(tris, 40, lopsided=4)
    elif shape == "too-deep-for-the-chunk-stack":   # ~28 levels: more than the 32-entry LDS stack leaves room for below the caller's leaves
        nodes, idx = _py_tree(tris, 40, lopsided=8)
    else:   # the reference builder's tree with every box made wrong in its own way: the boxes no longer contain their triangles
        nodes, idx = base.bvh_nodes.copy(), base.bvh_indices.copy()
        c = (nodes["aabb_min"] + nodes["aabb_max"]) * np.float32(0.5)
        h = (nodes["aabb_max"] - nodes["aabb_min"]) * np.float32(0.5)
        if shape == "boxes-shrunk":
            h = h * rng.uniform(0.3, 0.9, h.shape).astype(np.float32)
        elif shape == "boxes-grown":
            h = h * rng.uniform(1.0, 3.0, h.shape).astype(np.float32)
        else:
            c = c + h * rng.uniform(-0.8, 0.8, h.shape).astype(np.float32)
        nodes["aabb_min"], nodes["aabb_max"] = c - h, c + h
        nodes["aabb_min"][0], nodes["aabb_max"][0] = base.bvh_nodes["aabb_min"][0], base.bvh_nodes["aabb_max"][0]   # keep the root: something must be seen
    s = _with_tree(base, nodes, idx)
    assert len(s.bvh_nodes) > 1
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    assert o_st["mesh_hits"] > 0
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, **kw)
    f = e.render(rc)
    acc, st, name = e.read_accumulation(), e.stats(), e.last_kernel_name()
    e.close()
    if not kw:   # the default: the chunked walk, or -- when its tree would not fit the stack -- the reference walk
        assert name == ("k_trace_bvh" if shape == "too-deep-for-the-chunk-stack" else "k_trace_chunk")
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (shape, name)
    assert np.array_equal(f.pixels, o_rgba) and st["segments"] == o_st["segments"], (shape, name)
