"""BASELINE.json sizes.  C1 is compared with the oracle in full (the GPU box's host has
enough cores); at C2's size the oracle would need minutes, so the full-size checks are
size-independent properties: determinism, launch-chunk invariance, row-shard invariance,
exact work counts, and oracle agreement on a sub-rectangle."""
import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, abi, scenes
from renderbaby_amd import dist as rdist
from tests import _oracle

pytestmark = pytest.mark.gpu


def test_c1_full_config_is_bit_exact():
    s = scenes.cornell_c1()  # 512x512, 64 spp, depth 4
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    f = e.render(rc)
    acc = e.read_accumulation()
    st = e.stats()
    e.close()
    assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32))
    assert np.array_equal(f.pixels, o_rgba)
    assert st["segments"] == o_st["segments"] and st["paths"] == 512 * 512 * 64


def test_c2_size_properties():
    spp = 16  # full 1920x1080 frame, depth 8; the sample count is the only thing scaled down
    s = scenes.cornell(1920, 1080, spp, 8)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc)
    f1 = e.render(rc).pixels.copy()
    a1 = e.read_accumulation()
    st = e.stats()
    f2 = e.render(RenderConfig.from_scene(s, create=False)).pixels
    assert np.array_equal(f1, f2)  # deterministic
    assert st["paths"] == 1920 * 1080 * spp
    assert np.all(a1[..., 3] == spp) and np.all(f1[..., 3] == 255)
    e.close()
    # launch chunking (7 + 7 + 2 passes) and the per-pixel kernel give the same bits
    for kw in (dict(passes_per_launch=7), dict(kernel=abi.KERNEL_QUEUE), dict(color_budget_mib=256)):
        e = Engine.new(rc, **kw)
        assert np.array_equal(e.render(rc).pixels, f1), kw
        assert np.array_equal(e.read_accumulation().view(np.uint32), a1.view(np.uint32)), kw
        e.close()
    # oracle agreement on a sub-rectangle of the full-size frame (rows 500..507)
    o_acc, _, _, _ = _oracle.render(s, rows=(500, 508))
    assert np.array_equal(a1[500:508].view(np.uint32), o_acc[500:508].view(np.uint32))
    # 8-way row sharding reassembles the same frame
    import torch
    parts = []
    for r in range(8):
        e = Engine.new(rc, shard_rank=r, shard_count=8, stripe_rows=1)
        parts.append(torch.from_numpy(e.render(rc).pixels.copy()))
        e.close()
    assert np.array_equal(rdist.assemble(parts, 1080, 1).numpy(), f1)


def test_c3_mesh_subrectangle_matches_oracle():
    s = scenes.mesh_scene(112, 112, 1920, 1080, 2, 5)  # the 50 176-triangle mesh at full resolution
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, stats=True, reference_walk=True)
    e.render(rc)
    acc = e.read_accumulation()
    e.close()
    o_acc, _, _, _ = _oracle.render(s, rows=(700, 704))
    assert np.array_equal(acc[700:704].view(np.uint32), o_acc[700:704].view(np.uint32))


def test_c3_fast_bvh_equals_reference_walk_on_the_full_frame():
    # 16.6 M paths / 28.8 M segments through the 50 176-triangle mesh: the opt-in fast walk,
    # over the host-built or the device-built tree, must not change a single bit of the frame
    s = scenes.mesh_scene(112, 112, 1920, 1080, 8, 5)
    rc = RenderConfig.from_scene(s)
    out = {}
    for mode in ("exact", "chunk", "host-sah", "device-ploc", "device-lbvh"):
        e = Engine.new(rc, reference_walk=(mode == "exact"), host_bvh=(mode == "host-sah"), device_bvh=mode.startswith("device"),
                       device_lbvh=(mode == "device-lbvh"))
        e.render(rc)
        out[mode] = (e.read_accumulation(), e.stats()["segments"])
        assert e.fast_bvh_builder()[0] == ("" if mode in ("exact", "chunk") else mode)
        assert e.last_kernel_name() == {"exact": "k_trace_bvh", "chunk": "k_trace_chunk"}.get(mode, "k_trace_fast")
        e.close()
    for mode in ("chunk", "host-sah", "device-ploc", "device-lbvh"):
        diff = (out["exact"][0].view(np.uint32) != out[mode][0].view(np.uint32)).any(axis=-1)
        assert diff.sum() == 0, f"{mode}: {int(diff.sum())} of {diff.size} pixels differ"
        assert out["exact"][1] == out[mode][1]


def test_reference_lamp_scene_fast_walk_equals_reference_walk_at_full_size():
    # the reference's own largest fixture scene at its real 2056 x 2056 frame (4 of its 512 spp,
    # 68 M segments through 68 768 triangles, huge wall triangles next to sub-millimetre ones): the
    # opt-in walk over the host- and the device-built tree must not change a single bit of the frame
    from tests import _refscenes
    s = _refscenes.ref_lamp(spp=4)
    rc = RenderConfig.from_scene(s)
    out = {}
    for mode in ("exact", "chunk", "host-sah", "device-ploc"):
        e = Engine.new(rc, reference_walk=(mode == "exact"), host_bvh=(mode == "host-sah"), device_bvh=(mode == "device-ploc"))
        e.render(rc)
        out[mode] = (e.read_accumulation(), e.stats()["segments"])
        e.close()
    for mode in ("chunk", "host-sah", "device-ploc"):
        diff = (out["exact"][0].view(np.uint32) != out[mode][0].view(np.uint32)).any(axis=-1)
        assert diff.sum() == 0, f"{mode}: {int(diff.sum())} of {diff.size} pixels differ"
        assert out["exact"][1] == out[mode][1]
