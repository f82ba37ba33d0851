"""k_trace's colour stores and work queue (round 2): the per-row LDS ring that combines the stores
(ColorRing in rb_kernels.hip: on for reservations that are multiples of 256 items), the lanes it leaves to store
on their own (paths longer than four rows of the ring last), the waves' own first reservations, and the camera
worked out on the host -- every form against the oracle and against each other, bit for bit."""
import dataclasses

import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu


def _render(scene, **kw):
    rc = RenderConfig.from_scene(scene)
    e = Engine.new(rc, **kw)
    f = e.render(rc)
    acc, st, name = e.read_accumulation(), e.stats(), e.last_kernel_name()
    e.close()
    return acc, f.pixels, st, name


@pytest.mark.parametrize("size,spp,depth", [((256, 192), 32, 8), ((203, 77), 24, 5), ((64, 40), 64, 40)],
                         ids=["tiles", "ragged-edges", "deep-paths"])
def test_ring_and_direct_stores_agree_with_the_oracle(size, spp, depth):
    # queue_batch 256 / 512: the ring; 64 / 192: every lane stores its own 16 bytes (the r01 form).
    # "deep-paths": at depth 40 some paths outlive four rows of their wave's ring and are sent to store directly.
    s = scenes.cornell(size[0], size[1], spp, depth)
    o_acc, _, o_rgba, o_st = _oracle.render(s)
    for batch in (256, 512, 64, 192, 0, 100, 1000, 16384):   # the last three: not multiples of a row / larger than the frame's share
        acc, px, st, name = _render(s, queue_batch=batch)
        assert name == "k_trace"
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), batch
        assert np.array_equal(px, o_rgba), batch
        assert st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"], batch


def test_ring_with_fewer_items_than_one_round_of_reservations():
    # 16 x 16 pixels: 4 tiles, a handful of waves get a first reservation of their own, the rest find the queue
    # exhausted at once; the reservation is larger than the whole frame
    for spp, batch in ((1, 0), (3, 256), (5, 1024), (2, 64)):
        s = scenes.cornell(16, 16, spp, 6)
        o_acc, _, o_rgba, _ = _oracle.render(s)
        acc, px, _, _ = _render(s, queue_batch=batch)
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (spp, batch)
        assert np.array_equal(px, o_rgba), (spp, batch)


@pytest.mark.parametrize("cam", [((0, 3, 5), (0, 0, -1)), ((1.25, 2.5, 4.0), (-0.3, -0.2, -1.0)),
                                 ((0, 9, 0.5), (0.01, -1.0, -0.02)), ((-3, 1, 7), (0.9, 0.1, -2.0))],
                         ids=["c1", "oblique", "steep", "wide"])
def test_host_camera_matches_the_oracle(cam):
    # Cam (right / up / forward, fov, 1 / (w - 1), 1 / (h - 1)) is now worked out by launch_render on the host;
    # odd sizes put the two pixel-coordinate divisions on awkward denominators
    pos, direction = cam
    for w, h in ((97, 61), (128, 3), (2, 2), (1, 4)):
        s = scenes.cornell(w, h, 4, 4)
        u = s.uniforms.copy()
        u["camera"]["pos"] = pos
        u["camera"]["dir"] = direction
        s = dataclasses.replace(s, uniforms=u)
        o_acc, _, o_rgba, _ = _oracle.render(s)
        for kernel in (0, 1, 2):
            rc = RenderConfig.from_scene(s)
            e = Engine.new(rc, kernel=kernel)
            f = e.render(rc)
            acc = e.read_accumulation()
            e.close()
            assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (w, h, kernel)
            assert np.array_equal(f.pixels, o_rgba), (w, h, kernel)
