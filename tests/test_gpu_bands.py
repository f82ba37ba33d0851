"""The stepped walks deal a large launch to the 8 XCDs in stripes of tile rows, one work-queue word per group of blocks
(rb_kernels.hip, ItemQueue::band_item).  Items are independent, so nothing may change: every path is
traced exactly once, and the rows at the top, in the middle and in the last, partial stripe are the oracle's bits.  The
launches here are large enough for the banded queues (items >= 32 stripes: 1920 x 1080 at 2 spp is 4.1 M items in stripes of
123 k); small frames -- every other parity test -- take the single word."""
import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig, scenes
from tests import _oracle

pytestmark = pytest.mark.gpu

ROWS = ((0, 2), (537, 540), (1077, 1080))


def _check_rows(s, acc):
    for r in ROWS:
        o_acc, _, _, _ = _oracle.render(s, rows=r)
        assert np.array_equal(acc[r[0]:r[1]].view(np.uint32), o_acc[r[0]:r[1]].view(np.uint32)), r


@pytest.mark.parametrize("walk", ["chunk", "reference", "own-tree"])
def test_banded_queues_trace_every_item_once_mesh(walk):
    s = scenes.mesh_scene(112, 112, 1920, 1080, 2, 5)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, stats=True, reference_walk=(walk == "reference"), fast_bvh=(walk == "own-tree"))
    e.render(rc)
    acc, st, kn = e.read_accumulation(), e.stats(), e.last_kernel_name()
    e.close()
    assert kn == {"chunk": "k_trace_chunk", "reference": "k_trace_bvh", "own-tree": "k_trace_fast"}[walk]
    assert st["paths"] == 1920 * 1080 * 2
    _check_rows(s, acc)


def test_banded_queues_trace_every_item_once_sphere_tree():
    s = scenes.spheres_scene(n=3000, width=1920, height=1080, spp=2, max_depth=4, seed=5, extent=30.0)
    rc = RenderConfig.from_scene(s)
    e = Engine.new(rc, stats=True)
    e.render(rc)
    acc, st, kn = e.read_accumulation(), e.stats(), e.last_kernel_name()
    e.close()
    assert kn == "k_trace_sph"
    assert st["paths"] == 1920 * 1080 * 2
    _check_rows(s, acc)
