"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the CPU oracle): the oracle must keep reproducing them, and the HIP path
must reproduce them bit for bit through the C ABI."""
import glob
import os

import numpy as np
import pytest

from tests import _oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLDEN]


def test_fixtures_present():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_oracle_reproduces_golden(path):
    scene, accum, rgba, stats = _oracle.load_golden(path)
    a, _, r, st = _oracle.render(scene)
    assert np.array_equal(a.view(np.uint32), accum.view(np.uint32))
    assert np.array_equal(r, rgba)
    assert st == stats


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [1, 2, 3], ids=["pixel", "queue", "stream"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_reproduces_golden(path, kernel):
    from renderbaby_amd import Engine, RenderConfig
    scene, accum, rgba, stats = _oracle.load_golden(path)
    rc = RenderConfig.from_scene(scene)
    eng = Engine.new(rc, kernel=kernel, stats=True, reference_walk=True)   # the golden counters are the reference walk's
    frame = eng.render(rc)
    acc = eng.read_accumulation()
    st = eng.stats()
    eng.close()
    assert np.array_equal(acc.view(np.uint32), accum.view(np.uint32))
    assert np.array_equal(frame.pixels, rgba)
    assert {k: st[k] for k in _oracle.STAT_KEYS} == stats
