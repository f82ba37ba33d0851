"""Differential fuzzing (tools/fuzz_parity.py) in a small dose: random scenes with nasty geometry, every
kernel variant that applies, bit for bit against the oracle."""
import importlib.util
import os

import numpy as np
import pytest

from renderbaby_amd import Engine, RenderConfig
from tests import _oracle

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize("first", [0, 1000, 2000])
def test_random_scenes_all_variants(first):
    for seed in range(first, first + 25):
        s = fuzz.random_scene(seed)
        o_acc, _, o_rgba, o_st = _oracle.render(s)
        rc = RenderConfig.from_scene(s)
        for name, kw in fuzz.variants(s):
            e = Engine.new(rc, stats=True, **kw)
            frame = e.render(rc)
            acc, st = e.read_accumulation(), e.stats()
            e.close()
            assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)), (seed, name)
            assert np.array_equal(frame.pixels, o_rgba), (seed, name)
            assert st["segments"] == o_st["segments"] and st["paths"] == o_st["paths"], (seed, name)
