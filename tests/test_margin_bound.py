"""The inequality both culled walks rest on (DESIGN.md sections 4.1 / 4.2), checked by brute force instead of by reading
(tools/margin_check.py): whenever the reference's f32 Moller-Trumbore accepts a hit, the reported point lies within the
walks' margin of the triangle's box.  A small dose here; profiles/r03_margin_check.txt has 180 M rays.  No GPU."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("margin_check", os.path.join(ROOT, "tools", "margin_check.py"))
mc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mc)


@pytest.mark.parametrize("regime", ["floor", "grazing", "steep"])
def test_accepted_hits_stay_within_the_margin(regime):
    rng = np.random.default_rng({"floor": 1, "grazing": 2, "steep": 3}[regime])
    accepted, worst, _, case, across, along = mc.check(rng, 600_000, regime)
    assert accepted > 5_000                      # the sampler really produces accepted hits in this regime
    assert worst <= 1.0, case                    # a counterexample to the bound would be a counterexample to both walks
    # the chunked walk's two parts: the exact plane point against the box, the reported t against the exact one
    assert across <= 1.0 and along <= 1.0, (across, along)


def test_the_emulated_triangle_test_is_the_oracles():
    # the numpy float32 restatement used above against the oracle's C restatement of shader.wgsl:248-280, bit for bit
    from tests import _oracle
    rng = np.random.default_rng(9)
    o, d, v0, e1, e2, v1, v2 = mc.batch(rng, 2000, "grazing")
    ok, t, a = mc.mt32(o, d, v0, e1, e2)
    n_hit = 0
    for i in range(2000):
        col = lambda c: [float(c[0][i]), float(c[1][i]), float(c[2][i])]
        to, _, _ = _oracle.isect_triangle(col(o), col(d), col(v0), col(v1), col(v2))
        hit_o = to > 0.001
        assert bool(ok[i]) == bool(hit_o), i
        if ok[i]:
            n_hit += 1
            assert np.float32(to).view(np.uint32) == t[i].view(np.uint32), i
    assert n_hit > 10


def test_a_short_adversarial_search_stays_within_the_bounds():
    # tools/margin_search.py in miniature: the worst of a random sample, nudged by a few ulps per generation towards larger
    # error / margin; profiles/r03_margin_search.txt has the long run.  The search must climb (it is a search) and stay below 1.
    ms_spec = importlib.util.spec_from_file_location("margin_search", os.path.join(ROOT, "tools", "margin_search.py"))
    ms = importlib.util.module_from_spec(ms_spec)
    ms_spec.loader.exec_module(ms)
    rng = np.random.default_rng(5)
    o, d, v0, e1, e2, v1, v2 = mc.batch(rng, 300_000, "floor")
    S = np.stack(list(o) + list(d) + list(v0) + list(v1) + list(v2)).astype(np.float32)
    for which in range(3):
        r = ms.evaluate(S, which)
        top = np.argsort(r)[-300:]
        P, best = S[:, top].copy(), r[top].copy()
        start = best.max()
        for g in range(6):
            kids = ms.nudge(rng, P, big=False)
            rk = ms.evaluate(kids, which)
            better = rk > best
            P[:, better], best[better] = kids[:, better], rk[better]
        assert start > 0.0 and best.max() >= start and best.max() <= 1.0, (which, start, best.max())
