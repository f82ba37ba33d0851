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


# ---- the culling rule itself (rb_kernels.hip, chunk_child), restated on numpy float32 for a chunk of ONE triangle: its tight box,
# its own normal as the cone, the bounds rb_bvh.cpp would store (|e1| |e2| in place of L^2, bf16 rounded up).  Every accepted hit
# must survive the three tests with itself as the best hit so far.  (The GPU suites test the product; this pins the formula.)
F32 = np.float32
U = F32(5.9604645e-8)
KP, KT, KS, KD = (F32(12.0) * U * F32(1.01), F32(11.0) * U * F32(1.01), F32(24.0) * U * F32(1.01), F32(16.0) * U * F32(1.01))
C0 = F32(0.03)


def _fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def _bf16_up(v):
    """float64 -> float32 whose low 16 bits are zero, rounded up; +inf beyond 1.5e5 (rb_bvh.cpp bf16_up)."""
    f = v.astype(F32)
    f = np.where(f.astype(np.float64) < v, np.nextafter(f, F32(np.inf)), f)
    b = f.view(np.uint32)
    b = ((b >> 16) + ((b & 0xFFFF) != 0)).astype(np.uint32) << 16
    return np.where(v <= 1.5e5, b.view(F32), F32(np.inf))


def _rule_keeps(o, d, v0, e1, e2, v1, v2, t_hat):
    lo = [np.minimum(np.minimum(v0[i], v1[i]), v2[i]) for i in range(3)]
    hi = [np.maximum(np.maximum(v0[i], v1[i]), v2[i]) for i in range(3)]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = [F32(1.0) / d[i] for i in range(3)]
        a = [lo[i] - o[i] for i in range(3)]
        b = [hi[i] - o[i] for i in range(3)]
        t0 = [a[i] * inv[i] for i in range(3)]
        t1 = [b[i] * inv[i] for i in range(3)]
        near = [np.fmin(t0[i], t1[i]) for i in range(3)]
        far = [np.fmax(t0[i], t1[i]) for i in range(3)]
        # the stored bounds and cone of this one triangle (host side, double precision, rounded outwards)
        e1d, e2d = [x.astype(np.float64) for x in e1], [x.astype(np.float64) for x in e2]
        l1, l2 = sum(x * x for x in e1d), sum(x * x for x in e2d)
        n = (e1d[1] * e2d[2] - e1d[2] * e2d[1], e1d[2] * e2d[0] - e1d[0] * e2d[2], e1d[0] * e2d[1] - e1d[1] * e2d[0])
        nn = np.sqrt(sum(x * x for x in n))
        g = np.sqrt(l1) * np.sqrt(l2) * (1.0 + 1e-12)
        cap = _bf16_up(g * 1e6 * (1.0 + 1e-5) * (1.0 + 1e-6))
        fa = _bf16_up(g / nn / (0.95 * 0.03) * (1.0 + 1e-5) * (1.0 + 1e-9))
        cos_a = np.cos(1e-9) * (1.0 - 1e-6) - 1e-7
        cone = [(n[i] / nn * cos_a).astype(F32) for i in range(3)]
        cone_w = F32(np.sqrt(max(0.0, 1.0 - cos_a * cos_a)) / cos_a * (1.0 + 1e-5) + 1e-7)
        # cone_cos_bound (rb_device_intersect.hpp)
        y = np.abs(_fma(d[2], cone[2], _fma(d[1], cone[1], d[0] * cone[0])))
        k2 = _fma(cone[2], cone[2], _fma(cone[1], cone[1], cone[0] * cone[0]))
        root = F32(1.000001) * np.sqrt(np.fmax(_fma(-y, y, k2), F32(0.0)) + F32(4e-6) * k2)
        lb = _fma(np.full_like(root, -cone_w), root, y - F32(1e-6))
        fl = fa * (C0 * F32(1.00001)) * (F32(1.0) / lb)
        f = np.where((lb > F32(1e-6)) & (fl < cap), fl, cap)
        mx = [np.fmax(np.abs(a[i]), np.abs(b[i])) for i in range(3)]
        sp = F32(1.001) * np.sqrt(_fma(mx[0], mx[0], _fma(mx[1], mx[1], mx[2] * mx[2]))) + \
            F32(0.5) * (((b[0] - a[0]) + (b[1] - a[1])) + (b[2] - a[2]))
        fin = f <= F32(1.5e5)
        mm = np.where(fin, sp * _fma(np.full_like(f, KP), f, np.full_like(f, KS)), F32(1e30))
        dt = np.where(fin, sp * _fma(np.full_like(f, KT), f, np.full_like(f, KD)), F32(1e30))
        ai = [np.abs(inv[i]) for i in range(3)]
        tn = np.fmax(np.fmax(_fma(-mm, ai[0], near[0]), _fma(-mm, ai[1], near[1])), _fma(-mm, ai[2], near[2]))
        tf = np.fmin(np.fmin(_fma(mm, ai[0], far[0]), _fma(mm, ai[1], far[1])), _fma(mm, ai[2], far[2]))
        return ~(tf < tn) & ~(tf < -dt) & ~(tn - dt > t_hat)


@pytest.mark.parametrize("regime", ["floor", "grazing", "steep"])
def test_the_culling_rule_keeps_every_accepted_hit(regime):
    rng = np.random.default_rng({"floor": 11, "grazing": 12, "steep": 13}[regime])
    o, d, v0, e1, e2, v1, v2 = mc.batch(rng, 600_000, regime)
    ok, t, _ = mc.mt32(o, d, v0, e1, e2)
    idx = np.nonzero(ok)[0]
    assert len(idx) > 5_000
    pick = lambda c: tuple(x[idx] for x in c)
    keeps = _rule_keeps(pick(o), pick(d), pick(v0), pick(e1), pick(e2), pick(v1), pick(v2), t[idx])
    assert keeps.all(), int((~keeps).sum())


def test_the_culling_rule_keeps_the_worst_cases_of_the_search():
    # the input bits profiles/r03_margin_search.txt ends its three searches with
    import re
    text = open(os.path.join(ROOT, "profiles", "r03_margin_search.txt")).read()
    rows = re.findall(r"inputs \(f32 bits\): ((?:[0-9a-f]{8} ?){15})", text)
    assert len(rows) == 3
    P = np.array([[int(w, 16) for w in r.split()] for r in rows], dtype=np.uint32).view(F32).T.copy()   # [15][3]
    o, d, v0, v1, v2 = (tuple(P[3 * k + i] for i in range(3)) for k in range(5))
    e1 = tuple(v1[i] - v0[i] for i in range(3))
    e2 = tuple(v2[i] - v0[i] for i in range(3))
    ok, t, _ = mc.mt32(o, d, v0, e1, e2)
    assert ok.all()
    assert _rule_keeps(o, d, v0, e1, e2, v1, v2, t).all()


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_two_inequalities_follow_from_the_standard_model_of_rounding():
    # tools/margin_certify.py: the error analysis of shader.wgsl:248-280 carried out in exact rational arithmetic -- no rays,
    # no roundings sampled -- ends in comparisons between rationals; every one must hold with the kernel's constants
    mcert = _tool("margin_certify")
    for name, need, has in mcert.checks:
        if has is None:
            assert need >= mcert.Fr(95, 100), name
        else:
            assert need <= has, (name, float(need), float(has))
    # the derivation reproduces the hand-made constants of DESIGN.md section 4.1 (E7) to the digit they were quoted with
    assert abs(float(mcert.c1) - 11.27) < 0.02 and abs(float(mcert.c2) - 4.60) < 0.02
    assert abs(float(mcert.c1t) - 10.27) < 0.02 and abs(float(mcert.c2t) - 4.60) < 0.02
    # and the kernel's constants are the ones it was run against
    src = open(os.path.join(ROOT, "renderbaby_amd", "csrc", "rb_kernels.hip")).read()
    for k, v in (("kChunkKP", "12.0f"), ("kChunkKT", "11.0f"), ("kChunkKS", "24.0f"), ("kChunkKD", "16.0f")):
        assert f"constexpr float {k} = {v} * 5.9604645e-8f * 1.01f;" in src, k
    assert "constexpr float kChunkFMax = 1.5e5f;" in src


def test_reported_sphere_hits_stay_within_the_sphere_walks_margin():
    # tools/sphere_margin_check.py, a small dose: the reference's intersect_sphere in numpy float32 against exact values on
    # near-tangent rays; the discriminant's error constant and both parts of the margin (profiles/r04_sphere_margin_check.txt: 60 M rays)
    smc = _tool("sphere_margin_check")
    rng = np.random.default_rng(5)
    o, d, c, r = smc.batch(rng, 400_000)
    t, disc, a = smc.sphere32(o, d, c, r)
    o64, d64, c64 = [np.stack([x.astype(np.float64) for x in v]) for v in (o, d, c)]
    r64 = r.astype(np.float64)
    oc = o64 - c64
    a_, hb, D2 = (d64 * d64).sum(0), (oc * d64).sum(0), (oc * oc).sum(0)
    Df = np.maximum(np.sqrt(D2), r64)
    rep = t > 0
    assert rep.sum() > 100_000
    E = np.abs(disc.astype(np.float64) - (hb * hb - a_ * (D2 - r64 * r64))) / (smc.U * a_ * Df * Df)
    assert E[rep].max() < 26.0
    b = np.sqrt(np.maximum(D2 - hb * hb / a_, 0.0))
    tc, s = -hb / a_, np.sqrt(np.maximum(r64 * r64 - np.maximum(D2 - hb * hb / a_, 0.0), 0.0) / a_)
    th = t.astype(np.float64)
    along = np.maximum(np.maximum((tc - s) - th, th - (tc + s)), 0.0) * np.sqrt(a_) / Df
    across = np.maximum(b - r64, 0.0) / Df
    assert across[rep].max() < smc.K and along[rep].max() < smc.K
    src = open(os.path.join(ROOT, "renderbaby_amd", "csrc", "rb_device_shade.hpp")).read()
    assert "constexpr float kSphK = 1.25e-3f * 1.001f;" in src
