"""Numerical stress test of the inequality the culled walks rest on (used by section 4.2):

    if the reference's f32 Moller-Trumbore (shader.wgsl:248-280) ACCEPTS a hit of triangle k with parameter t^,
    then X = o + t^ d lies within  dist_inf(X, box(k)) <= Sp (27 u F + 24 u)  of the triangle's box, where
    F = min(L^2 / 1e-6, (L^2 / N) / (0.95 |cos(d, n)|)),  Sp >= |o - v0| + 2 L,  u = 2^-24.

E1-E7 are first-order bounds added up by hand; this checks the end result by brute force instead of by reading: random
triangles (sizes, aspect ratios and distances over several decades) and rays aimed at or just past them, most of them
within a fraction of a degree of the triangle's plane -- the regime in which |a^| approaches the 1e-6 floor and the
reported hit wanders -- evaluated with exactly the shader's operations (numpy float32: every +, -, * is one IEEE binary32
operation, no FMA; the reciprocal is the correctly rounded 1 / a).  Reports the largest observed  dist / margin  ratio per
regime; anything above 1 would be a counterexample to the bound (and to both culled walks).  CPU only.

    python tools/margin_check.py [millions of rays, default 40] > profiles/<tag>_margin_check.txt
"""
import sys
import time

import numpy as np

F32 = np.float32
U = 2.0 ** -24


def cross32(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def dot32(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def mt32(o, d, v0, e1, e2):
    """shader.wgsl:248-280 on float32 component arrays -> (accepted mask, t, a)."""
    h = cross32(d, e2)
    a = dot32(e1, h)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        f = F32(1.0) / a
        s = (o[0] - v0[0], o[1] - v0[1], o[2] - v0[2])
        u = f * dot32(s, h)
        q = cross32(s, e1)
        v = f * dot32(d, q)
        t = f * dot32(e2, q)
        ok = ~(np.abs(a) < F32(1e-6)) & ~(u < 0) & ~(u > 1) & ~(v < 0) & ~(u + v > 1) & (t > 0) & (t > F32(0.001))
    return ok, t, a


def batch(rng, n, regime):
    """n random (triangle, ray) pairs; all arrays float32 [3][n]."""
    f = lambda lo, hi: rng.uniform(lo, hi, n)
    # triangle: v0, two edges with length L = 10^[-3, -0.42] (L^2 / 1e-6 <= 1.5e5), aspect down to 1:50
    L = 10.0 ** f(-3.0, -0.42)
    v0 = rng.uniform(-50, 50, (3, n))
    dir1 = rng.normal(size=(3, n)); dir1 /= np.linalg.norm(dir1, axis=0)
    dir2 = rng.normal(size=(3, n)); dir2 -= (dir2 * dir1).sum(0) * dir1; dir2 /= np.linalg.norm(dir2, axis=0)
    skew = f(-0.9, 0.9)
    e1 = dir1 * L
    e2 = (dir1 * skew + dir2 * 10.0 ** f(-1.7, 0.0)) * L * f(0.3, 1.0)
    nrm = np.cross(e1.T, e2.T).T
    nn = np.linalg.norm(nrm, axis=0)
    nrm /= nn
    # a point near the triangle (barycentrics a little beyond it), an origin at distance 10^[-2, 2] from it
    bu, bv = f(-0.3, 1.3), f(-0.3, 1.3)
    target = v0 + e1 * bu + e2 * bv
    dist = 10.0 ** f(-2.0, 2.0)
    inplane = dir1 * f(-1, 1) + dir2 * f(-1, 1)
    inplane /= np.linalg.norm(inplane, axis=0)
    if regime == "grazing":      # |cos| from 1e-8 up to 3e-2: the determinant near its floor
        tilt = 10.0 ** f(-8.0, -1.5) * rng.choice([-1.0, 1.0], n)
    elif regime == "floor":      # aimed so that |a| = N |cos| lands within a factor 30 of 1e-6
        tilt = (1e-6 * 10.0 ** f(0.0, 1.5)) / np.maximum(nn, 1e-30) * rng.choice([-1.0, 1.0], n)
    else:                        # "steep": ordinary incidence
        tilt = f(0.05, 5.0) * rng.choice([-1.0, 1.0], n)
    dd = inplane + nrm * tilt
    dd /= np.linalg.norm(dd, axis=0)
    o = target - dd * dist + nrm * (f(-1, 1) * 10.0 ** f(-9.0, -3.0) * dist)   # a hair off, so that hits and near misses both occur
    to32 = lambda m: tuple(np.ascontiguousarray(m[i], dtype=F32) for i in range(3))
    o32, v032 = to32(o), to32(v0)
    d32 = to32(dd)
    # the kernels' d is normalize(...) in f32: |d| = 1 +- 4u; renormalise in f32 the way the shader does
    ln = np.sqrt(dot32(d32, d32))
    d32 = (d32[0] / ln, d32[1] / ln, d32[2] / ln)
    v1, v2 = to32(v0 + e1), to32(v0 + e2)
    e132 = (v1[0] - v032[0], v1[1] - v032[1], v1[2] - v032[2])          # the f32 edges of k_prep_tris
    e232 = (v2[0] - v032[0], v2[1] - v032[1], v2[2] - v032[2])
    return o32, d32, v032, e132, e232, v1, v2


def ratios(o, d, v0, e1, e2, v1, v2):
    """Per (ray, triangle) pair, float32 component arrays: (accepted mask, and for the accepted ones in order: the three
    ratios error / margin -- reported point against section 4.1's margin, exact plane point and |t^ - t*| against section
    4.2's two parts -- and a dict of arrays describing the cases)."""
    ok, t, a = mt32(o, d, v0, e1, e2)
    idx = np.nonzero(ok)[0]
    g = lambda c: tuple(x[idx].astype(np.float64) for x in c)
    o, d, v0, e1, e2, v1, v2 = g(o), g(d), g(v0), g(e1), g(e2), g(v1), g(v2)
    t, a = t[idx].astype(np.float64), a[idx].astype(np.float64)
    X = [o[i] + t * d[i] for i in range(3)]
    # the triangle's box from the f32 vertices the builders use
    dist = np.zeros(len(idx))
    for i in range(3):
        lo, hi = np.minimum(np.minimum(v0[i], v1[i]), v2[i]), np.maximum(np.maximum(v0[i], v1[i]), v2[i])
        dist = np.maximum(dist, np.maximum(lo - X[i], X[i] - hi))
    dist = np.maximum(dist, 0.0)
    L2 = np.maximum(sum(e1[i] ** 2 for i in range(3)), sum(e2[i] ** 2 for i in range(3)))
    nx = (e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0])
    N = np.sqrt(sum(c * c for c in nx))
    cosn = np.abs(sum(d[i] * nx[i] for i in range(3))) / np.maximum(N, 1e-300)
    F = np.minimum(L2 / 1e-6, (L2 / np.maximum(N, 1e-300)) / np.maximum(0.95 * cosn, 1e-300))
    s = np.sqrt(sum((o[i] - v0[i]) ** 2 for i in range(3)))
    Sp = s + 2.0 * np.sqrt(L2)
    margin = Sp * (27.0 * U * F + 24.0 * U)
    ratio = dist / margin
    # the same bound in the two parts the chunked walk uses: the exact plane point Q* = o + t* d of
    # the f32 inputs (double precision: 2^-53 against the 2^-24 under test) within S (12 u F' + 24 u) of the box, and
    # |t^ - t*| <= S (11 u F' + 16 u), with S = |o - v0| + L / 2 and F' = F with |e1| |e2| in place of L^2
    h = (d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0])
    ax = sum(e1[i] * h[i] for i in range(3))
    sv = [o[i] - v0[i] for i in range(3)]
    q = (sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0])
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = sum(e2[i] * q[i] for i in range(3)) / ax
    Q = [o[i] + tx * d[i] for i in range(3)]
    dq = np.zeros(len(idx))
    for i in range(3):
        lo, hi = np.minimum(np.minimum(v0[i], v1[i]), v2[i]), np.maximum(np.maximum(v0[i], v1[i]), v2[i])
        dq = np.maximum(dq, np.maximum(lo - Q[i], Q[i] - hi))
    A = np.sqrt(sum(e1[i] ** 2 for i in range(3)) * sum(e2[i] ** 2 for i in range(3)))    # |e1| |e2| in place of L^2
    Fa = np.minimum(A / 1e-6, (A / np.maximum(N, 1e-300)) / np.maximum(0.95 * cosn, 1e-300))
    Sh = s + 0.5 * np.sqrt(L2)
    across = np.maximum(dq, 0.0) / (Sh * (12.0 * U * Fa + 24.0 * U))
    along = np.abs(t - tx) / (Sh * (11.0 * U * Fa + 16.0 * U))
    valid = (F <= 1.5e5) & np.isfinite(across) & np.isfinite(along)   # beyond 1.5e5 no bound is claimed (always entered)
    ratio, across, along = np.where(valid, ratio, 0.0), np.where(valid, across, 0.0), np.where(valid, along, 0.0)
    info = dict(dist=dist, margin=margin, a=a, cos=cosn, L=np.sqrt(L2), s=s, F=F, t=t)
    return ok, ratio, across, along, info


def check(rng, n, regime):
    ok, ratio, across, along, info = ratios(*batch(rng, n, regime))
    if len(ratio) == 0:
        return 0, 0.0, 0.0, None, 0.0, 0.0
    k = int(np.argmax(ratio))
    worst = dict(ratio=float(ratio[k]), **{key: float(v[k]) for key, v in info.items()})
    return len(ratio), float(ratio.max()), float(np.percentile(ratio, 99.9)), worst, float(across.max()), float(along.max())


def main():
    millions = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    rng = np.random.default_rng(20241004)
    per = 2_000_000
    print(f"# margin_check: {millions:g} M rays per regime, numpy float32 = the shader's single IEEE operations; seed 20241004")
    print("# bound: dist_inf(o + t^ d, box(triangle)) <= (|o - v0| + 2 L) (27 u F + 24 u),  F = min(L^2 / 1e-6, (L^2 / N) / (0.95 |cos|))")
    print("# in two parts (the chunked walk): dist_inf(o + t* d, box) <= S (12 u F' + 24 u) and |t^ - t*| <= S (11 u F' + 16 u); t* = the exact plane point's,\n# S = |o - v0| + L / 2, F' = F with |e1| |e2| in place of L^2")
    t0 = time.time()
    overall = 0.0
    for regime in ("floor", "grazing", "steep"):
        acc, mx, p999, worst, across, along = 0, 0.0, 0.0, None, 0.0, 0.0
        for _ in range(int(millions * 1e6 / per)):
            n_ok, m, p, w, ac, al = check(rng, per, regime)
            acc += n_ok
            p999 = max(p999, p)
            across, along = max(across, ac), max(along, al)
            if m > mx:
                mx, worst = m, w
        overall = max(overall, mx, across, along)
        print(f"{regime:8s}: {acc:10d} accepted hits of {int(millions * 1e6)} rays; largest dist / margin = {mx:.4f}; 99.9th percentile <= {p999:.4f}")
        print(f"          in two parts: plane point to box / S (12 u F' + 24 u) = {across:.4f}; |t^ - t*| / S (11 u F' + 16 u) = {along:.4f}")
        if worst:
            print("          worst case: " + ", ".join(f"{k} {v:.4g}" for k, v in worst.items()))
    print(f"# largest ratio overall {overall:.4f} ({'within the bound' if overall <= 1.0 else 'COUNTEREXAMPLE'}); {time.time() - t0:.0f} s")
    return 0 if overall <= 1.0 else 1


if __name__ == "__main__":
    sys.exit(main())
