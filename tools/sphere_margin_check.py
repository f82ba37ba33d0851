"""Brute-force check of the two inequalities the sphere walks cull with (rb_device_shade.hpp, sphere_child; CPU only).

The reference's intersect_sphere (shader.wgsl:193-215) is evaluated with numpy float32 -- every operation one IEEE
binary32 operation in the shader's order, as oracle/rb_oracle.c and the kernels do -- on random rays aimed at or just past
random spheres (sizes and distances over several decades, origins outside, on and inside the sphere, most rays within a
hair of tangency, where the discriminant cancels), and compared with the exact values (float64 on the same f32 inputs:
its rounding is 1e-9 of the f32 one).  For every REPORTED hit (t^ > 0.001), with D = max(|o - c|, r):

  E       |disc^ - disc*| / (u a D^2)                                  the walks assume E < 26
  across  (distance of the ray's line from the centre - r) / D          must stay below kSphK = 1.25e-3
  along   |d| * (distance of t^ from the chord [t_c - s, t_c + s],
          or from t_c when the exact line misses) / D                   must stay below kSphK = 1.25e-3

usage: python tools/sphere_margin_check.py [millions of rays, default 40]
"""
import sys
import numpy as np

F = np.float32
U = 2.0 ** -24
K = 1.25e-3


def dot32(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def sphere32(o, d, c, r):
    """isect_sphere in binary32; returns (t or -1, disc)."""
    oc = [o[i] - c[i] for i in range(3)]
    a = dot32(d, d)
    hb = dot32(oc, d)
    cc = dot32(oc, oc) - r * r
    disc = hb * hb - a * cc
    with np.errstate(invalid="ignore", divide="ignore"):
        sq = np.sqrt(np.maximum(disc, F(0)))
        r1 = (-hb - sq) / a
        r2 = (-hb + sq) / a
    t = np.where(r1 > F(0.001), r1, np.where(r2 > F(0.001), r2, F(-1)))
    t = np.where(disc < F(0), F(-1), t)
    return t, disc, a


def batch(rng, n):
    logu = lambda lo, hi, size: np.exp(rng.uniform(np.log(lo), np.log(hi), size))
    D = logu(0.05, 2000.0, n)
    kind = rng.integers(0, 4, n)
    r = np.where(kind == 0, D * logu(1e-4, 0.9, n),          # origin outside
        np.where(kind == 1, D * (1 + rng.normal(0, 1e-4, n)),  # origin on the surface (a path that just scattered there)
        np.where(kind == 2, D * logu(1.0, 50.0, n),            # origin inside
                 logu(0.05, 0.5, n))))                         # BASELINE C4's radii
    u = rng.normal(size=(3, n)); u /= np.linalg.norm(u, axis=0)
    c = u * D + rng.normal(size=(3, n)) * logu(1e-3, 100.0, n)   # the origin anywhere, not at 0
    o = c - u * D
    # a direction whose line passes at distance b from the centre, b = r (1 + eps), eps mostly tiny
    eps = rng.normal(0, 1, n) * logu(1e-9, 1e-1, n)
    b = np.minimum(np.abs(r * (1 + eps)), D * (1 - 1e-12))
    w = rng.normal(size=(3, n)); w -= u * (w * u).sum(0); w /= np.linalg.norm(w, axis=0)
    sin_ = b / D
    dirn = u * np.sqrt(np.maximum(1 - sin_ * sin_, 0)) + w * sin_
    dirn *= np.where(rng.random(n) < 0.1, -1.0, 1.0)   # a tenth look away (second root / behind)
    # f32 inputs; d normalised in f32 as the kernels do
    o32 = [o[i].astype(F) for i in range(3)]
    c32 = [c[i].astype(F) for i in range(3)]
    d32 = [dirn[i].astype(F) for i in range(3)]
    ln = np.sqrt(dot32(d32, d32))
    d32 = [x / ln for x in d32]
    return o32, d32, c32, r.astype(F)


def main():
    millions = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
    rng = np.random.default_rng(20240917)
    worst = {"E": 0.0, "across": 0.0, "along": 0.0}
    hits = total = 0
    chunk = 2_000_000
    while total < millions * 1e6:
        o, d, c, r = batch(rng, chunk)
        t, disc, a = sphere32(o, d, c, r)
        o64, d64, c64 = [np.stack([x.astype(np.float64) for x in v]) for v in (o, d, c)]
        r64 = r.astype(np.float64)
        oc = o64 - c64
        a_ = (d64 * d64).sum(0)
        hb = (oc * d64).sum(0)
        D2 = (oc * oc).sum(0)
        Df = np.maximum(np.sqrt(D2), r64)
        disc_ = hb * hb - a_ * (D2 - r64 * r64)
        rep = t > 0
        E = np.abs(disc.astype(np.float64) - disc_) / (U * a_ * Df * Df)
        # geometry of the reported hits
        tc = -hb / a_
        b2 = np.maximum(D2 - hb * hb / a_, 0.0)
        bb = np.sqrt(b2)
        s = np.sqrt(np.maximum(r64 * r64 - b2, 0.0) / a_)
        th = t.astype(np.float64)
        along = np.maximum(np.maximum((tc - s) - th, th - (tc + s)), 0.0) * np.sqrt(a_) / Df
        across = np.maximum(bb - r64, 0.0) / Df
        if rep.any():
            worst["E"] = max(worst["E"], float(E[rep].max()))
            worst["across"] = max(worst["across"], float(across[rep].max()))
            worst["along"] = max(worst["along"], float(along[rep].max()))
        hits += int(rep.sum())
        total += chunk
    print(f"{total / 1e6:.0f} M rays, {hits / 1e6:.1f} M reported hits")
    print(f"largest E = |disc^ - disc*| / (u a D^2) over reported hits: {worst['E']:.2f}   (assumed < 26)")
    print(f"largest across / D: {worst['across']:.3e} = {worst['across'] / K:.3f} of kSphK")
    print(f"largest along  / D: {worst['along']:.3e} = {worst['along'] / K:.3f} of kSphK")
    ok = worst["E"] < 26 and worst["across"] < K and worst["along"] < K
    print("OK" if ok else "VIOLATION")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
