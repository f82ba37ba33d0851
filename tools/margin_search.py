"""Adversarial search on the inequality the culled walks rest on (used in two parts by section
4.2) -- tools/margin_check.py samples rays at random, and random rays do not find worst-case roundings; this climbs towards
them.  Start from the worst cases of a random sample (per criterion: the reported point against 4.1's margin, the exact
plane point against 4.2's across-the-ray part, |t^ - t*| against its along-the-ray part), then, generation after
generation, nudge the 15 input floats of each survivor (origin, direction, three vertices) by a few ulps -- and now and
then by a relative 1e-3, to move between basins -- evaluate the shader's f32 triangle test on every child (numpy float32 =
the contract's single IEEE operations) and keep a child if its ratio error / margin is larger.  |d| stays within 1 +- 4 u
(the kernels' normalised directions), children whose hit is no longer accepted drop out.  A ratio above 1 would be a
counterexample.  CPU only.

    python tools/margin_search.py [generations, default 300] [population, default 4000] [seed, default 777] > profiles/<tag>_margin_search.txt
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import margin_check as mc  # noqa: E402

F32 = np.float32


def derive(o, d, v0, v1, v2):
    e1 = tuple(v1[i] - v0[i] for i in range(3))   # the f32 edges of k_prep_tris
    e2 = tuple(v2[i] - v0[i] for i in range(3))
    return o, d, v0, e1, e2, v1, v2


def evaluate(P, which):
    """P: float32 [15][n] (o, d, v0, v1, v2) -> ratio per column (0 where the hit is not accepted or |d| is off)."""
    o, d, v0, v1, v2 = (tuple(P[3 * k + i] for i in range(3)) for k in range(5))
    ok, ratio, across, along, _ = mc.ratios(*derive(o, d, v0, v1, v2))
    out = np.zeros(P.shape[1])
    out[np.nonzero(ok)[0]] = (ratio, across, along)[which]
    dn = np.sqrt(sum(x.astype(np.float64) ** 2 for x in d))
    out[np.abs(dn - 1.0) > 4.0 * mc.U] = 0.0
    return out


def nudge(rng, P, big):
    """children of P: a few of the 15 floats moved by +-1..4 ulps (or by a relative 1e-3 when `big`); the direction is
    renormalised in f32 the way the shader does."""
    C = P.copy()
    n = C.shape[1]
    mask = rng.random(C.shape) < 0.25
    if big:
        C = np.where(mask, C * (1.0 + rng.normal(0.0, 1e-3, C.shape)).astype(F32), C).astype(F32)
    else:
        steps = rng.integers(-4, 5, C.shape)
        bits = C.view(np.int32) + np.where(mask, steps, 0).astype(np.int32) * np.where(C >= 0, 1, -1).astype(np.int32)
        C = bits.view(F32).copy()
    d = [C[3 + i] for i in range(3)]
    ln = np.sqrt(mc.dot32(d, d))
    for i in range(3):
        C[3 + i] = d[i] / ln
    return C


def main():
    gens = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    pop = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 777
    rng = np.random.default_rng(seed)
    names = ("reported point / Sp (27 u F + 24 u)   [section 4.1, k_trace_fast]",
             "exact plane point / S (12 u F' + 24 u) [section 4.2, across the ray]",
             "|t^ - t*| / S (11 u F' + 16 u)         [section 4.2, along the ray]")
    print(f"# margin_search: {gens} generations, {pop} survivors x 8 children per criterion; seeds = the worst of 6 M random rays; seed {seed}")
    t0 = time.time()
    # one random sample serves as the seed population of all three searches
    cols = []
    for regime in ("floor", "grazing", "steep"):
        for _ in range(2):
            o, d, v0, e1, e2, v1, v2 = mc.batch(rng, 1_000_000, regime)
            cols.append(np.stack(list(o) + list(d) + list(v0) + list(v1) + list(v2)).astype(F32))
    S = np.concatenate(cols, axis=1)
    overall = 0.0
    for which in range(3):
        r = evaluate(S, which)
        top = np.argsort(r)[-pop:]
        P, best = S[:, top].copy(), r[top].copy()
        start = float(best.max())
        for g in range(gens):
            kids = np.concatenate([nudge(rng, P, big=(k == 7 and g % 4 == 0)) for k in range(8)], axis=1)
            rk = evaluate(kids, which).reshape(8, -1)
            kb = rk.argmax(axis=0)
            cand = rk[kb, np.arange(P.shape[1])]
            better = cand > best
            sel = kids.reshape(15, 8, -1)[:, kb, np.arange(P.shape[1])]
            P[:, better], best[better] = sel[:, better], cand[better]
            if g % 25 == 24:      # the weak half makes room for copies of the strong half
                order = np.argsort(best)
                half = len(order) // 2
                P[:, order[:half]], best[order[:half]] = P[:, order[half:half * 2]], best[order[half:half * 2]]
        k = int(np.argmax(best))
        o, d, v0, v1, v2 = (tuple(P[3 * j + i][k:k + 1] for i in range(3)) for j in range(5))
        _, _, _, _, info = mc.ratios(*derive(o, d, v0, v1, v2))
        overall = max(overall, float(best.max()))
        print(f"{names[which]}: random sample {start:.4f} -> after the search {best.max():.4f}; median survivor {np.median(best):.4f}")
        print("    worst case: " + ", ".join(f"{key} {float(v[0]):.4g}" for key, v in info.items()))
        print("    inputs (f32 bits): " + " ".join(f"{int(x):08x}" for x in P[:, k].view(np.uint32)), flush=True)
    print(f"# largest ratio reached {overall:.4f} ({'within the bounds' if overall <= 1.0 else 'COUNTEREXAMPLE'}); {time.time() - t0:.0f} s")
    return 0 if overall <= 1.0 else 1


if __name__ == "__main__":
    sys.exit(main())
