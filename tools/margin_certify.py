"""Machine-checked derivation of the inequalities the culled walks rest on: the two of the chunked walk (rb_kernels.hip chunk_child;
DESIGN.md 4.1 / 4.2) and, at the end, the sphere walks' (rb_device_shade.hpp sphere_child).

    Claim.  Let the reference's triangle test (shader.wgsl:248-280: every operation one IEEE binary32 operation, round to
    nearest, no FMA; dot = (x x' + y y') + z z'; f = the correctly rounded 1 / a) ACCEPT a hit of the triangle (v0, e1, e2)
    for the ray (o, d) and report t^.  Let t* be the exact parameter at which the ray meets the triangle's plane,
    Q* = o + t* d, and F >= |e1| |e2| / |a^|, S >= max(|o - v0| + L / 2, L), L = max(|e1|, |e2|).  Then

        across:  dist(Q*, box of the triangle) + (what the kernel's own slab arithmetic can lose)  <=  S (KP F + KS)
        along:   |t^ - t*| + (what the kernel's comparison can lose)                               <=  S (KT F + KD)

    with KP = 12 u, KS = 24 u, KT = 11 u, KD = 16 u (times the kernel's 1.01), u = 2^-24.

This script does not sample rays and does not sample roundings.  It carries out the error analysis itself -- the standard
model fl(x op y) = (x op y)(1 + delta), |delta| <= u, through the operation sequence of the shader, norm-wise, with every
second-order term kept -- in exact rational arithmetic (fractions.Fraction; sqrt 2 replaced by a rational upper bound),
and then checks the claim as a handful of comparisons between rationals.  What remains to be trusted is the text of the
lemmas below (each is two lines), not an addition done by hand: r02's derivation absorbed a second-order term it should
not have, which is exactly the kind of slip this removes.

Hypotheses (where each comes from):
  H1  |a^| >= 1e-6 and u^ >= 0, v^ >= 0, fl(u^ + v^) <= 1, t^ > 0              the shader's own accept conditions (:253,:259,:265,:268)
  H2  | |d| - 1 | <= 4 u                                                        every ray direction is the output of normalize()
  H3  L^2 / |a^| <= 1.5e5                                                       the kernel culls only with F <= kChunkFMax = 1.5e5, and
      the tree stores F so that this follows: F_floor = |e1||e2| / 1e-6 only when L^2 / 1e-6 <= 1.5e5 (rb_bvh.cpp pack_fac), else
      the cone bound with L^2 in place of |e1||e2| -- checked below (LEMMA cone) without circularity
  H4  no underflow in the products (the scene is not 1e-30 units across), boxes contain their triangles (checked at build
      time: a slot whose box does not gets an unbounded margin, rb_bvh.cpp fill_child)

usage: python tools/margin_certify.py        (prints the derivation's constants and PASS / FAIL; exit status 1 on FAIL)
"""
import sys
from fractions import Fraction as Fr

u = Fr(1, 2 ** 24)
u1 = u / (1 - u)                      # (1 + delta)^-1 - 1 <= u1
SQRT2 = Fr(14142136, 10 ** 7)         # > sqrt(2)
assert SQRT2 * SQRT2 > 2


def gamma(k):                         # (1 + u)^k - 1: k roundings compounded
    return (1 + u) ** k - 1


# ---------------------------------------------------------------------------------------------------------------------
# LEMMA cross.  c^ = fl-cross(x, y) on the vectors x, y actually used (component i: fl(fl(p) - fl(q)), p, q products):
#   |c^_i - c_i| <= u (1 + u)(|p| + |q|) + u |c_i|,  and  sum_i (|p_i| + |q_i|)^2 <= 2 |x|^2 |y|^2  (Cauchy-Schwarz), so
#   |c^ - c|_2 <= [ (1 + u) sqrt2 + 1 ] u |x| |y|.
CROSS = (1 + u) * SQRT2 + 1           # in units of u |x| |y|

# LEMMA dot3.  r^ = fl(fl(fl(x1 y1) + fl(x2 y2)) + fl(x3 y3)) on the vectors actually used:
#   r^ - r = [first two products through two roundings] + [third through one] + delta5 (A + B),  A + B = r^ / (1 + delta5)
#   |r^ - r| <= gamma2 (|p1| + |p2|) + u |p3| + u1 |r^|,  2 (|p1| + |p2|) + |p3| <= 2 |x| |y|,  |r^| <= |r| + |r^ - r|
#   =>  |r^ - r| <= ( gamma2 / 2 * 2 |x| |y| ... ) -- kept simple and safe:  (gamma2 |x| |y| + u1 |r|) / (1 - u1)
def dot3(nx_ny, r_abs):
    """rounding error of a 3-term dot product; arguments are coefficients of the same monomial (or a pair, see below)"""
    return (gamma(2) * nx_ny + u1 * r_abs) / (1 - u1)


# Quantities are tracked as  bound = coefficient x (a named monomial of the magnitudes s = |o - v0|, D = |d|, P = |e1|, Q = |e2|)
# ---------------------------------------------------------------------------------------------------------------------
# step 1   s^ = fl(o - v0):            |s^ - s| <= u s,   |s^| <= (1 + u) s
E_s = u
N_s = 1 + u
# step 2   h^ = fl(d x e2):            |h^ - h| <= CROSS u D Q,   |h^| <= (1 + CROSS u) D Q
E_h = CROSS * u
N_h = 1 + CROSS * u
# step 3   a^ = fl(e1 . h^):           rounding dot3(P |h^|, |e1 . h^|) with |e1 . h^| <= |a| + P E_h;  propagation P E_h
#          |a^ - a| <= alpha1 D A + alpha2 |a|,   A = P Q
alpha1 = (gamma(2) * N_h + u1 * E_h) / (1 - u1) + E_h
alpha2 = u1 / (1 - u1)
# step 4   numerators.
#   Nu = s . h,  Nu^ = fl(s^ . h^):   propagation |s^ - s| |h^| + |s| |h^ - h| = (E_s N_h + E_h) s D Q =: prop_u s D Q
prop_u = E_s * N_h + E_h
a_u = (gamma(2) * N_s * N_h + u1 * prop_u) / (1 - u1) + prop_u      # |Nu^ - Nu| <= a_u s D Q + b |Nu|
b_n = u1 / (1 - u1)
#   q^ = fl(s^ x e1):                 rounding CROSS u |s^| P,  propagation |s^ - s| P:   |q^ - q| <= E_q s P,  |q^| <= N_q s P
E_q = CROSS * u * N_s + E_s
N_q = 1 + E_q
#   Nv = d . q,  Nv^ = fl(d . q^)  and  Nt = e2 . q,  Nt^ = fl(e2 . q^):  rounding dot3(|x| |q^|, .), propagation |x| E_q
a_v = (gamma(2) * N_q + u1 * E_q) / (1 - u1) + E_q                  # |Nv^ - Nv| <= a_v s D P + b |Nv|;  |Nt^ - Nt| <= a_v s P Q + b |Nt|
a_t = a_v
g = (1 + gamma(2)) * max(a_u, a_v) / u                              # in units of u: the numerators' constant after f^ and the product
g_t = (1 + gamma(2)) * a_t / u

# ---------------------------------------------------------------------------------------------------------------------
# step 5   quotients.  f^ = fl(1 / a^) = (1 + delta) / a^,  x^ = fl(f^ N^) = N^ / a^ (1 + eps),  |eps| <= gamma2.
#   x^ - x* = (N^ - N) / a^ (1 + eps) + N (1 / a^ - 1 / a)(1 + eps) + x* eps
#   |x^ - x*| <= (1 + gamma2) [a_x (monomial) + b |N|] / |a^| + (1 + gamma2) |x*| |a^ - a| / |a^| + gamma2 |x*|
# With Z = u D / |a^|, W = A Z, lam = L^2 Z >= W:   |a| / |a^| <= kappa(W) = (1 + (alpha1 / u) W) / (1 - alpha2)
#   |x^ - x*| <= g (monomial without D) Z + |x*| B(W),   B(W) = b1 W + b0
# H3 and H2:  lam <= 1.5e5 u (1 + 4 u)
D_max, D_min = 1 + 4 * u, 1 - 4 * u
lam_max = Fr(150000) * u * D_max


def kappa(W):
    return (1 + (alpha1 / u) * W) / (1 - alpha2)


def B(W):
    return (1 + gamma(2)) * ((b_n + alpha2) * kappa(W) + (alpha1 / u) * W) + gamma(2)


b0 = B(Fr(0))
b1 = (B(lam_max) - b0) / lam_max      # B is affine in W
B_max = B(lam_max)
assert 0 < B_max < Fr(1, 10)

# ---------------------------------------------------------------------------------------------------------------------
# step 6   acceptance (H1): u^ + v^ <= 1 + u1;  Sigma := |u*| + |v*| <= 1 + u1 + du + dv,  du + dv <= g s (P + Q) Z + Sigma B
#   =>  Sigma <= (1 + u1 + g s (P + Q) Z) / (1 - B)
inv1mB = 1 / (1 - B_max)

# step 7   across.  Q* = v0 + u* e1 + v* e2;  the point P' = v0 + u' e1 + v' e2 with (u', v') = (u^, v^) / (1 + u1) is in the triangle;
#   |Q* - P'| <= (du + u1) P + (dv + u1) Q,   du P + dv Q <= 2 g s W + Sigma L B,   L (P + Q) Z <= 2 lam
#   =>  across <= s W c1 + L W c2 + u (s c3 + L c4)       (then W <= u D F)
c1 = 2 * g + 2 * g * b1 * lam_max * inv1mB
c2 = b1 * (1 + u1) * inv1mB
c3 = 2 * g * lam_max * b0 * inv1mB / u
c4 = (b0 * (1 + u1) * inv1mB + 2 * u1) / u
c4 += u1 / u                          # the triangle of the f32 edges against the box of the vertices: |e1 - (v1 - v0)| <= u1 P
#          along.  D |t^ - t*| <= g_t s W + D |t*| B,   D |t*| = |Q* - o| <= s + Sigma L
#   =>  D |t^ - t*| <= s W c1t + L W c2t + u (s c3t + L c4t)
c1t = g_t + b1 + 2 * g * b1 * lam_max * inv1mB
c2t = b1 * (1 + u1) * inv1mB
c3t = (b0 + 2 * g * lam_max * b0 * inv1mB) / u
c4t = b0 * (1 + u1) * inv1mB / u

# ---------------------------------------------------------------------------------------------------------------------
# The kernel's side (chunk_child): mm = S (KP f + KS), dt = S (KT f + KD) with f >= F, S = 1.001 sqrt(...) + half extents,
# slab values  t = fl(fl(lo - o) fl(1 / d))  (three roundings),  tn = fma(-mm, |1 / d|, n)  (one more; mm |1 / d| carries the
# reciprocal's), and fl(tn - dt) in the last comparison.  In space (times |d_axis|) a slab plane is misplaced by at most
#   |lo - o| (gamma3 + u (1 + gamma3)) + mm (2 u + ...)  <=  k_slab u S,     mm <= S (KP 1.5e5 + KS)
KP, KS, KT, KD = 12 * u * Fr(1009, 1000), 24 * u * Fr(1009, 1000), 11 * u * Fr(1009, 1000), 16 * u * Fr(1009, 1000)   # the f32 constants are >= 1.009 x
mm_over_S = KP * 150000 + KS
k_slab = (gamma(3) + u * (1 + gamma(3))) / u + mm_over_S * (2 * u + gamma(2)) / u
k_cmp = 2 + k_slab                    # along: the entry it compares (k_slab) and fl(tn - dt) itself (|t| D <= 2 S)

# S >= s + L / 2 (the box holds v0, and half its extents hold L / 2) and S >= L (the farthest corner of a box is at least half its
# diagonal away from ANY point, the half extents add at least another half, and the box holds two vertices L apart):
#   sup over s, L of (x s + y L) / max(s + L / 2, L) = max(x / 2 + y, x)
def worst(x, y):
    return max(x / 2 + y, x)


checks = [
    ("across, F part, |o - v0| term", c1 * D_max, KP / u),
    ("across, F part, L term (S >= L / 2 + ...)", c2 * D_max, KP / u / 2),
    ("across, u part + slab arithmetic", worst(c3, c4) + k_slab, KS / u),
    ("along,  F part, |o - v0| term", c1t / D_min, KT / u),
    ("along,  F part, L term", c2t / D_min, KT / u / 2),
    ("along,  u part + comparison", worst(c3t, c4t) / D_min + k_cmp, KD / u),
]

# LEMMA cone (H3 for a triangle too large for the determinant floor alone; rb_bvh.cpp pack_fac stores the cone bound with L^2):
#   the kernel enters unless F_L = (L^2 / N) / (0.95 lb) <= 1.5e5, lb <= |cos(d, n)|.  |a| = D N |cos| >= D N lb, so
#   L^2 <= 1.5e5 0.95 |a| / D  =>  |a^ - a| <= alpha1 D L^2 + alpha2 |a| <= eta |a|  =>  |a^| >= (1 - eta) |a|
#   =>  L^2 / |a^| <= 1.5e5 0.95 / ((1 - eta) D)  <=  1.5e5     (H3, not assumed)      and  F_L >= L^2 / |a^| as claimed if (1 - eta) D >= 0.95
eta = alpha1 * Fr(150000) * Fr(95, 100) + alpha2
checks.append(("cone bound: (1 - eta) |d| >= 0.95", Fr(95, 100), (1 - eta) * D_min))
checks.append(("determinant floor: |a^| >= 0.95 |a| under H3", 1 / kappa(lam_max), None))


# =====================================================================================================================
# The sphere walks (rb_device_shade.hpp sphere_child, kSphK): intersect_sphere's discriminant, shader.wgsl:193-199,
#   oc^ = fl(o - c);  hb^ = dot3(oc^, d);  cc^ = fl(dot3(oc^, oc^) - fl(r r));  disc^ = fl(fl(hb^ hb^) - fl(a^ cc^)),  a^ = dot3(d, d)
# |disc^ - disc*| <= E u a Dm^2 with Dm = max(|o - c|, r), a = |d|^2.  Everything below in units of a Dm^2 (|oc| <= Dm, r <= Dm).
E_oc = u                                                    # |oc^ - oc| <= u |oc|
N_oc = 1 + u
# hb: rounding of the dot product on (oc^, d) with |r| <= |hb| + propagation, propagation |oc^ - oc| |d|       [units: Dm |d|]
prop_hb = E_oc
E_hb = (gamma(2) * N_oc + u1 * (1 + prop_hb)) / (1 - u1) + prop_hb
# hb^2: |hb^^2 - hb^2| <= 2 |hb| E_hb + E_hb^2, then one rounding of the product                                   [units: a Dm^2]
E_hb2 = (2 * E_hb + E_hb * E_hb) * (1 + u) + u
# oc^ . oc^ against |oc|^2: rounding (gamma2 + u1) |oc^|^2 / (1 - u1), propagation (2 u + u^2) |oc|^2                 [units: Dm^2]
E_oo = (gamma(2) + u1) * N_oc * N_oc / (1 - u1) + (2 * u + u * u)
# cc^ = fl(oo^ - fl(r r)): u r^2 for the product, u |oo^ - rr^| <= u (1 + E_oo + u) max(|oc|^2, r^2) for the difference
E_cc = E_oo + u + u * (1 + E_oo + u)
# a^ cc^: |a^ - a| <= (gamma2 + u1) / (1 - u1) a;  |cc*| <= Dm^2;  one rounding of the product
E_a = (gamma(2) + u1) / (1 - u1)
E_ac = (E_a * 1 + (1 + E_a) * E_cc) * (1 + u) + u * (1 + E_cc)
# the final difference: both errors, and one rounding of |hb^^2 - (a cc)^| <= (1 + E_hb2) + (1 + E_ac)
E_disc = E_hb2 + E_ac + u * (2 + E_hb2 + E_ac)
E_sphere = E_disc / u
# what the walk needs (rb_device_shade.hpp): a reported hit has disc^ >= 0, so b^2 <= r^2 + E u Dm^2 (b: line to centre) and the reported
# root is within sqrt(E u) Dm / |d| of the chord.  Beside that: |hb^ - hb| / a in t_c (E_hb Dm), the two roundings and the division of
# the root (gamma(3) (|t_c| + s) |d| <= gamma(3) 2 Dm), the kernel's slab values (fl(c - o), an approximate reciprocal within 1 ulp, a
# product: gamma(3) + u on |c - o| <= D) and the outward rounding of the stored boxes (inside D's factor 1.0000004): all relative to D >= Dm
import math
sqrtEu = Fr(math.isqrt(int(E_disc * 10 ** 30)) + 1, 10 ** 15)          # > sqrt(E u), a rational upper bound
assert sqrtEu * sqrtEu > E_disc
k_need = sqrtEu + E_hb + 2 * gamma(3) + (gamma(3) + 2 * u)
k_has = Fr(125, 100000) * Fr(1001, 1000)                               # kSphK = 1.25e-3f * 1.001f (|d| = 1 +- 4 u and v_sqrt's ulp are in the 1.001)
k_has_eff = k_has / (1 + 5 * u) / (1 + 2 * u)
checks.append(("sphere: sqrt(E u) + the rest <= kSphK", k_need, k_has_eff))


def main():
    f = lambda x: f"{float(x):.4f}"
    print("u = 2^-24;  units below: u for the constants, 1 for ratios")
    print(f"  cross product   |c^ - c| <= {f(CROSS)} u |x||y|        a^: |a^ - a| <= {f(alpha1 / u)} u |d| A + {f(alpha2 / u)} u |a|")
    print(f"  numerators      s.h: {f(a_u / u)} u s|d||e2|   d.q, e2.q: {f(a_v / u)} u s|d||e1|, .. s A     (+ {f(b_n / u)} u relative)")
    print(f"  H3: L^2 Z <= {f(lam_max)}   B(W) = {f(b1)} W + {f(b0 / u)} u   <= {f(B_max)}    |a| / |a^| <= {f(kappa(lam_max))}")
    print(f"  across:  dist(Q*, box) <= u F |d| ({f(c1)} s + {f(c2)} L) + u ({f(c3)} s + {f(c4)} L)     [DESIGN r03 by hand: 11.2 s + 4.6 L; 10 u L]")
    print(f"  along:   |d||t^ - t*|  <= u F |d| ({f(c1t)} s + {f(c2t)} L) + u ({f(c3t)} s + {f(c4t)} L)     [DESIGN r03 by hand: 10.2 s + 4.6 L; 4 u t^]")
    print(f"  kernel:  slab planes misplaced by <= {f(k_slab)} u S;  the along comparison loses <= {f(k_cmp)} u S")
    print(f"  spheres: |disc^ - disc*| <= {f(E_sphere)} u a max(|o - c|, r)^2   (assumed < 26 by kSphK; sampled: 8.9)   sqrt(E u) = {float(sqrtEu):.4e}")
    ok = True
    print("\n  check                                              needs      has        used")
    for name, need, has in checks:
        if has is None:
            good = need >= Fr(95, 100)
            print(f"  {name:50s} {f(need)}  >= 0.95    {'ok' if good else 'FAIL'}")
        else:
            good = need <= has
            fmt = (lambda x: f"{float(x):.4e}") if float(has) < 0.01 else f
            print(f"  {name:50s} {fmt(need):>8s}   {fmt(has):>8s}   {float(need / has):6.3f}   {'ok' if good else 'FAIL'}")
        ok &= good
    print("\nPASS: every accepted hit lies inside what chunk_child keeps, for every ray and triangle that meet H1-H4;\n      every reported sphere hit inside what sphere_child keeps (normalised directions, no underflow)" if ok else "\nFAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
