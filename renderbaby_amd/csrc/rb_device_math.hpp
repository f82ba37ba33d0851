// rb_device_math.hpp -- vec3, IEEE-exact fast reciprocal / division / sqrt, conversions, PCG RNG, colour output.
// Part of the single device translation unit rb_kernels.hip (numerics contract: see there).
#pragma once
#include "rb_device_common.hpp"

#pragma clang fp contract(off)

namespace rb {
namespace {

// ------------------------------------------------------------------ vec3 --
struct f3 {
    float x, y, z;
};
DEV f3 mk(float x, float y, float z) { return f3{x, y, z}; }
DEV f3 ld3(const float* p) { return f3{p[0], p[1], p[2]}; }
DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEV f3 divs(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
DEV float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV f3 div3_exact(f3 a, float b);
DEV float sqrt_exact(float x);
DEV f3 normalize(f3 a) { return div3_exact(a, sqrt_exact(dot(a, a))); }

// ---- IEEE-exact 1/b in 3-5 instructions instead of the 12-instruction a/b expansion.
// v_rcp_f32 is accurate to 1 ulp; one (RB_RCP_STEPS=1) or two Newton steps with FMA give the
// correctly rounded reciprocal for every significand except a few (e.g. all ones), which is a
// property of the significand alone as long as b and 1/b are normal.  Lanes outside
// [2^-100, 2^100] or with a significand the exhaustive device check (rb_debug_rcp_exhaustive,
// tests/test_gpu_parity.py::test_fast_reciprocal_is_exhaustively_exact) has not cleared fall back
// to the compiler's division, so the result is `1.0f / b` bit for bit in every case.
#ifndef RB_RCP_STEPS
#define RB_RCP_STEPS 1
#endif
#ifndef RB_FAST_RCP
#define RB_FAST_RCP 1
#endif
DEV float rcp_newton(float b) {
    float r = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
#if RB_RCP_STEPS >= 2
    e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
#endif
    return r;
}
DEV bool rcp_safe(float b) {
    const uint32_t x = __float_as_uint(b) & 0x7FFFFFFFu;
    // 2^-100 <= |b| < 2^100 and significand not all ones
    return (x - 0x0D800000u) < 0x64000000u && (x & 0x007FFFFFu) != 0x007FFFFFu;
}
DEV float rcp_exact(float b) {
#if RB_FAST_RCP
    if (rcp_safe(b)) return rcp_newton(b);
#endif
    return 1.0f / b;
}

// ---- IEEE-exact a/b from the exact reciprocal: q0 = RN(a*y), r = a - b*q0 (exact in an FMA),
// q = RN(q0 + r*y) with y = RN(1/b).  Whether q is the correctly rounded quotient depends only on
// the two significands while a, b, a/b and r stay clear of the subnormal range; the device check
// rb_debug_div_exhaustive walked ALL 2^23 x 2^23 significand pairs with zero mismatches
// (profiles/r01_div_exhaustive_2p46.log; sampled again by the test suite).  Used where one
// denominator serves three numerators (normalize), so the range checks amortise; anything
// outside the checked ranges takes the compiler's division, so results never change.
#ifndef RB_FAST_DIV
#define RB_FAST_DIV 1
#endif
DEV float div_newton(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r, y, q0);
}
// b in [2^-60, 2^60), significand not all ones
DEV bool div_safe_den(float b) {
    const uint32_t x = __float_as_uint(b) & 0x7FFFFFFFu;
    return (x - 0x21800000u) < 0x3C000000u && (x & 0x007FFFFFu) != 0x007FFFFFu;
}
// v / len for len = sqrt(dot(v, v)) (normalize).  len < 2^59 bounds every |component| below 2^60
// (anything larger would have made len infinite); a non-zero component must be >= 2^-100 in
// magnitude so that q0 and the exact remainder stay representable.  A zero numerator keeps its
// sign through the final copysign, which is also the sign of every non-zero quotient (len > 0).
DEV f3 div3_exact(f3 a, float b) {
#if RB_FAST_DIV
    // (x << 1) - 2 wraps a zero to 0xFFFFFFFE, so the unsigned minimum flags only 0 < |x| < 2^-100
    const uint32_t tx = (__float_as_uint(a.x) << 1) - 2u, ty = (__float_as_uint(a.y) << 1) - 2u,
                   tz = (__float_as_uint(a.z) << 1) - 2u;
    const bool num_ok = min(min(tx, ty), tz) >= (0x0D800000u << 1) - 2u;
    const uint32_t xb = __float_as_uint(b);  // b >= 0: sign bit clear unless -0 / NaN payloads
    const bool den_ok = (xb - 0x21800000u) < 0x3B800000u && (xb & 0x007FFFFFu) != 0x007FFFFFu;  // [2^-60, 2^59)
    if (num_ok && den_ok) {
        const float y = rcp_newton(b);
        const float qx = div_newton(a.x, b, y), qy = div_newton(a.y, b, y), qz = div_newton(a.z, b, y);
        return mk(__builtin_copysignf(qx, a.x), __builtin_copysignf(qy, a.y), __builtin_copysignf(qz, a.z));
    }
#endif
    return mk(a.x / b, a.y / b, a.z / b);
}
// ---- IEEE-exact sqrt without the subnormal / zero / infinity handling of the compiler's
// expansion: v_sqrt_f32 (1 ulp), then pick among s-1ulp, s, s+1ulp by the sign of the exact
// residuals x - s_lo*s and x - s_hi*s (the same selection the compiler emits).  Valid for
// x in [2^-60, 2^60); checked for all 2^23 significands at an even and an odd exponent by
// rb_debug_rcp_exhaustive (mode 1).  Everything else takes sqrtf.
#ifndef RB_FAST_SQRT
#define RB_FAST_SQRT 1
#endif
DEV float sqrt_newton(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_lo = __uint_as_float(__float_as_uint(s) - 1u);
    const float s_hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_lo = __builtin_fmaf(-s_lo, s, x);
    const float r_hi = __builtin_fmaf(-s_hi, s, x);
    float out = (r_lo <= 0.0f) ? s_lo : s;
    out = (r_hi > 0.0f) ? s_hi : out;
    return out;
}
DEV float sqrt_exact(float x) {
#if RB_FAST_SQRT
    if ((__float_as_uint(x) - 0x21800000u) < 0x3C000000u) return sqrt_newton(x);  // positive, [2^-60, 2^60)
#endif
    return sqrtf(x);
}

// 1/a for the triangle test: the reference rejects |a| < 1e-6 first (its reciprocal is never used),
// so only the upper range and the significand need checking.
DEV float rcp_tri(float a) {
#if RB_FAST_RCP
    const uint32_t x = __float_as_uint(a) & 0x7FFFFFFFu;
    if (x < 0x71800000u && (x & 0x007FFFFFu) != 0x007FFFFFu) return rcp_newton(a);
#endif
    return 1.0f / a;
}

// WGSL u32(f32) / i32(f32): truncate + saturate, NaN -> 0
DEV uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
DEV int32_t f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)f;
}

// ------------------------------------------------------------------- RNG --
// shader.wgsl:417-421
DEV uint32_t pcg(uint32_t seed) {
    uint32_t state = seed * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
// shader.wgsl:423-426
DEV float rnd(uint32_t& seed) {
    seed = pcg(seed);
    return (float)seed / 4294967296.0f;
}
// shader.wgsl:429-446
DEV f3 random_unit_vector(uint32_t& seed) {
    f3 p;
    for (;;) {
        float px = rnd(seed) * 2.0f - 1.0f;
        float py = rnd(seed) * 2.0f - 1.0f;
        float pz = rnd(seed) * 2.0f - 1.0f;
        p = mk(px, py, pz);
        if (dot(p, p) < 1.0f) break;
    }
    return normalize(p);
}

// --------------------------------------------------------- colour output --
DEV float linear_to_gamma(float c) { return (c > 0.0f) ? sqrtf(c) : 0.0f; }  // :137-142
DEV uint32_t color_map(f3 c) {                                                // :144-151
    uint32_t r = f2u(linear_to_gamma(c.x) * 255.999f);
    uint32_t g = f2u(linear_to_gamma(c.y) * 255.999f);
    uint32_t b = f2u(linear_to_gamma(c.z) * 255.999f);
    return (255u << 24) | (b << 16) | (g << 8) | r;
}
DEV f3 hash_to_color(uint32_t n) {  // :394-400
    uint32_t h = n * 2654435761u;
    return mk((float)(h % 41u) / 40.0f, (float)(h % 29u) / 28.0f, (float)(h % 19u) / 18.0f);
}

}  // namespace
}  // namespace rb
