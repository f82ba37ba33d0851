// rb_kernels.hip -- gfx950 (CDNA4, wave64) kernels for RenderBaby's path-tracing
// hot path: the per-pixel / per-sample loop of
// crates/engine-pathtracer/src/shader.wgsl (main :673-723, trace_ray :522-662,
// intersect_bvh :282-392, intersect_* :193-280,402-414,664-671, PCG :417-446,
// scatter :459-490, sample_texture :153-191, color_map :137-151).
//
// Numerics contract (bit-exact against oracle/rb_oracle.c; DESIGN.md "Numerics"):
//   every f32 operation is a single IEEE binary32 op, round-to-nearest-even, no FMA
//   contraction (this TU is built with -ffp-contract=off and the pragma below),
//   correctly-rounded / and sqrt (-fhip-fp32-correctly-rounded-divide-sqrt),
//   subnormals kept (hipcc default float mode), dot = (x*x' + y*y') + z*z'.
//
// No MFMA: the work is branchy traversal and 3-wide dot products.  What matters
// here is lane utilisation (path regeneration keeps all 64 lanes on the
// intersection code), scalar/LDS residency of the scene, and 16-byte loads.
#include <hip/hip_runtime.h>

#include "rb_internal.hpp"

#pragma clang fp contract(off)

#define DEV __device__ __forceinline__

// Cost-attribution builds (tools/ablate.sh): RB_ABLATE=n repeats one stage on perturbed-but-equal
// inputs and folds the result into nothing observable, so (time[n] - time[0]) is that stage's cost.
#ifndef RB_ABLATE
#define RB_ABLATE 0
#endif

// Scene data is immutable for the duration of a launch.  Reading it through the
// constant address space lets the compiler use scalar loads (s_load_*: one
// fetch per wavefront, operands land in SGPRs) whenever the address is
// wave-uniform -- spheres, lights, a single-leaf BVH -- and ordinary vector loads
// otherwise.  Without this every lane issues its own VMEM load of the same
// address, because the kernels also store (accumulation, traversal stack).
#define RB_CONST __attribute__((address_space(4)))
template <class T>
DEV const RB_CONST T* cptr(const T* p) {
    return (const RB_CONST T*)p;
}
// native vector types: HIP's float4/uint4 classes cannot be copy-constructed from address space 4
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef const RB_CONST v4f* cf4p;
typedef const RB_CONST v4u* cu4p;
typedef v4f nt_f4;  // nontemporal builtins want a native vector

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

// -------------------------------------------------------------- textures --
// shader.wgsl:153-191.  pow(c, 2.2) over the 256 possible channel values is a
// host-computed table (same libm as the oracle), so textured hits stay bit-exact.
DEV f3 sample_texture(const KParams& p, int32_t index, float uvx, float uvy) {
    if (index < 0) {
        if (p.u.checkerboard_enabled > 0u) {
            int32_t u2 = f2i(floorf(uvx * 10.0f));
            int32_t v2 = f2i(floorf(uvy * 10.0f));
            int32_t sum = (int32_t)((uint32_t)u2 + (uint32_t)v2);
            return (sum % 2 == 0) ? ld3(p.u.checkerboard_color_1) : ld3(p.u.checkerboard_color_2);
        }
        return mk(1.0f, 1.0f, 1.0f);
    }
    if ((uint32_t)index >= p.n_tex) return mk(0.0f, 0.0f, 0.0f);
    const v4u iw = ((cu4p)p.tex_info)[index];
    rb_texture_info info;
    info.offset = iw.x;
    info.width = iw.y;
    info.height = iw.z;
    float u = uvx - floorf(uvx);
    float v = uvy - floorf(uvy);
    uint32_t x = min(f2u(u * (float)info.width), info.width - 1u);
    uint32_t y = min(f2u((1.0f - v) * (float)info.height), info.height - 1u);
    const uint32_t pixel = cptr(p.tex_data)[info.offset + y * info.width + x];
    const RB_CONST float* lut = cptr(p.srgb_lut);
    return mk(lut[pixel & 255u], lut[(pixel >> 8) & 255u], lut[(pixel >> 16) & 255u]);
}

// ---------------------------------------------------------- intersection --
// shader.wgsl:193-215 / :217-239
DEV float isect_sphere(f3 o, f3 d, float a, f3 center, float radius) {
    f3 oc = o - center;
    float half_b = dot(oc, d);
    float c = dot(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return -1.0f;
    float sqrtd = sqrtf(disc);
    float root = (-half_b - sqrtd) / a;
    if (root <= 0.001f) {
        root = (-half_b + sqrtd) / a;
        if (root <= 0.001f) return -1.0f;
    }
    return root;
}

// shader.wgsl:248-280 with edge1/edge2 supplied (v1 - v0, v2 - v0).  RB_TRI_BRANCHFREE=1
// evaluates everything and folds the four early returns into one predicate (same
// comparisons on the same values); measured slower on gfx950 (34.8 vs 33.5 ms on C2-short)
// because the early-outs do skip whole-wave work, so the branchy form is the default.
#ifndef RB_TRI_BRANCHFREE
#define RB_TRI_BRANCHFREE 0
#endif
DEV float isect_triangle(f3 o, f3 d, f3 v0, f3 edge1, f3 edge2, float& uo, float& vo) {
#if RB_TRI_BRANCHFREE
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    const float f = rcp_tri(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    const float t = f * dot(edge2, q);
    const bool miss = (fabsf(a) < 1e-6f) | (u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | !(t > 0.0f);
    uo = u;
    vo = v;
    return miss ? -1.0f : t;
#else
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    if (fabsf(a) < 1e-6f) return -1.0f;
    const float f = rcp_tri(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    const float t = f * dot(edge2, q);
    if (t > 0.0f) {
        uo = u;
        vo = v;
        return t;
    }
    return -1.0f;
#endif
}

// shader.wgsl:664-671 with inv_dir = 1/dir hoisted per ray (pure function of dir)
DEV bool isect_aabb(f3 o, f3 inv, f3 bmin, f3 bmax) {
    f3 t0 = (bmin - o) * inv;
    f3 t1 = (bmax - o) * inv;
    float tmin = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
    float tmax = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
    return tmax >= fmaxf(tmin, 0.0f);
}

// shader.wgsl:402-414
DEV float isect_ground(f3 o, f3 d, float ground_height) {
    if (fabsf(d.y) < 1e-6f) return -1.0f;
    float t = (ground_height - o.y) / d.y;
    return (t > 0.0f) ? t : -1.0f;
}

// ------------------------------------------------------------ statistics --
template <bool STATS>
struct Tally {
    uint32_t segments = 0, paths = 0;
    unsigned long long nodes = 0, tris = 0, spheres = 0, lights = 0, mesh_hits = 0;
};
template <>
struct Tally<false> {
    uint32_t segments = 0, paths = 0;
};

DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <bool STATS>
DEV void flush_tally(const Tally<STATS>& t, unsigned long long* counters) {
    unsigned long long seg = wave_sum((unsigned long long)t.segments);
    unsigned long long pth = wave_sum((unsigned long long)t.paths);
    const bool lead = (__lane_id() == 0);
    if (lead) {
        atomicAdd(&counters[C_SEGMENTS], seg);
        atomicAdd(&counters[C_PATHS], pth);
    }
    if constexpr (STATS) {
        unsigned long long a = wave_sum(t.nodes), b = wave_sum(t.tris), c = wave_sum(t.spheres),
                           d = wave_sum(t.lights), e = wave_sum(t.mesh_hits);
        if (lead) {
            atomicAdd(&counters[C_NODES], a);
            atomicAdd(&counters[C_TRIS], b);
            atomicAdd(&counters[C_SPHERES], c);
            atomicAdd(&counters[C_LIGHTS], d);
            atomicAdd(&counters[C_MESH_HITS], e);
        }
    }
}

// -------------------------------------------------------- BVH traversal --
struct TriHit {
    float t, u, v;
    uint32_t slot;  // position in bvh_indices (prepared-triangle index)
    bool hit;
};

// shader.wgsl:282-392.  Same visit order (left pushed first, right popped first),
// same strict `t > 0.001 && t < hit.t` acceptance, so the winner is the same
// triangle.  Shading data of shader.wgsl:350-372 depends only on the final
// winner and is produced afterwards (tri_shade).  `stack` is this lane's column
// of an LDS array [kStackDepth][blockDim]; the host has verified that the tree
// fits (rb_bvh.cpp).
DEV void test_slot(const v4f a, const v4f b, const v4f c, uint32_t slot, f3 o, f3 d, TriHit& h) {
    float u, v;
    const float t = isect_triangle(o, d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), u, v);
    if (t > 0.001f && t < h.t) {
        h.hit = true;
        h.t = t;
        h.u = u;
        h.v = v;
        h.slot = slot;
    }
}

#ifndef RB_TRI_PAIRS
#define RB_TRI_PAIRS 1
#endif
// Two triangles per step with packed f32 math (v_pk_mul_f32 / v_pk_add_f32: two IEEE
// binary32 operations per lane per instruction).  Element 0 is the triangle at `slot`,
// element 1 the one at `slot + 1`; every element goes through exactly the operations of
// isect_triangle (shader.wgsl:248-280), so each t, u, v is bit-identical to the one-at-a-time
// form, and the two candidates are offered to the closest-hit test in slot order.
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool STATS>
DEV void test_pair(const v4f a0, const v4f b0, const v4f c0, bool ok0, const v4f a1, const v4f b1, const v4f c1,
                   bool ok1, uint32_t slot, f3 o, f3 d, TriHit& h, Tally<STATS>& tl) {
    const f2 v0x = {a0.x, a1.x}, v0y = {a0.y, a1.y}, v0z = {a0.z, a1.z};
    const f2 e1x = {b0.x, b1.x}, e1y = {b0.y, b1.y}, e1z = {b0.z, b1.z};
    const f2 e2x = {c0.x, c1.x}, e2y = {c0.y, c1.y}, e2z = {c0.z, c1.z};
    // h = cross(d, edge2)
    const f2 hx = d.y * e2z - d.z * e2y;
    const f2 hy = d.z * e2x - d.x * e2z;
    const f2 hz = d.x * e2y - d.y * e2x;
    const f2 a = (e1x * hx + e1y * hy) + e1z * hz;
    const f2 f = {rcp_tri(a.x), rcp_tri(a.y)};
    const f2 sx = o.x - v0x, sy = o.y - v0y, sz = o.z - v0z;
    const f2 u = f * ((sx * hx + sy * hy) + sz * hz);
    // q = cross(s, edge1)
    const f2 qx = sy * e1z - sz * e1y;
    const f2 qy = sz * e1x - sx * e1z;
    const f2 qz = sx * e1y - sy * e1x;
    const f2 v = f * ((d.x * qx + d.y * qy) + d.z * qz);
    const f2 t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    const f2 uv = u + v;
    const bool hit0 = ok0 && !(fabsf(a.x) < 1e-6f) && !(u.x < 0.0f) && !(u.x > 1.0f) && !(v.x < 0.0f) && !(uv.x > 1.0f) &&
                      (t.x > 0.0f);
    const bool hit1 = ok1 && !(fabsf(a.y) < 1e-6f) && !(u.y < 0.0f) && !(u.y > 1.0f) && !(v.y < 0.0f) && !(uv.y > 1.0f) &&
                      (t.y > 0.0f);
    if (hit0 && t.x > 0.001f && t.x < h.t) {
        h.hit = true;
        h.t = t.x;
        h.u = u.x;
        h.v = v.x;
        h.slot = slot;
        if constexpr (STATS) tl.mesh_hits++;
    }
    if (hit1 && t.y > 0.001f && t.y < h.t) {
        h.hit = true;
        h.t = t.y;
        h.u = u.y;
        h.v = v.y;
        h.slot = slot + 1u;
        if constexpr (STATS) tl.mesh_hits++;
    }
}

// Opt-in fast walk (RB_FLAG_FAST_BVH) of the library's own SAH tree over the same triangles
// (rb_bvh.cpp, fast_bvh_build): nearer child first, subtrees skipped when missed or entered
// beyond the best t.  It reproduces the reference walk's winner:
//  * candidates are evaluated with the reference's isect_triangle, so t, u, v are the same bits;
//  * equal t resolves by the triangle's rank in the reference's visit order;
//  * the reference only tests a triangle if every node from the root to its leaf passes
//    intersect_aabb: an improving candidate is accepted only after that chain has been
//    re-checked with the reference's own slab arithmetic on the reference's boxes;
//  * boxes are inflated by a margin so rounding cannot cull a triangle the reference would hit.
// Not a proof (an ill-conditioned Moller-Trumbore hit far outside its triangle could be missed),
// which is why it is opt-in; the tests compare it bit for bit with the reference walk.
template <bool STATS>
DEV TriHit intersect_bvh_fast(const KParams& p, f3 o, f3 d, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    TriHit h;
    h.hit = false;
    h.t = 1e20f;
    h.u = 0.0f;
    h.v = 0.0f;
    h.slot = 0u;
    uint32_t best_rank = 0xFFFFFFFFu;
    const f3 inv = mk(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z));
    const cf4p nodes = (cf4p)p.fast_nodes;
    const cf4p ftris = (cf4p)p.fast_tris;
    const cf4p rnodes = (cf4p)p.nodes;
    const RB_CONST uint32_t* fslots = cptr(p.fast_slots);
    const RB_CONST uint32_t* meta = cptr(p.slot_meta);
    const RB_CONST uint32_t* parent = cptr(p.ref_parent);
    const float m = p.fast_margin;
    // S: farthest the ray origin can be from any point of the mesh (>= |origin - v0| for every
    // triangle).  A child's box is inflated by m + 0.01 * S * (largest |e1||e2| below it): how far
    // from its triangle a Moller-Trumbore hit with |a| >= 4.2e-5 can be reported (see rb_bvh.cpp)
    const f3 fb0 = ld3(p.fast_bmin), fb1 = ld3(p.fast_bmax);
    const float sx_ = fmaxf(fabsf(o.x - fb0.x), fabsf(o.x - fb1.x)), sy_ = fmaxf(fabsf(o.y - fb0.y), fabsf(o.y - fb1.y)),
                sz_ = fmaxf(fabsf(o.z - fb0.z), fabsf(o.z - fb1.z));
    const float S = 0.01f * sqrtf(sx_ * sx_ + sy_ * sy_ + sz_ * sz_);

    auto entry = [&](v4f lo, v4f hi, float amax, float& tn) -> bool {
        const float mm = m + S * amax;
        const f3 t0 = (mk(lo.x - mm, lo.y - mm, lo.z - mm) - o) * inv;
        const f3 t1 = (mk(hi.x + mm, hi.y + mm, hi.z + mm) - o) * inv;
        tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
        const float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
        return !(tf < fmaxf(tn, 0.0f)) && !(tn > h.t);
    };
    auto reference_would_test = [&](uint32_t leaf_node) -> bool {
        // Shortcut: if the ray passes through the reference LEAF's box shrunk by m on every side
        // (m is far above the rounding error of a slab test), it passes through the interior of
        // every ancestor's box, so each of the reference's slab tests succeeds; only a ray that
        // merely grazes the leaf box needs the exact walk up the chain.
        {
            const v4f n0 = rnodes[leaf_node * 3u], n1 = rnodes[leaf_node * 3u + 1u];
            if constexpr (STATS) tl.nodes++;
            const bool thick = (n1.x - n0.x >= 2.0f * m) && (n1.y - n0.y >= 2.0f * m) && (n1.z - n0.z >= 2.0f * m);
            const f3 t0 = (mk(n0.x + m, n0.y + m, n0.z + m) - o) * inv;
            const f3 t1 = (mk(n1.x - m, n1.y - m, n1.z - m) - o) * inv;
            const float tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
            const float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
            const bool finite = (t0.x == t0.x) && (t0.y == t0.y) && (t0.z == t0.z) && (t1.x == t1.x) && (t1.y == t1.y) &&
                                (t1.z == t1.z);
            if (thick && finite && tf >= fmaxf(tn, 0.0f)) return true;
        }
        uint32_t n = leaf_node;
        for (;;) {
            const v4f n0 = rnodes[n * 3u], n1 = rnodes[n * 3u + 1u];
            if constexpr (STATS) tl.nodes++;
            if (!isect_aabb(o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) return false;
            if (n == 0u) return true;
            n = parent[n];
        }
    };
    auto leaf = [&](uint32_t ref) {
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 3u) + 1u;
        for (uint32_t j = first; j < first + count; j++) {
            const v4f a = ftris[j * 4u], b = ftris[j * 4u + 1u], c = ftris[j * 4u + 2u];
            if constexpr (STATS) tl.tris++;
            float u, v;
            const float t = isect_triangle(o, d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), u, v);
            if (t > 0.001f && !(t > h.t)) {
                const uint32_t slot = fslots[j];
                const uint32_t leaf_node = meta[slot * 2u], rank = meta[slot * 2u + 1u];
                if ((t < h.t || rank < best_rank) && reference_would_test(leaf_node)) {
                    h.hit = true;
                    h.t = t;
                    h.u = u;
                    h.v = v;
                    h.slot = slot;
                    best_rank = rank;
                    if constexpr (STATS) tl.mesh_hits++;
                }
            }
        }
    };

    uint32_t cur = p.fast_root;
    int sp = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            leaf(cur);
            if (sp == 0) break;
            sp--;
            cur = stack[sp * stride];
            continue;
        }
        const v4f l0 = nodes[cur * 4u], l1 = nodes[cur * 4u + 1u], r0 = nodes[cur * 4u + 2u], r1 = nodes[cur * 4u + 3u];
        if constexpr (STATS) tl.nodes++;
        const uint32_t lref = __float_as_uint(l0.w), rref = __float_as_uint(l1.w);
        float tl_, tr_;
        const bool hl = entry(l0, l1, r0.w, tl_), hr = entry(r0, r1, r1.w, tr_);
        if (hl && hr) {
            const bool left_first = !(tr_ < tl_);
            stack[sp * stride] = left_first ? rref : lref;
            sp++;
            cur = left_first ? lref : rref;
        } else if (hl) {
            cur = lref;
        } else if (hr) {
            cur = rref;
        } else {
            if (sp == 0) break;
            sp--;
            cur = stack[sp * stride];
        }
    }
    return h;
}

template <bool STATS>
DEV TriHit intersect_bvh(const KParams& p, f3 o, f3 d, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    TriHit h;
    h.hit = false;
    h.t = 1e20f;
    h.u = 0.0f;
    h.v = 0.0f;
    h.slot = 0u;
    const uint32_t node_count = p.u.bvh_node_count;
    if (node_count == 0u) return h;
    if (p.fast_nodes != nullptr) return intersect_bvh_fast<STATS>(p, o, d, stack, stride, tl);
    const f3 inv = mk(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z));
    const cf4p nodes = (cf4p)p.nodes;
    const cf4p ptris = (cf4p)p.ptris;  // 4 x float4 per triangle

    if (node_count == 1u) {
        // Single-node tree (the Cornell box): no stack; the node and its triangles are
        // wave-uniform, so they are fetched with scalar loads and every lane that is
        // inside the box walks the same primitive list.  Same tests, same order.  The next
        // triangle's record is requested before the current one is tested.
        const v4f n0 = nodes[0], n1 = nodes[1];
        const v4u n2 = ((cu4p)p.nodes)[2];
        if constexpr (STATS) tl.nodes++;
        const uint32_t first = n2.z, count = n2.w;
        const uint32_t end = (first + count < p.index_len) ? first + count : p.index_len;  // guard :331
        if (first < end && isect_aabb(o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) {
#if RB_TRI_PAIRS
            uint32_t slot = first;
            cf4p tp = ptris + (size_t)first * 4u;  // running pointer: one scalar add per step, immediate offsets
            for (; slot + 2u <= end; slot += 2u, tp += 8) {
                const v4f a0 = tp[0], b0 = tp[1], c0 = tp[2];
                const v4f a1 = tp[4], b1 = tp[5], c1 = tp[6];
                const bool ok0 = __float_as_uint(c0.w) != 0u, ok1 = __float_as_uint(c1.w) != 0u;
                if constexpr (STATS) tl.tris += (ok0 ? 1u : 0u) + (ok1 ? 1u : 0u);
                test_pair(a0, b0, c0, ok0, a1, b1, c1, ok1, slot, o, d, h, tl);
            }
            for (; slot < end; slot++, tp += 4) {
                const v4f a = tp[0], b = tp[1], c = tp[2];
#else
            for (uint32_t slot = first; slot < end; slot++) {
                const v4f a = ptris[slot * 4u], b = ptris[slot * 4u + 1u], c = ptris[slot * 4u + 2u];
#endif
                if (__float_as_uint(c.w) != 0u) {  // guard :336
                    if constexpr (STATS) tl.tris++;
                    const float before = h.t;
                    test_slot(a, b, c, slot, o, d, h);
                    if constexpr (STATS) tl.mesh_hits += (h.t != before) ? 1u : 0u;
                }
            }
        }
        return h;
    }

    int sp = 0;
    stack[0] = 0u;
    sp = 1;
    while (sp > 0) {
        sp--;
        const uint32_t node_idx = stack[sp * stride];
        if (node_idx >= node_count) continue;
        const v4f n0 = nodes[node_idx * 3u], n1 = nodes[node_idx * 3u + 1u];
        const v4u n2 = ((cu4p)p.nodes)[node_idx * 3u + 2u];
        if constexpr (STATS) tl.nodes++;
        if (!isect_aabb(o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) continue;
        const uint32_t left = n2.x, right = n2.y, first = n2.z, count = n2.w;
        if (count > 0u) {
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t slot = first + i;
                if (slot >= p.index_len) continue;
                const v4f a = ptris[slot * 4u], b = ptris[slot * 4u + 1u], c = ptris[slot * 4u + 2u];
                if (__float_as_uint(c.w) == 0u) continue;  // guard :336
                if constexpr (STATS) tl.tris++;
                const float before = h.t;
                test_slot(a, b, c, slot, o, d, h);
                if constexpr (STATS) tl.mesh_hits += (h.t != before) ? 1u : 0u;
            }
        } else {
            if (left < node_count) {
                stack[sp * stride] = left;
                sp++;
            }
            if (right < node_count) {
                stack[sp * stride] = right;
                sp++;
            }
        }
    }
    return h;
}

DEV float uv_at(const KParams& p, uint32_t i) { return (i < p.n_uvs) ? cptr(p.uvs)[i] : 0.0f; }

// shader.wgsl:353-361
DEV void tri_uv(const KParams& p, const TriHit& h, float& uvx, float& uvy) {
    const v4u s0 = ((cu4p)p.pshade)[h.slot];
    const uint32_t i0 = s0.x, i1 = s0.y, i2 = s0.z;  // v0_index, v1_index, v2_index
    const float w = 1.0f - h.u - h.v;
    const float uv0x = uv_at(p, i0 * 2u), uv0y = uv_at(p, i0 * 2u + 1u);
    const float uv1x = uv_at(p, i1 * 2u), uv1y = uv_at(p, i1 * 2u + 1u);
    const float uv2x = uv_at(p, i2 * 2u), uv2y = uv_at(p, i2 * 2u + 1u);
    uvx = (w * uv0x + h.u * uv1x) + h.v * uv2x;
    uvy = (w * uv0y + h.u * uv1y) + h.v * uv2y;
}

// ------------------------------------------------------------- materials --
struct Mat {
    f3 diffuse, specular, emissive;
    float fuzz;     // clamp(1 - shininess / 1000, 0, 1)   (shader.wgsl:637)
    bool metal;     // mean(specular) > 0.01 && mean(diffuse) < 0.01   (:615-623)
    int32_t tex;
};
DEV Mat load_mat(const rb_material* m) {
    const cf4p q = (cf4p)m;
    const v4f d = q[1], s = q[2], e = q[3];
    const v4u t = ((cu4p)m)[4];  // opacity, illum, texture_index, pad
    Mat r;
    r.diffuse = mk(d.x, d.y, d.z);
    r.specular = mk(s.x, s.y, s.z);
    r.fuzz = d.w;          // device copy's _pad1, filled by k_prep_materials
    r.emissive = mk(e.x, e.y, e.z);
    r.tex = (int32_t)t.z;
    r.metal = t.w != 0u;   // device copy's _pad2
    return r;
}
DEV int32_t load_tex_index(const rb_material* m) {
    const v4u t = ((cu4p)m)[4];
    return (int32_t)t.z;
}

enum Kind : uint32_t { K_NONE = 0, K_GROUND = 1, K_TRI = 2, K_SPHERE = 3, K_LIGHT = 4 };

struct Path {
    f3 o, d;
    f3 color, att;
    uint32_t seed;
    uint32_t depth;
};

DEV f3 reflect_vector(f3 v, f3 n) { return v - (2.0f * dot(v, n)) * n; }  // :459-461
DEV bool near_zero(f3 v) {                                                // :463-466
    const float s = 1e-8f;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}

// Sphere acceleration structure (rb_bvh.cpp, sphere_bvh_build): closest sphere with the
// semantics of the reference's linear scan (shader.wgsl:574-586).
//  * every candidate is evaluated with the reference's exact intersect_sphere;
//  * the scan accepts `t > 0.001 && t < closest.t` in index order, i.e. the winner is the
//    smallest t below the incoming closest_t, ties going to the lowest index: here
//    `t < best || (t == best && id < best_id)`;
//  * a subtree is skipped only if the ray misses its box inflated by m, or enters it beyond
//    best_t.  m covers the rounding error of the reference's own arithmetic: its discriminant
//    hb^2 - a*(|oc|^2 - r^2) carries an absolute error <= 16 u a |oc|^2 (u = 2^-24), so a sphere
//    can be reported hit by a ray passing up to sqrt(r^2 + 1e-6 D^2) from its centre and the
//    reported t can be early by about the same amount; D = the farthest the ray origin can be
//    from any sphere.  m = 3e-3 * D (> 2 * sqrt(1e-6) * D) bounds both.
DEV void intersect_spheres_bvh(const KParams& p, f3 o, f3 d, float a, float& closest_t, uint32_t& sphere_idx,
                               uint32_t* stack, uint32_t stride, unsigned long long* n_tested) {
    const cf4p nodes = (cf4p)p.sph_nodes;
    const cf4p leafs = (cf4p)p.sph_leaf;
    const RB_CONST uint32_t* ids = cptr(p.sph_id);
    const f3 bmin = ld3(p.sph_bmin), bmax = ld3(p.sph_bmax);
    const float dx = fmaxf(fabsf(o.x - bmin.x), fabsf(o.x - bmax.x));
    const float dy = fmaxf(fabsf(o.y - bmin.y), fabsf(o.y - bmax.y));
    const float dz = fmaxf(fabsf(o.z - bmin.z), fabsf(o.z - bmax.z));
    const float m = 3e-3f * sqrtf(dx * dx + dy * dy + dz * dz) + 1e-4f;
    const f3 inv = mk(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z));
    float best = closest_t;
    uint32_t best_id = 0xFFFFFFFFu;

    // slab test of a box inflated by m: visit unless missed or entered beyond `best`
    // (comparisons are written so that a NaN means "visit")
    auto entry = [&](v4f lo, v4f hi, float& tn) -> bool {
        const f3 t0 = (mk(lo.x - m, lo.y - m, lo.z - m) - o) * inv;
        const f3 t1 = (mk(hi.x + m, hi.y + m, hi.z + m) - o) * inv;
        tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
        const float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
        return !(tf < fmaxf(tn, 0.0f)) && !(tn > best);
    };
    auto leaf = [&](uint32_t ref) {
        const uint32_t first = ref & 0x0FFFFFFFu, count = ((ref >> 28) & 3u) + 1u;
        for (uint32_t j = first; j < first + count; j++) {
            const v4f cr = leafs[j];
            const uint32_t id = ids[j];
            if (n_tested) (*n_tested)++;
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);
            if (t > 0.001f && (t < best || (t == best && id < best_id))) {
                best = t;
                best_id = id;
            }
        }
    };

    uint32_t cur = p.sph_root;
    int sp = 0;
    for (;;) {
        if (cur & 0x80000000u) {
            leaf(cur);
            if (sp == 0) break;
            sp--;
            cur = stack[sp * stride];
            continue;
        }
        const v4f l0 = nodes[cur * 4u], l1 = nodes[cur * 4u + 1u], r0 = nodes[cur * 4u + 2u], r1 = nodes[cur * 4u + 3u];
        const uint32_t lref = __float_as_uint(l0.w), rref = __float_as_uint(l1.w);
        float tl_, tr_;
        const bool hl = entry(l0, l1, tl_), hr = entry(r0, r1, tr_);
        if (hl && hr) {
            // nearer child first; the other waits on the stack
            const bool left_first = !(tr_ < tl_);
            stack[sp * stride] = left_first ? rref : lref;
            sp++;
            cur = left_first ? lref : rref;
        } else if (hl) {
            cur = lref;
        } else if (hr) {
            cur = rref;
        } else {
            if (sp == 0) break;
            sp--;
            cur = stack[sp * stride];
        }
    }
    if (best_id != 0xFFFFFFFFu) {
        closest_t = best;
        sphere_idx = best_id;
    }
}

// One iteration of the bounce loop, shader.wgsl:534-660.  Returns true when the
// path continues.  The closest-hit search keeps the reference's category order
// (ground, BVH, spheres, lights) and strict comparisons, so ties resolve the
// same way; per-hit data that only the final winner needs (position, normal,
// material, uv) is produced once, after the search.
template <bool STATS>
DEV bool segment_finish(const KParams& p, Path& pt, const TriHit th, uint32_t* stack, uint32_t stride,
                        Tally<STATS>& tl) {
    const f3 o = pt.o, d = pt.d;
    tl.segments++;

    float closest_t = 1e20f;
    uint32_t kind = K_NONE;
    // state of closest_hit.uv / use_texture after the ground + BVH stage
    float uvx = 0.0f, uvy = 0.0f;
    bool use_tex = false;

    // Ground :552-565
    if (p.u.ground_enabled > 0u) {
        const float t = isect_ground(o, d, p.u.ground_height);
        if (t > 0.001f && t < closest_t) {
            closest_t = t;
            kind = K_GROUND;
            const f3 gp = o + t * d;
            uvx = gp.x;
            uvy = gp.z;
            use_tex = true;
        }
    }

    // BVH triangles :568-571 (th: the traversal's winner, produced by the caller)
    const bool tri_won_a = th.hit && th.t < closest_t;  // closest_hit = bvh_hit
    if (tri_won_a) {
        closest_t = th.t;
        kind = K_TRI;
    }

    // Spheres :574-586 and point lights :590-601.  Two passes with the reference's arithmetic:
    // pass 1 evaluates the discriminant of every sphere with wave-uniform scalar loads and
    // records the candidates (disc >= 0) in a per-lane bit mask; pass 2 runs the sqrt/divide
    // tail only for a lane's own candidates, in ascending index order, so the strict `<`
    // keeps the same winner.  Most lanes have no candidate, so the expensive tail is issued
    // once or twice per segment instead of once per sphere.
    const float a = dot(d, d);
    uint32_t sphere_idx = 0xFFFFFFFFu;
    const uint32_t ns = p.u.spheres_count;
    const cf4p sph4 = (cf4p)p.spheres;  // 96 B = 6 x float4 per sphere; [0] = centre, radius
    if (p.sph_nodes != nullptr) {
        unsigned long long* cnt = nullptr;
        if constexpr (STATS) cnt = &tl.spheres;
        intersect_spheres_bvh(p, o, d, a, closest_t, sphere_idx, stack, stride, cnt);
    } else
    for (uint32_t base = 0; base < ns; base += 32u) {
        const uint32_t n = (ns - base < 32u) ? ns - base : 32u;
        uint32_t cand = 0u;
        cf4p sp_ = sph4 + (size_t)base * 6u;
        for (uint32_t k = 0; k < n; k++, sp_ += 6) {
            const v4f cr = sp_[0];
            if constexpr (STATS) tl.spheres++;
            const f3 oc = o - mk(cr.x, cr.y, cr.z);
            const float half_b = dot(oc, d);
            const float c = dot(oc, oc) - cr.w * cr.w;
            const float disc = half_b * half_b - a * c;
            cand |= (disc < 0.0f) ? 0u : (1u << k);
        }
        while (cand != 0u) {
            const uint32_t k = (uint32_t)__ffs((int)cand) - 1u;
            cand &= cand - 1u;
            const v4f cr = sph4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);
            if (t > 0.001f && t < closest_t) {
                closest_t = t;
                sphere_idx = base + k;
            }
        }
    }
    if (sphere_idx != 0xFFFFFFFFu) kind = K_SPHERE;

    uint32_t light_idx = 0xFFFFFFFFu;
    const cf4p lgt4 = (cf4p)p.lights;
    for (uint32_t base = 0; base < p.n_lights; base += 32u) {
        const uint32_t n = (p.n_lights - base < 32u) ? p.n_lights - base : 32u;
        uint32_t cand = 0u;
        for (uint32_t k = 0; k < n; k++) {
            const v4f cr = lgt4[(base + k) * 6u];
            if constexpr (STATS) tl.lights++;
            const f3 oc = o - mk(cr.x, cr.y, cr.z);
            const float half_b = dot(oc, d);
            const float c = dot(oc, oc) - cr.w * cr.w;
            const float disc = half_b * half_b - a * c;
            cand |= (disc < 0.0f) ? 0u : (1u << k);
        }
        while (cand != 0u) {
            const uint32_t k = (uint32_t)__ffs((int)cand) - 1u;
            cand &= cand - 1u;
            const v4f cr = lgt4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);
            if (t > 0.001f && t < closest_t) {
                closest_t = t;
                light_idx = base + k;
            }
        }
    }
    if (light_idx != 0xFFFFFFFFu) kind = K_LIGHT;
#if RB_ABLATE == 2
    {
        f3 o2 = o;
        asm volatile("" : "+v"(o2.x));
        float ct = 1e20f;
        uint32_t si = 0;
        for (uint32_t base = 0; base < ns; base += 32u) {
            const uint32_t n = (ns - base < 32u) ? ns - base : 32u;
            uint32_t cand = 0u;
            for (uint32_t k = 0; k < n; k++) {
                const v4f cr = sph4[(base + k) * 6u];
                const f3 oc = o2 - mk(cr.x, cr.y, cr.z);
                const float half_b = dot(oc, d);
                const float c = dot(oc, oc) - cr.w * cr.w;
                const float disc = half_b * half_b - a * c;
                cand |= (disc < 0.0f) ? 0u : (1u << k);
            }
            while (cand != 0u) {
                const uint32_t k = (uint32_t)__ffs((int)cand) - 1u;
                cand &= cand - 1u;
                const v4f cr = sph4[(base + k) * 6u];
                const float t = isect_sphere(o2, d, a, mk(cr.x, cr.y, cr.z), cr.w);
                if (t > 0.001f && t < ct) {
                    ct = t;
                    si = base + k;
                }
            }
        }
        asm volatile("" ::"v"(ct), "v"(si));
    }
#endif

    // Sky :604-608
    if (kind == K_NONE) {
        pt.color = pt.color + pt.att * ld3(p.u.sky_color);
        return false;
    }

    // ---- resolve the winner's HitRecord fields (:555-563, :348-372, :579-584, :595-599)
    const f3 pos = o + closest_t * d;
    f3 normal = mk(0.0f, 1.0f, 0.0f);
    Mat m;
    m.diffuse = mk(0, 0, 0);
    m.specular = mk(0, 0, 0);
    m.emissive = mk(0, 0, 0);
    m.fuzz = 1.0f;
    m.metal = false;  // ground (diffuse 0.5) and colour-hash triangles (specular 0) are never metal
    m.tex = -1;
    if (tri_won_a) {
        // the BVH hit replaced closest_hit, including uv and use_texture, even if a
        // sphere or light wins later (those never reset uv; lights never reset use_texture)
        const cu4p pr = (cu4p)p.ptris + th.slot * 4u;  // [0].w = tri_id, [1].w = mesh_index
        if (p.u.color_hash_enabled != 0u) {
            use_tex = false;
            if (kind == K_TRI) {
                const v4u p0 = pr[0];
                m.diffuse = hash_to_color(p0.w + 1u);
            }
        } else {
            const v4u p1 = pr[1];
            const rb_material* mm = &p.meshes[p1.w].material;
            if (kind == K_TRI) {
                m = load_mat(mm);
                use_tex = m.tex >= 0;
            } else {
                use_tex = load_tex_index(mm) >= 0;
            }
        }
    }
    if (kind == K_GROUND) {
        m.diffuse = mk(0.5f, 0.5f, 0.5f);
    } else if (kind == K_TRI) {
        const v4f s = ((cf4p)p.ptris)[th.slot * 4u + 3u];
        normal = mk(s.x, s.y, s.z);
    } else {
        if (sphere_idx != 0xFFFFFFFFu) {
            const rb_sphere* s = p.spheres + sphere_idx;
            if (kind == K_SPHERE) {
                m = load_mat(&s->material);
                use_tex = m.tex >= 0;
                const v4f cr = ((cf4p)s)[0];
                normal = normalize(pos - mk(cr.x, cr.y, cr.z));
            } else {
                use_tex = load_tex_index(&s->material) >= 0;
            }
        }
        if (kind == K_LIGHT) {
            const rb_point_light* l = p.lights + light_idx;
            m = load_mat(&l->material);
            const v4f cr = ((cf4p)l)[0];
            normal = normalize(pos - mk(cr.x, cr.y, cr.z));
        }
    }

    // is_metal / fuzz (:615-623,637) are pure functions of the material: evaluated once per
    // material at upload (k_prep_materials) with the shader's arithmetic
    const bool is_metal = m.metal;

    pt.color = pt.color + pt.att * m.emissive;  // :626

#if RB_ABLATE == 3
    {
        uint32_t s2 = pt.seed;
        asm volatile("" : "+v"(s2));
        const f3 r2 = random_unit_vector(s2);
        asm volatile("" ::"v"(r2.x), "v"(r2.y), "v"(r2.z));
    }
#endif
    // Both scatter branches draw exactly one random unit vector (:472, :488) and nothing else
    // touches the seed, so the rejection loop runs once for the whole wavefront instead of once
    // per branch; likewise the final normalize below is shared.
    const f3 ruv = random_unit_vector(pt.seed);
    f3 scattered, albedo;
    bool absorbed = false;
    if (is_metal) {
        const f3 reflected = reflect_vector(normalize(d), normal);
        scattered = reflected + m.fuzz * ruv;
        absorbed = dot(scattered, normal) <= 0.0f;  // :640-642
        albedo = m.specular;
    } else {
        const f3 sd = normal + ruv;
        scattered = near_zero(sd) ? normal : normalize(sd);
        albedo = m.diffuse;
        if (use_tex) {
            if (tri_won_a) tri_uv(p, th, uvx, uvy);
            albedo = albedo * sample_texture(p, m.tex, uvx, uvy);
        }
    }
    if (absorbed) return false;
    pt.att = pt.att * albedo;
    pt.o = pos + 0.001f * normal;
    pt.d = normalize(scattered);
    pt.depth++;
    return pt.depth < p.u.max_depth;
}

// One whole iteration of the bounce loop: traversal + everything else.
template <bool STATS>
DEV bool segment(const KParams& p, Path& pt, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    const TriHit th = intersect_bvh<STATS>(p, pt.o, pt.d, stack, stride, tl);
#if RB_ABLATE == 1
    {
        f3 o2 = pt.o;
        asm volatile("" : "+v"(o2.x));
        Tally<STATS> t2;
        const TriHit th2 = intersect_bvh<STATS>(p, o2, pt.d, stack, stride, t2);
        asm volatile("" ::"v"(th2.t), "v"(th2.slot));
    }
#endif
    return segment_finish<STATS>(p, pt, th, stack, stride, tl);
}

// ----------------------------------------------------------------- camera --
struct Cam {
    f3 pos, right, up, fwd;
    float fov, aspect, wm1, hm1;
};
// shader.wgsl:690,702-708 (per-launch invariants of the sample loop)
DEV Cam make_cam(const KParams& p) {
    Cam c;
    c.aspect = (float)p.u.width / (float)p.u.height;
    c.pos = ld3(p.u.camera.pos);
    c.fwd = normalize(ld3(p.u.camera.dir));
    c.right = normalize(cross(mk(0.0f, 1.0f, 0.0f), c.fwd));
    c.up = cross(c.fwd, c.right);
    c.fov = p.u.camera.pane_width / (2.0f * p.u.camera.pane_distance * c.aspect);
    c.wm1 = (float)(p.u.width - 1u);
    c.hm1 = (float)(p.u.height - 1u);
    return c;
}
// shader.wgsl:693-709
DEV void start_path(const KParams& p, const Cam& c, uint32_t x, uint32_t y, uint32_t pixel_index,
                    uint32_t sample_offset, Path& pt) {
    uint32_t seed = pcg(pixel_index + pcg(sample_offset));
    const float off_x = rnd(seed) - 0.5f;
    const float off_y = rnd(seed) - 0.5f;
    const float u = ((((float)x + off_x) / c.wm1) * 2.0f - 1.0f) * c.aspect;
    const float v = 1.0f - (((float)y + off_y) / c.hm1) * 2.0f;
    pt.o = c.pos;
    pt.d = normalize(((c.fov * u) * c.right + (c.fov * v) * c.up) + c.fwd);
    pt.seed = seed;
    pt.color = mk(0, 0, 0);
    pt.att = mk(1, 1, 1);
    pt.depth = 0;
}

// global image row of local row `ly` (interleaved stripes)
DEV uint32_t global_row(const KParams& p, uint32_t ly) {
    if (p.shard_count <= 1u) return ly;
    const uint32_t s = ly / p.stripe_rows, r = ly % p.stripe_rows;
    return (s * p.shard_count + p.shard_rank) * p.stripe_rows + r;
}

// shader.wgsl:716-722 + the x mirror of gpu_wrapper.rs:446-458
DEV void store_pixel(const KParams& p, uint32_t x, uint32_t ly, f3 acc, uint32_t total_samples) {
    const size_t li = (size_t)ly * p.u.width + x;
    const float ts = (float)total_samples;
    reinterpret_cast<float4*>(p.accum)[li] = make_float4(acc.x, acc.y, acc.z, ts);
    const f3 fin = divs(acc, ts);
    const f3 mapped = mk(fin.x / (fin.x + 1.0f), fin.y / (fin.y + 1.0f), fin.z / (fin.z + 1.0f));
    p.out_rgba[(size_t)ly * p.u.width + (p.u.width - 1u - x)] = color_map(mapped);
}

// ======================================================= kernel: PIXEL ====
// One thread per pixel, 8x8 pixels per wavefront, nested sample / bounce loops:
// the shape of the reference's dispatch (one invocation = one pixel,
// gpu_wrapper.rs:380-384) with all passes of a launch folded into the kernel.
constexpr uint32_t kPixelBlock = 256;

template <bool STATS>
__global__ void __launch_bounds__(kPixelBlock) k_pixel(const KParams p) {
    extern __shared__ uint32_t s_stack[];  // [stack_depth][blockDim.x]; empty for single-node trees
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t tiles_x = (p.u.width + 15u) / 16u;
    const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const uint32_t x = bx * 16u + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t ly = by * 16u + (wave >> 1) * 8u + (lane >> 3);
    Tally<STATS> tl;
    bool active = (x < p.u.width) && (ly < p.local_rows);
    uint32_t y = 0;
    if (active) {
        y = global_row(p, ly);
        active = y < p.u.height;
    }
    if (active) {
        const Cam cam = make_cam(p);
        const uint32_t pixel_index = y * p.u.width + x;
        const float4 a4 = reinterpret_cast<const float4*>(p.accum)[(size_t)ly * p.u.width + x];
        f3 acc = mk(a4.x, a4.y, a4.z);
        uint32_t total = f2u(a4.w);
        for (uint32_t pass = p.first_pass; pass < p.first_pass + p.n_passes; pass++) {
            for (uint32_t s = 0; s < p.samples_per_pass; s++) {
                Path pt;
                start_path(p, cam, x, y, pixel_index, pass * p.samples_per_pass + s, pt);
                if (p.u.max_depth > 0u) {
                    while (segment<STATS>(p, pt, &s_stack[tid], kPixelBlock, tl)) {
                    }
                }
                tl.paths++;
                acc = acc + pt.color;
                total = total + 1u;
            }
        }
        store_pixel(p, x, ly, acc, total);
    }
    flush_tally<STATS>(tl, p.counters);
}

// ======================================================= kernel: QUEUE ====
// Persistent wavefronts.  Each lane owns one pixel at a time and walks its
// samples in order (so the f32 accumulation order is the reference's); a lane
// whose path ends starts its next sample immediately (path regeneration), and a
// lane whose pixel is finished takes the next pixel from a global queue: the
// wave ballots the idle lanes, lane 0 reserves popcount(idle) pixels with one
// atomic, and each idle lane picks its slot by prefix count.  All live lanes
// therefore execute the intersection code together on every iteration.
constexpr uint32_t kQueueBlock = 256;

template <bool STATS>
__global__ void __launch_bounds__(kQueueBlock) k_queue(const KParams p) {
    extern __shared__ uint32_t s_stack[];  // [stack_depth][blockDim.x]; empty for single-node trees
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t total_items = tiles_x * tiles_y * 64u;  // host guarantees < 2^32
    const uint32_t samples_total = p.n_passes * p.samples_per_pass;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    const Cam cam = make_cam(p);
    Tally<STATS> tl;

    bool have_pixel = false;   // lane owns a pixel
    bool exhausted = false;    // queue is empty for this wave
    uint32_t x = 0, ly = 0, y = 0, pixel_index = 0, sample = 0, total = 0;
    f3 acc = mk(0, 0, 0);
    Path pt;
    pt.depth = 0;
    bool in_path = false;

    for (;;) {
        // ---- refill idle lanes from the queue
        if (!exhausted) {
            const unsigned long long idle = __ballot(!have_pixel);
            if (idle != 0ull) {
                const uint32_t n = (uint32_t)__popcll(idle);
                uint32_t base = 0;
                if (lane == (uint32_t)(__ffsll((long long)idle) - 1)) base = atomicAdd(p.queue, n);
                base = __shfl(base, __ffsll((long long)idle) - 1, 64);
                if (!have_pixel) {
                    const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    const uint32_t item = base + rank;
                    if (item < total_items) {
                        const uint32_t tile = item >> 6, in = item & 63u;
                        x = (tile % tiles_x) * 8u + (in & 7u);
                        ly = (tile / tiles_x) * 8u + (in >> 3);
                        bool ok = (x < width) && (ly < p.local_rows);
                        if (ok) {
                            y = global_row(p, ly);
                            ok = y < p.u.height;
                        }
                        if (ok) {
                            pixel_index = y * width + x;
                            const float4 a4 = reinterpret_cast<const float4*>(p.accum)[(size_t)ly * width + x];
                            acc = mk(a4.x, a4.y, a4.z);
                            total = f2u(a4.w);
                            sample = 0;
                            have_pixel = true;
                            in_path = false;
                        }
                    }
                }
                if (base + n >= total_items) exhausted = true;
            }
        }
        if (__ballot(have_pixel) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (have_pixel) {
            if (!in_path) {
                start_path(p, cam, x, y, pixel_index, sample_base + sample, pt);
                in_path = p.u.max_depth > 0u;
            }
            if (in_path) in_path = segment<STATS>(p, pt, &s_stack[tid], kQueueBlock, tl);
            if (!in_path) {
                tl.paths++;
                acc = acc + pt.color;
                total = total + 1u;
                sample++;
                if (sample == samples_total) {
                    store_pixel(p, x, ly, acc, total);
                    have_pixel = false;
                }
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// ============================================== kernels: TRACE + ACCUMULATE ==
// Two-phase form of the same computation.  Phase 1 (k_trace): persistent
// wavefronts pull single (pixel, sample) items from a global queue -- a ballot of
// the idle lanes, one atomic by the first idle lane for popcount(idle) items, a
// prefix count to hand them out -- trace the path and store its radiance (16 B)
// into an HBM buffer laid out [tile][sample][64 pixels].  A lane that finishes a
// path regenerates at once, so all 64 lanes stay on the intersection code and
// the launch tail is one path, not one pixel's worth of samples.  Nothing in
// phase 1 depends on order.  Phase 2 (k_accumulate): one lane per pixel adds the
// samples to the accumulation IN SAMPLE ORDER (f32 addition is not associative:
// this is what keeps the frame bit-identical to the reference order,
// shader.wgsl:712-717), then tone-maps and packs (:720-722).  Phase 2 is a
// coalesced 1 KiB-per-wave stream and is HBM-bound; phase 1 is ALU-bound.
constexpr uint32_t kTraceBlock = 256;
#ifndef RB_TRACE_WAVES
#define RB_TRACE_WAVES 6
#endif

template <bool STATS>
__global__ void __launch_bounds__(kTraceBlock, RB_TRACE_WAVES) k_trace(const KParams p) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;  // host keeps this < 2^31
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    const Cam cam = make_cam(p);
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    Tally<STATS> tl;

    bool active = false, exhausted = false;
    uint32_t item = 0;
    uint32_t loc_next = 0, loc_end = 0;  // this wave's reserved item range (wave-uniform)
    const uint32_t batch = p.queue_batch;
    Path pt;
    pt.depth = 0;

    for (;;) {
        // ---- hand items to idle lanes.  The global queue is touched once per `batch`
        // items (one word sustains only ~90 M atomics/s chip-wide); inside a batch the
        // wave allocates with ballot + prefix count, no memory traffic.
        unsigned long long idle = __ballot(!active);
        for (int round = 0; round < 2 && idle != 0ull; round++) {
            if (loc_next == loc_end) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0u) b = atomicAdd(p.queue, batch);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= total_items) {
                    exhausted = true;
                    break;
                }
                loc_next = b;
                loc_end = (total_items - b < batch) ? total_items : b + batch;
            }
            const uint32_t avail = loc_end - loc_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const bool take = !active && rank < avail;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t taken = n_idle < avail ? n_idle : avail;
            if (take) {
                const uint32_t it = loc_next + rank;
                const uint32_t in = it & 63u, ts = it >> 6;
                const uint32_t tile = ts / S, smp = ts - tile * S;
                const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
                const uint32_t x = tx * 8u + (in & 7u), ly = ty * 8u + (in >> 3);
                bool ok = (x < width) && (ly < p.local_rows);
                uint32_t y = 0;
                if (ok) {
                    y = global_row(p, ly);
                    ok = y < p.u.height;
                }
                if (ok) {
                    start_path(p, cam, x, y, y * width + x, sample_base + smp, pt);
#if RB_ABLATE == 4
                    {
                        Path p2;
                        uint32_t x2 = x;
                        asm volatile("" : "+v"(x2));
                        start_path(p, cam, x2, y, y * width + x2, sample_base + smp, p2);
                        asm volatile("" ::"v"(p2.d.x), "v"(p2.d.y), "v"(p2.d.z), "v"(p2.seed));
                    }
#endif
                    item = it;
                    if (p.u.max_depth > 0u) {
                        active = true;
                    } else {
                        colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        tl.paths++;
                    }
                }
            }
            loc_next += taken;
            idle = __ballot(!active) & ~0ull;
            if (taken == n_idle) break;  // everyone who asked was served (or got a padding item)
        }
        if (__ballot(active) == 0ull) {
            if (exhausted && loc_next == loc_end) break;
            continue;
        }
        if (active) {
            const bool alive = segment<STATS>(p, pt, &s_stack[tid], kTraceBlock, tl);
            if (!alive) {
                colors[item] = make_float4(pt.color.x, pt.color.y, pt.color.z, 0.0f);
                tl.paths++;
                active = false;
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// k_trace for multi-node BVHs.  With 128-triangle leaves and no t-culling (the reference's
// traversal, kept for exactness) a ray visits a handful of leaves, but the number varies a lot
// between rays; run per segment, the wavefront waits for its slowest lane (measured ~1/3 lane
// utilisation on the 50k-triangle scene).  Here the scheduling unit is ONE LEAF: every loop
// iteration each traversing lane advances its own stack to its next leaf (node phase) and tests
// that leaf's triangles (leaf phase); a lane whose stack is empty finishes its segment (spheres,
// lights, shading) and starts the next segment or a new path at once, while the other lanes keep
// walking.  Visit order per ray is unchanged, so the winner is the same triangle.
template <bool STATS>
__global__ void __launch_bounds__(kTraceBlock) k_trace_bvh(const KParams p) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const uint32_t total_items = tiles_x * tiles_y * S * 64u;
    const uint32_t sample_base = p.first_pass * p.samples_per_pass;
    const uint32_t node_count = p.u.bvh_node_count;
    const Cam cam = make_cam(p);
    float4* __restrict__ colors = reinterpret_cast<float4*>(p.colors);
    const cf4p nodes = (cf4p)p.nodes;
    const cf4p ptris = (cf4p)p.ptris;
    uint32_t* const stack = &s_stack[tid];
    Tally<STATS> tl;

    enum : uint32_t { IDLE = 0, BEGIN = 1, TRAV = 2, FINISH = 3 };
    uint32_t state = IDLE;
    bool exhausted = false;
    uint32_t item = 0, loc_next = 0, loc_end = 0;
    const uint32_t batch = p.queue_batch;
    Path pt;
    pt.depth = 0;
    TriHit th;
    th.hit = false;
    th.t = 1e20f;
    th.u = th.v = 0.0f;
    th.slot = 0u;
    f3 inv = mk(0, 0, 0);
    int sp = 0;

    for (;;) {
        // ---- (1) hand items to idle lanes (same scheme as k_trace)
        unsigned long long idle = __ballot(state == IDLE);
        for (int round = 0; round < 2 && idle != 0ull; round++) {
            if (loc_next == loc_end) {
                if (exhausted) break;
                uint32_t b = 0;
                if (lane == 0u) b = atomicAdd(p.queue, batch);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= total_items) {
                    exhausted = true;
                    break;
                }
                loc_next = b;
                loc_end = (total_items - b < batch) ? total_items : b + batch;
            }
            const uint32_t avail = loc_end - loc_next;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const bool take = (state == IDLE) && rank < avail;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t taken = n_idle < avail ? n_idle : avail;
            if (take) {
                const uint32_t it = loc_next + rank;
                const uint32_t in = it & 63u, ts = it >> 6;
                const uint32_t tile = ts / S, smp = ts - tile * S;
                const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
                const uint32_t x = tx * 8u + (in & 7u), ly = ty * 8u + (in >> 3);
                bool ok = (x < width) && (ly < p.local_rows);
                uint32_t y = 0;
                if (ok) {
                    y = global_row(p, ly);
                    ok = y < p.u.height;
                }
                if (ok) {
                    start_path(p, cam, x, y, y * width + x, sample_base + smp, pt);
                    item = it;
                    if (p.u.max_depth > 0u) {
                        state = BEGIN;
                    } else {
                        colors[it] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        tl.paths++;
                    }
                }
            }
            loc_next += taken;
            idle = __ballot(state == IDLE);
            if (taken == n_idle) break;
        }
        if (__ballot(state != IDLE) == 0ull) {
            if (exhausted && loc_next == loc_end) break;
            continue;
        }

        // ---- (2) start of a segment: reset the traversal (shader.wgsl:283-307)
        if (state == BEGIN) {
            th.hit = false;
            th.t = 1e20f;
            th.u = th.v = 0.0f;
            th.slot = 0u;
            inv = mk(rcp_exact(pt.d.x), rcp_exact(pt.d.y), rcp_exact(pt.d.z));
            stack[0] = 0u;
            sp = 1;
            state = TRAV;
        }

        // ---- (3) node phase: pop until this lane has a leaf to test or its stack is empty
        uint32_t first = 0, count = 0;
        while (state == TRAV && count == 0u) {
            if (sp == 0) {
                state = FINISH;
                break;
            }
            sp--;
            const uint32_t node_idx = stack[sp * kTraceBlock];
            if (node_idx >= node_count) continue;
            const v4f n0 = nodes[node_idx * 3u], n1 = nodes[node_idx * 3u + 1u];
            const v4u n2 = ((cu4p)p.nodes)[node_idx * 3u + 2u];
            if constexpr (STATS) tl.nodes++;
            if (!isect_aabb(pt.o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) continue;
            if (n2.w > 0u) {
                first = n2.z;
                count = n2.w;
            } else {
                if (n2.x < node_count) {
                    stack[sp * kTraceBlock] = n2.x;
                    sp++;
                }
                if (n2.y < node_count) {
                    stack[sp * kTraceBlock] = n2.y;
                    sp++;
                }
            }
        }

        // ---- (4) leaf phase: this lane's leaf (shader.wgsl:327-374), two triangles per step so
        // that six 16-byte loads are in flight per lane; candidates are offered in slot order
        {
            const uint32_t end = (first + count < p.index_len) ? first + count : p.index_len;  // guard :331
            uint32_t slot = first;
#if RB_TRI_PAIRS
            for (; slot + 2u <= end; slot += 2u) {
                const v4f a0 = ptris[slot * 4u], b0 = ptris[slot * 4u + 1u], c0 = ptris[slot * 4u + 2u];
                const v4f a1 = ptris[slot * 4u + 4u], b1 = ptris[slot * 4u + 5u], c1 = ptris[slot * 4u + 6u];
                const bool ok0 = __float_as_uint(c0.w) != 0u, ok1 = __float_as_uint(c1.w) != 0u;
                if constexpr (STATS) tl.tris += (ok0 ? 1u : 0u) + (ok1 ? 1u : 0u);
                test_pair(a0, b0, c0, ok0, a1, b1, c1, ok1, slot, pt.o, pt.d, th, tl);
            }
#endif
            for (; slot < end; slot++) {
                const v4f a = ptris[slot * 4u], b = ptris[slot * 4u + 1u], c = ptris[slot * 4u + 2u];
                if (__float_as_uint(c.w) == 0u) continue;  // guard :336
                if constexpr (STATS) tl.tris++;
                const float before = th.t;
                test_slot(a, b, c, slot, pt.o, pt.d, th);
                if constexpr (STATS) tl.mesh_hits += (th.t != before) ? 1u : 0u;
            }
        }

        // ---- (5) traversal complete: ground, spheres, lights, shading, next ray
        if (state == FINISH) {
            const bool alive = segment_finish<STATS>(p, pt, th, stack, kTraceBlock, tl);
            if (alive) {
                state = BEGIN;
            } else {
                colors[item] = make_float4(pt.color.x, pt.color.y, pt.color.z, 0.0f);
                tl.paths++;
                state = IDLE;
            }
        }
    }
    flush_tally<STATS>(tl, p.counters);
}

// Phase 2: ordered accumulation + tone map + pack.  One wavefront per 8x8 tile,
// lane = pixel; each sample row is a contiguous 1 KiB read.
__global__ void __launch_bounds__(256) k_accumulate(const KParams p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t width = p.u.width;
    const uint32_t tiles_x = (width + 7u) / 8u;
    const uint32_t tiles_y = (p.local_rows + 7u) / 8u;
    if (tile >= tiles_x * tiles_y) return;
    const uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u), ly = ty * 8u + (lane >> 3);
    if (x >= width || ly >= p.local_rows) return;
    if (global_row(p, ly) >= p.u.height) return;
    const uint32_t S = p.n_passes * p.samples_per_pass;
    const float4 a4 = reinterpret_cast<const float4*>(p.accum)[(size_t)ly * width + x];
    f3 acc = mk(a4.x, a4.y, a4.z);
    uint32_t total = f2u(a4.w);
    const nt_f4* __restrict__ c = reinterpret_cast<const nt_f4*>(p.colors) + ((size_t)tile * S) * 64u + lane;
    uint32_t s = 0;
    for (; s + 8u <= S; s += 8u) {
        nt_f4 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(&c[(size_t)(s + k) * 64u]);
#pragma unroll
        for (int k = 0; k < 8; k++) acc = acc + mk(v[k].x, v[k].y, v[k].z);
    }
    for (; s < S; s++) {
        const nt_f4 v = __builtin_nontemporal_load(&c[(size_t)s * 64u]);
        acc = acc + mk(v.x, v.y, v.z);
    }
    total += S;
    store_pixel(p, x, ly, acc, total);
}

// ============================================================ prep kernel ==
// Gathers triangles into bvh_indices order and hoists the per-triangle
// invariants (edge1, edge2, geometric normal) of shader.wgsl:249-250,351.
__global__ void k_prep_tris(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices,
                            uint32_t index_len, PrepTri* out, PrepTriShade* shade) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= index_len) return;
    PrepTri t;
    PrepTriShade s;
    const uint32_t id = indices[slot];
    if (id >= tri_count) {  // shader.wgsl:336 `continue`
        t = PrepTri{};
        s = PrepTriShade{};
        t.valid = 0u;
    } else {
        const rb_gpu_triangle g = tris[id];
        const f3 v0 = ld3(g.v0), v1 = ld3(g.v1), v2 = ld3(g.v2);
        const f3 e1 = v1 - v0, e2 = v2 - v0;
        const f3 n = normalize(cross(e1, e2));
        t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
        t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
        t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
        t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
        t.tri_id = id;
        t.mesh_index = g.mesh_index;
        t.valid = 1u;
        t._pad = 0u;
        s.v0_index = g.v0_index;
        s.v1_index = g.v1_index;
        s.v2_index = g.v2_index;
        s._pad = 0u;
    }
    out[slot] = t;
    shade[slot] = s;
}

// Per-material invariants of the shading branch (shader.wgsl:615-623,637), written into the pad
// words of the DEVICE copy of each Material (the caller's buffers are never touched).
__global__ void k_prep_materials(unsigned char* first_material, uint32_t stride, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rb_material* m = reinterpret_cast<rb_material*>(first_material + (size_t)i * stride);
    const float specular_strength = (m->specular[0] + m->specular[1] + m->specular[2]) / 3.0f;
    const float diffuse_strength = (m->diffuse[0] + m->diffuse[1] + m->diffuse[2]) / 3.0f;
    const bool is_metal = specular_strength > 0.01f && diffuse_strength < 0.01f;
    const float fuzz = fminf(fmaxf(1.0f - (m->shininess / 1000.0f), 0.0f), 1.0f);
    m->_pad1 = fuzz;
    m->_pad2 = is_metal ? 1u : 0u;
}

// Copies prepared triangles into the fast tree's leaf order.
__global__ void k_gather_tris(const PrepTri* ptris, const uint32_t* slots, uint32_t n, PrepTri* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = ptris[slots[j]];
}

// Exhaustive check of rcp_newton against the compiler's correctly rounded 1/b: every one of the
// 2^23 significands at biased exponent `expo`, both signs.  mismatch[0] counts differing results
// among rcp_safe inputs; mismatch[1..] records up to 15 offending bit patterns.
__global__ void k_rcp_exhaustive(uint32_t expo, uint32_t* mismatch) {
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= (1u << 23)) return;
    for (uint32_t sign = 0; sign < 2; sign++) {
        const uint32_t bits = (sign << 31) | ((expo & 0xFFu) << 23) | m;
        const float b = __uint_as_float(bits);
        float want, got;
        if (expo & 0x100u) {  // mode 1: sqrt (positive operands only)
            if (sign) continue;
            want = sqrtf(b);
            got = sqrt_exact(b);
        } else {
            if (!rcp_safe(b)) continue;
            want = 1.0f / b;
            got = rcp_newton(b);
        }
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const uint32_t k = atomicAdd(&mismatch[0], 1u);
            if (k < 15u) mismatch[1u + k] = bits;
        }
    }
}

// Exhaustive check of div_newton: thread = one denominator significand (biased exponent eb),
// loop over `a_count` numerator significands starting at a_begin (biased exponent ea).
__global__ void k_div_exhaustive(uint32_t b_begin, uint32_t ea, uint32_t eb, uint32_t a_begin, uint32_t a_count,
                                 unsigned long long* mismatch) {
    const uint32_t mb = b_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (mb >= (1u << 23)) return;
    const float b = __uint_as_float((eb << 23) | mb);
    if (!div_safe_den(b)) return;
    const float y = rcp_newton(b);
    unsigned long long bad = 0;
    uint32_t first_bad = 0;
    for (uint32_t i = 0; i < a_count; i++) {
        const float a = __uint_as_float((ea << 23) | ((a_begin + i) & 0x7FFFFFu));
        const float want = a / b;
        const float got = div_newton(a, b, y);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (bad == 0) first_bad = __float_as_uint(a);
            bad++;
        }
    }
    if (bad) {
        const unsigned long long k = atomicAdd(&mismatch[0], bad);
        if (k < 7ull) {
            mismatch[1 + 2 * k] = first_bad;
            mismatch[2 + 2 * k] = __float_as_uint(b);
        }
    }
}

// Exposes the device's /, sqrt, normalize and u32->f32 to the parity tests.
__global__ void k_debug_math(const float* a, const float* b, float* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b[i];
    out[i] = x / y;
    out[n + i] = sqrtf(fabsf(x));
    const f3 v = normalize(mk(x, y, x * y));
    out[2 * n + i] = v.x;
    out[3 * n + i] = v.y;
    out[4 * n + i] = v.z;
    out[5 * n + i] = (float)__float_as_uint(x) / 4294967296.0f;
    out[6 * n + i] = fminf(fmaxf(x, y), x * 0.0f);
    out[7 * n + i] = dot(mk(x, y, x), mk(y, y, x));
}

}  // namespace

// ================================================================ launchers ==
int device_cu_count(int device) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
}

static uint32_t persistent_blocks(uint64_t items, uint32_t block, uint32_t blocks_per_cu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t blocks = (uint32_t)device_cu_count(dev) * blocks_per_cu;
    const uint64_t needed = (items + block - 1) / block;
    if (needed < blocks) blocks = (uint32_t)needed;
    return blocks;
}

int launch_render(const KParams& p, uint32_t kernel, bool stats, void* stream_, LaunchInfo* info, void* ev_after_trace) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    LaunchInfo li{};
    const size_t lds = sizeof(uint32_t) * p.stack_depth * 256u;
    li.lds_bytes = lds;
    if (kernel == RB_KERNEL_PIXEL) {
        const uint32_t tiles_x = (p.u.width + 15u) / 16u, tiles_y = (p.local_rows + 15u) / 16u;
        li.grid = tiles_x * tiles_y;
        li.block = kPixelBlock;
        li.kernel_name = "k_pixel";
        if (li.grid == 0) return 0;
        if (stats)
            hipLaunchKernelGGL(k_pixel<true>, dim3(li.grid), dim3(li.block), lds, stream, p);
        else
            hipLaunchKernelGGL(k_pixel<false>, dim3(li.grid), dim3(li.block), lds, stream, p);
    } else if (kernel == RB_KERNEL_QUEUE) {
        const uint64_t items = (uint64_t)((p.u.width + 7u) / 8u) * ((p.local_rows + 7u) / 8u) * 64u;
        li.grid = persistent_blocks(items, kQueueBlock, p.blocks_per_cu ? p.blocks_per_cu : 8u);  // residency is set by VGPRs
        li.block = kQueueBlock;
        li.kernel_name = "k_queue";
        if (li.grid == 0) return 0;
        hipError_t e = hipMemsetAsync(p.queue, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return (int)e;
        if (stats)
            hipLaunchKernelGGL(k_queue<true>, dim3(li.grid), dim3(li.block), lds, stream, p);
        else
            hipLaunchKernelGGL(k_queue<false>, dim3(li.grid), dim3(li.block), lds, stream, p);
    } else {
        const uint64_t tiles = (uint64_t)((p.u.width + 7u) / 8u) * ((p.local_rows + 7u) / 8u);
        const uint64_t items = tiles * 64u * p.n_passes * p.samples_per_pass;
        li.grid = persistent_blocks(items, kTraceBlock, p.blocks_per_cu ? p.blocks_per_cu : 8u);
        li.block = kTraceBlock;
        if (li.grid == 0) return 0;
        // batch: >= 64 reservations per wave for balance, <= 4096 items, multiple of 64
        KParams q = p;
        const uint64_t waves = (uint64_t)li.grid * (kTraceBlock / 64u);
        uint64_t batch = items / (waves * 64u);
        batch = (batch / 64u) * 64u;
        if (batch < 64u) batch = 64u;
        if (batch > 4096u) batch = 4096u;
        if (p.queue_batch) batch = p.queue_batch;
        q.queue_batch = (uint32_t)batch;
        hipError_t e = hipMemsetAsync(p.queue, 0, sizeof(uint32_t), stream);
        if (e != hipSuccess) return (int)e;
        const bool stepped = p.u.bvh_node_count > 1u && !p.no_leaf_stepping && p.fast_nodes == nullptr;
        li.kernel_name = stepped ? "k_trace_bvh" : "k_trace";
        if (stepped) {
            if (stats)
                hipLaunchKernelGGL(k_trace_bvh<true>, dim3(li.grid), dim3(li.block), lds, stream, q);
            else
                hipLaunchKernelGGL(k_trace_bvh<false>, dim3(li.grid), dim3(li.block), lds, stream, q);
        } else if (stats) {
            hipLaunchKernelGGL(k_trace<true>, dim3(li.grid), dim3(li.block), lds, stream, q);
        } else {
            hipLaunchKernelGGL(k_trace<false>, dim3(li.grid), dim3(li.block), lds, stream, q);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        if (ev_after_trace) (void)hipEventRecord(static_cast<hipEvent_t>(ev_after_trace), stream);
        hipLaunchKernelGGL(k_accumulate, dim3((uint32_t)((tiles + 3u) / 4u)), dim3(256), 0, stream, p);
    }
    if (info) *info = li;
    return (int)hipGetLastError();
}

int launch_prep_tris(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices,
                     uint32_t index_len, PrepTri* out, PrepTriShade* shade, void* stream_) {
    if (index_len == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const uint32_t block = 256, grid = (index_len + block - 1) / block;
    hipLaunchKernelGGL(k_prep_tris, dim3(grid), dim3(block), 0, stream, tris, tri_count, indices, index_len, out,
                       shade);
    return (int)hipGetLastError();
}

int launch_prep_materials(void* first_material, uint32_t stride, uint32_t n, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_prep_materials, dim3((n + 255) / 256), dim3(256), 0, stream,
                       static_cast<unsigned char*>(first_material), stride, n);
    return (int)hipGetLastError();
}

int launch_gather_tris(const PrepTri* ptris, const uint32_t* slots, uint32_t n, PrepTri* out, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_gather_tris, dim3((n + 255) / 256), dim3(256), 0, stream, ptris, slots, n, out);
    return (int)hipGetLastError();
}

int launch_div_exhaustive(uint32_t b_begin, uint32_t b_count, uint32_t ea, uint32_t eb, uint32_t a_begin,
                          uint32_t a_count, unsigned long long* mismatch, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_div_exhaustive, dim3((b_count + 255) / 256), dim3(256), 0, stream, b_begin, ea, eb, a_begin,
                       a_count, mismatch);
    return (int)hipGetLastError();
}

int launch_rcp_exhaustive(uint32_t expo, uint32_t* mismatch, void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_rcp_exhaustive, dim3((1u << 23) / 256), dim3(256), 0, stream, expo, mismatch);
    return (int)hipGetLastError();
}

int launch_debug_math(const float* a, const float* b, float* out, uint32_t n, void* stream_) {
    if (n == 0) return 0;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, stream, a, b, out, n);
    return (int)hipGetLastError();
}

}  // namespace rb
