// rb_device_intersect.hpp -- textures, primitive intersection tests, work counters, BVH walks (reference order, fast tree, sphere tree).
// Part of the single device translation unit rb_kernels.hip (numerics contract: see there).
#pragma once
#include "rb_device_math.hpp"

#pragma clang fp contract(off)

namespace rb {
namespace {

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

// shader.wgsl:248-280 with edge1/edge2 supplied (v1 - v0, v2 - v0).  (A branch-free form that evaluates
// everything and folds the four early returns into one predicate was measured slower on gfx950, 34.8 vs
// 33.5 ms on C2-short: the early-outs do skip whole-wave work.  tools/ablate/rb_forks.patch.)
DEV float isect_triangle(f3 o, f3 d, f3 v0, f3 edge1, f3 edge2, float& uo, float& vo) {
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

// (One triangle per step.  r01-r02 tested two per step with packed f32 math: on gfx950 a packed f32 instruction issues at
// half the rate of a plain one (MI355X_MICROARCH.md), so the pairing bought no throughput and paid for the register
// pairs -- 4.9 % slower on C2, profiles/r03_noslp.txt.  That arm lives in tools/ablate/rb_forks.patch.)

// Walk of the library's own tree over the same triangles (rb_bvh.cpp / rb_build.hip; RB_FLAG_FAST_BVH -- r02's default
// for large meshes, since r03 the chunked walk of rb_kernels.hip is the default): nearer child first, subtrees
// skipped when missed or entered beyond the best t.  It reproduces the reference walk's winner
// (shader.wgsl:282-392) bit for bit -- DESIGN.md section 4 has the full argument:
//  * candidates are evaluated with the reference's isect_triangle, so t, u, v are the same bits;
//  * equal t resolves by the triangle's rank in the reference's visit order;
//  * the reference only tests a triangle if every node from the root to its leaf passes intersect_aabb: an
//    improving candidate is accepted only after that chain has been re-checked with the reference's own slab
//    arithmetic on the reference's boxes;
//  * no triangle the reference would report as hit is ever lost to culling.  The reference's f32
//    Moller-Trumbore can report a hit some distance from its triangle; that distance is bounded as long as the
//    ray is not nearly parallel to the triangle's plane.  Hits are therefore split by c0 = kFastGrazeCos:
//      (A) |cos(ray, normal)| >= c0: a subtree of the library's tree is skipped only if the ray misses its box
//          inflated by a margin that covers every such hit below it (entry());
//      (B) |cos| < c0 (the ray is within ~1.7 degrees of the triangle's plane; no useful bound exists): found by a
//          second pass over the REFERENCE tree with the reference's own slab tests, entering only nodes whose
//          normal cone admits such a triangle, and testing only the triangles of a leaf that are (gnode_step / gleaf_step).
//    Both passes offer their candidates to the same closest-hit rule, so the reference's winner is always among them.
// The walk is a resumable object so that the same steps serve the per-segment kernels (intersect_bvh_fast
// below: run to completion) and the stepped kernel (k_trace_fast: lanes that finish are shaded and refilled
// while the others keep walking).
//
// (A) in numbers.  For the ray (o, d), |d| = 1 +- 4 ulp, and a triangle k with L = max(|e1|, |e2|), N = |e1 x e2|,
// s = o - v0, a = d . (e2 x e1) = N cos(d, n_k) and the reference's computed determinant a^ (a hit is accepted only
// if |a^| >= 1e-6), the point X = o + t^ d of an ACCEPTED hit satisfies (u = 2^-24, first order, every rounding of
// shader.wgsl:248-280 in the no-FMA f32 arithmetic of the contract)
//     dist_inf(X, box(triangle k)) <= 26 u (|s| + L) L^2 / |a_k|  +  7.5 u (|s| + 2 L)
// (r03's sharper sum, second order included: (21.4 |s| + 9.1 L) u L^2 / |a_k|, of which 11.2 |s| + 4.6 L across the ray and
// 10.2 |s| + 4.6 L along it -- k_trace_chunk uses the two parts separately, this walk keeps the round figure)
// as long as 5.42 u L^2 / |a^| <= 0.05.  With |cos| >= c0: L^2 / |a_k| <= (max_k L_k^2 / N_k) / c0, a per-child
// constant FA (stored in the node's pad words with the 0.95 that bounds |a^| >= 0.95 |a|; +inf = always enter,
// beyond 1.5e5 where no bound is claimed or for a triangle with N = 0); |s| + 2 L <= Sp = (distance from o to the
// child box's farthest corner) + 2 (sum of its extents); plus 10 u Sp for the rounding of the slab test itself:
//     margin = Sp (27 u FA + 20 u),
// and X is inside the box so inflated, entered no later than t^: the slab test with that margin and the cull
// `entry > best t` are safe.
// Can a triangle whose normal lies in `cone` = {c cos(alpha), tan(alpha)} (every unit normal below is within alpha of
// +-c) have |cos(d, n)| < c0?  |cos(d, n)| >= cos(alpha) (x - tan(alpha) sqrt(1 - x^2)) with x = |d . c|.  y = x cos(alpha)
// is known to 6e-7, k2 - y^2 to 4e-6 k2; all zeros = no cone = always possible; tan(alpha) = -1 = nothing below = never.
// cone_cos_bound: that lower bound of |cos(d, n)| over the cone (<= 0, or NaN, when it says nothing).
DEV float cone_cos_bound(f3 d, v4f cone) {
    const float y = fabsf(__builtin_fmaf(d.z, cone.z, __builtin_fmaf(d.y, cone.y, d.x * cone.x)));
    const float k2 = __builtin_fmaf(cone.z, cone.z, __builtin_fmaf(cone.y, cone.y, cone.x * cone.x));
    const float root = 1.000001f * __builtin_amdgcn_sqrtf(fmaxf(__builtin_fmaf(-y, y, k2), 0.0f) + 4e-6f * k2);
    return __builtin_fmaf(-cone.w, root, y - 1e-6f);
}
DEV bool cone_admits_grazing(f3 d, v4f cone) {
    if (cone.w < 0.0f) return false;
    return !(cone_cos_bound(d, cone) >= kFastGrazeCos);
}
constexpr float kFastKF = 27.0f * 5.9604645e-8f * 1.01f;
constexpr float kFastKS = 20.0f * 5.9604645e-8f * 1.01f;
template <bool STATS>
struct FastWalk {
    f3 o, d, inv;
    float m_ref;        // shrink of a reference leaf box for the chain shortcut (reference_would_test)
    TriHit h;
    uint32_t best_rank;
    uint32_t cur;       // current child reference (leaf: bit 31); second pass: reference node, or bit 31 | next slot of a leaf
    uint32_t gend;      // second pass: end of the leaf being scanned
    bool graze;         // false: the culled walk of the library's tree (A); true: the second pass over the reference tree (B)
    int sp;

    DEV void begin(const KParams& p, f3 o_, f3 d_) {
        o = o_;
        d = d_;
        h.hit = false;
        h.t = 1e20f;
        h.u = 0.0f;
        h.v = 0.0f;
        h.slot = 0u;
        best_rank = 0xFFFFFFFFu;
        inv = mk(rcp_exact(d.x), rcp_exact(d.y), rcp_exact(d.z));
        // farthest the ray origin can be from any point of the mesh: the rounding error of a slab test on any of
        // the reference's boxes is below 8 ulp of that
        const f3 fb0 = ld3(p.fast_bmin), fb1 = ld3(p.fast_bmax);
        const float sx_ = fmaxf(fabsf(o.x - fb0.x), fabsf(o.x - fb1.x)), sy_ = fmaxf(fabsf(o.y - fb0.y), fabsf(o.y - fb1.y)),
                    sz_ = fmaxf(fabsf(o.z - fb0.z), fabsf(o.z - fb1.z));
        m_ref = 1e-6f * sqrtf(sx_ * sx_ + sy_ * sy_ + sz_ * sz_) + 1e-30f;
        cur = p.fast_root;
        gend = 0u;
        graze = false;
        sp = 0;
    }
    DEV bool at_leaf() const { return (cur & 0x80000000u) != 0u; }
    // which step this lane needs next: 0 node / 1 leaf of the library's tree, 2 node / 3 leaf chunk of the second pass
    DEV uint32_t kind() const { return (graze ? 2u : 0u) | (cur >> 31); }

    // Entries beyond the LDS stack (trees deeper than kStackDepth: only device-built ones, the host
    // builder limits its depth) spill to this lane's column of a global scratch array.
    // (volatile on the spill side keeps the compiler from merging the two accesses into one through a
    // generic pointer, which would turn every LDS stack access into a flat_load / flat_store)
    DEV static volatile uint32_t* spill(const KParams& p, int at) {
        return p.stack_overflow + ((size_t)(uint32_t)(at - (int)kStackDepth) * (gridDim.x * blockDim.x) + blockIdx.x * blockDim.x + threadIdx.x);
    }
    DEV static void push(const KParams& p, uint32_t* stack, uint32_t stride, int at, uint32_t v) {
        if (__builtin_expect(at < (int)kStackDepth, 1)) stack[at * stride] = v;
        else *spill(p, at) = v;
    }
    DEV static uint32_t peek(const KParams& p, const uint32_t* stack, uint32_t stride, int at) {
        uint32_t v;
        if (__builtin_expect(at < (int)kStackDepth, 1)) v = stack[at * stride];
        else v = *spill(p, at);
        return v;
    }
    // next pending subtree; when the library's tree is done the second pass starts at the reference root;
    // false when both are complete
    DEV bool pop(const KParams& p, const uint32_t* stack, uint32_t stride) {
        if (sp == 0) {
            if (graze || p.fast_skip_second_pass != 0u) return false;
            graze = true;
            cur = 0u;
            return true;
        }
        sp--;
        cur = peek(p, stack, stride, sp);
        return true;
    }

    // visit the child unless the ray misses its inflated box or enters it beyond the best t (comparisons are
    // written so that a NaN means "visit").  This arithmetic only steers the walk: fused operations and the
    // hardware's approximate sqrt / rcp are fine as long as every rounding is on the safe side.
    DEV bool entry(v4f lo, v4f hi, float fa, float& tn) const {
        const f3 a = mk(lo.x, lo.y, lo.z) - o, b = mk(hi.x, hi.y, hi.z) - o;
        const float fx = fmaxf(fabsf(a.x), fabsf(b.x)), fy = fmaxf(fabsf(a.y), fabsf(b.y)), fz = fmaxf(fabsf(a.z), fabsf(b.z));
        // Sp >= |o - v0| + 2 L for every triangle below: farthest corner (v_sqrt_f32 is within 1 ulp) + box extents
        const float sp_ = 1.001f * __builtin_amdgcn_sqrtf(__builtin_fmaf(fx, fx, __builtin_fmaf(fy, fy, fz * fz))) +
                          2.0f * (((b.x - a.x) + (b.y - a.y)) + (b.z - a.z));
        const float mm = (fa <= 1.5e5f) ? sp_ * __builtin_fmaf(kFastKF, fa, kFastKS) : 1e30f;   // NaN -> 1e30
        const f3 t0 = mk(a.x - mm, a.y - mm, a.z - mm) * inv;
        const f3 t1 = mk(b.x + mm, b.y + mm, b.z + mm) * inv;
        tn = fmaxf(fmaxf(fminf(t0.x, t1.x), fminf(t0.y, t1.y)), fminf(t0.z, t1.z));
        const float tf = fminf(fminf(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y)), fmaxf(t0.z, t1.z));
        return !(tf < fmaxf(tn, 0.0f)) && !(tn > h.t);
    }

    // (B): can a triangle with |cos(d, n)| < c0 lie below a reference node?  cone = {c cos(alpha), tan(alpha)},
    // every unit normal below is within alpha of +-c, so |cos(d, n)| >= cos(alpha) (x - tan(alpha) sqrt(1 - x^2)) with
    // x = |d . c|.  y = x cos(alpha) is known to 6e-7, k2 - y^2 to 4e-6 k2; all zeros = no cone = always possible;
    // tan(alpha) = -1 = no large triangle below = never.
    DEV bool graze_possible(v4f cone) const { return cone_admits_grazing(d, cone); }
    // The second pass: the reference's own walk (shader.wgsl:309-389: same nodes, same slab arithmetic, so a leaf
    // is reached exactly when the reference tests its triangles), restricted to nodes that can hold a class-(B)
    // triangle, and in a leaf to the triangles that are class (B) for this ray (the prepared unit normal is the
    // reference's f32 normalize(cross(e1, e2))).  Candidates go through the reference's intersect_triangle and the
    // closest-hit rule of leaf_step; the visit order does not matter because equal t resolves by reference rank.
    // Resumable like the first pass: one reference node per gnode_step, up to kGrazeChunk triangles per gleaf_step,
    // on the same (by now empty) stack.
    static constexpr uint32_t kGrazeChunk = 16u;
    DEV bool gnode_step(const KParams& p, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
        const cf4p gn = (cf4p)p.gnodes + (size_t)cur * 4u;
        const v4f cone = gn[2];
        if (graze_possible(cone)) {
            const v4f n0 = gn[0], n1 = gn[1];
            if constexpr (STATS) tl.nodes++;
            if (isect_aabb(o, inv, mk(n0.x, n0.y, n0.z), mk(n1.x, n1.y, n1.z))) {
                const v4f n3 = gn[3];   // first, count (the leaf's large triangles in gslots), cap, is_leaf
                // A near-degenerate hit of a triangle below is reported no farther than margin(cap) from this box
                // (entry(): the same bound with L^2 / |a^| <= L^2 / 1e-6), so it cannot beat the best t if the ray
                // enters the inflated box beyond it.
                float tn;
                if (entry(n0, n1, n3.z, tn)) {
                    const uint32_t node_count = p.u.bvh_node_count;
                    if (__float_as_uint(n3.w) != 0u) {   // leaf
                        const uint32_t first = __float_as_uint(n3.x), count = __float_as_uint(n3.y);
                        if (count > 0u) {
                            cur = 0x80000000u | first;
                            gend = first + count;
                            return true;
                        }
                    } else {
                        const uint32_t l = __float_as_uint(n0.w), r = __float_as_uint(n1.w);
                        const bool hl = l < node_count, hr = r < node_count;   // guards :376-387
                        if (hl && hr) {
                            stack[sp * stride] = r;
                            sp++;
                            cur = l;
                            return true;
                        }
                        if (hl || hr) {
                            cur = hl ? l : r;
                            return true;
                        }
                    }
                }
            }
        }
        return pop(p, stack, stride);
    }
    DEV bool gleaf_step(const KParams& p, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
        const cf4p ptris = (cf4p)p.ptris;
        const RB_CONST uint32_t* meta = cptr(p.slot_meta);
        const RB_CONST uint32_t* gslots = cptr(p.gslots);
        uint32_t j = cur & 0x7FFFFFFFu;
        const uint32_t stop = (j + kGrazeChunk < gend) ? j + kGrazeChunk : gend;
        for (; j < stop; j++) {
            const uint32_t slot = gslots[j];
            const v4f nr = ptris[slot * 4u + 3u];
            // prepared normal vs true normal: the f32 cross product is off by <= 2.42 u L^2, i.e. the direction by
            // <= 2.5 u q + 3 u <= 6.5e-4 rad for every triangle (A) has a finite margin for (q <= 4275); a
            // triangle beyond that (or without a normal: NaN here) has FA = +inf and is never culled by (A)
            if (!(fabsf(__builtin_fmaf(d.z, nr.z, __builtin_fmaf(d.y, nr.y, d.x * nr.x))) < kFastGrazeCos * 1.03f)) continue;
            const v4f a = ptris[slot * 4u], b = ptris[slot * 4u + 1u], c = ptris[slot * 4u + 2u];
            if (__float_as_uint(c.w) == 0u) continue;  // guard :336
            if constexpr (STATS) tl.tris++;
            float u, v;
            const float t = isect_triangle(o, d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), u, v);
            if (t > 0.001f && !(t > h.t)) {
                const uint32_t rank = meta[slot * 2u + 1u] & ~kSlotLarge;
                if (t < h.t || rank < best_rank) {
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
        if (j < gend) {
            cur = 0x80000000u | j;
            return true;
        }
        return pop(p, stack, stride);
    }
    // one step of whatever kind this lane needs; false when the walk is complete
    DEV bool step(const KParams& p, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
        switch (kind()) {
            case 0u: return node_step(p, stack, stride, tl);
            case 1u: return leaf_step(p, stack, stride, tl);
            case 2u: return gnode_step(p, stack, stride, tl);
            default: return gleaf_step(p, stack, stride, tl);
        }
    }

    // cur is an internal node: descend into the nearer child that is hit, remember the other.
    // Returns false when the walk is complete.
    DEV bool node_step(const KParams& p, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
        const cf4p nodes = (cf4p)p.fast_nodes;
        const v4f l0 = nodes[cur * 4u], l1 = nodes[cur * 4u + 1u], r0 = nodes[cur * 4u + 2u], r1 = nodes[cur * 4u + 3u];
        if constexpr (STATS) tl.nodes++;
        const uint32_t lref = __float_as_uint(l0.w), rref = __float_as_uint(l1.w);
        float tl_, tr_;
        const bool hl = entry(l0, l1, r0.w, tl_), hr = entry(r0, r1, r1.w, tr_);
        if (hl && hr) {
            const bool left_first = !(tr_ < tl_);
            push(p, stack, stride, sp, left_first ? rref : lref);
            sp++;
            cur = left_first ? lref : rref;
            return true;
        }
        if (hl) {
            cur = lref;
            return true;
        }
        if (hr) {
            cur = rref;
            return true;
        }
        return pop(p, stack, stride);
    }

    DEV bool reference_would_test(const KParams& p, uint32_t leaf_node, Tally<STATS>& tl) const {
        const cf4p rnodes = (cf4p)p.nodes;
        const RB_CONST uint32_t* parent = cptr(p.ref_parent);
        const float m = m_ref;
        // Shortcut: if the ray passes through the reference LEAF's box shrunk by m on every side (m = 1e-6 of
        // the farthest the origin can be from the mesh: twice the rounding error of this slab test plus that of
        // the reference's), it passes through the interior of every ancestor's box with room to spare, so each of
        // the reference's slab tests succeeds; only a ray that merely grazes the leaf box needs the exact walk
        // up the chain.
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
    }

    // cur is a leaf (1 or 2 triangles): test them, then take the next pending subtree.
    // Returns false when the walk is complete.
    DEV bool leaf_step(const KParams& p, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
        const cf4p ftris = (cf4p)p.fast_tris;
        const RB_CONST uint32_t* fslots = cptr(p.fast_slots);
        const RB_CONST uint32_t* meta = cptr(p.slot_meta);
        const uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 3u) + 1u;
        for (uint32_t j = first; j < first + count; j++) {
            const v4f a = ftris[j * 4u], b = ftris[j * 4u + 1u], c = ftris[j * 4u + 2u];
            if constexpr (STATS) tl.tris++;
            float u, v;
            const float t = isect_triangle(o, d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), u, v);
            if (t > 0.001f && !(t > h.t)) {
                const uint32_t slot = fslots[j];
                const uint32_t leaf_node = meta[slot * 2u], rank = meta[slot * 2u + 1u] & ~kSlotLarge;
                if ((t < h.t || rank < best_rank) && reference_would_test(p, leaf_node, tl)) {
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
        return pop(p, stack, stride);
    }
};

template <bool STATS>
DEV TriHit intersect_bvh_fast(const KParams& p, f3 o, f3 d, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    FastWalk<STATS> w;
    w.begin(p, o, d);
    while (w.step(p, stack, stride, tl)) {
    }
    return w.h;
}

// MULTI = false: the caller only ever sees trees of at most one node (k_trace, the Cornell kernel): the walks of
// multi-node trees are left out of its code altogether -- inlined there they cost the hot loop registers (their
// live ranges pushed 73 VGPRs of k_trace into scratch: 74 B of HBM traffic per segment, -10 %).
template <bool STATS, bool MULTI = true>
DEV TriHit intersect_bvh(const KParams& p, f3 o, f3 d, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    TriHit h;
    h.hit = false;
    h.t = 1e20f;
    h.u = 0.0f;
    h.v = 0.0f;
    h.slot = 0u;
    const uint32_t node_count = p.u.bvh_node_count;
    if (node_count == 0u) return h;
    if constexpr (MULTI) {
        if (p.fast_nodes != nullptr) return intersect_bvh_fast<STATS>(p, o, d, stack, stride, tl);
    }
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
            for (uint32_t slot = first; slot < end; slot++) {
                const v4f a = ptris[slot * 4u], b = ptris[slot * 4u + 1u], c = ptris[slot * 4u + 2u];
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
    if constexpr (!MULTI) return h;

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

}  // namespace
}  // namespace rb
