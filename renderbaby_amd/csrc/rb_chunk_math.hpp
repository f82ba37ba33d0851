// rb_chunk_math.hpp -- the per-triangle and per-subtree quantities of the chunked walk's tree, as
// host + device functions: the host builder (rb_bvh.cpp chunk_tree_build) and the device builder (rb_build.hip
// device_chunk_tree_build) compute the margins' inputs with the same source.  Double arithmetic throughout, no contraction
// (both translation units are compiled with -ffp-contract=off); every result a kernel culls with is rounded UP on the way
// to its stored form, so a last-bit difference between the host's libm and the device's only moves a bound by what the
// (1 + 1e-6) / + 1e-9 slack in these functions already covers.
#pragma once

#include <cmath>
#include <cstdint>

#include "rb_internal.hpp"

namespace rb {
namespace chunkmath {

RB_HD inline double dmin(double a, double b) { return a < b ? a : b; }
RB_HD inline double dmax(double a, double b) { return a > b ? a : b; }
constexpr float kBig = 3.402823466e38f;   // boxes start at +-inf in the host builder; +-FLT_MAX would do, inf is kept
RB_HD inline float f_inf() { return __builtin_huge_valf(); }

// F_k of one triangle and whether it is "large" (see the comment above FBuilder in rb_bvh.cpp), from the f32 edges the
// kernels use (k_prep_tris subtracts in f32, so do these).
RB_HD inline TriBound tri_bound_hd(const float v0[3], const float v1[3], const float v2[3], float small_cap) {
    double e1[3], e2[3], l1 = 0, l2 = 0;
    for (int a = 0; a < 3; ++a) {
        e1[a] = double(v1[a] - v0[a]);
        e2[a] = double(v2[a] - v0[a]);
        l1 += e1[a] * e1[a];
        l2 += e2[a] * e2[a];
    }
    TriBound b;
    const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
    const double nn = sqrt(nx * nx + ny * ny + nz * nz), ll = dmax(l1, l2);
    const double cap = ll * 1e6 * (1.0 + 1e-5);   // L^2 / fl(1e-6), rounded up
    b.has_normal = nn > 0.0 && nn <= 1.7976931348623157e308;
    if (b.has_normal) {
        b.n[0] = nx / nn; b.n[1] = ny / nn; b.n[2] = nz / nn;
    }
    b.large = !(cap <= double(small_cap));
    if (!b.large) b.f = static_cast<float>(cap * (1.0 + 1e-6));
    else if (b.has_normal) b.f = static_cast<float>(ll / nn / (0.95 * double(kFastGrazeCos)) * (1.0 + 1e-5));   // 0.95: |a^| >= 0.95 |a| in the bound's range
    else b.f = f_inf();
    return b;
}

// cone of unit normals: axis c, half-angle alpha; `valid` false = "no useful cone" (wider than ~89 degrees)
struct DCone {
    double c[3] = {0, 0, 0}, alpha = 4.0;
    bool valid = false;
    double cap = 0.0;   // largest L^2 / 1e-6 over the large triangles below
};
RB_HD inline DCone merge(const DCone& a, const DCone& b_) {
    DCone out;
    out.cap = dmax(a.cap, b_.cap);
    if (!a.valid || !b_.valid) return out;
    DCone b = b_;
    if (a.c[0] * b.c[0] + a.c[1] * b.c[1] + a.c[2] * b.c[2] < 0.0)
        for (int i = 0; i < 3; ++i) b.c[i] = -b.c[i];
    double s[3] = {a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]};
    const double len = sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    if (!(len > 1e-9)) return out;
    for (int i = 0; i < 3; ++i) s[i] /= len;
    const double ang_a = acos(dmin(1.0, dmax(-1.0, s[0] * a.c[0] + s[1] * a.c[1] + s[2] * a.c[2])));
    const double ang_b = acos(dmin(1.0, dmax(-1.0, s[0] * b.c[0] + s[1] * b.c[1] + s[2] * b.c[2])));
    out.alpha = dmax(ang_a + a.alpha, ang_b + b.alpha) + 1e-9;
    if (!(out.alpha < 1.55)) { DCone bad; bad.cap = out.cap; return bad; }
    for (int i = 0; i < 3; ++i) out.c[i] = s[i];
    out.valid = true;
    return out;
}
RB_HD inline DCone empty_cone() {
    DCone c;
    c.valid = true;
    c.alpha = -1.0;
    c.c[0] = 1.0;
    return c;
}
RB_HD inline DCone merge_cones(const DCone& a, const DCone& b) {
    if (a.valid && a.alpha < 0.0) return b;
    if (b.valid && b.alpha < 0.0) return a;
    return merge(a, b);
}
// {axis cos(alpha), tan(alpha)} as FastWalk::graze_possible reads it; all zeros = "any ray may graze a triangle below"
RB_HD inline void encode_cone(const DCone& c, float o[4]) {
    o[0] = o[1] = o[2] = o[3] = 0.0f;
    if (!c.valid) return;
    if (c.alpha < 0.0) { o[3] = -1.0f; return; }   // nothing below
    const double cos_a = cos(c.alpha) * (1.0 - 1e-6) - 1e-7;
    if (!(cos_a > 0.0175)) return;
    const double sin_a = sqrt(dmax(0.0, 1.0 - cos_a * cos_a));
    o[0] = static_cast<float>(c.c[0] * cos_a);
    o[1] = static_cast<float>(c.c[1] * cos_a);
    o[2] = static_cast<float>(c.c[2] * cos_a);
    o[3] = static_cast<float>(sin_a / cos_a * (1.0 + 1e-5) + 1e-7);
}
// E7's leading terms carry |e1| |e2| where r02 wrote L^2 = max(|e1|, |e2|)^2: the chunked walk stores its bounds with
// G = |e1| |e2| in place of L^2.  The bounds are claimed while 5.42 u L^2 / |a^| <= 0.05 -- with L^2.  An accepted hit has
// |a^| >= 1e-6, so that holds for every triangle with L^2 / 1e-6 <= 1.5e5 whatever the ray; a child slot with a larger
// triangle below it stores +inf as its floor bound and its |cos| >= c0 bound with L^2, as r02 did, so that the kernel's test
// "F <= 1.5e5" is the proviso itself there (pack_fac).
RB_HD inline double chunk_g(double l1sq, double l2sq) { return sqrt(l1sq) * sqrt(l2sq) * (1.0 + 1e-12); }
// v >= 0 as bf16, rounded up; beyond 1.5e5 (where the bound is not claimed) +inf
RB_HD inline uint32_t bf16_up(double v) {
    if (!(v <= 1.5e5)) return 0x7F80u;
    float f = static_cast<float>(v);
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    if (static_cast<double>(f) < v) b += 1u;   // f >= 0 and finite here: the next float up
    return (b >> 16) + ((b & 0xFFFFu) ? 1u : 0u);
}

// what a subtree hands to its parent
struct ChunkInfo {
    uint32_t ref = kChunkNone;
    double cap = 0.0, fa = 0.0;      // with G = max(|e1| |e2|, L^2 / 4) (chunk_g)
    double cap_l = 0.0, fa_l = 0.0;  // with L^2: what is stored where a triangle below is beyond the range the bounds are claimed for
    DCone cone;
    uint32_t depth = 0;   // internal nodes on the longest path below (= stack entries the walk may need)
    float mn[3] = {f_inf(), f_inf(), f_inf()}, mx[3] = {-f_inf(), -f_inf(), -f_inf()};   // tight bounds of the triangles below
};
RB_HD inline uint32_t pack_fac(const ChunkInfo& i) {
    if (!(i.cap_l * (1.0 + 1e-6) <= 1.5e5)) return (0x7F80u << 16) | bf16_up(i.fa_l);
    return (bf16_up(i.cap * (1.0 + 1e-6)) << 16) | bf16_up(i.fa);
}
// one child slot of a ChunkNode: the box the walk tests (mn, mx), the subtree's reference, margins and cone
RB_HD inline void fill_child(const ChunkInfo& i, const float mn[3], const float mx[3], float bmin[3], uint32_t& ref, float bmax[3],
                             uint32_t& fac, float cone[4]) {
    for (int a = 0; a < 3; ++a) { bmin[a] = mn[a]; bmax[a] = mx[a]; }
    ref = i.ref;
    fac = pack_fac(i);
    // the margins bound the distance of a reported hit from its TRIANGLE; culling on a box needs the triangles
    // inside it.  The reference's builder guarantees that (bvh.rs:100-123), a caller's own tree need not: such a
    // child is always entered (its reference box still decides, exactly, whether the leaf below is reached)
    for (int a = 0; a < 3; ++a)
        if (i.ref != kChunkNone && !(mn[a] <= i.mn[a] && i.mx[a] <= mx[a])) fac = 0x7F807F80u;
    encode_cone(i.cone, cone);
}
// the parent's info from its two children's (either may be absent: ref == kChunkNone with the defaults)
RB_HD inline void combine(const ChunkInfo& l, const ChunkInfo& r, ChunkInfo& out) {
    out.cap = dmax(l.cap, r.cap);
    out.fa = dmax(l.fa, r.fa);
    out.cap_l = dmax(l.cap_l, r.cap_l);
    out.fa_l = dmax(l.fa_l, r.fa_l);
    out.cone = merge_cones(l.ref == kChunkNone ? empty_cone() : l.cone, r.ref == kChunkNone ? empty_cone() : r.cone);
    out.depth = 1 + (l.depth > r.depth ? l.depth : r.depth);
    for (int a = 0; a < 3; ++a) {
        out.mn[a] = l.mn[a] < r.mn[a] ? l.mn[a] : r.mn[a];
        out.mx[a] = l.mx[a] > r.mx[a] ? l.mx[a] : r.mx[a];
    }
}

// one triangle as the builders see it
struct ChunkItem {
    uint32_t slot, rank;
    float mn[3], mx[3];
    double cap, fa, cap_l, fa_l;
    double n[3];
    bool has_normal;
};
RB_HD inline void make_item(const rb_gpu_triangle& t, uint32_t slot, uint32_t rank, ChunkItem& it) {
    it.slot = slot;
    it.rank = rank;
    double ll = 0, l2 = 0;
    for (int a = 0; a < 3; ++a) {
        const float lo = t.v0[a] < t.v1[a] ? t.v0[a] : t.v1[a], hi = t.v0[a] > t.v1[a] ? t.v0[a] : t.v1[a];
        it.mn[a] = lo < t.v2[a] ? lo : t.v2[a];
        it.mx[a] = hi > t.v2[a] ? hi : t.v2[a];
        ll += double(t.v1[a] - t.v0[a]) * double(t.v1[a] - t.v0[a]);   // the f32 edges of k_prep_tris, exactly
        l2 += double(t.v2[a] - t.v0[a]) * double(t.v2[a] - t.v0[a]);
    }
    const TriBound b = tri_bound_hd(t.v0, t.v1, t.v2, 0.0f);   // threshold 0: every triangle keeps its |cos| >= c0 bound (of L^2 / |a|)
    const double g = chunk_g(ll, l2), lmax = dmax(ll, l2);
    it.cap = g * 1e6 * (1.0 + 1e-5);
    it.fa = (b.has_normal && lmax > 0.0) ? double(b.f) * (g / lmax) * (1.0 + 1e-9) : double(b.f);   // no normal: +inf; a point: 0
    it.cap_l = lmax * 1e6 * (1.0 + 1e-5);
    it.fa_l = b.f;
    it.has_normal = b.has_normal;
    for (int a = 0; a < 3; ++a) it.n[a] = b.n[a];
}

// direct cone of a handful of normals (either orientation): axis = normalised sum of the sign-aligned normals.  `item(i)`
// returns the i-th ChunkItem of the chunk.
template <class ItemAt>
RB_HD inline DCone cone_of(const ItemAt& item, uint32_t n) {
    DCone c;
    if (n == 0) return empty_cone();
    double sum[3] = {0, 0, 0};
    const ChunkItem& first = item(0);
    for (uint32_t i = 0; i < n; ++i) {
        const ChunkItem& it = item(i);
        if (!it.has_normal) return c;   // no normal: any direction grazes it
        const double sg = (it.n[0] * first.n[0] + it.n[1] * first.n[1] + it.n[2] * first.n[2]) < 0.0 ? -1.0 : 1.0;
        for (int a = 0; a < 3; ++a) sum[a] += sg * it.n[a];
    }
    const double len = sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
    if (!(len > 1e-9)) return c;
    double cmin = 1.0;
    for (int a = 0; a < 3; ++a) c.c[a] = sum[a] / len;
    for (uint32_t i = 0; i < n; ++i) {
        const ChunkItem& it = item(i);
        cmin = dmin(cmin, fabs(it.n[0] * c.c[0] + it.n[1] * c.c[1] + it.n[2] * c.c[2]));
    }
    c.alpha = acos(dmin(1.0, cmin)) + 1e-9;
    c.valid = c.alpha < 1.55;
    return c;
}

}  // namespace chunkmath
}  // namespace rb

// ---- pieces of the build both builders run on the host (rb_bvh.cpp)
#include <vector>
namespace rb {
bool chunk_visit_order(const rb_bvh_node* ref_nodes, uint32_t node_count, std::vector<uint32_t>& order, std::vector<uint32_t>& leaves);
void chunk_top_pass(const rb_bvh_node* ref_nodes, uint32_t node_count, const std::vector<uint32_t>& order,
                    std::vector<chunkmath::ChunkInfo>& info, std::vector<ChunkNode>& nodes, uint32_t first_index);
}  // namespace rb
