// rb_device_shade.hpp -- one iteration of the bounce loop (closest-hit resolution + shading), camera / primary rays, pixel store.
// Part of the single device translation unit rb_kernels.hip (numerics contract: see there).
#pragma once
#include "rb_device_intersect.hpp"

#pragma clang fp contract(off)

namespace rb {
namespace {

// ------------------------------------------------------------- materials --
// The launch parameters, fetched again from the kernel-argument segment (every render kernel takes one KParams by
// value, so it sits at offset 0).  A kernel that keeps every field it will ever need in scalar registers across its
// whole loop runs out of them (102 per wave): the allocator spills to lanes of a vector register and every use
// costs a v_readlane, a vector-unit slot.  Starting a phase of the loop from a pointer the compiler cannot see
// through makes that phase load what it needs with scalar loads (the scalar cache holds them) and ends the live
// ranges at the phase's end.  k_trace on C2: 411 -> 78 v_readlane in the loop body, + 5.6 % (profiles/r02_fresh_params.txt).
DEV const KParams& fresh_params(const KParams&) {
    const RB_CONST char* k = (const RB_CONST char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const KParams*)k;
}

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

// Sphere acceleration structure (rb_bvh.cpp sphere_bvh_build on the host, rb_build.hip device_sphere_bvh_build on the
// device): closest sphere with the semantics of the reference's linear scan (shader.wgsl:574-586).
//  * every candidate is evaluated with the reference's exact intersect_sphere;
//  * the scan accepts `t > 0.001 && t < closest.t` in index order, i.e. the winner is the
//    smallest t below the incoming closest_t, ties going to the lowest index: here
//    `t < best || (t == best && id < best_id)`;
//  * a subtree is skipped only if no sphere below it can be REPORTED hit nearer than the best t so far.  The reference's
//    discriminant hb^2 - a (|oc|^2 - r^2) carries an absolute error <= E u a D^2 (u = 2^-24; D >= max(|oc|, r)) with E <= 22.0:
//    tools/margin_certify.py derives it -- the standard model of rounding through the seven operations, in exact rational
//    arithmetic -- and checks that sqrt(E u) plus everything else the walk can lose stays below kSphK (92 % of it);
//    tools/sphere_margin_check.py samples 8.9 at most over 43 M reported hits.  So
//      across the ray: a sphere is reported hit only by a ray whose LINE passes within r + sqrt(E u) D of its centre:
//        the line's point nearest the centre (parameter t_c) lies in the sphere's box grown by mm = kSphK D;
//      along the ray: the reported t^ = t_c -+ sqrt(disc^) / a is within dt = kSphK D / |d| of a point of the chord
//        (or of t_c itself when the exact line misses the sphere); every ray direction is the output of normalize(), |d| = 1 +- 4 u.
//    So with [tn, tf] = where the line is inside the child's box grown by mm, every reported t^ of a sphere below lies
//    in [tn - dt, tf + dt]; it must be positive and, to win, below the best t.  D = distance from the origin to the box's
//    farthest corner >= max(|oc|, r) for every sphere inside the box.  kSphK = 1.25e-3 = sqrt(26 u): both parts of
//    r03's single 3e-3 D box inflation, each now where it belongs.
constexpr float kSphK = 1.25e-3f * 1.001f;   // * 1.001: v_sqrt_f32 / v_rsq_f32 are within 1 ulp, the slab arithmetic a few more
constexpr float kSphAbs = 1e-4f;             // absolute part: the boxes' own rounding (c -+ r in f32), tiny scenes
// one child box (centre c, half extents h): enter?  `tn` = where the line enters the grown box.  NaN anywhere means "enter" (a ray
// parallel to a slab it is outside of gets inf - inf there and is not rejected by that axis: it visits, which is always correct).
DEV bool sphere_child(float cx, float cy, float cz, float hx, float hy, float hz, f3 o, f3 inv, float best, float& tn) {
    const f3 a = mk(cx, cy, cz) - o;
    const float mx = fabsf(a.x) + hx, my = fabsf(a.y) + hy, mz = fabsf(a.z) + hz;   // the farthest corner, per axis
    // (D by the sum of the components instead -- two adds for three fma and a square root -- tests 2.3 % more spheres for 0.6 %: not taken)
    const float D = __builtin_amdgcn_sqrtf(__builtin_fmaf(mx, mx, __builtin_fmaf(my, my, mz * mz)));
    const float mm = __builtin_fmaf(kSphK, D, kSphAbs), dt = mm;   // dt = kSphK D / |d|, and |d| = 1 +- 4 u: inside kSphK's 1.001
    const f3 tc = a * inv;
    const float ex = (hx + mm) * fabsf(inv.x), ey = (hy + mm) * fabsf(inv.y), ez = (hz + mm) * fabsf(inv.z);
    tn = fmaxf(fmaxf(tc.x - ex, tc.y - ey), tc.z - ez);
    const float tf = fminf(fminf(tc.x + ex, tc.y + ey), tc.z + ez);
    return !(tf < tn) && !(tf < -dt) && !(tn - dt > best);
}

// Leaf reference: bit 31, (count - 1) << 27, first position in leaf order; at most kSphLeaf spheres, consecutive
// 16-byte {centre, radius} records (sph_leaf) with their original indices beside them (sph_id).
DEV uint32_t sph_leaf_first(uint32_t ref) { return ref & 0x07FFFFFFu; }
DEV uint32_t sph_leaf_count(uint32_t ref) { return ((ref >> 27) & 15u) + 1u; }
// which components of the direction are negative: a node's children are visited in the order of these signs
DEV uint32_t sph_dir_signs(f3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }

// One 4-wide node (rb_internal.hpp SphereNode4): the up to four child boxes against the ray; returns the entered child the
// ray meets first (kSphNone: none) and hands the other entered ones to `push`, farthest first -- "first" by the split
// planes: the half on the ray's side of the top split before the other, and inside a half the child on its side of that
// half's split (any order is correct: the walk culls on the best t; this one costs a few selects instead of a sort).
template <class Push>
DEV uint32_t sphere_node_step(const KParams& p, uint32_t node, f3 o, f3 inv, uint32_t dneg, float best, Push&& push) {
    const cf4p q = (cf4p)p.sph_nodes + (size_t)node * 8u;
    const v4f cx = q[0], cy = q[1], cz = q[2], hx = q[3], hy = q[4], hz = q[5];
    const v4u rf = ((cu4p)p.sph_nodes)[(size_t)node * 8u + 6u], mt = ((cu4p)p.sph_nodes)[(size_t)node * 8u + 7u];
    float t0, t1, t2, t3;
    // a child that is not entered (or is not there) becomes "no child": the order below then moves one word per child
    uint32_t r0 = sphere_child(cx.x, cy.x, cz.x, hx.x, hy.x, hz.x, o, inv, best, t0) ? rf.x : kSphNone;
    uint32_t r1 = sphere_child(cx.y, cy.y, cz.y, hx.y, hy.y, hz.y, o, inv, best, t1) ? rf.y : kSphNone;
    uint32_t r2 = sphere_child(cx.z, cy.z, cz.z, hx.z, hy.z, hz.z, o, inv, best, t2) ? rf.z : kSphNone;
    uint32_t r3 = sphere_child(cx.w, cy.w, cz.w, hx.w, hy.w, hz.w, o, inv, best, t3) ? rf.w : kSphNone;
    const uint32_t ax = mt.x;
    const bool f0 = ((dneg >> (ax & 3u)) & 1u) != 0u, f1 = ((dneg >> ((ax >> 2) & 3u)) & 1u) != 0u, f2 = ((dneg >> ((ax >> 4) & 3u)) & 1u) != 0u;
    auto swap_if = [](bool c, uint32_t& a, uint32_t& b) {
        const uint32_t t = c ? b : a;
        b = c ? a : b;
        a = t;
    };
    swap_if(f1, r0, r1);   // a ray going down the lower half's axis meets child 1 first
    swap_if(f2, r2, r3);
    swap_if(f0, r0, r2);   // ... and one going down the top split's axis the upper half
    swap_if(f0, r1, r3);
    // r0 .. r3 are now in the order the ray meets them: the first child there is walked next, the others wait, farthest first
    const bool h0 = r0 != kSphNone, h1 = r1 != kSphNone, h2 = r2 != kSphNone;
    if (r3 != kSphNone && (h0 || h1 || h2)) push(r3);
    if (h2 && (h0 || h1)) push(r2);
    if (h1 && h0) push(r1);
    return h0 ? r0 : h1 ? r1 : h2 ? r2 : r3;
}

// The per-lane walk, run to completion: the per-segment kernels and segment_finish of the mesh walks (k_trace_sph has
// its own pooled form of the leaves, rb_kernels.hip).
DEV void intersect_spheres_bvh(const KParams& p, f3 o, f3 d, float a, float& closest_t, uint32_t& sphere_idx,
                               uint32_t* stack, uint32_t stride, unsigned long long* n_tested) {
    const f3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));   // steers only (1 ulp: within the margin)
    const uint32_t dneg = sph_dir_signs(d);
    const cf4p leafs = (cf4p)p.sph_leaf;
    const RB_CONST uint32_t* ids = cptr(p.sph_id);
    float best = closest_t;
    uint32_t best_id = 0xFFFFFFFFu, cur = p.sph_root;
    int sp = 0;
    for (;;) {
        if (cur != kSphNone && (cur & 0x80000000u) == 0u) {
            cur = sphere_node_step(p, cur, o, inv, dneg, best, [&](uint32_t ref) {
                stack[sp * stride] = ref;
                sp++;
            });
            if (cur != kSphNone) continue;
        } else if (cur != kSphNone) {
            const uint32_t first = sph_leaf_first(cur), count = sph_leaf_count(cur);
            for (uint32_t j = first; j < first + count; j++) {
                const v4f cr = leafs[j];
                const uint32_t id = ids[j];
                if (n_tested) (*n_tested)++;
                const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);
                // strictly nearer than everything so far; equal t only displaces a SPHERE of higher index (the scan's order):
                // a tie with the ground / triangle hit the walk started from loses, as in the reference (best_id still unset)
                if (t > 0.001f && (t < best || (t == best && best_id != 0xFFFFFFFFu && id < best_id))) {
                    best = t;
                    best_id = id;
                }
            }
        }
        if (sp == 0) break;
        sp--;
        cur = stack[sp * stride];
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
// segment_finish in three parts, so that the stepped sphere kernel (k_trace_sph) can run the sphere
// walk between them one step at a time.
struct SegState {
    float closest_t;
    uint32_t kind;
    float uvx, uvy;   // closest_hit.uv / use_texture after the ground + BVH stage
    bool use_tex, tri_won_a;
};

// Ground and the BVH winner, shader.wgsl:552-571
template <bool STATS>
DEV SegState segment_pre(const KParams& p, const Path& pt, const TriHit th, Tally<STATS>& tl) {
    const f3 o = pt.o, d = pt.d;
    tl.segments++;

    float closest_t = 1e20f;
    uint32_t kind = K_NONE;
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

    SegState st;
    st.closest_t = closest_t;
    st.kind = kind;
    st.uvx = uvx;
    st.uvy = uvy;
    st.use_tex = use_tex;
    st.tri_won_a = tri_won_a;
    return st;
}

// Spheres :574-586 and point lights :590-601.  Two passes with the reference's arithmetic:
// pass 1 evaluates the discriminant of every sphere with wave-uniform scalar loads and
// records the candidates (disc >= 0) in a per-lane bit mask; pass 2 runs the sqrt/divide
// tail only for a lane's own candidates, in ascending index order, so the strict `<`
// keeps the same winner.  Most lanes have no candidate, so the expensive tail is issued
// once or twice per segment instead of once per sphere.
// Spheres, shader.wgsl:574-586 -- per-segment form (the sphere tree run to completion, or the scan)
// SPHTREE = false: an instantiation for launches without a sphere tree (at most 64 spheres): the tree's walk, with its 32
// registers of node, stays out of the kernel's register allocation (k_trace for C2: 32 spilled registers with it, 4 without)
template <bool STATS, bool SPHTREE = true>
DEV void segment_spheres(const KParams& p, f3 o, f3 d, float a, float& closest_t, uint32_t& sphere_idx, uint32_t* stack,
                         uint32_t stride, Tally<STATS>& tl) {
    const uint32_t ns = p.u.spheres_count;
    const cf4p sph4 = (cf4p)p.spheres;  // 96 B = 6 x float4 per sphere; [0] = centre, radius
    if constexpr (SPHTREE) {
        if (p.sph_nodes != nullptr) {
            unsigned long long* cnt = nullptr;
            if constexpr (STATS) cnt = &tl.spheres;
            intersect_spheres_bvh(p, o, d, a, closest_t, sphere_idx, stack, stride, cnt);
            return;
        }
    }
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
}

// Point lights, sky, the winner's HitRecord, shading and scatter, shader.wgsl:590-660
template <bool STATS>
DEV bool segment_post(const KParams& p, Path& pt, const TriHit th, const SegState st, float closest_t, uint32_t sphere_idx,
                      Tally<STATS>& tl) {
    const f3 o = pt.o, d = pt.d;
    const float a = dot(d, d);
    uint32_t kind = st.kind;
    float uvx = st.uvx, uvy = st.uvy;
    bool use_tex = st.use_tex;
    const bool tri_won_a = st.tri_won_a;
    [[maybe_unused]] const uint32_t ns = p.u.spheres_count;
    [[maybe_unused]] const cf4p sph4 = (cf4p)p.spheres;
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

// One iteration of the bounce loop after the triangle traversal (`th`: its winner).
template <bool STATS, bool SPHTREE = true>
DEV bool segment_finish(const KParams& p, Path& pt, const TriHit th, uint32_t* stack, uint32_t stride,
                        Tally<STATS>& tl) {
    const SegState st = segment_pre<STATS>(fresh_params(p), pt, th, tl);
    float closest_t = st.closest_t;
    uint32_t sphere_idx = 0xFFFFFFFFu;
    segment_spheres<STATS, SPHTREE>(fresh_params(p), pt.o, pt.d, dot(pt.d, pt.d), closest_t, sphere_idx, stack, stride, tl);
    return segment_post<STATS>(fresh_params(p), pt, th, st, closest_t, sphere_idx, tl);
}

// One whole iteration of the bounce loop: traversal + everything else.  MULTI = false: trees of at most one node AND no
// sphere tree (k_trace's default instantiations).
template <bool STATS, bool MULTI = true>
DEV bool segment(const KParams& p, Path& pt, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    const TriHit th = intersect_bvh<STATS, MULTI>(fresh_params(p), pt.o, pt.d, stack, stride, tl);
    return segment_finish<STATS, MULTI>(p, pt, th, stack, stride, tl);
}

// ----------------------------------------------------------------- camera --
// shader.wgsl:693-709; `sample_hash` = hash(current_pass * samples_per_pass + sample), the part of the seed
// that does not depend on the pixel
// (p.cam: rb_internal.hpp; callers pass fresh_params(p) so that the camera is not held in registers between path starts)
DEV void start_path_hashed(const KParams& p, uint32_t x, uint32_t y, uint32_t pixel_index, uint32_t sample_hash, Path& pt) {
    const Cam& c = p.cam;
    uint32_t seed = pcg(pixel_index + sample_hash);
    const float off_x = rnd(seed) - 0.5f;
    const float off_y = rnd(seed) - 0.5f;
    // the two divisions by launch constants: the exact quotient from the exact reciprocal (div_newton; the numerators
    // are zero or at least 2^-33 in magnitude, multiples of the sampler's 2^-32 grid), the compiler's expansion otherwise
    const float ax = (float)x + off_x, ay = (float)y + off_y;
    float qx, qy;
    if (c.fast_wh) {
        qx = __builtin_copysignf(div_newton(ax, c.wm1, c.inv_wm1), ax);
        qy = __builtin_copysignf(div_newton(ay, c.hm1, c.inv_hm1), ay);
    } else {
        qx = ax / c.wm1;
        qy = ay / c.hm1;
    }
    const float u = ((qx * 2.0f) - 1.0f) * c.aspect;
    const float v = 1.0f - qy * 2.0f;
    pt.o = ld3(c.pos);
    pt.d = normalize(((c.fov * u) * ld3(c.right) + (c.fov * v) * ld3(c.up)) + ld3(c.fwd));
    pt.seed = seed;
    pt.color = mk(0, 0, 0);
    pt.att = mk(1, 1, 1);
    pt.depth = 0;
}

DEV void start_path(const KParams& p, uint32_t x, uint32_t y, uint32_t pixel_index, uint32_t sample_offset, Path& pt) {
    start_path_hashed(p, x, y, pixel_index, pcg(sample_offset), pt);
}

// global image row of local row `ly` (interleaved stripes)
DEV uint32_t global_row(const KParams& p, uint32_t ly) {
    if (p.shard_count <= 1u) return ly;
    if (p.stripe_rows == 1u) return ly * p.shard_count + p.shard_rank;  // no division for one-row stripes
    const uint32_t s = ly / p.stripe_rows, r = ly % p.stripe_rows;
    return (s * p.shard_count + p.shard_rank) * p.stripe_rows + r;
}

// shader.wgsl:716-722 + the x mirror of gpu_wrapper.rs:446-458
DEV void store_pixel(const KParams& p, uint32_t x, uint32_t ly, f3 acc, uint32_t total_samples) {
    const size_t li = (size_t)ly * p.u.width + x;
    const float ts = (float)total_samples;
    reinterpret_cast<float4*>(p.accum_out)[li] = make_float4(acc.x, acc.y, acc.z, ts);
    const f3 fin = divs(acc, ts);
    const f3 mapped = mk(fin.x / (fin.x + 1.0f), fin.y / (fin.y + 1.0f), fin.z / (fin.z + 1.0f));
    p.out_rgba[(size_t)ly * p.u.width + (p.u.width - 1u - x)] = color_map(mapped);
}

// ---- (pixel, sample) items of the stream kernels.  Item = (tile * S + sample) * 64 + pixel-in-tile.
// A refill round hands out at most 64 consecutive items starting at the wave-uniform `base`, so a
// lane's item lies in the 64-item row of `base` or in the next one: tile, sample and the sample's
// hash are worked out once per round for those two rows with scalar arithmetic (division by the
// launch constants via host-made reciprocals), and a lane only selects between them.
DEV uint32_t udiv_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t& rem) {
    uint32_t q = __umulhi(n, magic);  // magic = floor(2^32 / d) (0xFFFFFFFF for d = 1): q is exact or one low
    uint32_t r = n - q * d;
    if (r >= d) {
        q++;
        r -= d;
    }
    rem = r;
    return q;
}
struct ItemRows {
    uint32_t in0;
    uint32_t tx[2], ty[2], hs[2];
};
DEV ItemRows item_rows(const KParams& p, uint32_t base, uint32_t S, uint32_t tiles_x, uint32_t sample_base) {
    ItemRows r;
    r.in0 = base & 63u;
    uint32_t smp, tx;
    const uint32_t tile = udiv_magic(base >> 6, S, p.magic_S, smp);
    const uint32_t ty = udiv_magic(tile, tiles_x, p.magic_tiles_x, tx);
    r.tx[0] = tx;
    r.ty[0] = ty;
    r.hs[0] = pcg(sample_base + smp);
    uint32_t smp1 = smp + 1u, tx1 = tx, ty1 = ty;
    if (smp1 == S) {
        smp1 = 0u;
        tx1++;
        if (tx1 == tiles_x) {
            tx1 = 0u;
            ty1++;
        }
    }
    r.tx[1] = tx1;
    r.ty[1] = ty1;
    r.hs[1] = pcg(sample_base + smp1);
    return r;
}
// this lane's pixel for the item `base + rank`; false for the padding pixels of edge tiles and stripes
DEV bool item_pixel(const KParams& p, const ItemRows& r, uint32_t rank, uint32_t& x, uint32_t& y, uint32_t& sample_hash) {
    const uint32_t idx = r.in0 + rank, in = idx & 63u;
    const bool next = idx >= 64u;
    const uint32_t tx = next ? r.tx[1] : r.tx[0], ty = next ? r.ty[1] : r.ty[0];
    sample_hash = next ? r.hs[1] : r.hs[0];
    x = tx * 8u + (in & 7u);
    const uint32_t ly = ty * 8u + (in >> 3);
    if (!((x < p.u.width) && (ly < p.local_rows))) return false;
    y = global_row(p, ly);
    return y < p.u.height;
}

}  // namespace
}  // namespace rb
