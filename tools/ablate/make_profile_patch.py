"""Regenerates tools/ablate/rb_profile.patch from the current product sources: the counting code of the profiling build
(tools/walk_profile.sh) is written HERE, as edits of a copy of renderbaby_amd/csrc, and shipped as the diff -- so that it can
be re-made when the sources move and never lives in them.     python tools/ablate/make_profile_patch.py"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = "/tmp/rb_profile_patch"
shutil.rmtree(W, ignore_errors=True)
for side in "ab":
    shutil.copytree(os.path.join(ROOT, "renderbaby_amd", "csrc"), os.path.join(W, side, "renderbaby_amd", "csrc"))
R = os.path.join(W, "b", "renderbaby_amd", "csrc") + "/"


def edit(name, pairs):
    s = open(R + name).read()
    for old, new, *cnt in pairs:
        assert old in s, (name, old[:60])
        s = s.replace(old, new, *cnt)
    open(R + name, "w").write(s)


edit("rb_device_common.hpp", [("#define DEV __device__ __forceinline__", '''#define DEV __device__ __forceinline__
// ---- tools/ablate/rb_profile.patch: pass / phase occupancy counters (never in the product build)
#define RB_WALK_PROFILE 1
static __device__ unsigned long long g_walk_prof[64];
// one count and the active lanes at this point of the code, whatever the divergence: slot 2 i = executions, 2 i + 1 = lanes
#define KPROF(i) do { const unsigned long long m_ = __ballot(1); \\
    if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)(__ffsll((long long)m_) - 1)) { \\
        atomicAdd(&g_walk_prof[2 * (i)], 1ull); atomicAdd(&g_walk_prof[2 * (i) + 1], (unsigned long long)__popcll(m_)); } } while (0)
''', 1)])
# k_trace's phases.  pairs of slots: 8 path start, 9 segment, 10 triangle test, 11 past |a|, 12 past u, 13 past v, 14 sphere pass 1,
# 15 sphere pass 2, 16 light pass 1, 17 light pass 2, 18 shading, 19 one try of the rejection loop, 20 metal, 21 lambert, 22 texture, 23 path end
edit("rb_device_math.hpp", [('''    for (;;) {
        float px = rnd(seed) * 2.0f - 1.0f;''', '''    for (;;) {
        KPROF(19);
        float px = rnd(seed) * 2.0f - 1.0f;''')])
edit("rb_device_intersect.hpp", [('''    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    if (fabsf(a) < 1e-6f) return -1.0f;
    const float f = rcp_tri(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;''', '''    KPROF(10);
    const f3 h = cross(d, edge2);
    const float a = dot(edge1, h);
    if (fabsf(a) < 1e-6f) return -1.0f;
    KPROF(11);
    const float f = rcp_tri(a);
    const f3 s = o - v0;
    const float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    KPROF(12);
    const f3 q = cross(s, edge1);
    const float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    KPROF(13);''')])
edit("rb_device_shade.hpp", [
    ('''        for (uint32_t k = 0; k < n; k++, sp_ += 6) {
            const v4f cr = sp_[0];''', '''        for (uint32_t k = 0; k < n; k++, sp_ += 6) {
            KPROF(14);
            const v4f cr = sp_[0];'''),
    ('''            const v4f cr = sph4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);''', '''            KPROF(15);
            const v4f cr = sph4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);'''),
    ('''            const v4f cr = lgt4[(base + k) * 6u];
            if constexpr (STATS) tl.lights++;''', '''            KPROF(16);
            const v4f cr = lgt4[(base + k) * 6u];
            if constexpr (STATS) tl.lights++;'''),
    ('''            const v4f cr = lgt4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);''', '''            KPROF(17);
            const v4f cr = lgt4[(base + k) * 6u];
            const float t = isect_sphere(o, d, a, mk(cr.x, cr.y, cr.z), cr.w);'''),
    ('''    // ---- resolve the winner's HitRecord fields (:555-563, :348-372, :579-584, :595-599)
    const f3 pos = o + closest_t * d;''', '''    KPROF(18);
    // ---- resolve the winner's HitRecord fields (:555-563, :348-372, :579-584, :595-599)
    const f3 pos = o + closest_t * d;'''),
    ('''    if (is_metal) {
        const f3 reflected = reflect_vector(normalize(d), normal);''', '''    if (is_metal) {
        KPROF(20);
        const f3 reflected = reflect_vector(normalize(d), normal);'''),
    ('''    } else {
        const f3 sd = normal + ruv;''', '''    } else {
        KPROF(21);
        const f3 sd = normal + ruv;'''),
    ('''            if (tri_won_a) tri_uv(p, th, uvx, uvy);
            albedo = albedo * sample_texture(p, m.tex, uvx, uvy);''', '''            KPROF(22);
            if (tri_won_a) tri_uv(p, th, uvx, uvy);
            albedo = albedo * sample_texture(p, m.tex, uvx, uvy);'''),
    ('''DEV void start_path_hashed(const KParams& p, uint32_t x, uint32_t y, uint32_t pixel_index, uint32_t sample_hash, Path& pt) {
    const Cam& c = p.cam;''', '''DEV void start_path_hashed(const KParams& p, uint32_t x, uint32_t y, uint32_t pixel_index, uint32_t sample_hash, Path& pt) {
    KPROF(8);
    const Cam& c = p.cam;'''),
    ('''DEV bool segment(const KParams& p, Path& pt, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
''', '''DEV bool segment(const KParams& p, Path& pt, uint32_t* stack, uint32_t stride, Tally<STATS>& tl) {
    KPROF(9);
''')])
# k_trace's path end; k_trace_sph's passes (slots 0..11 as plain counters, summed per wave)
s = open(R + "rb_kernels.hip").read()
s = s.replace('''                cring.finish(colors, item, pt.color);
                tl.paths++;''', '''                KPROF(23);
                cring.finish(colors, item, pt.color);
                tl.paths++;''')
# chunk_child: how wide the margins are that the chunked walk culls with -- slots 56..61: child tests, of them with no cone bound for the
# ray (the determinant floor's margin), with a relative margin 12u F above 1e-2 / 1e-3 / 1e-4, and entered
s = s.replace('''    order = tmin;
    return !(tf < tn) && !(tf < -dt) && !(tn - dt > best_t);''', '''    order = tmin;
    {
        const float rel = kChunkKP * f;
        const bool ent = !(tf < tn) && !(tf < -dt) && !(tn - dt > best_t);
        const unsigned long long all_ = __ballot(1);
        const bool first_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)(__ffsll((long long)all_) - 1);
        const unsigned long long c1 = __ballot(!(lb > 1e-6f)), c2 = __ballot(!(rel <= 1e-2f)), c3 = __ballot(!(rel <= 1e-3f)), c4 = __ballot(!(rel <= 1e-4f)), c5 = __ballot(ent);
        if (first_) {
            atomicAdd(&g_walk_prof[56], (unsigned long long)__popcll(all_));
            atomicAdd(&g_walk_prof[57], (unsigned long long)__popcll(c1));
            atomicAdd(&g_walk_prof[58], (unsigned long long)__popcll(c2));
            atomicAdd(&g_walk_prof[59], (unsigned long long)__popcll(c3));
            atomicAdd(&g_walk_prof[60], (unsigned long long)__popcll(c4));
            atomicAdd(&g_walk_prof[61], (unsigned long long)__popcll(c5));
        }
    }
    return !(tf < tn) && !(tf < -dt) && !(tn - dt > best_t);''')
# ... and of the children that ARE chunks (slots 62, 63: tests, tests without a cone bound for the ray)
s = s.replace('''    const bool vl = lref != kChunkNone && chunk_child(l0, l1, __float_as_uint(r0.w), lc, exact, o, d, inv, best_t, kl);''', '''    {
        const bool ll = lref != kChunkNone && (lref & kChunkLeaf) != 0u, rl_ = rref != kChunkNone && (rref & kChunkLeaf) != 0u;
        const unsigned n_leaf = (unsigned)__popcll(__ballot(ll)) + (unsigned)__popcll(__ballot(rl_));
        const unsigned n_nocone = (unsigned)__popcll(__ballot(ll && !(cone_cos_bound(d, lc) > 1e-6f))) + (unsigned)__popcll(__ballot(rl_ && !(cone_cos_bound(d, rc) > 1e-6f)));
        const unsigned long long all_ = __ballot(1);
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (unsigned)(__ffsll((long long)all_) - 1)) {
            atomicAdd(&g_walk_prof[62], (unsigned long long)n_leaf);
            atomicAdd(&g_walk_prof[63], (unsigned long long)n_nocone);
        }
    }
    const bool vl = lref != kChunkNone && chunk_child(l0, l1, __float_as_uint(r0.w), lc, exact, o, d, inv, best_t, kl);''')
# k_trace_chunk's leaf rounds: what a slab along the chunk's (first triangle's) normal would cull of the (ray, chunk) pairs the boxes
# let through -- slots 48.. as plain counters: pairs, pairs with a hit, pairs whose (margin-grown) box the ray no longer enters
# before the best t, pairs the slab culls beyond that for margin 0 / 1e-3 S / 7e-3 S, culled pairs that had a hit (must be 0)
probe_old = '''                    if (r.valid && t > 0.001f) {
                        const unsigned long long k = ((unsigned long long)__float_as_uint(t) << 32) | __float_as_uint(r.a.w);'''
probe_new = '''                    {
                        static_assert(kChunkTris == 16u, "the probe reduces over groups of 16 lanes");
                        auto bc = [&](float x, uint32_t from) { return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(from * 4u), __float_as_int(x))); };
                        const uint32_t gl = lane & ~15u;
                        const f3 pv0 = mk(r.a.x, r.a.y, r.a.z), pe1 = mk(r.b.x, r.b.y, r.b.z), pe2 = mk(r.c.x, r.c.y, r.c.z);
                        f3 pn = cross(pe1, pe2);
                        const float pl = sqrtf(dot(pn, pn));
                        pn = pl > 0.0f ? (1.0f / pl) * pn : mk(0.0f, 1.0f, 0.0f);
                        const f3 nn = mk(bc(pn.x, gl), bc(pn.y, gl), bc(pn.z, gl));
                        const f3 q1 = pv0 + pe1, q2 = pv0 + pe2;
                        const float s0 = dot(nn, pv0), s1 = dot(nn, q1), s2 = dot(nn, q2);
                        const float big = 3.0e38f;
                        float v8[8] = {fminf(s0, fminf(s1, s2)), -fmaxf(s0, fmaxf(s1, s2)), fminf(pv0.x, fminf(q1.x, q2.x)), fminf(pv0.y, fminf(q1.y, q2.y)),
                                       fminf(pv0.z, fminf(q1.z, q2.z)), -fmaxf(pv0.x, fmaxf(q1.x, q2.x)), -fmaxf(pv0.y, fmaxf(q1.y, q2.y)), -fmaxf(pv0.z, fmaxf(q1.z, q2.z))};
                        for (int q = 0; q < 8; q++) {
                            float x = r.valid ? v8[q] : big;
                            for (uint32_t sh = 1; sh < 16u; sh <<= 1) x = fminf(x, bc(x, lane ^ sh));
                            v8[q] = x;
                        }
                        const unsigned long long hitm = __ballot(r.valid && t > 0.001f);
                        const bool any_hit = ((hitm >> gl) & 0xFFFFull) != 0ull;
                        const bool leader = (lane & 15u) == 0u && (g0 + lane / kChunkTris) < n_units;
                        if (leader) {
                            const f3 ro = mk(r.r0.x, r.r0.y, r.r0.z), rd = mk(r.r1.x, r.r1.y, r.r1.z);
                            const float tbest = __uint_as_float((uint32_t)(best[r.rl] >> 32));
                            const f3 bmn = mk(v8[2], v8[3], v8[4]), bmx = mk(-v8[5], -v8[6], -v8[7]);
                            const f3 far = mk(fmaxf(fabsf(bmn.x - ro.x), fabsf(bmx.x - ro.x)), fmaxf(fabsf(bmn.y - ro.y), fabsf(bmx.y - ro.y)), fmaxf(fabsf(bmn.z - ro.z), fabsf(bmx.z - ro.z)));
                            const float S = sqrtf(dot(far, far));
                            const float so = dot(nn, ro), sd = dot(nn, rd);
                            auto test = [&](float M, bool& box_in, bool& slab_in) {
                                float tb0 = 0.0f, tb1 = tbest;
                                const float oo[3] = {ro.x, ro.y, ro.z}, dd[3] = {rd.x, rd.y, rd.z}, lo3[3] = {bmn.x, bmn.y, bmn.z}, hi3[3] = {bmx.x, bmx.y, bmx.z};
                                for (int ax = 0; ax < 3; ax++) {
                                    const float iv = 1.0f / dd[ax];
                                    float a0 = ((lo3[ax] - M) - oo[ax]) * iv, a1 = ((hi3[ax] + M) - oo[ax]) * iv;
                                    if (a0 > a1) { const float tt = a0; a0 = a1; a1 = tt; }
                                    if (a0 == a0) tb0 = fmaxf(tb0, a0);
                                    if (a1 == a1) tb1 = fminf(tb1, a1);
                                }
                                box_in = tb0 <= tb1;
                                const float Ms = 1.7321f * M, l0 = v8[0] - Ms, h0 = -v8[1] + Ms;
                                float ts0 = tb0, ts1 = tb1;
                                if (fabsf(sd) > 1e-12f) {
                                    float a0 = (l0 - so) / sd, a1 = (h0 - so) / sd;
                                    if (a0 > a1) { const float tt = a0; a0 = a1; a1 = tt; }
                                    ts0 = fmaxf(ts0, a0);
                                    ts1 = fminf(ts1, a1);
                                } else if (so < l0 || so > h0) {
                                    ts1 = -1.0f;
                                }
                                slab_in = box_in && ts0 <= ts1;
                            };
                            bool b0, s0_, b1, s1_, b2, s2_;
                            test(0.0f, b0, s0_);
                            test(1e-3f * S, b1, s1_);
                            test(7e-3f * S, b2, s2_);
                            atomicAdd(&g_walk_prof[48], 1ull);
                            if (any_hit) atomicAdd(&g_walk_prof[49], 1ull);
                            if (!b1) atomicAdd(&g_walk_prof[50], 1ull);
                            if (b0 && !s0_) atomicAdd(&g_walk_prof[51], 1ull);
                            if (b1 && !s1_) atomicAdd(&g_walk_prof[52], 1ull);
                            if (b2 && !s2_) atomicAdd(&g_walk_prof[53], 1ull);
                            if (any_hit && b1 && !s1_) atomicAdd(&g_walk_prof[54], 1ull);
                            if (any_hit && !b1) atomicAdd(&g_walk_prof[55], 1ull);
                        }
                    }
''' + probe_old
assert s.count(probe_old) == 1
s = s.replace(probe_old, probe_new)
a = s.index('template <bool STATS>\n__global__ void __launch_bounds__(kTraceBlock, RB_SPH_WAVES) k_trace_sph(const KParams p) {')
b = s.index('// Phase 2: ordered accumulation + tone map + pack.')
k = s[a:b]
for old, new, *cnt in [
    ('''    auto set_aside = [&]() {''', '''    unsigned long long prof[12] = {0};   // outer iterations; begin passes, lanes; node passes, lanes; leaf phases, pairs, rounds; flushes, survivors; finish passes, lanes
    auto set_aside = [&]() {''', 1),
    ('''        // ---- (2) start of a segment: ground and the triangle list''', '''        prof[0]++;
        { const uint32_t nb = (uint32_t)__popcll(__ballot(state == BEGIN)); if (nb) { prof[1]++; prof[2] += nb; } }
        // ---- (2) start of a segment: ground and the triangle list'''),
    ('''            if (n == 0u || (it > 0 && n < (uint32_t)RB_SPH_NODE_LANES)) break;
''', '''            if (n == 0u || (it > 0 && n < (uint32_t)RB_SPH_NODE_LANES)) break;
            prof[3]++; prof[4] += n;
'''),
    ('''                const unsigned long long below = (1ull << lane) - 1ull;
''', '''                const unsigned long long below = (1ull << lane) - 1ull;
                prof[5]++; prof[6] += n_units; prof[7] += (n_units + 64u * kSphPerLane / kSphLeaf - 1u) / (64u * kSphPerLane / kSphLeaf);
''', 1),
    ('''                auto flush = [&](uint32_t k) {
''', '''                auto flush = [&](uint32_t k) {
                    prof[8]++; prof[9] += k;
'''),
    ('''            if (n_fin != 0u && (n_fin >= (uint32_t)RB_SPH_FINISH_LANES || n_trav == 0u) && state == FINISH) {
                float closest_t = st.closest_t;''', '''            if (n_fin != 0u && (n_fin >= (uint32_t)RB_SPH_FINISH_LANES || n_trav == 0u)) { prof[10]++; prof[11] += n_fin; }
            if (n_fin != 0u && (n_fin >= (uint32_t)RB_SPH_FINISH_LANES || n_trav == 0u) && state == FINISH) {
                float closest_t = st.closest_t;''')]:
    assert old in k, old[:60]
    k = k.replace(old, new, *cnt)
j = k.rindex('    flush_tally<STATS>(tl, p.counters);\n}')
k = k[:j] + '''    flush_tally<STATS>(tl, p.counters);
    if (lane == 0u) for (int i = 0; i < 12; i++) atomicAdd(&g_walk_prof[i], prof[i]);
}''' + k[j + len('    flush_tally<STATS>(tl, p.counters);\n}'):]
s = s[:a] + k + s[b:]
old = '''#ifndef RB_WALK_PROFILE
int debug_walk_profile(unsigned long long*, int) { return -1; }
#endif'''
assert old in s
s = s.replace(old, '''#ifndef RB_WALK_PROFILE
int debug_walk_profile(unsigned long long*, int) { return -1; }
#else
int debug_walk_profile(unsigned long long* out64, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_walk_prof), 64 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[64] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_walk_prof), z, sizeof(z));
    }
    return (int)e;
}
#endif''')
open(R + "rb_kernels.hip", "w").write(s)
out = subprocess.run(["diff", "-ru", "a/renderbaby_amd/csrc", "b/renderbaby_amd/csrc"], cwd=W, capture_output=True, text=True).stdout
lines = []
for l in out.split("\n"):
    if l.startswith("--- a/"): l = "--- " + l[6:]
    elif l.startswith("+++ b/"): l = "+++ " + l[6:]
    elif l.startswith("diff -ru a/"): l = "diff " + l[11:].split(" b/")[0]
    lines.append(l)
open(os.path.join(ROOT, "tools", "ablate", "rb_profile.patch"), "w").write("\n".join(lines))
print("tools/ablate/rb_profile.patch:", len(lines), "lines")
