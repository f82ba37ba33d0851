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
