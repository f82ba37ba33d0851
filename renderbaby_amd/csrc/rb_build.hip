// rb_build.hip -- device-side build of the library's own triangle tree (RB_FLAG_DEVICE_BVH).
//
// the reference rebuilds its BVH on the CPU for every render
// (scene_engine_adapter.rs:435-440, engine-bvh/src/bvh.rs:87-150).  The tree the opt-in fast walk
// uses (rb_device_intersect.hpp, intersect_bvh_fast) can instead be built on the GPU from the
// triangles that are already resident there:
//
//   k_lbvh_prims      tight box, |e1||e2| bound and centroid per triangle; mesh / centroid bounds
//   k_lbvh_keys       63-bit Morton code of the centroid (21 bits per axis)
//   rocprim radix sort of (key, item) pairs; equal keys are told apart by their sorted position
//   k_lbvh_hierarchy  one thread per internal node: range and split from common key prefixes
//                     (Karras, "Maximizing parallelism in the construction of BVHs", 2012)
//   k_lbvh_refit      bottom-up: the second thread to reach a node merges its children's boxes
//   k_lbvh_emit       64-B two-box nodes in the format of the host builder (rb_bvh.cpp); an
//                     internal node over exactly two triangles becomes a two-triangle leaf
//
// The tree only steers the walk: which triangle wins, its t/u/v and all tie-breaking follow the
// reference's tree and arithmetic (see intersect_bvh_fast), so frames are the same bits whichever
// builder produced the tree.  A binned-SAH tree (host) is the better tree; this one is built in
// milliseconds.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <utility>

#include "rb_internal.hpp"

#pragma clang fp contract(off)

namespace rb {
namespace {

constexpr uint32_t kLeafTag = 0x80000000u;

__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// bounds[0..2] mesh min, [3..5] mesh max, [6..8] centroid min, [9..11] centroid max (ordered uints)
__global__ void __launch_bounds__(256) k_lbvh_prims(const rb_gpu_triangle* __restrict__ tris,
                                                     const uint32_t* __restrict__ indices,
                                                     const uint32_t* __restrict__ slots,
                                                     const uint32_t* __restrict__ slot_meta, uint32_t n,
                                                     float4* __restrict__ pmin, float4* __restrict__ pmax,
                                                     uint32_t* bounds) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float inf = __builtin_inff();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    float cn[3] = {inf, inf, inf}, cx[3] = {-inf, -inf, -inf};
    if (i < n) {
        const rb_gpu_triangle t = tris[indices[slots[i]]];
        double l1 = 0.0, l2 = 0.0, e1[3], e2[3];
        for (int a = 0; a < 3; ++a) {
            e1[a] = double(t.v1[a] - t.v0[a]);   // the f32 edges of k_prep_tris, exactly
            e2[a] = double(t.v2[a] - t.v0[a]);
            l1 += e1[a] * e1[a];
            l2 += e2[a] * e2[a];
            mn[a] = fminf(t.v0[a], fminf(t.v1[a], t.v2[a]));
            mx[a] = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a]));
            cn[a] = cx[a] = 0.5f * (mn[a] + mx[a]);
        }
        // F_k as the host builder's tri_bound (rb_bvh.cpp), with the host's small / large decision (slot_meta): a
        // small triangle's bound L^2 / 1e-6 covers all its hits, a large one's (L^2 / N) / (0.95 c0) those with
        // |cos| >= c0.  Its largest value below a child scales that child's culling margin (FastWalk::entry).
        const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
        const double nn = sqrt(nx * nx + ny * ny + nz * nz), ll = fmax(l1, l2);
        const bool large = (slot_meta[(size_t)slots[i] * 2u + 1u] & kSlotLarge) != 0u;
        float q = inf;
        if (!large) q = static_cast<float>(ll * 1e6 * (1.0 + 1e-5) * (1.0 + 1e-6));
        else if (nn > 0.0 && nn < 1e300) q = static_cast<float>(ll / nn / (0.95 * double(kFastGrazeCos)) * (1.0 + 1e-5));
        pmin[i] = make_float4(mn[0], mn[1], mn[2], q);
        pmax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
            cn[a] = fminf(cn[a], __shfl_xor(cn[a], off, 64));
            cx[a] = fmaxf(cx[a], __shfl_xor(cx[a], off, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bounds[a], f2ord(mn[a]));
            atomicMax(&bounds[3 + a], f2ord(mx[a]));
            atomicMin(&bounds[6 + a], f2ord(cn[a]));
            atomicMax(&bounds[9 + a], f2ord(cx[a]));
        }
    }
}

__device__ __forceinline__ unsigned long long spread21(uint32_t x) {  // 21 bits -> every third bit
    unsigned long long v = x & 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(256) k_lbvh_keys(const float4* __restrict__ pmin, const float4* __restrict__ pmax,
                                                    uint32_t n, const uint32_t* __restrict__ bounds,
                                                    unsigned long long* __restrict__ keys, uint32_t* __restrict__ items) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = pmin[i], b = pmax[i];
    const float c[3] = {0.5f * (a.x + b.x), 0.5f * (a.y + b.y), 0.5f * (a.z + b.z)};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(bounds[6 + k]), hi = ord2f(bounds[9 + k]);
        const float ext = hi - lo;
        float x = (ext > 0.0f) ? (c[k] - lo) / ext * 2097152.0f : 0.0f;
        x = fminf(fmaxf(x, 0.0f), 2097151.0f);  // NaN -> 0
        q[k] = static_cast<uint32_t>(x);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    items[i] = i;
}

// Internal node i of the n-1 (Karras 2012, section 4).  Equal keys are extended by their sorted
// position, as in the paper, so a run of k identical codes becomes a subtree of depth ~log2 k.
// Children: leaf `pos` is encoded as kLeafTag | pos.
__global__ void __launch_bounds__(256) k_lbvh_hierarchy(const unsigned long long* __restrict__ keys, uint32_t n,
                                                         uint32_t* __restrict__ left, uint32_t* __restrict__ right,
                                                         uint32_t* __restrict__ range_first,
                                                         uint32_t* __restrict__ range_size,
                                                         uint32_t* __restrict__ parent,
                                                         uint32_t* __restrict__ leaf_parent) {
    const int i = static_cast<int>(blockIdx.x * 256u + threadIdx.x);
    const int last = static_cast<int>(n) - 1;
    if (i >= last) return;
    const unsigned long long ki = keys[i];
    auto delta = [&](int j) -> int {
        if (j < 0 || j > last) return -1;
        const unsigned long long x = ki ^ keys[j];
        return x ? __clzll(static_cast<long long>(x)) : 64 + __clz(i ^ j);
    };
    const int d = (delta(i + 1) - delta(i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(i - d);
    int lmax = 2;
    while (delta(i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    if (lo == gamma) {
        left[i] = kLeafTag | static_cast<uint32_t>(gamma);
        leaf_parent[gamma] = static_cast<uint32_t>(i);
    } else {
        left[i] = static_cast<uint32_t>(gamma);
        parent[gamma] = static_cast<uint32_t>(i);
    }
    if (hi == gamma + 1) {
        right[i] = kLeafTag | static_cast<uint32_t>(gamma + 1);
        leaf_parent[gamma + 1] = static_cast<uint32_t>(i);
    } else {
        right[i] = static_cast<uint32_t>(gamma + 1);
        parent[gamma + 1] = static_cast<uint32_t>(i);
    }
    range_first[i] = static_cast<uint32_t>(lo);
    range_size[i] = static_cast<uint32_t>(hi - lo + 1);
    if (i == 0) parent[0] = 0u;
}

// FA of a child = the largest F_k below it (rb_device_intersect.hpp, FastWalk::entry); +inf (always enter) when
// no finite bound exists or beyond the range the bound is claimed for
__device__ __forceinline__ uint32_t fa_bits(float f) {
    return __float_as_uint((f <= 1.5e5f) ? f : __builtin_inff());
}

// nmin[i] = {box min, largest F_k below}, nmax[i] = {box max, height as uint bits}.
__global__ void __launch_bounds__(256) k_lbvh_refit(const uint32_t* __restrict__ items, uint32_t n,
                                                     const float4* __restrict__ pmin, const float4* __restrict__ pmax,
                                                     const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                     const uint32_t* __restrict__ range_size,
                                                     const uint32_t* __restrict__ parent,
                                                     const uint32_t* __restrict__ leaf_parent, uint32_t* flags,
                                                     float4* nmin, float4* nmax) {
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= n) return;
    uint32_t cur = leaf_parent[pos];
    for (;;) {
        __threadfence();
        if (atomicAdd(&flags[cur], 1u) == 0u) return;  // the sibling subtree is not finished yet
        __threadfence();
        float4 lo[2], hi[2];
        uint32_t hgt[2];
        const uint32_t ch[2] = {left[cur], right[cur]};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & kLeafTag) {
                const uint32_t item = items[ch[k] & ~kLeafTag];
                lo[k] = pmin[item];
                hi[k] = pmax[item];
                hgt[k] = 0u;
            } else {
                lo[k] = nmin[ch[k]];  // written by another CU: the fence above has invalidated L1
                hi[k] = nmax[ch[k]];
                hgt[k] = __float_as_uint(hi[k].w);
            }
        }
        // a node over exactly two triangles is emitted as a leaf: it adds no level
        const uint32_t h = (range_size[cur] == 2u) ? 0u : (hgt[0] > hgt[1] ? hgt[0] : hgt[1]) + 1u;
        nmin[cur] = make_float4(fminf(lo[0].x, lo[1].x), fminf(lo[0].y, lo[1].y), fminf(lo[0].z, lo[1].z),
                                fmaxf(lo[0].w, lo[1].w));
        nmax[cur] = make_float4(fmaxf(hi[0].x, hi[1].x), fmaxf(hi[0].y, hi[1].y), fmaxf(hi[0].z, hi[1].z),
                                __uint_as_float(h));
        if (cur == 0u) return;
        cur = parent[cur];
    }
}

__global__ void __launch_bounds__(256) k_lbvh_emit(const uint32_t* __restrict__ items, uint32_t n,
                                                    const uint32_t* __restrict__ slots,
                                                    const float4* __restrict__ pmin, const float4* __restrict__ pmax,
                                                    const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                    const uint32_t* __restrict__ range_first,
                                                    const uint32_t* __restrict__ range_size,
                                                    const float4* __restrict__ nmin, const float4* __restrict__ nmax,
                                                    const uint32_t* __restrict__ bounds, SphereNode* __restrict__ nodes,
                                                    uint32_t* __restrict__ fast_slots, DeviceTreeInfo* info) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) fast_slots[i] = slots[items[i]];
    if (i + 1u < n) {
        float4 o[4];
        uint32_t ref[2];
        const uint32_t ch[2] = {left[i], right[i]};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & kLeafTag) {
                const uint32_t pos = ch[k] & ~kLeafTag;
                const uint32_t item = items[pos];
                o[2 * k] = pmin[item];
                o[2 * k + 1] = pmax[item];
                ref[k] = kLeafTag | pos;
            } else {
                o[2 * k] = nmin[ch[k]];
                o[2 * k + 1] = nmax[ch[k]];
                ref[k] = (range_size[ch[k]] == 2u) ? (kLeafTag | (1u << 28) | range_first[ch[k]]) : ch[k];
            }
        }
        SphereNode nd;
        nd.lmin[0] = o[0].x; nd.lmin[1] = o[0].y; nd.lmin[2] = o[0].z; nd.left = ref[0];
        nd.lmax[0] = o[1].x; nd.lmax[1] = o[1].y; nd.lmax[2] = o[1].z; nd.right = ref[1];
        nd.rmin[0] = o[2].x; nd.rmin[1] = o[2].y; nd.rmin[2] = o[2].z; nd._pad0 = fa_bits(o[0].w);
        nd.rmax[0] = o[3].x; nd.rmax[1] = o[3].y; nd.rmax[2] = o[3].z; nd._pad1 = fa_bits(o[2].w);
        nodes[i] = nd;
    }
    if (i == 0u) {
        const float4 r0 = nmin[0], r1 = nmax[0];
        info->root = (range_size[0] == 2u) ? (kLeafTag | (1u << 28)) : 0u;
        info->depth = __float_as_uint(r1.w) + 2u;  // pending far children <= internal levels
        info->root_amax = 0.0f;
        for (int a = 0; a < 3; ++a) {
            info->bmin[a] = ord2f(bounds[a]);
            info->bmax[a] = ord2f(bounds[3 + a]);
        }
        info->margin = 0.0f;
        (void)r0;
    }
}


// ---- PLOC (Meister & Bittner, "Parallel Locally-Ordered Clustering for BVH Construction", 2018):
// bottom-up agglomeration over the Morton order.  Every cluster looks kPlocRadius positions to
// either side for the neighbour whose union with it has the smallest surface area; mutual nearest
// neighbours merge; the survivors are compacted and the search repeats.  About log_1.6(n)
// rounds; the tree quality is close to a SAH build's.
#ifndef RB_PLOC_RADIUS
#define RB_PLOC_RADIUS 16
#endif
constexpr int kPlocRadius = RB_PLOC_RADIUS;

struct PlocClusters {
    float4* lo;      // box min, largest F_k below
    float4* hi;      // box max, height (uint bits)
    uint32_t* ref;   // child reference a parent would store
    uint32_t* run;   // single triangle: its position in the Morton order; otherwise ~0
};

__global__ void __launch_bounds__(256) k_ploc_init(const uint32_t* __restrict__ items, const uint32_t* __restrict__ slots,
                                                    uint32_t n, const float4* __restrict__ pmin,
                                                    const float4* __restrict__ pmax, PlocClusters c,
                                                    uint32_t* __restrict__ fast_slots, uint32_t* counters) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0u) {
        counters[0] = n;   // live clusters
        counters[1] = 0u;  // nodes written
    }
    if (i >= n) return;
    const uint32_t item = items[i];
    const float4 a = pmin[item], b = pmax[item];
    c.lo[i] = a;
    c.hi[i] = make_float4(b.x, b.y, b.z, __uint_as_float(0u));
    c.ref[i] = kLeafTag | i;
    c.run[i] = i;
    fast_slots[i] = slots[item];
}

__global__ void __launch_bounds__(256) k_ploc_nn(const uint32_t* __restrict__ counters, PlocClusters c,
                                                  uint32_t* __restrict__ nn) {
    const uint32_t count = counters[0];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float4 a0 = c.lo[i], a1 = c.hi[i];
    const uint32_t jb = i > (uint32_t)kPlocRadius ? i - (uint32_t)kPlocRadius : 0u;
    const uint32_t je = (i + (uint32_t)kPlocRadius + 1u < count) ? i + (uint32_t)kPlocRadius + 1u : count;
    float best = __builtin_inff();
    uint32_t bj = (i + 1u < count) ? i + 1u : (i > 0u ? i - 1u : 0u);
    for (uint32_t j = jb; j < je; ++j) {
        if (j == i) continue;
        const float4 b0 = c.lo[j], b1 = c.hi[j];
        const float dx = fmaxf(a1.x, b1.x) - fminf(a0.x, b0.x), dy = fmaxf(a1.y, b1.y) - fminf(a0.y, b0.y),
                    dz = fmaxf(a1.z, b1.z) - fminf(a0.z, b0.z);
        const float area = dx * dy + dy * dz + dz * dx;
        // ties (coincident or equal boxes): the nearer position, and at equal distance the even
        // positions look right and the odd ones left, so that a run of identical clusters still
        // pairs up (0,1) (2,3) ... in one round instead of merging one pair per round
        const uint32_t dist = j > i ? j - i : i - j, bdist = bj > i ? bj - i : i - bj;
        const bool prefer = dist < bdist || (dist == bdist && ((i & 1u) ? j < i : j > i));
        if (area < best || (area == best && prefer)) {
            best = area;
            bj = j;
        }
    }
    nn[i] = bj;
}

__global__ void __launch_bounds__(256) k_ploc_merge(uint32_t* counters, uint32_t n, PlocClusters c,
                                                     const uint32_t* __restrict__ nn, uint32_t* __restrict__ keep,
                                                     SphereNode* __restrict__ nodes) {
    const uint32_t count = counters[0];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (i >= count) {
        keep[i] = 0u;
        return;
    }
    const uint32_t j = nn[i];
    const bool mutual = (count > 1u) && nn[j] == i;
    if (!mutual) {
        keep[i] = 1u;
        return;
    }
    if (i > j) {  // merged into the lower position by that thread
        keep[i] = 0u;
        return;
    }
    const float4 a0 = c.lo[i], a1 = c.hi[i], b0 = c.lo[j], b1 = c.hi[j];
    const uint32_t ra = c.ref[i], rb = c.ref[j];
    const uint32_t pa = c.run[i], pb = c.run[j];
    const uint32_t ha = __float_as_uint(a1.w), hb = __float_as_uint(b1.w);
    uint32_t ref, h;
    if (pa != 0xFFFFFFFFu && pb == pa + 1u) {
        ref = kLeafTag | (1u << 28) | pa;  // neighbours in the triangle order: one two-triangle leaf
        h = 0u;
    } else {
        const uint32_t id = atomicAdd(&counters[1], 1u);
        SphereNode nd;
        nd.lmin[0] = a0.x; nd.lmin[1] = a0.y; nd.lmin[2] = a0.z; nd.left = ra;
        nd.lmax[0] = a1.x; nd.lmax[1] = a1.y; nd.lmax[2] = a1.z; nd.right = rb;
        nd.rmin[0] = b0.x; nd.rmin[1] = b0.y; nd.rmin[2] = b0.z; nd._pad0 = fa_bits(a0.w);
        nd.rmax[0] = b1.x; nd.rmax[1] = b1.y; nd.rmax[2] = b1.z; nd._pad1 = fa_bits(b0.w);
        nodes[id] = nd;
        ref = id;
        h = (ha > hb ? ha : hb) + 1u;
    }
    c.lo[i] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), fmaxf(a0.w, b0.w));
    c.hi[i] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), __uint_as_float(h));
    c.ref[i] = ref;
    c.run[i] = 0xFFFFFFFFu;
    keep[i] = 1u;
}

__global__ void __launch_bounds__(256) k_ploc_scatter(uint32_t* counters, uint32_t n, PlocClusters from, PlocClusters to,
                                                       const uint32_t* __restrict__ keep,
                                                       const uint32_t* __restrict__ pos) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (keep[i]) {
        const uint32_t d = pos[i];
        to.lo[d] = from.lo[i];
        to.hi[d] = from.hi[i];
        to.ref[d] = from.ref[i];
        to.run[d] = from.run[i];
    }
    if (i == n - 1u) counters[2] = pos[i] + keep[i];  // becomes counters[0] in k_ploc_commit
}

__global__ void k_ploc_commit(uint32_t* counters) { counters[0] = counters[2]; }

__global__ void k_ploc_finish(PlocClusters c, const uint32_t* __restrict__ bounds, DeviceTreeInfo* info) {
    const float4 r0 = c.lo[0], r1 = c.hi[0];
    info->root = c.ref[0];
    info->depth = __float_as_uint(r1.w) + 2u;
    info->root_amax = 0.0f;
    for (int a = 0; a < 3; ++a) {
        info->bmin[a] = ord2f(bounds[a]);
        info->bmax[a] = ord2f(bounds[3 + a]);
    }
    info->margin = 0.0f;
    (void)r0;
}

inline size_t align256(size_t x) { return (x + 255u) & ~size_t(255); }

// ---- The sphere tree (BASELINE C4: 10^6 spheres; rb_device_shade.hpp SphereWalk, rb_kernels.hip k_trace_sph) built on
// the device: Morton order of the centres, leaves = kSphLeaf consecutive spheres of that order (one 16-byte {centre,
// radius} record each, so the lanes that test a leaf read consecutive bytes), and the LBVH of this file over the leaves'
// first keys.  The tree only steers the walk -- every candidate goes through the reference's intersect_sphere and ties
// resolve by the original index -- so the frame is the same bits as with the host's median-split tree (rb_bvh.cpp), which
// stays as the fallback (a tree deeper than the LDS stack) and as the checker (tests/test_gpu_sphere_tree.py).
__global__ void __launch_bounds__(256) k_sph_bounds(const rb_sphere* __restrict__ spheres, uint32_t n, uint32_t* bounds) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float inf = __builtin_inff();
    float cn[3] = {inf, inf, inf}, cx[3] = {-inf, -inf, -inf};
    if (i < n) {
        const float4 cr = reinterpret_cast<const float4*>(spheres)[(size_t)i * 6u];
        cn[0] = cx[0] = cr.x; cn[1] = cx[1] = cr.y; cn[2] = cx[2] = cr.z;
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            cn[a] = fminf(cn[a], __shfl_xor(cn[a], off, 64));
            cx[a] = fmaxf(cx[a], __shfl_xor(cx[a], off, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&bounds[6 + a], f2ord(cn[a]));
            atomicMax(&bounds[9 + a], f2ord(cx[a]));
        }
    }
}

__global__ void __launch_bounds__(256) k_sph_keys(const rb_sphere* __restrict__ spheres, uint32_t n, const uint32_t* __restrict__ bounds,
                                                   unsigned long long* __restrict__ keys, uint32_t* __restrict__ items) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 cr = reinterpret_cast<const float4*>(spheres)[(size_t)i * 6u];
    const float c[3] = {cr.x, cr.y, cr.z};
    // one scale for the three axes (the largest extent): the cells are cubes, so a flat scene (C4: 200 x 20 x 200) is cut
    // along its long axes first instead of into slabs a tenth as thick as they are wide
    float ext = 0.0f;
    for (int k = 0; k < 3; ++k) ext = fmaxf(ext, ord2f(bounds[9 + k]) - ord2f(bounds[6 + k]));
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = ord2f(bounds[6 + k]);
        float x = (ext > 0.0f) ? (c[k] - lo) / ext * 2097152.0f : 0.0f;
        x = fminf(fmaxf(x, 0.0f), 2097151.0f);  // NaN -> 0
        q[k] = static_cast<uint32_t>(x);
    }
    keys[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    items[i] = i;
}

// one thread per sphere of the sorted order: its leaf record and original index; one thread per leaf: the leaf's box
// (c -+ r in f32: what the walk's margin allows for, rb_device_shade.hpp kSphAbs) and first key
__global__ void __launch_bounds__(256) k_sph_leaves(const rb_sphere* __restrict__ spheres, const uint32_t* __restrict__ items,
                                                     const unsigned long long* __restrict__ keys, uint32_t n, uint32_t n_leaf,
                                                     float4* __restrict__ leaf_out, uint32_t* __restrict__ id_out,
                                                     float4* __restrict__ lmin, float4* __restrict__ lmax,
                                                     unsigned long long* __restrict__ lkeys, uint32_t* __restrict__ ident) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        const uint32_t id = items[i];
        leaf_out[i] = reinterpret_cast<const float4*>(spheres)[(size_t)id * 6u];
        id_out[i] = id;
    }
    if (i < n_leaf) {
        const uint32_t first = i * kSphLeaf, end = (first + kSphLeaf < n) ? first + kSphLeaf : n;
        const float inf = __builtin_inff();
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (uint32_t j = first; j < end; ++j) {
            const float4 cr = reinterpret_cast<const float4*>(spheres)[(size_t)items[j] * 6u];
            const float c[3] = {cr.x, cr.y, cr.z};
            for (int a = 0; a < 3; ++a) {
                mn[a] = fminf(mn[a], c[a] - cr.w);
                mx[a] = fmaxf(mx[a], c[a] + cr.w);
            }
        }
        lmin[i] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        lmax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
        lkeys[i] = keys[first];
        ident[i] = i;
    }
}

// bottom-up boxes and heights over the leaves' LBVH (as k_lbvh_refit, every internal node a real node)
__global__ void __launch_bounds__(256) k_sph_refit(uint32_t n_leaf, const float4* __restrict__ lmin, const float4* __restrict__ lmax,
                                                    const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
                                                    const uint32_t* __restrict__ parent, const uint32_t* __restrict__ leaf_parent,
                                                    uint32_t* flags, float4* nmin, float4* nmax) {
    const uint32_t pos = blockIdx.x * 256u + threadIdx.x;
    if (pos >= n_leaf) return;
    uint32_t cur = leaf_parent[pos];
    for (;;) {
        __threadfence();
        if (atomicAdd(&flags[cur], 1u) == 0u) return;  // the sibling subtree is not finished yet
        __threadfence();
        float4 lo[2], hi[2];
        uint32_t hgt[2];
        const uint32_t ch[2] = {left[cur], right[cur]};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & kLeafTag) {
                lo[k] = lmin[ch[k] & ~kLeafTag];
                hi[k] = lmax[ch[k] & ~kLeafTag];
                hgt[k] = 0u;
            } else {
                lo[k] = nmin[ch[k]];  // written by another CU: the fence above has invalidated L1
                hi[k] = nmax[ch[k]];
                hgt[k] = __float_as_uint(hi[k].w);
            }
        }
        const uint32_t h = (hgt[0] > hgt[1] ? hgt[0] : hgt[1]) + 1u;
        nmin[cur] = make_float4(fminf(lo[0].x, lo[1].x), fminf(lo[0].y, lo[1].y), fminf(lo[0].z, lo[1].z), 0.0f);
        nmax[cur] = make_float4(fmaxf(hi[0].x, hi[1].x), fmaxf(hi[0].y, hi[1].y), fmaxf(hi[0].z, hi[1].z), __uint_as_float(h));
        if (cur == 0u) return;
        cur = parent[cur];
    }
}

__device__ __forceinline__ uint32_t sph_leaf_ref(uint32_t leaf, uint32_t n) {
    const uint32_t first = leaf * kSphLeaf, cnt = (n - first < kSphLeaf) ? n - first : kSphLeaf;
    return kLeafTag | ((cnt - 1u) << 27) | first;
}

__global__ void __launch_bounds__(256) k_sph_emit(uint32_t n, uint32_t n_leaf, const float4* __restrict__ lmin,
                                                   const float4* __restrict__ lmax, const uint32_t* __restrict__ left,
                                                   const uint32_t* __restrict__ right, const float4* __restrict__ nmin,
                                                   const float4* __restrict__ nmax, SphereNode* __restrict__ nodes,
                                                   DeviceSphereTreeInfo* info) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1u < n_leaf) {
        float4 o[4];
        uint32_t ref[2];
        const uint32_t ch[2] = {left[i], right[i]};
        for (int k = 0; k < 2; ++k) {
            if (ch[k] & kLeafTag) {
                const uint32_t leaf = ch[k] & ~kLeafTag;
                o[2 * k] = lmin[leaf];
                o[2 * k + 1] = lmax[leaf];
                ref[k] = sph_leaf_ref(leaf, n);
            } else {
                o[2 * k] = nmin[ch[k]];
                o[2 * k + 1] = nmax[ch[k]];
                ref[k] = ch[k];
            }
        }
        SphereNode nd;
        nd.lmin[0] = o[0].x; nd.lmin[1] = o[0].y; nd.lmin[2] = o[0].z; nd.left = ref[0];
        nd.lmax[0] = o[1].x; nd.lmax[1] = o[1].y; nd.lmax[2] = o[1].z; nd.right = ref[1];
        nd.rmin[0] = o[2].x; nd.rmin[1] = o[2].y; nd.rmin[2] = o[2].z; nd._pad0 = 0u;
        nd.rmax[0] = o[3].x; nd.rmax[1] = o[3].y; nd.rmax[2] = o[3].z; nd._pad1 = 0u;
        nodes[i] = nd;
    }
    if (i == 0u) {
        info->root = n_leaf > 1u ? 0u : sph_leaf_ref(0u, n);
        info->depth = n_leaf > 1u ? __float_as_uint(nmax[0].w) + 1u : 1u;   // entries the walk's stack can hold at once: <= internal levels
    }
}

}  // namespace

int device_sphere_bvh_build(const rb_sphere* spheres, uint32_t n, SphereNode* nodes_out, float* leaf_out, uint32_t* id_out,
                            DeviceSphereTreeInfo* info_out, void* stream_) {
    if (n == 0u || n >= (1u << 27)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using key_t = unsigned long long;
    const uint32_t n_leaf = (n + kSphLeaf - 1u) / kSphLeaf;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<key_t*>(nullptr), static_cast<key_t*>(nullptr),
                                             static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 63, stream);
    if (e != hipSuccess) return static_cast<int>(e);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    const size_t o_keys_in = carve(sizeof(key_t) * n), o_keys = carve(sizeof(key_t) * n);
    const size_t o_items_in = carve(4u * n), o_items = carve(4u * n);
    const size_t o_lmin = carve(16u * n_leaf), o_lmax = carve(16u * n_leaf), o_nmin = carve(16u * n_leaf), o_nmax = carve(16u * n_leaf);
    const size_t o_lkeys = carve(sizeof(key_t) * n_leaf), o_ident = carve(4u * n_leaf);
    const size_t o_left = carve(4u * n_leaf), o_right = carve(4u * n_leaf), o_first = carve(4u * n_leaf), o_size = carve(4u * n_leaf);
    const size_t o_parent = carve(4u * n_leaf), o_leafpar = carve(4u * n_leaf), o_flags = carve(4u * n_leaf);
    const size_t o_bounds = carve(64), o_info = carve(sizeof(DeviceSphereTreeInfo)), o_sort = carve(sort_bytes);
    char* base = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&base), off);
    if (e != hipSuccess) return static_cast<int>(e);
    auto at = [&](size_t o) { return base + o; };
    auto done = [&](hipError_t err) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        return static_cast<int>(err);
    };
    key_t* keys_in = reinterpret_cast<key_t*>(at(o_keys_in));
    key_t* keys = reinterpret_cast<key_t*>(at(o_keys));
    uint32_t* items_in = reinterpret_cast<uint32_t*>(at(o_items_in));
    uint32_t* items = reinterpret_cast<uint32_t*>(at(o_items));
    float4 *lmin = reinterpret_cast<float4*>(at(o_lmin)), *lmax = reinterpret_cast<float4*>(at(o_lmax));
    float4 *nmin = reinterpret_cast<float4*>(at(o_nmin)), *nmax = reinterpret_cast<float4*>(at(o_nmax));
    key_t* lkeys = reinterpret_cast<key_t*>(at(o_lkeys));
    uint32_t* ident = reinterpret_cast<uint32_t*>(at(o_ident));
    uint32_t *left = reinterpret_cast<uint32_t*>(at(o_left)), *right = reinterpret_cast<uint32_t*>(at(o_right));
    uint32_t *rfirst = reinterpret_cast<uint32_t*>(at(o_first)), *rsize = reinterpret_cast<uint32_t*>(at(o_size));
    uint32_t *parent = reinterpret_cast<uint32_t*>(at(o_parent)), *leafpar = reinterpret_cast<uint32_t*>(at(o_leafpar));
    uint32_t* flags = reinterpret_cast<uint32_t*>(at(o_flags));
    uint32_t* bounds = reinterpret_cast<uint32_t*>(at(o_bounds));
    DeviceSphereTreeInfo* d_info = reinterpret_cast<DeviceSphereTreeInfo*>(at(o_info));
    const uint32_t init[16] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u,
                               0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const dim3 grid((n + 255u) / 256u), lgrid((n_leaf + 255u) / 256u), block(256);
    e = hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return done(e);
    e = hipMemsetAsync(flags, 0, 4u * n_leaf, stream);
    if (e != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_sph_bounds, grid, block, 0, stream, spheres, n, bounds);
    hipLaunchKernelGGL(k_sph_keys, grid, block, 0, stream, spheres, n, bounds, keys_in, items_in);
    e = rocprim::radix_sort_pairs(at(o_sort), sort_bytes, keys_in, keys, items_in, items, n, 0, 63, stream);
    if (e != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_sph_leaves, grid, block, 0, stream, spheres, items, keys, n, n_leaf, reinterpret_cast<float4*>(leaf_out),
                       id_out, lmin, lmax, lkeys, ident);
    if (n_leaf > 1u) {
        hipLaunchKernelGGL(k_lbvh_hierarchy, lgrid, block, 0, stream, lkeys, n_leaf, left, right, rfirst, rsize, parent, leafpar);
        hipLaunchKernelGGL(k_sph_refit, lgrid, block, 0, stream, n_leaf, lmin, lmax, left, right, parent, leafpar, flags, nmin, nmax);
    }
    hipLaunchKernelGGL(k_sph_emit, lgrid, block, 0, stream, n, n_leaf, lmin, lmax, left, right, nmin, nmax, nodes_out, d_info);
    e = hipGetLastError();
    if (e != hipSuccess) return done(e);
    e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(info_out, d_info, sizeof(DeviceSphereTreeInfo), hipMemcpyDeviceToHost);
    return done(e);
}

// All work is queued on `stream`; `info_out` (host) is valid when this returns (it synchronises).
int device_fast_bvh_build(const rb_gpu_triangle* tris, const uint32_t* indices, const uint32_t* slots, uint32_t n,
                          const uint32_t* slot_meta, SphereNode* nodes_out, uint32_t* fast_slots_out, DeviceTreeInfo* info_out,
                          void* stream_, bool plain_lbvh) {
    if (n < 2u || n >= (1u << 28)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using key_t = unsigned long long;
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<key_t*>(nullptr), static_cast<key_t*>(nullptr),
                                             static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 63,
                                             stream);
    if (e != hipSuccess) return static_cast<int>(e);
    // one allocation, carved up
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    const size_t o_keys_in = carve(sizeof(key_t) * n), o_keys = carve(sizeof(key_t) * n);
    const size_t o_items_in = carve(4u * n), o_items = carve(4u * n);
    const size_t o_pmin = carve(16u * n), o_pmax = carve(16u * n), o_nmin = carve(16u * n), o_nmax = carve(16u * n);
    const size_t o_left = carve(4u * n), o_right = carve(4u * n), o_first = carve(4u * n), o_size = carve(4u * n);
    const size_t o_parent = carve(4u * n), o_leafpar = carve(4u * n), o_flags = carve(4u * n);
    const size_t o_bounds = carve(64), o_info = carve(sizeof(DeviceTreeInfo)), o_sort = carve(sort_bytes);
    // PLOC: a second set of cluster arrays, neighbour / keep / position words, a scan workspace
    size_t scan_bytes = 0;
    if (!plain_lbvh) {
        e = rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u,
                                    n, rocprim::plus<uint32_t>(), stream);
        if (e != hipSuccess) return static_cast<int>(e);
    }
    const size_t o_blo = carve(plain_lbvh ? 0 : 16u * n), o_bhi = carve(plain_lbvh ? 0 : 16u * n);
    const size_t o_aref = carve(plain_lbvh ? 0 : 4u * n), o_bref = carve(plain_lbvh ? 0 : 4u * n);
    const size_t o_arun = carve(plain_lbvh ? 0 : 4u * n), o_brun = carve(plain_lbvh ? 0 : 4u * n);
    const size_t o_scan = carve(scan_bytes), o_counters = carve(64);
    char* base = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&base), off);
    if (e != hipSuccess) return static_cast<int>(e);
    auto at = [&](size_t o) { return base + o; };
    key_t* keys_in = reinterpret_cast<key_t*>(at(o_keys_in));
    key_t* keys = reinterpret_cast<key_t*>(at(o_keys));
    uint32_t* items_in = reinterpret_cast<uint32_t*>(at(o_items_in));
    uint32_t* items = reinterpret_cast<uint32_t*>(at(o_items));
    float4* pmin = reinterpret_cast<float4*>(at(o_pmin));
    float4* pmax = reinterpret_cast<float4*>(at(o_pmax));
    float4* nmin = reinterpret_cast<float4*>(at(o_nmin));
    float4* nmax = reinterpret_cast<float4*>(at(o_nmax));
    uint32_t* left = reinterpret_cast<uint32_t*>(at(o_left));
    uint32_t* right = reinterpret_cast<uint32_t*>(at(o_right));
    uint32_t* rfirst = reinterpret_cast<uint32_t*>(at(o_first));
    uint32_t* rsize = reinterpret_cast<uint32_t*>(at(o_size));
    uint32_t* parent = reinterpret_cast<uint32_t*>(at(o_parent));
    uint32_t* leafpar = reinterpret_cast<uint32_t*>(at(o_leafpar));
    uint32_t* flags = reinterpret_cast<uint32_t*>(at(o_flags));
    uint32_t* bounds = reinterpret_cast<uint32_t*>(at(o_bounds));
    DeviceTreeInfo* d_info = reinterpret_cast<DeviceTreeInfo*>(at(o_info));

    const uint32_t init[16] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u,
                               0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const dim3 grid((n + 255u) / 256u), block(256);
    auto done = [&](hipError_t err) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        return static_cast<int>(err);
    };
    e = hipMemcpyAsync(bounds, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return done(e);
    e = hipMemsetAsync(flags, 0, 4u * n, stream);
    if (e != hipSuccess) return done(e);
    hipLaunchKernelGGL(k_lbvh_prims, grid, block, 0, stream, tris, indices, slots, slot_meta, n, pmin, pmax, bounds);
    hipLaunchKernelGGL(k_lbvh_keys, grid, block, 0, stream, pmin, pmax, n, bounds, keys_in, items_in);
    e = rocprim::radix_sort_pairs(at(o_sort), sort_bytes, keys_in, keys, items_in, items, n, 0, 63, stream);
    if (e != hipSuccess) return done(e);
    if (!plain_lbvh) {
        // nn / keep / pos reuse the LBVH link arrays
        uint32_t *nn = left, *keep = right, *pos = rfirst;
        uint32_t* counters = reinterpret_cast<uint32_t*>(at(o_counters));
        PlocClusters A{nmin, nmax, reinterpret_cast<uint32_t*>(at(o_aref)), reinterpret_cast<uint32_t*>(at(o_arun))};
        PlocClusters B{reinterpret_cast<float4*>(at(o_blo)), reinterpret_cast<float4*>(at(o_bhi)),
                       reinterpret_cast<uint32_t*>(at(o_bref)), reinterpret_cast<uint32_t*>(at(o_brun))};
        hipLaunchKernelGGL(k_ploc_init, grid, block, 0, stream, items, slots, n, pmin, pmax, A, fast_slots_out, counters);
        uint32_t live = n;
        for (int round = 0; live > 1u && round < 4096; ++round) {
            const dim3 g((live + 255u) / 256u);  // `live` is an upper bound of the device-side count
            hipLaunchKernelGGL(k_ploc_nn, g, block, 0, stream, counters, A, nn);
            hipLaunchKernelGGL(k_ploc_merge, g, block, 0, stream, counters, live, A, nn, keep, nodes_out);
            e = rocprim::exclusive_scan(at(o_scan), scan_bytes, keep, pos, 0u, live, rocprim::plus<uint32_t>(), stream);
            if (e != hipSuccess) return done(e);
            hipLaunchKernelGGL(k_ploc_scatter, g, block, 0, stream, counters, live, A, B, keep, pos);
            hipLaunchKernelGGL(k_ploc_commit, dim3(1), dim3(1), 0, stream, counters);
            std::swap(A, B);
            if ((round & 3) == 3 || live <= 4096u) {  // tighten the bound now and then (one 4-byte read-back)
                e = hipStreamSynchronize(stream);
                if (e == hipSuccess) e = hipMemcpy(&live, counters, 4, hipMemcpyDeviceToHost);
                if (e != hipSuccess) return done(e);
            }
        }
        if (live > 1u) {  // not reached in practice (every round merges at least the closest pair)
            e = hipStreamSynchronize(stream);
            if (e == hipSuccess) e = hipMemcpy(&live, counters, 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return done(e);
            if (live > 1u) return done(hipErrorNotReady);  // the caller falls back to the host builder
        }
        hipLaunchKernelGGL(k_ploc_finish, dim3(1), dim3(1), 0, stream, A, bounds, d_info);
        e = hipGetLastError();
        if (e != hipSuccess) return done(e);
        e = hipStreamSynchronize(stream);
        if (e == hipSuccess) e = hipMemcpy(info_out, d_info, sizeof(DeviceTreeInfo), hipMemcpyDeviceToHost);
        return done(e);
    }
    hipLaunchKernelGGL(k_lbvh_hierarchy, grid, block, 0, stream, keys, n, left, right, rfirst, rsize, parent, leafpar);
    hipLaunchKernelGGL(k_lbvh_refit, grid, block, 0, stream, items, n, pmin, pmax, left, right, rsize, parent, leafpar,
                       flags, nmin, nmax);
    hipLaunchKernelGGL(k_lbvh_emit, grid, block, 0, stream, items, n, slots, pmin, pmax, left, right, rfirst, rsize,
                       nmin, nmax, bounds, nodes_out, fast_slots_out, d_info);
    e = hipGetLastError();
    if (e != hipSuccess) return done(e);
    e = hipStreamSynchronize(stream);
    if (e == hipSuccess) e = hipMemcpy(info_out, d_info, sizeof(DeviceTreeInfo), hipMemcpyDeviceToHost);
    return done(e);
}

}  // namespace rb
