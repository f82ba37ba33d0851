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
#include "rb_chunk_math.hpp"

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

// ---- The sphere tree (BASELINE C4: 10^6 spheres; rb_internal.hpp SphereNode4, rb_device_shade.hpp sphere_node_step,
// rb_kernels.hip k_trace_sph) built on the device: the host builder's median splits (rb_bvh.cpp), level by level.  The
// spheres of level l's segment s are positions [s n / 2^l, (s + 1) n / 2^l) of the current order -- a complete tree, nothing
// to store --; one pass per level finds every segment's longest axis (bounds of the centres), keys every sphere with
// (segment, centre along that axis) and sorts: the halves of every segment are then the next level's segments.  Splitting
// stops at the first level whose segments hold at most kSphLeaf spheres; two levels of splits make one 4-wide node.
// (r04's first device builder -- Morton order, leaves of 16 consecutive spheres, an LBVH above them -- built in 7 ms and made
// the walk test 2.4 times as many spheres as the median splits: 123 against 51 per segment on C4.)
// The tree only steers the walk -- every candidate goes through the reference's intersect_sphere and ties resolve by the
// original index -- so the frame is the same bits as with the host's tree, which stays as the fallback and the checker.
__device__ __forceinline__ uint32_t ms_start(uint32_t s, uint32_t n, uint32_t l) { return (uint32_t)(((unsigned long long)s * n) >> l); }
__device__ __forceinline__ uint32_t ms_segment(uint32_t i, uint32_t n, uint32_t l) {   // the s with ms_start(s) <= i < ms_start(s + 1)
    return (uint32_t)(((((unsigned long long)i + 1ull) << l) - 1ull) / n);
}

__global__ void __launch_bounds__(256) k_ms_init(const rb_sphere* __restrict__ spheres, uint32_t n, float4* __restrict__ cen,
                                                  uint32_t* __restrict__ items) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    cen[i] = reinterpret_cast<const float4*>(spheres)[(size_t)i * 6u];   // {centre, radius}
    items[i] = i;
}

// bounds of the centres per segment (ordered uints: [s][0..2] min, [s][3..5] max).  Large segments: one thread per
// sphere, a wavefront that lies inside one segment reduces first; small ones (SMALL): one thread per segment.
template <bool SMALL>
__global__ void __launch_bounds__(256) k_ms_bounds(const float4* __restrict__ cen, const uint32_t* __restrict__ items, uint32_t n,
                                                    uint32_t l, uint32_t* __restrict__ segb) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const float inf = __builtin_inff();
    if constexpr (SMALL) {
        if (i >= (1u << l)) return;
        const uint32_t first = ms_start(i, n, l), end = ms_start(i + 1u, n, l);
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (uint32_t j = first; j < end; ++j) {
            const float4 c = cen[items[j]];
            mn[0] = fminf(mn[0], c.x); mn[1] = fminf(mn[1], c.y); mn[2] = fminf(mn[2], c.z);
            mx[0] = fmaxf(mx[0], c.x); mx[1] = fmaxf(mx[1], c.y); mx[2] = fmaxf(mx[2], c.z);
        }
        for (int a = 0; a < 3; ++a) {
            segb[i * 6u + a] = f2ord(mn[a]);
            segb[i * 6u + 3u + a] = f2ord(mx[a]);
        }
    } else {
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        uint32_t s = 0xFFFFFFFFu;
        if (i < n) {
            const float4 c = cen[items[i]];
            mn[0] = mx[0] = c.x; mn[1] = mx[1] = c.y; mn[2] = mx[2] = c.z;
            s = ms_segment(i, n, l);
        }
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s);
        if (__ballot(s != s0) == 0ull) {   // the whole wavefront in one segment (or past the end)
            if (s0 == 0xFFFFFFFFu) return;
            for (int a = 0; a < 3; ++a)
                for (int off = 32; off > 0; off >>= 1) {
                    mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64));
                    mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64));
                }
            if ((threadIdx.x & 63u) == 0u)
                for (int a = 0; a < 3; ++a) {
                    atomicMin(&segb[s0 * 6u + a], f2ord(mn[a]));
                    atomicMax(&segb[s0 * 6u + 3u + a], f2ord(mx[a]));
                }
        } else if (i < n) {
            for (int a = 0; a < 3; ++a) {
                atomicMin(&segb[s * 6u + a], f2ord(mn[a]));
                atomicMax(&segb[s * 6u + 3u + a], f2ord(mx[a]));
            }
        }
    }
}

// key = segment << 32 | centre along the segment's longest axis (the host builder's rule: x if it is the strictly longest,
// else y if longer than z, else z); the axis of every segment is kept for the nodes' visiting order
__global__ void __launch_bounds__(256) k_ms_keys(const float4* __restrict__ cen, const uint32_t* __restrict__ items, uint32_t n, uint32_t l,
                                                  const uint32_t* __restrict__ segb, unsigned long long* __restrict__ keys,
                                                  unsigned char* __restrict__ axes_l) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = ms_segment(i, n, l);
    const float ex = ord2f(segb[s * 6u + 3u]) - ord2f(segb[s * 6u]), ey = ord2f(segb[s * 6u + 4u]) - ord2f(segb[s * 6u + 1u]),
                ez = ord2f(segb[s * 6u + 5u]) - ord2f(segb[s * 6u + 2u]);
    const uint32_t axis = (ex > ey && ex > ez) ? 0u : ((ey > ez) ? 1u : 2u);
    const float4 c = cen[items[i]];
    const float v = axis == 0u ? c.x : axis == 1u ? c.y : c.z;
    keys[i] = ((unsigned long long)s << 32) | f2ord(v);
    if (i == ms_start(s, n, l)) axes_l[s] = (unsigned char)axis;
}

// the final order: leaf records, original indices, and the box of every leaf (heap index 2^L - 1 + leaf; c -+ r in f32: what
// the walk's margin allows for, rb_device_shade.hpp kSphAbs)
__global__ void __launch_bounds__(256) k_ms_leaves(const float4* __restrict__ cen, const uint32_t* __restrict__ items, uint32_t n, uint32_t L,
                                                    float4* __restrict__ leaf_out, uint32_t* __restrict__ id_out, float4* __restrict__ hmin,
                                                    float4* __restrict__ hmax) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) {
        const uint32_t id = items[i];
        leaf_out[i] = cen[id];
        id_out[i] = id;
    }
    if (i < (1u << L)) {
        const uint32_t first = ms_start(i, n, L), end = ms_start(i + 1u, n, L);
        const float inf = __builtin_inff();
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (uint32_t j = first; j < end; ++j) {
            const float4 c = cen[items[j]];
            mn[0] = fminf(mn[0], c.x - c.w); mn[1] = fminf(mn[1], c.y - c.w); mn[2] = fminf(mn[2], c.z - c.w);
            mx[0] = fmaxf(mx[0], c.x + c.w); mx[1] = fmaxf(mx[1], c.y + c.w); mx[2] = fmaxf(mx[2], c.z + c.w);
        }
        const uint32_t h = (1u << L) - 1u + i;
        hmin[h] = make_float4(mn[0], mn[1], mn[2], 0.0f);
        hmax[h] = make_float4(mx[0], mx[1], mx[2], 0.0f);
    }
}

__global__ void __launch_bounds__(256) k_ms_up(uint32_t l, float4* __restrict__ hmin, float4* __restrict__ hmax) {   // level l from level l + 1
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= (1u << l)) return;
    const uint32_t h = (1u << l) - 1u + s, c = (2u << l) - 1u + 2u * s;
    const float4 a0 = hmin[c], a1 = hmax[c], b0 = hmin[c + 1u], b1 = hmax[c + 1u];
    hmin[h] = make_float4(fminf(a0.x, b0.x), fminf(a0.y, b0.y), fminf(a0.z, b0.z), 0.0f);
    hmax[h] = make_float4(fmaxf(a1.x, b1.x), fmaxf(a1.y, b1.y), fmaxf(a1.z, b1.z), 0.0f);
}

// node index of segment s of the even level l: the even levels back to back (1 + 4 + 16 + ... nodes before level l)
__device__ __host__ __forceinline__ uint32_t ms_node_index(uint32_t l, uint32_t s) { return (uint32_t)((((1ull << l) - 1ull) / 3ull) + s); }

__global__ void __launch_bounds__(256) k_ms_emit(uint32_t n, uint32_t L, uint32_t l, const float4* __restrict__ hmin, const float4* __restrict__ hmax,
                                                  const unsigned char* __restrict__ axes, SphereNode4* __restrict__ nodes) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= (1u << l)) return;
    SphereNode4 nd{};
    for (int k = 0; k < 4; ++k) nd.ref[k] = kSphNone;
    auto child = [&](int k, uint32_t cl, uint32_t cs) {   // slot k = segment cs of level cl
        const uint32_t h = (1u << cl) - 1u + cs;
        const float4 lo = hmin[h], hi = hmax[h];
        const float l3[3] = {lo.x, lo.y, lo.z}, h3[3] = {hi.x, hi.y, hi.z};
        nd.set_box(k, l3, h3);
        if (cl == L) {
            const uint32_t first = ms_start(cs, n, L), cnt = ms_start(cs + 1u, n, L) - first;
            nd.ref[k] = kLeafTag | ((cnt - 1u) << 27) | first;
        } else {
            nd.ref[k] = ms_node_index(cl, cs);
        }
    };
    // axes[] holds level 0's segment, then level 1's two, ...: level l starts at 2^l - 1
    nd.axes = axes[(1u << l) - 1u + s];
    if (l + 1u == L) {   // an odd number of levels: the last nodes have two children, the leaves
        child(0, l + 1u, 2u * s);
        child(2, l + 1u, 2u * s + 1u);
    } else {
        nd.axes |= ((uint32_t)axes[(2u << l) - 1u + 2u * s] << 2) | ((uint32_t)axes[(2u << l) - 1u + 2u * s + 1u] << 4);
        for (uint32_t k = 0; k < 4u; ++k) child((int)k, l + 2u, 4u * s + k);
    }
    nodes[ms_node_index(l, s)] = nd;
}

}  // namespace

// levels of median splits for n spheres: the first level whose segments hold at most kSphLeaf
static uint32_t sphere_tree_levels(size_t n) {
    uint32_t L = 0;
    while (((n + (size_t(1) << L) - 1) >> L) > kSphLeaf) ++L;
    return L;
}
size_t sphere_tree_node_capacity(size_t n) {
    const uint32_t L = sphere_tree_levels(n);
    size_t nodes = 0;
    for (uint32_t l = 0; l < L; l += 2) nodes += size_t(1) << l;
    return nodes ? nodes : 1;
}

int device_sphere_bvh_build(const rb_sphere* spheres, uint32_t n, SphereNode4* nodes_out, float* leaf_out, uint32_t* id_out,
                            DeviceSphereTreeInfo* info_out, void* stream_) {
    if (n == 0u || n >= (1u << 27)) return static_cast<int>(hipErrorInvalidValue);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    using key_t = unsigned long long;
    const uint32_t L = sphere_tree_levels(n);   // leaves = the 2^L segments of level L, every one non-empty (n > kSphLeaf 2^(L-1) >= 2^L)
    if (L == 0u) {   // one leaf (not reached through the runtime: it scans up to 64 spheres)
        info_out->root = kLeafTag | ((n - 1u) << 27);
        info_out->depth = 0;
        info_out->n_nodes = 0;
    }
    size_t sort_bytes = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, sort_bytes, static_cast<key_t*>(nullptr), static_cast<key_t*>(nullptr),
                                             static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), n, 0, 64, stream);
    if (e != hipSuccess) return static_cast<int>(e);
    const size_t heap = size_t(2) << L;   // segments of levels 0 .. L: 2^(L+1) - 1
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    const size_t o_keys_a = carve(sizeof(key_t) * n), o_keys_b = carve(sizeof(key_t) * n);
    const size_t o_items_a = carve(4u * n), o_items_b = carve(4u * n), o_cen = carve(16u * size_t(n));
    const size_t o_segb = carve(24u * (size_t(1) << (L ? L - 1u : 0u))), o_axes = carve(heap);
    const size_t o_hmin = carve(16u * heap), o_hmax = carve(16u * heap), o_sort = carve(sort_bytes);
    char* base = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&base), off);
    if (e != hipSuccess) return static_cast<int>(e);
    auto at = [&](size_t o) { return base + o; };
    auto done = [&](hipError_t err) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        return static_cast<int>(err);
    };
    key_t *keys_a = reinterpret_cast<key_t*>(at(o_keys_a)), *keys_b = reinterpret_cast<key_t*>(at(o_keys_b));
    uint32_t *items = reinterpret_cast<uint32_t*>(at(o_items_a)), *items_b = reinterpret_cast<uint32_t*>(at(o_items_b));
    float4* cen = reinterpret_cast<float4*>(at(o_cen));
    uint32_t* segb = reinterpret_cast<uint32_t*>(at(o_segb));
    unsigned char* axes = reinterpret_cast<unsigned char*>(at(o_axes));
    float4 *hmin = reinterpret_cast<float4*>(at(o_hmin)), *hmax = reinterpret_cast<float4*>(at(o_hmax));
    const dim3 grid((n + 255u) / 256u), block(256);
    hipLaunchKernelGGL(k_ms_init, grid, block, 0, stream, spheres, n, cen, items);
    for (uint32_t l = 0; l < L; ++l) {
        const uint32_t segs = 1u << l;
        if (n / segs >= 256u) {   // (ordered uints: min starts at all ones, max at zero)
            e = hipMemsetAsync(segb, 0, 24u * size_t(segs), stream);
            if (e != hipSuccess) return done(e);
            e = hipMemset2DAsync(segb, 24, 0xFF, 12, segs, stream);
            if (e != hipSuccess) return done(e);
            hipLaunchKernelGGL(k_ms_bounds<false>, grid, block, 0, stream, cen, items, n, l, segb);
        } else {
            hipLaunchKernelGGL(k_ms_bounds<true>, dim3((segs + 255u) / 256u), block, 0, stream, cen, items, n, l, segb);
        }
        hipLaunchKernelGGL(k_ms_keys, grid, block, 0, stream, cen, items, n, l, segb, keys_a, axes + (segs - 1u));
        e = rocprim::radix_sort_pairs(at(o_sort), sort_bytes, keys_a, keys_b, items, items_b, n, 0, 32u + l, stream);
        if (e != hipSuccess) return done(e);
        std::swap(items, items_b);
    }
    hipLaunchKernelGGL(k_ms_leaves, grid, block, 0, stream, cen, items, n, L, reinterpret_cast<float4*>(leaf_out), id_out, hmin, hmax);
    for (uint32_t l = L; l-- > 0;) hipLaunchKernelGGL(k_ms_up, dim3(((1u << l) + 255u) / 256u), block, 0, stream, l, hmin, hmax);
    uint32_t levels4 = 0;
    for (uint32_t l = 0; l < L; l += 2, ++levels4)
        hipLaunchKernelGGL(k_ms_emit, dim3(((1u << l) + 255u) / 256u), block, 0, stream, n, L, l, hmin, hmax, axes, nodes_out);
    e = hipGetLastError();
    if (L != 0u) {
        info_out->root = 0u;
        info_out->depth = levels4;
        info_out->n_nodes = static_cast<uint32_t>(sphere_tree_node_capacity(n));
    }
    return done(e);
}

namespace {
// ---------------------------------------------------------------------------------------------------------------------
// The chunked walk's tree on the device.  What rb_bvh.cpp chunk_tree_build does per reference leaf
// -- triangles whose determinant-floor bound is far above the leaf's median split off, median splits of the rest down to
// chunks of kChunkTris, per-slot boxes, margins and cones -- one thread block does for one leaf here, the leaf's triangles
// in LDS: a reference leaf holds at most 128 triangles (bvh.rs:12), a caller's own tree with leaves beyond
// kChunkDeviceLeafMax goes to the host builder.  The arithmetic is rb_chunk_math.hpp's, the same source the host builder
// compiles; the two trees differ only in how equal sort keys fall.  Two launches: <true> counts every leaf's valid
// slots and nodes (scanned into each leaf's first rank / first node), <false> builds in place.  The caller's own internal
// nodes (reference boxes, kChunkExact) are made by the host's bottom-up pass from the per-leaf summaries read back
// (chunk_top_pass): 8 191 nodes for C5's 10^6 triangles.
constexpr uint32_t kChunkDevBlock = 128;
constexpr uint32_t kChunkDevTMax = 2u * (kChunkDeviceLeafMax / kChunkTris + 10u);   // tree entries of one leaf: chunks <= count / 16 + one per split-off set (<= 9)
constexpr uint32_t kChunkDevSets = 12;

struct ChunkDevT {   // one entry of the leaf's tree while it is made: items [lo, hi) of the current order
    uint16_t lo, hi, k;      // k = chunks below (1 = a chunk)
    int16_t left, right;     // entries (-1: not split yet / a chunk)
    uint16_t level, ord;     // depth below the leaf's root; ordinal among the inner entries (= node index - the leaf's first)
    uint16_t inner;
};

// ascending bitonic sort of (key, val) pairs in LDS; n2 a power of two; every thread of the block calls it
__device__ void chunk_bitonic(unsigned long long* key, uint32_t* val, uint32_t n2, uint32_t t) {
    __syncthreads();
    for (uint32_t k = 2; k <= n2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < n2; i += kChunkDevBlock) {
                const uint32_t x = i ^ j;
                if (x > i) {
                    const unsigned long long a = key[i], b = key[x];
                    if ((a > b) == ((i & k) == 0u)) {
                        key[i] = b;
                        key[x] = a;
                        const uint32_t v = val[i];
                        val[i] = val[x];
                        val[x] = v;
                    }
                }
            }
            __syncthreads();
        }
}

template <bool COUNT>
__global__ void __launch_bounds__(kChunkDevBlock) k_chunk_leaves(const rb_gpu_triangle* __restrict__ tris, uint32_t tri_count,
                                                                 const uint32_t* __restrict__ indices, uint32_t index_len,
                                                                 const uint2* __restrict__ leaf_desc, uint32_t n_leaves,
                                                                 uint32_t* __restrict__ leaf_valid, uint32_t* __restrict__ leaf_nodes,
                                                                 const uint32_t* __restrict__ rank0, const uint32_t* __restrict__ node0,
                                                                 ChunkNode* __restrict__ nodes, uint32_t* __restrict__ pos_slot,
                                                                 uint32_t* __restrict__ pos_rank, uint32_t* __restrict__ rank_slot,
                                                                 chunkmath::ChunkInfo* __restrict__ leaf_info) {
    using chunkmath::ChunkInfo;
    using chunkmath::ChunkItem;
    __shared__ ChunkItem items[kChunkDeviceLeafMax];
    __shared__ unsigned long long key[kChunkDeviceLeafMax];
    __shared__ uint32_t perm[kChunkDeviceLeafMax];
    __shared__ uint32_t slots[kChunkDeviceLeafMax];
    __shared__ uint8_t segof[kChunkDeviceLeafMax];
    __shared__ ChunkDevT tn[kChunkDevTMax];
    __shared__ alignas(8) unsigned char info_raw[sizeof(ChunkInfo) * kChunkDevTMax];   // (ChunkInfo has default member initialisers)
    ChunkInfo* info = reinterpret_cast<ChunkInfo*>(info_raw);
    __shared__ uint16_t pend[kChunkDevTMax], pend_next[kChunkDevTMax];
    __shared__ uint8_t pend_axis[kChunkDevTMax];
    __shared__ uint16_t set_lo[kChunkDevSets], set_hi[kChunkDevSets];
    __shared__ uint32_t s_count, s_wave[2], s_nsets, s_nt, s_npend, s_maxlevel, s_root;
    const uint32_t t = threadIdx.x, li = blockIdx.x;
    if (li >= n_leaves) return;
    const uint32_t first = leaf_desc[li].x, pc = leaf_desc[li].y;   // pc <= kChunkDeviceLeafMax (the host checked)
    // ---- the leaf's valid slots, in slot order (guards shader.wgsl:331, :336): rank = first rank + position here
    if (t == 0) s_count = 0;
    __syncthreads();
    for (uint32_t base = 0; base < pc; base += kChunkDevBlock) {
        const uint32_t i = base + t, slot = first + i;
        const bool v = i < pc && slot < index_len && slot >= first && indices[slot] < tri_count;
        const unsigned long long b = __ballot(v);
        if ((t & 63u) == 0u) s_wave[t >> 6] = static_cast<uint32_t>(__popcll(b));
        __syncthreads();
        const uint32_t at = s_count + ((t >> 6) ? s_wave[0] : 0u) + static_cast<uint32_t>(__popcll(b & ((1ull << (t & 63u)) - 1ull)));
        if (v) slots[at] = slot;
        __syncthreads();
        if (t == 0) s_count += s_wave[0] + s_wave[1];
        __syncthreads();
    }
    const uint32_t count = s_count;
    if (count == 0u) {
        if (t == 0) {
            if (COUNT) {
                leaf_valid[li] = 0u;
                leaf_nodes[li] = 0u;
            } else {
                leaf_info[li] = ChunkInfo();   // ref = kChunkNone: nothing to hit below
            }
        }
        return;
    }
    const uint32_t r0 = COUNT ? 0u : rank0[li];
    for (uint32_t j = t; j < count; j += kChunkDevBlock) {
        chunkmath::make_item(tris[indices[slots[j]]], slots[j], r0 + j, items[j]);
        if (!COUNT) rank_slot[r0 + j] = slots[j];
    }
    uint32_t n2 = 2;
    while (n2 < count) n2 <<= 1;
    __syncthreads();
    // ---- ascending by the determinant-floor bound: the split-off sets are ranges of this order
    for (uint32_t j = t; j < n2; j += kChunkDevBlock) {
        key[j] = j < count ? static_cast<unsigned long long>(__double_as_longlong(items[j].cap)) : ~0ull;
        perm[j] = j;
    }
    chunk_bitonic(key, perm, n2, t);
    if (t == 0) {
        uint32_t ns = 0, lo = 0;
        const uint32_t hi = count;
        while (hi - lo >= 2u && ns + 1u < kChunkDevSets) {
            const double thr = 8.0 * items[perm[lo + (hi - lo) / 2u]].cap;   // 8 x the median (rb_bvh.cpp build_leaf)
            uint32_t mid = lo;
            while (mid < hi && items[perm[mid]].cap <= thr) ++mid;
            if (!(mid > lo && mid < hi)) break;
            set_lo[ns] = static_cast<uint16_t>(lo);
            set_hi[ns] = static_cast<uint16_t>(mid);
            ++ns;
            lo = mid;
        }
        set_lo[ns] = static_cast<uint16_t>(lo);
        set_hi[ns] = static_cast<uint16_t>(hi);
        s_nsets = ns + 1u;
    }
    __syncthreads();
    const uint32_t nsets = s_nsets;
    if (COUNT) {
        if (t == 0) {
            uint32_t chunks = 0;
            for (uint32_t s = 0; s < nsets; ++s) chunks += (set_hi[s] - set_lo[s] + kChunkTris - 1u) / kChunkTris;
            leaf_valid[li] = count;
            leaf_nodes[li] = chunks - 1u;
        }
        return;
    }
    // ---- the skeleton: a chain of joins over the sets (small bounds first), every set an unsplit entry
    if (t == 0) {
        uint32_t nt = 0, n_inner = 0, npend = 0, maxlevel = 0;
        auto entry = [&](uint32_t lo, uint32_t hi, uint32_t k, uint32_t level, bool chain) {
            ChunkDevT e;
            e.lo = static_cast<uint16_t>(lo);
            e.hi = static_cast<uint16_t>(hi);
            e.k = static_cast<uint16_t>(k);
            e.left = e.right = -1;
            e.level = static_cast<uint16_t>(level);
            e.inner = (chain || k > 1u) ? 1u : 0u;
            e.ord = e.inner ? static_cast<uint16_t>(n_inner++) : 0u;
            if (level > maxlevel) maxlevel = level;
            tn[nt] = e;
            if (!chain && k > 1u) pend[npend++] = static_cast<uint16_t>(nt);
            return nt++;
        };
        auto chunks_of = [&](uint32_t s) { return (set_hi[s] - set_lo[s] + kChunkTris - 1u) / kChunkTris; };
        if (nsets == 1u) {
            entry(set_lo[0], set_hi[0], chunks_of(0), 0u, false);
        } else {
            uint32_t cur = entry(0u, count, 0u, 0u, true);
            for (uint32_t s = 0; s + 1u < nsets; ++s) {
                const uint32_t lv = tn[cur].level + 1u;
                const uint32_t l = entry(set_lo[s], set_hi[s], chunks_of(s), lv, false);
                const uint32_t r = (s + 2u < nsets) ? entry(set_lo[s + 1u], count, 0u, lv, true) : entry(set_lo[s + 1u], set_hi[s + 1u], chunks_of(s + 1u), lv, false);
                tn[cur].left = static_cast<int16_t>(l);
                tn[cur].right = static_cast<int16_t>(r);
                cur = r;
            }
        }
        s_root = 0u;
        s_nt = nt;
        s_npend = npend;
        s_maxlevel = maxlevel;
        // (n_inner continues below through tn[].ord of the entries the splits make; kept in s_wave[0])
        s_wave[0] = n_inner;
    }
    __syncthreads();
    // ---- median splits, a level of every unsplit entry at a time: longest axis of the centroids' extent, the items of the
    // entry sorted along it (one sort of the whole leaf, keyed by the entry's first position so that nothing leaves its range)
    while (s_npend > 0u) {
        const uint32_t npend = s_npend;
        for (uint32_t j = t; j < count; j += kChunkDevBlock) segof[j] = 0xFFu;
        __syncthreads();
        if (t < npend) {
            const ChunkDevT e = tn[pend[t]];
            float cmn[3] = {chunkmath::f_inf(), chunkmath::f_inf(), chunkmath::f_inf()}, cmx[3] = {-chunkmath::f_inf(), -chunkmath::f_inf(), -chunkmath::f_inf()};
            for (uint32_t j = e.lo; j < e.hi; ++j) {
                const ChunkItem& it = items[perm[j]];
                for (int a = 0; a < 3; ++a) {
                    const float c = 0.5f * (it.mn[a] + it.mx[a]);
                    cmn[a] = c < cmn[a] ? c : cmn[a];
                    cmx[a] = c > cmx[a] ? c : cmx[a];
                }
                segof[j] = static_cast<uint8_t>(t);
            }
            const float ex = cmx[0] - cmn[0], ey = cmx[1] - cmn[1], ez = cmx[2] - cmn[2];
            pend_axis[t] = (ex > ey && ex > ez) ? 0u : ((ey > ez) ? 1u : 2u);
        }
        __syncthreads();
        for (uint32_t j = t; j < n2; j += kChunkDevBlock) {
            unsigned long long k = ~0ull;
            if (j < count) {
                const uint32_t sg = segof[j];
                if (sg == 0xFFu) {
                    k = static_cast<unsigned long long>(j) << 32;
                } else {
                    const ChunkItem& it = items[perm[j]];
                    const uint32_t a = pend_axis[sg];
                    k = (static_cast<unsigned long long>(tn[pend[sg]].lo) << 32) | f2ord(0.5f * (it.mn[a] + it.mx[a]));
                }
            }
            key[j] = k;
        }
        chunk_bitonic(key, perm, n2, t);
        if (t == 0) {
            uint32_t nt = s_nt, n_inner = s_wave[0], nnext = 0, maxlevel = s_maxlevel;
            for (uint32_t i = 0; i < npend; ++i) {
                const uint32_t p = pend[i];
                const uint32_t lo = tn[p].lo, hi = tn[p].hi, k = tn[p].k, cnt = hi - lo, kl = k / 2u;
                uint32_t take = cnt * kl / k;   // the left half gets floor(k / 2) of the k chunks: sizes stay within one of count / k
                if (take < 1u) take = 1u;
                const uint32_t mid = lo + take, lv = tn[p].level + 1u;
                const uint32_t kk[2] = {kl, k - kl}, a[2] = {lo, mid}, b[2] = {mid, hi};
                for (int side = 0; side < 2; ++side) {
                    ChunkDevT e;
                    e.lo = static_cast<uint16_t>(a[side]);
                    e.hi = static_cast<uint16_t>(b[side]);
                    e.k = static_cast<uint16_t>(kk[side]);
                    e.left = e.right = -1;
                    e.level = static_cast<uint16_t>(lv);
                    e.inner = kk[side] > 1u ? 1u : 0u;
                    e.ord = e.inner ? static_cast<uint16_t>(n_inner++) : 0u;
                    tn[nt] = e;
                    if (e.inner) pend_next[nnext++] = static_cast<uint16_t>(nt);
                    if (side == 0) tn[p].left = static_cast<int16_t>(nt);
                    else tn[p].right = static_cast<int16_t>(nt);
                    ++nt;
                }
                if (lv > maxlevel) maxlevel = lv;
            }
            for (uint32_t i = 0; i < nnext; ++i) pend[i] = pend_next[i];
            s_nt = nt;
            s_wave[0] = n_inner;
            s_npend = nnext;
            s_maxlevel = maxlevel;
        }
        __syncthreads();
    }
    // ---- the chunks: positions in the final order, tight box, margins, cone
    const uint32_t nt = s_nt, p0 = r0, nd0 = node0[li];   // a leaf's first position = its first rank: every valid slot is in exactly one chunk
    for (uint32_t j = t; j < count; j += kChunkDevBlock) {
        pos_slot[p0 + j] = items[perm[j]].slot;
        pos_rank[p0 + j] = items[perm[j]].rank;
    }
    for (uint32_t i = t; i < nt; i += kChunkDevBlock) {
        const ChunkDevT e = tn[i];
        if (e.inner) continue;
        ChunkInfo r;
        for (uint32_t j = e.lo; j < e.hi; ++j) {
            const ChunkItem& it = items[perm[j]];
            r.cap = chunkmath::dmax(r.cap, it.cap);
            r.fa = chunkmath::dmax(r.fa, it.fa);
            r.cap_l = chunkmath::dmax(r.cap_l, it.cap_l);
            r.fa_l = chunkmath::dmax(r.fa_l, it.fa_l);
            for (int a = 0; a < 3; ++a) {
                r.mn[a] = it.mn[a] < r.mn[a] ? it.mn[a] : r.mn[a];
                r.mx[a] = it.mx[a] > r.mx[a] ? it.mx[a] : r.mx[a];
            }
        }
        const uint32_t lo = e.lo;
        r.cone = chunkmath::cone_of([&](uint32_t q) -> const ChunkItem& { return items[perm[lo + q]]; }, static_cast<uint32_t>(e.hi - e.lo));
        r.ref = kChunkLeaf | ((static_cast<uint32_t>(e.hi - e.lo) - 1u) << 26) | (p0 + lo);
        info[i] = r;
    }
    __syncthreads();
    // ---- the library's own nodes, deepest level first (children carry the library's boxes: no kChunkExact)
    for (uint32_t lv = s_maxlevel + 1u; lv-- > 0u;) {
        for (uint32_t i = t; i < nt; i += kChunkDevBlock) {
            const ChunkDevT e = tn[i];
            if (!e.inner || e.level != lv) continue;
            const ChunkInfo l = info[e.left], r = info[e.right];
            ChunkNode n;
            chunkmath::fill_child(l, l.mn, l.mx, n.lmin, n.lref, n.lmax, n.lfac, n.lcone);
            chunkmath::fill_child(r, r.mn, r.mx, n.rmin, n.rref, n.rmax, n.rfac, n.rcone);
            nodes[nd0 + e.ord] = n;
            ChunkInfo up;
            chunkmath::combine(l, r, up);
            up.ref = nd0 + e.ord;
            info[i] = up;
        }
        __syncthreads();
    }
    if (t == 0) leaf_info[li] = info[s_root];
}

__global__ void k_chunk_totals(const uint32_t* leaf_valid, const uint32_t* leaf_nodes, const uint32_t* rank0, const uint32_t* node0,
                               uint32_t n_leaves, uint32_t* totals) {
    totals[0] = rank0[n_leaves - 1u] + leaf_valid[n_leaves - 1u];
    totals[1] = node0[n_leaves - 1u] + leaf_nodes[n_leaves - 1u];
}
}  // namespace

int device_chunk_tree_build(const rb_gpu_triangle* d_tris, uint32_t tri_count, const uint32_t* d_indices, uint32_t index_len,
                            const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, DeviceChunkTree* out, void* stream_) {
    using chunkmath::ChunkInfo;
    *out = DeviceChunkTree{};
    if (node_count < 2u || index_len == 0u || tri_count == 0u || ref_nodes[0].primitive_count > 0u) return -1;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    std::vector<uint32_t> order, leaves;
    if (!chunk_visit_order(ref_nodes, node_count, order, leaves) || leaves.empty() || leaves.size() >= (1u << 30)) return -1;
    const uint32_t n_leaves = static_cast<uint32_t>(leaves.size());
    std::vector<uint2> desc(n_leaves);
    for (uint32_t i = 0; i < n_leaves; ++i) {
        const rb_bvh_node& n = ref_nodes[leaves[i]];
        if (n.primitive_count > kChunkDeviceLeafMax) return -1;   // a caller's tree with fatter leaves than the block holds: host builder
        desc[i] = make_uint2(n.first_primitive, n.primitive_count);
    }
    size_t scan_bytes = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u, n_leaves,
                                           rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return static_cast<int>(e);
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += align256(bytes); return o; };
    const size_t o_desc = carve(8u * size_t(n_leaves)), o_valid = carve(4u * size_t(n_leaves)), o_nodes = carve(4u * size_t(n_leaves));
    const size_t o_rank0 = carve(4u * size_t(n_leaves)), o_node0 = carve(4u * size_t(n_leaves)), o_info = carve(sizeof(ChunkInfo) * size_t(n_leaves));
    const size_t o_tot = carve(8), o_scan = carve(scan_bytes);
    char* base = nullptr;
    e = hipMalloc(reinterpret_cast<void**>(&base), off);
    if (e != hipSuccess) return static_cast<int>(e);
    ChunkNode* d_nodes = nullptr;
    uint32_t *d_pos_slot = nullptr, *d_pos_rank = nullptr, *d_rank_slot = nullptr;
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(base);
        if (d_nodes) (void)hipFree(d_nodes);
        if (d_pos_slot) (void)hipFree(d_pos_slot);
        if (d_pos_rank) (void)hipFree(d_pos_rank);
        if (d_rank_slot) (void)hipFree(d_rank_slot);
        return code;
    };
    auto at = [&](size_t o) { return base + o; };
    uint2* d_desc = reinterpret_cast<uint2*>(at(o_desc));
    uint32_t *d_valid = reinterpret_cast<uint32_t*>(at(o_valid)), *d_lnodes = reinterpret_cast<uint32_t*>(at(o_nodes));
    uint32_t *d_rank0 = reinterpret_cast<uint32_t*>(at(o_rank0)), *d_node0 = reinterpret_cast<uint32_t*>(at(o_node0));
    ChunkInfo* d_info = reinterpret_cast<ChunkInfo*>(at(o_info));
    uint32_t* d_tot = reinterpret_cast<uint32_t*>(at(o_tot));
    e = hipMemcpyAsync(d_desc, desc.data(), 8u * size_t(n_leaves), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return fail(static_cast<int>(e));
    hipLaunchKernelGGL(k_chunk_leaves<true>, dim3(n_leaves), dim3(kChunkDevBlock), 0, stream, d_tris, tri_count, d_indices, index_len, d_desc, n_leaves,
                       d_valid, d_lnodes, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    e = rocprim::exclusive_scan(at(o_scan), scan_bytes, d_valid, d_rank0, 0u, n_leaves, rocprim::plus<uint32_t>(), stream);
    if (e == hipSuccess) e = rocprim::exclusive_scan(at(o_scan), scan_bytes, d_lnodes, d_node0, 0u, n_leaves, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return fail(static_cast<int>(e));
    hipLaunchKernelGGL(k_chunk_totals, dim3(1), dim3(1), 0, stream, d_valid, d_lnodes, d_rank0, d_node0, n_leaves, d_tot);
    uint32_t tot[2] = {0, 0};
    e = hipMemcpyAsync(tot, d_tot, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(static_cast<int>(e));
    const uint32_t rank = tot[0], n_dev_nodes = tot[1];
    // (the scans are 32-bit sums of at most index_len <= 2^31 slots; the limits below are the reference formats': rank in 26 bits)
    if (rank == 0u || rank >= (1u << 26) - 64u) return fail(-1);
    const size_t n_top_max = order.size() - leaves.size();
    if (size_t(n_dev_nodes) + n_top_max >= (1u << 30)) return fail(-1);
    e = hipMalloc(reinterpret_cast<void**>(&d_nodes), sizeof(ChunkNode) * (size_t(n_dev_nodes) + n_top_max));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_pos_slot), 4u * size_t(rank));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_pos_rank), 4u * size_t(rank));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_rank_slot), 4u * size_t(rank));
    if (e != hipSuccess) return fail(static_cast<int>(e));
    hipLaunchKernelGGL(k_chunk_leaves<false>, dim3(n_leaves), dim3(kChunkDevBlock), 0, stream, d_tris, tri_count, d_indices, index_len, d_desc, n_leaves,
                       nullptr, nullptr, d_rank0, d_node0, d_nodes, d_pos_slot, d_pos_rank, d_rank_slot, d_info);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(static_cast<int>(e));
    std::vector<ChunkInfo> leaf_info(n_leaves), info(node_count);
    e = hipMemcpyAsync(leaf_info.data(), d_info, sizeof(ChunkInfo) * size_t(n_leaves), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return fail(static_cast<int>(e));
    for (uint32_t i = 0; i < n_leaves; ++i) info[leaves[i]] = leaf_info[i];
    std::vector<ChunkNode> top;
    top.reserve(n_top_max);
    chunk_top_pass(ref_nodes, node_count, order, info, top, n_dev_nodes);
    const uint32_t root = info[0].ref, depth = info[0].depth;
    if (root == kChunkNone || depth + 1u > stack_limit) return fail(-1);
    if (!top.empty()) {
        e = hipMemcpyAsync(d_nodes + n_dev_nodes, top.data(), sizeof(ChunkNode) * top.size(), hipMemcpyHostToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);   // `top` is a local
        if (e != hipSuccess) return fail(static_cast<int>(e));
    }
    (void)hipFree(base);
    out->nodes = d_nodes;
    out->n_nodes = size_t(n_dev_nodes) + top.size();
    out->nodes_capacity = size_t(n_dev_nodes) + n_top_max;
    out->pos_slot = d_pos_slot;
    out->pos_rank = d_pos_rank;
    out->rank_slot = d_rank_slot;
    out->n_pos = rank;
    out->root = root;
    out->depth = depth;
    return 0;
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
