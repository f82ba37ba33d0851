// rb_bvh.cpp -- host-side producer and validator of the BVH the kernels walk.
//
// build: the reference's median-split builder, crates/engine-bvh/src/bvh.rs:87-150
//   (bounds over the three vertices, leaf if count <= 128, split on the strictly
//   longest axis x > y > z, nth_element on the triangle centroid, left then right,
//   pre-order numbering).  `select_nth_unstable_by` leaves the order inside each
//   half unspecified, so tree *bytes* are not reproducible across implementations;
//   parity is defined on hits.
// validate: the WGSL traversal (shader.wgsl:282-392) tolerates any node array thanks
//   to robust buffer access and a 1024-entry stack; a HIP kernel does not, so a
//   malformed tree (cycle, out-of-range child, depth beyond the kernel stack) is
//   rejected on the host before it can hang the GPU.
#include "rb_internal.hpp"
#include "rb_chunk_math.hpp"

#include <algorithm>
#include <array>
#include <functional>
#include <future>
#include <thread>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace rb {

namespace {

constexpr size_t kMaxLeaf = 128;  // bvh.rs:12

// BVH::new / build_node, bvh.rs:87-150, restated: bounds over the three vertices of every triangle of the range, a leaf at
// <= 128 triangles, else the longest axis of those bounds (x only if strictly the longest, else y if longer than z, else z),
// mid = first + count / 2, the range partitioned around the element of rank mid by centroid[axis] (select_nth_unstable_by:
// std::nth_element here; the order inside the halves is unspecified there too), left then right, pre-order numbering.
// The halves are independent, and a subtree's node count follows from its triangle count alone, so every node's index is
// known before anything is built: the top levels fork threads (r04: C5's 10^6 triangles 133 -> ~ 25 ms on 16 CPUs; the
// reference rebuilds this tree on the CPU for every render, scene_engine_adapter.rs:435-440).  Centroids and per-triangle
// bounds are worked out once (the same f32 operations as bvh.rs:152-154 and aabb.rs), not once per comparison.
struct Builder {
    const rb_gpu_triangle* tris;
    std::vector<uint32_t>& idx;
    std::vector<rb_bvh_node>& nodes;
    std::vector<float> cen[3];       // centroid per triangle and axis
    std::vector<float> tmn, tmx;     // bounds per triangle (3 floats each)
    uint32_t fork_levels = 0;

    static float centroid(const rb_gpu_triangle& t, int axis) {
        return ((t.v0[axis] + t.v1[axis]) + t.v2[axis]) / 3.0f;  // bvh.rs:152-154
    }
    static size_t nodes_of(size_t count) {   // nodes of the subtree over `count` triangles
        return count <= kMaxLeaf ? 1 : 1 + nodes_of(count / 2) + nodes_of(count - count / 2);
    }
    void prepare(size_t n, size_t threads) {
        for (int a = 0; a < 3; ++a) cen[a].resize(n);
        tmn.resize(3 * n);
        tmx.resize(3 * n);
        auto work = [&](size_t first, size_t last) {
            for (size_t i = first; i < last; ++i) {
                const rb_gpu_triangle& t = tris[i];
                for (int a = 0; a < 3; ++a) {
                    cen[a][i] = centroid(t, a);
                    tmn[3 * i + a] = std::min(t.v0[a], std::min(t.v1[a], t.v2[a]));
                    tmx[3 * i + a] = std::max(t.v0[a], std::max(t.v1[a], t.v2[a]));
                }
            }
        };
        if (threads <= 1 || n < 65536) {
            work(0, n);
            return;
        }
        std::vector<std::thread> pool;
        const size_t per = (n + threads - 1) / threads;
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(work, std::min(t * per, n), std::min((t + 1) * per, n));
        for (std::thread& th : pool) th.join();
    }

    void node(size_t first, size_t count, uint32_t me, uint32_t level) {
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i) {
            const size_t t = idx[i];
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], tmn[3 * t + a]);
                mx[a] = std::max(mx[a], tmx[3 * t + a]);
            }
        }
        rb_bvh_node n;
        std::memset(&n, 0, sizeof n);
        std::memcpy(n.aabb_min, mn, sizeof mn);
        std::memcpy(n.aabb_max, mx, sizeof mx);
        if (count <= kMaxLeaf) {
            n.first_primitive = static_cast<uint32_t>(first);
            n.primitive_count = static_cast<uint32_t>(count);
            nodes[me] = n;
            return;
        }
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);  // bvh.rs:125-135
        const size_t mid = first + count / 2;
        const float* c = cen[axis].data();
        std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                         [c](uint32_t a, uint32_t b) { return c[a] < c[b]; });
        const uint32_t l = me + 1u, r = me + 1u + static_cast<uint32_t>(nodes_of(mid - first));
        n.left = l;
        n.right = r;
        nodes[me] = n;
        if (level < fork_levels && count >= 32768) {
            std::thread left([&] { node(first, mid - first, l, level + 1); });
            node(mid, first + count - mid, r, level + 1);
            left.join();
        } else {
            node(first, mid - first, l, level + 1);
            node(mid, first + count - mid, r, level + 1);
        }
    }
};

}  // namespace

void bvh_build(const rb_gpu_triangle* tris, size_t n_tris, std::vector<rb_bvh_node>& nodes,
               std::vector<uint32_t>& indices) {
    nodes.clear();
    indices.resize(n_tris);
    for (size_t i = 0; i < n_tris; ++i) indices[i] = static_cast<uint32_t>(i);
    if (n_tris == 0) return;  // the adapter never builds an empty tree (scene_engine_adapter.rs:435-440)
    const char* seq = std::getenv("RB_HOST_BUILD_SEQUENTIAL");
    const size_t threads = (seq && seq[0] == '1') ? 1u : std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    Builder b{tris, indices, nodes, {}, {}, {}, 0};
    for (size_t t = 1; t < threads; t *= 2) b.fork_levels++;   // 2^levels subtrees in flight
    b.prepare(n_tris, threads);
    nodes.resize(Builder::nodes_of(n_tris));
    b.node(0, n_tris, 0u, 0u);
}

// Iterative DFS from node 0 following exactly the children the shader would push
// (shader.wgsl:376-387: a child index >= node_count is skipped, not an error).
// Returns false with `why` set when the traversal could revisit a node (cycle or
// DAG -- a DAG would double-test triangles but terminate; a cycle never does) or
// needs more stack than the kernels have.
bool bvh_validate(const rb_bvh_node* nodes, uint32_t node_count, uint32_t max_stack, std::string& why,
                  uint32_t* depth_out) {
    if (depth_out) *depth_out = 0;
    if (node_count == 0) return true;
    std::vector<uint8_t> seen(node_count, 0);
    struct Item { uint32_t node, sp; };
    std::vector<Item> st;
    st.push_back({0u, 1u});
    uint32_t max_sp = 1;
    while (!st.empty()) {
        const Item it = st.back();
        st.pop_back();
        if (seen[it.node]) {
            why = "BVH node " + std::to_string(it.node) + " is reachable twice (cycle or shared child)";
            return false;
        }
        seen[it.node] = 1;
        const rb_bvh_node& n = nodes[it.node];
        if (n.primitive_count > 0) continue;
        // stack occupancy when this node is popped is it.sp - 1; it then pushes up to two
        uint32_t sp = it.sp - 1;
        if (n.left < node_count) st.push_back({n.left, 0});
        if (n.right < node_count) st.push_back({n.right, 0});
        uint32_t pushed = (n.left < node_count) + (n.right < node_count);
        // left is pushed first, so it sits below right: left is popped with occupancy sp+1,
        // right with occupancy sp+2 (when both exist)
        size_t k = st.size();
        if (pushed == 2) {
            st[k - 2].sp = sp + 1;
            st[k - 1].sp = sp + 2;
        } else if (pushed == 1) {
            st[k - 1].sp = sp + 1;
        }
        max_sp = std::max(max_sp, sp + pushed);
    }
    if (depth_out) *depth_out = max_sp;
    if (max_sp > max_stack) {
        why = "BVH traversal needs a stack of " + std::to_string(max_sp) + " entries; kernels provide " +
              std::to_string(max_stack);
        return false;
    }
    return true;
}


// ---------------------------------------------------------------- sphere tree --
// The reference scans every sphere on every segment (shader.wgsl:574-586), which is
// intractable beyond a few thousand spheres (BASELINE config C4 has 10^6).  This builds
// a tree over the spheres' tight boxes: median splits on the longest axis of the centres'
// bounds down to leaves of <= kSphLeaf spheres, two levels of splits per 4-wide node
// (rb_internal.hpp SphereNode4).  The traversal (rb_device_shade.hpp sphere_node_step, rb_kernels.hip
// k_trace_sph) grows the boxes per ray by a margin that covers the rounding error of the reference's own
// discriminant, runs the reference's exact intersect_sphere on every candidate and
// breaks ties by the lower original index -- the winner of the linear scan.  The host builder is the fallback and the
// checker of the device builder (rb_build.hip), which is the default from 1024 spheres up.
namespace {
// The library's two-box triangle builder forks big subtrees onto threads: the halves touch disjoint ranges of the
// item array, the left one is built into its own node array, and appending left then right with
// their node indices shifted reproduces the sequential (pre-order) numbering exactly.
constexpr size_t kParallelCount = 16384;
inline void append_subtree(std::vector<SphereNode>& nodes, const std::vector<SphereNode>& v, uint32_t& ref) {
    const uint32_t off = static_cast<uint32_t>(nodes.size());
    for (SphereNode c : v) {
        if (!(c.left & 0x80000000u)) c.left += off;
        if (!(c.right & 0x80000000u)) c.right += off;
        nodes.push_back(c);
    }
    if (!(ref & 0x80000000u)) ref += off;
}

struct SBuilder {
    const rb_sphere* sph;
    std::vector<uint32_t>& order;
    std::vector<SphereNode4>& nodes;
    uint32_t levels = 0;   // 4-wide levels on the deepest path

    void bounds(size_t first, size_t count, float mn[3], float mx[3]) const {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i) {
            const rb_sphere& s = sph[order[i]];
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], s.center[a] - s.radius);
                mx[a] = std::max(mx[a], s.center[a] + s.radius);
            }
        }
    }
    // median split of [first, first + count) along the longest axis of the centres; returns the axis
    uint32_t split(size_t first, size_t count, size_t& mid) {
        float cmn[3], cmx[3];
        for (int a = 0; a < 3; ++a) {
            cmn[a] = std::numeric_limits<float>::infinity();
            cmx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i)
            for (int a = 0; a < 3; ++a) {
                cmn[a] = std::min(cmn[a], sph[order[i]].center[a]);
                cmx[a] = std::max(cmx[a], sph[order[i]].center[a]);
            }
        const float ex = cmx[0] - cmn[0], ey = cmx[1] - cmn[1], ez = cmx[2] - cmn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
        mid = first + count / 2;
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                         [&](uint32_t a, uint32_t b) { return sph[a].center[axis] < sph[b].center[axis]; });
        return static_cast<uint32_t>(axis);
    }
    static uint32_t leaf_ref(size_t first, size_t count) {
        return 0x80000000u | (static_cast<uint32_t>(count - 1) << 27) | static_cast<uint32_t>(first);
    }
    void set_child(SphereNode4& n, int k, size_t first, size_t count, uint32_t ref) const {
        float mn[3], mx[3];
        bounds(first, count, mn, mx);
        n.set_box(k, mn, mx);
        n.ref[k] = ref;
    }
    // returns a child reference: a leaf (count <= kSphLeaf) or the index of a node over two levels of splits
    uint32_t build(size_t first, size_t count, uint32_t level) {
        if (count <= kSphLeaf) return leaf_ref(first, count);
        levels = std::max(levels, level);
        const uint32_t me = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        SphereNode4 n{};
        for (int k = 0; k < 4; ++k) n.ref[k] = kSphNone;
        size_t mid = 0;
        n.axes = split(first, count, mid);
        const size_t hf[2] = {first, mid}, hc[2] = {mid - first, first + count - mid};
        for (int h = 0; h < 2; ++h) {
            if (hc[h] <= kSphLeaf) {   // this half is a leaf already: one child
                set_child(n, 2 * h, hf[h], hc[h], leaf_ref(hf[h], hc[h]));
                continue;
            }
            size_t m2 = 0;
            n.axes |= split(hf[h], hc[h], m2) << (2 + 2 * h);
            const uint32_t r0 = build(hf[h], m2 - hf[h], level + 1), r1 = build(m2, hf[h] + hc[h] - m2, level + 1);
            set_child(n, 2 * h, hf[h], m2 - hf[h], r0);
            set_child(n, 2 * h + 1, m2, hf[h] + hc[h] - m2, r1);
        }
        nodes[me] = n;
        return me;
    }
};
}  // namespace

void sphere_bvh_build(const rb_sphere* spheres, size_t n, std::vector<SphereNode4>& nodes,
                      std::vector<uint32_t>& order, uint32_t* root_ref, uint32_t* depth) {
    nodes.clear();
    order.resize(n);
    for (size_t i = 0; i < n; ++i) order[i] = static_cast<uint32_t>(i);
    SBuilder b{spheres, order, nodes};
    *root_ref = n ? b.build(0, n, 1) : 0x80000000u;
    *depth = b.levels;
}


// ------------------------------------------------------------ the library's own triangle tree --
// The reference walks its 128-triangle-leaf tree without t-culling or ordering (shader.wgsl:282-392); on a
// 50k-triangle mesh that is ~430 triangle tests per segment.  This builds a second tree over the SAME triangles
// (binned SAH, <= 2 per leaf, both children's boxes in the parent) that the kernels walk near-child-first with
// culling.  To keep the reference's winner it also emits, per triangle slot, the reference leaf that holds it
// and its rank in the reference's visit order, plus a parent array of the reference tree: an improving
// candidate is accepted only if every reference node from its leaf up to the root passes the reference's own
// slab test, and equal t resolves by rank.
//
// Culling must never lose a triangle the reference would report as hit, and the reference's f32
// Moller-Trumbore can report a hit some distance away from the triangle: at most
// 26 u (|s| + L) L^2 / |a| + ... (rb_device_intersect.hpp, FastWalk::entry).  Two things
// are made here for that, rounded outwards:
//   per child of the library's tree  FA = the largest F_k below it, F_k bounding L_k^2 / |a^| over the hits of
//       triangle k that the tree's walk answers for (L = longer of the two edges at v0, N = |e1 x e2|):
//         "small" k (L_k^2 / 1e-6 <= kFastSmallCap): F_k = L_k^2 / 1e-6 -- the reference accepts no |a^| < 1e-6, so
//                 this covers EVERY accepted hit of k;
//         "large" k: F_k = (L_k^2 / N_k) / (0.95 c0), c0 = kFastGrazeCos -- covers its hits with
//                 |cos(ray, normal)| >= c0;
//       +inf (always enter) beyond 1.5e5 or when a triangle below has N = 0;
//   per node of the REFERENCE tree  a cone {c cos(alpha), tan(alpha)} that contains the normal of every LARGE
//       triangle below it (either orientation): their hits with |cos| < c0 -- a ray within ~1.7 degrees of the
//       triangle's plane, where no useful bound exists -- are found by a second walk over the reference's own
//       tree that only enters nodes whose cone admits such a triangle (FastWalk::gnode_step / gleaf_step).  A
//       mesh of small triangles has no such node at all.
namespace {
constexpr float kInf = std::numeric_limits<float>::infinity();

struct FBuilder {
    const std::vector<float>& bmn;   // per item: tight box min (3 floats)
    const std::vector<float>& bmx;
    const std::vector<float>& q;     // per item: F_k (see above; +inf if N == 0)
    std::vector<uint32_t>& items;    // permuted in place
    std::vector<SphereNode>& nodes;
    uint32_t limit;
    uint32_t max_depth = 0;
    uint32_t par_levels = 0;   // levels below this call that may still fork a thread

    void bounds(size_t first, size_t count, float mn[3], float mx[3]) const {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i) {
            const uint32_t it = items[i];
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], bmn[it * 3 + a]);
                mx[a] = std::max(mx[a], bmx[it * 3 + a]);
            }
        }
    }
    float centroid(uint32_t it, int a) const { return 0.5f * (bmn[it * 3 + a] + bmx[it * 3 + a]); }
    float fa(size_t first, size_t count) const {
        float m = 0.0f;
        for (size_t i = first; i < first + count; ++i) m = std::max(m, q[items[i]]);
        return (m <= 1.5e5f) ? m : kInf;
    }
    static float area(const float mn[3], const float mx[3]) {
        const float x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
        return (x < 0 || y < 0 || z < 0) ? 0.0f : 2.0f * (x * y + y * z + z * x);
    }

    uint32_t build(size_t first, size_t count, uint32_t depth) {
        max_depth = std::max(max_depth, depth);
        if (count <= 2) return 0x80000000u | (static_cast<uint32_t>(count - 1) << 28) | static_cast<uint32_t>(first);
        float cmn[3], cmx[3];
        for (int a = 0; a < 3; ++a) {
            cmn[a] = std::numeric_limits<float>::infinity();
            cmx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i)
            for (int a = 0; a < 3; ++a) {
                const float c = centroid(items[i], a);
                cmn[a] = std::min(cmn[a], c);
                cmx[a] = std::max(cmx[a], c);
            }
        size_t mid = first + count / 2;
        int split_axis = -1;
        // levels left if we balance from here; switch to median splits before the stack limit
        uint32_t need = 0;
        for (size_t c = count; c > 2; c = (c + 1) / 2) ++need;
        const bool force_median = depth + need + 2 >= limit;
        if (!force_median) {
            constexpr int B = 16;
            float best = std::numeric_limits<float>::infinity();
            int best_bin = -1;
            for (int a = 0; a < 3; ++a) {
                const float ext = cmx[a] - cmn[a];
                if (!(ext > 0.0f)) continue;
                uint32_t cnt[B] = {0};
                float bmin_[B][3], bmax_[B][3];
                for (int k = 0; k < B; ++k)
                    for (int j = 0; j < 3; ++j) {
                        bmin_[k][j] = std::numeric_limits<float>::infinity();
                        bmax_[k][j] = -std::numeric_limits<float>::infinity();
                    }
                const float scale = B / ext;
                for (size_t i = first; i < first + count; ++i) {
                    const uint32_t it = items[i];
                    int k = static_cast<int>((centroid(it, a) - cmn[a]) * scale);
                    k = std::min(std::max(k, 0), B - 1);
                    cnt[k]++;
                    for (int j = 0; j < 3; ++j) {
                        bmin_[k][j] = std::min(bmin_[k][j], bmn[it * 3 + j]);
                        bmax_[k][j] = std::max(bmax_[k][j], bmx[it * 3 + j]);
                    }
                }
                float lmn[3], lmx[3], rarea[B];
                uint32_t rcnt[B];
                float rmn[3], rmx[3];
                for (int j = 0; j < 3; ++j) { rmn[j] = std::numeric_limits<float>::infinity(); rmx[j] = -rmn[j]; }
                uint32_t acc = 0;
                for (int k = B - 1; k >= 1; --k) {
                    acc += cnt[k];
                    for (int j = 0; j < 3; ++j) { rmn[j] = std::min(rmn[j], bmin_[k][j]); rmx[j] = std::max(rmx[j], bmax_[k][j]); }
                    rcnt[k] = acc;
                    rarea[k] = area(rmn, rmx);
                }
                for (int j = 0; j < 3; ++j) { lmn[j] = std::numeric_limits<float>::infinity(); lmx[j] = -lmn[j]; }
                acc = 0;
                for (int k = 0; k < B - 1; ++k) {
                    acc += cnt[k];
                    for (int j = 0; j < 3; ++j) { lmn[j] = std::min(lmn[j], bmin_[k][j]); lmx[j] = std::max(lmx[j], bmax_[k][j]); }
                    if (acc == 0 || rcnt[k + 1] == 0) continue;
                    const float cost = area(lmn, lmx) * acc + rarea[k + 1] * rcnt[k + 1];
                    if (cost < best) { best = cost; best_bin = k; split_axis = a; }
                }
            }
            if (split_axis >= 0) {
                const int a = split_axis;
                const float scale = B / (cmx[a] - cmn[a]);
                auto it = std::partition(items.begin() + first, items.begin() + first + count, [&](uint32_t id) {
                    int k = static_cast<int>((centroid(id, a) - cmn[a]) * scale);
                    k = std::min(std::max(k, 0), B - 1);
                    return k <= best_bin;
                });
                mid = static_cast<size_t>(it - items.begin());
                if (mid == first || mid == first + count) split_axis = -1;
            }
        }
        if (split_axis < 0) {  // median on the longest centroid axis (also the depth-limit path)
            const float ex = cmx[0] - cmn[0], ey = cmx[1] - cmn[1], ez = cmx[2] - cmn[2];
            const int a = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
            mid = first + count / 2;
            std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + count,
                             [&](uint32_t x, uint32_t y) { return centroid(x, a) < centroid(y, a); });
        }
        const uint32_t me = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        SphereNode n;
        bounds(first, mid - first, n.lmin, n.lmax);
        bounds(mid, first + count - mid, n.rmin, n.rmax);
        if (count >= kParallelCount && par_levels > 0u) {
            std::vector<SphereNode> lv, rv;
            FBuilder lb{bmn, bmx, q, items, lv, limit, 0, par_levels - 1u};
            FBuilder rb{bmn, bmx, q, items, rv, limit, 0, par_levels - 1u};
            auto fut = std::async(std::launch::async, [&] { return lb.build(first, mid - first, depth + 1); });
            uint32_t rref = rb.build(mid, first + count - mid, depth + 1);
            uint32_t lref = fut.get();
            append_subtree(nodes, lv, lref);
            append_subtree(nodes, rv, rref);
            n.left = lref;
            n.right = rref;
            max_depth = std::max(max_depth, std::max(lb.max_depth, rb.max_depth));
        } else {
            n.left = build(first, mid - first, depth + 1);
            n.right = build(mid, first + count - mid, depth + 1);
        }
        // the two pad words: FA of each child (scales the inflation of that child's box)
        const float al = fa(first, mid - first), ar = fa(mid, first + count - mid);
        std::memcpy(&n._pad0, &al, 4);
        std::memcpy(&n._pad1, &ar, 4);
        nodes[me] = n;
        return me;
    }
};
}  // namespace

// F_k of one triangle and whether it is "large" (see the comment above FBuilder), from the f32 edges the kernels
// use.  The device builder gets the flag from slot_meta and repeats the arithmetic in double (rb_build.hip).
TriBound tri_bound(const rb_gpu_triangle& t, float small_cap) { return chunkmath::tri_bound_hd(t.v0, t.v1, t.v2, small_cap); }

using chunkmath::DCone;   // cone of unit normals (rb_chunk_math.hpp: shared with the device builder)
using chunkmath::merge;

bool fast_bvh_prepare(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      const rb_bvh_node* ref_nodes, uint32_t node_count, FastTree& out, float small_cap) {
    out = FastTree{};
    out.small_cap = small_cap;
    if (node_count == 0 || index_len == 0) return false;
    // ---- reference visit order (right child first, shader.wgsl:376-387) and per-slot metadata
    out.ref_parent.assign(node_count, 0u);
    out.slot_meta.assign(static_cast<size_t>(index_len) * 2, 0xFFFFFFFFu);
    std::vector<uint32_t> st{0u};
    std::vector<uint32_t> slots, order;   // order: reachable nodes, parents before children
    std::vector<TriBound> bound(index_len);   // per slot (valid slots only), made once: the cones below use it again
    uint32_t rank = 0;
    while (!st.empty()) {
        const uint32_t ni = st.back();
        st.pop_back();
        order.push_back(ni);
        const rb_bvh_node& n = ref_nodes[ni];
        if (n.primitive_count > 0) {
            for (uint32_t i = 0; i < n.primitive_count; ++i) {
                const uint32_t slot = n.first_primitive + i;
                if (slot >= index_len) continue;                     // guard :331
                if (indices[slot] >= tri_count) { ++rank; continue; }  // guard :336
                if (out.slot_meta[slot * 2] != 0xFFFFFFFFu) return false;  // slot shared by two leaves: keep the reference walk
                bound[slot] = tri_bound(tris[indices[slot]], small_cap);
                const bool large = bound[slot].large;
                out.slot_meta[slot * 2] = ni;
                out.slot_meta[slot * 2 + 1] = rank++ | (large ? kSlotLarge : 0u);
                out.n_large += large ? 1u : 0u;
                slots.push_back(slot);
            }
        } else {
            if (n.left < node_count) { out.ref_parent[n.left] = ni; st.push_back(n.left); }
            if (n.right < node_count) { out.ref_parent[n.right] = ni; st.push_back(n.right); }
        }
    }
    if (slots.empty() || slots.size() >= (1u << 28)) return false;
    // ---- This walk leans on two properties every tree of the reference's builder has (bvh.rs:100-123) and a caller's own
    // tree need not: a child's box lies inside its parent's (the chain shortcut of reference_would_test), and a leaf's
    // box contains its triangles (the second pass culls on the reference's boxes).  A tree without them keeps the
    // reference walk (found by tests/test_gpu_parity.py::test_caller_made_trees, r03: boxes shrunk by hand).
    for (uint32_t ni : order) {
        const rb_bvh_node& n = ref_nodes[ni];
        auto inside = [&](const float* mn, const float* mx) {
            for (int a = 0; a < 3; ++a)
                if (!(n.aabb_min[a] <= mn[a] && mx[a] <= n.aabb_max[a])) return false;
            return true;
        };
        if (n.primitive_count > 0) {
            for (uint32_t i = 0; i < n.primitive_count; ++i) {
                const uint32_t slot = n.first_primitive + i;
                if (slot >= index_len || indices[slot] >= tri_count) continue;
                const rb_gpu_triangle& t = tris[indices[slot]];
                if (!inside(t.v0, t.v0) || !inside(t.v1, t.v1) || !inside(t.v2, t.v2)) return false;
            }
        } else {
            if (n.left < node_count && !inside(ref_nodes[n.left].aabb_min, ref_nodes[n.left].aabb_max)) return false;
            if (n.right < node_count && !inside(ref_nodes[n.right].aabb_min, ref_nodes[n.right].aabb_max)) return false;
        }
    }
    // ---- per reference node: the cone of the triangle normals below it (bottom-up: leaves from their triangles,
    // inner nodes by merging their children's cones), for the walk's second pass over near-degenerate hits
    std::vector<DCone> cones(node_count);
    std::vector<std::pair<uint32_t, uint32_t>> grange(node_count, {0u, 0u});
    for (size_t k = order.size(); k-- > 0;) {
        const uint32_t ni = order[k];
        const rb_bvh_node& n = ref_nodes[ni];
        DCone c;
        if (n.primitive_count > 0) {
            grange[ni].first = static_cast<uint32_t>(out.gslots.size());
            // direct, over the LARGE triangles of the leaf: axis = normalised sum of the sign-aligned normals,
            // alpha = largest angle to it
            std::vector<std::array<double, 3>> nrm;
            bool ok = true;
            for (uint32_t i = 0; i < n.primitive_count; ++i) {   // every large triangle goes into gslots, cone or no cone
                const uint32_t slot = n.first_primitive + i;
                if (slot >= index_len || indices[slot] >= tri_count) continue;
                const rb_gpu_triangle& t = tris[indices[slot]];
                const TriBound& b = bound[slot];
                if (!b.large) continue;
                out.gslots.push_back(slot);
                double l1 = 0, l2 = 0;
                for (int a = 0; a < 3; ++a) {
                    l1 += double(t.v1[a] - t.v0[a]) * double(t.v1[a] - t.v0[a]);
                    l2 += double(t.v2[a] - t.v0[a]) * double(t.v2[a] - t.v0[a]);
                }
                c.cap = std::max(c.cap, std::max(l1, l2) * 1e6 * (1.0 + 1e-5));
                if (!b.has_normal) { ok = false; continue; }   // no normal: any direction "grazes" it
                nrm.push_back({b.n[0], b.n[1], b.n[2]});
            }
            grange[ni].second = static_cast<uint32_t>(out.gslots.size()) - grange[ni].first;
            if (ok && nrm.empty()) {   // no large triangle in this leaf: never needs a visit
                c.valid = true; c.alpha = -1.0; c.c[0] = 1.0;   // alpha < 0 marks "empty" for the merges above it
            } else if (ok) {
                double sum[3] = {0, 0, 0};
                for (const auto& v : nrm) {
                    const double sg = (v[0] * nrm[0][0] + v[1] * nrm[0][1] + v[2] * nrm[0][2]) < 0.0 ? -1.0 : 1.0;
                    for (int a = 0; a < 3; ++a) sum[a] += sg * v[a];
                }
                const double len = std::sqrt(sum[0] * sum[0] + sum[1] * sum[1] + sum[2] * sum[2]);
                if (len > 1e-9) {
                    double cmin = 1.0;
                    for (int a = 0; a < 3; ++a) c.c[a] = sum[a] / len;
                    for (const auto& v : nrm) cmin = std::min(cmin, std::fabs(v[0] * c.c[0] + v[1] * c.c[1] + v[2] * c.c[2]));
                    c.alpha = std::acos(std::min(1.0, cmin)) + 1e-9;
                    c.valid = c.alpha < 1.55;
                }
            }
        } else {
            const bool hl = n.left < node_count, hr = n.right < node_count;
            if (hl && hr) {
                const DCone &a = cones[n.left], &b = cones[n.right];
                if (a.valid && a.alpha < 0.0) c = b;          // an empty side adds nothing
                else if (b.valid && b.alpha < 0.0) c = a;
                else c = merge(a, b);
                c.cap = std::max(a.cap, b.cap);
            } else if (hl) c = cones[n.left];
            else if (hr) c = cones[n.right];
            else { c.valid = true; c.alpha = -1.0; c.c[0] = 1.0; }
        }
        cones[ni] = c;
    }
    // one 64-B record per reference node for that pass: the reference's box and links, bit for bit, plus the cone
    out.gnodes.assign(node_count, GrazeNode{});
    for (uint32_t ni = 0; ni < node_count; ++ni) {
        const rb_bvh_node& n = ref_nodes[ni];
        GrazeNode& g = out.gnodes[ni];
        std::memcpy(g.bmin, n.aabb_min, 12);
        std::memcpy(g.bmax, n.aabb_max, 12);
        g.left = n.left;
        g.right = n.right;
        g.is_leaf = n.primitive_count > 0 ? 1u : 0u;
        g.first = grange[ni].first;
        g.count = grange[ni].second;
        const DCone& c = cones[ni];
        g.cap = (c.cap <= 1.5e5) ? static_cast<float>(c.cap * (1.0 + 1e-6)) : std::numeric_limits<float>::infinity();
        float* o = g.cone;   // all zeros = "always possible"
        if (!c.valid) continue;
        if (c.alpha < 0.0) { o[3] = -1.0f; continue; }   // no large triangle below: tan = -1 tells the walk never to enter
        const double cos_a = std::cos(c.alpha) * (1.0 - 1e-6) - 1e-7;
        if (!(cos_a > 0.0175)) continue;
        const double sin_a = std::sqrt(std::max(0.0, 1.0 - cos_a * cos_a));
        o[0] = static_cast<float>(c.c[0] * cos_a);
        o[1] = static_cast<float>(c.c[1] * cos_a);
        o[2] = static_cast<float>(c.c[2] * cos_a);
        o[3] = static_cast<float>(sin_a / cos_a * (1.0 + 1e-5) + 1e-7);
    }
    out.slots = std::move(slots);
    return true;
}

bool fast_bvh_build(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                    const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, FastTree& out, float small_cap) {
    if (!fast_bvh_prepare(tris, tri_count, indices, index_len, ref_nodes, node_count, out, small_cap)) return false;
    const std::vector<uint32_t> slots = std::move(out.slots);
    // ---- tight boxes per item; items are indices into `slots`
    const size_t n = slots.size();
    std::vector<float> bmn(n * 3), bmx(n * 3), q(n);
    float smn[3] = {1e30f, 1e30f, 1e30f}, smx[3] = {-1e30f, -1e30f, -1e30f};
    for (size_t i = 0; i < n; ++i) {
        const rb_gpu_triangle& t = tris[indices[slots[i]]];
        q[i] = tri_bound(t, small_cap).f;
        for (int a = 0; a < 3; ++a) {
            bmn[i * 3 + a] = std::min(t.v0[a], std::min(t.v1[a], t.v2[a]));
            bmx[i * 3 + a] = std::max(t.v0[a], std::max(t.v1[a], t.v2[a]));
            smn[a] = std::min(smn[a], bmn[i * 3 + a]);
            smx[a] = std::max(smx[a], bmx[i * 3 + a]);
        }
    }
    std::vector<uint32_t> items(n);
    for (size_t i = 0; i < n; ++i) items[i] = static_cast<uint32_t>(i);
    // up to 64 threads at the sixth level; RB_HOST_BUILD_SEQUENTIAL=1 (debug) builds on one thread -- the
    // tree is the same either way (tests/test_gpu_parity.py compares the walk's counters)
    const char* seq = std::getenv("RB_HOST_BUILD_SEQUENTIAL");
    FBuilder fb{bmn, bmx, q, items, out.nodes, stack_limit, 0, (seq && seq[0] == '1') ? 0u : 6u};
    out.root = fb.build(0, n, 1);
    out.depth = fb.max_depth + 1;
    out.slots.resize(n);
    for (size_t i = 0; i < n; ++i) out.slots[i] = slots[items[i]];
    out.margin = 0.0f;
    for (int a = 0; a < 3; ++a) {
        out.bmin[a] = smn[a];
        out.bmax[a] = smx[a];
    }
    out.root_amax = 0.0f;
    return out.depth <= stack_limit;
}


// ------------------------------------------------------------ the chunked walk's tree (ChunkTree) --
// The caller's tree on top -- every reference node becomes one two-box node whose child slots carry the children's
// reference boxes bit for bit, so the kernel can repeat shader.wgsl:664-671 on them -- and below every reference
// leaf a small median-split tree of the library's own (tight f32 boxes) down to chunks of at most kChunkTris
// triangles.  Per child slot, rounded outwards: the two bounds of L^2 / |a^| (tri_bound: the determinant floor, which
// covers every accepted hit, and the |cos| >= c0 bound, far smaller) and the cone of the normals below, which tells
// the walk which of the two a given ray needs.  Triangles are renumbered chunk by chunk; the reference's visit order
// survives as the rank that breaks ties in t.
namespace {
// the per-triangle and per-subtree arithmetic lives in rb_chunk_math.hpp (host + device)
using chunkmath::ChunkInfo;
using chunkmath::ChunkItem;
using chunkmath::chunk_g;
using chunkmath::empty_cone;
using chunkmath::merge_cones;

struct ChunkBuilder {
    ChunkTree& out;
    std::vector<ChunkItem>& items;

    void tight(size_t lo, size_t hi, float mn[3], float mx[3]) const {
        for (int a = 0; a < 3; ++a) { mn[a] = kInf; mx[a] = -kInf; }
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                mn[a] = std::min(mn[a], items[i].mn[a]);
                mx[a] = std::max(mx[a], items[i].mx[a]);
            }
    }
    // the library's own levels below one reference leaf: items [lo, hi)
    ChunkInfo build(size_t lo, size_t hi) {
        const size_t count = hi - lo;
        ChunkInfo r;
        if (count <= kChunkTris) {
            const uint32_t first = static_cast<uint32_t>(out.pos_slot.size());
            for (size_t i = lo; i < hi; ++i) {
                out.pos_slot.push_back(items[i].slot);
                out.pos_rank.push_back(items[i].rank);
                r.cap = std::max(r.cap, items[i].cap);
                r.fa = std::max(r.fa, items[i].fa);
                r.cap_l = std::max(r.cap_l, items[i].cap_l);
                r.fa_l = std::max(r.fa_l, items[i].fa_l);
            }
            tight(lo, hi, r.mn, r.mx);
            r.cone = chunkmath::cone_of([&](uint32_t i) -> const ChunkItem& { return items[lo + i]; }, static_cast<uint32_t>(count));
            r.ref = kChunkLeaf | (static_cast<uint32_t>(count - 1) << 26) | first;
            return r;
        }
        // k chunks in the end; the left half gets floor(k / 2) of them: sizes stay within one of count / k
        const size_t k = (count + kChunkTris - 1) / kChunkTris, kl = k / 2;
        const size_t mid = lo + std::max<size_t>(1, count * kl / k);
        float cmn[3] = {kInf, kInf, kInf}, cmx[3] = {-kInf, -kInf, -kInf};
        auto centroid = [&](const ChunkItem& it, int a) { return 0.5f * (it.mn[a] + it.mx[a]); };
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                cmn[a] = std::min(cmn[a], centroid(items[i], a));
                cmx[a] = std::max(cmx[a], centroid(items[i], a));
            }
        const float ex = cmx[0] - cmn[0], ey = cmx[1] - cmn[1], ez = cmx[2] - cmn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                         [&](const ChunkItem& x, const ChunkItem& y) { return centroid(x, axis) < centroid(y, axis); });
        return join(build(lo, mid), build(mid, hi), lo, mid, hi);
    }
    // one of the library's own nodes over the subtrees of items [lo, mid) and [mid, hi)
    ChunkInfo join(const ChunkInfo& l, const ChunkInfo& rr, size_t lo, size_t mid, size_t hi) {
        ChunkInfo r;
        ChunkNode n{};
        float mn[3], mx[3];
        tight(lo, mid, mn, mx);
        chunkmath::fill_child(l, mn, mx, n.lmin, n.lref, n.lmax, n.lfac, n.lcone);
        tight(mid, hi, mn, mx);
        chunkmath::fill_child(rr, mn, mx, n.rmin, n.rref, n.rmax, n.rfac, n.rcone);
        r.ref = static_cast<uint32_t>(out.nodes.size());   // children carry the library's own boxes: no kChunkExact
        out.nodes.push_back(n);
        chunkmath::combine(l, rr, r);
        tight(lo, hi, r.mn, r.mx);
        return r;
    }
    // The triangles of one reference leaf.  A chunk's margin is its worst triangle's: one wall-sized triangle among a
    // lamp's thousands (the reference's median splits put them in the same leaf) gives the whole chunk an unbounded
    // margin and a cone that admits every ray, so it would be tested on every visit of the leaf.  Triangles whose
    // determinant-floor bound is far above the leaf's typical one are therefore split off first, into subtrees of
    // their own (and again among themselves).
    ChunkInfo build_leaf(size_t lo, size_t hi) {
        const size_t count = hi - lo;
        if (count >= 2) {
            std::vector<double> caps(count);
            for (size_t i = 0; i < count; ++i) caps[i] = items[lo + i].cap;
            std::nth_element(caps.begin(), caps.begin() + count / 2, caps.end());
            const double thr = 8.0 * caps[count / 2];
            const auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const ChunkItem& x) { return x.cap <= thr; });
            const size_t mid = static_cast<size_t>(it - items.begin());
            if (mid > lo && mid < hi) return join(build(lo, mid), build_leaf(mid, hi), lo, mid, hi);
        }
        return build(lo, hi);
    }
};
}  // namespace

// The reference's visit order (right child first, shader.wgsl:376-387): nodes parents-first, leaves in visit order.
bool chunk_visit_order(const rb_bvh_node* ref_nodes, uint32_t node_count, std::vector<uint32_t>& order, std::vector<uint32_t>& leaves) {
    std::vector<uint32_t> st{0u};
    order.clear();
    leaves.clear();
    while (!st.empty()) {
        const uint32_t ni = st.back();
        st.pop_back();
        order.push_back(ni);
        if (order.size() > node_count) return false;
        const rb_bvh_node& n = ref_nodes[ni];
        if (n.primitive_count > 0) {
            leaves.push_back(ni);
        } else {
            if (n.left < node_count) st.push_back(n.left);
            if (n.right < node_count) st.push_back(n.right);
        }
    }
    return true;
}

// Bottom-up over the caller's internal nodes, children before parents: `info` holds what the subtrees below the reference
// leaves handed up; the nodes made here carry the REFERENCE boxes (kChunkExact) and are appended to `nodes`, whose first
// element has index `first_index` in the tree's node array.
void chunk_top_pass(const rb_bvh_node* ref_nodes, uint32_t node_count, const std::vector<uint32_t>& order, std::vector<ChunkInfo>& info,
                    std::vector<ChunkNode>& nodes, uint32_t first_index) {
    for (size_t k = order.size(); k-- > 0;) {
        const uint32_t ni = order[k];
        const rb_bvh_node& n = ref_nodes[ni];
        ChunkInfo& r = info[ni];
        if (n.primitive_count > 0) continue;
        const bool hl = n.left < node_count, hr = n.right < node_count;
        const ChunkInfo none;
        const ChunkInfo& l = hl ? info[n.left] : none;
        const ChunkInfo& rr = hr ? info[n.right] : none;
        if (l.ref == kChunkNone && rr.ref == kChunkNone) continue;   // nothing to hit below
        ChunkNode c{};
        const float zero[3] = {0, 0, 0};
        chunkmath::fill_child(l, hl ? ref_nodes[n.left].aabb_min : zero, hl ? ref_nodes[n.left].aabb_max : zero, c.lmin, c.lref, c.lmax,
                              c.lfac, c.lcone);
        chunkmath::fill_child(rr, hr ? ref_nodes[n.right].aabb_min : zero, hr ? ref_nodes[n.right].aabb_max : zero, c.rmin, c.rref,
                              c.rmax, c.rfac, c.rcone);
        r.ref = (first_index + static_cast<uint32_t>(nodes.size())) | kChunkExact;
        nodes.push_back(c);
        chunkmath::combine(l, rr, r);
    }
}

bool chunk_tree_build(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, ChunkTree& out) {
    out = ChunkTree{};
    if (node_count < 2 || index_len == 0 || ref_nodes[0].primitive_count > 0) return false;
    // one thread per CPU the process is granted (RB_HOST_BUILD_SEQUENTIAL=1: one thread, for debugging); fn(first, last)
    const char* seq = std::getenv("RB_HOST_BUILD_SEQUENTIAL");
    const size_t max_threads = (seq && seq[0] == '1') ? 1u : std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    auto parallel_for = [&](size_t n, const std::function<void(size_t, size_t)>& fn) {
        const size_t threads = (index_len < 16384u) ? 1u : std::min(max_threads, n);   // (C3's 50 176 slots: 1.9 ms on all threads, 3.3 ms on six)
        if (threads <= 1) {
            fn(0, n);
            return;
        }
        std::vector<std::thread> pool;
        const size_t per = (n + threads - 1) / threads;
        for (size_t t = 0; t < threads; ++t) pool.emplace_back(fn, std::min(t * per, n), std::min((t + 1) * per, n));
        for (std::thread& th : pool) th.join();
    };
    std::vector<uint32_t> order, leaves;
    if (!chunk_visit_order(ref_nodes, node_count, order, leaves)) return false;
    // ranks: a leaf's first rank = the valid slots (guards :331, :336) of the leaves visited before it
    auto valid = [&](uint32_t slot) { return slot < index_len && indices[slot] < tri_count; };
    std::vector<uint32_t> leaf_rank0(leaves.size() + 1, 0u);
    parallel_for(leaves.size(), [&](size_t first, size_t last) {
        for (size_t li = first; li < last; ++li) {
            const rb_bvh_node& n = ref_nodes[leaves[li]];
            uint32_t c = 0;
            for (uint32_t i = 0; i < n.primitive_count; ++i) c += valid(n.first_primitive + i) ? 1u : 0u;
            leaf_rank0[li + 1] = c;
        }
    });
    for (size_t li = 0; li < leaves.size(); ++li) {
        if (leaf_rank0[li] > (1u << 26)) return false;
        leaf_rank0[li + 1] += leaf_rank0[li];
    }
    const uint32_t rank = leaf_rank0[leaves.size()];
    if (rank == 0 || rank >= (1u << 26) - 64u) return false;
    out.rank_slot.resize(rank);
    std::vector<ChunkInfo> info(node_count);
    // ---- the library's own levels below every reference leaf.  The leaves are independent, so they are built in parallel
    // (C5's 8 192 leaves of 128 triangles: 85 ms on one thread), each into arrays of its own with references that count from
    // zero; laid end to end afterwards, the references shifted by where a leaf's arrays landed.
    struct LeafOut {
        ChunkTree t;
        ChunkInfo info;
    };
    std::vector<LeafOut> lout(leaves.size());
    parallel_for(leaves.size(), [&](size_t first, size_t last) {
        std::vector<ChunkItem> items;
        for (size_t li = first; li < last; ++li) {
            const rb_bvh_node& n = ref_nodes[leaves[li]];
            items.clear();
            uint32_t rk = leaf_rank0[li];
            for (uint32_t i = 0; i < n.primitive_count; ++i) {
                const uint32_t slot = n.first_primitive + i;
                if (!valid(slot)) continue;
                ChunkItem it;
                out.rank_slot[rk] = slot;
                chunkmath::make_item(tris[indices[slot]], slot, rk++, it);
                items.push_back(it);
            }
            if (items.empty()) continue;
            ChunkBuilder cb{lout[li].t, items};
            lout[li].info = cb.build_leaf(0, items.size());
        }
    });
    {
        std::vector<uint32_t> node_off(leaves.size() + 1, 0u), pos_off(leaves.size() + 1, 0u);
        for (size_t li = 0; li < leaves.size(); ++li) {
            node_off[li + 1] = node_off[li] + static_cast<uint32_t>(lout[li].t.nodes.size());
            pos_off[li + 1] = pos_off[li] + static_cast<uint32_t>(lout[li].t.pos_slot.size());
        }
        const size_t n_nodes = node_off[leaves.size()], n_pos = pos_off[leaves.size()];
        if (n_pos >= (1u << 26) || n_nodes >= (1u << 30)) return false;
        out.nodes.reserve(n_nodes + order.size());
        out.nodes.resize(n_nodes);
        out.pos_slot.resize(n_pos);
        out.pos_rank.resize(n_pos);
        parallel_for(leaves.size(), [&](size_t first, size_t last) {
            for (size_t li = first; li < last; ++li) {
                LeafOut& l = lout[li];
                const uint32_t no = node_off[li], po = pos_off[li];
                auto moved = [&](uint32_t ref) {   // leaf: first position += po; node: index += no
                    if (ref == kChunkNone) return ref;
                    return (ref & kChunkLeaf) ? ref + po : ref + no;
                };
                for (size_t k = 0; k < l.t.nodes.size(); ++k) {
                    ChunkNode c = l.t.nodes[k];
                    c.lref = moved(c.lref);
                    c.rref = moved(c.rref);
                    out.nodes[no + k] = c;
                }
                std::copy(l.t.pos_slot.begin(), l.t.pos_slot.end(), out.pos_slot.begin() + po);
                std::copy(l.t.pos_rank.begin(), l.t.pos_rank.end(), out.pos_rank.begin() + po);
                l.info.ref = moved(l.info.ref);
                info[leaves[li]] = l.info;
                l.t = ChunkTree{};
            }
        });
    }
    chunk_top_pass(ref_nodes, node_count, order, info, out.nodes, 0u);
    out.root = info[0].ref;
    out.depth = info[0].depth;
    if (out.root == kChunkNone || out.nodes.size() >= (1u << 30)) return false;
    return out.depth + 1u <= stack_limit;
}


// Structural invariants of a ChunkTree against the triangles it was made from -- what k_trace_chunk relies on without
// checking: every valid slot in exactly one chunk, ranks and positions consistent, references in range, no node reached
// twice, depth within the stack, and per child slot either an unbounded margin or (a) a box that contains every triangle
// below it and (b) a determinant-floor bound no smaller than any of theirs.  `why` names the first violation.
bool chunk_tree_check(const ChunkTree& t, const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      uint32_t stack_limit, std::string& why) {
    const size_t n = t.pos_slot.size();
    if (t.pos_rank.size() != n || t.rank_slot.size() != n) { why = "position / rank arrays differ in length"; return false; }
    std::vector<uint8_t> seen_rank(n, 0), seen_pos(n, 0), seen_node(t.nodes.size(), 0);
    for (size_t p = 0; p < n; ++p) {
        const uint32_t r = t.pos_rank[p], slot = t.pos_slot[p];
        if (r >= n || seen_rank[r]) { why = "rank " + std::to_string(r) + " missing or used twice"; return false; }
        seen_rank[r] = 1;
        if (t.rank_slot[r] != slot) { why = "rank_slot disagrees with pos_slot at position " + std::to_string(p); return false; }
        if (slot >= index_len || indices[slot] >= tri_count) { why = "position " + std::to_string(p) + " holds an invalid slot"; return false; }
    }
    auto bf16 = [](uint32_t h) { const uint32_t b = h << 16; float f; std::memcpy(&f, &b, 4); return f; };
    struct Range { float mn[3], mx[3]; double cap; uint32_t depth; };
    // post-order over the tree: the ranges of a node's two children are checked against its slots
    std::function<bool(uint32_t, Range&)> visit = [&](uint32_t ref, Range& out) -> bool {
        for (int a = 0; a < 3; ++a) { out.mn[a] = kInf; out.mx[a] = -kInf; }
        out.cap = 0.0;
        out.depth = 0;
        if (ref & kChunkLeaf) {
            const uint32_t first = ref & 0x03FFFFFFu, count = ((ref >> 26) & 31u) + 1u;
            if (count > kChunkTris || static_cast<size_t>(first) + count > n) { why = "chunk [" + std::to_string(first) + ", +" + std::to_string(count) + ") out of range"; return false; }
            for (uint32_t p = first; p < first + count; ++p) {
                if (seen_pos[p]) { why = "position " + std::to_string(p) + " is in two chunks"; return false; }
                seen_pos[p] = 1;
                const rb_gpu_triangle& g = tris[indices[t.pos_slot[p]]];
                double l1 = 0, l2 = 0;
                for (int a = 0; a < 3; ++a) {
                    out.mn[a] = std::min(out.mn[a], std::min(g.v0[a], std::min(g.v1[a], g.v2[a])));
                    out.mx[a] = std::max(out.mx[a], std::max(g.v0[a], std::max(g.v1[a], g.v2[a])));
                    l1 += double(g.v1[a] - g.v0[a]) * double(g.v1[a] - g.v0[a]);
                    l2 += double(g.v2[a] - g.v0[a]) * double(g.v2[a] - g.v0[a]);
                }
                out.cap = std::max(out.cap, chunk_g(l1, l2) * 1e6);
            }
            return true;
        }
        const uint32_t ni = ref & 0x3FFFFFFFu;
        if (ni >= t.nodes.size() || seen_node[ni]) { why = "node " + std::to_string(ni) + " out of range or reached twice"; return false; }
        seen_node[ni] = 1;
        const ChunkNode& c = t.nodes[ni];
        const struct { const float* mn; const float* mx; uint32_t ref, fac; } slot[2] = {{c.lmin, c.lmax, c.lref, c.lfac}, {c.rmin, c.rmax, c.rref, c.rfac}};
        for (const auto& sl : slot) {
            if (sl.ref == kChunkNone) continue;
            Range r;
            if (!visit(sl.ref, r)) return false;
            out.depth = std::max(out.depth, r.depth);
            out.cap = std::max(out.cap, r.cap);
            for (int a = 0; a < 3; ++a) { out.mn[a] = std::min(out.mn[a], r.mn[a]); out.mx[a] = std::max(out.mx[a], r.mx[a]); }
            if (sl.fac == 0x7F807F80u) continue;   // always entered
            for (int a = 0; a < 3; ++a)
                if (!(sl.mn[a] <= r.mn[a] && r.mx[a] <= sl.mx[a])) { why = "a child box of node " + std::to_string(ni) + " does not contain its triangles and may still be culled"; return false; }
            if (!(double(bf16(sl.fac >> 16)) >= r.cap)) { why = "a child of node " + std::to_string(ni) + " understates the determinant-floor bound below it"; return false; }
        }
        out.depth += 1;
        return true;
    };
    Range root;
    if (t.root == kChunkNone || !visit(t.root, root)) { if (why.empty()) why = "no root"; return false; }
    for (size_t p = 0; p < n; ++p)
        if (!seen_pos[p]) { why = "position " + std::to_string(p) + " is in no chunk"; return false; }
    if (root.depth != t.depth) { why = "depth " + std::to_string(t.depth) + " recorded, " + std::to_string(root.depth) + " found"; return false; }
    if (t.depth + 1u > stack_limit) { why = "deeper than the stack"; return false; }
    return true;
}

}  // namespace rb
