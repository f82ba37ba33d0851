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

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace rb {

namespace {

constexpr size_t kMaxLeaf = 128;  // bvh.rs:12

struct Builder {
    const rb_gpu_triangle* tris;
    std::vector<uint32_t>& idx;
    std::vector<rb_bvh_node>& nodes;

    static float centroid(const rb_gpu_triangle& t, int axis) {
        return ((t.v0[axis] + t.v1[axis]) + t.v2[axis]) / 3.0f;  // bvh.rs:152-154
    }

    uint32_t node(size_t first, size_t count) {
        const uint32_t me = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        float mn[3], mx[3];
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::numeric_limits<float>::infinity();
            mx[a] = -std::numeric_limits<float>::infinity();
        }
        for (size_t i = first; i < first + count; ++i) {
            const rb_gpu_triangle& t = tris[idx[i]];
            const float* vs[3] = {t.v0, t.v1, t.v2};
            for (auto v : vs)
                for (int a = 0; a < 3; ++a) {
                    mn[a] = std::min(mn[a], v[a]);
                    mx[a] = std::max(mx[a], v[a]);
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
            return me;
        }
        const float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        const int axis = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);  // bvh.rs:125-135
        const size_t mid = first + count / 2;
        std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count,
                         [&](uint32_t a, uint32_t b) { return centroid(tris[a], axis) < centroid(tris[b], axis); });
        const uint32_t l = node(first, mid - first);
        const uint32_t r = node(mid, first + count - mid);
        n.left = l;
        n.right = r;
        nodes[me] = n;
        return me;
    }
};

}  // namespace

void bvh_build(const rb_gpu_triangle* tris, size_t n_tris, std::vector<rb_bvh_node>& nodes,
               std::vector<uint32_t>& indices) {
    nodes.clear();
    indices.resize(n_tris);
    for (size_t i = 0; i < n_tris; ++i) indices[i] = static_cast<uint32_t>(i);
    if (n_tris == 0) return;  // the adapter never builds an empty tree (scene_engine_adapter.rs:435-440)
    nodes.reserve(2 * (n_tris / (kMaxLeaf / 2) + 1));
    Builder b{tris, indices, nodes};
    b.node(0, n_tris);
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


// ---------------------------------------------------------------- sphere BVH --
// The reference scans every sphere on every segment (shader.wgsl:574-586), which is
// intractable beyond a few thousand spheres (BASELINE config C4 has 10^6).  This builds
// a binary tree over the spheres' tight boxes (median split on the longest axis of the
// centroid bounds, <= 4 spheres per leaf).  A node stores BOTH children's boxes, so one
// visit decides both; the traversal (rb_kernels.hip, intersect_spheres_bvh) inflates the
// boxes per ray by a margin that covers the rounding error of the reference's own
// discriminant, runs the reference's exact intersect_sphere on every candidate and
// breaks ties by the lower original index -- the winner of the linear scan.
namespace {
struct SBuilder {
    const rb_sphere* sph;
    std::vector<uint32_t>& order;
    std::vector<SphereNode>& nodes;
    uint32_t max_depth = 0;

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
    // returns a child reference: leaf = 0x80000000 | (count-1) << 28 | first ; node = index
    uint32_t build(size_t first, size_t count, uint32_t depth) {
        max_depth = std::max(max_depth, depth);
        if (count <= 4) return 0x80000000u | (static_cast<uint32_t>(count - 1) << 28) | static_cast<uint32_t>(first);
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
        const size_t mid = first + count / 2;
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                         [&](uint32_t a, uint32_t b) { return sph[a].center[axis] < sph[b].center[axis]; });
        const uint32_t me = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        SphereNode n;
        bounds(first, mid - first, n.lmin, n.lmax);
        bounds(mid, first + count - mid, n.rmin, n.rmax);
        n.left = build(first, mid - first, depth + 1);
        n.right = build(mid, first + count - mid, depth + 1);
        n._pad0 = n._pad1 = 0;
        nodes[me] = n;
        return me;
    }
};
}  // namespace

void sphere_bvh_build(const rb_sphere* spheres, size_t n, std::vector<SphereNode>& nodes,
                      std::vector<uint32_t>& order, uint32_t* root_ref, uint32_t* depth, float bmin[3],
                      float bmax[3]) {
    nodes.clear();
    order.resize(n);
    for (size_t i = 0; i < n; ++i) order[i] = static_cast<uint32_t>(i);
    SBuilder b{spheres, order, nodes};
    b.bounds(0, n, bmin, bmax);
    *root_ref = n ? b.build(0, n, 1) : 0x80000000u;
    *depth = b.max_depth + 1;
}

}  // namespace rb
