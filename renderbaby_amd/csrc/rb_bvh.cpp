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

}  // namespace rb
