// rb_internal.hpp -- declarations shared by the runtime (rb_runtime.cpp), the
// BVH helper (rb_bvh.cpp) and the kernels (rb_kernels.hip).  Not part of the ABI.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rb_abi.h"

#if defined(__HIPCC__) || defined(__HIP__)
#define RB_HD __host__ __device__
#else
#define RB_HD
#endif

namespace rb {

constexpr uint32_t kStackDepth = 32;      // per-lane traversal stack entries (LDS)
// The traversal stacks of a block are ONE region of LDS, s_stack[entry][thread]: 4-byte entries in columns of
// KParams::stack_depth entries, sized by launch_render.  Every walk a kernel runs -- the caller's tree, the chunked or the
// library's tree, and the sphere tree inside segment_finish -- uses the calling lane's column, one walk at a time.  An entry
// wider than kStackEntryBytes, or a walk deeper than stack_depth, runs into the neighbouring lane's column (r03: an
// experiment with 8-byte entries in k_trace_chunk did, and a lane popped a corrupted reference: a GPU memory fault).  The
// kernels assert the entry size where they take their column (StackColumn), the runtime the depth where it sets
// stack_depth (stack_depth_covers).
constexpr uint32_t kStackEntryBytes = 4;
constexpr uint32_t kDefaultStripeRows = 8;   // = the kernel tile's 8 rows: a tile's rays stay neighbours, the shards stay balanced (profiles/r04_shard_rehearsal.txt)

// ---- rb_bvh.cpp
void bvh_build(const rb_gpu_triangle* tris, size_t n_tris, std::vector<rb_bvh_node>& nodes,
               std::vector<uint32_t>& indices);
bool bvh_validate(const rb_bvh_node* nodes, uint32_t node_count, uint32_t max_stack, std::string& why,
                  uint32_t* depth_out);

// The two-box node of the library's own TRIANGLE tree (leaves of <= 2: 0x80000000 | (count-1) << 28 | first).
struct alignas(16) SphereNode {
    float lmin[3];
    uint32_t left;    // child reference: node index, or leaf = 0x80000000 | (count-1) << 28 | first
    float lmax[3];
    uint32_t right;
    float rmin[3];
    uint32_t _pad0;
    float rmax[3];
    uint32_t _pad1;
};
static_assert(sizeof(SphereNode) == 64, "SphereNode is 64 B");
// ---- Sphere acceleration structure (no reference counterpart: shader.wgsl:574-586 scans; BASELINE C4 has 10^6 spheres).
// A 4-wide tree: every node holds the boxes of up to four children (two levels of median splits collapsed into one), so a
// ray's walk makes half as many DEPENDENT node fetches as with two-box nodes -- the walk waits on them (r04: 48 % of wave
// cycles with two-box nodes).  128 B, component-major, so the four box tests read like one.  Leaves are runs of at most
// kSphLeaf spheres of the leaf order: consecutive 16-byte {centre, radius} records (sph_leaf) with their original indices
// beside them (sph_id), tested one sphere per lane by k_trace_sph.
#ifndef RB_SPH_LEAF
#define RB_SPH_LEAF 16
#endif
constexpr uint32_t kSphLeaf = RB_SPH_LEAF;   // spheres per leaf = lanes per (ray, leaf) pair of k_trace_sph: 8 or 16
static_assert(kSphLeaf == 8 || kSphLeaf == 16, "a sphere leaf is tested by 8 or 16 lanes");
constexpr uint32_t kSphNone = 0xFFFFFFFFu;
struct alignas(16) SphereNode4 {
    float cx[4], cy[4], cz[4], hx[4], hy[4], hz[4];   // child k's box: centre (cx[k], cy[k], cz[k]) -+ half extents (hx[k], hy[k], hz[k]), rounded outwards
    uint32_t ref[4];   // child reference: kSphNone | node index | leaf = 0x80000000 | (count-1) << 27 | first (count <= kSphLeaf)
    uint32_t axes;     // children 0,1 = the lower half of the split along axis (axes & 3), cut along (axes >> 2) & 3; children 2,3 = the
    uint32_t _pad[3];  // upper half, cut along (axes >> 4) & 3: a ray visits the halves, and the children of a half, in the order of its direction's signs
    // lo .. hi -> centre and half extents that contain it whatever the rounding (the slab test then costs 6 operations per axis instead of 8)
    RB_HD void set_box(int k, const float lo[3], const float hi[3]) {
        float* c[3] = {cx, cy, cz};
        float* h[3] = {hx, hy, hz};
        for (int a = 0; a < 3; ++a) {
            const float m = 0.5f * lo[a] + 0.5f * hi[a];
            const float d0 = m - lo[a], d1 = hi[a] - m, d = d0 > d1 ? d0 : d1;
            c[a][k] = m;
            h[a][k] = d >= 0.0f ? d * 1.0000004f + 1e-37f : 0.0f;   // (d rounded up: one ulp for the subtraction, one for this product)
        }
    }
};
static_assert(sizeof(SphereNode4) == 128, "SphereNode4 is 128 B");
constexpr uint32_t kSphereBvhThreshold = 64;  // use the linear two-pass scan up to this many spheres
constexpr uint32_t kSphereDeviceBuildMin = 1024;  // from here up the sphere tree is built on the device by default
// host builder (rb_bvh.cpp): median splits on the longest axis of the centres.  depth = 4-wide levels.
void sphere_bvh_build(const rb_sphere* spheres, size_t n, std::vector<SphereNode4>& nodes,
                      std::vector<uint32_t>& order, uint32_t* root_ref, uint32_t* depth);
// The same tree made on the device (rb_build.hip): level by level, every level one segmented sort of the centres along each
// segment's longest axis -- the median splits of the host builder, a complete tree.  `spheres` is the device copy; nodes_out
// holds sphere_tree_node_capacity(n) nodes, leaf_out n float4 {centre, radius}, id_out n original indices.  Synchronises `stream`.
struct DeviceSphereTreeInfo {
    uint32_t root, depth, n_nodes;
};
size_t sphere_tree_node_capacity(size_t n);
int device_sphere_bvh_build(const rb_sphere* spheres, uint32_t n, SphereNode4* nodes_out, float* leaf_out, uint32_t* id_out,
                            DeviceSphereTreeInfo* info_out, void* stream);
// a node leaves at most three children waiting, so 3 entries per level is all a walk can hold at once (not one more: for
// C4's eight levels the 24-entry columns plus k_trace_sph's 15 KiB of wave corners are 39.4 KiB per block, four blocks per
// CU; a 25th entry would make it three)
inline uint32_t sphere_stack_entries(uint32_t levels4) { return 3u * levels4; }

// Walk of multi-node meshes when the caller's flags do not say: the chunked walk (ChunkTree below),
// at every size.  Should its tree not be buildable (a tree deeper than the LDS stack, a leaf root), the library's own
// tree of section 4.1 takes over from this many triangles up -- where the reference walk's ~450 triangle fetches per
// segment go to HBM -- and the reference walk below.  RB_FLAG_FAST_BVH / RB_FLAG_REFERENCE_WALK / RB_FLAG_CHUNK_WALK choose explicitly.
constexpr uint32_t kOwnTreeDefaultMinTriangles = 393216;
constexpr uint32_t kDeviceBuildMinTriangles = 16384;  // from here up the library's tree is built on the device by default

// c0 of the library's triangle walk (rb_device_intersect.hpp, FastWalk): hits with
// |cos(ray, triangle normal)| >= c0 are covered by the culling margin of the library's tree (FA in each node's two
// pad words), the others by a second pass over the reference tree guided by per-node normal cones.
constexpr float kFastGrazeCos = 0.03f;
// A triangle whose determinant bound L^2 / 1e-6 (the reference rejects |a^| < 1e-6) is at most this is "small": the
// culling margin of the library's tree then covers ALL its accepted hits, near-degenerate ones included, and the
// second pass skips it.  16000 <=> L <= 0.126: a margin of at most 2.6e-2 of the distance to the box.  A tuning
// constant, not part of the proof (any value up to 1.5e5 is covered): larger = fewer triangles left to the second
// pass but wider margins in the first (with RB_FLAG_SKIP_NEAR_DEGENERATE there is no second pass to relieve and the
// runtime passes 0: every triangle keeps the class (A) margin).  Measured (profiles/r02_small_cap_sweep.txt, M segments/s at 1500 / 8000 /
// 16000 / 32000 / 64000 / 150000): C5 744 / 1603 / 1600 / 1608 / 1604 / 1601, reference lamp scene 669 / 750 / 1066 /
// 1065 / 1102 / 909, C3 448 / 450 / 460 / 244 / 181 / 216.
#ifndef RB_FAST_SMALL_CAP
#define RB_FAST_SMALL_CAP 16000.0f
#endif
constexpr float kFastSmallCap = RB_FAST_SMALL_CAP;
constexpr uint32_t kSlotLarge = 0x80000000u;   // slot_meta[2 * slot + 1]: bit 31 = "large" triangle, low bits = reference rank
// A node of the reference tree as that second pass reads it: the caller's box and links bit for bit (the pass
// repeats the reference's own slab test), plus the cone {axis cos(alpha), tan(alpha)} of the triangle normals below.
struct alignas(16) GrazeNode {
    float bmin[3];
    uint32_t left;
    float bmax[3];
    uint32_t right;
    float cone[4];
    uint32_t first, count;   // leaf: range in FastTree::gslots (the leaf's LARGE triangles only)
    float cap;               // largest L^2 / 1e-6 over the large triangles below (+inf beyond 1.5e5): bounds how far from the
                             // node's box a near-degenerate hit can be reported, so the pass can cull on the best t
    uint32_t is_leaf;
};
static_assert(sizeof(GrazeNode) == 64, "GrazeNode is 64 B");

struct TriBound {
    double n[3] = {0, 0, 0};   // unit normal (if has_normal)
    float f = 0.0f;            // F_k: bound of L^2 / |a^| over the hits the library's tree answers for
    bool large = false, has_normal = false;
};
TriBound tri_bound(const rb_gpu_triangle& t, float small_cap = kFastSmallCap);

// The library's own triangle tree (rb_bvh.cpp).  Same 64-B two-box node as the sphere tree.
struct FastTree {
    std::vector<SphereNode> nodes;
    std::vector<GrazeNode> gnodes;     // per REFERENCE node: its box, links and the cone of the normals below it
    std::vector<uint32_t> gslots;      // the large triangles' slots, leaf by leaf (GrazeNode::first / count)
    uint32_t n_large = 0;              // triangles the second pass answers for ("large", rb_bvh.cpp)
    std::vector<uint32_t> slots;       // leaf order -> slot in bvh_indices order
    std::vector<uint32_t> slot_meta;   // per slot: {reference leaf node, rank in the reference visit order}
    std::vector<uint32_t> ref_parent;  // reference tree: parent of each node (root: 0)
    uint32_t root = 0x80000000u, depth = 0;
    float margin = 0.0f;               // slab-rounding part of the box inflation
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};  // mesh bounds (for S, the per-ray distance bound)
    float root_amax = 0.0f;
    float small_cap = kFastSmallCap;   // the "small triangle" threshold the per-triangle bounds were made with
};
bool fast_bvh_build(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                    const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, FastTree& out,
                    float small_cap = kFastSmallCap);
// The part of it that depends on the reference tree only: ref_parent, slot_meta, and `slots` = the
// valid slots in the reference's visit order (the items a builder then arranges into a tree).
bool fast_bvh_prepare(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      const rb_bvh_node* ref_nodes, uint32_t node_count, FastTree& out, float small_cap = kFastSmallCap);

// ---- The chunked walk (k_trace_chunk): the caller's tree on top, walked with the reference's
// own slab arithmetic on the reference's boxes (so a leaf is reached exactly when shader.wgsl:309-389 reaches it),
// nearer child first, and below every reference leaf the library's own small tree down to chunks of at most
// kChunkTris triangles, which a wavefront then tests cooperatively (one triangle per lane, coalesced records).
// A subtree is skipped only when the ray misses its box inflated by the margin that bounds how far from its
// triangle the reference's f32 Moller-Trumbore can report a hit (rb_device_intersect.hpp, FastWalk::entry), or
// enters it beyond the best t.  One 96-B node decides both children:
struct alignas(16) ChunkNode {
    float lmin[3];
    uint32_t lref;   // child reference: kChunkNone | leaf = bit 31, (count - 1) << 26, first position | node index, bit 30 = that node's children carry reference boxes
    float lmax[3];
    uint32_t rref;
    float rmin[3];
    uint32_t lfac;   // two bf16, rounded up: high = largest L^2 / 1e-6 below the child (bounds EVERY accepted hit), low = largest
    float rmax[3];   //   (L^2 / N) / (0.95 c0) (bounds the hits with |cos(ray, normal)| >= c0); 0x7F80 = +inf = always enter
    uint32_t rfac;
    float lcone[4];  // {axis cos(alpha), tan(alpha)} of the normals below the child (GrazeNode::cone): which of the two applies
    float rcone[4];
};
static_assert(sizeof(ChunkNode) == 96, "ChunkNode is 96 B");
#ifndef RB_CHUNK_TRIS
#define RB_CHUNK_TRIS 16
#endif
constexpr uint32_t kChunkTris = RB_CHUNK_TRIS;   // triangles per chunk = lanes per (ray, chunk) unit: 8, 16 or 32
static_assert(kChunkTris == 8 || kChunkTris == 16 || kChunkTris == 32, "a chunk is tested by 8, 16 or 32 lanes");
constexpr uint32_t kChunkNone = 0xFFFFFFFFu;
constexpr uint32_t kChunkLeaf = 0x80000000u, kChunkExact = 0x40000000u;
struct ChunkTree {
    std::vector<ChunkNode> nodes;
    std::vector<uint32_t> pos_slot;    // chunk order -> slot in bvh_indices order
    std::vector<uint32_t> pos_rank;    // chunk order -> rank in the reference's visit order (ties in t go to the lower rank)
    std::vector<uint32_t> rank_slot;   // rank -> slot (the winner's record for shading)
    uint32_t root = kChunkNone, depth = 0;
};
bool chunk_tree_build(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, ChunkTree& out);
bool chunk_tree_check(const ChunkTree& t, const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices, uint32_t index_len,
                      uint32_t stack_limit, std::string& why);   // the structural invariants k_trace_chunk relies on

// ---- rb_build.hip: the chunked walk's tree built on the device: one thread block per reference leaf for the library's own
// levels, the caller's internal nodes by the host's bottom-up pass.  Returns 0 (built: the device arrays are the caller's
// to hipFree), -1 (not for this builder: a leaf beyond kChunkDeviceLeafMax triangles, or a tree the chunked walk does not
// take at all -- the host builder decides), or a hipError_t.
constexpr uint32_t kChunkDeviceLeafMax = 256;       // bvh.rs:12 MAX_LEAF_SIZE = 128
constexpr uint32_t kChunkDeviceBuildMin = 16384;    // slots from which the device builder is the default (below, the host's is microseconds)
struct DeviceChunkTree {
    ChunkNode* nodes = nullptr;
    size_t n_nodes = 0, nodes_capacity = 0;
    uint32_t *pos_slot = nullptr, *pos_rank = nullptr, *rank_slot = nullptr;
    size_t n_pos = 0;
    uint32_t root = kChunkNone, depth = 0;
};
int device_chunk_tree_build(const rb_gpu_triangle* d_tris, uint32_t tri_count, const uint32_t* d_indices, uint32_t index_len,
                            const rb_bvh_node* ref_nodes, uint32_t node_count, uint32_t stack_limit, DeviceChunkTree* out, void* stream);

// ---- rb_build.hip: the same tree built on the device (RB_FLAG_DEVICE_BVH), Morton order + LBVH
struct DeviceTreeInfo {
    uint32_t root, depth;
    float margin, root_amax;
    float bmin[3], bmax[3];
};
int device_fast_bvh_build(const rb_gpu_triangle* tris, const uint32_t* indices, const uint32_t* slots, uint32_t n,
                          const uint32_t* slot_meta, SphereNode* nodes_out, uint32_t* fast_slots_out, DeviceTreeInfo* info_out,
                          void* stream, bool plain_lbvh);

// ---- device-side counters (one block of u64 in device memory)
enum Counter : uint32_t {
    C_SEGMENTS = 0, C_PATHS, C_NODES, C_TRIS, C_SPHERES, C_LIGHTS, C_MESH_HITS, C_COUNT
};

// Triangle in bvh_indices order with the per-triangle invariants of
// shader.wgsl:249-250,351 hoisted out of the per-ray loop.  edge1/edge2/normal
// are computed on the device by the same f32 operations the shader performs per
// test, so every ray sees bit-identical values.
struct alignas(16) PrepTri {   // 64 B: one s_load_dwordx16 (uniform) or four 16-B vector loads
    float v0[3];
    uint32_t tri_id;    // index into bvh_triangles (for hash_to_color and the :336 guard)
    float e1[3];        // v1 - v0
    uint32_t mesh_index;
    float e2[3];        // v2 - v0
    uint32_t valid;     // 0 when a guard of shader.wgsl:331,336 would `continue`
    float n[3];         // normalize(cross(e1, e2))  (shader.wgsl:351)
    uint32_t _pad;
};
struct alignas(16) PrepTriShade {   // 16 B: uv lookup indices (shader.wgsl:357-359)
    uint32_t v0_index;
    uint32_t v1_index;
    uint32_t v2_index;
    uint32_t _pad;
};

// The per-launch invariants of the sample loop (shader.wgsl:690,702-708), worked out once on the host by
// launch_render (host_cam in rb_kernels.hip: the same single IEEE binary32 operations, in the shader's order) so
// that no kernel holds twenty registers for them: a path start reads them from the kernel arguments.
struct Cam {
    float pos[3], right[3], up[3], fwd[3];
    float fov, aspect, wm1, hm1;
    float inv_wm1, inv_hm1;  // RN(1 / wm1), RN(1 / hm1) when fast_wh
    uint32_t fast_wh;        // both denominators inside the ranges the exact-reciprocal division was checked for
    uint32_t _pad;
};

// Everything a render launch needs.  Passed by value (kernarg segment => scalar loads).
// the stream kernels' queue words: one per group of blocks that share an XCD, 64 bytes apart
constexpr uint32_t kQueueGroups = 8, kQueueStride = 16, kQueueWords = kQueueGroups * kQueueStride;
struct KParams {
    rb_uniforms u;  // counts already patched (gpu_wrapper.rs:475-495) and clamped to buffer lengths
    const rb_sphere* spheres;
    const rb_point_light* lights;
    const rb_mesh* meshes;
    const rb_bvh_node* nodes;
    const uint32_t* indices;
    const rb_gpu_triangle* tris;
    const PrepTri* ptris;          // index_len entries, bvh_indices order
    const PrepTriShade* pshade;    // index_len entries
    const float* uvs;
    const uint32_t* tex_data;
    const rb_texture_info* tex_info;
    const float* srgb_lut;         // 256 entries: powf(i/255, 2.2) computed on the host
    const SphereNode* fast_nodes;  // fast triangle tree (nullptr => the reference walk)
    const GrazeNode* gnodes;       // per reference node: box, links, normal cone (the walk's second pass)
    const uint32_t* gslots;        // slots of the large triangles, leaf by leaf
    uint32_t fast_skip_second_pass; // RB_FLAG_SKIP_NEAR_DEGENERATE
    const float* fast_tris;        // PrepTri records gathered into fast-leaf order (64 B each)
    const uint32_t* fast_slots;    // fast-leaf order -> slot
    const uint32_t* slot_meta;     // per slot {reference leaf node, reference rank}
    const uint32_t* ref_parent;    // reference tree parents
    uint32_t fast_root;
    float fast_margin;
    float fast_bmin[3];
    float fast_root_amax;
    float fast_bmax[3];
    uint32_t _pad_fast;
    const ChunkNode* chunk_nodes;  // the chunked walk's tree (nullptr => another walk)
    const float* chunk_a;          // float4 per chunk-order position: v0, rank (as bits)
    const float* chunk_b;          // float4: e1, -
    const float* chunk_c;          // float4: e2, -
    const uint32_t* chunk_rank_slot; // rank -> slot
    uint32_t chunk_root;           // root child reference (the root's own box is nodes[0]'s)
    uint32_t chunk_n;              // positions
    const SphereNode4* sph_nodes;  // sphere tree (nullptr => linear scan)
    const float* sph_leaf;         // float4 {centre, radius} in leaf order
    const uint32_t* sph_id;        // original sphere index per leaf-order slot
    uint32_t sph_root;             // root child reference
    uint32_t _pad_sph;
    const float* accum_in;         // local_rows_padded * width * 4: the accumulation this launch resumes ...
    float* accum_out;              // ... and the one it writes (the same buffer, or the other frame slot when a pass runs ahead)
    uint32_t* out_rgba;            // local_rows_padded * width, x mirrored
    unsigned long long* counters;  // C_COUNT
    uint32_t* queue;               // work-queue head (RB_KERNEL_QUEUE / RB_KERNEL_STREAM)
    float* colors;                 // RB_KERNEL_STREAM: per-path radiance as float4, [tile][sample][64 pixels]
    uint32_t n_lights;             // arrayLength(&point_lights): >= 1 unless deleted
    uint32_t n_meshes;
    uint32_t index_len;            // arrayLength(&bvh_indices)
    uint32_t n_uvs;
    uint32_t n_tex;
    uint32_t first_pass, n_passes, samples_per_pass;
    uint32_t shard_rank, shard_count, stripe_rows, local_rows;  // local_rows: rows owned (unpadded)
    uint32_t stack_depth;          // LDS traversal-stack entries per lane (0 for single-node trees)
    uint32_t blocks_per_cu;        // persistent grid density (0 = default)
    uint32_t queue_batch;          // items a wave reserves per global atomic (RB_KERNEL_STREAM)
    uint32_t queue_groups;         // 1, or 8: one queue word per group of blocks that share an XCD (blockIdx % 8), each with its own stripes of items (set by launch_render)
    uint32_t queue_region;         // items per stripe (a multiple of queue_batch); stripe k belongs to band k % queue_groups
    uint32_t no_leaf_stepping;     // RB_KERNEL_STREAM: 1 = per-segment traversal even for multi-node trees
    uint32_t lds_mode;             // LDS staging of small meshes: 0 = when it fits, 1 = never
    uint32_t magic_S, magic_tiles_x; // floor(2^32 / d) for the item decode of the stream kernels (set by launch_render)
    uint32_t* stack_overflow;      // fast walk: entries beyond kStackDepth, [entry][grid * block] (nullptr if never needed)
    Cam cam;                       // set by launch_render
};

struct LaunchInfo {
    uint32_t grid, block;
    size_t lds_bytes;
    const char* kernel_name;
};

// ---- rb_kernels.hip
// All return hipError_t as int (0 = success); launches are asynchronous on `stream`.
int launch_render(const KParams& p, uint32_t kernel, bool stats, void* stream, LaunchInfo* info,
                  void* ev_after_trace = nullptr);
int launch_prep_tris(const rb_gpu_triangle* tris, uint32_t tri_count, const uint32_t* indices,
                     uint32_t index_len, PrepTri* out, PrepTriShade* shade, void* stream);
int launch_prep_materials(void* first_material, uint32_t stride, uint32_t n, void* stream);
int launch_gather_tris(const PrepTri* ptris, const uint32_t* slots, uint32_t n, PrepTri* out, void* stream);
// prepared triangles -> the chunked walk's three float4 arrays in chunk order (A: v0 | rank, B: e1, C: e2)
int launch_chunk_gather(const PrepTri* ptris, const uint32_t* pos_slot, const uint32_t* pos_rank, uint32_t n, float* a,
                        float* b, float* c, void* stream);
// Multi-GPU assembly on the root device: gathered[rank][local row][x] (every rank's padded stripe buffer back to
// back) -> frame[global row][x]
int launch_deinterleave(const uint32_t* gathered, uint32_t* frame, uint32_t width, uint32_t height, uint32_t padded_rows,
                        uint32_t stripe_rows, uint32_t shard_count, void* stream);
size_t max_dynamic_lds(int device);   // hipDeviceAttributeMaxSharedMemoryPerBlock
int launch_div_exhaustive(uint32_t b_begin, uint32_t b_count, uint32_t ea, uint32_t eb, uint32_t a_begin,
                          uint32_t a_count, unsigned long long* mismatch16, void* stream);
int launch_rcp_exhaustive(uint32_t expo, uint32_t* mismatch16, void* stream);
int launch_debug_math(const float* a, const float* b, float* out, uint32_t n, void* stream);
int debug_walk_profile(unsigned long long* out64, int reset);   // pass occupancy counters of a profiling build, tools/ablate/rb_profile.patch (-1 otherwise)
double measure_l1_gather(size_t table_bytes, uint32_t rounds);   // divergent 16-byte gathers from an L2-resident table: lane accesses / s
int device_cu_count(int device);
uint32_t stream_kernel_max_threads(uint32_t blocks_per_cu);  // upper bound of grid * block of the stream kernels

}  // namespace rb
