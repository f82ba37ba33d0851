// rb_runtime.cpp -- the C-ABI runtime of librenderbaby_hip.so (include/rb_abi.h).
//
// Replaces, for the HIP backend, crates/engine-wgpu-wrapper (GpuWrapper,
// GpuBuffers, ProgressiveRenderHelper) and the host half of
// crates/engine-pathtracer/src/lib.rs: device buffers mirroring the 14 wgpu
// buffers (buffers.rs:32-61), the Change<T> state machine
// (gpu_wrapper.rs:116-300), count patch-up and uploads (:469-576), the pass loop
// (:365-426) and read-back (:432-463; the x mirror is done by the kernel's
// store).  Every entry point selects its device first (HIP's current device is
// per-thread and the reference drives the iterator from a worker thread,
// frame_buffer.rs:141-148) and reports failures as status + message instead of
// panicking.
//
// Beyond the reference (one wgpu device, one synchronous pass per frame):
//  * two frame slots (accumulation + RGBA8 each) so that the progressive iterator can run pass k+1
//    while frame k is copied to the caller;
//  * row-stripe sharding over several devices behind this same boundary -- one engine per device inside
//    one process (rb_create_multi) or one process per device (rb_comm_init_rank) -- with ONE RCCL gather
//    of the RGBA8 stripes to the root per delivered frame; rccl_gather.cpp.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rb_internal.hpp"
#include "rb_rccl.hpp"

namespace {

thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T* ptr = nullptr;
    size_t count = 0;     // elements allocated
    ~DevBuf() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    hipError_t resize(size_t n) {
        if (n == count && ptr) return hipSuccess;
        release();
        if (n == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
    void adopt(T* p, size_t n) {   // take over an allocation made elsewhere (rb_build.hip)
        release();
        ptr = p;
        count = n;
    }
    // scratch that is sized per launch: keep an allocation that is large enough and not wastefully so
    hipError_t reserve(size_t n) {
        if (ptr && n <= count && count <= 4 * std::max<size_t>(n, 1)) return hipSuccess;
        return resize(n);
    }
};

// One frame: the accumulation (vec4<f32> per pixel: sum of radiance, sample count) and the packed RGBA8
// image the kernels derive from it.  `done` is recorded after the launches that produced this slot.
struct FrameSlot {
    DevBuf<float> accum;
    DevBuf<uint32_t> rgba;
    hipEvent_t done = nullptr;
};

}  // namespace

struct rb_engine {
    std::mutex mu;
    mutable std::mutex err_mu;       // guards `error` for the const getters (rb_get_size, rb_last_error)
    mutable std::string error;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // read-backs into page-locked caller memory
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;   // per launch chunk: begin, after-trace, end
    uint32_t ev_used = 0;
    const char* last_kernel_name = "";
    rb_options opt{};

    bool initialized = false;        // GpuWrapper::initialized (gpu_wrapper.rs:69,117)
    bool have_uniforms = false;      // last update carried Create/Update uniforms (:303-329)
    bool scene_valid = false;        // the buffers hold a scene that passed validate_scene (set by the last update)
    rb_uniforms uniforms{};          // as handed over (before count patch-up)
    rb_progressive prh{};            // gpu_wrapper.rs:19-53
    bool iter_initialized = false;   // RaytracerFrameIterator::initialized (lib.rs:131)
    uint32_t iter_passes_per_frame = 1;  // rb_iter_set_passes_per_frame (1 = the reference's one frame per pass)

    // element counts = what arrayLength() / the patched uniforms see
    uint32_t n_spheres = 0, n_lights = 0, n_meshes = 0, n_nodes = 0, n_indices = 0, n_tris = 0, n_uvs = 0,
             n_tex = 0;
    // Change of the last update for the three patched counts (gpu_wrapper.rs:475-495)
    uint32_t last_change_spheres = RB_KEEP, last_change_nodes = RB_KEEP, last_change_tris = RB_KEEP;
    bool prep_dirty = true;
    uint32_t prep_tri_count = 0xFFFFFFFFu;  // uniforms.bvh_triangle_count (patched) the prepared triangles were made for

    DevBuf<rb_sphere> spheres;
    DevBuf<rb_point_light> lights;
    DevBuf<rb_mesh> meshes;
    DevBuf<rb_bvh_node> nodes;
    DevBuf<uint32_t> indices;
    DevBuf<rb_gpu_triangle> tris;
    DevBuf<rb::PrepTri> ptris;
    DevBuf<rb::PrepTriShade> pshade;
    DevBuf<float> uvs;
    DevBuf<uint32_t> tex_data;
    DevBuf<rb_texture_info> tex_info;
    DevBuf<float> srgb_lut;
    FrameSlot slot[2];               // slot[cur] holds the committed frame
    int cur = 0;
    bool spec_valid = false;         // slot[1 - cur] holds passes [spec_first, +spec_n) run ahead on top of slot[cur]
    uint32_t spec_first = 0, spec_n = 0;
    DevBuf<unsigned long long> counters;
    DevBuf<uint32_t> queue;
    DevBuf<rb::SphereNode> fast_nodes; // the library's own triangle tree (walk mode "fast")
    DevBuf<rb::PrepTri> fast_tris;
    DevBuf<uint32_t> fast_slots, slot_meta, ref_parent, stack_overflow;
    DevBuf<rb::GrazeNode> gnodes;
    DevBuf<uint32_t> gslots;
    uint32_t fast_root = 0, fast_depth = 0;
    float fast_margin = 0.0f, fast_root_amax = 0.0f;
    float fast_bmin[3] = {0, 0, 0}, fast_bmax[3] = {0, 0, 0};
    bool fast_ready = false;
    float fast_build_ms = 0.0f;
    const char* fast_builder = "";  // which builder produced the fast tree ("host-sah" / "device-ploc" / "device-lbvh")
    DevBuf<rb::ChunkNode> chunk_nodes; // the chunked walk (rb_internal.hpp, ChunkTree)
    DevBuf<float> chunk_a, chunk_b, chunk_c;
    DevBuf<uint32_t> chunk_rank_slot;
    DevBuf<uint32_t> chunk_pos_slot, chunk_pos_rank;   // chunk order -> slot / rank: read by the gather at build time, kept for rb_debug_engine_chunk_tree
    size_t chunk_n_nodes = 0;
    const char* chunk_builder = "";   // "device" | "host"
    uint32_t chunk_root = 0, chunk_depth = 0;
    bool chunk_ready = false;
    float chunk_build_ms = 0.0f;
    // The host's copy of the mesh, for the host builders and the checkers.  A large mesh (>= kChunkDeviceBuildMin elements: the
    // device builder's territory) is NOT copied at rb_update -- a second 88 MB in host memory cost C5's update 10 of its 14 ms --
    // but fetched back from the device buffer if a host builder turns out to be needed after all (ensure_host_mesh).
    std::vector<rb_gpu_triangle> host_tris;
    std::vector<uint32_t> host_indices;
    size_t host_tri_len = 0, host_index_len = 0;   // what the vectors hold, or would hold (0: the engine keeps no copy)
    bool host_tris_stale = false, host_indices_stale = false;
    DevBuf<rb::SphereNode4> sph_nodes;  // own sphere acceleration structure (n_spheres > threshold)
    DevBuf<float> sph_leaf;
    DevBuf<uint32_t> sph_id;
    uint32_t sph_root = 0, sph_depth = 0;
    bool sph_bvh = false;
    bool stack_depth_covers = true;    // set with KParams::stack_depth: every walk in use fits its LDS column
    const char* sph_builder = "";      // "device-median" | "host-median" | "" (linear scan)
    float sph_build_ms = 0.0f;
    DevBuf<float> colors;            // RB_KERNEL_STREAM: float4 per (pixel, sample) of one launch chunk
    uint64_t color_budget = 0;       // bytes `colors` may take (0 = ask the device at the next dispatch)
    uint32_t bvh_stack = 0;          // traversal-stack entries the current tree needs

    std::vector<rb_bvh_node> host_nodes;  // kept for validation when nodes/indices change separately
    uint32_t width = 0, height = 0, local_rows = 0, padded_rows = 0;

    rb_stats stats{};
    float last_dispatch_ms = 0.0f;
    uint32_t last_launches = 0;
    bool timing_pending = false;
    uint32_t max_mesh_index = 0;  // over the uploaded triangles

    // ---- several devices behind one handle (rb_create_multi): this engine only coordinates; every part is a
    // complete engine for one shard on one device.  Or one process per device (rb_comm_init_rank): this engine
    // is shard `opt.shard_rank` and `net` holds its communicator.
    std::vector<std::unique_ptr<rb_engine>> parts;
    rb::Gather net;
};

namespace {

int fail(const rb_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) {
        std::lock_guard<std::mutex> g(e->err_mu);
        e->error = buf;
    } else {
        g_create_error = buf;
    }
    return code;
}

#define HIP_TRY(e, call)                                                                          \
    do {                                                                                          \
        hipError_t _st = (call);                                                                  \
        if (_st != hipSuccess)                                                                    \
            return fail((e), RB_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(_st));      \
    } while (0)

const char* kFieldNames[9] = {"uniforms", "spheres", "uvs", "meshes", "lights",
                              "bvh_nodes", "bvh_indices", "bvh_triangles", "textures"};

const rb_field* field_at(const rb_config* c, int i) {
    const rb_field* f[9] = {&c->uniforms, &c->spheres, &c->uvs, &c->meshes, &c->lights,
                            &c->bvh_nodes, &c->bvh_indices, &c->bvh_triangles, &c->textures};
    return f[i];
}

int check_fields(rb_engine* e, const rb_config* cfg) {
    if (!cfg) return fail(e, RB_ERR_NULL_ARGUMENT, "config is NULL");
    for (int i = 0; i < 9; ++i) {
        const rb_field* f = field_at(cfg, i);
        if (f->change > RB_DELETE) return fail(e, RB_ERR_NULL_ARGUMENT, "%s: bad change tag %u", kFieldNames[i], f->change);
        if ((f->change == RB_CREATE || f->change == RB_UPDATE) && f->count > 0 && !f->ptr)
            return fail(e, RB_ERR_NULL_ARGUMENT, "%s: count %zu with NULL pointer", kFieldNames[i], f->count);
    }
    if ((cfg->uniforms.change == RB_CREATE || cfg->uniforms.change == RB_UPDATE) && cfg->uniforms.count != 1)
        return fail(e, RB_ERR_INVALID_UNIFORMS, "uniforms: expected exactly one rb_uniforms, got %zu", cfg->uniforms.count);
    return RB_OK;
}

// RenderConfig::validate_init -- render_config.rs:163-185
int validate_init(rb_engine* e, const rb_config* c) {
    if (c->uniforms.change != RB_CREATE) return fail(e, RB_ERR_INVALID_UNIFORMS, "Invalid Uniforms");
    if (c->spheres.change != RB_CREATE) return fail(e, RB_ERR_INVALID_SPHERES, "Invalid Spheres");
    if (c->uvs.change != RB_CREATE) return fail(e, RB_ERR_INVALID_UVS, "Invalid UVs");
    if (c->meshes.change != RB_CREATE) return fail(e, RB_ERR_INVALID_MESHES, "Invalid Meshes");
    if (c->lights.change != RB_CREATE) return fail(e, RB_ERR_INVALID_LIGHTS, "Invalid Lights");
    if (c->textures.change != RB_CREATE) return fail(e, RB_ERR_INVALID_TEXTURES, "Invalid Textures");
    return RB_OK;
}

bool has_data(const rb_field& f) { return f.change == RB_CREATE || f.change == RB_UPDATE; }

// RenderConfig::validate -- render_config.rs:187-268
int validate(rb_engine* e, const rb_config* c) {
    if (has_data(c->uniforms)) {
        const rb_uniforms* u = static_cast<const rb_uniforms*>(c->uniforms.ptr);
        if (!(u->camera.pane_distance >= 0.0f && u->camera.pane_distance <= 100.0f))
            return fail(e, RB_ERR_PANE_DISTANCE_OUT_OF_BOUNDS, "Pane-Distance is out of bounds");
        if (!(u->camera.pane_width >= 0.0f && u->camera.pane_width <= 1000.0f))
            return fail(e, RB_ERR_PANE_WIDTH_OUT_OF_BOUNDS, "Pane-Distance is out of bounds");  // sic, :631-633
        const float* d = u->camera.dir;
        const float len_sq = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (len_sq < 1.1920929e-07f) return fail(e, RB_ERR_INVALID_CAMERA_DIRECTION, "Invalid camera direction");
    } else if (c->uniforms.change == RB_DELETE) {
        return fail(e, RB_ERR_CANNOT_DELETE_NONEXISTENT, "Cannot delete none existent");
    }
    if (has_data(c->spheres)) {
        const rb_sphere* s = static_cast<const rb_sphere*>(c->spheres.ptr);
        for (size_t i = 0; i < c->spheres.count; ++i)
            if (s[i].radius <= 0.0f) return fail(e, RB_ERR_INVALID_SPHERES, "Invalid Spheres");
    }
    if (has_data(c->uvs)) {
        if (c->uvs.count % 2 != 0) return fail(e, RB_ERR_INVALID_UVS, "Invalid UVs");
    } else if (c->uvs.change == RB_DELETE) {
        return fail(e, RB_ERR_UNSUPPORTED_DELETE, "not yet implemented: Implement UVs Deletion");
    }
    if (c->meshes.change == RB_DELETE)
        return fail(e, RB_ERR_UNSUPPORTED_DELETE, "not yet implemented: Implement meshes Deletion");
    if (has_data(c->lights)) {
        const rb_point_light* l = static_cast<const rb_point_light*>(c->lights.ptr);
        for (size_t i = 0; i < c->lights.count; ++i)
            if (l[i].radius <= 0.0f) return fail(e, RB_ERR_INVALID_LIGHTS, "Invalid Lights");
    } else if (c->lights.change == RB_DELETE) {
        return fail(e, RB_ERR_UNSUPPORTED_DELETE, "not yet implemented: Implement lights Deletion");
    }
    if (c->textures.change == RB_DELETE)
        return fail(e, RB_ERR_UNSUPPORTED_DELETE, "not yet implemented: Implement textures Deletion");
    return RB_OK;
}

// create_storage_buffer -- buffers.rs:232-249: an empty slice still allocates one
// zero-filled element (wgpu zero-initialises), so arrayLength() is 1.
// The copy is queued on the engine's stream from caller memory: every path that calls this ends in
// update_locked's hipStreamSynchronize (or an earlier one) before the caller gets its buffers back.
template <typename T>
int upload(rb_engine* e, DevBuf<T>& buf, const void* src, size_t count, uint32_t* visible_len, bool pad_empty) {
    const size_t alloc = (count == 0 && pad_empty) ? 1 : count;
    HIP_TRY(e, buf.resize(alloc));
    if (count > 0) {
        HIP_TRY(e, hipMemcpyAsync(buf.ptr, src, count * sizeof(T), hipMemcpyHostToDevice, e->stream));
    } else if (alloc > 0) {
        HIP_TRY(e, hipMemsetAsync(buf.ptr, 0, alloc * sizeof(T), e->stream));
    }
    if (visible_len) *visible_len = static_cast<uint32_t>(alloc);
    return RB_OK;
}

struct StripeGeometry {
    uint32_t local = 0, padded = 0;
};
StripeGeometry stripe_geometry(const rb_options& opt, uint32_t h) {
    const uint32_t sc = opt.shard_count > 1 ? opt.shard_count : 1;
    const uint32_t sr = opt.stripe_rows ? opt.stripe_rows : rb::kDefaultStripeRows;
    StripeGeometry g{h, h};
    if (sc > 1) {
        const uint32_t stripes = (h + sr - 1) / sr;
        const uint32_t per_rank = (stripes + sc - 1) / sc;  // equal on every rank (padded)
        g.padded = per_rank * sr;
        uint32_t owned = 0;  // stripes this rank renders
        for (uint32_t s = opt.shard_rank; s < stripes; s += sc) owned++;
        g.local = owned * sr;  // the kernels additionally bound rows by global y < height
    }
    return g;
}

// grow_resolution -- buffers.rs:171-180 (+ the stripe geometry of the sharded case)
int resize_frame(rb_engine* e, uint32_t w, uint32_t h) {
    const StripeGeometry g = stripe_geometry(e->opt, h);
    const uint64_t px = static_cast<uint64_t>(w) * g.padded;
    if (px >= (1ull << 31)) return fail(e, RB_ERR_INVALID_UNIFORMS, "frame of %u x %u pixels is too large", w, h);
    e->spec_valid = false;
    e->cur = 0;
    e->slot[1].accum.release();   // the run-ahead slot is (re)allocated when the iterator first needs it
    e->slot[1].rgba.release();
    FrameSlot& s = e->slot[0];
    HIP_TRY(e, s.accum.resize(px * 4));
    HIP_TRY(e, s.rgba.resize(px));
    if (px) {
        HIP_TRY(e, hipMemsetAsync(s.accum.ptr, 0, px * 16, e->stream));
        HIP_TRY(e, hipMemsetAsync(s.rgba.ptr, 0, px * 4, e->stream));
    }
    e->width = w;
    e->height = h;
    e->local_rows = g.local;
    e->padded_rows = g.padded;
    return RB_OK;
}

int upload_textures(rb_engine* e, const rb_field& f) {
    const rb_texture* t = static_cast<const rb_texture*>(f.ptr);
    std::vector<uint32_t> data;
    std::vector<rb_texture_info> info;
    uint32_t offset = 0;
    for (size_t i = 0; i < f.count; ++i) {  // process_textures, buffers.rs:151-168
        const size_t n = static_cast<size_t>(t[i].width) * t[i].height;
        info.push_back(rb_texture_info{offset, t[i].width, t[i].height, 0});
        data.insert(data.end(), t[i].rgba_data, t[i].rgba_data + n);
        offset += t[i].width * t[i].height;
    }
    int rc = upload(e, e->tex_data, data.data(), data.size(), nullptr, true);
    if (rc) return rc;
    rc = upload(e, e->tex_info, info.data(), info.size(), nullptr, true);
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));  // `data`/`info` are locals
    e->n_tex = static_cast<uint32_t>(f.count);
    return RB_OK;
}

int prep_materials(rb_engine* e, rb_material* first, size_t stride, size_t n) {
    if (!first || n == 0) return RB_OK;
    int rc = rb::launch_prep_materials(first, static_cast<uint32_t>(stride), static_cast<uint32_t>(n), e->stream);
    if (rc) return fail(e, RB_ERR_DEVICE, "material prep launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
    return RB_OK;
}

// Spheres beyond kSphereBvhThreshold get the library's own acceleration structure; the
// reference's linear scan (shader.wgsl:574-586) stays the rule for small counts.  From kSphereDeviceBuildMin spheres
// up the tree is made on the device from the copy that is already there (rb_build.hip: the same median splits, one
// segmented sort per level; 10^6 spheres in milliseconds where the host takes 0.1 s);
// RB_FLAG_SPHERE_TREE_HOST / RB_FLAG_SPHERE_TREE_DEVICE force either builder.  The frame does not depend on which one ran.
int build_sphere_bvh(rb_engine* e, const rb_sphere* s, size_t n) {
    e->sph_bvh = false;
    e->sph_builder = "";
    if (n <= rb::kSphereBvhThreshold || n >= (1u << 27) || (e->opt.flags & RB_FLAG_NO_SPHERE_BVH)) return RB_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    const bool force_host = (e->opt.flags & RB_FLAG_SPHERE_TREE_HOST) != 0u, force_dev = (e->opt.flags & RB_FLAG_SPHERE_TREE_DEVICE) != 0u;
    if (!force_host && (force_dev || n >= rb::kSphereDeviceBuildMin)) {
        HIP_TRY(e, e->sph_nodes.resize(rb::sphere_tree_node_capacity(n)));
        HIP_TRY(e, e->sph_leaf.resize(n * 4));
        HIP_TRY(e, e->sph_id.resize(n));
        rb::DeviceSphereTreeInfo info{};
        const int rc = rb::device_sphere_bvh_build(e->spheres.ptr, static_cast<uint32_t>(n), e->sph_nodes.ptr, e->sph_leaf.ptr, e->sph_id.ptr,
                                                   &info, e->stream);
        if (rc) return fail(e, RB_ERR_DEVICE, "device sphere tree build failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
        if (rb::sphere_stack_entries(info.depth) > rb::kStackDepth) return RB_OK;   // beyond 16 M spheres: the scan (the host's tree is as deep)
        e->sph_root = info.root;
        e->sph_depth = rb::sphere_stack_entries(info.depth);
        e->sph_bvh = true;
        e->sph_builder = "device-median";
        e->sph_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        return RB_OK;
    }
    std::vector<rb::SphereNode4> nodes;
    std::vector<uint32_t> order;
    uint32_t levels4 = 0;
    rb::sphere_bvh_build(s, n, nodes, order, &e->sph_root, &levels4);
    if (rb::sphere_stack_entries(levels4) > rb::kStackDepth) return RB_OK;
    e->sph_depth = rb::sphere_stack_entries(levels4);
    std::vector<float> leaf(n * 4);
    for (size_t j = 0; j < n; ++j) {
        const rb_sphere& sp = s[order[j]];
        leaf[j * 4 + 0] = sp.center[0];
        leaf[j * 4 + 1] = sp.center[1];
        leaf[j * 4 + 2] = sp.center[2];
        leaf[j * 4 + 3] = sp.radius;
    }
    int rc = upload(e, e->sph_nodes, nodes.data(), nodes.size(), nullptr, true);
    if (!rc) rc = upload(e, e->sph_leaf, leaf.data(), leaf.size(), nullptr, true);
    if (!rc) rc = upload(e, e->sph_id, order.data(), order.size(), nullptr, true);
    if (rc) return rc;
    HIP_TRY(e, hipStreamSynchronize(e->stream));  // the vectors above are locals
    e->sph_bvh = true;
    e->sph_builder = "host-median";
    e->sph_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return RB_OK;
}

// What an update does with one non-uniform field.  `first` = the engine's first update
// (gpu_wrapper.rs:117-163: only Create is acted on); otherwise :196-294.
enum class Act { None, Take, Delete };
Act field_action(int idx, const rb_field& f, bool first) {
    const bool bvh_field = (idx >= 5 && idx <= 7);
    if (first) return f.change == RB_CREATE ? Act::Take : Act::None;
    if (f.change == RB_UPDATE) return Act::Take;
    if (f.change == RB_DELETE) return Act::Delete;
    if (f.change == RB_CREATE) return bvh_field ? Act::Take : Act::None;  // "Create not allowed after initialization" except BVH (:242-280)
    return Act::None;
}

// may the library's own tree be wanted for this engine's meshes?  (host copies of triangles / indices are kept then)
bool may_want_own_tree(const rb_engine* e) { return !(e->opt.flags & RB_FLAG_REFERENCE_WALK); }
// a mesh of `n` elements whose host copy can wait (ensure_host_mesh): large enough for the device builder, and no flag that
// sends the build to a host builder anyway
bool host_copy_can_wait(const rb_engine* e, size_t n) {
    return n >= rb::kChunkDeviceBuildMin && !(e->opt.flags & (RB_FLAG_CHUNK_TREE_HOST | RB_FLAG_FAST_BVH | RB_FLAG_HOST_BVH | RB_FLAG_DEVICE_BVH));
}
// is it wanted for a mesh of n_tris triangles?
bool wants_own_tree(const rb_engine* e, uint32_t n_tris) {   // (asked only when the chunked walk is not in use)
    if (e->opt.flags & RB_FLAG_REFERENCE_WALK) return false;
    if (e->opt.flags & (RB_FLAG_FAST_BVH | RB_FLAG_DEVICE_BVH | RB_FLAG_HOST_BVH)) return true;
    return n_tris >= rb::kOwnTreeDefaultMinTriangles;
}

// The chunked walk (k_trace_chunk) is the walk of every multi-node mesh unless the caller names another one: it is
// the fastest exact walk at every size measured (profiles/r03_walks.txt: 576 to 1 048 578 triangles).
bool wants_chunk_walk(const rb_engine* e) {
    const uint32_t kern = e->opt.kernel ? e->opt.kernel : RB_KERNEL_STREAM;
    if (kern != RB_KERNEL_STREAM) return false;   // the one-pixel-per-lane dispatch shapes walk per segment
    if (e->opt.flags & RB_FLAG_CHUNK_WALK) return true;
    return !(e->opt.flags & (RB_FLAG_REFERENCE_WALK | RB_FLAG_FAST_BVH | RB_FLAG_DEVICE_BVH | RB_FLAG_HOST_BVH));
}

int apply_field(rb_engine* e, int idx, const rb_field& f, bool first) {
    const Act act = field_action(idx, f, first);
    if (act == Act::None) return RB_OK;
    const bool del = act == Act::Delete;
    const void* src = del ? nullptr : f.ptr;
    const size_t n = del ? 0 : f.count;
    int rc = RB_OK;
    switch (idx) {
        case 1:
            rc = upload(e, e->spheres, src, n, nullptr, true);
            e->n_spheres = static_cast<uint32_t>(n);
            if (!rc) rc = prep_materials(e, e->spheres.ptr ? &e->spheres.ptr->material : nullptr, sizeof(rb_sphere), n);
            if (!rc) rc = build_sphere_bvh(e, static_cast<const rb_sphere*>(src), n);
            break;
        case 2: rc = upload(e, e->uvs, src, n, nullptr, true); e->n_uvs = static_cast<uint32_t>(n); break;
        case 3:
            rc = upload(e, e->meshes, src, n, nullptr, true);
            e->n_meshes = static_cast<uint32_t>(n);
            if (!rc) rc = prep_materials(e, e->meshes.ptr ? &e->meshes.ptr->material : nullptr, sizeof(rb_mesh), n);
            break;
        case 4:
            // delete_lights creates a 4-byte buffer (buffers.rs:389-391): arrayLength() == 0
            rc = upload(e, e->lights, src, n, &e->n_lights, !del);
            if (del) e->n_lights = 0;
            if (!rc) rc = prep_materials(e, e->lights.ptr ? &e->lights.ptr->material : nullptr, sizeof(rb_point_light), e->n_lights);
            break;
        case 5:
            rc = upload(e, e->nodes, src, n, nullptr, true);
            e->n_nodes = static_cast<uint32_t>(n);
            e->host_nodes.assign(static_cast<const rb_bvh_node*>(src), static_cast<const rb_bvh_node*>(src) + n);
            e->prep_dirty = true;
            break;
        case 6:
            rc = upload(e, e->indices, src, n, &e->n_indices, true);
            e->prep_dirty = true;
            if (may_want_own_tree(e)) {
                e->host_index_len = n;
                e->host_indices_stale = host_copy_can_wait(e, n);
                if (e->host_indices_stale) std::vector<uint32_t>().swap(e->host_indices);
                else e->host_indices.assign(static_cast<const uint32_t*>(src), static_cast<const uint32_t*>(src) + n);
            }
            break;
        case 7:
            rc = upload(e, e->tris, src, n, nullptr, true);
            e->n_tris = static_cast<uint32_t>(n);
            e->prep_dirty = true;
            if (may_want_own_tree(e)) {
                e->host_tri_len = n;
                e->host_tris_stale = host_copy_can_wait(e, n);
                if (e->host_tris_stale) std::vector<rb_gpu_triangle>().swap(e->host_tris);
                else e->host_tris.assign(static_cast<const rb_gpu_triangle*>(src), static_cast<const rb_gpu_triangle*>(src) + n);
            }
            break;
        case 8:
            if (del) { rb_field empty{RB_UPDATE, nullptr, 0}; rc = upload_textures(e, empty); }
            else rc = upload_textures(e, f);
            break;
        default: break;
    }
    return rc;
}

// Host-side checks that stand in for WGSL's robust buffer access: anything that would make a HIP kernel
// read out of bounds or loop forever is refused.  They run on the scene the update WOULD produce -- the
// incoming fields merged with the kept host copies -- before a single buffer is touched, so a refused
// rb_update leaves the previous scene live and renderable.
struct ScenePlan {
    uint32_t bvh_stack = 0, max_mesh_index = 0;
};
int validate_scene(rb_engine* e, const rb_config* cfg, bool first, ScenePlan& plan) {
    const Act a_nodes = field_action(5, cfg->bvh_nodes, first), a_idx = field_action(6, cfg->bvh_indices, first),
              a_tris = field_action(7, cfg->bvh_triangles, first), a_meshes = field_action(3, cfg->meshes, first);
    const bool take_uniforms = first ? (cfg->uniforms.change == RB_CREATE) : (cfg->uniforms.change == RB_UPDATE);
    // ---- the tree the kernels would walk
    const rb_bvh_node* nodes = e->host_nodes.data();
    uint32_t n_nodes = static_cast<uint32_t>(e->host_nodes.size());
    if (a_nodes == Act::Take) {
        nodes = static_cast<const rb_bvh_node*>(cfg->bvh_nodes.ptr);
        if (cfg->bvh_nodes.count >= (1ull << 31)) return fail(e, RB_ERR_INVALID_BVH, "too many BVH nodes");
        n_nodes = static_cast<uint32_t>(cfg->bvh_nodes.count);
    } else if (a_nodes == Act::Delete) {
        n_nodes = 0;
    }
    uint64_t index_len = e->n_indices;  // arrayLength(&bvh_indices): an empty vector still has one element
    if (a_idx == Act::Take) index_len = std::max<uint64_t>(cfg->bvh_indices.count, 1);
    else if (a_idx == Act::Delete) index_len = 1;
    if (index_len >= (1ull << 31)) return fail(e, RB_ERR_INVALID_BVH, "too many BVH indices");
    plan.bvh_stack = e->bvh_stack;
    if (n_nodes > 0) {
        std::string why;
        uint32_t depth = 0;
        if (!rb::bvh_validate(nodes, n_nodes, rb::kStackDepth, why, &depth)) return fail(e, RB_ERR_INVALID_BVH, "%s", why.c_str());
        plan.bvh_stack = depth;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const rb_bvh_node& n = nodes[i];
            if (n.primitive_count > 0 && static_cast<uint64_t>(n.first_primitive) + n.primitive_count > index_len)
                return fail(e, RB_ERR_INVALID_BVH, "leaf %u covers [%u, +%u) of %llu bvh_indices", i, n.first_primitive,
                            n.primitive_count, static_cast<unsigned long long>(index_len));
        }
    }
    // ---- every triangle's material must exist (shader.wgsl:370 reads meshes[tri.mesh_index])
    uint64_t n_tris = e->n_tris;
    plan.max_mesh_index = e->max_mesh_index;
    if (a_tris == Act::Take) {
        const rb_gpu_triangle* t = static_cast<const rb_gpu_triangle*>(cfg->bvh_triangles.ptr);
        uint32_t mx = 0;
        for (size_t i = 0; i < cfg->bvh_triangles.count; ++i) mx = std::max(mx, t[i].mesh_index);
        plan.max_mesh_index = mx;
        n_tris = cfg->bvh_triangles.count;
        if (n_tris >= (1ull << 31)) return fail(e, RB_ERR_INVALID_BVH, "too many triangles");
    } else if (a_tris == Act::Delete) {
        n_tris = 0;
        plan.max_mesh_index = 0;
    }
    uint64_t n_meshes = e->n_meshes;
    if (a_meshes == Act::Take) n_meshes = cfg->meshes.count;
    const uint32_t color_hash = take_uniforms ? static_cast<const rb_uniforms*>(cfg->uniforms.ptr)->color_hash_enabled
                                              : e->uniforms.color_hash_enabled;
    if (n_tris > 0 && color_hash == 0 && plan.max_mesh_index >= n_meshes)
        return fail(e, RB_ERR_INVALID_MESHES, "a triangle references mesh %u of %llu", plan.max_mesh_index,
                    static_cast<unsigned long long>(n_meshes));
    // ---- textures and the frame
    if (field_action(8, cfg->textures, first) == Act::Take) {
        const rb_texture* t = static_cast<const rb_texture*>(cfg->textures.ptr);
        for (size_t i = 0; i < cfg->textures.count; ++i) {
            if (t[i].width == 0 || t[i].height == 0) return fail(e, RB_ERR_INVALID_TEXTURES, "texture %zu is empty", i);
            if (!t[i].rgba_data) return fail(e, RB_ERR_INVALID_TEXTURES, "texture %zu has no data", i);
        }
    }
    if (take_uniforms) {
        const rb_uniforms* u = static_cast<const rb_uniforms*>(cfg->uniforms.ptr);
        const uint64_t px = static_cast<uint64_t>(u->width) * stripe_geometry(e->opt, u->height).padded;
        if (px >= (1ull << 31)) return fail(e, RB_ERR_INVALID_UNIFORMS, "frame of %u x %u pixels is too large", u->width, u->height);
    }
    return RB_OK;
}

void set_device(const rb_engine* e) { (void)hipSetDevice(e->device); }

// update_uniforms count patch-up -- gpu_wrapper.rs:475-495: Create/Update overwrite the count with the
// vector length, Delete zeroes it, Keep leaves the caller's value (clamped to the buffer here so that a
// stale count cannot read out of bounds; the WGSL relies on robust buffer access for that).
uint32_t patch_count(uint32_t change, uint32_t given, uint32_t len) {
    if (change == RB_CREATE || change == RB_UPDATE) return len;
    if (change == RB_DELETE) return 0u;
    return std::min(given, len);
}

// The host builders and checkers read host_tris / host_indices: bring back what rb_update left on the device only.
int ensure_host_mesh(rb_engine* e) {
    if (!e->host_tris_stale && !e->host_indices_stale) return RB_OK;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (e->host_tris_stale) {
        if (e->host_tri_len > e->tris.count) return fail(e, RB_ERR_DEVICE, "the triangle buffer is shorter than the mesh it was made from");
        e->host_tris.resize(e->host_tri_len);
        HIP_TRY(e, hipMemcpy(e->host_tris.data(), e->tris.ptr, sizeof(rb_gpu_triangle) * e->host_tri_len, hipMemcpyDeviceToHost));
        e->host_tris_stale = false;
    }
    if (e->host_indices_stale) {
        if (e->host_index_len > e->indices.count) return fail(e, RB_ERR_DEVICE, "the index buffer is shorter than the mesh it was made from");
        e->host_indices.resize(e->host_index_len);
        HIP_TRY(e, hipMemcpy(e->host_indices.data(), e->indices.ptr, 4u * e->host_index_len, hipMemcpyDeviceToHost));
        e->host_indices_stale = false;
    }
    return RB_OK;
}

int ensure_prepared(rb_engine* e) {
    // shader.wgsl:336 skips triangle ids >= uniforms.bvh_triangle_count: the prepared triangles carry that
    // guard as their `valid` word, so they depend on the (patched) count as well as on the buffers
    const uint32_t tri_count = patch_count(e->last_change_tris, e->uniforms.bvh_triangle_count, e->n_tris);
    if (!e->prep_dirty && e->prep_tri_count == tri_count) return RB_OK;
    const uint32_t len = e->n_indices;  // arrayLength(&bvh_indices) >= 1
    HIP_TRY(e, e->ptris.resize(len));
    HIP_TRY(e, e->pshade.resize(len));
    int rc = rb::launch_prep_tris(e->tris.ptr, tri_count, e->indices.ptr, len, e->ptris.ptr, e->pshade.ptr, e->stream);
    if (rc) return fail(e, RB_ERR_DEVICE, "prep kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
    e->prep_dirty = false;
    e->prep_tri_count = tri_count;
    // ---- the chunked walk's tree: the caller's tree with the library's own levels below its leaves
    e->chunk_ready = false;
    e->fast_ready = false;
    if (wants_chunk_walk(e) && e->host_nodes.size() > 1 && e->host_tri_len > 0 && e->host_index_len > 0 && tri_count > 0) {
        const auto t_begin = std::chrono::steady_clock::now();
        const uint32_t n_tris = std::min<uint32_t>(tri_count, static_cast<uint32_t>(e->host_tri_len));
        const uint32_t n_idx = static_cast<uint32_t>(e->host_index_len), n_nodes = static_cast<uint32_t>(e->host_nodes.size());
        // which builder: the device one from kChunkDeviceBuildMin slots up (one block per reference leaf; C5's 10^6 triangles
        // in a few ms where the host's threads take 11-15), the host's below; either can be forced.  Same walk, same frames.
        const bool force_host = (e->opt.flags & RB_FLAG_CHUNK_TREE_HOST) != 0u, force_dev = (e->opt.flags & RB_FLAG_CHUNK_TREE_DEVICE) != 0u;
        bool built = false;
        size_t n = 0;
        e->chunk_builder = "";
        if (!force_host && (force_dev || n_idx >= rb::kChunkDeviceBuildMin) && n_idx <= e->indices.count && n_tris <= e->tris.count) {
            rb::DeviceChunkTree dt;
            const int brc = rb::device_chunk_tree_build(e->tris.ptr, n_tris, e->indices.ptr, n_idx, e->host_nodes.data(), n_nodes, rb::kStackDepth, &dt, e->stream);
            if (brc > 0) return fail(e, RB_ERR_DEVICE, "chunk tree build failed: %s", hipGetErrorString(static_cast<hipError_t>(brc)));
            if (brc == 0) {
                e->chunk_nodes.adopt(dt.nodes, dt.nodes_capacity);
                e->chunk_rank_slot.adopt(dt.rank_slot, dt.n_pos);
                e->chunk_pos_slot.adopt(dt.pos_slot, dt.n_pos);
                e->chunk_pos_rank.adopt(dt.pos_rank, dt.n_pos);
                e->chunk_n_nodes = dt.n_nodes;
                e->chunk_root = dt.root;
                e->chunk_depth = dt.depth;
                n = dt.n_pos;
                built = true;
                e->chunk_builder = "device";
            }
        }
        if (!built) {
            rb::ChunkTree ct;
            rc = ensure_host_mesh(e);
            if (rc) return rc;
            if (rb::chunk_tree_build(e->host_tris.data(), n_tris, e->host_indices.data(), n_idx, e->host_nodes.data(), n_nodes, rb::kStackDepth, ct)) {
                n = ct.pos_slot.size();
                rc = upload(e, e->chunk_nodes, ct.nodes.data(), ct.nodes.size(), nullptr, true);
                if (!rc) rc = upload(e, e->chunk_rank_slot, ct.rank_slot.data(), ct.rank_slot.size(), nullptr, true);
                if (!rc) rc = upload(e, e->chunk_pos_slot, ct.pos_slot.data(), n, nullptr, true);
                if (!rc) rc = upload(e, e->chunk_pos_rank, ct.pos_rank.data(), n, nullptr, true);
                if (rc) return rc;
                HIP_TRY(e, hipStreamSynchronize(e->stream));  // `ct` is a local
                e->chunk_n_nodes = ct.nodes.size();
                e->chunk_root = ct.root;
                e->chunk_depth = ct.depth;
                built = true;
                e->chunk_builder = "host";
            }
        }
        if (built) {
            HIP_TRY(e, e->chunk_a.resize(n * 4));
            HIP_TRY(e, e->chunk_b.resize(n * 4));
            HIP_TRY(e, e->chunk_c.resize(n * 4));
            rc = rb::launch_chunk_gather(e->ptris.ptr, e->chunk_pos_slot.ptr, e->chunk_pos_rank.ptr, static_cast<uint32_t>(n), e->chunk_a.ptr, e->chunk_b.ptr,
                                         e->chunk_c.ptr, e->stream);
            if (rc) return fail(e, RB_ERR_DEVICE, "chunk gather launch failed");
            HIP_TRY(e, hipStreamSynchronize(e->stream));
            e->chunk_ready = true;
            e->chunk_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        }
    }
    // ---- the library's own tree over the same triangles
    if (!e->chunk_ready && wants_own_tree(e, tri_count) && e->host_nodes.size() > 1 && e->host_tri_len > 0 && e->host_index_len > 0 && tri_count > 0) {
        rc = ensure_host_mesh(e);
        if (rc) return rc;
        rb::FastTree ft;
        const auto t_begin = std::chrono::steady_clock::now();
        const uint32_t n_idx = static_cast<uint32_t>(e->host_indices.size());
        const uint32_t n_nodes = static_cast<uint32_t>(e->host_nodes.size());
        const uint32_t n_tris = std::min<uint32_t>(tri_count, static_cast<uint32_t>(e->host_tris.size()));
        bool built = false;
        e->fast_builder = "";
        // which builder: the device one from kDeviceBuildMinTriangles up (milliseconds instead of ~0.15 s per
        // million triangles), the host's binned SAH below; either can be forced
        const float small_cap = (e->opt.flags & RB_FLAG_SKIP_NEAR_DEGENERATE) ? 0.0f : rb::kFastSmallCap;
        const bool force_host = (e->opt.flags & RB_FLAG_HOST_BVH) != 0u, force_dev = (e->opt.flags & RB_FLAG_DEVICE_BVH) != 0u;
        const bool try_device = !force_host && (force_dev || n_tris >= rb::kDeviceBuildMinTriangles);
        if (try_device) {
            // reference-order metadata on the host (one pass over the caller's tree), the tree on the device
            if (rb::fast_bvh_prepare(e->host_tris.data(), n_tris, e->host_indices.data(), n_idx, e->host_nodes.data(), n_nodes, ft, small_cap) &&
                ft.slots.size() >= 1024) {
                const uint32_t n = static_cast<uint32_t>(ft.slots.size());
                DevBuf<uint32_t> visit_slots;
                rc = upload(e, visit_slots, ft.slots.data(), ft.slots.size(), nullptr, true);
                if (!rc) rc = upload(e, e->slot_meta, ft.slot_meta.data(), ft.slot_meta.size(), nullptr, true);
                if (rc) return rc;
                HIP_TRY(e, e->fast_nodes.resize(n - 1));
                HIP_TRY(e, e->fast_slots.resize(n));
                rb::DeviceTreeInfo info{};
                rc = rb::device_fast_bvh_build(e->tris.ptr, e->indices.ptr, visit_slots.ptr, n, e->slot_meta.ptr, e->fast_nodes.ptr,
                                               e->fast_slots.ptr, &info, e->stream, (e->opt.flags & RB_FLAG_DEVICE_LBVH) != 0u);
                if (rc && rc != static_cast<int>(hipErrorNotReady))
                    return fail(e, RB_ERR_DEVICE, "device BVH build failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
                if (rc) info.depth = 0xFFFFFFFFu;  // clustering did not converge within its round limit: host builder
                // a tree deeper than the LDS stack spills to a global scratch column per lane, which only the
                // persistent kernels (bounded grid) get; otherwise use the depth-limited host builder
                const uint32_t kern = e->opt.kernel ? e->opt.kernel : RB_KERNEL_STREAM;
                bool usable = info.depth <= rb::kStackDepth;
                if (!usable && kern != RB_KERNEL_PIXEL && info.depth <= 128u) {
                    const size_t lanes = rb::stream_kernel_max_threads(e->opt._reserved[0]);
                    HIP_TRY(e, e->stack_overflow.resize(lanes * (info.depth - rb::kStackDepth)));
                    usable = true;
                }
                if (usable) {
                    ft.root = info.root;
                    ft.depth = info.depth;
                    ft.margin = info.margin;
                    ft.root_amax = info.root_amax;
                    for (int i = 0; i < 3; ++i) {
                        ft.bmin[i] = info.bmin[i];
                        ft.bmax[i] = info.bmax[i];
                    }
                    built = true;
                    e->fast_builder = (e->opt.flags & RB_FLAG_DEVICE_LBVH) ? "device-lbvh" : "device-ploc";
                }
            }
        }
        if (!built) {
            if (!rb::fast_bvh_build(e->host_tris.data(), n_tris, e->host_indices.data(), n_idx, e->host_nodes.data(), n_nodes,
                                    rb::kStackDepth, ft, small_cap))
                return RB_OK;  // keep the reference walk
            rc = upload(e, e->fast_nodes, ft.nodes.data(), ft.nodes.size(), nullptr, true);
            if (!rc) rc = upload(e, e->fast_slots, ft.slots.data(), ft.slots.size(), nullptr, true);
            if (rc) return rc;
            e->fast_builder = "host-sah";
        }
        rc = upload(e, e->slot_meta, ft.slot_meta.data(), ft.slot_meta.size(), nullptr, true);
        if (!rc) rc = upload(e, e->ref_parent, ft.ref_parent.data(), ft.ref_parent.size(), nullptr, true);
        if (!rc) rc = upload(e, e->gnodes, ft.gnodes.data(), ft.gnodes.size(), nullptr, true);
        if (!rc) rc = upload(e, e->gslots, ft.gslots.data(), ft.gslots.size(), nullptr, true);
        if (rc) return rc;
        const size_t n_items = e->fast_slots.count;
        HIP_TRY(e, e->fast_tris.resize(n_items));
        rc = rb::launch_gather_tris(e->ptris.ptr, e->fast_slots.ptr, static_cast<uint32_t>(n_items), e->fast_tris.ptr, e->stream);
        if (rc) return fail(e, RB_ERR_DEVICE, "gather kernel launch failed");
        HIP_TRY(e, hipStreamSynchronize(e->stream));  // `ft` is a local
        e->fast_build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        e->fast_root = ft.root;
        e->fast_depth = ft.depth;
        e->fast_margin = ft.margin;
        e->fast_root_amax = ft.root_amax;
        for (int i = 0; i < 3; ++i) {
            e->fast_bmin[i] = ft.bmin[i];
            e->fast_bmax[i] = ft.bmax[i];
        }
        e->fast_ready = true;
    }
    return RB_OK;
}

rb::KParams make_params(rb_engine* e, uint32_t first_pass, uint32_t n_passes, int src, int dst) {
    rb::KParams p{};
    p.u = e->uniforms;
    p.u.spheres_count = patch_count(e->last_change_spheres, e->uniforms.spheres_count, e->n_spheres);
    p.u.bvh_node_count = patch_count(e->last_change_nodes, e->uniforms.bvh_node_count, e->n_nodes);
    p.u.bvh_triangle_count = patch_count(e->last_change_tris, e->uniforms.bvh_triangle_count, e->n_tris);
    p.spheres = e->spheres.ptr;
    p.lights = e->lights.ptr;
    p.meshes = e->meshes.ptr;
    p.nodes = e->nodes.ptr;
    p.indices = e->indices.ptr;
    p.tris = e->tris.ptr;
    p.ptris = e->ptris.ptr;
    p.pshade = e->pshade.ptr;
    p.uvs = e->uvs.ptr;
    p.tex_data = e->tex_data.ptr;
    p.tex_info = e->tex_info.ptr;
    p.srgb_lut = e->srgb_lut.ptr;
    p.accum_in = e->slot[src].accum.ptr;
    p.accum_out = e->slot[dst].accum.ptr;
    p.out_rgba = e->slot[dst].rgba.ptr;
    p.counters = e->counters.ptr;
    p.queue = e->queue.ptr;
    p.n_lights = e->n_lights;
    p.n_meshes = e->n_meshes;
    p.index_len = e->n_indices;
    p.n_uvs = e->n_uvs;
    p.n_tex = e->n_tex;
    p.first_pass = first_pass;
    p.n_passes = n_passes;
    p.samples_per_pass = e->prh.samples_per_pass;
    p.shard_rank = e->opt.shard_rank;
    p.shard_count = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    p.stripe_rows = e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows;
    p.local_rows = e->local_rows;
    p.colors = e->colors.ptr;
    const bool use_chunk = e->chunk_ready && p.u.bvh_node_count == e->n_nodes && p.u.bvh_node_count > 1u;
    p.chunk_nodes = use_chunk ? e->chunk_nodes.ptr : nullptr;
    p.chunk_a = e->chunk_a.ptr;
    p.chunk_b = e->chunk_b.ptr;
    p.chunk_c = e->chunk_c.ptr;
    p.chunk_rank_slot = e->chunk_rank_slot.ptr;
    p.chunk_root = e->chunk_root;
    p.chunk_n = static_cast<uint32_t>(e->chunk_rank_slot.count);
    const bool use_fast = !use_chunk && e->fast_ready && p.u.bvh_node_count == e->n_nodes && p.u.bvh_node_count > 1u;
    p.fast_nodes = use_fast ? e->fast_nodes.ptr : nullptr;
    p.gnodes = e->gnodes.ptr;
    p.gslots = e->gslots.ptr;
    p.fast_skip_second_pass = (e->opt.flags & RB_FLAG_SKIP_NEAR_DEGENERATE) ? 1u : 0u;
    p.fast_tris = reinterpret_cast<const float*>(e->fast_tris.ptr);
    p.fast_slots = e->fast_slots.ptr;
    p.slot_meta = e->slot_meta.ptr;
    p.ref_parent = e->ref_parent.ptr;
    p.fast_root = e->fast_root;
    p.fast_margin = e->fast_margin;
    p.fast_root_amax = e->fast_root_amax;
    for (int i = 0; i < 3; ++i) {
        p.fast_bmin[i] = e->fast_bmin[i];
        p.fast_bmax[i] = e->fast_bmax[i];
    }
    const bool use_sph_bvh = e->sph_bvh && p.u.spheres_count == e->n_spheres;
    p.sph_nodes = use_sph_bvh ? e->sph_nodes.ptr : nullptr;
    p.sph_leaf = e->sph_leaf.ptr;
    p.sph_id = e->sph_id.ptr;
    p.sph_root = e->sph_root;
    // a single-node tree is walked without a stack (rb_kernels.hip, intersect_bvh)
    p.stack_depth = (p.u.bvh_node_count <= 1u) ? 0u : std::max(e->bvh_stack, 1u);
    if (use_sph_bvh) p.stack_depth = std::max(p.stack_depth, e->sph_depth);
    if (use_fast) p.stack_depth = std::max(p.stack_depth, std::min(e->fast_depth, rb::kStackDepth));
    // (max with what is there already: with no_leaf_stepping the launch falls to the per-segment kernel, which walks the
    // CALLER's tree and needs bvh_stack entries -- the chunk tree can be shallower where it prunes empty subtrees)
    if (use_chunk) p.stack_depth = std::max(p.stack_depth, e->chunk_depth + 1u);
    // every walk a launch can run must fit the column it shares with the others (rb_internal.hpp, kStackEntryBytes)
    e->stack_depth_covers = p.stack_depth <= rb::kStackDepth && (p.u.bvh_node_count <= 1u || p.stack_depth >= e->bvh_stack) &&
                            (!use_sph_bvh || p.stack_depth >= e->sph_depth) && (!use_chunk || p.stack_depth >= e->chunk_depth + 1u) &&
                            (!use_fast || p.stack_depth >= std::min(e->fast_depth, rb::kStackDepth));
    p.stack_overflow = e->stack_overflow.ptr;
    p.blocks_per_cu = e->opt._reserved[0];
    // the caller's reservation size: a multiple of 64 items, at most 4096 (the launcher's own range; beyond it the
    // 32-bit queue arithmetic of the stream kernels could wrap and hand items out twice)
    p.queue_batch = e->opt._reserved[2] ? std::min<uint32_t>(((std::min<uint32_t>(e->opt._reserved[2], 4096u) + 63u) / 64u) * 64u, 4096u) : 0u;
    p.no_leaf_stepping = e->opt._reserved[3];
    p.lds_mode = e->opt._reserved[4];
    return p;
}

int require_ready(rb_engine* e) {
    if (!e->initialized) return fail(e, RB_ERR_NOT_INITIALIZED, "engine has not received its first update");
    if (!e->scene_valid) return fail(e, RB_ERR_DEVICE, "the last update failed half-way on the device; send the scene again");
    if (!e->have_uniforms) return fail(e, RB_ERR_UNIFORMS_NOT_INITIALIZED, "Uniforms must be initialized");
    return RB_OK;
}

int clear_accum(rb_engine* e) {
    const size_t px = static_cast<size_t>(e->width) * e->padded_rows;
    e->spec_valid = false;
    if (px) HIP_TRY(e, hipMemsetAsync(e->slot[e->cur].accum.ptr, 0, px * 16, e->stream));
    return RB_OK;
}

int accumulate_timing(rb_engine* e) {
    float ms = 0.0f;
    if (e->last_launches > 0) {
        HIP_TRY(e, hipEventSynchronize(e->ev_end));
        HIP_TRY(e, hipEventElapsedTime(&ms, e->ev_begin, e->ev_end));
    }
    e->last_dispatch_ms = ms;
    if (e->timing_pending) {
        e->stats.kernel_ms += ms;
        for (uint32_t i = 0; i + 3 <= e->ev_used; i += 3) {
            float t = 0.0f, a = 0.0f;
            HIP_TRY(e, hipEventElapsedTime(&t, e->ev_pool[i], e->ev_pool[i + 1]));
            HIP_TRY(e, hipEventElapsedTime(&a, e->ev_pool[i + 1], e->ev_pool[i + 2]));
            e->stats.trace_ms += t;
            e->stats.accumulate_ms += a;
        }
        e->timing_pending = false;
    }
    return RB_OK;
}

// Colour-buffer budget of the stream kernels (one float4 per (pixel, sample) of a launch chunk): the caller's
// figure, else 4 GiB but never more than half of what the device has free right now -- eight launches per C2
// frame instead of one cost 0.4 %, and a library that sits behind a GUI should not take 34 GB for a 1080p frame.
uint64_t color_budget_bytes(rb_engine* e) {
    if (e->opt._reserved[1]) return static_cast<uint64_t>(e->opt._reserved[1]) << 20;
    if (e->color_budget == 0) {   // asked once per update: hipMemGetInfo is a driver round trip, and the iterator dispatches per pass
        uint64_t budget = 4ull << 30;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min<uint64_t>(budget, (free_b + e->colors.count * sizeof(float)) / 2);
        e->color_budget = std::max<uint64_t>(budget, 1ull << 20);
    }
    return e->color_budget;
}

// The stream kernels' colour buffer for a group of n_passes passes, and the passes per launch it allows: one float4 per
// (pixel, sample) of a launch chunk.  Keeps the item count below 2^31 and, unless the caller fixed the chunk, the buffer
// within the budget; if the device cannot give even that, halves.  Allocates (lazily: the first dispatch, or rb_reserve).
int reserve_colors(rb_engine* e, uint32_t n_passes, uint32_t* chunk_out) {
    const uint32_t kernel = e->opt.kernel ? e->opt.kernel : RB_KERNEL_STREAM;
    uint32_t chunk = e->opt.passes_per_launch ? e->opt.passes_per_launch : n_passes;
    if (kernel == RB_KERNEL_STREAM && n_passes != 0 && e->width != 0 && e->local_rows != 0) {
        const uint64_t tiles = static_cast<uint64_t>((e->width + 7) / 8) * ((e->local_rows + 7) / 8);
        const uint64_t per_pass = tiles * 64ull * e->prh.samples_per_pass;  // items per pass
        const uint64_t budget_items = color_budget_bytes(e) / 16ull;
        uint64_t max_chunk = std::min<uint64_t>((1ull << 31) / std::max<uint64_t>(per_pass, 1) , 0xFFFFFFFFull);
        if (!e->opt.passes_per_launch) max_chunk = std::min(max_chunk, std::max<uint64_t>(budget_items / std::max<uint64_t>(per_pass, 1), 1));
        if (max_chunk == 0) return fail(e, RB_ERR_INVALID_UNIFORMS, "frame too large for one launch");
        chunk = static_cast<uint32_t>(std::min<uint64_t>(chunk, max_chunk));
        for (;;) {
            const hipError_t st = e->colors.reserve(per_pass * chunk * 4);
            if (st == hipSuccess) break;
            (void)hipGetLastError();  // clear the sticky out-of-memory status
            if (st != hipErrorOutOfMemory || chunk == 1)
                return fail(e, RB_ERR_DEVICE, "colour buffer of %llu bytes: %s", static_cast<unsigned long long>(per_pass * chunk * 16ull),
                            hipGetErrorString(st));
            chunk = (chunk + 1) / 2;
        }
    }
    *chunk_out = chunk;
    return RB_OK;
}

// dispatch_compute_progressive without the host sync -- gpu_wrapper.rs:365-400: passes
// [first_pass, first_pass + n_passes) on top of slot `src`, into slot `dst` (the same slot, or the other one
// when the iterator runs a pass ahead).
int dispatch(rb_engine* e, uint32_t first_pass, uint32_t n_passes, int src, int dst) {
    int rc = ensure_prepared(e);
    if (rc) return rc;
    if (n_passes == 0 || e->width == 0 || e->local_rows == 0) {
        e->last_launches = 0;
        return RB_OK;
    }
    if (e->timing_pending) {  // fold the previous group's events before they are recorded again
        rc = accumulate_timing(e);
        if (rc) return rc;
    }
    const uint32_t kernel = e->opt.kernel ? e->opt.kernel : RB_KERNEL_STREAM;
    const bool stats = (e->opt.flags & RB_FLAG_STATS) != 0;
    uint32_t chunk = 0;
    rc = reserve_colors(e, n_passes, &chunk);
    if (rc) return rc;
    HIP_TRY(e, hipEventRecord(e->ev_begin, e->stream));
    uint32_t launches = 0;
    e->ev_used = 0;
    for (uint32_t done = 0; done < n_passes;) {
        const uint32_t n = std::min(chunk, n_passes - done);
        // the first chunk resumes `src`; later chunks of the same group continue in `dst`
        rb::KParams p = make_params(e, first_pass + done, n, done == 0 ? src : dst, dst);
        if (!e->stack_depth_covers) return fail(e, RB_ERR_DEVICE, "internal: a traversal is deeper than its LDS stack column (%u entries)", p.stack_depth);
        rb::LaunchInfo li{};
        // per-chunk timing events (first 256 chunks of a group; later ones only count in the total)
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        if (e->ev_used + 3 <= 768) {
            while (e->ev_pool.size() < e->ev_used + 3) {
                hipEvent_t x;
                HIP_TRY(e, hipEventCreate(&x));
                e->ev_pool.push_back(x);
            }
            for (int i = 0; i < 3; ++i) ev[i] = e->ev_pool[e->ev_used + i];
            e->ev_used += 3;
            HIP_TRY(e, hipEventRecord(ev[0], e->stream));
        }
        rc = rb::launch_render(p, kernel, stats, e->stream, &li, ev[1]);
        if (rc) return fail(e, RB_ERR_DEVICE, "render kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
        if (ev[2]) {
            if (kernel != RB_KERNEL_STREAM) HIP_TRY(e, hipEventRecord(ev[1], e->stream));
            HIP_TRY(e, hipEventRecord(ev[2], e->stream));
        }
        if (li.kernel_name) e->last_kernel_name = li.kernel_name;
        done += n;
        launches++;
    }
    HIP_TRY(e, hipEventRecord(e->ev_end, e->stream));
    HIP_TRY(e, hipEventRecord(e->slot[dst].done, e->stream));
    e->last_launches = launches;
    e->timing_pending = true;
    e->stats.launches += launches;
    return RB_OK;
}

// Copies the committed frame's RGBA8 rows (local stripe order when sharded) to caller memory: the host waits
// for the launches that produced the slot, then a blocking copy.  The engine's stream is non-blocking, so a
// pass the iterator has already started on the OTHER slot keeps running underneath this copy.
int read_slot_rgba(rb_engine* e, int slot, uint8_t* out) {
    if (!out) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    const size_t bytes = static_cast<size_t>(e->width) * 4 * (sc == 1 ? e->height : e->padded_rows);
    if (bytes == 0) return RB_OK;
    // Page-locked destination (rb_host_alloc, or memory the caller registered with HIP): a DMA on the copy
    // stream straight into it, behind the slot's event -- no staging, no host-side copy.
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, out) == hipSuccess && attr.type == hipMemoryTypeHost) {
        HIP_TRY(e, hipStreamWaitEvent(e->copy_stream, e->slot[slot].done, 0));
        HIP_TRY(e, hipMemcpyAsync(out, e->slot[slot].rgba.ptr, bytes, hipMemcpyDeviceToHost, e->copy_stream));
        HIP_TRY(e, hipStreamSynchronize(e->copy_stream));
        return RB_OK;
    }
    (void)hipGetLastError();  // an unregistered pointer makes the query fail: that is the ordinary case
    HIP_TRY(e, hipEventSynchronize(e->slot[slot].done));
    HIP_TRY(e, hipMemcpy(out, e->slot[slot].rgba.ptr, bytes, hipMemcpyDeviceToHost));
    return RB_OK;
}

int read_rgba(rb_engine* e, uint8_t* out) {
    // uploads and clears queued after the slot's last launch group must be over as well
    HIP_TRY(e, hipEventRecord(e->slot[e->cur].done, e->stream));
    return read_slot_rgba(e, e->cur, out);
}

int update_fields(rb_engine* e, const rb_config* cfg) {
    int rc = check_fields(e, cfg);
    if (rc) return rc;
    const bool first = !e->initialized;
    rc = first ? validate_init(e, cfg) : validate(e, cfg);
    if (rc) return rc;
    ScenePlan plan;
    rc = validate_scene(e, cfg, first, plan);
    if (rc) return rc;   // nothing has been touched: the previous scene stays live

    // ---- from here on the buffers change; a device failure half-way leaves the engine refusing to render
    e->scene_valid = false;
    e->spec_valid = false;
    e->color_budget = 0;
    // uniforms (gpu_wrapper.rs:122-136 / :165-192)
    const bool take_uniforms = first ? (cfg->uniforms.change == RB_CREATE) : (cfg->uniforms.change == RB_UPDATE);
    if (take_uniforms) {
        const rb_uniforms* u = static_cast<const rb_uniforms*>(cfg->uniforms.ptr);
        if (u->width != e->width || u->height != e->height || e->slot[0].accum.ptr == nullptr) {
            rc = resize_frame(e, u->width, u->height);
            if (rc) return rc;
        }
        e->uniforms = *u;
        e->prh.total_samples = u->total_samples;  // ProgressiveRenderHelper::update (:47-52)
        e->prh.total_passes = (u->total_samples + e->prh.samples_per_pass - 1) / e->prh.samples_per_pass;
    }
    // self.rc = new_rc (:298): width()/height()/update_uniforms panic unless the *latest*
    // config carried Create/Update uniforms (:303-329,470-473).
    e->have_uniforms = has_data(cfg->uniforms);

    for (int i = 1; i < 9; ++i) {
        rc = apply_field(e, i, *field_at(cfg, i), first);
        if (rc) return rc;
    }
    e->last_change_spheres = cfg->spheres.change;
    e->last_change_nodes = cfg->bvh_nodes.change;
    e->last_change_tris = cfg->bvh_triangles.change;
    e->bvh_stack = plan.bvh_stack;
    e->max_mesh_index = plan.max_mesh_index;
    e->initialized = true;
    e->scene_valid = true;
    return RB_OK;
}

// Inputs are borrowed only for this call: whatever update_fields has queued from the caller's
// buffers must have left them before we return -- also when it stops half-way with an error.
// (Without Create/Update uniforms the reference panics at the next use, gpu_wrapper.rs:303-329;
// here that is require_ready's error.)
int update_locked(rb_engine* e, const rb_config* cfg) {
    const int rc = update_fields(e, cfg);
    const hipError_t st = hipStreamSynchronize(e->stream);
    if (rc) return rc;
    if (st != hipSuccess) return fail(e, RB_ERR_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(st));
    return RB_OK;
}

// zero the accumulation, run every pass (dispatch_compute, gpu_wrapper.rs:406-426) -- no read-back
int render_async(rb_engine* e) {
    int rc = require_ready(e);
    if (rc) return rc;
    rc = clear_accum(e);  // :407-411
    if (rc) return rc;
    e->prh.current_pass = 0;
    rc = dispatch(e, 0, e->prh.total_passes, e->cur, e->cur);
    if (rc) return rc;
    e->prh.current_pass = e->prh.total_passes ? e->prh.total_passes - 1 : 0;  // loop variable's last value (:415)
    return RB_OK;
}

int render_locked(rb_engine* e, uint8_t* rgba_out) {
    if (!rgba_out && !(e->net.nranks > 1 && e->net.rank != 0)) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    int rc = render_async(e);
    if (rc) return rc;
    if (e->net.nranks > 1) {  // one process per device: the frame is assembled on rank 0
        std::string why;
        if (rb::gather_process(e->net, e->slot[e->cur].rgba.ptr, e->width, e->height, e->padded_rows,
                               e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows, e->copy_stream, e->slot[e->cur].done,
                               rgba_out, why))
            return fail(e, RB_ERR_DEVICE, "%s", why.c_str());
    } else {
        rc = read_rgba(e, rgba_out);
        if (rc) return rc;
    }
    return accumulate_timing(e);
}

// One step of the progressive iterator on one engine, without the delivery: passes [current_pass, +n) end up in
// slot[cur].  They are taken from the run-ahead slot when the previous call started exactly these passes there (and
// nothing has touched the scene or the accumulation since); then the next group is started on the other slot, so that
// it computes while the caller's frame is exchanged and copied out (frame_buffer.rs:164-221 pumps frames from a worker
// thread; lib.rs:200-205 syncs, maps and mirrors per pass).  The exchange and the read-back run on the engine's second
// stream behind the slot's event, so a sharded engine -- one process per device, or a part of a multi-device handle --
// runs ahead like a whole-frame one.
int iter_advance(rb_engine* e, uint32_t per_frame) {
    int rc = require_ready(e);
    if (rc) return rc;
    if (!e->iter_initialized) {  // lib.rs:181-192
        rc = clear_accum(e);
        if (rc) return rc;
        e->iter_initialized = true;
    }
    const uint32_t per = std::max(per_frame, 1u);
    const uint32_t n = std::min(per, e->prh.total_passes - e->prh.current_pass);
    if (e->spec_valid && e->spec_first == e->prh.current_pass && e->spec_n == n) {
        e->cur = 1 - e->cur;   // commit the pass group that has been running since the previous call
        e->spec_valid = false;
    } else {
        e->spec_valid = false;
        rc = dispatch(e, e->prh.current_pass, n, e->cur, e->cur);  // lib.rs:200-203 (n = 1 there)
        if (rc) return rc;
    }
    e->prh.current_pass += n;  // lib.rs:213
    rc = accumulate_timing(e);
    if (rc) return rc;
    // ---- run ahead: the next group on the other slot
    if (!(e->opt.flags & RB_FLAG_NO_RUN_AHEAD) && e->prh.current_pass < e->prh.total_passes) {
        FrameSlot& o = e->slot[1 - e->cur];
        const size_t px = static_cast<size_t>(e->width) * e->padded_rows;
        if (o.accum.count != px * 4 || o.rgba.count != px) {
            HIP_TRY(e, o.accum.resize(px * 4));
            HIP_TRY(e, o.rgba.resize(px));
            // rows a sharded engine pads its stripes with are never written by a kernel: callers that read the
            // slot must not see what the allocator left there
            if (px) HIP_TRY(e, hipMemsetAsync(o.accum.ptr, 0, px * 16, e->stream));
            if (px) HIP_TRY(e, hipMemsetAsync(o.rgba.ptr, 0, px * 4, e->stream));
        }
        const uint32_t n2 = std::min(per, e->prh.total_passes - e->prh.current_pass);
        rc = dispatch(e, e->prh.current_pass, n2, e->cur, 1 - e->cur);
        if (rc) return rc;
        e->spec_valid = true;
        e->spec_first = e->prh.current_pass;
        e->spec_n = n2;
    }
    return RB_OK;
}

int iter_next_locked(rb_engine* e, uint8_t* rgba_out) {
    if (!(e->prh.current_pass < e->prh.total_passes))
        return fail(e, RB_ERR_NO_MORE_FRAMES, "No more frames available");  // lib.rs:170-177
    const bool multiproc = e->net.nranks > 1;
    if (!rgba_out && !(multiproc && e->net.rank != 0)) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    int rc = iter_advance(e, e->iter_passes_per_frame);
    if (rc) return rc;
    if (multiproc) {
        std::string why;
        if (rb::gather_process(e->net, e->slot[e->cur].rgba.ptr, e->width, e->height, e->padded_rows,
                               e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows, e->copy_stream, e->slot[e->cur].done,
                               rgba_out, why))
            return fail(e, RB_ERR_DEVICE, "%s", why.c_str());
        return RB_OK;
    }
    return read_slot_rgba(e, e->cur, rgba_out);  // lib.rs:205
}

rb_engine* create_single(const rb_config* cfg, const rb_options& opt) {
    if (opt.shard_count > 1 && opt.shard_rank >= opt.shard_count) {
        fail(nullptr, RB_ERR_INVALID_OPTIONS, "shard_rank %u >= shard_count %u", opt.shard_rank, opt.shard_count);
        return nullptr;
    }
    if (opt.kernel > RB_KERNEL_STREAM) { fail(nullptr, RB_ERR_INVALID_OPTIONS, "unknown kernel %u", opt.kernel); return nullptr; }
    int dev = opt.device;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) { fail(nullptr, RB_ERR_DEVICE, "no HIP device available"); return nullptr; }
    }
    if (hipSetDevice(dev) != hipSuccess) { fail(nullptr, RB_ERR_DEVICE, "hipSetDevice(%d) failed", dev); return nullptr; }
    rb_engine* e = new rb_engine();
    e->device = dev;
    e->opt = opt;
    // RB_REFERENCE_WALK=1 in the environment: every engine of this process walks meshes exactly as shader.wgsl:282-392 does,
    // whatever the host program's flags say -- the escape hatch from the culled walks (whose exactness is derived and fuzzed,
    // DESIGN.md section 4.2) that needs no rebuild of the host
    if (const char* rw = std::getenv("RB_REFERENCE_WALK"); rw && rw[0] == '1')
        e->opt.flags = (e->opt.flags & ~(RB_FLAG_FAST_BVH | RB_FLAG_DEVICE_BVH | RB_FLAG_DEVICE_LBVH | RB_FLAG_HOST_BVH | RB_FLAG_CHUNK_WALK |
                                         RB_FLAG_SKIP_NEAR_DEGENERATE)) | RB_FLAG_REFERENCE_WALK;
    auto bail = [&](const char* what, hipError_t st) -> rb_engine* {
        fail(nullptr, RB_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(st));
        rb_destroy(e);
        return nullptr;
    };
    hipError_t st;
    if ((st = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", st);
    if ((st = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", st);
    if ((st = hipEventCreate(&e->ev_begin)) != hipSuccess) return bail("hipEventCreate", st);
    if ((st = hipEventCreate(&e->ev_end)) != hipSuccess) return bail("hipEventCreate", st);
    for (FrameSlot& s : e->slot)
        if ((st = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", st);
    if ((st = e->counters.resize(rb::C_COUNT)) != hipSuccess) return bail("hipMalloc(counters)", st);
    if ((st = e->queue.resize(rb::kQueueWords)) != hipSuccess) return bail("hipMalloc(queue)", st);
    if ((st = hipMemsetAsync(e->counters.ptr, 0, sizeof(unsigned long long) * rb::C_COUNT, e->stream)) != hipSuccess)
        return bail("hipMemset(counters)", st);
    // sRGB -> linear table for sample_texture's pow(c, 2.2) (shader.wgsl:185-190)
    float lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = powf(static_cast<float>(i) / 255.0f, 2.2f);
    if ((st = e->srgb_lut.resize(256)) != hipSuccess) return bail("hipMalloc(lut)", st);
    if ((st = hipMemcpy(e->srgb_lut.ptr, lut, sizeof lut, hipMemcpyHostToDevice)) != hipSuccess) return bail("hipMemcpy(lut)", st);
    // ProgressiveRenderHelper::new (gpu_wrapper.rs:38-45); SAMPLES_PER_PASS = 1 (:12)
    const rb_uniforms* u = static_cast<const rb_uniforms*>(cfg->uniforms.ptr);
    e->prh.samples_per_pass = 1;
    e->prh.total_samples = u->total_samples;
    e->prh.total_passes = u->total_samples;
    e->prh.current_pass = 0;
    return e;
}

bool check_create(const rb_config* cfg) {
    g_create_error.clear();
    if (!cfg) { fail(nullptr, RB_ERR_NULL_ARGUMENT, "config is NULL"); return false; }
    if (check_fields(nullptr, cfg)) return false;
    // GpuBuffers::new panics unless these are Create (buffers.rs:74-97)
    if (validate_init(nullptr, cfg)) return false;
    return true;
}

rb_engine* create_impl(const rb_config* cfg, const rb_options* opt_in) {
    if (!check_create(cfg)) return nullptr;
    rb_options opt{};
    opt.device = -1;
    if (opt_in) opt = *opt_in;
    return create_single(cfg, opt);
}

// ------------------------------------------------------------------ several devices, one handle ----
bool is_group(const rb_engine* e) { return !e->parts.empty(); }

void copy_error(rb_engine* g, const rb_engine* part) {
    std::string msg;
    {
        std::lock_guard<std::mutex> l(part->err_mu);
        msg = part->error;
    }
    std::lock_guard<std::mutex> l(g->err_mu);
    g->error = "device " + std::to_string(part->device) + ": " + msg;
}

#define PART_TRY(g, part, call)            \
    do {                                   \
        set_device(part);                  \
        const int _rc = (call);            \
        if (_rc) {                         \
            copy_error((g), (part));       \
            return _rc;                    \
        }                                  \
    } while (0)

int group_update(rb_engine* g, const rb_config* cfg) {
    for (auto& p : g->parts) PART_TRY(g, p.get(), update_locked(p.get(), cfg));
    rb_engine* p0 = g->parts[0].get();
    g->width = p0->width;
    g->height = p0->height;
    g->have_uniforms = p0->have_uniforms;
    g->initialized = p0->initialized;
    g->prh = p0->prh;
    return RB_OK;
}

// the one exchange step: every part's RGBA8 stripes to the root device, de-interleaved there, then read back
int group_deliver(rb_engine* g, uint8_t* rgba_out, bool fold_timing) {
    std::vector<rb::GatherSource> src;
    for (auto& p : g->parts) src.push_back(rb::GatherSource{p->device, p->copy_stream, p->slot[p->cur].done, p->slot[p->cur].rgba.ptr});
    rb_engine* p0 = g->parts[0].get();
    const uint32_t sr = p0->opt.stripe_rows ? p0->opt.stripe_rows : rb::kDefaultStripeRows;
    std::string why;
    if (rb::gather_group(g->net, src, p0->width, p0->height, p0->padded_rows, sr, rgba_out, why))
        return fail(g, RB_ERR_DEVICE, "%s", why.c_str());
    // (the iterator folds a group's timing when it commits it: waiting for the events here would wait for the pass
    // that has just been started ahead)
    if (fold_timing)
        for (auto& p : g->parts) PART_TRY(g, p.get(), accumulate_timing(p.get()));
    return RB_OK;
}

int group_render(rb_engine* g, uint8_t* rgba_out) {
    if (!rgba_out) return fail(g, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    for (auto& p : g->parts) PART_TRY(g, p.get(), render_async(p.get()));  // all devices render concurrently
    g->prh = g->parts[0]->prh;
    return group_deliver(g, rgba_out, true);
}

int group_iter_next(rb_engine* g, uint8_t* rgba_out) {
    rb_engine* p0 = g->parts[0].get();
    if (!(p0->prh.current_pass < p0->prh.total_passes)) return fail(g, RB_ERR_NO_MORE_FRAMES, "No more frames available");
    if (!rgba_out) return fail(g, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    for (auto& pp : g->parts) PART_TRY(g, pp.get(), iter_advance(pp.get(), g->iter_passes_per_frame));   // every part runs ahead
    g->prh = p0->prh;
    return group_deliver(g, rgba_out, false);
}

}  // namespace

// ============================================================== C ABI ======
extern "C" {

rb_engine* rb_create(const rb_config* cfg) { return create_impl(cfg, nullptr); }
rb_engine* rb_create_ex(const rb_config* cfg, const rb_options* opt) { return create_impl(cfg, opt); }

rb_engine* rb_create_multi(const rb_config* cfg, const rb_options* opt_in, const int32_t* devices, uint32_t n_devices) {
    if (!check_create(cfg)) return nullptr;
    if (!devices || n_devices == 0 || n_devices > 64) {
        fail(nullptr, RB_ERR_INVALID_OPTIONS, "rb_create_multi needs 1..64 devices");
        return nullptr;
    }
    rb_options opt{};
    if (opt_in) opt = *opt_in;
    if (opt.shard_count > 1) {
        fail(nullptr, RB_ERR_INVALID_OPTIONS, "rb_create_multi shards by itself: leave shard_rank / shard_count zero");
        return nullptr;
    }
    std::unique_ptr<rb_engine> g(new rb_engine());
    g->opt = opt;
    g->device = devices[0];
    for (uint32_t r = 0; r < n_devices; ++r) {
        rb_options po = opt;
        po.device = devices[r];
        po.shard_rank = r;
        po.shard_count = n_devices;
        rb_engine* p = create_single(cfg, po);
        if (!p) {   // g_create_error is set
            rb_destroy(g.release());
            return nullptr;
        }
        g->parts.emplace_back(p);
    }
    std::vector<int> devs(devices, devices + n_devices);
    std::string why;
    if (rb::gather_init_group(g->net, devs, (opt.flags & RB_FLAG_GATHER_PEER_COPY) != 0u, why)) {
        rb_destroy(g.release());
        fail(nullptr, RB_ERR_DEVICE, "%s", why.c_str());
        return nullptr;
    }
    rb_engine* out = g.release();
    return out;
}

int rb_comm_available(void) {
    std::string why;
    if (rb::gather_available(why)) {
        g_create_error = why;
        return RB_ERR_DEVICE;
    }
    return RB_OK;
}

int rb_comm_unique_id(uint8_t id_out[RB_COMM_ID_BYTES]) {
    if (!id_out) return RB_ERR_NULL_ARGUMENT;
    std::string why;
    if (rb::gather_unique_id(id_out, why)) {
        g_create_error = why;
        return RB_ERR_DEVICE;
    }
    return RB_OK;
}

int rb_comm_init_rank(rb_engine* e, const uint8_t id[RB_COMM_ID_BYTES], uint32_t rank, uint32_t nranks) {
    if (!e || !id) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) return fail(e, RB_ERR_INVALID_OPTIONS, "rb_comm_init_rank is for single-device engines");
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    if (nranks != sc || rank != (sc > 1 ? e->opt.shard_rank : 0u))
        return fail(e, RB_ERR_INVALID_OPTIONS, "communicator rank %u of %u does not match shard %u of %u", rank, nranks,
                    e->opt.shard_rank, sc);
    set_device(e);
    std::string why;
    if (rb::gather_init_rank(e->net, e->device, id, rank, nranks, why)) return fail(e, RB_ERR_DEVICE, "%s", why.c_str());
    return RB_OK;
}

int rb_comm_info(rb_engine* e, uint32_t* rccl_ranks, uint32_t* rccl_rank, float* last_gather_ms) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    std::string why;
    if (rb::gather_comm_info(e->net, rccl_ranks, rccl_rank, why)) return fail(e, RB_ERR_DEVICE, "%s", why.c_str());
    if (last_gather_ms) *last_gather_ms = e->net.last_ms;
    return RB_OK;
}

void rb_destroy(rb_engine* e) {
    if (!e) return;
    if (is_group(e)) {
        for (auto& p : e->parts) {
            set_device(p.get());
            if (p->stream) (void)hipStreamSynchronize(p->stream);
        }
        rb::gather_destroy(e->net);
        for (auto& p : e->parts) rb_destroy(p.release());
        delete e;
        return;
    }
    set_device(e);
    // every launch, copy and event record of this engine was queued on its one stream: when that has
    // drained nothing on the device refers to the buffers, events or communicator any more
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) {
        (void)hipStreamSynchronize(e->copy_stream);
        (void)hipStreamDestroy(e->copy_stream);
    }
    rb::gather_destroy(e->net);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    for (FrameSlot& s : e->slot)
        if (s.done) (void)hipEventDestroy(s.done);
    for (hipEvent_t x : e->ev_pool) (void)hipEventDestroy(x);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* rb_last_error(const rb_engine* e) {
    if (!e) return g_create_error.c_str();
    // a copy per calling thread: another thread's failing call may replace the engine's text at any moment
    // (the reference's GUI polls from its own thread), and a pointer into that string would dangle
    thread_local std::string mine;
    {
        std::lock_guard<std::mutex> g(e->err_mu);
        mine = e->error;
    }
    return mine.c_str();
}

int rb_update(rb_engine* e, const rb_config* cfg) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) return group_update(e, cfg);
    set_device(e);
    return update_locked(e, cfg);
}

int rb_render(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) return group_render(e, rgba_out);
    set_device(e);
    return render_locked(e, rgba_out);
}

int rb_render_config(rb_engine* e, const rb_config* cfg, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        const int rc = group_update(e, cfg);
        return rc ? rc : group_render(e, rgba_out);
    }
    set_device(e);
    int rc = update_locked(e, cfg);
    if (rc) return rc;
    return render_locked(e, rgba_out);
}

int rb_iter_begin(rb_engine* e, const rb_config* cfg) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        int rc = group_update(e, cfg);
        if (rc) return rc;
        for (auto& p : e->parts) {
            PART_TRY(e, p.get(), require_ready(p.get()));
            p->prh.current_pass = 0;
            p->iter_initialized = false;
        }
        e->prh = e->parts[0]->prh;
        return RB_OK;
    }
    set_device(e);
    int rc = update_locked(e, cfg);
    if (rc) return rc;
    rc = require_ready(e);
    if (rc) return rc;
    e->prh.current_pass = 0;       // lib.rs:91
    e->iter_initialized = false;   // RaytracerFrameIterator::new (lib.rs:144-150)
    e->spec_valid = false;
    return RB_OK;
}

int rb_iter_has_next(rb_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lock(e->mu);
    const rb_engine* s = is_group(e) ? e->parts[0].get() : e;
    return s->prh.current_pass < s->prh.total_passes ? 1 : 0;  // lib.rs:153-156
}

int rb_iter_next(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) return group_iter_next(e, rgba_out);
    set_device(e);
    return iter_next_locked(e, rgba_out);
}

void rb_iter_destroy(rb_engine* e) { (void)e; }  // lib.rs:231-233: logs only

int rb_iter_set_passes_per_frame(rb_engine* e, uint32_t n) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    e->iter_passes_per_frame = n;
    return RB_OK;
}

int rb_get_size(const rb_engine* e, uint32_t* width, uint32_t* height) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(const_cast<rb_engine*>(e)->mu);
    if (!e->have_uniforms) return fail(e, RB_ERR_UNIFORMS_NOT_INITIALIZED, "Uniforms must be initialized");  // gpu_wrapper.rs:313,321
    if (width) *width = e->width;
    if (height) *height = e->height;
    return RB_OK;
}

int rb_clear(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        for (auto& p : e->parts) {
            PART_TRY(e, p.get(), require_ready(p.get()));
            PART_TRY(e, p.get(), clear_accum(p.get()));
        }
        return RB_OK;
    }
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    return clear_accum(e);
}

int rb_dispatch(rb_engine* e, uint32_t first_pass, uint32_t n_passes) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        for (auto& p : e->parts) {
            PART_TRY(e, p.get(), require_ready(p.get()));
            p->spec_valid = false;
            PART_TRY(e, p.get(), dispatch(p.get(), first_pass, n_passes, p->cur, p->cur));
        }
        return RB_OK;
    }
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    e->spec_valid = false;
    return dispatch(e, first_pass, n_passes, e->cur, e->cur);
}

int rb_reserve(rb_engine* e, uint32_t n_passes) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    auto one = [&](rb_engine* p) {
        set_device(p);
        int rc = require_ready(p);
        if (!rc) rc = ensure_prepared(p);
        uint32_t chunk = 0;
        if (!rc) rc = reserve_colors(p, n_passes, &chunk);
        if (!rc && hipStreamSynchronize(p->stream) != hipSuccess) rc = fail(p, RB_ERR_DEVICE, "synchronise failed");
        return rc;
    };
    if (is_group(e)) {
        for (auto& p : e->parts) PART_TRY(e, p.get(), one(p.get()));
        return RB_OK;
    }
    return one(e);
}

int rb_sync(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        for (auto& p : e->parts) {
            set_device(p.get());
            HIP_TRY(e, hipStreamSynchronize(p->stream));
        }
        return RB_OK;
    }
    set_device(e);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RB_OK;
}

int rb_read_rgba(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        if (!rgba_out) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
        return group_deliver(e, rgba_out, false);
    }
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    return read_rgba(e, rgba_out);
}

int rb_read_accumulation(rb_engine* e, float* accum_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (!accum_out) return fail(e, RB_ERR_NULL_ARGUMENT, "accum_out is NULL");
    if (is_group(e)) {
        // debugging / checkpoint path: every part's rows through the host, in image order
        rb_engine* p0 = e->parts[0].get();
        const uint32_t w = p0->width, h = p0->height, sr = p0->opt.stripe_rows ? p0->opt.stripe_rows : rb::kDefaultStripeRows;
        const uint32_t n = static_cast<uint32_t>(e->parts.size());
        std::vector<float> tmp(static_cast<size_t>(w) * p0->padded_rows * 4);
        for (uint32_t r = 0; r < n; ++r) {
            rb_engine* p = e->parts[r].get();
            PART_TRY(e, p, require_ready(p));
            HIP_TRY(e, hipStreamSynchronize(p->stream));
            HIP_TRY(e, hipMemcpy(tmp.data(), p->slot[p->cur].accum.ptr, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (uint32_t lr = 0; lr < p->padded_rows; ++lr) {
                const uint32_t y = ((lr / sr) * n + r) * sr + lr % sr;
                if (y < h) std::memcpy(accum_out + static_cast<size_t>(y) * w * 4, tmp.data() + static_cast<size_t>(lr) * w * 4, static_cast<size_t>(w) * 16);
            }
        }
        return RB_OK;
    }
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    const uint32_t rows = (e->opt.shard_count > 1) ? e->padded_rows : e->height;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(accum_out, e->slot[e->cur].accum.ptr, static_cast<size_t>(e->width) * rows * 16, hipMemcpyDeviceToHost));
    return RB_OK;
}

void* rb_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void rb_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int rb_device_rgba(rb_engine* e, void** d_ptr, size_t* bytes) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {  // the assembled frame on the root device (valid after a render / iterator step)
        if (d_ptr) *d_ptr = rb::gather_frame_ptr(e->net);
        if (bytes) *bytes = static_cast<size_t>(e->width) * e->height * 4;
        return RB_OK;
    }
    if (d_ptr) *d_ptr = e->slot[e->cur].rgba.ptr;
    if (bytes) *bytes = static_cast<size_t>(e->width) * e->padded_rows * 4;
    return RB_OK;
}

int rb_local_rows(const rb_engine* e, uint32_t* rows, uint32_t* padded_rows) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    if (is_group(e)) {  // the handle delivers whole frames
        if (rows) *rows = e->height;
        if (padded_rows) *padded_rows = e->height;
        return RB_OK;
    }
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    const uint32_t sr = e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows;
    uint32_t owned = e->height;
    if (sc > 1) {
        owned = 0;
        const uint32_t stripes = (e->height + sr - 1) / sr;
        for (uint32_t s = e->opt.shard_rank; s < stripes; s += sc) owned += std::min(sr, e->height - s * sr);
    }
    if (rows) *rows = owned;
    if (padded_rows) *padded_rows = e->padded_rows;
    return RB_OK;
}

int rb_global_row(const rb_engine* e, uint32_t local_row, uint32_t* global_row) {
    if (!e || !global_row) return RB_ERR_NULL_ARGUMENT;
    if (is_group(e)) { *global_row = local_row; return RB_OK; }
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    const uint32_t sr = e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows;
    if (sc == 1) { *global_row = local_row; return RB_OK; }
    *global_row = ((local_row / sr) * sc + e->opt.shard_rank) * sr + local_row % sr;
    return RB_OK;
}

int rb_shard_layout(uint32_t height, uint32_t shard_rank, uint32_t shard_count, uint32_t stripe_rows,
                    uint32_t* owned_rows, uint32_t* padded_rows) {
    const uint32_t sc = shard_count > 1 ? shard_count : 1;
    const uint32_t sr = stripe_rows ? stripe_rows : rb::kDefaultStripeRows;
    if (shard_rank >= sc) return RB_ERR_INVALID_OPTIONS;
    uint32_t owned = height, padded = height;
    if (sc > 1) {
        const uint32_t stripes = (height + sr - 1) / sr;
        padded = ((stripes + sc - 1) / sc) * sr;
        owned = 0;
        for (uint32_t s = shard_rank; s < stripes; s += sc) owned += std::min(sr, height - s * sr);
    }
    if (owned_rows) *owned_rows = owned;
    if (padded_rows) *padded_rows = padded;
    return RB_OK;
}

uint32_t rb_shard_global_row(uint32_t shard_rank, uint32_t shard_count, uint32_t stripe_rows, uint32_t local_row) {
    const uint32_t sc = shard_count > 1 ? shard_count : 1;
    const uint32_t sr = stripe_rows ? stripe_rows : rb::kDefaultStripeRows;
    if (sc == 1) return local_row;
    return ((local_row / sr) * sc + shard_rank) * sr + local_row % sr;
}

static int part_stats(rb_engine* e, rb_stats* out);

int rb_get_stats(rb_engine* e, rb_stats* out) {
    if (!e || !out) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {  // work counters add up; the devices run side by side, so times are the slowest part's
        rb_stats sum{};
        for (auto& p : e->parts) {
            rb_stats s{};
            PART_TRY(e, p.get(), part_stats(p.get(), &s));
            sum.segments += s.segments; sum.paths += s.paths; sum.nodes_popped += s.nodes_popped;
            sum.tris_tested += s.tris_tested; sum.spheres_tested += s.spheres_tested;
            sum.lights_tested += s.lights_tested; sum.mesh_hits += s.mesh_hits;
            sum.launches = std::max(sum.launches, s.launches);
            sum.kernel_ms = std::max(sum.kernel_ms, s.kernel_ms);
            sum.trace_ms = std::max(sum.trace_ms, s.trace_ms);
            sum.accumulate_ms = std::max(sum.accumulate_ms, s.accumulate_ms);
        }
        *out = sum;
        return RB_OK;
    }
    set_device(e);
    return part_stats(e, out);
}

static int part_stats(rb_engine* e, rb_stats* out) {
    unsigned long long c[rb::C_COUNT];
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    {
        const int rc = accumulate_timing(e);
        if (rc) return rc;
    }
    HIP_TRY(e, hipMemcpy(c, e->counters.ptr, sizeof c, hipMemcpyDeviceToHost));
    e->stats.segments = c[rb::C_SEGMENTS];
    e->stats.paths = c[rb::C_PATHS];
    e->stats.nodes_popped = c[rb::C_NODES];
    e->stats.tris_tested = c[rb::C_TRIS];
    e->stats.spheres_tested = c[rb::C_SPHERES];
    e->stats.lights_tested = c[rb::C_LIGHTS];
    e->stats.mesh_hits = c[rb::C_MESH_HITS];
    *out = e->stats;
    return RB_OK;
}

static int part_reset_stats(rb_engine* e);

int rb_reset_stats(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        for (auto& p : e->parts) PART_TRY(e, p.get(), part_reset_stats(p.get()));
        return RB_OK;
    }
    set_device(e);
    return part_reset_stats(e);
}

static int part_reset_stats(rb_engine* e) {
    {   // a launch group whose events have not been read yet belongs to the period that ends here
        const int rc = accumulate_timing(e);
        if (rc) return rc;
    }
    HIP_TRY(e, hipMemsetAsync(e->counters.ptr, 0, sizeof(unsigned long long) * rb::C_COUNT, e->stream));
    e->stats = rb_stats{};
    return RB_OK;
}

int rb_last_dispatch_ms(rb_engine* e, float* ms) {
    if (!e || !ms) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (is_group(e)) {
        *ms = 0.0f;
        for (auto& p : e->parts) {
            PART_TRY(e, p.get(), accumulate_timing(p.get()));
            *ms = std::max(*ms, p->last_dispatch_ms);
        }
        return RB_OK;
    }
    set_device(e);
    int rc = accumulate_timing(e);
    if (rc) return rc;
    *ms = e->last_dispatch_ms;
    return RB_OK;
}

int rb_bvh_build(const rb_gpu_triangle* tris, size_t n_tris, rb_bvh_node* nodes_out, size_t nodes_capacity,
                 size_t* n_nodes, uint32_t* indices_out) {
    if (!n_nodes || (n_tris > 0 && !tris)) return RB_ERR_NULL_ARGUMENT;
    std::vector<rb_bvh_node> nodes;
    std::vector<uint32_t> indices;
    rb::bvh_build(tris, n_tris, nodes, indices);
    *n_nodes = nodes.size();
    if (!nodes_out) return RB_OK;
    if (nodes_capacity < nodes.size()) return RB_ERR_INVALID_BVH;
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rb_bvh_node));
    if (indices_out) std::memcpy(indices_out, indices.data(), indices.size() * sizeof(uint32_t));
    return RB_OK;
}

const char* rb_version(void) { return "renderbaby-hip 0.3 (gfx950)"; }

const char* rb_last_kernel_name(const rb_engine* e) {
    if (!e) return "";
    return is_group(e) ? e->parts[0]->last_kernel_name : e->last_kernel_name;
}

const char* rb_fast_bvh_builder(const rb_engine* e, float* build_ms) {
    if (e && is_group(e)) e = e->parts[0].get();
    if (build_ms) *build_ms = e ? e->fast_build_ms : 0.0f;
    return (e && e->fast_ready) ? e->fast_builder : "";
}

const char* rb_sphere_tree_builder(const rb_engine* e, float* build_ms) {
    if (e && is_group(e)) e = e->parts[0].get();
    if (build_ms) *build_ms = (e && e->sph_bvh) ? e->sph_build_ms : 0.0f;
    return (e && e->sph_bvh) ? e->sph_builder : "";
}

int rb_device_name(int device, char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return RB_ERR_NULL_ARGUMENT;
    hipDeviceProp_t prop;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return RB_ERR_DEVICE;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RB_ERR_DEVICE;
    snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return RB_OK;
}

// Test hook: exhaustive device check of the fast reciprocal (all 2^23 significands, both signs)
// at one biased exponent.  out16[0] = number of mismatches, out16[1..15] = offending bit patterns.
int rb_debug_rcp_exhaustive(uint32_t biased_exponent, uint32_t* out16) {
    if (!out16) return RB_ERR_NULL_ARGUMENT;
    uint32_t* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 64) != hipSuccess) return RB_ERR_DEVICE;
    (void)hipMemset(d, 0, 64);
    int rc = rb::launch_rcp_exhaustive(biased_exponent, d, nullptr);
    hipError_t st = hipDeviceSynchronize();
    (void)hipMemcpy(out16, d, 64, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (rc || st != hipSuccess) ? RB_ERR_DEVICE : RB_OK;
}

// Test hook: device check of the fast exact division over denominators [b_begin, b_begin+b_count)
// x numerators [a_begin, a_begin+a_count) (significands; biased exponents ea / eb).
// out16[0] = mismatch count, then up to 7 (a, b) bit-pattern pairs.
int rb_debug_div_exhaustive(uint32_t b_begin, uint32_t b_count, uint32_t ea, uint32_t eb, uint32_t a_begin,
                            uint32_t a_count, unsigned long long* out16) {
    if (!out16) return RB_ERR_NULL_ARGUMENT;
    unsigned long long* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), 128) != hipSuccess) return RB_ERR_DEVICE;
    (void)hipMemset(d, 0, 128);
    int rc = rb::launch_div_exhaustive(b_begin, b_count, ea, eb, a_begin, a_count, d, nullptr);
    hipError_t st = hipDeviceSynchronize();
    (void)hipMemcpy(out16, d, 128, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (rc || st != hipSuccess) ? RB_ERR_DEVICE : RB_OK;
}

namespace {
void chunk_tree_census(const rb::ChunkTree& t, uint64_t out6[6]) {
    uint64_t chunks = 0, unbounded = 0;
    for (const rb::ChunkNode& c : t.nodes) {
        chunks += ((c.lref != rb::kChunkNone && (c.lref & rb::kChunkLeaf)) ? 1 : 0) + ((c.rref != rb::kChunkNone && (c.rref & rb::kChunkLeaf)) ? 1 : 0);
        unbounded += ((c.lref != rb::kChunkNone && (c.lfac >> 16) == 0x7F80u) ? 1 : 0) + ((c.rref != rb::kChunkNone && (c.rfac >> 16) == 0x7F80u) ? 1 : 0);
    }
    out6[0] = 1;
    out6[1] = t.nodes.size();
    out6[2] = t.pos_slot.size();
    out6[3] = t.depth;
    out6[4] = chunks;
    out6[5] = unbounded;
}
}  // namespace

// Test aid (host only): the chunked walk's tree for a mesh and a caller tree, with its invariants checked.
int rb_debug_chunk_tree(const rb_gpu_triangle* tris, size_t n_tris, const rb_bvh_node* nodes, size_t n_nodes, const uint32_t* indices,
                        size_t n_indices, uint64_t out6[6]) {
    if (!tris || !nodes || !indices || !out6) return RB_ERR_NULL_ARGUMENT;
    if (n_tris >= (1ull << 31) || n_nodes >= (1ull << 31) || n_indices >= (1ull << 31)) return RB_ERR_INVALID_BVH;
    std::string why;
    if (!rb::bvh_validate(nodes, static_cast<uint32_t>(n_nodes), rb::kStackDepth, why, nullptr)) return fail(nullptr, RB_ERR_INVALID_BVH, "%s", why.c_str());
    rb::ChunkTree t;
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (!rb::chunk_tree_build(tris, static_cast<uint32_t>(n_tris), indices, static_cast<uint32_t>(n_indices), nodes,
                              static_cast<uint32_t>(n_nodes), rb::kStackDepth, t))
        return RB_OK;   // out6[0] == 0: this tree is left to another walk
    if (!rb::chunk_tree_check(t, tris, static_cast<uint32_t>(n_tris), indices, static_cast<uint32_t>(n_indices), rb::kStackDepth, why))
        return fail(nullptr, RB_ERR_INVALID_BVH, "chunk tree: %s", why.c_str());
    chunk_tree_census(t, out6);
    return RB_OK;
}

int rb_debug_engine_chunk_tree(rb_engine* e, uint64_t out6[6]) {
    if (!e || !out6) return RB_ERR_NULL_ARGUMENT;
    if (is_group(e)) e = e->parts[0].get();
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    for (int i = 0; i < 6; ++i) out6[i] = 0;
    if (!e->chunk_ready) return RB_OK;
    rb::ChunkTree t;
    const size_t n = e->chunk_rank_slot.count;
    t.nodes.resize(e->chunk_n_nodes);
    t.pos_slot.resize(n);
    t.pos_rank.resize(n);
    t.rank_slot.resize(n);
    t.root = e->chunk_root;
    t.depth = e->chunk_depth;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(t.nodes.data(), e->chunk_nodes.ptr, sizeof(rb::ChunkNode) * t.nodes.size(), hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(t.pos_slot.data(), e->chunk_pos_slot.ptr, 4u * n, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(t.pos_rank.data(), e->chunk_pos_rank.ptr, 4u * n, hipMemcpyDeviceToHost));
    HIP_TRY(e, hipMemcpy(t.rank_slot.data(), e->chunk_rank_slot.ptr, 4u * n, hipMemcpyDeviceToHost));
    std::string why;
    {
        const int hrc = ensure_host_mesh(e);
        if (hrc) return hrc;
    }
    const uint32_t n_tris = std::min<uint32_t>(e->prep_tri_count, static_cast<uint32_t>(e->host_tris.size()));
    if (!rb::chunk_tree_check(t, e->host_tris.data(), n_tris, e->host_indices.data(), static_cast<uint32_t>(e->host_indices.size()), rb::kStackDepth, why))
        return fail(e, RB_ERR_INVALID_BVH, "chunk tree (%s builder): %s", e->chunk_builder, why.c_str());
    chunk_tree_census(t, out6);
    return RB_OK;
}

const char* rb_chunk_tree_builder(const rb_engine* e, float* build_ms) {
    if (e && is_group(e)) e = e->parts[0].get();
    if (build_ms) *build_ms = (e && e->chunk_ready) ? e->chunk_build_ms : 0.0f;
    return (e && e->chunk_ready) ? e->chunk_builder : "";
}

int rb_measure_l1_gather(int32_t device, uint64_t table_bytes, double* accesses_per_s) {
    if (!accesses_per_s) return RB_ERR_NULL_ARGUMENT;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return RB_ERR_DEVICE;
    *accesses_per_s = rb::measure_l1_gather(table_bytes ? static_cast<size_t>(table_bytes) : (2u << 20), 512u);
    return *accesses_per_s > 0.0 ? RB_OK : RB_ERR_DEVICE;
}

int rb_debug_walk_profile(uint64_t out64[64], int reset) {
    if (!out64) return RB_ERR_NULL_ARGUMENT;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    if (hipDeviceSynchronize() != hipSuccess) return RB_ERR_DEVICE;
    return rb::debug_walk_profile(reinterpret_cast<unsigned long long*>(out64), reset) == 0 ? RB_OK : RB_ERR_DEVICE;
}

// Debug hook for tests/test_gpu_parity.py: device /, sqrt, normalize, u32->f32, min/max, dot.
int rb_debug_math(const float* a, const float* b, float* out8n, uint32_t n) {
    if (!a || !b || !out8n) return RB_ERR_NULL_ARGUMENT;
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&da), n * 4) != hipSuccess) return RB_ERR_DEVICE;
    if (hipMalloc(reinterpret_cast<void**>(&db), n * 4) != hipSuccess) return RB_ERR_DEVICE;
    if (hipMalloc(reinterpret_cast<void**>(&dout), static_cast<size_t>(n) * 32) != hipSuccess) return RB_ERR_DEVICE;
    (void)hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
    int rc = rb::launch_debug_math(da, db, dout, n, nullptr);
    hipError_t st = hipDeviceSynchronize();
    (void)hipMemcpy(out8n, dout, static_cast<size_t>(n) * 32, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    return (rc || st != hipSuccess) ? RB_ERR_DEVICE : RB_OK;
}

}  // extern "C"
