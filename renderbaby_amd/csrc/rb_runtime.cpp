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
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rb_internal.hpp"

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
};

}  // namespace

struct rb_engine {
    std::mutex mu;
    mutable std::string error;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pool;   // per launch chunk: begin, after-trace, end
    uint32_t ev_used = 0;
    const char* last_kernel_name = "";
    rb_options opt{};

    bool initialized = false;        // GpuWrapper::initialized (gpu_wrapper.rs:69,117)
    bool have_uniforms = false;      // last update carried Create/Update uniforms (:303-329)
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
    DevBuf<float> accum;
    DevBuf<uint32_t> out_rgba;
    DevBuf<unsigned long long> counters;
    DevBuf<uint32_t> queue;
    DevBuf<rb::SphereNode> fast_nodes; // opt-in fast triangle tree (RB_FLAG_FAST_BVH)
    DevBuf<rb::PrepTri> fast_tris;
    DevBuf<uint32_t> fast_slots, slot_meta, ref_parent, stack_overflow;
    uint32_t fast_root = 0, fast_depth = 0;
    float fast_margin = 0.0f, fast_root_amax = 0.0f;
    float fast_bmin[3] = {0, 0, 0}, fast_bmax[3] = {0, 0, 0};
    bool fast_ready = false;
    float fast_build_ms = 0.0f;
    const char* fast_builder = "";  // which builder produced the fast tree ("host-sah" / "device-lbvh")
    std::vector<rb_gpu_triangle> host_tris;  // kept only when RB_FLAG_FAST_BVH is set
    std::vector<uint32_t> host_indices;
    DevBuf<rb::SphereNode> sph_nodes;  // own sphere acceleration structure (n_spheres > threshold)
    DevBuf<float> sph_leaf;
    DevBuf<uint32_t> sph_id;
    uint32_t sph_root = 0, sph_depth = 0;
    float sph_bmin[3] = {0, 0, 0}, sph_bmax[3] = {0, 0, 0};
    bool sph_bvh = false;
    DevBuf<float> colors;            // RB_KERNEL_STREAM: float4 per (pixel, sample) of one launch chunk
    uint32_t bvh_stack = 0;          // traversal-stack entries the current tree needs

    std::vector<rb_bvh_node> host_nodes;  // kept for validation when nodes/indices change separately
    uint32_t width = 0, height = 0, local_rows = 0, padded_rows = 0;

    rb_stats stats{};
    float last_dispatch_ms = 0.0f;
    uint32_t last_launches = 0;
    bool timing_pending = false;
    uint32_t max_mesh_index = 0;  // over the uploaded triangles
};

namespace {

int fail(rb_engine* e, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->error = buf; else g_create_error = buf;
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

// grow_resolution -- buffers.rs:171-180 (+ the stripe geometry of the sharded case)
int resize_frame(rb_engine* e, uint32_t w, uint32_t h) {
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    const uint32_t sr = e->opt.stripe_rows ? e->opt.stripe_rows : rb::kDefaultStripeRows;
    uint32_t local = h, padded = h;
    if (sc > 1) {
        const uint32_t stripes = (h + sr - 1) / sr;
        const uint32_t per_rank = (stripes + sc - 1) / sc;  // equal on every rank (padded)
        padded = per_rank * sr;
        uint32_t owned = 0;  // stripes this rank renders
        for (uint32_t s = e->opt.shard_rank; s < stripes; s += sc) owned++;
        local = owned * sr;  // the kernels additionally bound rows by global y < height
    }
    const uint64_t px = static_cast<uint64_t>(w) * padded;
    if (px >= (1ull << 31)) return fail(e, RB_ERR_INVALID_UNIFORMS, "frame of %u x %u pixels is too large", w, h);
    HIP_TRY(e, e->accum.resize(px * 4));
    HIP_TRY(e, e->out_rgba.resize(px));
    if (px) {
        HIP_TRY(e, hipMemsetAsync(e->accum.ptr, 0, px * 16, e->stream));
        HIP_TRY(e, hipMemsetAsync(e->out_rgba.ptr, 0, px * 4, e->stream));
    }
    e->width = w;
    e->height = h;
    e->local_rows = local;
    e->padded_rows = padded;
    return RB_OK;
}

int upload_textures(rb_engine* e, const rb_field& f) {
    const rb_texture* t = static_cast<const rb_texture*>(f.ptr);
    std::vector<uint32_t> data;
    std::vector<rb_texture_info> info;
    uint32_t offset = 0;
    for (size_t i = 0; i < f.count; ++i) {  // process_textures, buffers.rs:151-168
        const size_t n = static_cast<size_t>(t[i].width) * t[i].height;
        if (n > 0 && !t[i].rgba_data) return fail(e, RB_ERR_INVALID_TEXTURES, "texture %zu has no data", i);
        if (t[i].width == 0 || t[i].height == 0) return fail(e, RB_ERR_INVALID_TEXTURES, "texture %zu is empty", i);
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
// reference's linear scan (shader.wgsl:574-586) stays the rule for small counts.
int build_sphere_bvh(rb_engine* e, const rb_sphere* s, size_t n) {
    e->sph_bvh = false;
    if (n <= rb::kSphereBvhThreshold || n >= (1u << 28) || (e->opt.flags & RB_FLAG_NO_SPHERE_BVH)) return RB_OK;
    std::vector<rb::SphereNode> nodes;
    std::vector<uint32_t> order;
    rb::sphere_bvh_build(s, n, nodes, order, &e->sph_root, &e->sph_depth, e->sph_bmin, e->sph_bmax);
    if (e->sph_depth > rb::kStackDepth) return RB_OK;  // degenerate input: keep the linear scan
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
    return RB_OK;
}

// Applies one non-uniform field.  `first` = the engine's first update
// (gpu_wrapper.rs:117-163: only Create is acted on); otherwise :196-294.
int apply_field(rb_engine* e, int idx, const rb_field& f, bool first) {
    const bool bvh_field = (idx >= 5 && idx <= 7);
    bool take = false, del = false;
    if (first) {
        take = (f.change == RB_CREATE);
    } else {
        if (f.change == RB_UPDATE) take = true;
        else if (f.change == RB_DELETE) del = true;
        else if (f.change == RB_CREATE) take = bvh_field;  // "Create not allowed after initialization" except BVH (:242-280)
    }
    if (!take && !del) return RB_OK;
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
            if (e->opt.flags & RB_FLAG_FAST_BVH) e->host_indices.assign(static_cast<const uint32_t*>(src), static_cast<const uint32_t*>(src) + n);
            break;
        case 7:
            rc = upload(e, e->tris, src, n, nullptr, true);
            e->n_tris = static_cast<uint32_t>(n);
            e->prep_dirty = true;
            if (e->opt.flags & RB_FLAG_FAST_BVH) e->host_tris.assign(static_cast<const rb_gpu_triangle*>(src), static_cast<const rb_gpu_triangle*>(src) + n);
            break;
        case 8:
            if (del) { rb_field empty{RB_UPDATE, nullptr, 0}; rc = upload_textures(e, empty); }
            else rc = upload_textures(e, f);
            break;
        default: break;
    }
    return rc;
}

// Host-side checks that stand in for WGSL's robust buffer access: anything that
// would make a HIP kernel read out of bounds or loop forever is refused here.
int validate_scene(rb_engine* e, const rb_config* cfg) {
    std::string why;
    uint32_t depth = 0;
    if (has_data(cfg->bvh_triangles)) {
        const rb_gpu_triangle* t = static_cast<const rb_gpu_triangle*>(cfg->bvh_triangles.ptr);
        uint32_t mx = 0;
        for (size_t i = 0; i < cfg->bvh_triangles.count; ++i) mx = std::max(mx, t[i].mesh_index);
        e->max_mesh_index = mx;
    }
    const uint32_t n_nodes = static_cast<uint32_t>(e->host_nodes.size());
    if (n_nodes > 0) {
        if (!rb::bvh_validate(e->host_nodes.data(), n_nodes, rb::kStackDepth, why, &depth))
            return fail(e, RB_ERR_INVALID_BVH, "%s", why.c_str());
        e->bvh_stack = depth;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            const rb_bvh_node& n = e->host_nodes[i];
            if (n.primitive_count > 0 &&
                static_cast<uint64_t>(n.first_primitive) + n.primitive_count > e->n_indices)
                return fail(e, RB_ERR_INVALID_BVH, "leaf %u covers [%u, +%u) of %u bvh_indices", i, n.first_primitive,
                            n.primitive_count, e->n_indices);
        }
    }
    if (e->n_tris > 0 && e->uniforms.color_hash_enabled == 0 && e->max_mesh_index >= e->n_meshes)
        return fail(e, RB_ERR_INVALID_MESHES, "a triangle references mesh %u of %u", e->max_mesh_index, e->n_meshes);
    return RB_OK;
}

void set_device(rb_engine* e) { (void)hipSetDevice(e->device); }

int ensure_prepared(rb_engine* e) {
    if (!e->prep_dirty) return RB_OK;
    const uint32_t len = e->n_indices;  // arrayLength(&bvh_indices) >= 1
    HIP_TRY(e, e->ptris.resize(len));
    HIP_TRY(e, e->pshade.resize(len));
    int rc = rb::launch_prep_tris(e->tris.ptr, e->n_tris, e->indices.ptr, len, e->ptris.ptr, e->pshade.ptr, e->stream);
    if (rc) return fail(e, RB_ERR_DEVICE, "prep kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(rc)));
    e->prep_dirty = false;
    // ---- opt-in fast tree over the same triangles
    e->fast_ready = false;
    if ((e->opt.flags & RB_FLAG_FAST_BVH) && e->host_nodes.size() > 1 && !e->host_tris.empty() && !e->host_indices.empty()) {
        rb::FastTree ft;
        const auto t_begin = std::chrono::steady_clock::now();
        const uint32_t n_tris = static_cast<uint32_t>(e->host_tris.size()), n_idx = static_cast<uint32_t>(e->host_indices.size());
        const uint32_t n_nodes = static_cast<uint32_t>(e->host_nodes.size());
        bool built = false;
        e->fast_builder = "";
        if (e->opt.flags & RB_FLAG_DEVICE_BVH) {
            // reference-order metadata on the host (one pass over the caller's tree), the tree on the device
            if (rb::fast_bvh_prepare(n_tris, e->host_indices.data(), n_idx, e->host_nodes.data(), n_nodes, ft) &&
                ft.slots.size() >= 1024) {
                const uint32_t n = static_cast<uint32_t>(ft.slots.size());
                DevBuf<uint32_t> visit_slots;
                rc = upload(e, visit_slots, ft.slots.data(), ft.slots.size(), nullptr, true);
                if (rc) return rc;
                HIP_TRY(e, e->fast_nodes.resize(n - 1));
                HIP_TRY(e, e->fast_slots.resize(n));
                rb::DeviceTreeInfo info{};
                rc = rb::device_fast_bvh_build(e->tris.ptr, e->indices.ptr, visit_slots.ptr, n, e->fast_nodes.ptr,
                                               e->fast_slots.ptr, &info, e->stream,
                                               (e->opt.flags & RB_FLAG_DEVICE_LBVH) != 0u);
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
                                    rb::kStackDepth, ft))
                return RB_OK;  // keep the exact walk
            rc = upload(e, e->fast_nodes, ft.nodes.data(), ft.nodes.size(), nullptr, true);
            if (!rc) rc = upload(e, e->fast_slots, ft.slots.data(), ft.slots.size(), nullptr, true);
            if (rc) return rc;
            e->fast_builder = "host-sah";
        }
        rc = upload(e, e->slot_meta, ft.slot_meta.data(), ft.slot_meta.size(), nullptr, true);
        if (!rc) rc = upload(e, e->ref_parent, ft.ref_parent.data(), ft.ref_parent.size(), nullptr, true);
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

rb::KParams make_params(rb_engine* e, uint32_t first_pass, uint32_t n_passes) {
    rb::KParams p{};
    p.u = e->uniforms;
    // update_uniforms count patch-up -- gpu_wrapper.rs:475-495.  Keep leaves the caller's
    // value; it is clamped to the buffer so a stale count cannot read out of bounds.
    auto patch = [](uint32_t change, uint32_t given, uint32_t len) -> uint32_t {
        if (change == RB_CREATE || change == RB_UPDATE) return len;
        if (change == RB_DELETE) return 0u;
        return std::min(given, len);
    };
    p.u.spheres_count = patch(e->last_change_spheres, e->uniforms.spheres_count, e->n_spheres);
    p.u.bvh_node_count = patch(e->last_change_nodes, e->uniforms.bvh_node_count, e->n_nodes);
    p.u.bvh_triangle_count = patch(e->last_change_tris, e->uniforms.bvh_triangle_count, e->n_tris);
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
    p.accum = e->accum.ptr;
    p.out_rgba = e->out_rgba.ptr;
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
    const bool use_fast = e->fast_ready && p.u.bvh_node_count == e->n_nodes && p.u.bvh_node_count > 1u;
    p.fast_nodes = use_fast ? e->fast_nodes.ptr : nullptr;
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
    for (int i = 0; i < 3; ++i) {
        p.sph_bmin[i] = e->sph_bmin[i];
        p.sph_bmax[i] = e->sph_bmax[i];
    }
    // a single-node tree is walked without a stack (rb_kernels.hip, intersect_bvh)
    p.stack_depth = (p.u.bvh_node_count <= 1u) ? 0u : std::max(e->bvh_stack, 1u);
    if (use_sph_bvh) p.stack_depth = std::max(p.stack_depth, e->sph_depth);
    if (use_fast) p.stack_depth = std::max(p.stack_depth, std::min(e->fast_depth, rb::kStackDepth));
    p.stack_overflow = e->stack_overflow.ptr;
    p.blocks_per_cu = e->opt._reserved[0];
    p.queue_batch = e->opt._reserved[2];
    p.no_leaf_stepping = e->opt._reserved[3];
    p.lds_mode = e->opt._reserved[4];
    return p;
}

int require_ready(rb_engine* e) {
    if (!e->initialized) return fail(e, RB_ERR_NOT_INITIALIZED, "engine has not received its first update");
    if (!e->have_uniforms) return fail(e, RB_ERR_UNIFORMS_NOT_INITIALIZED, "Uniforms must be initialized");
    return RB_OK;
}

int clear_accum(rb_engine* e) {
    const size_t px = static_cast<size_t>(e->width) * e->padded_rows;
    if (px) HIP_TRY(e, hipMemsetAsync(e->accum.ptr, 0, px * 16, e->stream));
    return RB_OK;
}

// dispatch_compute_progressive without the host sync -- gpu_wrapper.rs:365-400.
// The bvh_node_count patched for Keep is only known after make_params.
int dispatch(rb_engine* e, uint32_t first_pass, uint32_t n_passes) {
    int rc = ensure_prepared(e);
    if (rc) return rc;
    if (n_passes == 0 || e->width == 0 || e->local_rows == 0) return RB_OK;
    const uint32_t kernel = e->opt.kernel ? e->opt.kernel : RB_KERNEL_STREAM;
    const bool stats = (e->opt.flags & RB_FLAG_STATS) != 0;
    uint32_t chunk = e->opt.passes_per_launch ? e->opt.passes_per_launch : n_passes;
    if (kernel == RB_KERNEL_STREAM) {
        // One float4 per (pixel, sample) of a launch chunk.  Keep the item count below 2^31
        // and, unless the caller fixed the chunk, the buffer within a budget (default 40 GiB of the
        // 288 GB of HBM: the whole C2 frame -- 1024 spp, 34 GB -- is then one launch; 32 / 8 / 2 / 1
        // launches per frame measured 23.5 / 24.6 / 24.7 / 24.7 G segments/s).
        const uint64_t tiles = static_cast<uint64_t>((e->width + 7) / 8) * ((e->local_rows + 7) / 8);
        const uint64_t per_pass = tiles * 64ull * e->prh.samples_per_pass;  // items per pass
        const uint64_t budget_items = (e->opt._reserved[1] ? static_cast<uint64_t>(e->opt._reserved[1]) : 40960ull) * (1ull << 20) / 16ull;
        uint64_t max_chunk = std::min<uint64_t>((1ull << 31) / std::max<uint64_t>(per_pass, 1) , 0xFFFFFFFFull);
        if (!e->opt.passes_per_launch) max_chunk = std::min(max_chunk, std::max<uint64_t>(budget_items / std::max<uint64_t>(per_pass, 1), 1));
        if (max_chunk == 0) return fail(e, RB_ERR_INVALID_UNIFORMS, "frame too large for one launch");
        chunk = static_cast<uint32_t>(std::min<uint64_t>(chunk, max_chunk));
        HIP_TRY(e, e->colors.resize(per_pass * chunk * 4));
    }
    HIP_TRY(e, hipEventRecord(e->ev_begin, e->stream));
    uint32_t launches = 0;
    e->ev_used = 0;
    for (uint32_t done = 0; done < n_passes;) {
        const uint32_t n = std::min(chunk, n_passes - done);
        rb::KParams p = make_params(e, first_pass + done, n);
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
    e->last_launches = launches;
    e->timing_pending = true;
    e->stats.launches += launches;
    return RB_OK;
}

int read_rgba(rb_engine* e, uint8_t* out) {
    if (!out) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    const uint32_t sc = e->opt.shard_count > 1 ? e->opt.shard_count : 1;
    const size_t row_bytes = static_cast<size_t>(e->width) * 4;
    // The destination is caller-owned, normally pageable memory: finish the stream's work, then a
    // blocking copy.  (An async copy into pageable memory followed by a stream wait leaves it to the
    // runtime when the bytes reach the caller's buffer.)
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out, e->out_rgba.ptr, row_bytes * (sc == 1 ? e->height : e->padded_rows), hipMemcpyDeviceToHost));
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

int update_fields(rb_engine* e, const rb_config* cfg) {
    int rc = check_fields(e, cfg);
    if (rc) return rc;
    const bool first = !e->initialized;
    rc = first ? validate_init(e, cfg) : validate(e, cfg);
    if (rc) return rc;

    // ---- uniforms (gpu_wrapper.rs:122-136 / :165-192)
    const bool take_uniforms = first ? (cfg->uniforms.change == RB_CREATE) : (cfg->uniforms.change == RB_UPDATE);
    if (take_uniforms) {
        const rb_uniforms* u = static_cast<const rb_uniforms*>(cfg->uniforms.ptr);
        if (u->width != e->width || u->height != e->height || e->accum.ptr == nullptr) {
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
    rc = validate_scene(e, cfg);
    if (rc) return rc;
    e->initialized = true;
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

int render_locked(rb_engine* e, uint8_t* rgba_out) {
    int rc = require_ready(e);
    if (rc) return rc;
    if (!rgba_out) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    rc = clear_accum(e);  // dispatch_compute, gpu_wrapper.rs:407-411
    if (rc) return rc;
    e->prh.current_pass = 0;
    rc = dispatch(e, 0, e->prh.total_passes);
    if (rc) return rc;
    e->prh.current_pass = e->prh.total_passes ? e->prh.total_passes - 1 : 0;  // loop variable's last value (:415)
    rc = read_rgba(e, rgba_out);
    if (rc) return rc;
    return accumulate_timing(e);
}

rb_engine* create_impl(const rb_config* cfg, const rb_options* opt_in) {
    g_create_error.clear();
    if (!cfg) { fail(nullptr, RB_ERR_NULL_ARGUMENT, "config is NULL"); return nullptr; }
    if (check_fields(nullptr, cfg)) return nullptr;
    // GpuBuffers::new panics unless these are Create (buffers.rs:74-97)
    if (validate_init(nullptr, cfg)) return nullptr;
    rb_options opt{};
    opt.device = -1;
    if (opt_in) opt = *opt_in;
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
    auto bail = [&](const char* what, hipError_t st) -> rb_engine* {
        fail(nullptr, RB_ERR_DEVICE, "%s failed: %s", what, hipGetErrorString(st));
        delete e;
        return nullptr;
    };
    hipError_t st;
    if ((st = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", st);
    if ((st = hipEventCreate(&e->ev_begin)) != hipSuccess) return bail("hipEventCreate", st);
    if ((st = hipEventCreate(&e->ev_end)) != hipSuccess) return bail("hipEventCreate", st);
    if ((st = e->counters.resize(rb::C_COUNT)) != hipSuccess) return bail("hipMalloc(counters)", st);
    if ((st = e->queue.resize(4)) != hipSuccess) return bail("hipMalloc(queue)", st);
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

}  // namespace

// ============================================================== C ABI ======
extern "C" {

rb_engine* rb_create(const rb_config* cfg) { return create_impl(cfg, nullptr); }
rb_engine* rb_create_ex(const rb_config* cfg, const rb_options* opt) { return create_impl(cfg, opt); }

void rb_destroy(rb_engine* e) {
    if (!e) return;
    set_device(e);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    for (hipEvent_t x : e->ev_pool) (void)hipEventDestroy(x);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* rb_last_error(const rb_engine* e) { return e ? e->error.c_str() : g_create_error.c_str(); }

int rb_update(rb_engine* e, const rb_config* cfg) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    return update_locked(e, cfg);
}

int rb_render(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    return render_locked(e, rgba_out);
}

int rb_render_config(rb_engine* e, const rb_config* cfg, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = update_locked(e, cfg);
    if (rc) return rc;
    return render_locked(e, rgba_out);
}

int rb_iter_begin(rb_engine* e, const rb_config* cfg) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = update_locked(e, cfg);
    if (rc) return rc;
    rc = require_ready(e);
    if (rc) return rc;
    e->prh.current_pass = 0;       // lib.rs:91
    e->iter_initialized = false;   // RaytracerFrameIterator::new (lib.rs:144-150)
    return RB_OK;
}

int rb_iter_has_next(rb_engine* e) {
    if (!e) return 0;
    std::lock_guard<std::mutex> lock(e->mu);
    return e->prh.current_pass < e->prh.total_passes ? 1 : 0;  // lib.rs:153-156
}

int rb_iter_next(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    if (!(e->prh.current_pass < e->prh.total_passes))
        return fail(e, RB_ERR_NO_MORE_FRAMES, "No more frames available");  // lib.rs:170-177
    int rc = require_ready(e);
    if (rc) return rc;
    if (!rgba_out) return fail(e, RB_ERR_NULL_ARGUMENT, "rgba_out is NULL");
    if (!e->iter_initialized) {  // lib.rs:181-192
        rc = clear_accum(e);
        if (rc) return rc;
        e->iter_initialized = true;
    }
    const uint32_t left = e->prh.total_passes - e->prh.current_pass;
    const uint32_t n = std::min(std::max(e->iter_passes_per_frame, 1u), left);
    rc = dispatch(e, e->prh.current_pass, n);  // lib.rs:200-203 (n = 1 there)
    if (rc) return rc;
    rc = read_rgba(e, rgba_out);  // lib.rs:205
    if (rc) return rc;
    e->prh.current_pass += n;  // lib.rs:213
    return accumulate_timing(e);
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
    if (!e->have_uniforms) {
        e->error = "Uniforms must be initialized";  // gpu_wrapper.rs:313,321
        return RB_ERR_UNIFORMS_NOT_INITIALIZED;
    }
    if (width) *width = e->width;
    if (height) *height = e->height;
    return RB_OK;
}

int rb_clear(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    return clear_accum(e);
}

int rb_dispatch(rb_engine* e, uint32_t first_pass, uint32_t n_passes) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    return dispatch(e, first_pass, n_passes);
}

int rb_sync(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    return RB_OK;
}

int rb_read_rgba(rb_engine* e, uint8_t* rgba_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    return read_rgba(e, rgba_out);
}

int rb_read_accumulation(rb_engine* e, float* accum_out) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    int rc = require_ready(e);
    if (rc) return rc;
    if (!accum_out) return fail(e, RB_ERR_NULL_ARGUMENT, "accum_out is NULL");
    const uint32_t rows = (e->opt.shard_count > 1) ? e->padded_rows : e->height;
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(accum_out, e->accum.ptr, static_cast<size_t>(e->width) * rows * 16, hipMemcpyDeviceToHost));
    return RB_OK;
}

int rb_device_rgba(rb_engine* e, void** d_ptr, size_t* bytes) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    if (d_ptr) *d_ptr = e->out_rgba.ptr;
    if (bytes) *bytes = static_cast<size_t>(e->width) * e->padded_rows * 4;
    return RB_OK;
}

int rb_local_rows(const rb_engine* e, uint32_t* rows, uint32_t* padded_rows) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
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

int rb_get_stats(rb_engine* e, rb_stats* out) {
    if (!e || !out) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    unsigned long long c[rb::C_COUNT];
    HIP_TRY(e, hipStreamSynchronize(e->stream));
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

int rb_reset_stats(rb_engine* e) {
    if (!e) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
    set_device(e);
    HIP_TRY(e, hipMemsetAsync(e->counters.ptr, 0, sizeof(unsigned long long) * rb::C_COUNT, e->stream));
    e->stats = rb_stats{};
    return RB_OK;
}

int rb_last_dispatch_ms(rb_engine* e, float* ms) {
    if (!e || !ms) return RB_ERR_NULL_ARGUMENT;
    std::lock_guard<std::mutex> lock(e->mu);
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

const char* rb_version(void) { return "renderbaby-hip 0.1 (gfx950)"; }

const char* rb_last_kernel_name(const rb_engine* e) { return e ? e->last_kernel_name : ""; }

const char* rb_fast_bvh_builder(const rb_engine* e, float* build_ms) {
    if (build_ms) *build_ms = e ? e->fast_build_ms : 0.0f;
    return (e && e->fast_ready) ? e->fast_builder : "";
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

// Debug hook for tests/test_gpu_math.py: device /, sqrt, normalize, u32->f32, min/max, dot.
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
