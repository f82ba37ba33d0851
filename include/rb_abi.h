/*
 * rb_abi.h -- C ABI of librenderbaby_hip.so, the MI355X (gfx950) backend for
 * RenderBaby's path-tracing hot path.
 *
 * This is the drop-in boundary: the entry points below are what a Rust shim
 * crate (`engine-hip`) binds with `extern "C"` to implement
 * `engine_config::Renderer` and `frame_buffer::FrameIterator`.  All citations
 * are file:line under the reference checkout (crates/... , src/...).
 *
 * Plain C: pointers, sizes, PODs.  No torch, no C++ types.  Inputs are borrowed
 * for the duration of a call only (copied to the device before return); the
 * library owns every device allocation; outputs go to caller-owned buffers.
 */
#ifndef RB_ABI_H
#define RB_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* POD layouts -- byte-identical to the reference's #[repr(C)] GPU ABI types */
/* ------------------------------------------------------------------------ */

/* crates/engine-config/src/camera.rs:27-61 (48 B) */
typedef struct rb_camera {
    float pane_distance;
    float pane_width;
    float _pad0[2];
    float pos[3];
    float _pad1;
    float dir[3];
    float _pad2;
} rb_camera;

/* crates/engine-config/src/uniforms.rs:26-67 (144 B) */
typedef struct rb_uniforms {
    uint32_t width;
    uint32_t height;
    uint32_t total_samples;
    uint32_t color_hash_enabled;
    rb_camera camera;
    uint32_t spheres_count;
    uint32_t triangles_count;
    uint32_t bvh_node_count;
    uint32_t bvh_triangle_count;
    uint32_t bvh_root;
    float ground_height;
    uint32_t ground_enabled;
    uint32_t checkerboard_enabled;
    float sky_color[3];
    uint32_t max_depth;
    float checkerboard_color_1[3];
    uint32_t _pad1;
    float checkerboard_color_2[3];
    uint32_t _pad2;
} rb_uniforms;

/* crates/engine-config/src/material.rs:34-65 (80 B) */
typedef struct rb_material {
    float ambient[3];
    float _pad0;
    float diffuse[3];
    float _pad1;
    float specular[3];
    float shininess;
    float emissive[3];
    float ior;
    float opacity;
    uint32_t illum;
    int32_t texture_index;
    uint32_t _pad2;
} rb_material;

/* crates/engine-config/src/sphere.rs:34-43 (96 B) */
typedef struct rb_sphere {
    float center[3];
    float radius;
    rb_material material;
} rb_sphere;

/* crates/engine-config/src/point_lights.rs:23-33 (96 B) */
typedef struct rb_point_light {
    float center[3];
    float radius;
    rb_material material;
} rb_point_light;

/* crates/engine-config/src/mesh.rs:21-32 (96 B) */
typedef struct rb_mesh {
    uint32_t triangle_index_start;
    uint32_t triangle_count;
    uint32_t _pad[2];
    rb_material material;
} rb_mesh;

/* crates/engine-bvh/src/bvh.rs:18-34 (48 B) */
typedef struct rb_bvh_node {
    float aabb_min[3];
    uint32_t _pad0;
    float aabb_max[3];
    uint32_t _pad1;
    uint32_t left;
    uint32_t right;
    uint32_t first_primitive;
    uint32_t primitive_count;
} rb_bvh_node;

/* crates/engine-bvh/src/triangle.rs:9-29 (64 B) */
typedef struct rb_gpu_triangle {
    float v0[3];
    uint32_t v0_index;
    float v1[3];
    uint32_t v1_index;
    float v2[3];
    uint32_t v2_index;
    uint32_t mesh_index;
    uint32_t _pad0;
    uint32_t _pad1;
    uint32_t _pad2;
} rb_gpu_triangle;

/* crates/engine-config/src/texture.rs:26-36 -- Vec<u32> becomes ptr + w*h */
typedef struct rb_texture {
    uint32_t width;
    uint32_t height;
    const uint32_t* rgba_data; /* width*height texels, R in the low byte */
} rb_texture;

/* crates/engine-wgpu-wrapper/src/buffers.rs:13-25 (16 B); offset is in texels
 * (buffers.rs:151-168) */
typedef struct rb_texture_info {
    uint32_t offset;
    uint32_t width;
    uint32_t height;
    uint32_t _pad;
} rb_texture_info;

/* crates/engine-wgpu-wrapper/src/gpu_wrapper.rs:19-31 (16 B) */
typedef struct rb_progressive {
    uint32_t total_passes;
    uint32_t current_pass;
    uint32_t total_samples;
    uint32_t samples_per_pass;
} rb_progressive;

/* ------------------------------------------------------------------------ */
/* RenderConfig = 9 x Change<T>  (crates/engine-config/src/render_config.rs:37-57,99-109) */
/* ------------------------------------------------------------------------ */

enum {
    RB_KEEP = 0,   /* Change::Keep   */
    RB_CREATE = 1, /* Change::Create */
    RB_UPDATE = 2, /* Change::Update */
    RB_DELETE = 3  /* Change::Delete */
};

typedef struct rb_field {
    uint32_t change;  /* RB_KEEP.. */
    const void* ptr;  /* element array (may be NULL when count == 0) */
    size_t count;     /* number of ELEMENTS (uvs: number of f32) */
} rb_field;

typedef struct rb_config {
    rb_field uniforms;      /* 1 x rb_uniforms */
    rb_field spheres;       /* rb_sphere[] */
    rb_field uvs;           /* float[] (pairs) */
    rb_field meshes;        /* rb_mesh[] */
    rb_field lights;        /* rb_point_light[] */
    rb_field bvh_nodes;     /* rb_bvh_node[] */
    rb_field bvh_indices;   /* uint32_t[] */
    rb_field bvh_triangles; /* rb_gpu_triangle[] */
    rb_field textures;      /* rb_texture[] */
} rb_config;

/* ------------------------------------------------------------------------ */
/* Status codes.  1..10 mirror RenderConfigBuilderError
 * (crates/engine-config/src/render_config.rs:608-619); the rest replace the
 * reference's panics / anyhow errors.                                        */
/* ------------------------------------------------------------------------ */
enum {
    RB_OK = 0,
    RB_ERR_PANE_DISTANCE_OUT_OF_BOUNDS = 1,
    RB_ERR_PANE_WIDTH_OUT_OF_BOUNDS = 2,
    RB_ERR_INVALID_CAMERA_DIRECTION = 3,
    RB_ERR_INVALID_UNIFORMS = 4,
    RB_ERR_INVALID_SPHERES = 5,
    RB_ERR_INVALID_UVS = 6,
    RB_ERR_INVALID_MESHES = 7,
    RB_ERR_INVALID_LIGHTS = 8,
    RB_ERR_INVALID_TEXTURES = 9,
    RB_ERR_CANNOT_DELETE_NONEXISTENT = 10,
    RB_ERR_UNIFORMS_NOT_INITIALIZED = 11, /* gpu_wrapper.rs:313,321 panic */
    RB_ERR_NO_MORE_FRAMES = 12,           /* engine-pathtracer/src/lib.rs:170-177 */
    RB_ERR_INVALID_BVH = 13,              /* malformed tree (cycle, range, depth) */
    RB_ERR_UNSUPPORTED_DELETE = 14,       /* render_config.rs `todo!()` arms */
    RB_ERR_NULL_ARGUMENT = 15,
    RB_ERR_DEVICE = 16,                   /* HIP runtime failure */
    RB_ERR_NOT_INITIALIZED = 17,          /* render before the first update */
    RB_ERR_INVALID_OPTIONS = 18
};

typedef struct rb_engine rb_engine;

/* Options that have no counterpart in the reference (one wgpu device, whole
 * frame): device choice, row-stripe sharding for multi-GPU, launch shape. */
typedef struct rb_options {
    int32_t device;             /* HIP device ordinal; -1 = current */
    uint32_t shard_rank;        /* this engine renders stripes s with s % shard_count == shard_rank */
    uint32_t shard_count;       /* 0 or 1 = whole frame */
    uint32_t stripe_rows;       /* rows per stripe; 0 = default (8: the height of a kernel tile) */
    uint32_t passes_per_launch; /* samples per pixel folded into one kernel launch; 0 = default */
    uint32_t kernel;            /* RB_KERNEL_* ; 0 = default */
    uint32_t flags;             /* RB_FLAG_* */
    uint32_t _reserved[5];      /* tuning / ablation knobs, 0 = default: [0] persistent blocks per CU, [1] colour-buffer
                                   budget in MiB (default 4096, at most half of the free device memory), [2] items a wave
                                   reserves per queue atomic (a multiple of 64; multiples of 256 let the default kernel combine
                                   its colour stores), [3] 1 = no leaf stepping, [4] LDS staging of small meshes (1 = never) */
} rb_options;

enum {
    RB_KERNEL_DEFAULT = 0,
    RB_KERNEL_PIXEL = 1,  /* one thread per pixel, nested sample/depth loops */
    RB_KERNEL_QUEUE = 2,  /* persistent wavefronts, pixel queue + path regeneration */
    RB_KERNEL_STREAM = 3  /* persistent wavefronts over (pixel, sample) items + ordered accumulate pass */
};

enum {
    RB_FLAG_STATS = 1u, /* count nodes/tris/spheres/lights per segment (slower) */
    RB_FLAG_NO_SPHERE_BVH = 2u, /* always use the reference's linear sphere scan (shader.wgsl:574-586) */
    RB_FLAG_SPHERE_TREE_HOST = 2048u,   /* more than 64 spheres: build the library's sphere tree on the host (median splits) ... */
    RB_FLAG_SPHERE_TREE_DEVICE = 4096u, /* ... or on the device (the same splits, a segmented sort per level) whatever the count; by default the device
                                           builds it from 1024 spheres up.  The frame does not depend on the builder. */
    RB_FLAG_FAST_BVH = 4u, /* multi-node meshes: walk the library's own tree over the triangles (culling, near-first) plus
                              a second pass over the caller's tree for hits reported from near-zero determinants, and accept
                              a hit only if the reference's traversal would have tested it: argued and fuzzed to deliver the
                              reference walk's frames.  Without a walk flag the library uses the
                              chunked walk (RB_FLAG_CHUNK_WALK) */
    RB_FLAG_DEVICE_BVH = 8u, /* with RB_FLAG_FAST_BVH: build that tree on the GPU (Morton order + locally-ordered
                                clustering) instead of on the host (binned SAH): milliseconds instead of ~0.5 s per
                                million triangles, the same frames */
    RB_FLAG_DEVICE_LBVH = 16u, /* with RB_FLAG_DEVICE_BVH: plain LBVH instead of the clustering (ablation: faster
                                 build, slower walk) */
    RB_FLAG_REFERENCE_WALK = 32u, /* multi-node meshes: walk the caller's tree exactly as shader.wgsl:282-392 does
                                     (128-triangle leaves, fixed order, no culling); same frames, 2-5x slower */
    RB_FLAG_HOST_BVH = 64u, /* build the library's tree on the host (binned SAH) whatever the triangle count */
    RB_FLAG_GATHER_PEER_COPY = 128u, /* rb_create_multi: move the stripes with hipMemcpyPeerAsync instead of RCCL
                                        (hosts without librccl; several shards on one device in the tests) */
    RB_FLAG_NO_RUN_AHEAD = 256u, /* progressive iterator: do not start the next pass while a frame is read back */
    RB_FLAG_CHUNK_WALK = 1024u, /* multi-node meshes: the chunked walk, THE DEFAULT -- the caller's tree walked with the
                                   reference's own slab arithmetic (nearer child first, subtrees culled on the best t by the
                                   margin that bounds the reference's reported hits), the library's own levels below its
                                   leaves down to 16-triangle chunks, which a wavefront tests cooperatively, one triangle
                                   per lane.  Same frames: every hit goes through the reference's own
                                   triangle test, and the two rounding-error inequalities the culling rests on are derived
                                   mechanically (tools/margin_certify.py) and fuzzed against the oracle, not proven in a proof
                                   assistant -- RB_FLAG_REFERENCE_WALK, or RB_REFERENCE_WALK=1 in the environment of a host that
                                   cannot be rebuilt, walks the reference's way.  The flag only makes the choice explicit */
    RB_FLAG_CHUNK_TREE_HOST = 8192u,    /* the chunked walk's tree: build it on the host (one thread per granted CPU) ... */
    RB_FLAG_CHUNK_TREE_DEVICE = 16384u, /* ... or on the device (one thread block per reference leaf) whatever the mesh's size; by default the
                                           device builds it from 16 384 triangle slots up.  A caller's tree with leaves of more than 256
                                           triangles is built on the host either way.  The frame does not depend on the builder. */
    RB_FLAG_SKIP_NEAR_DEGENERATE = 512u /* with the library's tree: skip its second pass.  The walk then answers only for
                                           hits whose ray is more than ~1.7 degrees off the plane of a LARGE triangle
                                           (L^2 > 1.6e-2); a hit the reference reports from a near-zero determinant there can be
                                           missed.  Several times faster on coarse meshes; frames validated equal on the
                                           BASELINE scenes, but this is the one mode the exactness argument does not cover. */
};

/* Work counters, summed over every launch since the last rb_reset_stats.
 * `segments` is the throughput unit (one executed iteration of the bounce
 * loop, shader.wgsl:534); the rest feed the algorithmic-bytes formula of
 * SURVEY.md section 8(d) and are only filled when RB_FLAG_STATS is set. */
typedef struct rb_stats {
    uint64_t segments;
    uint64_t paths;
    uint64_t nodes_popped;
    uint64_t tris_tested;
    uint64_t spheres_tested;
    uint64_t lights_tested;
    uint64_t mesh_hits;
    uint64_t launches;
    double kernel_ms; /* sum of HIP-event durations of the render launch groups */
    double trace_ms;  /* of which: the trace / render kernels alone (k_trace*, k_queue, k_pixel) */
    double accumulate_ms; /* of which: k_accumulate (RB_KERNEL_STREAM only) */
} rb_stats;

/* ------------------------------------------------------------------------ */
/* Entry points                                                             */
/* ------------------------------------------------------------------------ */

/* Engine::new(rc) -- crates/engine-pathtracer/src/lib.rs:111-119 ->
 * GpuWrapper::new, gpu_wrapper.rs:82-105.  Requires Create for uniforms,
 * spheres, uvs, meshes, lights, textures (buffers.rs:74-97 panics otherwise;
 * here: NULL + rb_last_error(NULL)).  The engine is not yet "initialized":
 * the first rb_update must again carry Create (gpu_wrapper.rs:117-121). */
rb_engine* rb_create(const rb_config* cfg);
rb_engine* rb_create_ex(const rb_config* cfg, const rb_options* opt);

/* The same engine over several devices of this process (the reference has one wgpu
 * device, gpu_device.rs:27-70): the frame's rows are sharded in interleaved stripes (stripe s -> devices[s % n]),
 * every device renders all samples of its rows with global pixel indices, and each delivered frame is ONE RCCL
 * gather of the RGBA8 stripes to devices[0] (grouped ncclSend / ncclRecv over a communicator made with
 * ncclCommInitAll), de-interleaved there and read back -- so rb_render / rb_iter_next return the whole frame,
 * bit-identical to a single-device engine's.  opt->shard_* must be zero; every other entry point works on the
 * handle as on a single-device engine (statistics add up over the devices). */
rb_engine* rb_create_multi(const rb_config* cfg, const rb_options* opt, const int32_t* devices, uint32_t n_devices);

/* One process per device instead: every process creates its shard (rb_create_ex with shard_rank / shard_count),
 * rank 0 makes an id, the host program hands it to the others (any transport: MPI, a file, torch.distributed)
 * and all call rb_comm_init_rank.  From then on rb_render / rb_iter_next gather the stripes to rank 0, which
 * receives the whole frame; on the other ranks rgba_out may be NULL and nothing is written. */
#define RB_COMM_ID_BYTES 128
/* Can this process load RCCL at all (RB_OK) -- so that every rank can say so BEFORE the collective rb_comm_init_rank, and
 * one that cannot does not leave the others waiting inside it.  Loads the library and resolves its symbols, nothing else:
 * no id is made (ncclGetUniqueId opens a listening socket and starts a thread that waits for the ranks to check in). */
int rb_comm_available(void);
/* On rank 0 only. */
int rb_comm_unique_id(uint8_t id_out[RB_COMM_ID_BYTES]);
int rb_comm_init_rank(rb_engine* e, const uint8_t id[RB_COMM_ID_BYTES], uint32_t rank, uint32_t nranks);

/* What the exchange of a sharded engine looks like from inside: the communicator's size and this handle's rank AS RCCL
 * REPORTS THEM (ncclCommCount / ncclCommUserRank; 0 ranks when nothing goes through RCCL: a whole-frame engine, or the
 * peer-copy transport), and the duration of this rank's share of the last gather (root: receives + de-interleave, on
 * its exchange stream, HIP events).  Any pointer may be NULL.  No reference counterpart (one wgpu device). */
int rb_comm_info(rb_engine* e, uint32_t* rccl_ranks, uint32_t* rccl_rank, float* last_gather_ms);

/* drop(Engine) */
void rb_destroy(rb_engine* e);

/* GpuWrapper::update + update_uniforms -- gpu_wrapper.rs:116-300,469-576:
 * Change state machine, validate_init / validate
 * (render_config.rs:163-268), count patch-up, uploads. */
int rb_update(rb_engine* e, const rb_config* cfg);

/* GpuWrapper::dispatch_compute + read_pixels -- gpu_wrapper.rs:406-426,432-463:
 * zero accumulation, run all total_samples passes, return RGBA8 (w*h*4 bytes,
 * row-major, top row first, x mirrored, A=255) into rgba_out. */
int rb_render(rb_engine* e, uint8_t* rgba_out);

/* <Engine as Renderer>::render(rc) -- engine-pathtracer/src/lib.rs:58-71:
 * rb_update + rb_render in one call. */
int rb_render_config(rb_engine* e, const rb_config* cfg, uint8_t* rgba_out);

/* <Engine as Renderer>::frame_iterator(rc) -- lib.rs:86-96: rb_update, then
 * current_pass = 0.  One iterator per engine at a time (the reference shares
 * one GpuWrapper behind a mutex the same way). */
int rb_iter_begin(rb_engine* e, const rb_config* cfg);
/* RaytracerFrameIterator::has_next -- lib.rs:153-156 */
int rb_iter_has_next(rb_engine* e);
/* RaytracerFrameIterator::next -- lib.rs:169-228: first call zeroes the
 * accumulation; one pass; read back; current_pass += 1.
 * Exhausted => RB_ERR_NO_MORE_FRAMES ("No more frames available"). */
int rb_iter_next(rb_engine* e, uint8_t* rgba_out);
/* RaytracerFrameIterator::destroy -- lib.rs:231-233 */
void rb_iter_destroy(rb_engine* e);
/* Extension (per-N-pass delivery): every rb_iter_next advances `n` passes
 * (the last one whatever is left) in one launch chunk and reads back once, so a frame is delivered
 * every n samples instead of after each one -- the frames are the reference's frames n-1, 2n-1, ...
 * and the last.  n = 0 or 1 is the reference's behaviour (the default; restored by rb_create only).
 * Applies from the next rb_iter_next. */
int rb_iter_set_passes_per_frame(rb_engine* e, uint32_t n);

/* anyhow error text of the last failing call on `e` (or of rb_create when
 * e == NULL): a copy owned by the calling thread, valid until that thread's next rb_last_error. */
const char* rb_last_error(const rb_engine* e);

/* GpuWrapper::get_width/get_height -- gpu_wrapper.rs:317-329 */
int rb_get_size(const rb_engine* e, uint32_t* width, uint32_t* height);

/* ---- Lower-level control used by bench.py, the parity tests and the
 * multi-GPU gather.  No reference counterpart: the reference can only run
 * whole renders synchronously. ---- */

/* Zero the accumulation buffer (gpu_wrapper.rs:407-411). */
int rb_clear(rb_engine* e);
/* Launch passes [first_pass, first_pass+n_passes) asynchronously on the
 * engine's stream (dispatch_compute_progressive, gpu_wrapper.rs:365-400,
 * without the per-pass host sync). */
int rb_dispatch(rb_engine* e, uint32_t first_pass, uint32_t n_passes);
/* Optional: everything a dispatch of n_passes passes would set up lazily -- the prepared triangles and the library's own
 * levels of the tree (as rb_dispatch(e, 0, 0) does), and the stream kernels' colour buffer (up to 4 GiB of device memory) --
 * without tracing anything.  A host that times its first frame calls this first and does not time hipMalloc. */
int rb_reserve(rb_engine* e, uint32_t n_passes);
/* Wait for the engine's stream. */
int rb_sync(rb_engine* e);
/* Copy this engine's RGBA8 rows (mirrored, local stripe order) to the host. */
int rb_read_rgba(rb_engine* e, uint8_t* rgba_out);
/* Copy the f32 accumulation (vec4 per pixel, shader x order, local stripe
 * order) to the host: local_rows*width*4 floats. */
int rb_read_accumulation(rb_engine* e, float* accum_out);
/* Page-locked host memory for frames (extension): rb_render / rb_iter_next / rb_read_rgba recognise a
 * page-locked rgba_out (from here, or registered by the caller with hipHostRegister) and copy into it by DMA on
 * a second stream, without the staging and host-side copy a pageable destination costs -- with the iterator's
 * run-ahead pass this is what lets frame delivery run at the compute rate.  Plain malloc'ed buffers keep
 * working (blocking copy).  Free with rb_host_free. */
void* rb_host_alloc(size_t bytes);
void rb_host_free(void* p);

/* Device pointer + byte size of the RGBA8 buffer of the committed frame (a single engine: its local stripe
 * buffer, valid until the next iterator step; a multi-device handle: the assembled frame on devices[0]). */
int rb_device_rgba(rb_engine* e, void** d_ptr, size_t* bytes);
/* Number of image rows this engine owns (== height when not sharded) and
 * the padded row count of the local buffer (equal on every rank). */
int rb_local_rows(const rb_engine* e, uint32_t* rows, uint32_t* padded_rows);
/* Global row index of local row `local_row`. */
int rb_global_row(const rb_engine* e, uint32_t local_row, uint32_t* global_row);

/* Row-stripe sharding geometry as pure functions (no engine, no device): stripe s
 * (rows [s*stripe_rows, (s+1)*stripe_rows)) belongs to rank s % shard_count; a
 * rank stores its stripes back to back.  owned_rows = image rows the rank
 * renders; padded_rows = rows of its local buffer, equal on every rank so that
 * one equal-size gather moves the frame. */
int rb_shard_layout(uint32_t height, uint32_t shard_rank, uint32_t shard_count, uint32_t stripe_rows,
                    uint32_t* owned_rows, uint32_t* padded_rows);
/* Global image row of a rank's local row (may be >= height in the padding). */
uint32_t rb_shard_global_row(uint32_t shard_rank, uint32_t shard_count, uint32_t stripe_rows,
                             uint32_t local_row);

/* Counters and times of every launch since the last rb_reset_stats.  While the progressive iterator is running they
 * INCLUDE the pass group it has started ahead of the delivered frame (and one it later drops because the scene
 * changed): the device counts what it traces. */
int rb_get_stats(rb_engine* e, rb_stats* out);
int rb_reset_stats(rb_engine* e);
/* Duration of the most recent rb_dispatch launch group, HIP events on the
 * engine's stream (ms).  Synchronises. */
int rb_last_dispatch_ms(rb_engine* e, float* ms);

/* BVH::new -- crates/engine-bvh/src/bvh.rs:87-150: median split on the longest
 * axis, leaves of <= 128 triangles, pre-order numbering.  Two-call protocol:
 * pass nodes_out == NULL to query sizes.  indices_out must hold n_tris u32.
 * The top levels are built on several threads (a subtree's node count follows from its triangle count, so every
 * index is known beforehand): 10^6 triangles in 12 ms on 16 cores where one takes 133 -- the tree the reference
 * rebuilds on the CPU for every render (scene_engine_adapter.rs:435-440). */
int rb_bvh_build(const rb_gpu_triangle* tris, size_t n_tris,
                 rb_bvh_node* nodes_out, size_t nodes_capacity, size_t* n_nodes,
                 uint32_t* indices_out);

/* Test aid (host only, no device): builds the chunked walk's tree for this mesh and this caller tree and checks the
 * structural invariants the kernel relies on (every valid slot in exactly one chunk, ranks consistent, references in range,
 * depth within the stack, per child slot an unbounded margin or a box and a bound that cover the triangles below).
 * out6 = {built (0: this tree is left to another walk), nodes, positions, depth, chunks, child slots with an unbounded
 * margin}; a violated invariant is RB_ERR_INVALID_BVH with rb_last_error(NULL) naming it. */
int rb_debug_chunk_tree(const rb_gpu_triangle* tris, size_t n_tris, const rb_bvh_node* nodes, size_t n_nodes, const uint32_t* indices,
                        size_t n_indices, uint64_t out6[6]);
/* The same check on the tree an engine is walking (after the first rb_dispatch / rb_render that follows an update): the
 * chunked walk's arrays are read back from the device -- whichever builder made them -- and checked against the engine's
 * copy of the mesh.  out6 as above (built = 0: the engine has no chunked tree). */
int rb_debug_engine_chunk_tree(rb_engine* e, uint64_t out6[6]);
/* Which builder produced the chunked walk's tree: "device", "host", or "" when the engine walks another way.  Valid after the
 * first rb_dispatch / rb_render that follows an update; `build_ms`, if not NULL, receives the wall time of that build with
 * its uploads and the gather of the chunks' triangle records. */
const char* rb_chunk_tree_builder(const rb_engine* e, float* build_ms);
/* Measurement aid for the roofline record (bench.py): the rate at which this device serves divergent 16-byte gathers --
 * every lane its own 128-byte line of a table of `table_bytes` (0 = 2 MiB, L2-resident) -- in lane accesses per second:
 * the ceiling of the L1 / texture-address path that a lane-per-ray tree walk runs into. */
int rb_measure_l1_gather(int32_t device, uint64_t table_bytes, double* accesses_per_s);
/* Test hook: evaluates the device's f32 /, sqrt, normalize, u32->f32, min/max and
 * dot on n input pairs (out8n: 8*n floats) so tests can check them against
 * IEEE-754 results computed on the host. */
int rb_debug_math(const float* a, const float* b, float* out8n, uint32_t n);
/* Measurement aid: the pass / phase occupancy counters of a library built from sources with tools/ablate/rb_profile.patch
 * applied (tools/walk_profile.sh: k_trace_sph's passes and the lanes, pairs, rounds and survivors in them; k_trace's lanes per
 * phase of its loop body), summed over every launch since the last reset.  The product build carries no counting code and
 * returns RB_ERR_DEVICE. */
int rb_debug_walk_profile(uint64_t out64[64], int reset);
/* Test hook: checks the kernels' fast exact reciprocal against the compiler's correctly rounded
 * 1/b for all 2^23 significands (both signs) at one biased exponent; out16[0] = mismatch count. */
int rb_debug_rcp_exhaustive(uint32_t biased_exponent, uint32_t* out16);
/* Test hook: the same for the fast exact division a/b over a block of significand pairs
 * (denominators [b_begin, +b_count) x numerators [a_begin, +a_count), biased exponents ea, eb);
 * out16[0] = mismatch count, then up to 7 (a, b) bit patterns. */
int rb_debug_div_exhaustive(uint32_t b_begin, uint32_t b_count, uint32_t ea, uint32_t eb, uint32_t a_begin,
                            uint32_t a_count, unsigned long long* out16);

/* Name of the render kernel the most recent rb_dispatch used ("k_trace", "k_trace_bvh",
 * "k_queue", "k_pixel"). */
const char* rb_last_kernel_name(const rb_engine* e);

/* Which builder produced the library's own tree: "host-sah", "device-ploc", "device-lbvh", or "" when
 * there is none (flag not set, single-node tree, or the scene keeps the exact walk).  Valid after the
 * first rb_dispatch / rb_render that follows an update.  `build_ms`, if not NULL, receives the wall
 * time of that build including its uploads. */
const char* rb_fast_bvh_builder(const rb_engine* e, float* build_ms);

/* Which builder produced the library's sphere tree (scenes with more than 64 spheres): "device-median", "host-median", or
 * "" when the engine scans (<= 64 spheres, RB_FLAG_NO_SPHERE_BVH).  Valid after the rb_update that brought the spheres.
 * `build_ms`, if not NULL, receives the wall time of that build including its uploads. */
const char* rb_sphere_tree_builder(const rb_engine* e, float* build_ms);

/* Library / device identification for logs. */
const char* rb_version(void);
int rb_device_name(int device, char* buf, size_t buf_len);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(rb_camera) == 48, "Camera is 48 B");
static_assert(sizeof(rb_uniforms) == 144, "Uniforms is 144 B");
static_assert(sizeof(rb_material) == 80, "Material is 80 B");
static_assert(sizeof(rb_sphere) == 96, "Sphere is 96 B");
static_assert(sizeof(rb_point_light) == 96, "PointLight is 96 B");
static_assert(sizeof(rb_mesh) == 96, "Mesh is 96 B");
static_assert(sizeof(rb_bvh_node) == 48, "BVHNode is 48 B");
static_assert(sizeof(rb_gpu_triangle) == 64, "GPUTriangle is 64 B");
static_assert(sizeof(rb_texture_info) == 16, "TextureInfo is 16 B");
static_assert(sizeof(rb_progressive) == 16, "ProgressiveRenderHelper is 16 B");
static_assert(offsetof(rb_uniforms, camera) == 16, "camera @16");
static_assert(offsetof(rb_uniforms, spheres_count) == 64, "spheres_count @64");
static_assert(offsetof(rb_uniforms, ground_height) == 84, "ground_height @84");
static_assert(offsetof(rb_uniforms, sky_color) == 96, "sky_color @96");
static_assert(offsetof(rb_uniforms, max_depth) == 108, "max_depth @108");
static_assert(offsetof(rb_uniforms, checkerboard_color_1) == 112, "cb1 @112");
static_assert(offsetof(rb_uniforms, checkerboard_color_2) == 128, "cb2 @128");
static_assert(offsetof(rb_camera, pos) == 16, "pos @16");
static_assert(offsetof(rb_camera, dir) == 32, "dir @32");
static_assert(offsetof(rb_material, diffuse) == 16, "diffuse @16");
static_assert(offsetof(rb_material, specular) == 32, "specular @32");
static_assert(offsetof(rb_material, shininess) == 44, "shininess @44");
static_assert(offsetof(rb_material, emissive) == 48, "emissive @48");
static_assert(offsetof(rb_material, texture_index) == 72, "texture_index @72");
static_assert(offsetof(rb_bvh_node, left) == 32, "left @32");
static_assert(offsetof(rb_gpu_triangle, mesh_index) == 48, "mesh_index @48");
#else
_Static_assert(sizeof(rb_camera) == 48, "Camera is 48 B");
_Static_assert(sizeof(rb_uniforms) == 144, "Uniforms is 144 B");
_Static_assert(sizeof(rb_material) == 80, "Material is 80 B");
_Static_assert(sizeof(rb_sphere) == 96, "Sphere is 96 B");
_Static_assert(sizeof(rb_point_light) == 96, "PointLight is 96 B");
_Static_assert(sizeof(rb_mesh) == 96, "Mesh is 96 B");
_Static_assert(sizeof(rb_bvh_node) == 48, "BVHNode is 48 B");
_Static_assert(sizeof(rb_gpu_triangle) == 64, "GPUTriangle is 64 B");
_Static_assert(sizeof(rb_texture_info) == 16, "TextureInfo is 16 B");
_Static_assert(sizeof(rb_progressive) == 16, "ProgressiveRenderHelper is 16 B");
#endif

#endif /* RB_ABI_H */
