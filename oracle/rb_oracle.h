/*
 * rb_oracle.h -- CPU oracle for RenderBaby's path-tracing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * WGSL compute shader (crates/engine-pathtracer/src/shader.wgsl) and of the
 * host-side conventions around it (pass numbering, count patch-up, x-mirror,
 * empty-buffer rule).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (librenderbaby_hip.so) never does.
 *
 * PARITY PINNING: the reference holds no render test, golden image or
 * known-answer vector for this path, and it cannot
 * be built here (Rust + wgpu, no toolchain).  The oracle is therefore pinned
 * only by known answers derived by hand from the shader's source text (PCG
 * hash values, hash_to_color(1), struct sizes, analytic sphere / triangle /
 * sky / emissive cases -- tests/test_oracle_kat.py).  Against the reference's
 * own tests: "parity unpinned".
 *
 * Numeric conventions where WGSL leaves latitude (documented in DESIGN.md):
 *   - every f32 op is a single IEEE-754 binary32 op, round-to-nearest-even,
 *     no fused multiply-add (-ffp-contract=off), subnormals kept;
 *   - dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z;  cross by the textbook formula;
 *   - normalize(v) = v / sqrt(dot(v,v))  (three correctly-rounded divisions);
 *   - sqrt and / are correctly rounded;  min/max are IEEE minNum/maxNum;
 *   - u32(f32) / i32(f32) truncate and saturate, NaN -> 0;
 *   - pow(c, 2.2) on a texel channel is libm powf of the host (256 possible
 *     inputs per channel).
 */
#ifndef RB_ORACLE_H
#define RB_ORACLE_H

#include "../include/rb_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A fully-resolved scene, i.e. the contents of the 13 shader bindings after
 * GpuWrapper::update_uniforms (gpu_wrapper.rs:469-576).  Counts are taken
 * from the array lengths exactly as the reference patches them. */
typedef struct rbo_scene {
    rb_uniforms uniforms;          /* spheres_count / bvh_* counts are overwritten */
    const rb_sphere* spheres;      size_t n_spheres;
    const rb_point_light* lights;  size_t n_lights;   /* 0 => one zero-filled phantom (buffers.rs:232-240) */
    const rb_mesh* meshes;         size_t n_meshes;
    const rb_bvh_node* nodes;      size_t n_nodes;
    const uint32_t* indices;       size_t n_indices;
    const rb_gpu_triangle* tris;   size_t n_tris;
    const float* uvs;              size_t n_uvs;
    const rb_texture* textures;    size_t n_textures;
    uint32_t samples_per_pass;     /* reference constant 1 (gpu_wrapper.rs:12); 0 => 1 */
    uint32_t counts_kept;          /* update_uniforms' patch-up rule (gpu_wrapper.rs:475-495): a count is overwritten with
                                      its vector's length when that field came as Create / Update, and the caller's value
                                      in `uniforms` STAYS IN FORCE when the field was Keep.  Bit 0: spheres_count, bit 1:
                                      bvh_node_count, bit 2: bvh_triangle_count were kept (0 = all three from the lengths).
                                      A kept count beyond its array is clamped to the length (WGSL leaves such reads to
                                      robust buffer access; the library clamps the same way). */
} rbo_scene;

typedef struct rbo_stats {
    uint64_t segments;
    uint64_t paths;
    uint64_t nodes_popped;
    uint64_t tris_tested;
    uint64_t spheres_tested;
    uint64_t lights_tested;
    uint64_t mesh_hits;
} rbo_stats;

/* shader.wgsl:417-426 */
uint32_t rbo_hash(uint32_t seed);
float rbo_random_float(uint32_t* seed);
/* shader.wgsl:394-400 */
void rbo_hash_to_color(uint32_t n, float rgb[3]);
/* shader.wgsl:137-151 */
uint32_t rbo_color_map(const float rgb[3]);
/* shader.wgsl:193-215 */
float rbo_intersect_sphere(const float o[3], const float d[3], const float center[3], float radius);
/* shader.wgsl:248-280: returns t (or -1) and u,v */
float rbo_intersect_triangle(const float o[3], const float d[3], const float v0[3],
                             const float v1[3], const float v2[3], float* u, float* v);
/* shader.wgsl:664-671 */
int rbo_intersect_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3]);
/* shader.wgsl:402-414 */
float rbo_intersect_ground(const float o[3], const float d[3], float ground_height);
/* shader.wgsl:153-191 (scene supplies textures / checkerboard uniforms) */
void rbo_sample_texture(const rbo_scene* s, int32_t index, const float uv[2], float rgb[3]);
/* shader.wgsl:699-709: primary ray for pixel (x,y) with jitter offsets */
void rbo_primary_ray(const rb_uniforms* u, uint32_t x, uint32_t y, float off_x, float off_y,
                     float origin[3], float dir[3]);
/* shader.wgsl:522-662 */
void rbo_trace_ray(const rbo_scene* s, const float origin[3], const float dir[3], uint32_t seed,
                   float rgb[3], rbo_stats* st);

/* shader.wgsl:673-723 for rows [row_begin,row_end), passes
 * [first_pass, first_pass+n_passes) of the pass loop in
 * gpu_wrapper.rs:413-423.  accum: width*height vec4<f32> (in/out, shader
 * order); output: width*height packed u32 (out, shader order).  n_threads <= 0
 * => all cores.  Returns 0 or a negative error. */
int rbo_render(const rbo_scene* s, uint32_t first_pass, uint32_t n_passes,
               uint32_t row_begin, uint32_t row_end, float* accum, uint32_t* output,
               rbo_stats* stats, int n_threads);

/* Same, restricted to columns [col_begin, col_end) as well (for bounded CPU-baseline samples). */
int rbo_render_window(const rbo_scene* s, uint32_t first_pass, uint32_t n_passes, uint32_t col_begin,
                      uint32_t col_end, uint32_t row_begin, uint32_t row_end, float* accum, uint32_t* output,
                      rbo_stats* stats, int n_threads);

/* gpu_wrapper.rs:432-463: packed u32 (shader order) -> RGBA8 with x reversed, A=255 */
void rbo_read_pixels(const uint32_t* output, uint32_t width, uint32_t height, uint8_t* rgba);

/* engine-bvh/src/bvh.rs:87-150.  nodes_out may be NULL to query *n_nodes. */
int rbo_bvh_build(const rb_gpu_triangle* tris, size_t n_tris, rb_bvh_node* nodes_out,
                  size_t nodes_capacity, size_t* n_nodes, uint32_t* indices_out);

int rbo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
