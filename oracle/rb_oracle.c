/*
 * rb_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see rb_oracle.h).
 *
 * One C function per live WGSL function of
 * crates/engine-pathtracer/src/shader.wgsl, same operation order.  Build with
 * -O2 -ffp-contract=off and no fast-math (oracle/Makefile).
 *
 * Parity status vs the reference's own tests: UNPINNED (the reference has no
 * test or golden vector for this path); pinned by source-derived KATs only.
 */
#include "rb_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- vec3 -- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 Vp(const float* p) { v3 r = {p[0], p[1], p[2]}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize(v3 a) { return divs(a, sqrtf(dot(a, a))); }

/* IEEE minNum / maxNum (a NaN operand loses) */
static inline float fmin_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (b < a ? b : a)); }
static inline float fmax_(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? b : a)); }
static inline float clampf(float e, float lo, float hi) { return fmin_(fmax_(e, lo), hi); }

/* WGSL u32(f32) / i32(f32): truncate, saturate; NaN -> 0 */
static inline uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u; /* NaN, negatives, zero */
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static inline int32_t f2i(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)f;
}

/* ----------------------------------------------------------------- RNG -- */
/* shader.wgsl:417-421 */
uint32_t rbo_hash(uint32_t seed) {
    uint32_t state = seed * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
/* shader.wgsl:423-426 */
float rbo_random_float(uint32_t* seed) {
    *seed = rbo_hash(*seed);
    return (float)(*seed) / 4294967296.0f;
}

/* shader.wgsl:429-441 */
static v3 random_in_unit_sphere(uint32_t* seed) {
    for (;;) {
        float px = rbo_random_float(seed) * 2.0f - 1.0f;
        float py = rbo_random_float(seed) * 2.0f - 1.0f;
        float pz = rbo_random_float(seed) * 2.0f - 1.0f;
        v3 p = V(px, py, pz);
        if (dot(p, p) < 1.0f) return p;
    }
}
/* shader.wgsl:444-446 */
static v3 random_unit_vector(uint32_t* seed) { return normalize(random_in_unit_sphere(seed)); }

/* -------------------------------------------------------- colour output -- */
/* shader.wgsl:137-142 */
static inline float linear_to_gamma(float c) { return (c > 0.0f) ? sqrtf(c) : 0.0f; }
/* shader.wgsl:144-151 */
uint32_t rbo_color_map(const float rgb[3]) {
    uint32_t r = f2u(linear_to_gamma(rgb[0]) * 255.999f);
    uint32_t g = f2u(linear_to_gamma(rgb[1]) * 255.999f);
    uint32_t b = f2u(linear_to_gamma(rgb[2]) * 255.999f);
    return (255u << 24) | (b << 16) | (g << 8) | r;
}
/* shader.wgsl:394-400 */
void rbo_hash_to_color(uint32_t n, float rgb[3]) {
    uint32_t h = n * 2654435761u;
    rgb[0] = (float)(h % 41u) / 40.0f;
    rgb[1] = (float)(h % 29u) / 28.0f;
    rgb[2] = (float)(h % 19u) / 18.0f;
}

/* ------------------------------------------------------------- textures -- */
/* shader.wgsl:153-191; texture flattening per buffers.rs:151-168 (offsets in
 * texels) is implicit here because each texture keeps its own array. */
void rbo_sample_texture(const rbo_scene* s, int32_t index, const float uv[2], float rgb[3]) {
    const rb_uniforms* un = &s->uniforms;
    if (index < 0) {
        if (un->checkerboard_enabled > 0u) {
            const float n = 10.0f;
            int32_t u2 = f2i(floorf(uv[0] * n));
            int32_t v2 = f2i(floorf(uv[1] * n));
            int32_t sum = (int32_t)((uint32_t)u2 + (uint32_t)v2);
            const float* c = (sum % 2 == 0) ? un->checkerboard_color_1 : un->checkerboard_color_2;
            rgb[0] = c[0]; rgb[1] = c[1]; rgb[2] = c[2];
        } else {
            rgb[0] = rgb[1] = rgb[2] = 1.0f;
        }
        return;
    }
    if ((size_t)index >= s->n_textures) { /* out of contract; robust-access stand-in */
        rgb[0] = rgb[1] = rgb[2] = 0.0f;
        return;
    }
    const rb_texture* t = &s->textures[index];
    float u = uv[0] - floorf(uv[0]); /* fract */
    float v = uv[1] - floorf(uv[1]);
    uint32_t x = f2u(u * (float)t->width);
    if (x > t->width - 1u) x = t->width - 1u;
    uint32_t y = f2u((1.0f - v) * (float)t->height);
    if (y > t->height - 1u) y = t->height - 1u;
    uint32_t pixel = t->rgba_data[(size_t)y * t->width + x];
    float r = (float)(pixel & 255u) / 255.0f;
    float g = (float)((pixel >> 8) & 255u) / 255.0f;
    float b = (float)((pixel >> 16) & 255u) / 255.0f;
    rgb[0] = powf(r, 2.2f);
    rgb[1] = powf(g, 2.2f);
    rgb[2] = powf(b, 2.2f);
}

/* --------------------------------------------------------- intersection -- */
/* shader.wgsl:193-215 (intersect_pointlight :217-239 is the same math) */
static inline float isect_sphere(v3 o, v3 d, v3 center, float radius) {
    v3 oc = sub(o, center);
    float a = dot(d, d);
    float half_b = dot(oc, d);
    float c = dot(oc, oc) - radius * radius;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return -1.0f;
    float sqrtd = sqrtf(disc);
    float root = (-half_b - sqrtd) / a;
    if (root <= 0.001f) {
        root = (-half_b + sqrtd) / a;
        if (root <= 0.001f) return -1.0f;
    }
    return root;
}
float rbo_intersect_sphere(const float o[3], const float d[3], const float c[3], float r) {
    return isect_sphere(Vp(o), Vp(d), Vp(c), r);
}

/* shader.wgsl:248-280 */
static inline float isect_triangle(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float* uo, float* vo) {
    v3 edge1 = sub(v1, v0);
    v3 edge2 = sub(v2, v0);
    v3 h = cross(d, edge2);
    float a = dot(edge1, h);
    *uo = 0.0f; *vo = 0.0f;
    if (fabsf(a) < 1e-6f) return -1.0f;
    float f = 1.0f / a;
    v3 s = sub(o, v0);
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return -1.0f;
    v3 q = cross(s, edge1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return -1.0f;
    float t = f * dot(edge2, q);
    if (t > 0.0f) { *uo = u; *vo = v; return t; }
    return -1.0f;
}
float rbo_intersect_triangle(const float o[3], const float d[3], const float v0[3],
                             const float v1[3], const float v2[3], float* u, float* v) {
    return isect_triangle(Vp(o), Vp(d), Vp(v0), Vp(v1), Vp(v2), u, v);
}

/* shader.wgsl:664-671 */
static inline int isect_aabb(v3 o, v3 d, v3 bmin, v3 bmax) {
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    v3 t0 = mul(sub(bmin, o), inv);
    v3 t1 = mul(sub(bmax, o), inv);
    float tmin = fmax_(fmax_(fmin_(t0.x, t1.x), fmin_(t0.y, t1.y)), fmin_(t0.z, t1.z));
    float tmax = fmin_(fmin_(fmax_(t0.x, t1.x), fmax_(t0.y, t1.y)), fmax_(t0.z, t1.z));
    return tmax >= fmax_(tmin, 0.0f);
}
int rbo_intersect_aabb(const float o[3], const float d[3], const float bmin[3], const float bmax[3]) {
    return isect_aabb(Vp(o), Vp(d), Vp(bmin), Vp(bmax));
}

/* shader.wgsl:402-414 */
static inline float isect_ground(v3 o, v3 d, float ground_height) {
    if (fabsf(d.y) < 1e-6f) return -1.0f;
    float t = (ground_height - o.y) / d.y;
    if (t > 0.0f) return t;
    return -1.0f;
}
float rbo_intersect_ground(const float o[3], const float d[3], float gh) {
    return isect_ground(Vp(o), Vp(d), gh);
}

/* ------------------------------------------------------------ HitRecord -- */
/* shader.wgsl:53-76: only diffuse, specular, shininess, emissive and
 * texture_index of Material are ever read by the shader; ambient is written
 * (:365, :560) but never read (:612 is a comment) -- carried all the same, so
 * that every assignment of the shader has its line here. */
typedef struct {
    int hit;
    float t;
    v3 pos, normal;
    float uv[2];
    int use_texture;
    v3 ambient, diffuse, specular, emissive;
    float shininess;
    int32_t texture_index;
} hitrec;

static inline void hit_reset(hitrec* h) { /* shader.wgsl:283-297, 535-549 */
    h->hit = 0; h->t = 1e20f;
    h->pos = V(0, 0, 0); h->normal = V(0, 0, 0);
    h->uv[0] = h->uv[1] = 0.0f; h->use_texture = 0;
    h->ambient = V(0, 0, 0); h->diffuse = V(0, 0, 0); h->specular = V(0, 0, 0); h->emissive = V(0, 0, 0);
    h->shininess = 0.0f; h->texture_index = -1;
}
static inline void hit_set_material(hitrec* h, const rb_material* m) {
    h->ambient = Vp(m->ambient); h->diffuse = Vp(m->diffuse); h->specular = Vp(m->specular); h->emissive = Vp(m->emissive);
    h->shininess = m->shininess; h->texture_index = m->texture_index;
}

typedef struct {
    const rbo_scene* s;
    uint32_t spheres_count, node_count, tri_count, index_len, light_len;
    const rb_point_light* lights;
    const uint32_t* indices;
} ctx_t;

static const rb_point_light k_phantom_light; /* zero-filled (buffers.rs:232-240) */
static const uint32_t k_zero_index = 0u;

static inline float uv_at(const rbo_scene* s, uint32_t i) { /* OOB read => 0 (robust access stand-in) */
    return ((size_t)i < s->n_uvs) ? s->uvs[i] : 0.0f;
}

/* shader.wgsl:282-392 */
static void intersect_bvh(const ctx_t* c, v3 o, v3 d, hitrec* hit, rbo_stats* st) {
    const rbo_scene* s = c->s;
    hit_reset(hit);
    uint32_t stack[1024];
    int sp = 0;
    if (c->node_count == 0u) return;
    stack[sp++] = 0u;
    for (;;) {
        if (sp == 0) break;
        sp--;
        uint32_t node_idx = stack[sp];
        if (node_idx >= c->node_count) continue;
        const rb_bvh_node* node = &s->nodes[node_idx];
        st->nodes_popped++;
        if (!isect_aabb(o, d, Vp(node->aabb_min), Vp(node->aabb_max))) continue;
        if (node->primitive_count > 0u) {
            for (uint32_t i = 0; i < node->primitive_count; i++) {
                uint32_t tri_idx = node->first_primitive + i;
                if (tri_idx >= c->index_len) continue;
                uint32_t bvh_tri_idx = c->indices[tri_idx];
                if (bvh_tri_idx >= c->tri_count) continue;
                const rb_gpu_triangle* tri = &s->tris[bvh_tri_idx];
                v3 v0 = Vp(tri->v0), v1 = Vp(tri->v1), v2 = Vp(tri->v2);
                float u, v;
                st->tris_tested++;
                float t = isect_triangle(o, d, v0, v1, v2, &u, &v);
                if (t > 0.001f && t < hit->t) {
                    hit->hit = 1;
                    hit->t = t;
                    hit->pos = add(o, scale(t, d));
                    hit->normal = normalize(cross(sub(v1, v0), sub(v2, v0)));
                    float w = 1.0f - u - v;
                    float uv0x = uv_at(s, tri->v0_index * 2u), uv0y = uv_at(s, tri->v0_index * 2u + 1u);
                    float uv1x = uv_at(s, tri->v1_index * 2u), uv1y = uv_at(s, tri->v1_index * 2u + 1u);
                    float uv2x = uv_at(s, tri->v2_index * 2u), uv2y = uv_at(s, tri->v2_index * 2u + 1u);
                    hit->uv[0] = (w * uv0x + u * uv1x) + v * uv2x;
                    hit->uv[1] = (w * uv0y + u * uv1y) + v * uv2y;
                    st->mesh_hits++;
                    if (s->uniforms.color_hash_enabled != 0u) {
                        float rgb[3];
                        rbo_hash_to_color(bvh_tri_idx + 1u, rgb);
                        hit->diffuse = Vp(rgb);
                        hit->ambient = V(0, 0, 0);      /* :365 */
                        hit->specular = V(0, 0, 0);
                        hit->use_texture = 0;
                    } else {
                        if ((size_t)tri->mesh_index < s->n_meshes) {
                            hit_set_material(hit, &s->meshes[tri->mesh_index].material);
                        } else { /* out of contract: zero material */
                            rb_material z; memset(&z, 0, sizeof z);
                            hit_set_material(hit, &z);
                        }
                        hit->use_texture = hit->texture_index >= 0;
                    }
                }
            }
        } else {
            if (node->left < c->node_count) { if (sp < 1024) stack[sp++] = node->left; }
            if (node->right < c->node_count) { if (sp < 1024) stack[sp++] = node->right; }
        }
    }
}

/* ------------------------------------------------------------- scatter -- */
/* shader.wgsl:459-461 */
static inline v3 reflect_vector(v3 v, v3 n) { return sub(v, scale(2.0f * dot(v, n), n)); }
/* shader.wgsl:463-466 */
static inline int near_zero(v3 v) {
    const float s = 1e-8f;
    return (fabsf(v.x) < s) && (fabsf(v.y) < s) && (fabsf(v.z) < s);
}
/* shader.wgsl:468-479 */
static v3 scatter_lambertian(v3 normal, uint32_t* seed) {
    v3 dir = add(normal, random_unit_vector(seed));
    if (near_zero(dir)) return normal;
    return normalize(dir);
}
/* shader.wgsl:481-490 */
static v3 scatter_metal(v3 ray_dir, v3 normal, float fuzz, uint32_t* seed) {
    v3 reflected = reflect_vector(normalize(ray_dir), normal);
    return add(reflected, scale(fuzz, random_unit_vector(seed)));
}

/* ------------------------------------------------------------ trace_ray -- */
static void ctx_init(ctx_t* c, const rbo_scene* s) {
    c->s = s;
    /* gpu_wrapper.rs:475-495: counts from the vector lengths (Create / Update), or the caller's own value (Keep) */
    c->spheres_count = (uint32_t)s->n_spheres;
    c->node_count = (uint32_t)s->n_nodes;
    c->tri_count = (uint32_t)s->n_tris;   /* shader.wgsl:336 compares triangle ids with this */
    if ((s->counts_kept & 1u) && s->uniforms.spheres_count < c->spheres_count) c->spheres_count = s->uniforms.spheres_count;
    if ((s->counts_kept & 2u) && s->uniforms.bvh_node_count < c->node_count) c->node_count = s->uniforms.bvh_node_count;
    if ((s->counts_kept & 4u) && s->uniforms.bvh_triangle_count < c->tri_count) c->tri_count = s->uniforms.bvh_triangle_count;
    /* arrayLength(): an empty Vec still allocates one zero element (buffers.rs:232-240) */
    if (s->n_lights == 0) { c->lights = &k_phantom_light; c->light_len = 1u; }
    else { c->lights = s->lights; c->light_len = (uint32_t)s->n_lights; }
    if (s->n_indices == 0) { c->indices = &k_zero_index; c->index_len = 1u; }
    else { c->indices = s->indices; c->index_len = (uint32_t)s->n_indices; }
}

/* shader.wgsl:522-662 */
static v3 trace_ray(const ctx_t* c, v3 origin, v3 direction, uint32_t seed, rbo_stats* st) {
    const rbo_scene* s = c->s;
    const rb_uniforms* un = &s->uniforms;
    v3 color = V(0, 0, 0);
    v3 attenuation = V(1, 1, 1);
    for (uint32_t depth = 0; depth < un->max_depth; depth++) {
        hitrec closest;
        hit_reset(&closest);
        st->segments++;

        /* Ground :552-565 */
        if (un->ground_enabled > 0u) {
            float t = isect_ground(origin, direction, un->ground_height);
            if (t > 0.001f && t < closest.t) {
                closest.hit = 1;
                closest.t = t;
                closest.pos = add(origin, scale(t, direction));
                closest.normal = V(0.0f, 1.0f, 0.0f);
                closest.diffuse = V(0.5f, 0.5f, 0.5f);
                closest.ambient = V(0, 0, 0);       /* :560 */
                closest.specular = V(0, 0, 0);
                closest.uv[0] = closest.pos.x;
                closest.uv[1] = closest.pos.z;
                closest.use_texture = 1;
            }
        }

        /* BVH triangles :568-571 */
        {
            hitrec bh;
            intersect_bvh(c, origin, direction, &bh, st);
            if (bh.hit && bh.t < closest.t) closest = bh;
        }

        /* Spheres :574-586 */
        for (uint32_t k = 0; k < c->spheres_count; k++) {
            const rb_sphere* sp = &s->spheres[k];
            st->spheres_tested++;
            float t = isect_sphere(origin, direction, Vp(sp->center), sp->radius);
            if (t > 0.001f && t < closest.t) {
                closest.hit = 1;
                closest.t = t;
                closest.pos = add(origin, scale(t, direction));
                closest.normal = normalize(sub(closest.pos, Vp(sp->center)));
                hit_set_material(&closest, &sp->material);
                closest.use_texture = closest.texture_index >= 0;
            }
        }

        /* Point lights :590-601 (use_texture and uv are NOT reset) */
        for (uint32_t k = 0; k < c->light_len; k++) {
            const rb_point_light* pl = &c->lights[k];
            st->lights_tested++;
            float t = isect_sphere(origin, direction, Vp(pl->center), pl->radius);
            if (t > 0.001f && t < closest.t) {
                closest.hit = 1;
                closest.t = t;
                closest.pos = add(origin, scale(t, direction));
                closest.normal = normalize(sub(closest.pos, Vp(pl->center)));
                hit_set_material(&closest, &pl->material);
            }
        }

        /* Sky :604-608 */
        if (!closest.hit) {
            color = add(color, mul(attenuation, Vp(un->sky_color)));
            break;
        }

        float specular_strength = (closest.specular.x + closest.specular.y + closest.specular.z) / 3.0f;
        float diffuse_strength = (closest.diffuse.x + closest.diffuse.y + closest.diffuse.z) / 3.0f;
        int is_metal = specular_strength > 0.01f && diffuse_strength < 0.01f;

        /* emitted light :626 */
        color = add(color, mul(attenuation, closest.emissive));

        v3 scattered, albedo;
        if (is_metal) {
            float fuzz = clampf(1.0f - (closest.shininess / 1000.0f), 0.0f, 1.0f);
            scattered = scatter_metal(direction, closest.normal, fuzz, &seed);
            if (dot(scattered, closest.normal) <= 0.0f) break;
            albedo = closest.specular;
        } else {
            scattered = scatter_lambertian(closest.normal, &seed);
            albedo = closest.diffuse;
            if (closest.use_texture) {
                float rgb[3];
                rbo_sample_texture(s, closest.texture_index, closest.uv, rgb);
                albedo = mul(albedo, Vp(rgb));
            }
        }
        attenuation = mul(attenuation, albedo);
        origin = add(closest.pos, scale(0.001f, closest.normal));
        direction = normalize(scattered);
    }
    return color;
}

void rbo_trace_ray(const rbo_scene* s, const float origin[3], const float dir[3], uint32_t seed,
                   float rgb[3], rbo_stats* st) {
    ctx_t c; ctx_init(&c, s);
    rbo_stats local; memset(&local, 0, sizeof local);
    v3 col = trace_ray(&c, Vp(origin), Vp(dir), seed, st ? st : &local);
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
}

/* ----------------------------------------------------------------- main -- */
/* shader.wgsl:690,699-709 */
static inline void primary_ray(const rb_uniforms* un, uint32_t x, uint32_t y, float off_x, float off_y,
                               v3* origin, v3* dir) {
    float aspect = (float)un->width / (float)un->height;
    float u = ((((float)x + off_x) / (float)(un->width - 1u)) * 2.0f - 1.0f) * aspect;
    float v = 1.0f - (((float)y + off_y) / (float)(un->height - 1u)) * 2.0f;
    v3 camera_pos = Vp(un->camera.pos);
    v3 camera_forward = normalize(Vp(un->camera.dir));
    v3 world_up = V(0.0f, 1.0f, 0.0f);
    v3 camera_right = normalize(cross(world_up, camera_forward));
    v3 camera_up = cross(camera_forward, camera_right);
    float fov = un->camera.pane_width / (2.0f * un->camera.pane_distance * aspect);
    v3 a = scale(fov * u, camera_right);
    v3 b = scale(fov * v, camera_up);
    *dir = normalize(add(add(a, b), camera_forward));
    *origin = camera_pos;
}
void rbo_primary_ray(const rb_uniforms* u, uint32_t x, uint32_t y, float off_x, float off_y,
                     float origin[3], float dir[3]) {
    v3 o, d;
    primary_ray(u, x, y, off_x, off_y, &o, &d);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

static inline void stats_add(rbo_stats* a, const rbo_stats* b) {
    a->segments += b->segments; a->paths += b->paths; a->nodes_popped += b->nodes_popped;
    a->tris_tested += b->tris_tested; a->spheres_tested += b->spheres_tested;
    a->lights_tested += b->lights_tested; a->mesh_hits += b->mesh_hits;
}

int rbo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* shader.wgsl:673-723 over the pass loop of gpu_wrapper.rs:413-423.  The
 * reference runs pass-major (one dispatch per pass over all pixels); pixels
 * are independent, so pixel-major order here is the same arithmetic. */
int rbo_render(const rbo_scene* s, uint32_t first_pass, uint32_t n_passes,
               uint32_t row_begin, uint32_t row_end, float* accum, uint32_t* output,
               rbo_stats* stats, int n_threads) {
    return rbo_render_window(s, first_pass, n_passes, 0u, 0xFFFFFFFFu, row_begin, row_end, accum, output, stats,
                             n_threads);
}

int rbo_render_window(const rbo_scene* s, uint32_t first_pass, uint32_t n_passes, uint32_t col_begin,
                      uint32_t col_end, uint32_t row_begin, uint32_t row_end, float* accum, uint32_t* output,
                      rbo_stats* stats, int n_threads) {
    if (!s || !accum || !output) return -1;
    const rb_uniforms* un = &s->uniforms;
    if (row_end > un->height) row_end = un->height;
    ctx_t c; ctx_init(&c, s);
    const uint32_t spp = s->samples_per_pass ? s->samples_per_pass : 1u;
    const uint32_t width = un->width;
    rbo_stats total; memset(&total, 0, sizeof total);
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    (void)n_threads;
#pragma omp parallel num_threads(n_threads)
    {
        rbo_stats st; memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 1)
        for (int64_t yy = (int64_t)row_begin; yy < (int64_t)row_end; yy++) {
            uint32_t y = (uint32_t)yy;
            for (uint32_t x = col_begin; x < (col_end < width ? col_end : width); x++) {
                uint32_t pixel_index = y * width + x;
                float* acc = accum + (size_t)pixel_index * 4u;
                v3 accumulated = V(acc[0], acc[1], acc[2]);
                uint32_t total_samples = f2u(acc[3]);
                for (uint32_t pass = first_pass; pass < first_pass + n_passes; pass++) {
                    for (uint32_t sample = 0; sample < spp; sample++) {
                        uint32_t sample_offset = pass * spp + sample;
                        uint32_t seed = rbo_hash(pixel_index + rbo_hash(sample_offset));
                        float off_x = rbo_random_float(&seed) - 0.5f;
                        float off_y = rbo_random_float(&seed) - 0.5f;
                        v3 o, d;
                        primary_ray(un, x, y, off_x, off_y, &o, &d);
                        v3 col = trace_ray(&c, o, d, seed, &st);
                        st.paths++;
                        accumulated = add(accumulated, col);
                        total_samples = total_samples + 1u;
                    }
                }
                acc[0] = accumulated.x; acc[1] = accumulated.y; acc[2] = accumulated.z;
                acc[3] = (float)total_samples;
                float ts = (float)total_samples;
                v3 fin = divs(accumulated, ts);
                float mapped[3];
                mapped[0] = fin.x / (fin.x + 1.0f);
                mapped[1] = fin.y / (fin.y + 1.0f);
                mapped[2] = fin.z / (fin.z + 1.0f);
                output[pixel_index] = rbo_color_map(mapped);
            }
        }
#pragma omp critical
        stats_add(&total, &st);
    }
    if (stats) stats_add(stats, &total);
    return 0;
}

/* gpu_wrapper.rs:446-458 */
void rbo_read_pixels(const uint32_t* output, uint32_t width, uint32_t height, uint8_t* rgba) {
    size_t o = 0;
    for (uint32_t y = 0; y < height; y++) {
        for (uint32_t xi = 0; xi < width; xi++) {
            uint32_t x = width - 1u - xi;
            uint32_t p = output[(size_t)y * width + x];
            rgba[o++] = (uint8_t)(p & 255u);
            rgba[o++] = (uint8_t)((p >> 8) & 255u);
            rgba[o++] = (uint8_t)((p >> 16) & 255u);
            rgba[o++] = 255u;
        }
    }
}

/* ------------------------------------------------------------ BVH build -- */
/* engine-bvh/src/bvh.rs:87-150.  select_nth_unstable_by leaves the order
 * inside each half unspecified, so a different but equally valid partition
 * is produced by this quickselect; parity is defined on hits, not tree bytes. */
typedef struct {
    const rb_gpu_triangle* tris;
    uint32_t* indices;
    rb_bvh_node* nodes;
    size_t cap, count;
    int overflow;
} bvhb;

static inline float centroid_axis(const rb_gpu_triangle* t, int axis) {
    /* glam: (v0 + v1 + v2) / 3.0 component-wise (bvh.rs:152-154) */
    return ((t->v0[axis] + t->v1[axis]) + t->v2[axis]) / 3.0f;
}

static void select_nth(bvhb* b, size_t lo, size_t hi /*exclusive*/, size_t nth, int axis) {
    /* iterative quickselect with median-of-three pivot */
    uint32_t* a = b->indices;
    while (hi - lo > 1) {
        size_t mid = lo + (hi - lo) / 2;
        float cl = centroid_axis(&b->tris[a[lo]], axis);
        float cm = centroid_axis(&b->tris[a[mid]], axis);
        float ch = centroid_axis(&b->tris[a[hi - 1]], axis);
        float pivot = (cl < cm) ? ((cm < ch) ? cm : (cl < ch ? ch : cl)) : ((cl < ch) ? cl : (cm < ch ? ch : cm));
        size_t i = lo, j = hi - 1;
        for (;;) {
            while (centroid_axis(&b->tris[a[i]], axis) < pivot) i++;
            while (centroid_axis(&b->tris[a[j]], axis) > pivot) j--;
            if (i >= j) break;
            uint32_t tmp = a[i]; a[i] = a[j]; a[j] = tmp;
            i++; if (j == 0) break; j--;
        }
        /* now [lo, j] <= pivot <= [j+1, hi) (Hoare); keep both sides non-empty */
        if (j >= hi - 1) j = hi - 2;
        if (nth <= j) hi = j + 1; else lo = j + 1;
    }
}

static uint32_t build_node(bvhb* b, size_t first, size_t count) {
    uint32_t node_index = (uint32_t)b->count;
    b->count++;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = first; i < first + count; i++) {
        const rb_gpu_triangle* t = &b->tris[b->indices[i]];
        const float* vs[3] = {t->v0, t->v1, t->v2};
        for (int k = 0; k < 3; k++)
            for (int ax = 0; ax < 3; ax++) {
                if (vs[k][ax] < mn[ax]) mn[ax] = vs[k][ax];
                if (vs[k][ax] > mx[ax]) mx[ax] = vs[k][ax];
            }
    }
    rb_bvh_node n; memset(&n, 0, sizeof n);
    memcpy(n.aabb_min, mn, sizeof mn);
    memcpy(n.aabb_max, mx, sizeof mx);
    if (count <= 128) {
        n.first_primitive = (uint32_t)first;
        n.primitive_count = (uint32_t)count;
        if (b->nodes) { if (node_index < b->cap) b->nodes[node_index] = n; else b->overflow = 1; }
        return node_index;
    }
    float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    int axis = (ex > ey && ex > ez) ? 0 : ((ey > ez) ? 1 : 2);
    size_t mid = first + count / 2;
    select_nth(b, first, first + count, mid, axis);
    uint32_t left = build_node(b, first, mid - first);
    uint32_t right = build_node(b, mid, first + count - mid);
    n.left = left; n.right = right;
    if (b->nodes) { if (node_index < b->cap) b->nodes[node_index] = n; else b->overflow = 1; }
    return node_index;
}

int rbo_bvh_build(const rb_gpu_triangle* tris, size_t n_tris, rb_bvh_node* nodes_out,
                  size_t nodes_capacity, size_t* n_nodes, uint32_t* indices_out) {
    if (!n_nodes) return -1;
    if (n_tris == 0) { *n_nodes = 0; return 0; } /* the adapter never builds an empty tree (scene_engine_adapter.rs:435-440) */
    if (!tris) return -1;
    uint32_t* idx = indices_out;
    int own = 0;
    if (!idx) { idx = (uint32_t*)malloc(sizeof(uint32_t) * (n_tris ? n_tris : 1)); own = 1; if (!idx) return -2; }
    for (size_t i = 0; i < n_tris; i++) idx[i] = (uint32_t)i;
    bvhb b; b.tris = tris; b.indices = idx; b.nodes = nodes_out; b.cap = nodes_capacity; b.count = 0; b.overflow = 0;
    build_node(&b, 0, n_tris);
    *n_nodes = b.count;
    if (own) free(idx);
    return b.overflow ? -3 : 0;
}
