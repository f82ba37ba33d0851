// engine.hpp -- C++ host-side mirror of the reference's renderer interface over
// the C ABI (rb_abi.h).  Header-only; the reference's toolchain (Rust) is absent
// from this image, so this is the compiled-language host layer a C++ caller uses
// and the template for the Rust shim in INTEGRATION.md.
//
//   Change<T>, RenderConfig        crates/engine-config/src/render_config.rs:37-57,99-109
//   trait Renderer                 crates/engine-config/src/renderer.rs:35-66
//   Engine::new / render / frame_iterator
//                                  crates/engine-pathtracer/src/lib.rs:58-119
//   Frame, trait FrameIterator     crates/frame-buffer/src/frame_iterator.rs:3-51
//
// Errors are exceptions carrying the rb_abi.h status and the library's message
// (the reference returns anyhow::Error or panics).
#pragma once

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../rb_abi.h"

namespace renderbaby {

struct RenderError : std::runtime_error {
    int code;
    RenderError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// enum Change<T> { Keep, Create(T), Update(T), Delete }
template <typename T>
struct Change {
    uint32_t tag = RB_KEEP;
    T value{};
    static Change keep() { return {}; }
    static Change create(T v) { return {RB_CREATE, std::move(v)}; }
    static Change update(T v) { return {RB_UPDATE, std::move(v)}; }
    static Change remove() { Change c; c.tag = RB_DELETE; return c; }
};

struct TextureData {  // crates/engine-config/src/texture.rs:26-36
    uint32_t width = 0, height = 0;
    std::vector<uint32_t> rgba_data;
};

struct RenderConfig {
    Change<rb_uniforms> uniforms;
    Change<std::vector<rb_sphere>> spheres;
    Change<std::vector<float>> uvs;
    Change<std::vector<rb_mesh>> meshes;
    Change<std::vector<rb_point_light>> lights;
    Change<std::vector<rb_bvh_node>> bvh_nodes;
    Change<std::vector<uint32_t>> bvh_indices;
    Change<std::vector<rb_gpu_triangle>> bvh_triangles;
    Change<std::vector<TextureData>> textures;
};

// struct Frame { width, height, pixels: Vec<u8> } -- RGBA8, x mirrored, A = 255
struct Frame {
    size_t width = 0, height = 0;
    std::vector<uint8_t> pixels;
    size_t expected_size() const { return width * height * 4; }
    void validate() const {
        if (pixels.size() != expected_size())
            throw RenderError(0, "Frame pixel size mismatch: expected " + std::to_string(expected_size()) +
                                     " bytes, got " + std::to_string(pixels.size()));
    }
};

// trait FrameIterator: Send + 'static { has_next, next, destroy }
struct FrameIterator {
    virtual ~FrameIterator() = default;
    virtual bool has_next() const = 0;
    virtual Frame next() = 0;
    virtual void destroy() = 0;
};

// trait Renderer: Send { render, frame_iterator }
struct Renderer {
    virtual ~Renderer() = default;
    virtual Frame render(const RenderConfig& rc) = 0;
    virtual std::unique_ptr<FrameIterator> frame_iterator(const RenderConfig& rc) = 0;
};

namespace detail {
// Borrowed view of a RenderConfig as rb_config; lives for one call.
struct Marshal {
    rb_config c{};
    std::vector<rb_texture> tex;
    explicit Marshal(const RenderConfig& rc) {
        c.uniforms = {rc.uniforms.tag, &rc.uniforms.value, 1};
        if (rc.uniforms.tag == RB_KEEP || rc.uniforms.tag == RB_DELETE) c.uniforms = {rc.uniforms.tag, nullptr, 0};
        auto vec = [](auto& ch) { return rb_field{ch.tag, ch.value.empty() ? nullptr : ch.value.data(), ch.value.size()}; };
        c.spheres = vec(rc.spheres);
        c.uvs = vec(rc.uvs);
        c.meshes = vec(rc.meshes);
        c.lights = vec(rc.lights);
        c.bvh_nodes = vec(rc.bvh_nodes);
        c.bvh_indices = vec(rc.bvh_indices);
        c.bvh_triangles = vec(rc.bvh_triangles);
        for (const auto& t : rc.textures.value) tex.push_back(rb_texture{t.width, t.height, t.rgba_data.data()});
        c.textures = {rc.textures.tag, tex.empty() ? nullptr : tex.data(), tex.size()};
    }
};
}  // namespace detail

// engine_pathtracer::Engine for the HIP backend.  Like the reference's
// Arc<Mutex<GpuWrapper>>, the handle is shared with the iterator and every call is
// serialised inside the library; Engine is safe to move across threads.
class Engine final : public Renderer {
    struct Handle {
        rb_engine* e;
        explicit Handle(rb_engine* p) : e(p) {}
        ~Handle() { rb_destroy(e); }
    };
    std::shared_ptr<Handle> h_;

    static void check(rb_engine* e, int rc) {
        if (rc != RB_OK) throw RenderError(rc, rb_last_error(e));
    }
    static Frame make_frame(rb_engine* e) {
        uint32_t w = 0, h = 0;
        check(e, rb_get_size(e, &w, &h));
        Frame f;
        f.width = w;
        f.height = h;
        f.pixels.resize(static_cast<size_t>(w) * h * 4);
        return f;
    }

    class Iter final : public FrameIterator {
        std::shared_ptr<Handle> h_;
      public:
        explicit Iter(std::shared_ptr<Handle> h) : h_(std::move(h)) {}
        bool has_next() const override { return rb_iter_has_next(h_->e) != 0; }
        Frame next() override {
            if (!has_next()) throw RenderError(RB_ERR_NO_MORE_FRAMES, "No more frames available");
            Frame f = make_frame(h_->e);
            check(h_->e, rb_iter_next(h_->e, f.pixels.data()));
            return f;
        }
        void destroy() override { rb_iter_destroy(h_->e); }
    };
    // extension: one frame per `n` samples instead of per sample (rb_iter_set_passes_per_frame)
  public:
    void set_passes_per_frame(uint32_t n) { check(h_->e, rb_iter_set_passes_per_frame(h_->e, n)); }
  private:

  public:
    // Engine::new(rc)
    explicit Engine(const RenderConfig& rc, const rb_options* opt = nullptr) {
        detail::Marshal m(rc);
        rb_engine* e = rb_create_ex(&m.c, opt);
        if (!e) throw RenderError(RB_ERR_DEVICE, rb_last_error(nullptr));
        h_ = std::make_shared<Handle>(e);
    }
    Frame render(const RenderConfig& rc) override {
        detail::Marshal m(rc);
        check(h_->e, rb_update(h_->e, &m.c));
        Frame f = make_frame(h_->e);
        check(h_->e, rb_render(h_->e, f.pixels.data()));
        return f;
    }
    std::unique_ptr<FrameIterator> frame_iterator(const RenderConfig& rc) override {
        detail::Marshal m(rc);
        check(h_->e, rb_iter_begin(h_->e, &m.c));
        return std::make_unique<Iter>(h_);
    }
    rb_engine* raw() const { return h_->e; }
};

}  // namespace renderbaby
