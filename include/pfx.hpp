// pfx.hpp — C++17 host-side mirror of PaintFE's operator interface over the C ABI (include/pfx.h).
//
// Same names, argument meaning and failure behaviour as the reference's Rust types, so call sites and tests read
// like the reference's own:
//   pfx::GpuRenderer   <->  paintfe::gpu::GpuRenderer           (src/gpu/renderer.rs:212-947)
//   pfx::CanvasState   <->  paintfe::canvas::CanvasState::composite() over Layer{pixels, opacity, blend_mode, visible, mask}
//                                                               (src/canvas/canvas_state.rs:482, src/canvas/layers.rs:389-421)
//   pfx::ops::*        <->  the pure `_core` functions           (src/ops/filters.rs:130, effects/*.rs, transform.rs)
// Header-only; link with -lpfx.  Nothing here does pixel arithmetic.
#pragma once
#include <algorithm>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "pfx.h"

namespace pfx {

struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string& msg) : std::runtime_error(msg), status(st) {}
};

// image::RgbaImage: tight row-major straight-alpha RGBA8
struct RgbaImage {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    RgbaImage() = default;
    RgbaImage(uint32_t w, uint32_t h) : width(w), height(h), data((size_t)w * h * 4, 0) {}
    RgbaImage(uint32_t w, uint32_t h, std::vector<uint8_t> raw) : width(w), height(h), data(std::move(raw)) {}
    uint8_t* pixel(uint32_t x, uint32_t y) { return &data[((size_t)y * width + x) * 4]; }
    const uint8_t* pixel(uint32_t x, uint32_t y) const { return &data[((size_t)y * width + x) * 4]; }
    bool operator==(const RgbaImage& o) const { return width == o.width && height == o.height && data == o.data; }
};
using GrayImage = std::vector<uint8_t>; // w*h selection mask, 0 = leave the pixel alone

enum class BlendMode : uint8_t { // src/canvas/layers.rs:2-29, ids = to_u8 (:125-153)
    Normal = 0, Multiply, Screen, Additive, Reflect, Glow, ColorBurn, ColorDodge, Overlay, Difference, Negation, Lighten, Darken,
    Xor, Overwrite, HardLight, SoftLight, Exclusion, Subtract, Divide, LinearBurn, VividLight, LinearLight, PinLight, HardMix
};

class GpuRenderer {
public:
    // GpuRenderer::try_new (renderer.rs:261): None when no adapter
    static std::optional<GpuRenderer> try_new(int device = 0)
    {
        pfx_ctx* c = nullptr;
        if (pfx_ctx_create(device, &c) != PFX_OK) return std::nullopt;
        return GpuRenderer(c);
    }
    // GpuRenderer::new (renderer.rs:249): fails hard when nothing is available
    explicit GpuRenderer(int device = 0)
    {
        const int st = pfx_ctx_create(device, &ctx_);
        if (st != PFX_OK) throw Error(st, std::string("GpuRenderer: ") + pfx_last_error(nullptr));
    }
    GpuRenderer(GpuRenderer&& o) noexcept : ctx_(o.ctx_) { o.ctx_ = nullptr; }
    GpuRenderer& operator=(GpuRenderer&& o) noexcept { if (this != &o) { reset(); ctx_ = o.ctx_; o.ctx_ = nullptr; } return *this; }
    GpuRenderer(const GpuRenderer&) = delete;
    GpuRenderer& operator=(const GpuRenderer&) = delete;
    ~GpuRenderer() { reset(); }

    bool available() const { return ctx_ != nullptr; } // renderer.rs:241
    pfx_ctx* raw() const { return ctx_; }
    void set_exact(bool on) { check(pfx_ctx_set_exact(ctx_, on ? 1 : 0)); }

    // ---- filters (renderer.rs:915-947): failure => empty Vec, like the wgpu readback path (compositor.rs:781-788)
    std::vector<uint8_t> blur_rgba(const std::vector<uint8_t>& d, uint32_t w, uint32_t h, float sigma) const
    { return run(d, [&](uint8_t* o) { return pfx_blur_rgba(ctx_, d.data(), o, w, h, sigma); }); }
    std::vector<uint8_t> brightness_contrast_rgba(const std::vector<uint8_t>& d, uint32_t w, uint32_t h, float b, float c) const
    { return run(d, [&](uint8_t* o) { return pfx_brightness_contrast_rgba(ctx_, d.data(), o, w, h, b, c); }); }
    std::vector<uint8_t> hsl_rgba(const std::vector<uint8_t>& d, uint32_t w, uint32_t h, float hue, float sat, float light) const
    { return run(d, [&](uint8_t* o) { return pfx_hsl_rgba(ctx_, d.data(), o, w, h, hue, sat, light); }); }
    std::vector<uint8_t> invert_rgba(const std::vector<uint8_t>& d, uint32_t w, uint32_t h) const
    { return run(d, [&](uint8_t* o) { return pfx_invert_rgba(ctx_, d.data(), o, w, h); }); }
    std::optional<std::vector<uint8_t>> median_rgba(const std::vector<uint8_t>& d, uint32_t w, uint32_t h, uint32_t radius) const
    {
        std::vector<uint8_t> out(d.size());
        if (pfx_median_rgba(ctx_, d.data(), out.data(), w, h, radius) != PFX_OK) return std::nullopt; // None (renderer.rs:945)
        return out;
    }

    // ---- layer store + compositor (renderer.rs:324-586)
    void ensure_layer_texture(size_t idx, uint32_t w, uint32_t h, const std::vector<uint8_t>& d, uint64_t generation)
    { if (available()) (void)pfx_layer_upload(ctx_, (uint32_t)idx, w, h, d.data(), generation); }
    void update_layer_rect(size_t idx, uint32_t x, uint32_t y, uint32_t rw, uint32_t rh, const std::vector<uint8_t>& d)
    { if (available()) (void)pfx_layer_update_rect(ctx_, (uint32_t)idx, x, y, rw, rh, d.data()); }
    void remove_layer(size_t idx) { (void)pfx_layer_remove(ctx_, (uint32_t)idx); }
    void clear_layers() { (void)pfx_layer_clear(ctx_); }
    size_t active_texture_count() const { return pfx_layer_count(ctx_); }
    size_t active_texture_memory() const { return pfx_layer_memory(ctx_); }
    // composite(canvas_w, canvas_h, &[(layer_idx, opacity, visible, blend_mode_u8)]) -> Option<Vec<u8>>
    struct LayerInfo { size_t layer_idx; float opacity; bool visible; uint8_t blend_mode; };
    std::optional<std::vector<uint8_t>> composite(uint32_t w, uint32_t h, const std::vector<LayerInfo>& info)
    {
        if (!available()) return std::nullopt;
        std::vector<pfx_layer_info> li(info.size());
        for (size_t i = 0; i < info.size(); ++i) {
            li[i] = pfx_layer_info{};
            li[i].layer_idx = (uint32_t)info[i].layer_idx; li[i].opacity = info[i].opacity;
            li[i].visible = info[i].visible; li[i].blend_mode = info[i].blend_mode;
        }
        std::vector<uint8_t> out((size_t)w * h * 4);
        if (pfx_composite(ctx_, w, h, li.data(), (uint32_t)li.size(), out.data()) != PFX_OK) return std::nullopt;
        return out;
    }

    std::string last_error() const { return pfx_last_error(ctx_); }
    void check(int st) const { if (st != PFX_OK) throw Error(st, pfx_last_error(ctx_)); }

private:
    explicit GpuRenderer(pfx_ctx* c) : ctx_(c) {}
    void reset() { if (ctx_) pfx_ctx_destroy(ctx_); ctx_ = nullptr; }
    template <class F> std::vector<uint8_t> run(const std::vector<uint8_t>& d, F&& f) const
    {
        std::vector<uint8_t> out(d.size());
        if (f(out.data()) != PFX_OK) out.clear();
        return out;
    }
    pfx_ctx* ctx_ = nullptr;
};

// ---- document model: just enough of Layer / CanvasState to say `state.composite()` ----
struct Layer { // src/canvas/layers.rs:389-421
    std::string name;
    RgbaImage pixels;
    float opacity = 1.0f;
    BlendMode blend_mode = BlendMode::Normal;
    bool visible = true;
    std::optional<GrayImage> mask; // "conceal" alpha, applied when mask_enabled
    bool mask_enabled = false;
    // layers.rs `gpu_generation`: the renderer re-uploads a layer only when this changes (renderer.rs:336-342).  Every
    // Layer starts with a process-unique value; bump it (`mark_dirty`) after editing `pixels`.
    uint64_t gpu_generation = next_generation();
    void mark_dirty() { gpu_generation = next_generation(); }
    static uint64_t next_generation() { static uint64_t g = 0; return ++g; }
};

class CanvasState {
public:
    uint32_t width, height;
    std::vector<Layer> layers;
    size_t active_layer_index = 0;
    CanvasState(uint32_t w, uint32_t h) : width(w), height(h) { layers.push_back(Layer{"Background", RgbaImage(w, h)}); } // CanvasState::new
    // CanvasState::composite() (canvas_state.rs:482) on the device
    RgbaImage composite(GpuRenderer& gpu) const
    {
        std::vector<pfx_layer_info> li(layers.size());
        for (size_t i = 0; i < layers.size(); ++i) {
            const Layer& L = layers[i];
            gpu.check(pfx_layer_upload(gpu.raw(), (uint32_t)i, width, height, L.pixels.data.data(), L.gpu_generation));
            gpu.check(pfx_layer_set_mask(gpu.raw(), (uint32_t)i, (L.mask_enabled && L.mask) ? L.mask->data() : nullptr));
            li[i] = pfx_layer_info{};
            li[i].layer_idx = (uint32_t)i; li[i].opacity = L.opacity; li[i].visible = L.visible; li[i].blend_mode = (uint8_t)L.blend_mode;
        }
        RgbaImage out(width, height);
        gpu.check(pfx_composite(gpu.raw(), width, height, li.data(), (uint32_t)li.size(), out.data.data()));
        return out;
    }
};

// ---- project files: load_pfe / save_pfe (src/io.rs:242-499) over the library's document object ----
namespace io {
inline CanvasState load_pfe(const std::string& path) // io.rs:469: layers come back flattened to w*h RGBA8 (to_rgba_image)
{
    char why[512] = {0};
    pfx_project* p = pfx_project_load_file(path.c_str(), why, sizeof why);
    if (!p) throw Error(PFX_ERR_INVALID, std::string("PfeError: ") + why);
    CanvasState state(pfx_project_width(p), pfx_project_height(p));
    state.layers.clear();
    for (uint32_t i = 0; i < pfx_project_layer_count(p); ++i) {
        pfx_project_layer L{};
        pfx_project_layer_get(p, i, &L);
        Layer out{L.name, RgbaImage(state.width, state.height)};
        pfx_project_layer_pixels(p, i, out.pixels.data.data());
        out.opacity = L.opacity;
        out.blend_mode = (BlendMode)(L.blend_mode > 24 ? 0 : L.blend_mode); // BlendMode::from_u8
        out.visible = L.effectively_visible != 0;
        state.layers.push_back(std::move(out));
    }
    state.active_layer_index = pfx_project_active_layer(p);
    pfx_project_free(p);
    return state;
}
inline void save_pfe(const CanvasState& state, const std::string& path) // io.rs:242 (raster layers: a V1 file)
{
    pfx_project* p = pfx_project_new(state.width, state.height);
    if (!p) throw Error(PFX_ERR_INVALID, "save_pfe: bad document size");
    for (const Layer& L : state.layers)
        pfx_project_add_layer(p, L.name.c_str(), L.pixels.data.data(), L.opacity, (uint8_t)L.blend_mode, L.visible, PFX_LAYER_RASTER, nullptr);
    pfx_project_set_active_layer(p, (uint32_t)std::min<size_t>(state.active_layer_index, state.layers.size() - 1));
    const int st = pfx_project_save_file(p, path.c_str());
    pfx_project_free(p);
    if (st != PFX_OK) throw Error(st, "save_pfe: cannot write " + path);
}
} // namespace io

// ---- pure `_core` functions ----
namespace ops {
inline const uint8_t* m(const GrayImage* mask) { return mask ? mask->data() : nullptr; }
template <class F> RgbaImage apply(GpuRenderer& g, const RgbaImage& src, F&& f)
{
    RgbaImage out(src.width, src.height);
    g.check(f(out.data.data()));
    return out;
}
inline RgbaImage blur_with_selection_pub(GpuRenderer& g, const RgbaImage& flat, float sigma, const GrayImage* mask = nullptr) // filters.rs:130
{ return apply(g, flat, [&](uint8_t* o) { return pfx_gaussian_blur_core(g.raw(), flat.data.data(), o, flat.width, flat.height, sigma, m(mask)); }); }
inline RgbaImage parallel_gaussian_blur_pub(GpuRenderer& g, const RgbaImage& src, float sigma) { return blur_with_selection_pub(g, src, sigma); } // :238
inline RgbaImage box_blur_core(GpuRenderer& g, const RgbaImage& flat, float radius, const GrayImage* mask = nullptr)                       // blur.rs:233
{ return apply(g, flat, [&](uint8_t* o) { return pfx_box_blur_core(g.raw(), flat.data.data(), o, flat.width, flat.height, radius, m(mask)); }); }
inline RgbaImage median_core(GpuRenderer& g, const RgbaImage& flat, uint32_t radius, const GrayImage* mask = nullptr)                      // noise.rs:357
{ return apply(g, flat, [&](uint8_t* o) { return pfx_median_core(g.raw(), flat.data.data(), o, flat.width, flat.height, radius, m(mask)); }); }
inline RgbaImage pixelate_core(GpuRenderer& g, const RgbaImage& flat, uint32_t block, const GrayImage* mask = nullptr)                     // distort.rs:333
{ return apply(g, flat, [&](uint8_t* o) { return pfx_pixelate_core(g.raw(), flat.data.data(), o, flat.width, flat.height, block, m(mask)); }); }
inline RgbaImage adjust(GpuRenderer& g, const RgbaImage& flat, pfx_adjust_op op, const std::vector<float>& p, const uint8_t* lut = nullptr,
                        const GrayImage* mask = nullptr, pfx_sparse_mode sparse = PFX_FROM_FLAT)
{ return apply(g, flat, [&](uint8_t* o) { return pfx_adjust(g.raw(), flat.data.data(), o, flat.width, flat.height, op, p.data(), (uint32_t)p.size(), lut, m(mask), sparse); }); }
inline RgbaImage warp_mesh_catmull_rom(GpuRenderer& g, const RgbaImage& src, const std::vector<float>& orig_xy, const std::vector<float>& def_xy,
                                       uint32_t cols, uint32_t rows)                                                                    // transform.rs:1743
{ return apply(g, src, [&](uint8_t* o) { return pfx_warp_mesh_catmull_rom(g.raw(), src.data.data(), orig_xy.data(), def_xy.data(), cols, rows, src.width, src.height, o); }); }
inline RgbaImage warp_displacement_full(GpuRenderer& g, const RgbaImage& src, const std::vector<float>& disp_xy, uint32_t w, uint32_t h)    // transform.rs:1288
{
    RgbaImage out(w, h);
    g.check(pfx_warp_displacement(g.raw(), src.data.data(), src.width, src.height, disp_xy.data(), w, h, out.data.data()));
    return out;
}
} // namespace ops

} // namespace pfx
