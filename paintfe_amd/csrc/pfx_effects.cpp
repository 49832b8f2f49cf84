// pfx_effects.cpp — C ABI for the effects built from the hot-path kernels: sharpen, glow, bokeh,
// motion blur.  Host-side constants follow the reference expression by expression (f32, no contraction).
#include <algorithm>
#include <cmath>
#include <vector>

#include "pfx_internal.h"

namespace {

int check2(pfx_ctx* ctx, const void* a, const void* b, uint32_t w, uint32_t h, const char* who)
{
    if (!ctx) return PFX_ERR_INVALID;
    if (!a || !b) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: null image pointer", who);
    if (w == 0 || h == 0 || (uint64_t)w * h > 256000000ull) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: bad image size %ux%u", who, w, h);
    // the effect kernels read neighbourhoods / gather from src while other workgroups write dst: the buffers must not overlap (pfx.h)
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    const size_t bytes = (size_t)w * h * 4;
    if (x < y + bytes && y < x + bytes) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: src and dst overlap", who);
    return pfx_use(ctx);
}

inline int32_t f32_as_i32(float v) { return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)v)); }

// The Gaussian inside a composite effect: bit-exact unless the caller opted out (pfx_internal.h: gauss_fast_effects).  The reference's tests hold these
// effects at tolerance 0 (tests/visual_filters.rs:43-55,154-165), and the default-mode Gaussian's +-1 LSB would be multiplied by `amount` / `intensity`.
struct exact_gauss_scope {
    pfx_ctx* c; bool saved;
    explicit exact_gauss_scope(pfx_ctx* ctx) : c(ctx), saved(ctx->exact) { if (!ctx->gauss_fast_effects) ctx->exact = true; }
    ~exact_gauss_scope() { c->exact = saved; }
};
int effect_gaussian(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius)
{
    exact_gauss_scope g(ctx);
    return pfx_gaussian_blur_dev(ctx, src_dev, dst_dev, w, h, radius, nullptr);
}

// Gaussian of src into the context scratch, then the two-input pass (stylize.rs: `blurred = parallel_gaussian_blur_pub(flat, radius)`)
int blur_then_combine(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, int op, float p0,
                      const void* mask_dev, const char* timer)
{
    const size_t bytes = (size_t)w * h * 4;
    {   // small radii: the bit-exact Gaussian and the combine in one kernel — the blurred image never exists in memory
        exact_gauss_scope g(ctx);
        if (pfx_int_gauss_exact_combine_applies(ctx, src_dev, dst_dev, w, h, radius)) {   // the timer only around a launch that happens
            pfx_timer t(ctx, timer);
            const int ran = pfx_int_gauss_exact_combine(ctx, src_dev, dst_dev, w, h, radius, op == PFXK_FX_SHARPEN ? 1 : 2, p0, mask_dev);
            if (ran < 0) return ran;
            if (ran == 1) return PFX_OK;
        }
    }
    PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, bytes));
    PFX_TRY(effect_gaussian(ctx, src_dev, ctx->st_aux2.p, w, h, radius));
    pfx_timer t(ctx, timer);
    PFX_HIP(ctx, pfxk_combine(ctx->stream, (const uint8_t*)src_dev, (const uint8_t*)ctx->st_aux2.p, (const uint8_t*)mask_dev,
                              (uint8_t*)dst_dev, w, h, op, p0));
    return PFX_OK;
}

template <class F>
int host_wrap(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, const uint8_t* mask, const char* who, F&& dev_call)
{
    PFX_TRY(check2(ctx, src, dst, w, h, who));
    const size_t bytes = (size_t)w * h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, bytes));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, bytes));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, bytes));
    const void* d_mask = nullptr;
    if (mask) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_mask, (size_t)w * h));
        PFX_TRY(pfx_h2d(ctx, ctx->st_mask.p, mask, (size_t)w * h));
        d_mask = ctx->st_mask.p;
    }
    PFX_TRY(dev_call(ctx->st_in.p, ctx->st_out.p, d_mask));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, bytes));
    return pfx_sync(ctx);
}

} // namespace

extern "C" {

int pfx_sharpen_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float radius, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_sharpen_dev"));
    return blur_then_combine(ctx, src_dev, dst_dev, w, h, radius, PFXK_FX_SHARPEN, amount, mask_dev, "sharpen");
}

int pfx_glow_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, float intensity, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_glow_dev"));
    return blur_then_combine(ctx, src_dev, dst_dev, w, h, radius, PFXK_FX_GLOW, intensity, mask_dev, "glow");
}

int pfx_bokeh_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_bokeh_blur_dev"));
    if (radius < 0.5f) { // blur.rs:23: flat.clone()
        PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, (size_t)w * h * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return PFX_OK;
    }
    // one horizontal span per disc row (blur.rs:35-47)
    const int32_t r = f32_as_i32(ceilf(radius));
    PFX_REQUIRE(ctx, r <= 1500, "bokeh radius too large");
    const float r2 = radius * radius;
    std::vector<int32_t> spans;
    size_t sample_count = 0;
    for (int32_t dy = -r; dy <= r; ++dy) {
        const float remaining = r2 - (float)(dy * dy);
        if (remaining >= 0.0f) {
            const int32_t span = f32_as_i32(floorf(sqrtf(remaining)));
            spans.push_back(dy);
            spans.push_back(span);
            sample_count += (size_t)(span * 2 + 1);
        }
    }
    const float inv_count = 1.0f / (float)sample_count;
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, std::max<size_t>(spans.size() * 4, 64)));
    PFX_TRY(pfx_h2d(ctx, ctx->d_misc.p, spans.data(), spans.size() * 4));
    pfx_timer t(ctx, "bokeh");
    PFX_HIP(ctx, pfxk_bokeh(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, (const int32_t*)ctx->d_misc.p,
                            (int)(spans.size() / 2), inv_count, w, h));
    return PFX_OK;
}

int pfx_motion_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float angle_deg, float distance,
                        const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_motion_blur_dev"));
    if (distance < 1.0f) { // blur.rs:150: flat.clone()
        PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, (size_t)w * h * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return PFX_OK;
    }
    const float angle = angle_deg * (3.14159265358979323846f / 180.0f); // f32::to_radians
    const int32_t steps = f32_as_i32(ceilf(distance));
    // 2 * steps + 1 samples per pixel: a bound keeps a hostile distance from turning into a launch that never ends (a document side is <= 25 000 px, io.rs:500)
    if (steps > 65536) return pfx_fail(ctx, PFX_ERR_UNSUPPORTED, "motion blur distance %g above 65536 is not supported", (double)distance);
    const float dx = cosf(angle), dy = sinf(angle);                     // glibc, as Rust's f32::cos / sin on Linux
    const float inv_steps = 1.0f / (float)(steps * 2 + 1);
    pfx_timer t(ctx, "motion_blur");
    PFX_HIP(ctx, pfxk_motion(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, steps, dx, dy, inv_steps, w, h));
    return PFX_OK;
}

int pfx_sharpen_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float amount, float radius, const uint8_t* mask)
{
    return host_wrap(ctx, src, dst, w, h, mask, "pfx_sharpen_core",
                     [&](const void* s, void* d, const void* m) { return pfx_sharpen_dev(ctx, s, d, w, h, amount, radius, m); });
}

int pfx_glow_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius, float intensity, const uint8_t* mask)
{
    return host_wrap(ctx, src, dst, w, h, mask, "pfx_glow_core",
                     [&](const void* s, void* d, const void* m) { return pfx_glow_dev(ctx, s, d, w, h, radius, intensity, m); });
}

int pfx_bokeh_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius, const uint8_t* mask)
{
    return host_wrap(ctx, src, dst, w, h, mask, "pfx_bokeh_blur_core",
                     [&](const void* s, void* d, const void* m) { return pfx_bokeh_blur_dev(ctx, s, d, w, h, radius, m); });
}

int pfx_motion_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float angle_deg, float distance, const uint8_t* mask)
{
    return host_wrap(ctx, src, dst, w, h, mask, "pfx_motion_blur_core",
                     [&](const void* s, void* d, const void* m) { return pfx_motion_blur_dev(ctx, s, d, w, h, angle_deg, distance, m); });
}

} // extern "C"

// =====================================================================================================================
// The rest of the effect bank (k_effects2.hip).  Each *_dev entry point restates the reference's per-call constants on
// the host (f32, same association order), fills the kernel's parameter block and launches; *_core wraps it with staging.
// =====================================================================================================================
namespace {

inline uint32_t pack4(const uint8_t c[4]) { return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24); }
inline float to_radians(float deg) { return deg * (3.14159265358979323846f / 180.0f); } // f32::to_radians
inline float rs_clampf(float x, float lo, float hi) { if (x < lo) x = lo; if (x > hi) x = hi; return x; }
inline uint32_t clamp_u32(uint32_t v, uint32_t lo, uint32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

int launch_fx(pfx_ctx* ctx, int fx, const char* timer, const void* src_dev, void* dst_dev, const void* mask_dev, const pfxk_fx_params& P,
              uint32_t w, uint32_t h)
{
    pfx_timer t(ctx, timer);
    PFX_HIP(ctx, pfxk_fx(ctx->stream, fx, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, &P, w, h));
    return PFX_OK;
}

int copy_through(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h) // `flat.clone()`
{
    if (src_dev != dst_dev) PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, (size_t)w * h * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return PFX_OK;
}

// mersenne-free hash of effects.rs:143-161 (crystallize seed points are built on the host like the reference does)
inline uint32_t hash_u32(uint32_t x)
{
    x *= 0x9E3779B9u; x ^= x >> 16;
    x *= 0x85EBCA6Bu; x ^= x >> 13;
    x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
inline float hash_f32(uint32_t x, uint32_t y, uint32_t seed)
{
    return (float)(hash_u32(x * 374761393u + y * 668265263u + seed) & 0x00FFFFFFu) / 16777216.0f;
}

} // namespace

#define PFX_UNPAREN(...) __VA_ARGS__
#define PFX_FX_CORE(name, DECL, CALL)                                                                                                    \
    int pfx_##name##_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, PFX_UNPAREN DECL, const uint8_t* mask)   \
    {                                                                                                                                    \
        return host_wrap(ctx, src, dst, w, h, mask, "pfx_" #name "_core",                                                                 \
                         [&](const void* s, void* d, const void* m) { return pfx_##name##_dev(ctx, s, d, w, h, PFX_UNPAREN CALL, m); });  \
    }

extern "C" {

int pfx_zoom_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float center_x, float center_y, float strength,
                      uint32_t samples, const float tint_color[4], float tint_strength, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_zoom_blur_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_zoom_blur_dev: in-place not supported");
    if (strength < 0.001f) return copy_through(ctx, src_dev, dst_dev, w, h); // blur.rs:332
    PFX_REQUIRE(ctx, samples <= 4096, "zoom blur: too many samples");
    const float cx = center_x * (float)w, cy = center_y * (float)h;
    const uint32_t n = samples < 2 ? 2 : samples;
    const float fw = (float)w, fh = (float)h;
    const float corners[4][2] = {{cx, cy}, {fw - cx, cy}, {cx, fh - cy}, {fw - cx, fh - cy}};
    float max_dist = 0.0f;
    for (auto& c : corners) max_dist = fmaxf(max_dist, sqrtf(c[0] * c[0] + c[1] * c[1]));
    max_dist = fmaxf(max_dist, 1.0f);
    pfxk_fx_params P{};
    P.f[0] = cx; P.f[1] = cy; P.f[2] = rs_clampf(strength, 0.0f, 0.99f); P.f[3] = 1.0f / (float)n; P.f[4] = max_dist;
    for (int c = 0; c < 4; ++c) P.f[5 + c] = tint_color ? tint_color[c] * 255.0f : 0.0f;
    P.f[9] = tint_color ? tint_strength : 0.0f;
    P.i[0] = (int32_t)n;
    return launch_fx(ctx, PFXK_FX2_ZOOM, "zoom_blur", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_crystallize_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float cell_size, uint32_t seed,
                        const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_crystallize_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_crystallize_dev: in-place not supported");
    const float cs = fmaxf(cell_size, 2.0f);
    const int32_t cells_x = std::max(f32_as_i32(ceilf((float)w / cs)), 1), cells_y = std::max(f32_as_i32(ceilf((float)h / cs)), 1);
    const size_t n = (size_t)cells_x * cells_y;
    std::vector<float> seeds(n * 2); // distort.rs:52-61
    for (int32_t cy = 0; cy < cells_y; ++cy)
        for (int32_t cx = 0; cx < cells_x; ++cx) {
            const float base_x = (float)cx * cs, base_y = (float)cy * cs;
            const float jx = hash_f32((uint32_t)cx, (uint32_t)cy, seed), jy = hash_f32((uint32_t)cx, (uint32_t)cy, seed + 77u);
            const size_t i = (size_t)cy * cells_x + cx;
            seeds[i * 2] = base_x + jx * cs;
            seeds[i * 2 + 1] = base_y + jy * cs;
        }
    const size_t seeds_bytes = n * 8, acc_bytes = n * 40, avg_bytes = n * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->fx_a, seeds_bytes + acc_bytes + avg_bytes));
    uint8_t* base = (uint8_t*)ctx->fx_a.p;
    PFX_TRY(pfx_h2d(ctx, base + acc_bytes, seeds.data(), seeds_bytes));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `seeds` is pageable host memory about to go out of scope
    pfx_timer t(ctx, "crystallize");
    PFX_HIP(ctx, pfxk_crystallize(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, (const float*)(base + acc_bytes),
                                  (unsigned long long*)base, (uint32_t*)(base + acc_bytes + seeds_bytes), cells_x, cells_y, cs, w, h));
    return PFX_OK;
}

int pfx_dents_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float scale, float amount, uint32_t seed,
                  uint32_t octaves, float roughness, int pinch, int wrap, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_dents_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_dents_dev: in-place not supported");
    pfxk_fx_params P{};
    P.f[0] = 1.0f / fmaxf(scale, 0.5f); P.f[1] = amount; P.f[2] = scale; P.f[3] = roughness;
    P.u[0] = seed;
    P.i[0] = (int32_t)clamp_u32(octaves, 1, 8); P.i[1] = pinch ? 1 : 0; P.i[2] = wrap ? 1 : 0;
    return launch_fx(ctx, PFXK_FX2_DENTS, "dents", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_bulge_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float origin_x, float origin_y,
                  const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_bulge_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_bulge_dev: in-place not supported");
    const float fw = (float)w, fh = (float)h; // distort.rs:406-411
    const float cx = rs_clampf(origin_x, 0.0f, 1.0f) * fmaxf(fw - 1.0f, 0.0f), cy = rs_clampf(origin_y, 0.0f, 1.0f) * fmaxf(fh - 1.0f, 0.0f);
    pfxk_fx_params P{};
    P.f[0] = cx; P.f[1] = cy;
    P.f[2] = fmaxf(fmaxf(fmaxf(cx, fw - cx), fmaxf(cy, fh - cy)), 1.0f);
    P.f[3] = fmaxf(fabsf(amount), 0.0001f);
    P.f[4] = amount;
    return launch_fx(ctx, PFXK_FX2_BULGE, "bulge", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_twist_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float angle_deg, float origin_x, float origin_y,
                  const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_twist_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_twist_dev: in-place not supported");
    const float fw = (float)w, fh = (float)h; // distort.rs:470-477
    const float cx = rs_clampf(origin_x, 0.0f, 1.0f) * fmaxf(fw - 1.0f, 0.0f), cy = rs_clampf(origin_y, 0.0f, 1.0f) * fmaxf(fh - 1.0f, 0.0f);
    const float mx = fmaxf(cx, fw - cx), my = fmaxf(cy, fh - cy);
    pfxk_fx_params P{};
    P.f[0] = cx; P.f[1] = cy;
    P.f[2] = fmaxf(sqrtf(mx * mx + my * my), 1.0f);
    P.f[3] = to_radians(angle_deg);
    return launch_fx(ctx, PFXK_FX2_TWIST, "twist", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_add_noise_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, int noise_type, int monochrome,
                      uint32_t seed, float scale, uint32_t octaves, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_add_noise_dev"));
    PFX_REQUIRE(ctx, noise_type >= PFX_NOISE_UNIFORM && noise_type <= PFX_NOISE_PERLIN, "pfx_add_noise_dev: unknown noise type");
    pfxk_fx_params P{};
    P.f[0] = 1.0f / fmaxf(scale, 0.1f);
    P.f[1] = amount * 255.0f / 100.0f; // noise.rs:107
    P.i[0] = noise_type; P.i[1] = monochrome ? 1 : 0; P.i[2] = (int32_t)clamp_u32(octaves, 1, 8);
    P.u[0] = seed;
    return launch_fx(ctx, PFXK_FX2_NOISE, "add_noise", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_reduce_noise_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float strength, uint32_t radius,
                         const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_reduce_noise_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_reduce_noise_dev: in-place not supported");
    PFX_REQUIRE(ctx, radius <= 64, "reduce_noise: radius above 64 is not supported");
    const int32_t r = radius < 1 ? 1 : (int32_t)radius;
    const float sigma_s = (float)r, sigma_r = strength * 2.55f; // noise.rs:184-186
    pfxk_fx_params P{};
    P.f[0] = 2.0f * sigma_s * sigma_s;
    P.f[1] = 2.0f * sigma_r * sigma_r + 0.001f;
    P.i[0] = r;
    return launch_fx(ctx, PFXK_FX2_REDUCE_NOISE, "reduce_noise", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_vignette_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float softness, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_vignette_dev"));
    const float cx = (float)w / 2.0f, cy = (float)h / 2.0f; // stylize.rs:176-181
    pfxk_fx_params P{};
    P.f[0] = cx; P.f[1] = cy; P.f[2] = sqrtf(cx * cx + cy * cy); P.f[3] = fmaxf(softness, 0.01f); P.f[4] = amount;
    return launch_fx(ctx, PFXK_FX2_VIGNETTE, "vignette", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_halftone_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float dot_size, float angle_deg, int shape,
                     const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_halftone_dev"));
    PFX_REQUIRE(ctx, shape >= PFX_HALFTONE_CIRCLE && shape <= PFX_HALFTONE_LINE, "pfx_halftone_dev: unknown shape");
    const float angle = to_radians(angle_deg); // stylize.rs:249-252 (glibc cosf / sinf like Rust's f32::cos / sin)
    pfxk_fx_params P{};
    P.f[0] = fmaxf(dot_size, 2.0f); P.f[1] = cosf(angle); P.f[2] = sinf(angle);
    P.i[0] = shape;
    return launch_fx(ctx, PFXK_FX2_HALFTONE, "halftone", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_grid_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t cell_w, uint32_t cell_h, uint32_t line_width,
                 const uint8_t color[4], int style, float opacity, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_grid_dev"));
    PFX_REQUIRE(ctx, color != nullptr && (style == PFX_GRID_LINES || style == PFX_GRID_CHECKERBOARD), "pfx_grid_dev: bad colour / style");
    pfxk_fx_params P{};
    P.u[0] = std::max(cell_w, 2u); P.u[1] = std::max(cell_h, 2u); P.u[2] = std::max(line_width, 1u); P.u[3] = pack4(color);
    P.i[0] = style;
    P.f[0] = opacity;
    return launch_fx(ctx, PFXK_FX2_GRID, "grid", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_canvas_border_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4],
                          const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_canvas_border_dev"));
    PFX_REQUIRE(ctx, color != nullptr, "pfx_canvas_border_dev: null colour");
    pfxk_fx_params P{};
    P.u[0] = std::min(std::max(width, 1u), std::min(w, h)); // render.rs:126
    P.u[1] = pack4(color);
    return launch_fx(ctx, PFXK_FX2_BORDER, "canvas_border", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_shadow_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int32_t offset_x, int32_t offset_y, float blur_radius,
                   int widen_radius, const uint8_t color[4], float opacity, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_shadow_dev"));
    PFX_REQUIRE(ctx, color != nullptr, "pfx_shadow_dev: null colour");
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_shadow_dev: in-place not supported");
    const size_t n = (size_t)w * h;
    int32_t spread = 0;
    if (widen_radius) spread = f32_as_i32(roundf(fmaxf(blur_radius, 1.0f))); // render.rs:251
    PFX_REQUIRE(ctx, spread <= 4096, "drop shadow: spread too large");
    // The reference expands the shadow's alpha plane to (a, a, a, a), blurs all four channels and reads one back (render.rs:291-301).  Where the bit-exact fused
    // Gaussian applies (the effect's default mode, radius <= 16) and rows are dword-aligned, the PLANE is blurred instead — per element the same products and sums
    // (k_gauss_exact.hip: gauss_plane_exact_kernel) — and the composite reads the plane; otherwise the RGBA image goes through the Gaussian as before.
    const int g_radius = blur_radius > 0.5f ? pfx_host_gaussian_radius(blur_radius) : 0;
    const bool plane_path = (w & 3u) == 0 && !ctx->gauss_fast_effects && ctx->shadow_plane_blur &&
                            (g_radius == 0 || (g_radius >= 1 && g_radius <= pfxk_gauss_fused_exact_max_radius()));
    PFX_TRY(pfx_reserve(ctx, ctx->fx_a, 2 * n + 8));
    if (!plane_path) PFX_TRY(pfx_reserve(ctx, ctx->fx_b, 4 * n));
    uint8_t* plane_a = (uint8_t*)ctx->fx_a.p;
    uint8_t* plane_b = plane_a + ((n + 3) & ~(size_t)3);       // dword-aligned second plane
    const void* alpha_img = plane_path ? (const void*)plane_a : ctx->fx_b.p;
    {
        pfx_timer t(ctx, "shadow_alpha");
        PFX_HIP(ctx, pfxk_shadow_alpha(ctx->stream, (const uint8_t*)src_dev, plane_a, plane_b, plane_path ? nullptr : (uint8_t*)ctx->fx_b.p,
                                       offset_x, offset_y, spread, w, h));
    }
    if (blur_radius > 0.5f) { // render.rs:297
        if (plane_path) {
            const float* wts = nullptr;
            PFX_TRY(pfx_int_gauss_exact_weights(ctx, blur_radius, &wts));
            pfx_timer t(ctx, "gauss_plane");
            PFX_HIP(ctx, pfxk_gauss_plane_exact(ctx->stream, plane_a, plane_b, wts, g_radius, w, h));
            alpha_img = plane_b;
        } else {
            PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, 4 * n));
            PFX_TRY(effect_gaussian(ctx, ctx->fx_b.p, ctx->st_aux2.p, w, h, blur_radius));
            alpha_img = ctx->st_aux2.p;
        }
    }
    pfxk_fx_params P{};
    P.f[0] = opacity;
    P.u[0] = pack4(color);
    P.u[1] = plane_path ? 1u : 0u;
    P.aux0 = alpha_img;
    return launch_fx(ctx, PFXK_FX2_SHADOW, "shadow_composite", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_outline_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4], int mode,
                    int anti_alias, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_outline_dev"));
    PFX_REQUIRE(ctx, color != nullptr && mode >= PFX_OUTLINE_OUTSIDE && mode <= PFX_OUTLINE_CENTER, "pfx_outline_dev: bad colour / mode");
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_outline_dev: in-place not supported");
    PFX_REQUIRE(ctx, width <= 256, "outline: width above 256 is not supported");
    const float radius = (float)std::max(width, 1u); // render.rs:417-418
    pfxk_fx_params P{};
    P.f[0] = radius;
    P.i[0] = f32_as_i32(ceilf(radius)) + 1; P.i[1] = mode; P.i[2] = anti_alias ? 1 : 0;
    P.u[0] = pack4(color);
    if (P.i[0] <= 15 && ctx->outline_bits) { // search windows up to 31 columns: nearest filled / empty texel from a bit plane of alpha != 0
        const uint32_t stride = pfxk_alpha_bits_stride(w);
        PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)h * stride * 4));
        PFX_HIP(ctx, pfxk_alpha_bits(ctx->stream, (const uint8_t*)src_dev, (uint32_t*)ctx->st_tmp.p, w, h));
        P.aux0 = ctx->st_tmp.p;
        P.i[3] = (int32_t)stride;
    }
    return launch_fx(ctx, PFXK_FX2_OUTLINE, "outline", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_pixel_drag_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t seed, float amount, uint32_t distance,
                       float direction, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_pixel_drag_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_pixel_drag_dev: in-place not supported");
    const float dir_rad = to_radians(direction); // glitch.rs:64-67
    pfxk_fx_params P{};
    P.f[0] = cosf(dir_rad); P.f[1] = sinf(dir_rad); P.f[2] = (float)std::max(distance, 1u); P.f[3] = amount / 100.0f;
    P.u[0] = seed;
    return launch_fx(ctx, PFXK_FX2_PIXEL_DRAG, "pixel_drag", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_rgb_displace_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int32_t r_dx, int32_t r_dy, int32_t g_dx,
                         int32_t g_dy, int32_t b_dx, int32_t b_dy, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_rgb_displace_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_rgb_displace_dev: in-place not supported");
    pfxk_fx_params P{};
    const int32_t off[6] = {r_dx, r_dy, g_dx, g_dy, b_dx, b_dy};
    for (int k = 0; k < 6; ++k) P.i[k] = std::max(-(1 << 28), std::min(1 << 28, off[k])); // far beyond any image: same clamped texel, no i32 overflow
    return launch_fx(ctx, PFXK_FX2_RGB_DISPLACE, "rgb_displace", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_ink_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float edge_strength, float threshold, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_ink_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_ink_dev: in-place not supported");
    pfxk_fx_params P{};
    P.f[0] = edge_strength; P.f[1] = threshold;
    return launch_fx(ctx, PFXK_FX2_INK, "ink", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_oil_painting_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius, uint32_t levels,
                         const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_oil_painting_dev"));
    PFX_REQUIRE(ctx, src_dev != dst_dev, "pfx_oil_painting_dev: in-place not supported");
    pfx_timer t(ctx, "oil_painting"); // artistic.rs:135-136
    PFX_HIP(ctx, pfxk_oil_painting(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, (int)clamp_u32(radius, 1, 10),
                                   (int)clamp_u32(levels, 2, 64), w, h));
    return PFX_OK;
}

int pfx_color_filter_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, const uint8_t filter_color[4], float intensity,
                         int mode, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_color_filter_dev"));
    PFX_REQUIRE(ctx, filter_color != nullptr && mode >= PFX_COLOR_FILTER_MULTIPLY && mode <= PFX_COLOR_FILTER_SOFT_LIGHT,
                "pfx_color_filter_dev: bad colour / mode");
    pfxk_fx_params P{};
    for (int c = 0; c < 3; ++c) P.f[c] = (float)filter_color[c] / 255.0f; // artistic.rs:273-277
    P.f[3] = intensity;
    P.i[0] = mode;
    return launch_fx(ctx, PFXK_FX2_COLOR_FILTER, "color_filter", src_dev, dst_dev, mask_dev, P, w, h);
}

int pfx_contours_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float scale, float frequency, float line_width,
                     const uint8_t line_color[4], uint32_t seed, uint32_t octaves, float blend, const void* mask_dev)
{
    PFX_TRY(check2(ctx, src_dev, dst_dev, w, h, "pfx_contours_dev"));
    PFX_REQUIRE(ctx, line_color != nullptr, "pfx_contours_dev: null colour");
    const float inv_scale = 1.0f / fmaxf(scale, 0.5f); // contours.rs:73-82,95
    const float half_lw = fmaxf(line_width * 0.5f, 0.3f);
    pfxk_fx_params P{};
    P.f[0] = inv_scale; P.f[1] = fmaxf(frequency, 0.5f); P.f[2] = half_lw * inv_scale * 0.5f; P.f[3] = (float)line_color[3] / 255.0f; P.f[4] = blend;
    for (int c = 0; c < 3; ++c) P.f[5 + c] = (float)line_color[c];
    P.u[0] = seed;
    P.i[0] = (int32_t)clamp_u32(octaves, 1, 8);
    return launch_fx(ctx, PFXK_FX2_CONTOURS, "contours", src_dev, dst_dev, mask_dev, P, w, h);
}

PFX_FX_CORE(zoom_blur, (float center_x, float center_y, float strength, uint32_t samples, const float tint_color[4], float tint_strength),
            (center_x, center_y, strength, samples, tint_color, tint_strength))
PFX_FX_CORE(crystallize, (float cell_size, uint32_t seed), (cell_size, seed))
PFX_FX_CORE(dents, (float scale, float amount, uint32_t seed, uint32_t octaves, float roughness, int pinch, int wrap),
            (scale, amount, seed, octaves, roughness, pinch, wrap))
PFX_FX_CORE(bulge, (float amount, float origin_x, float origin_y), (amount, origin_x, origin_y))
PFX_FX_CORE(twist, (float angle_deg, float origin_x, float origin_y), (angle_deg, origin_x, origin_y))
PFX_FX_CORE(add_noise, (float amount, int noise_type, int monochrome, uint32_t seed, float scale, uint32_t octaves),
            (amount, noise_type, monochrome, seed, scale, octaves))
PFX_FX_CORE(reduce_noise, (float strength, uint32_t radius), (strength, radius))
PFX_FX_CORE(vignette, (float amount, float softness), (amount, softness))
PFX_FX_CORE(halftone, (float dot_size, float angle_deg, int shape), (dot_size, angle_deg, shape))
PFX_FX_CORE(grid, (uint32_t cell_w, uint32_t cell_h, uint32_t line_width, const uint8_t color[4], int style, float opacity),
            (cell_w, cell_h, line_width, color, style, opacity))
PFX_FX_CORE(canvas_border, (uint32_t width, const uint8_t color[4]), (width, color))
PFX_FX_CORE(shadow, (int32_t offset_x, int32_t offset_y, float blur_radius, int widen_radius, const uint8_t color[4], float opacity),
            (offset_x, offset_y, blur_radius, widen_radius, color, opacity))
PFX_FX_CORE(outline, (uint32_t width, const uint8_t color[4], int mode, int anti_alias), (width, color, mode, anti_alias))
PFX_FX_CORE(pixel_drag, (uint32_t seed, float amount, uint32_t distance, float direction), (seed, amount, distance, direction))
PFX_FX_CORE(rgb_displace, (int32_t r_dx, int32_t r_dy, int32_t g_dx, int32_t g_dy, int32_t b_dx, int32_t b_dy), (r_dx, r_dy, g_dx, g_dy, b_dx, b_dy))
PFX_FX_CORE(ink, (float edge_strength, float threshold), (edge_strength, threshold))
PFX_FX_CORE(oil_painting, (uint32_t radius, uint32_t levels), (radius, levels))
PFX_FX_CORE(color_filter, (const uint8_t filter_color[4], float intensity, int mode), (filter_color, intensity, mode))
PFX_FX_CORE(contours, (float scale, float frequency, float line_width, const uint8_t line_color[4], uint32_t seed, uint32_t octaves, float blend),
            (scale, frequency, line_width, line_color, seed, octaves, blend))

} // extern "C"
