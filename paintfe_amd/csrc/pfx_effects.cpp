// pfx_effects.cpp — C ABI for the effects built from the hot-path kernels: sharpen, glow, bokeh,
// motion blur.  Host-side constants follow the reference expression by expression (f32, no contraction).
#include <cmath>
#include <vector>

#include "pfx_internal.h"

namespace {

int check2(pfx_ctx* ctx, const void* a, const void* b, uint32_t w, uint32_t h, const char* who)
{
    if (!ctx) return PFX_ERR_INVALID;
    if (!a || !b) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: null image pointer", who);
    if (w == 0 || h == 0 || (uint64_t)w * h > 256000000ull) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: bad image size %ux%u", who, w, h);
    return pfx_use(ctx);
}

inline int32_t f32_as_i32(float v) { return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)v)); }

// Gaussian of src into the context scratch, then the two-input pass (stylize.rs: `blurred = parallel_gaussian_blur_pub(flat, radius)`)
int blur_then_combine(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, int op, float p0,
                      const void* mask_dev, const char* timer)
{
    const size_t bytes = (size_t)w * h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, bytes));
    PFX_TRY(pfx_gaussian_blur_dev(ctx, src_dev, ctx->st_aux2.p, w, h, radius, nullptr));
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
