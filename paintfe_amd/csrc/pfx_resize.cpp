// pfx_resize.cpp — C ABI of resize_image: `imageops::resize(&flat, new_w, new_h, filter)` (ref: src/ops/transform.rs:347-359,
// resize_image script function src/ops/scripting.rs:749-770, filter mapping :29-37,60-67).
//
// The sampling algorithm is the `image` crate's (0.25.9 per Cargo.lock; not part of the reference tree): per output sample a
// window of source samples around (o + 0.5) * ratio, kernel weights normalised by their f32 sum, vertical pass into f32 then
// horizontal pass with clamp and round-to-nearest.  The per-axis weight tables are built here on the host (sinf for
// Lanczos3 from glibc, as the crate's f32::sin resolves to on Linux) and the two passes run in k_resize.hip.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "pfx_internal.h"

namespace {

float k_box(float) { return 1.0f; }
float k_triangle(float x) { const float a = fabsf(x); return a < 1.0f ? 1.0f - a : 0.0f; }
float sinc(float t)
{
    const float a = t * 3.14159265358979323846f;
    return t == 0.0f ? 1.0f : sinf(a) / a;
}
float k_lanczos3(float x) { return fabsf(x) < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f; }
float k_catmullrom(float x) // bc_cubic_spline(x, 0.0, 0.5)
{
    const float b = 0.0f, c = 0.5f, a = fabsf(x);
    float k;
    if (a < 1.0f) k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f) k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    else k = 0.0f;
    return k / 6.0f;
}

struct AxisTable {
    std::vector<uint32_t> left, count, off;
    std::vector<float> wts;
};

void build_axis(uint32_t n_in, uint32_t n_out, int filter, AxisTable& t)
{
    float (*kernel)(float) = k_triangle;
    float support = 1.0f;
    switch (filter) {
    case PFX_RESIZE_NEAREST: kernel = k_box; support = 0.0f; break;
    case PFX_RESIZE_BICUBIC: kernel = k_catmullrom; support = 2.0f; break;
    case PFX_RESIZE_LANCZOS3: kernel = k_lanczos3; support = 3.0f; break;
    default: break;
    }
    const float ratio = (float)n_in / (float)n_out;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float src_support = support * sratio;
    t.left.resize(n_out);
    t.count.resize(n_out);
    t.off.resize(n_out);
    t.wts.clear();
    for (uint32_t o = 0; o < n_out; ++o) {
        float input = ((float)o + 0.5f) * ratio;
        int64_t l = (int64_t)floorf(input - src_support);
        l = std::min<int64_t>(std::max<int64_t>(l, 0), (int64_t)n_in - 1);
        int64_t r = (int64_t)ceilf(input + src_support);
        r = std::min<int64_t>(std::max<int64_t>(r, l + 1), (int64_t)n_in);
        input = input - 0.5f;
        t.left[o] = (uint32_t)l;
        t.count[o] = (uint32_t)(r - l);
        t.off[o] = (uint32_t)t.wts.size();
        float sum = 0.0f;
        const size_t base = t.wts.size();
        for (int64_t i = l; i < r; ++i) {
            const float w = kernel(((float)i - input) / sratio);
            t.wts.push_back(w);
            sum += w;
        }
        for (size_t k = base; k < t.wts.size(); ++k) t.wts[k] /= sum;
    }
}

} // namespace

extern "C" {

int pfx_resize_image_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, uint32_t new_w, uint32_t new_h, int filter)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src_dev && dst_dev && src_dev != dst_dev, "pfx_resize_image_dev: bad image pointers");
    PFX_REQUIRE(ctx, w && h && new_w && new_h && (uint64_t)w * h <= 256000000ull && (uint64_t)new_w * new_h <= 256000000ull, "pfx_resize_image_dev: bad size");
    PFX_REQUIRE(ctx, filter >= PFX_RESIZE_NEAREST && filter <= PFX_RESIZE_LANCZOS3, "pfx_resize_image_dev: unknown filter");
    PFX_REQUIRE(ctx, !pfx_ranges_overlap(src_dev, (size_t)w * h * 4, dst_dev, (size_t)new_w * new_h * 4), "pfx_resize_image_dev: src and dst overlap");
    PFX_TRY(pfx_use(ctx));
    if (new_w == w && new_h == h) { // the crate copies instead of resampling
        PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, (size_t)w * h * 4, hipMemcpyDeviceToDevice, ctx->stream));
        return PFX_OK;
    }
    AxisTable v, hz;
    build_axis(h, new_h, filter, v);
    build_axis(w, new_w, filter, hz);
    PFX_REQUIRE(ctx, v.wts.size() < 0xffffffffull && hz.wts.size() < 0xffffffffull, "pfx_resize_image_dev: weight table too large");
    // one parameter blob: [v.left | v.count | v.off | h.left | h.count | h.off | v.wts | h.wts]
    const size_t n_u32 = 3 * (size_t)new_h + 3 * (size_t)new_w;
    std::vector<uint32_t> blob(n_u32 + v.wts.size() + hz.wts.size());
    uint32_t* p = blob.data();
    auto put = [&](const std::vector<uint32_t>& a) { std::copy(a.begin(), a.end(), p); p += a.size(); };
    put(v.left); put(v.count); put(v.off); put(hz.left); put(hz.count); put(hz.off);
    std::memcpy(p, v.wts.data(), v.wts.size() * 4);
    std::memcpy(p + v.wts.size(), hz.wts.data(), hz.wts.size() * 4);
    PFX_TRY(pfx_reserve(ctx, ctx->fx_a, blob.size() * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->fx_a.p, blob.data(), blob.size() * 4));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `blob` is pageable host memory about to go out of scope
    // the widest source-column range one output tile's horizontal taps reach: decides between the fused kernel (vertical results in
    // LDS) and the two-pass path with its f32 intermediate in HBM (strong downscales)
    uint32_t span_max = 0;
    const uint32_t tile = (uint32_t)pfxk_resize_tile_cols();
    for (uint32_t ox0 = 0; ox0 < new_w; ox0 += tile) {
        const uint32_t last = std::min(ox0 + tile, new_w) - 1;
        span_max = std::max(span_max, hz.left[last] + hz.count[last] - hz.left[ox0]);
    }
    const bool fused = (size_t)span_max * 8 * 16 <= 64u * 1024u && !ctx->resize_two_pass;
    if (!fused) PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)w * new_h * 16));
    const uint32_t* d = (const uint32_t*)ctx->fx_a.p;
    const float* dw = (const float*)(d + n_u32);
    pfx_timer t(ctx, "resize");
    PFX_HIP(ctx, pfxk_resize(ctx->stream, (const uint8_t*)src_dev, fused ? nullptr : (float*)ctx->st_tmp.p, (uint8_t*)dst_dev, d, d + new_h, d + 2 * (size_t)new_h,
                             dw, d + 3 * (size_t)new_h, d + 3 * (size_t)new_h + new_w, d + 3 * (size_t)new_h + 2 * (size_t)new_w, dw + v.wts.size(), w, h,
                             new_w, new_h, fused ? span_max : 0u));
    return PFX_OK;
}

// apply_affine (ref: src/ops/transform.rs:826-946): the homography is built and inverted on the host exactly as the reference does
int pfx_affine_transform_dev(pfx_ctx* ctx, const void* src_dev, uint32_t src_w, uint32_t src_h, void* dst_dev, uint32_t canvas_w, uint32_t canvas_h,
                             float rotation_z, float rotation_x, float rotation_y, float scale, float offset_x, float offset_y, int interpolation)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src_dev && dst_dev && src_dev != dst_dev, "pfx_affine_transform_dev: bad image pointers");
    PFX_REQUIRE(ctx, canvas_w && canvas_h && (uint64_t)canvas_w * canvas_h <= 256000000ull && (uint64_t)src_w * src_h <= 256000000ull,
                "pfx_affine_transform_dev: bad size");
    PFX_REQUIRE(ctx, interpolation == PFX_RESIZE_NEAREST || interpolation == PFX_RESIZE_BILINEAR, "pfx_affine_transform_dev: nearest or bilinear only");
    PFX_REQUIRE(ctx, !pfx_ranges_overlap(src_dev, (size_t)src_w * src_h * 4, dst_dev, (size_t)canvas_w * canvas_h * 4), "pfx_affine_transform_dev: src and dst overlap");
    PFX_TRY(pfx_use(ctx));
    pfxk_affine_params P{};
    P.cx = (float)canvas_w * 0.5f;
    P.cy = (float)canvas_h * 0.5f;
    P.off_x = offset_x;
    P.off_y = offset_y;
    P.inv_scale = fabsf(scale) > 1e-6f ? 1.0f / scale : 1.0f;
    P.nearest = interpolation == PFX_RESIZE_NEAREST;
    const float focal = (float)std::max(canvas_w, canvas_h) * 1.5f;
    const float rad = 3.14159265358979323846f / 180.0f; // f32::to_radians
    const float az = rotation_z * rad, ax = rotation_x * rad, ay = rotation_y * rad;
    const float sz = sinf(az), cz = cosf(az), sxr = sinf(ax), cxr = cosf(ax), syr = sinf(ay), cyr = cosf(ay);
    const float r00 = cz * cyr, r01 = cz * syr * sxr - sz * cxr, r10 = sz * cyr, r11 = sz * syr * sxr + cz * cxr, r20 = -syr, r21 = cyr * sxr;
    const float a = focal * r00, b = focal * r01, c = 0.0f, d = focal * r10, e = focal * r11, f = 0.0f, g = r20, h = r21, i = focal;
    const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g); // invert_3x3, transform.rs:949-976
    if (fabsf(det) < 1e-12f) {
        const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(P.hi, id, sizeof id);
    } else {
        const float inv = 1.0f / det;
        P.hi[0] = (e * i - f * h) * inv; P.hi[1] = (c * h - b * i) * inv; P.hi[2] = (b * f - c * e) * inv;
        P.hi[3] = (f * g - d * i) * inv; P.hi[4] = (a * i - c * g) * inv; P.hi[5] = (c * d - a * f) * inv;
        P.hi[6] = (d * h - e * g) * inv; P.hi[7] = (b * g - a * h) * inv; P.hi[8] = (a * e - b * d) * inv;
    }
    pfx_timer t(ctx, "affine");
    PFX_HIP(ctx, pfxk_affine(ctx->stream, (const uint8_t*)src_dev, src_w, src_h, (uint8_t*)dst_dev, canvas_w, canvas_h, &P));
    return PFX_OK;
}

int pfx_affine_transform(pfx_ctx* ctx, const uint8_t* src, uint32_t src_w, uint32_t src_h, uint8_t* dst, uint32_t canvas_w, uint32_t canvas_h,
                         float rotation_z, float rotation_x, float rotation_y, float scale, float offset_x, float offset_y, int interpolation)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src && dst && pfx_dims_ok(src_w, src_h) && pfx_dims_ok(canvas_w, canvas_h), "pfx_affine_transform: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, (size_t)src_w * src_h * 4));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, (size_t)canvas_w * canvas_h * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, (size_t)src_w * src_h * 4));
    PFX_TRY(pfx_affine_transform_dev(ctx, ctx->st_in.p, src_w, src_h, ctx->st_out.p, canvas_w, canvas_h, rotation_z, rotation_x, rotation_y, scale, offset_x,
                                     offset_y, interpolation));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, (size_t)canvas_w * canvas_h * 4));
    return pfx_sync(ctx);
}

// flip / rotate one layer image (ref: src/ops/transform.rs flip_canvas_* / rotate_canvas_* = imageops::flip_* / rotate*)
int pfx_flip_rotate_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, int op)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src_dev && dst_dev && src_dev != dst_dev && w && h && (uint64_t)w * h <= 256000000ull, "pfx_flip_rotate_dev: bad arguments");
    PFX_REQUIRE(ctx, op >= PFX_CANVAS_FLIP_HORIZONTAL && op <= PFX_CANVAS_ROTATE_180, "pfx_flip_rotate_dev: unknown operation");
    PFX_REQUIRE(ctx, !pfx_ranges_overlap(src_dev, (size_t)w * h * 4, dst_dev, (size_t)w * h * 4), "pfx_flip_rotate_dev: src and dst overlap");
    PFX_TRY(pfx_use(ctx));
    static const int mode_of[5] = {0, 1, 3, 4, 2}; // FLIP_H, FLIP_V, ROTATE_90CW, ROTATE_90CCW, ROTATE_180 -> k_script.hip permutation modes
    pfx_timer t(ctx, "permute");
    PFX_HIP(ctx, pfxk_permute(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, mode_of[op], w, h));
    return PFX_OK;
}

int pfx_flip_rotate(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst, int op)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src && dst && pfx_dims_ok(w, h), "pfx_flip_rotate: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const size_t bytes = (size_t)w * h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, bytes));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, bytes));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, bytes));
    PFX_TRY(pfx_flip_rotate_dev(ctx, ctx->st_in.p, w, h, ctx->st_out.p, op));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, bytes));
    return pfx_sync(ctx);
}

// resize_canvas(state, new_w, new_h, anchor, fill) for one layer image (ref: src/ops/transform.rs:382-424)
int pfx_resize_canvas_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, uint32_t new_w, uint32_t new_h, uint32_t anchor_x,
                          uint32_t anchor_y, const uint8_t fill[4])
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src_dev && dst_dev && src_dev != dst_dev && w && h && new_w && new_h && (uint64_t)new_w * new_h <= 256000000ull,
                "pfx_resize_canvas_dev: bad arguments");
    PFX_REQUIRE(ctx, !pfx_ranges_overlap(src_dev, (size_t)w * h * 4, dst_dev, (size_t)new_w * new_h * 4), "pfx_resize_canvas_dev: src and dst overlap");
    PFX_TRY(pfx_use(ctx));
    const int32_t off_x = anchor_x == 0 ? 0 : (anchor_x == 1 ? ((int32_t)new_w - (int32_t)w) / 2 : (int32_t)new_w - (int32_t)w);
    const int32_t off_y = anchor_y == 0 ? 0 : (anchor_y == 1 ? ((int32_t)new_h - (int32_t)h) / 2 : (int32_t)new_h - (int32_t)h);
    const uint32_t rgba = fill ? ((uint32_t)fill[0] | ((uint32_t)fill[1] << 8) | ((uint32_t)fill[2] << 16) | ((uint32_t)fill[3] << 24)) : 0u;
    pfx_timer t(ctx, "resize_canvas");
    PFX_HIP(ctx, pfxk_recanvas(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, w, h, new_w, new_h, off_x, off_y, rgba));
    return PFX_OK;
}

int pfx_resize_canvas(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst, uint32_t new_w, uint32_t new_h, uint32_t anchor_x,
                      uint32_t anchor_y, const uint8_t fill[4])
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src && dst && pfx_dims_ok(w, h) && pfx_dims_ok(new_w, new_h), "pfx_resize_canvas: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, (size_t)w * h * 4));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, (size_t)new_w * new_h * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, (size_t)w * h * 4));
    PFX_TRY(pfx_resize_canvas_dev(ctx, ctx->st_in.p, w, h, ctx->st_out.p, new_w, new_h, anchor_x, anchor_y, fill));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, (size_t)new_w * new_h * 4));
    return pfx_sync(ctx);
}

int pfx_resize_image(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst, uint32_t new_w, uint32_t new_h, int filter)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src && dst && pfx_dims_ok(w, h) && pfx_dims_ok(new_w, new_h), "pfx_resize_image: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, (size_t)w * h * 4));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, (size_t)new_w * new_h * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, (size_t)w * h * 4));
    PFX_TRY(pfx_resize_image_dev(ctx, ctx->st_in.p, w, h, ctx->st_out.p, new_w, new_h, filter));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, (size_t)new_w * new_h * 4));
    return pfx_sync(ctx);
}

} // extern "C"
