/*
 * o_effects.c — oracle restatement of the effects that reuse the hot-path kernels.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ops/effects/stylize.rs:26-70    glow_core      (Gaussian + screen blend, `.round()`)
 *   src/ops/effects/stylize.rs:96-143   sharpen_core   (unsharp mask)
 *   src/ops/effects/blur.rs:22-115      bokeh_blur_core (equal-weight disc, integer sums, f32 scale)
 *   src/ops/effects/blur.rs:144-210     motion_blur_core (nearest samples along a direction, f32 sums)
 */
#include "o_common.h"

static inline uint8_t round_u8(float v) { return rs_f32_as_u8(rs_clampf(roundf(v), 0.0f, 255.0f)); }
static inline long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* stylize.rs:26-70 */
void pfxo_glow(const uint8_t* src, uint32_t w, uint32_t h, float radius, float intensity, const uint8_t* mask,
               uint8_t* dst, int threads)
{
    size_t n = (size_t)w * h;
    uint8_t* blur = (uint8_t*)malloc(n * 4);
    pfxo_gaussian_blur(src, w, h, radius, blur, threads);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const uint8_t* s = src + (size_t)i * 4;
        uint8_t* d = dst + (size_t)i * 4;
        if (mask && mask[i] == 0) { memcpy(d, s, 4); continue; }
        for (int c = 0; c < 3; ++c) {
            float sv = (float)s[c] / 255.0f;
            float b = (float)blur[(size_t)i * 4 + c] / 255.0f;
            float result = 1.0f - (1.0f - sv) * (1.0f - b * intensity);
            d[c] = round_u8(result * 255.0f);
        }
        d[3] = s[3];
    }
    free(blur);
}

/* stylize.rs:96-143 */
void pfxo_sharpen(const uint8_t* src, uint32_t w, uint32_t h, float amount, float radius, const uint8_t* mask,
                  uint8_t* dst, int threads)
{
    size_t n = (size_t)w * h;
    uint8_t* blur = (uint8_t*)malloc(n * 4);
    pfxo_gaussian_blur(src, w, h, radius, blur, threads);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const uint8_t* s = src + (size_t)i * 4;
        uint8_t* d = dst + (size_t)i * 4;
        if (mask && mask[i] == 0) { memcpy(d, s, 4); continue; }
        for (int c = 0; c < 3; ++c) {
            float sv = (float)s[c], b = (float)blur[(size_t)i * 4 + c];
            float v = sv + amount * (sv - b);
            d[c] = round_u8(v);
        }
        d[3] = s[3];
    }
    free(blur);
}

/* blur.rs:22-115; the sliding row sums of the reference equal direct clamped-span sums */
void pfxo_bokeh_blur(const uint8_t* src, uint32_t w32, uint32_t h32, float radius, const uint8_t* mask, uint8_t* dst,
                     int threads)
{
    long w = w32, h = h32;
    if (radius < 0.5f || w == 0 || h == 0) { memcpy(dst, src, (size_t)w * h * 4); return; }
    int32_t r = rs_f32_as_i32(ceilf(radius));
    float r2 = radius * radius;
    int32_t* span_dy = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * r + 1));
    int32_t* span_hw = (int32_t*)malloc(sizeof(int32_t) * (size_t)(2 * r + 1));
    int n_spans = 0;
    size_t sample_count = 0;
    for (int32_t dy = -r; dy <= r; ++dy) {
        float remaining = r2 - (float)(dy * dy);
        if (remaining >= 0.0f) {
            int32_t span = rs_f32_as_i32(floorf(sqrtf(remaining)));
            span_dy[n_spans] = dy; span_hw[n_spans] = span; ++n_spans;
            sample_count += (size_t)(span * 2 + 1);
        }
    }
    float inv_count = 1.0f / (float)sample_count;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (mask && mask[(size_t)y * w + x] == 0) { memcpy(dst + oi, src + oi, 4); continue; }
            uint64_t tot[4] = {0, 0, 0, 0};
            for (int k = 0; k < n_spans; ++k) {
                long sy = clampl(y + span_dy[k], 0, h - 1);
                for (long dx = -span_hw[k]; dx <= span_hw[k]; ++dx) {
                    long sx = clampl(x + dx, 0, w - 1);
                    const uint8_t* p = src + ((size_t)sy * w + sx) * 4;
                    tot[0] += p[0]; tot[1] += p[1]; tot[2] += p[2]; tot[3] += p[3];
                }
            }
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8((float)tot[c] * inv_count);
        }
    free(span_dy);
    free(span_hw);
}

/* blur.rs:144-210 */
void pfxo_motion_blur(const uint8_t* src, uint32_t w32, uint32_t h32, float angle_deg, float distance,
                      const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (distance < 1.0f || w == 0 || h == 0) { memcpy(dst, src, (size_t)w * h * 4); return; }
    float angle = angle_deg * (3.14159265358979323846f / 180.0f); /* f32::to_radians: self * (PI / 180.0) in f32 */
    int32_t steps = rs_f32_as_i32(ceilf(distance));
    float dx = cosf(angle), dy = sinf(angle);
    float inv_steps = 1.0f / (float)(steps * 2 + 1);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (mask && mask[(size_t)y * w + x] == 0) { memcpy(dst + oi, src + oi, 4); continue; }
            float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int32_t i = -steps; i <= steps; ++i) {
                long sx = clampl(rs_f32_as_i32(roundf((float)x + (float)i * dx)), 0, w - 1);
                long sy = clampl(rs_f32_as_i32(roundf((float)y + (float)i * dy)), 0, h - 1);
                const uint8_t* p = src + ((size_t)sy * w + sx) * 4;
                sum[0] += (float)p[0]; sum[1] += (float)p[1]; sum[2] += (float)p[2]; sum[3] += (float)p[3];
            }
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(sum[c] * inv_steps);
        }
}
