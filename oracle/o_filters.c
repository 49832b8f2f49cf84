/*
 * o_filters.c — oracle restatement of the stencil filters.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ops/filters.rs:214-234   build_gaussian_kernel
 *   src/ops/filters.rs:242-316   parallel_gaussian_blur
 *   src/ops/filters.rs:141-207   blur_with_selection
 *   src/ops/effects/blur.rs:233-318   box_blur_core
 *   src/ops/effects/noise.rs:357-410  median_core
 *   src/ops/effects/distort.rs:333-373 pixelate_core
 * Selection masks are w*h bytes with the image's own dimensions; 0 = leave pixel.
 */
#include "o_common.h"

static inline long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* filters.rs:214-234 */
int pfxo_gaussian_kernel(float sigma, float* out, int cap)
{
    uint32_t radius = rs_f32_as_u32(ceilf(sigma * 3.0f));
    if (radius == 0) {
        if (cap >= 1) out[0] = 1.0f;
        return 1;
    }
    int len = (int)radius * 2 + 1;
    if (len > cap) return -len;
    float s2 = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (int i = 0; i < len; ++i) {
        float x = (float)i - (float)radius;
        float v = expf(-x * x / s2);
        out[i] = v;
        sum += v;
    }
    float inv = 1.0f / sum;
    for (int i = 0; i < len; ++i) out[i] *= inv;
    return len;
}

/* filters.rs:242-316 */
void pfxo_gaussian_blur(const uint8_t* src, uint32_t w32, uint32_t h32, float sigma, uint8_t* dst, int threads)
{
    size_t w = w32, h = h32;
    if (w == 0 || h == 0) return;
    int cap = 2 * (int)rs_f32_as_u32(ceilf(sigma * 3.0f)) + 1;
    float* kernel = (float*)malloc(sizeof(float) * (size_t)cap);
    int klen = pfxo_gaussian_kernel(sigma, kernel, cap);
    long radius = klen / 2;
    float* buf_h = (float*)malloc(sizeof(float) * w * h * 4);
    o_set_threads(threads);

#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) { /* :258-283 horizontal, u8 -> f32 */
        const uint8_t* row_in = src + (size_t)y * w * 4;
        float* row_out = buf_h + (size_t)y * w * 4;
        for (long x = 0; x < (long)w; ++x) {
            float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
            for (int ki = 0; ki < klen; ++ki) {
                float kv = kernel[ki];
                long sx = clampl(x + ki - radius, 0, (long)w - 1);
                const uint8_t* p = row_in + sx * 4;
                r += (float)p[0] * kv;
                g += (float)p[1] * kv;
                b += (float)p[2] * kv;
                a += (float)p[3] * kv;
            }
            row_out[x * 4 + 0] = r; row_out[x * 4 + 1] = g; row_out[x * 4 + 2] = b; row_out[x * 4 + 3] = a;
        }
    }
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) { /* :286-313 vertical, f32 -> u8 */
        uint8_t* row_out = dst + (size_t)y * w * 4;
        for (long x = 0; x < (long)w; ++x) {
            float r = 0.0f, g = 0.0f, b = 0.0f, a = 0.0f;
            for (int ki = 0; ki < klen; ++ki) {
                float kv = kernel[ki];
                long sy = clampl(y + ki - radius, 0, (long)h - 1);
                const float* p = buf_h + ((size_t)sy * w + (size_t)x) * 4;
                r += p[0] * kv;
                g += p[1] * kv;
                b += p[2] * kv;
                a += p[3] * kv;
            }
            row_out[x * 4 + 0] = rs_f32_as_u8(rs_clampf(roundf(r), 0.0f, 255.0f));
            row_out[x * 4 + 1] = rs_f32_as_u8(rs_clampf(roundf(g), 0.0f, 255.0f));
            row_out[x * 4 + 2] = rs_f32_as_u8(rs_clampf(roundf(b), 0.0f, 255.0f));
            row_out[x * 4 + 3] = rs_f32_as_u8(rs_clampf(roundf(a), 0.0f, 255.0f));
        }
    }
    free(buf_h);
    free(kernel);
}

/* filters.rs:141-207 */
void pfxo_blur_with_selection(const uint8_t* src, uint32_t w, uint32_t h, float sigma,
                              const uint8_t* mask, uint8_t* dst, int threads)
{
    if (!mask) { pfxo_gaussian_blur(src, w, h, sigma, dst, threads); return; }
    uint32_t min_x = w, min_y = h, max_x = 0, max_y = 0;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x)
            if (mask[(size_t)y * w + x] > 0) {
                if (x < min_x) min_x = x;
                if (y < min_y) min_y = y;
                if (x > max_x) max_x = x;
                if (y > max_y) max_y = y;
            }
    memcpy(dst, src, (size_t)w * h * 4);
    if (min_x > max_x || min_y > max_y) return; /* nothing selected :163 */
    uint32_t pad = rs_f32_as_u32(ceilf(sigma * 3.0f));
    uint32_t crop_x = min_x > pad ? min_x - pad : 0, crop_y = min_y > pad ? min_y - pad : 0;
    uint32_t crop_x2 = max_x + 1 + pad < w ? max_x + 1 + pad : w;
    uint32_t crop_y2 = max_y + 1 + pad < h ? max_y + 1 + pad : h;
    uint32_t cw = crop_x2 - crop_x, ch = crop_y2 - crop_y;
    uint8_t* sub = (uint8_t*)malloc((size_t)cw * ch * 4);
    uint8_t* bl = (uint8_t*)malloc((size_t)cw * ch * 4);
    for (uint32_t y = 0; y < ch; ++y)
        memcpy(sub + (size_t)y * cw * 4, src + ((size_t)(crop_y + y) * w + crop_x) * 4, (size_t)cw * 4);
    pfxo_gaussian_blur(sub, cw, ch, sigma, bl, threads);
    for (uint32_t y = min_y; y <= max_y; ++y)
        for (uint32_t x = min_x; x <= max_x; ++x)
            if (mask[(size_t)y * w + x] > 0)
                memcpy(dst + ((size_t)y * w + x) * 4, bl + ((size_t)(y - crop_y) * cw + (x - crop_x)) * 4, 4);
    free(sub);
    free(bl);
}

/* effects/blur.rs:233-318.  The sliding sums of the reference equal the direct clamped-window sums. */
void pfxo_box_blur(const uint8_t* src, uint32_t w32, uint32_t h32, float radius, const uint8_t* mask,
                   uint8_t* dst, int threads)
{
    size_t w = w32, h = h32;
    if (radius < 0.5f || w == 0 || h == 0) { memcpy(dst, src, w * h * 4); return; }
    long r = (long)rs_f32_as_u32(ceilf(radius));
    uint32_t divisor = (uint32_t)(r * 2 + 1);
    uint8_t* h_buf = (uint8_t*)malloc(w * h * 4);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y) {
        const uint8_t* row = src + (size_t)y * w * 4;
        uint8_t* out = h_buf + (size_t)y * w * 4;
        uint32_t sums[4] = {0, 0, 0, 0};
        for (long k = -r; k <= r; ++k) {
            long sx = clampl(k, 0, (long)w - 1);
            for (int c = 0; c < 4; ++c) sums[c] += row[sx * 4 + c];
        }
        for (long x = 0; x < (long)w; ++x) {
            for (int c = 0; c < 4; ++c) out[x * 4 + c] = (uint8_t)((sums[c] + divisor / 2) / divisor);
            if (x + 1 < (long)w) {
                long rx = clampl(x - r, 0, (long)w - 1), ax = clampl(x + r + 1, 0, (long)w - 1);
                for (int c = 0; c < 4; ++c) sums[c] = sums[c] - row[rx * 4 + c] + row[ax * 4 + c];
            }
        }
    }
    /* vertical pass: serial `for x` in the reference (:284); columns are independent so it is parallel here */
#pragma omp parallel for schedule(static)
    for (long x = 0; x < (long)w; ++x) {
        uint32_t sums[4] = {0, 0, 0, 0};
        for (long k = -r; k <= r; ++k) {
            long sy = clampl(k, 0, (long)h - 1);
            for (int c = 0; c < 4; ++c) sums[c] += h_buf[((size_t)sy * w + x) * 4 + c];
        }
        for (long y = 0; y < (long)h; ++y) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (mask && mask[(size_t)y * w + x] == 0) memcpy(dst + oi, src + oi, 4);
            else for (int c = 0; c < 4; ++c) dst[oi + c] = (uint8_t)((sums[c] + divisor / 2) / divisor);
            if (y + 1 < (long)h) {
                long ry = clampl(y - r, 0, (long)h - 1), ay = clampl(y + r + 1, 0, (long)h - 1);
                for (int c = 0; c < 4; ++c)
                    sums[c] = sums[c] - h_buf[((size_t)ry * w + x) * 4 + c] + h_buf[((size_t)ay * w + x) * 4 + c];
            }
        }
    }
    free(h_buf);
}

/* effects/noise.rs:357-410.  sort + pick len/2 == rank selection; done with a 256-bin histogram. */
void pfxo_median(const uint8_t* src, uint32_t w32, uint32_t h32, uint32_t radius, const uint8_t* mask,
                 uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    long r = radius > 1 ? (long)radius : 1;
    long n = (2 * r + 1) * (2 * r + 1);
    long rank = n / 2;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y) {
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (mask && mask[(size_t)y * w + x] == 0) { memcpy(dst + oi, src + oi, 4); continue; }
            uint32_t hist[4][256];
            memset(hist, 0, sizeof hist);
            for (long dy = -r; dy <= r; ++dy) {
                long sy = clampl(y + dy, 0, h - 1);
                for (long dx = -r; dx <= r; ++dx) {
                    long sx = clampl(x + dx, 0, w - 1);
                    const uint8_t* p = src + ((size_t)sy * w + sx) * 4;
                    hist[0][p[0]]++; hist[1][p[1]]++; hist[2][p[2]]++; hist[3][p[3]]++;
                }
            }
            for (int c = 0; c < 4; ++c) {
                long acc = 0;
                int v = 0;
                for (; v < 256; ++v) { acc += hist[c][v]; if (acc > rank) break; }
                dst[oi + c] = (uint8_t)v;
            }
        }
    }
}

/* effects/distort.rs:333-373 */
void pfxo_pixelate(const uint8_t* src, uint32_t w, uint32_t h, uint32_t block, const uint8_t* mask,
                   uint8_t* dst, int threads)
{
    if (w == 0 || h == 0) return;
    uint32_t bs = block > 2 ? block : 2;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (mask && mask[(size_t)y * w + x] == 0) { memcpy(dst + oi, src + oi, 4); continue; }
            uint32_t bx = (x / bs) * bs + bs / 2, by = ((uint32_t)y / bs) * bs + bs / 2;
            uint32_t sx = bx < w - 1 ? bx : w - 1, sy = by < h - 1 ? by : h - 1;
            memcpy(dst + oi, src + ((size_t)sy * w + sx) * 4, 4);
        }
}
