/*
 * o_effects2.c — oracle restatement of the rest of the effect bank (the Rhai Effect API's
 * apply_noise/reduce_noise/crystallize/bulge/twist/vignette/halftone/ink/oil_painting and the dialog-only effects).
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ops/effects.rs:53-98,108-161          apply_per_pixel, sample_clamped, sample_bilinear, hash_u32, hash_f32
 *   src/ops/effects/blur.rs:322-427           zoom_blur_core
 *   src/ops/effects/distort.rs:26-169         crystallize_core
 *   src/ops/effects/distort.rs:229-310        turbulence_2d, dents_core
 *   src/ops/effects/distort.rs:400-437        bulge_core_at
 *   src/ops/effects/distort.rs:464-493        twist_core_at
 *   src/ops/effects/noise.rs:53-143           perlin_noise_2d, add_noise_core
 *   src/ops/effects/noise.rs:172-261          reduce_noise_core
 *   src/ops/effects/stylize.rs:170-191        vignette_core
 *   src/ops/effects/stylize.rs:242-277        halftone_core
 *   src/ops/effects/render.rs:52-92           grid_core
 *   src/ops/effects/render.rs:114-165         canvas_border_core
 *   src/ops/effects/render.rs:220-349         shadow_core
 *   src/ops/effects/render.rs:403-572         outline_core
 *   src/ops/effects/glitch.rs:44-99           pixel_drag_core
 *   src/ops/effects/glitch.rs:142-196         rgb_displace_core
 *   src/ops/effects/artistic.rs:31-99         ink_core
 *   src/ops/effects/artistic.rs:123-215       oil_painting_core
 *   src/ops/effects/artistic.rs:266-309       color_filter_core
 *   src/ops/effects/contours.rs:56-112        contours_core
 * Transcendentals are glibc's (Rust's f32::exp/ln/cos/sin/powf lower to the system libm on Linux).
 */
#include "o_common.h"

#define PI_F 3.14159265358979323846f
static inline uint8_t round_u8(float v) { return rs_f32_as_u8(rs_clampf(roundf(v), 0.0f, 255.0f)); }
static inline long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float to_radians(float deg) { return deg * (PI_F / 180.0f); }
static inline int masked_out(const uint8_t* mask, long w, long x, long y) { return mask && mask[(size_t)y * w + x] == 0; }

/* effects.rs:143-161 */
static inline uint32_t hash_u32(uint32_t x)
{
    x *= 0x9E3779B9u; x ^= x >> 16;
    x *= 0x85EBCA6Bu; x ^= x >> 13;
    x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
static inline float hash_f32(uint32_t x, uint32_t y, uint32_t seed)
{
    uint32_t h = hash_u32(x * 374761393u + y * 668265263u + seed);
    return (float)(h & 0x00FFFFFFu) / 16777216.0f;
}
float pfxo_hash_f32(uint32_t x, uint32_t y, uint32_t seed) { return hash_f32(x, y, seed); }

/* noise.rs:53-71 */
static float perlin_noise_2d(float x, float y, uint32_t seed)
{
    int32_t xi = rs_f32_as_i32(floorf(x)), yi = rs_f32_as_i32(floorf(y));
    float xf = x - (float)xi, yf = y - (float)yi;
    float u = xf * xf * xf * (xf * (xf * 6.0f - 15.0f) + 10.0f);
    float v = yf * yf * yf * (yf * (yf * 6.0f - 15.0f) + 10.0f);
    float n00 = hash_f32((uint32_t)xi, (uint32_t)yi, seed);
    float n10 = hash_f32((uint32_t)(xi + 1), (uint32_t)yi, seed);
    float n01 = hash_f32((uint32_t)xi, (uint32_t)(yi + 1), seed);
    float n11 = hash_f32((uint32_t)(xi + 1), (uint32_t)(yi + 1), seed);
    float nx0 = n00 + u * (n10 - n00);
    float nx1 = n01 + u * (n11 - n01);
    return nx0 + v * (nx1 - nx0);
}

/* distort.rs:229-246 */
static float turbulence_2d(float x, float y, uint32_t seed, uint32_t octaves, float roughness)
{
    float total = 0.0f, amplitude = 1.0f, frequency = 1.0f, max_amplitude = 0.0f;
    for (uint32_t i = 0; i < octaves; ++i) {
        uint32_t s = seed + i * 1000u;
        total += perlin_noise_2d(x * frequency, y * frequency, s) * amplitude;
        max_amplitude += amplitude;
        amplitude *= roughness;
        frequency *= 2.0f;
    }
    return max_amplitude > 0.0f ? total / max_amplitude : 0.0f;
}
float pfxo_turbulence_2d(float x, float y, uint32_t seed, uint32_t octaves, float roughness)
{
    return turbulence_2d(x, y, seed, octaves, roughness);
}

/* effects.rs:108-140 */
static inline void sample_clamped(const uint8_t* img, long w, long h, int32_t x, int32_t y, float out[4])
{
    long cx = clampl(x, 0, w - 1), cy = clampl(y, 0, h - 1);
    const uint8_t* p = img + ((size_t)cy * w + cx) * 4;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
}
static inline void sample_bilinear(const uint8_t* img, long w, long h, float fx, float fy, float out[4])
{
    int32_t x0 = rs_f32_as_i32(floorf(fx)), y0 = rs_f32_as_i32(floorf(fy));
    /* `x0 + 1` on i32::MAX would panic in debug / wrap in release; coordinates never get there */
    int32_t x1 = (int32_t)((uint32_t)x0 + 1u), y1 = (int32_t)((uint32_t)y0 + 1u);
    float dx = fx - (float)x0, dy = fy - (float)y0;
    float p00[4], p10[4], p01[4], p11[4];
    sample_clamped(img, w, h, x0, y0, p00);
    sample_clamped(img, w, h, x1, y0, p10);
    sample_clamped(img, w, h, x0, y1, p01);
    sample_clamped(img, w, h, x1, y1, p11);
    for (int c = 0; c < 4; ++c)
        out[c] = p00[c] * (1.0f - dx) * (1.0f - dy) + p10[c] * dx * (1.0f - dy) + p01[c] * (1.0f - dx) * dy + p11[c] * dx * dy;
}

/* rust f32::rem_euclid */
static inline float rem_euclid(float a, float b)
{
    float r = fmodf(a, b);
    return r < 0.0f ? r + fabsf(b) : r;
}

/* blur.rs:322-427 */
void pfxo_zoom_blur(const uint8_t* src, uint32_t w32, uint32_t h32, float center_x, float center_y, float strength,
                    uint32_t samples, const float tint_color[4], float tint_strength, const uint8_t* mask, uint8_t* dst,
                    int threads)
{
    long w = w32, h = h32;
    if (strength < 0.001f || w == 0 || h == 0) { memcpy(dst, src, (size_t)w * h * 4); return; }
    float cx = center_x * (float)w, cy = center_y * (float)h;
    float s = rs_clampf(strength, 0.0f, 0.99f);
    long n = samples < 2 ? 2 : samples;
    float inv_n = 1.0f / (float)n;
    float corners[4][2] = {{cx, cy}, {(float)w - cx, cy}, {cx, (float)h - cy}, {(float)w - cx, (float)h - cy}};
    float max_dist = 0.0f;
    for (int k = 0; k < 4; ++k) max_dist = fmaxf(max_dist, sqrtf(corners[k][0] * corners[k][0] + corners[k][1] * corners[k][1]));
    max_dist = fmaxf(max_dist, 1.0f);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float dx = (float)x - cx, dy = (float)y - cy;
            float sum[4] = {0, 0, 0, 0};
            for (long i = 0; i < n; ++i) {
                float t = 1.0f - s * ((float)i / (float)(n - 1));
                long sx = clampl(rs_f32_as_i32(roundf(cx + dx * t)), 0, w - 1);
                long sy = clampl(rs_f32_as_i32(roundf(cy + dy * t)), 0, h - 1);
                const uint8_t* p = src + ((size_t)sy * w + sx) * 4;
                for (int c = 0; c < 4; ++c) sum[c] += (float)p[c];
            }
            float v[4];
            for (int c = 0; c < 4; ++c) v[c] = sum[c] * inv_n;
            if (tint_strength > 0.001f) {
                float dist = sqrtf(dx * dx + dy * dy);
                float t = fmaxf(1.0f - dist / max_dist, 0.0f) * tint_strength;
                for (int c = 0; c < 4; ++c) v[c] = v[c] + (tint_color[c] * 255.0f - v[c]) * t;
            }
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(v[c]);
        }
}

/* distort.rs:26-169 */
static inline size_t nearest_seed(const float* seeds, int32_t cells_x, int32_t cells_y, float cs, long x, long y)
{
    int32_t gcx = rs_f32_as_i32((float)x / cs), gcy = rs_f32_as_i32((float)y / cs);
    float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float best_dist = 3.40282347e+38f;
    size_t best_idx = 0;
    for (int32_t dy = -1; dy <= 1; ++dy)
        for (int32_t dx = -1; dx <= 1; ++dx) {
            int32_t nx = gcx + dx, ny = gcy + dy;
            if (nx < 0 || ny < 0 || nx >= cells_x || ny >= cells_y) continue;
            size_t idx = (size_t)(ny * cells_x + nx);
            float sx = seeds[idx * 2], sy = seeds[idx * 2 + 1];
            float d = (px - sx) * (px - sx) + (py - sy) * (py - sy);
            if (d < best_dist) { best_dist = d; best_idx = idx; }
        }
    return best_idx;
}
void pfxo_crystallize(const uint8_t* src, uint32_t w32, uint32_t h32, float cell_size, uint32_t seed, const uint8_t* mask,
                      uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    float cs = fmaxf(cell_size, 2.0f);
    memcpy(dst, src, (size_t)w * h * 4);
    if (w == 0 || h == 0) return;
    int32_t cells_x = rs_f32_as_i32(ceilf((float)w / cs)), cells_y = rs_f32_as_i32(ceilf((float)h / cs));
    if (cells_x < 1) cells_x = 1;
    if (cells_y < 1) cells_y = 1;
    size_t num_cells = (size_t)cells_x * cells_y;
    float* seeds = (float*)malloc(sizeof(float) * 2 * num_cells);
    for (int32_t cy = 0; cy < cells_y; ++cy)
        for (int32_t cx = 0; cx < cells_x; ++cx) {
            float base_x = (float)cx * cs, base_y = (float)cy * cs;
            float jx = hash_f32((uint32_t)cx, (uint32_t)cy, seed), jy = hash_f32((uint32_t)cx, (uint32_t)cy, seed + 77u);
            size_t i = (size_t)cy * cells_x + cx;
            seeds[i * 2] = base_x + jx * cs;
            seeds[i * 2 + 1] = base_y + jy * cs;
        }
    double* sums = (double*)calloc(num_cells * 4, sizeof(double));
    uint32_t* counts = (uint32_t*)calloc(num_cells, sizeof(uint32_t));
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t b = nearest_seed(seeds, cells_x, cells_y, cs, x, y);
            const uint8_t* p = src + ((size_t)y * w + x) * 4;
            for (int c = 0; c < 4; ++c) sums[b * 4 + c] += (double)p[c];
            counts[b] += 1;
        }
    uint8_t* avg = (uint8_t*)calloc(num_cells, 4);
    for (size_t i = 0; i < num_cells; ++i)
        if (counts[i] > 0) {
            double inv = 1.0 / (double)counts[i];
            for (int c = 0; c < 4; ++c) {
                double v = round(sums[i * 4 + c] * inv);
                v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
                avg[i * 4 + c] = (uint8_t)v;
            }
        }
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            if (masked_out(mask, w, x, y)) continue;
            size_t b = nearest_seed(seeds, cells_x, cells_y, cs, x, y);
            memcpy(dst + ((size_t)y * w + x) * 4, avg + b * 4, 4);
        }
    free(seeds); free(sums); free(counts); free(avg);
}

/* distort.rs:248-310 */
void pfxo_dents(const uint8_t* src, uint32_t w32, uint32_t h32, float scale, float amount, uint32_t seed, uint32_t octaves,
                float roughness, int pinch, int wrap, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    uint32_t oct = octaves < 1 ? 1 : (octaves > 8 ? 8 : octaves);
    float inv_scale = 1.0f / fmaxf(scale, 0.5f);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float nx = turbulence_2d((float)x * inv_scale, (float)y * inv_scale, seed, oct, roughness) * 2.0f - 1.0f;
            float ny = turbulence_2d((float)x * inv_scale, (float)y * inv_scale, seed + 9999u, oct, roughness) * 2.0f - 1.0f;
            if (pinch) {
                float cx = (float)w * 0.5f, cy = (float)h * 0.5f;
                float dx = (float)x - cx, dy = (float)y - cy;
                float dist = fmaxf(sqrtf(dx * dx + dy * dy), 1.0f);
                float factor = (1.0f - dist / fmaxf(cx, cy)) * 0.5f;
                nx = nx + dx / dist * factor;
                ny = ny + dy / dist * factor;
            }
            float src_x = (float)x + nx * amount * scale;
            float src_y = (float)y + ny * amount * scale;
            if (wrap) { src_x = rem_euclid(src_x, (float)w); src_y = rem_euclid(src_y, (float)h); }
            float p[4];
            sample_bilinear(src, w, h, src_x, src_y, p);
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(p[c]);
        }
}

/* distort.rs:400-437 */
void pfxo_bulge(const uint8_t* src, uint32_t w32, uint32_t h32, float amount, float origin_x, float origin_y,
                const uint8_t* mask, uint8_t* dst, int threads)
{
    long wl = w32, hl = h32;
    if (wl == 0 || hl == 0) return;
    float w = (float)w32, h = (float)h32;
    float cx = rs_clampf(origin_x, 0.0f, 1.0f) * fmaxf(w - 1.0f, 0.0f);
    float cy = rs_clampf(origin_y, 0.0f, 1.0f) * fmaxf(h - 1.0f, 0.0f);
    float max_r = fmaxf(fmaxf(fmaxf(cx, w - cx), fmaxf(cy, h - cy)), 1.0f);
    float strength = fmaxf(fabsf(amount), 0.0001f);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < hl; ++y)
        for (long x = 0; x < wl; ++x) {
            size_t oi = ((size_t)y * wl + x) * 4;
            if (masked_out(mask, wl, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float dx = (float)x - cx, dy = (float)y - cy;
            float dist = sqrtf(dx * dx + dy * dy);
            float norm = fminf(dist / max_r, 1.0f);
            if (norm >= 1.0f) { memcpy(dst + oi, src + oi, 4); continue; } /* sample_clamped(x, y) -> round() is the identity */
            float falloff = 1.0f - norm;
            float factor = amount > 0.0f ? 1.0f - falloff * strength * 0.5f : (amount < 0.0f ? 1.0f + falloff * strength * 0.5f : 1.0f);
            float p[4];
            sample_bilinear(src, wl, hl, cx + dx * factor, cy + dy * factor, p);
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(p[c]);
        }
}

/* distort.rs:464-493 */
void pfxo_twist(const uint8_t* src, uint32_t w32, uint32_t h32, float angle_deg, float origin_x, float origin_y,
                const uint8_t* mask, uint8_t* dst, int threads)
{
    long wl = w32, hl = h32;
    if (wl == 0 || hl == 0) return;
    float w = (float)w32, h = (float)h32;
    float cx = rs_clampf(origin_x, 0.0f, 1.0f) * fmaxf(w - 1.0f, 0.0f);
    float cy = rs_clampf(origin_y, 0.0f, 1.0f) * fmaxf(h - 1.0f, 0.0f);
    float mx = fmaxf(cx, w - cx), my = fmaxf(cy, h - cy);
    float max_r = fmaxf(sqrtf(mx * mx + my * my), 1.0f);
    float twist_amount = to_radians(angle_deg);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < hl; ++y)
        for (long x = 0; x < wl; ++x) {
            size_t oi = ((size_t)y * wl + x) * 4;
            if (masked_out(mask, wl, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float dx = (float)x - cx, dy = (float)y - cy;
            float dist = sqrtf(dx * dx + dy * dy);
            float norm = dist / max_r;
            float rotation = twist_amount * (1.0f - norm);
            float cos_r = cosf(rotation), sin_r = sinf(rotation);
            float p[4];
            sample_bilinear(src, wl, hl, cx + dx * cos_r - dy * sin_r, cy + dx * sin_r + dy * cos_r, p);
            for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(p[c]);
        }
}

/* noise.rs:73-143; noise_type: 0 uniform, 1 gaussian, 2 perlin */
void pfxo_add_noise(const uint8_t* src, uint32_t w32, uint32_t h32, float amount, int noise_type, int monochrome,
                    uint32_t seed, float scale, uint32_t octaves, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    float inv_scale = 1.0f / fmaxf(scale, 0.1f);
    uint32_t oct = octaves < 1 ? 1 : (octaves > 8 ? 8 : octaves);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float r = src[oi], g = src[oi + 1], b = src[oi + 2], a = src[oi + 3];
            float sx = (float)x * inv_scale, sy = (float)y * inv_scale;
            uint32_t qx = rs_f32_as_u32(floorf((float)x * inv_scale)), qy = rs_f32_as_u32(floorf((float)y * inv_scale));
            float strength = amount * 255.0f / 100.0f;
            float nr, ng, nb;
            if (monochrome) {
                float noise_val;
                if (noise_type == 0) noise_val = hash_f32(qx, qy, seed) * 2.0f - 1.0f;
                else if (noise_type == 1) {
                    float u1 = fmaxf(hash_f32(qx, qy, seed), 0.0001f);
                    float u2 = hash_f32(qx, qy, seed + 7u);
                    noise_val = sqrtf(-2.0f * logf(u1)) * cosf(2.0f * PI_F * u2) * 0.33f;
                } else noise_val = turbulence_2d(sx, sy, seed, oct, 0.5f) * 2.0f - 1.0f;
                nr = ng = nb = noise_val * strength;
            } else if (noise_type == 2) {
                nr = (turbulence_2d(sx, sy, seed, oct, 0.5f) * 2.0f - 1.0f) * strength;
                ng = (turbulence_2d(sx, sy, seed + 1u, oct, 0.5f) * 2.0f - 1.0f) * strength;
                nb = (turbulence_2d(sx, sy, seed + 2u, oct, 0.5f) * 2.0f - 1.0f) * strength;
            } else { /* uniform AND gaussian: the colour branch draws uniform values (noise.rs:114-138) */
                nr = (hash_f32(qx, qy, seed) * 2.0f - 1.0f) * strength;
                ng = (hash_f32(qx, qy, seed + 1u) * 2.0f - 1.0f) * strength;
                nb = (hash_f32(qx, qy, seed + 2u) * 2.0f - 1.0f) * strength;
            }
            dst[oi] = round_u8(r + nr); dst[oi + 1] = round_u8(g + ng); dst[oi + 2] = round_u8(b + nb); dst[oi + 3] = round_u8(a);
        }
}

/* noise.rs:172-261 */
void pfxo_reduce_noise(const uint8_t* src, uint32_t w32, uint32_t h32, float strength, uint32_t radius, const uint8_t* mask,
                       uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    int32_t r = radius < 1 ? 1 : (int32_t)radius;
    float sigma_s = (float)r, sigma_r = strength * 2.55f;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float cr = src[oi], cg = src[oi + 1], cb = src[oi + 2];
            float sum[4] = {0, 0, 0, 0}, weight_sum = 0.0f;
            for (int32_t dy = -r; dy <= r; ++dy) {
                long sy = clampl(y + dy, 0, h - 1);
                for (int32_t dx = -r; dx <= r; ++dx) {
                    long sx = clampl(x + dx, 0, w - 1);
                    const uint8_t* p = src + ((size_t)sy * w + sx) * 4;
                    float pr = p[0], pg = p[1], pb = p[2], pa = p[3];
                    float spatial = (float)(dx * dx + dy * dy) / (2.0f * sigma_s * sigma_s);
                    float dr = cr - pr, dg = cg - pg, db = cb - pb;
                    float range = (dr * dr + dg * dg + db * db) / (2.0f * sigma_r * sigma_r + 0.001f);
                    float weight = expf(-spatial - range);
                    sum[0] += pr * weight; sum[1] += pg * weight; sum[2] += pb * weight; sum[3] += pa * weight;
                    weight_sum += weight;
                }
            }
            if (weight_sum > 0.0f) {
                float inv = 1.0f / weight_sum;
                for (int c = 0; c < 4; ++c) dst[oi + c] = round_u8(sum[c] * inv);
            } else memcpy(dst + oi, src + oi, 4);
        }
}

/* stylize.rs:170-191 */
void pfxo_vignette(const uint8_t* src, uint32_t w32, uint32_t h32, float amount, float softness, const uint8_t* mask,
                   uint8_t* dst, int threads)
{
    long wl = w32, hl = h32;
    if (wl == 0 || hl == 0) return;
    float cx = (float)w32 / 2.0f, cy = (float)h32 / 2.0f;
    float max_dist = sqrtf(cx * cx + cy * cy);
    float soft = fmaxf(softness, 0.01f);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < hl; ++y)
        for (long x = 0; x < wl; ++x) {
            size_t oi = ((size_t)y * wl + x) * 4;
            if (masked_out(mask, wl, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float dx = (float)x - cx, dy = (float)y - cy;
            float dist = sqrtf(dx * dx + dy * dy) / max_dist;
            float vf = rs_clampf(1.0f - (amount * powf(fminf(dist / soft, 1.0f), 2.0f)), 0.0f, 1.0f);
            for (int c = 0; c < 3; ++c) dst[oi + c] = round_u8((float)src[oi + c] * vf);
            dst[oi + 3] = src[oi + 3];
        }
}

/* stylize.rs:242-277; shape: 0 circle, 1 square, 2 diamond, 3 line */
void pfxo_halftone(const uint8_t* src, uint32_t w32, uint32_t h32, float dot_size, float angle_deg, int shape,
                   const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    float ds = fmaxf(dot_size, 2.0f);
    float angle = to_radians(angle_deg);
    float cos_a = cosf(angle), sin_a = sinf(angle);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float r = src[oi], g = src[oi + 1], b = src[oi + 2];
            float lum = (0.2126f * r + 0.7152f * g + 0.0722f * b) / 255.0f;
            float fx = (float)x * cos_a + (float)y * sin_a;
            float fy = -((float)x) * sin_a + (float)y * cos_a;
            float qx = fx / ds, qy = fy / ds;
            float cell_x = fabsf(qx - truncf(qx)), cell_y = fabsf(qy - truncf(qy));
            float cx = cell_x - 0.5f, cy = cell_y - 0.5f;
            float threshold;
            if (shape == 0) threshold = sqrtf(cx * cx + cy * cy) * 2.0f;
            else if (shape == 1) threshold = fmaxf(fabsf(cx), fabsf(cy)) * 2.0f;
            else if (shape == 2) threshold = fabsf(cx) + fabsf(cy);
            else threshold = fabsf(cy) * 2.0f;
            uint8_t val = threshold < lum ? 255 : 0;
            dst[oi] = dst[oi + 1] = dst[oi + 2] = val;
            dst[oi + 3] = src[oi + 3];
        }
}

/* render.rs:52-92; style: 0 lines, 1 checkerboard */
void pfxo_grid(const uint8_t* src, uint32_t w32, uint32_t h32, uint32_t cell_w, uint32_t cell_h, uint32_t line_width,
               const uint8_t color[4], int style, float opacity, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    uint32_t cw = cell_w < 2 ? 2 : cell_w, ch = cell_h < 2 ? 2 : cell_h, lw = line_width < 1 ? 1 : line_width;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            int draw = style == 0 ? (((uint32_t)x % cw) < lw || ((uint32_t)y % ch) < lw)
                                  : ((((uint32_t)x / cw) + ((uint32_t)y / ch)) % 2u == 0);
            for (int c = 0; c < 4; ++c) {
                float v = src[oi + c];
                if (draw) v = v * (1.0f - opacity) + (float)color[c] * opacity;
                dst[oi + c] = round_u8(v);
            }
        }
}

/* render.rs:114-165 */
void pfxo_canvas_border(const uint8_t* src, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4],
                        const uint8_t* mask, uint8_t* dst, int threads)
{
    (void)threads;
    memcpy(dst, src, (size_t)w * h * 4);
    if (w == 0 || h == 0) return;
    uint32_t border_w = width < 1 ? 1 : width;
    uint32_t m = w < h ? w : h;
    if (border_w > m) border_w = m;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            if (masked_out(mask, w, x, y)) continue;
            if (x < border_w || y < border_w || x >= w - border_w || y >= h - border_w) memcpy(dst + ((size_t)y * w + x) * 4, color, 4);
        }
}

/* render.rs:220-349 */
void pfxo_drop_shadow(const uint8_t* src, uint32_t w32, uint32_t h32, int32_t offset_x, int32_t offset_y, float blur_radius,
                      int widen_radius, const uint8_t color[4], float opacity, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    size_t n = (size_t)w * h;
    uint8_t* sa = (uint8_t*)calloc(n, 1);
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            long sx = x - offset_x, sy = y - offset_y;
            if (sx >= 0 && sx < w && sy >= 0 && sy < h) sa[(size_t)y * w + x] = src[((size_t)sy * w + sx) * 4 + 3];
        }
    if (widen_radius) {
        int32_t spread = rs_f32_as_i32(roundf(fmaxf(blur_radius, 1.0f)));
        if (spread > 0) {
            long r = spread;
            uint8_t* hz = (uint8_t*)calloc(n, 1);
            for (long y = 0; y < h; ++y)
                for (long x = 0; x < w; ++x) {
                    long x0 = x - r < 0 ? 0 : x - r, x1 = x + r > w - 1 ? w - 1 : x + r;
                    uint8_t m = 0;
                    for (long s = x0; s <= x1; ++s) if (sa[(size_t)y * w + s] > m) m = sa[(size_t)y * w + s];
                    hz[(size_t)y * w + x] = m;
                }
            for (long y = 0; y < h; ++y) {
                long y0 = y - r < 0 ? 0 : y - r, y1 = y + r > h - 1 ? h - 1 : y + r;
                for (long x = 0; x < w; ++x) {
                    uint8_t m = 0;
                    for (long s = y0; s <= y1; ++s) if (hz[(size_t)s * w + x] > m) m = hz[(size_t)s * w + x];
                    sa[(size_t)y * w + x] = m;
                }
            }
            free(hz);
        }
    }
    uint8_t* argba = (uint8_t*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) memset(argba + i * 4, sa[i], 4);
    uint8_t* blur = argba;
    if (blur_radius > 0.5f) {
        blur = (uint8_t*)malloc(n * 4);
        pfxo_gaussian_blur(argba, w32, h32, blur_radius, blur, threads);
    }
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        size_t si = (size_t)i * 4;
        if (mask && mask[i] == 0) { memcpy(dst + si, src + si, 4); continue; }
        float shadow_a = ((float)blur[si] / 255.0f) * opacity * ((float)color[3] / 255.0f);
        float src_a = (float)src[si + 3] / 255.0f;
        float out_a = src_a + shadow_a * (1.0f - src_a);
        for (int c = 0; c < 3; ++c) {
            float shadow_c = (float)color[c] / 255.0f, src_c = (float)src[si + c] / 255.0f;
            float out_c = out_a > 0.0f ? (src_c * src_a + shadow_c * shadow_a * (1.0f - src_a)) / out_a : 0.0f;
            dst[si + c] = round_u8(out_c * 255.0f);
        }
        dst[si + 3] = round_u8(out_a * 255.0f);
    }
    if (blur != argba) free(blur);
    free(argba); free(sa);
}

/* render.rs:403-572; mode: 0 outside, 1 inside, 2 center */
static int nearest_distance(const uint8_t* src, long w, long h, long x, long y, int32_t sr, int want_filled, float* out)
{
    int32_t best_sq = -1;
    for (int32_t dy = -sr; dy <= sr; ++dy)
        for (int32_t dx = -sr; dx <= sr; ++dx) {
            int32_t d = dx * dx + dy * dy;
            if (best_sq >= 0 && d > best_sq) continue;
            long sx = x + dx, sy = y + dy;
            if (sx < 0 || sy < 0 || sx >= w || sy >= h) continue;
            uint8_t a = src[((size_t)sy * w + sx) * 4 + 3];
            if (want_filled ? a > 0 : a == 0) best_sq = d;
        }
    if (best_sq < 0) return 0;
    *out = sqrtf((float)best_sq);
    return 1;
}
static inline float shell_coverage(float distance, float radius, int anti_alias)
{
    if (anti_alias) {
        float t = rs_clampf((radius + 0.5f - distance) / 1.0f, 0.0f, 1.0f);
        return t * t * (3.0f - 2.0f * t);
    }
    return distance <= radius ? 1.0f : 0.0f;
}
void pfxo_outline(const uint8_t* src, uint32_t w32, uint32_t h32, uint32_t width, const uint8_t color[4], int mode,
                  int anti_alias, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    memcpy(dst, src, (size_t)w * h * 4);
    if (w == 0 || h == 0) return;
    float radius = (float)(width < 1 ? 1 : width);
    int32_t sr = rs_f32_as_i32(ceilf(radius)) + 1;
    long min_x = w, min_y = h, max_x = 0, max_y = 0;
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x)
            if (src[((size_t)y * w + x) * 4 + 3] > 0) {
                if (x < min_x) min_x = x;
                if (y < min_y) min_y = y;
                if (x > max_x) max_x = x;
                if (y > max_y) max_y = y;
            }
    if (min_x == w) return;
    long pad = (long)sr + 1;
    long p0x = min_x - pad < 0 ? 0 : min_x - pad, p0y = min_y - pad < 0 ? 0 : min_y - pad;
    long p1x = max_x + pad > w - 1 ? w - 1 : max_x + pad, p1y = max_y + pad > h - 1 ? h - 1 : max_y + pad;
    float ca = (float)color[3] / 255.0f;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = p0y; y <= p1y; ++y)
        for (long x = p0x; x <= p1x; ++x) {
            if (masked_out(mask, w, x, y)) continue;
            size_t pi = ((size_t)y * w + x) * 4;
            float src_a = (float)src[pi + 3] / 255.0f;
            float d, outside_cov = 0.0f, inside_cov = 0.0f;
            if (nearest_distance(src, w, h, x, y, sr, 1, &d)) outside_cov = shell_coverage(fmaxf(d - 1.0f, 0.0f), radius, anti_alias);
            outside_cov = outside_cov * (1.0f - src_a);
            if (nearest_distance(src, w, h, x, y, sr, 0, &d)) inside_cov = shell_coverage(d, radius, anti_alias);
            inside_cov = inside_cov * src_a;
            float under_cov = mode == 1 ? 0.0f : outside_cov, over_cov = mode == 0 ? 0.0f : inside_cov;
            float a_under = ca * under_cov, a_over = ca * over_cov;
            float comp[3] = {(float)src[pi] / 255.0f, (float)src[pi + 1] / 255.0f, (float)src[pi + 2] / 255.0f};
            float comp_a = (float)src[pi + 3] / 255.0f;
            if (a_under > 0.0f) {
                float out_a = comp_a + a_under * (1.0f - comp_a);
                if (out_a > 0.0f)
                    for (int c = 0; c < 3; ++c)
                        comp[c] = (comp[c] * comp_a + ((float)color[c] / 255.0f) * a_under * (1.0f - comp_a)) / out_a;
                comp_a = out_a;
            }
            if (a_over > 0.0f) {
                float out_a = a_over + comp_a * (1.0f - a_over);
                if (out_a > 0.0f)
                    for (int c = 0; c < 3; ++c)
                        comp[c] = (((float)color[c] / 255.0f) * a_over + comp[c] * comp_a * (1.0f - a_over)) / out_a;
                comp_a = out_a;
            }
            for (int c = 0; c < 3; ++c) dst[pi + c] = rs_f32_as_u8(roundf(rs_clampf(comp[c], 0.0f, 1.0f) * 255.0f));
            dst[pi + 3] = rs_f32_as_u8(roundf(rs_clampf(comp_a, 0.0f, 1.0f) * 255.0f));
        }
}

/* glitch.rs:44-99 */
void pfxo_pixel_drag(const uint8_t* src, uint32_t w32, uint32_t h32, uint32_t seed, float amount, uint32_t distance,
                     float direction, const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    memcpy(dst, src, (size_t)w * h * 4);
    if (w == 0 || h == 0) return;
    float dir_rad = to_radians(direction);
    float dx_dir = cosf(dir_rad), dy_dir = sinf(dir_rad);
    float dist = (float)(distance < 1 ? 1 : distance);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y) {
        if (hash_f32((uint32_t)y, 0, seed) > amount / 100.0f) continue;
        int32_t drag = rs_f32_as_i32(hash_f32((uint32_t)y, 1, seed) * dist);
        for (long x = 0; x < w; ++x) {
            if (masked_out(mask, w, x, y)) continue;
            long sx = clampl(rs_f32_as_i32(roundf((float)x - (float)drag * dx_dir)), 0, w - 1);
            long sy = clampl(rs_f32_as_i32(roundf((float)y - (float)drag * dy_dir)), 0, h - 1);
            memcpy(dst + ((size_t)y * w + x) * 4, src + ((size_t)sy * w + sx) * 4, 4);
        }
    }
}

/* glitch.rs:142-196; offsets = {rx, ry, gx, gy, bx, by} */
void pfxo_rgb_displace(const uint8_t* src, uint32_t w32, uint32_t h32, const int32_t off[6], const uint8_t* mask, uint8_t* dst,
                       int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            for (int c = 0; c < 3; ++c) {
                long sx = clampl(x + off[c * 2], 0, w - 1), sy = clampl(y + off[c * 2 + 1], 0, h - 1);
                dst[oi + c] = src[((size_t)sy * w + sx) * 4 + c];
            }
            dst[oi + 3] = src[oi + 3];
        }
}

/* artistic.rs:31-99 */
static inline float ink_lum(const uint8_t* src, long w, long h, long px, long py)
{
    const uint8_t* p = src + ((size_t)clampl(py, 0, h - 1) * w + clampl(px, 0, w - 1)) * 4;
    return 0.2126f * (float)p[0] + 0.7152f * (float)p[1] + 0.0722f * (float)p[2];
}
void pfxo_ink(const uint8_t* src, uint32_t w32, uint32_t h32, float edge_strength, float threshold, const uint8_t* mask,
              uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
#define L(ax, ay) ink_lum(src, w, h, x + (ax), y + (ay))
            float gx = -L(-1, -1) - 2.0f * L(-1, 0) - L(-1, 1) + L(1, -1) + 2.0f * L(1, 0) + L(1, 1);
            float gy = -L(-1, -1) - 2.0f * L(0, -1) - L(1, -1) + L(-1, 1) + 2.0f * L(0, 1) + L(1, 1);
#undef L
            float edge = sqrtf(gx * gx + gy * gy) * edge_strength / 100.0f;
            uint8_t val = edge > threshold ? 0 : 255;
            dst[oi] = dst[oi + 1] = dst[oi + 2] = val;
            dst[oi + 3] = src[oi + 3];
        }
}

/* artistic.rs:123-215 */
void pfxo_oil_painting(const uint8_t* src, uint32_t w32, uint32_t h32, uint32_t radius, uint32_t levels, const uint8_t* mask,
                       uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    int32_t r = (int32_t)(radius < 1 ? 1 : (radius > 10 ? 10 : radius));
    uint32_t nl = levels < 2 ? 2 : (levels > 64 ? 64 : levels);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            uint32_t cnt[64] = {0}, sr[64] = {0}, sg[64] = {0}, sb[64] = {0};
            for (int32_t dy = -r; dy <= r; ++dy) {
                long sy = clampl(y + dy, 0, h - 1);
                for (int32_t dx = -r; dx <= r; ++dx) {
                    const uint8_t* p = src + ((size_t)sy * w + clampl(x + dx, 0, w - 1)) * 4;
                    uint32_t pr = p[0], pg = p[1], pb = p[2];
                    uint32_t k = (pr + pg + pb) / 3 * nl / 256;
                    if (k > nl - 1) k = nl - 1;
                    cnt[k] += 1; sr[k] += pr; sg[k] += pg; sb[k] += pb;
                }
            }
            uint32_t max_count = 0, max_idx = 0;
            for (uint32_t i = 0; i < nl; ++i)
                if (cnt[i] > max_count) { max_count = cnt[i]; max_idx = i; }
            dst[oi] = dst[oi + 1] = dst[oi + 2] = 0;
            if (max_count > 0) {
                dst[oi] = (uint8_t)(sr[max_idx] / max_count);
                dst[oi + 1] = (uint8_t)(sg[max_idx] / max_count);
                dst[oi + 2] = (uint8_t)(sb[max_idx] / max_count);
            }
            dst[oi + 3] = src[oi + 3];
        }
}

/* artistic.rs:266-309; mode: 0 multiply, 1 screen, 2 overlay, 3 soft light */
static inline float cf_blend(int mode, float s, float f)
{
    switch (mode) {
    case 0: return s * f;
    case 1: return 1.0f - (1.0f - s) * (1.0f - f);
    case 2: return s < 0.5f ? 2.0f * s * f : 1.0f - 2.0f * (1.0f - s) * (1.0f - f);
    default: return f < 0.5f ? s - (1.0f - 2.0f * f) * s * (1.0f - s) : s + (2.0f * f - 1.0f) * (sqrtf(s) - s);
    }
}
void pfxo_color_filter(const uint8_t* src, uint32_t w32, uint32_t h32, const uint8_t filter_color[4], float intensity, int mode,
                       const uint8_t* mask, uint8_t* dst, int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    float fc[3] = {(float)filter_color[0] / 255.0f, (float)filter_color[1] / 255.0f, (float)filter_color[2] / 255.0f};
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < w * h; ++i) {
        size_t oi = (size_t)i * 4;
        if (mask && mask[i] == 0) { memcpy(dst + oi, src + oi, 4); continue; }
        for (int c = 0; c < 3; ++c) {
            float s = (float)src[oi + c] / 255.0f;
            dst[oi + c] = round_u8((s * (1.0f - intensity) + cf_blend(mode, s, fc[c]) * intensity) * 255.0f);
        }
        dst[oi + 3] = src[oi + 3];
    }
}

/* contours.rs:56-112 */
void pfxo_contours(const uint8_t* src, uint32_t w32, uint32_t h32, float scale, float frequency, float line_width,
                   const uint8_t line_color[4], uint32_t seed, uint32_t octaves, float blend, const uint8_t* mask, uint8_t* dst,
                   int threads)
{
    long w = w32, h = h32;
    if (w == 0 || h == 0) return;
    float inv_scale = 1.0f / fmaxf(scale, 0.5f);
    uint32_t oct = octaves < 1 ? 1 : (octaves > 8 ? 8 : octaves);
    float half_lw = fmaxf(line_width * 0.5f, 0.3f);
    float lc[3] = {(float)line_color[0], (float)line_color[1], (float)line_color[2]};
    float la = (float)line_color[3] / 255.0f;
    float freq = fmaxf(frequency, 0.5f);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < h; ++y)
        for (long x = 0; x < w; ++x) {
            size_t oi = ((size_t)y * w + x) * 4;
            if (masked_out(mask, w, x, y)) { memcpy(dst + oi, src + oi, 4); continue; }
            float noise_val = turbulence_2d((float)x * inv_scale, (float)y * inv_scale, seed, oct, 0.5f);
            float level = noise_val * freq;
            float dist_to_contour = fabsf(level - roundf(level)) / freq;
            float edge = half_lw * inv_scale * 0.5f;
            float line_alpha = dist_to_contour < edge ? 1.0f
                             : (dist_to_contour < edge * 2.0f ? 1.0f - (dist_to_contour - edge) / edge : 0.0f);
            float alpha = line_alpha * la * blend;
            for (int c = 0; c < 3; ++c) dst[oi + c] = round_u8((float)src[oi + c] * (1.0f - alpha) + lc[c] * alpha);
            dst[oi + 3] = src[oi + 3];
        }
}
