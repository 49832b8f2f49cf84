/*
 * o_adjust.c — oracle restatement of the pointwise adjustment bank (three numeric flavours).
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ops/adjustments.rs:21-42    apply_pixel_transform (populated chunks only, `.round()`)
 *   src/ops/adjustments.rs:46-108   apply_pixel_transform_from_flat (+ from_rgba_image re-sparsify :105)
 *   src/ops/adjustments.rs:115-140  invert_colors / invert_alpha / sepia
 *   src/ops/adjustments.rs:144-256  auto_levels + build_stretch_lut
 *   src/ops/adjustments.rs:265-416  brightness_contrast / hsl / exposure / highlights_shadows
 *   src/ops/adjustments.rs:465-488  build_levels_lut        :517-526 temperature_tint
 *   src/ops/adjustments.rs:584-729  build_multi_channel_luts / build_curves_lut
 *   src/ops/adjustments.rs:944-1012 rgb_to_hsl / hsl_to_rgb / hue_to_rgb
 *   src/ops/adjustments.rs:1240-1441 threshold / posterize / color_balance / gradient_map / b&w / vibrance
 *   src/ops/filters.rs:321-378      desaturate_layer (BT.709, round)
 *   src/ops/scripting.rs:869-1075   the Rhai-inline flavour (truncating `as u8`, alpha untouched, mask ignored)
 */
#include "o_common.h"

/* ---------- colour helpers: adjustments.rs:944-1012 ---------- */
static void rgb_to_hsl(float r, float g, float b, float* oh, float* os, float* ol)
{
    float max = fmaxf(fmaxf(r, g), b);
    float min = fminf(fminf(r, g), b);
    float l = (max + min) / 2.0f;
    if (fabsf(max - min) < 1e-6f) { *oh = 0.0f; *os = 0.0f; *ol = l; return; }
    float d = max - min;
    float s = (l > 0.5f) ? d / (2.0f - max - min) : d / (max + min);
    float h;
    if (fabsf(max - r) < 1e-6f) {
        h = (g - b) / d;
        if (h < 0.0f) h += 6.0f;
        h = h / 6.0f;
    } else if (fabsf(max - g) < 1e-6f) {
        h = ((b - r) / d + 2.0f) / 6.0f;
    } else {
        h = ((r - g) / d + 4.0f) / 6.0f;
    }
    *oh = h; *os = s; *ol = l;
}
static float hue_to_rgb(float p, float q, float t)
{
    if (t < 0.0f) t += 1.0f;
    if (t > 1.0f) t -= 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 1.0f / 2.0f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}
static void hsl_to_rgb(float h, float s, float l, float* r, float* g, float* b)
{
    if (fabsf(s) < 1e-6f) { *r = l; *g = l; *b = l; return; }
    float q = (l < 0.5f) ? l * (1.0f + s) : l + s - l * s;
    float p = 2.0f * l - q;
    *r = hue_to_rgb(p, q, h + 1.0f / 3.0f);
    *g = hue_to_rgb(p, q, h);
    *b = hue_to_rgb(p, q, h - 1.0f / 3.0f);
}
static inline float rs_fract(float x) { return x - truncf(x); }
void pfxo_rgb_to_hsl(float r, float g, float b, float* h, float* s, float* l) { rgb_to_hsl(r, g, b, h, s, l); }
void pfxo_hsl_to_rgb(float h, float s, float l, float* r, float* g, float* b) { hsl_to_rgb(h, s, l, r, g, b); }

/* ---------- LUT builders ---------- */
/* adjustments.rs:465-488 */
void pfxo_levels_lut(float in_black, float in_white, float gamma, float out_black, float out_white, uint8_t lut[256])
{
    float in_range = fmaxf(in_white - in_black, 1.0f);
    float out_range = out_white - out_black;
    float inv_gamma = 1.0f / fmaxf(gamma, 0.01f);
    for (int i = 0; i < 256; ++i) {
        float v = (float)i;
        float normalized = rs_clampf((v - in_black) / in_range, 0.0f, 1.0f);
        float gamma_corrected = powf(normalized, inv_gamma);
        float output = out_black + gamma_corrected * out_range;
        lut[i] = rs_f32_as_u8(rs_clampf(roundf(output), 0.0f, 255.0f));
    }
}
/* scripting.rs:1050-1061 (apply_levels: truncating, output range fixed 0..255) */
void pfxo_rhai_levels_lut(float in_black, float in_white, float gamma, uint8_t lut[256])
{
    float in_range = fmaxf(in_white - in_black, 1.0f);
    float inv_gamma = 1.0f / fmaxf(gamma, 0.01f);
    for (int i = 0; i < 256; ++i) {
        float normalized = rs_clampf(((float)i - in_black) / in_range, 0.0f, 1.0f);
        float gamma_corrected = powf(normalized, inv_gamma);
        lut[i] = rs_f32_as_u8(rs_clampf(gamma_corrected * 255.0f, 0.0f, 255.0f));
    }
}
/* adjustments.rs:235-256 */
void pfxo_stretch_lut(uint8_t min, uint8_t max, uint8_t lut[256])
{
    if (max <= min) { for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i; return; }
    float range = (float)(max - min);
    for (int i = 0; i < 256; ++i) {
        float v;
        if ((uint8_t)i <= min) v = 0.0f;
        else if ((uint8_t)i >= max) v = 255.0f;
        else v = ((float)i - (float)min) / range * 255.0f;
        lut[i] = rs_f32_as_u8(rs_clampf(roundf(v), 0.0f, 255.0f));
    }
}
/* adjustments.rs:640-729 (Fritsch–Carlson monotone cubic) */
void pfxo_curves_lut(const float* pts, int n, uint8_t lut[256])
{
    if (n < 2) { for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i; return; }
#define PX(i) pts[(i) * 2]
#define PY(i) pts[(i) * 2 + 1]
    float* delta = (float*)malloc(sizeof(float) * (size_t)(n - 1));
    float* m = (float*)calloc((size_t)n, sizeof(float));
    for (int i = 0; i < n - 1; ++i) {
        float dx = PX(i + 1) - PX(i), dy = PY(i + 1) - PY(i);
        delta[i] = (fabsf(dx) < 1e-6f) ? 0.0f : dy / dx;
    }
    m[0] = delta[0];
    m[n - 1] = delta[n - 2];
    for (int i = 1; i < n - 1; ++i)
        m[i] = (delta[i - 1] * delta[i] <= 0.0f) ? 0.0f : (delta[i - 1] + delta[i]) / 2.0f;
    for (int i = 0; i < n - 1; ++i) {
        if (fabsf(delta[i]) < 1e-6f) { m[i] = 0.0f; m[i + 1] = 0.0f; }
        else {
            float alpha = m[i] / delta[i], beta = m[i + 1] / delta[i];
            float s = alpha * alpha + beta * beta;
            if (s > 9.0f) {
                float tau = 3.0f / sqrtf(s);
                m[i] = tau * alpha * delta[i];
                m[i + 1] = tau * beta * delta[i];
            }
        }
    }
    for (int i = 0; i < 256; ++i) {
        float x = (float)i;
        int seg = 0;
        for (int j = 0; j < n - 1; ++j) if (x >= PX(j)) seg = j;
        float val;
        if (x <= PX(0)) val = PY(0);
        else if (x >= PX(n - 1)) val = PY(n - 1);
        else {
            float x0 = PX(seg), x1 = PX(seg + 1), y0 = PY(seg), y1 = PY(seg + 1);
            float h = x1 - x0;
            if (fabsf(h) < 1e-6f) val = y0;
            else {
                float t = (x - x0) / h, t2 = t * t, t3 = t2 * t;
                float h00 = 2.0f * t3 - 3.0f * t2 + 1.0f;
                float h10 = t3 - 2.0f * t2 + t;
                float h01 = -2.0f * t3 + 3.0f * t2;
                float h11 = t3 - t2;
                val = h00 * y0 + h10 * h * m[seg] + h01 * y1 + h11 * h * m[seg + 1];
            }
        }
        lut[i] = rs_f32_as_u8(rs_clampf(roundf(val), 0.0f, 255.0f));
    }
    free(delta);
    free(m);
#undef PX
#undef PY
}
/* adjustments.rs:584-631: [RGB, R, G, B, A] curves -> composed R,G,B,A tables.
 * pts[k] may be NULL / n[k]==0 with enabled[k]==0 for a disabled channel. */
void pfxo_curves_luts_multi(const float* const pts[5], const int n[5], const int enabled[5], uint8_t out[4 * 256])
{
    uint8_t l[5][256];
    for (int k = 0; k < 5; ++k) {
        if (enabled[k]) pfxo_curves_lut(pts[k], n[k], l[k]);
        else for (int i = 0; i < 256; ++i) l[k][i] = (uint8_t)i;
    }
    for (int i = 0; i < 256; ++i) {
        out[0 * 256 + i] = l[1][l[0][i]];
        out[1 * 256 + i] = l[2][l[0][i]];
        out[2 * 256 + i] = l[3][l[0][i]];
        out[3 * 256 + i] = l[4][i];
    }
}
/* adjustments.rs:144-233: per-channel min/max over selected, non-transparent pixels -> stretch LUTs */
void pfxo_auto_levels_luts(const uint8_t* src, uint32_t w, uint32_t h, const uint8_t* mask, uint8_t out[4 * 256])
{
    uint8_t mn[3] = {255, 255, 255}, mx[3] = {0, 0, 0};
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (mask && mask[i] == 0) continue;
        if (src[i * 4 + 3] == 0) continue;
        for (int c = 0; c < 3; ++c) {
            uint8_t v = src[i * 4 + c];
            if (v < mn[c]) mn[c] = v;
            if (v > mx[c]) mx[c] = v;
        }
    }
    for (int c = 0; c < 3; ++c) pfxo_stretch_lut(mn[c], mx[c], out + c * 256);
    for (int i = 0; i < 256; ++i) out[3 * 256 + i] = (uint8_t)i;
}

/* ---------- the per-pixel closures (f32 in 0..255 -> f32) ---------- */
static void px_fn(int op, const float* p, const uint8_t* lut, float r, float g, float b, float a, float o[4])
{
    switch (op) {
    case PFXO_OP_INVERT: o[0] = 255.0f - r; o[1] = 255.0f - g; o[2] = 255.0f - b; o[3] = a; return;
    case PFXO_OP_INVERT_ALPHA: o[0] = r; o[1] = g; o[2] = b; o[3] = 255.0f - a; return;
    case PFXO_OP_SEPIA: {
        float sr = 0.393f * r + 0.769f * g + 0.189f * b;
        float sg = 0.349f * r + 0.686f * g + 0.168f * b;
        float sb = 0.272f * r + 0.534f * g + 0.131f * b;
        o[0] = fminf(sr, 255.0f); o[1] = fminf(sg, 255.0f); o[2] = fminf(sb, 255.0f); o[3] = a;
        return;
    }
    case PFXO_OP_BRIGHTNESS_CONTRAST: { /* p = [brightness, contrast] */
        float brightness = p[0], contrast = p[1];
        float factor = (259.0f * (contrast + 255.0f)) / (255.0f * (259.0f - contrast));
        o[0] = factor * (r + brightness - 128.0f) + 128.0f;
        o[1] = factor * (g + brightness - 128.0f) + 128.0f;
        o[2] = factor * (b + brightness - 128.0f) + 128.0f;
        o[3] = a;
        return;
    }
    case PFXO_OP_HSL: { /* p = [hue_shift deg, saturation, lightness] */
        float sat_factor = 1.0f + p[1] / 100.0f;
        float light_offset = p[2] * 255.0f / 100.0f;
        float h, s, l, nr, ng, nb;
        rgb_to_hsl(r / 255.0f, g / 255.0f, b / 255.0f, &h, &s, &l);
        float nh = rs_fract(h + p[0] / 360.0f);
        if (nh < 0.0f) nh = nh + 1.0f;
        float ns = rs_clampf(s * sat_factor, 0.0f, 1.0f);
        hsl_to_rgb(nh, ns, l, &nr, &ng, &nb);
        o[0] = nr * 255.0f + light_offset; o[1] = ng * 255.0f + light_offset; o[2] = nb * 255.0f + light_offset;
        o[3] = a;
        return;
    }
    case PFXO_OP_EXPOSURE: {
        float gain = powf(2.0f, p[0]);
        o[0] = r * gain; o[1] = g * gain; o[2] = b * gain; o[3] = a;
        return;
    }
    case PFXO_OP_HIGHLIGHTS_SHADOWS: { /* p = [shadows, highlights] */
        float shadow_amt = p[0] / 100.0f, highlight_amt = p[1] / 100.0f;
        float lum = (0.2126f * r + 0.7152f * g + 0.0722f * b) / 255.0f;
        float sw = (1.0f - lum) * (1.0f - lum);
        float hw = lum * lum;
        float adjustment = sw * shadow_amt * 128.0f + hw * highlight_amt * 128.0f;
        o[0] = r + adjustment; o[1] = g + adjustment; o[2] = b + adjustment; o[3] = a;
        return;
    }
    case PFXO_OP_TEMPERATURE_TINT: { /* p = [temperature, tint] */
        float temp_shift = p[0] * 1.5f, tint_shift = p[1] * 1.0f;
        o[0] = r + temp_shift; o[1] = g - tint_shift * 0.5f; o[2] = b - temp_shift; o[3] = a;
        return;
    }
    case PFXO_OP_THRESHOLD: {
        float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
        float v = (lum >= p[0]) ? 255.0f : 0.0f;
        o[0] = v; o[1] = v; o[2] = v; o[3] = a;
        return;
    }
    case PFXO_OP_POSTERIZE: { /* p[0] = levels; `levels.max(2) as f32` (adjustments.rs:1268) */
        float factor = p[0] < 2.0f ? 2.0f : p[0];
        o[0] = roundf(r / 255.0f * (factor - 1.0f)) / (factor - 1.0f) * 255.0f;
        o[1] = roundf(g / 255.0f * (factor - 1.0f)) / (factor - 1.0f) * 255.0f;
        o[2] = roundf(b / 255.0f * (factor - 1.0f)) / (factor - 1.0f) * 255.0f;
        o[3] = a;
        return;
    }
    case PFXO_OP_COLOR_BALANCE: { /* p = shadows[3], midtones[3], highlights[3] */
        float lum = (0.2126f * r + 0.7152f * g + 0.0722f * b) / 255.0f;
        float sw0 = fmaxf(1.0f - lum * 2.0f, 0.0f), hw0 = fmaxf(lum * 2.0f - 1.0f, 0.0f);
        float sw = sw0 * sw0, hw = hw0 * hw0;
        float mw = fmaxf(1.0f - sw - hw, 0.0f);
        float adj_r = sw * p[0] + mw * p[3] + hw * p[6];
        float adj_g = sw * p[1] + mw * p[4] + hw * p[7];
        float adj_b = sw * p[2] + mw * p[5] + hw * p[8];
        o[0] = r + adj_r * 1.28f; o[1] = g + adj_g * 1.28f; o[2] = b + adj_b * 1.28f; o[3] = a;
        return;
    }
    case PFXO_OP_GRADIENT_MAP: { /* lut = 256 x RGBA */
        uint32_t lum = rs_f32_as_u32(0.2126f * r + 0.7152f * g + 0.0722f * b);
        if (lum > 255) lum = 255;
        o[0] = (float)lut[lum * 4 + 0]; o[1] = (float)lut[lum * 4 + 1]; o[2] = (float)lut[lum * 4 + 2]; o[3] = a;
        return;
    }
    case PFXO_OP_BLACK_AND_WHITE: { /* p = weights r,g,b */
        float v = (r * p[0] + g * p[1] + b * p[2]) / 100.0f;
        v = rs_clampf(v, 0.0f, 255.0f);
        o[0] = v; o[1] = v; o[2] = v; o[3] = a;
        return;
    }
    case PFXO_OP_VIBRANCE: { /* p[0] = amount */
        float v = p[0] / 100.0f;
        float h, s, l, nr, ng, nb;
        rgb_to_hsl(r / 255.0f, g / 255.0f, b / 255.0f, &h, &s, &l);
        float boost = (v >= 0.0f) ? v * ((1.0f - s) * (1.0f - s)) : v * (s * s);
        float ns = rs_clampf(s + boost, 0.0f, 1.0f);
        hsl_to_rgb(h, ns, l, &nr, &ng, &nb);
        o[0] = nr * 255.0f; o[1] = ng * 255.0f; o[2] = nb * 255.0f; o[3] = a;
        return;
    }
    case PFXO_OP_LUT_RGBA: /* lut = R[256] G[256] B[256] A[256]  (levels / curves / auto-levels) */
        o[0] = (float)lut[(int)r]; o[1] = (float)lut[256 + (int)g]; o[2] = (float)lut[512 + (int)b];
        o[3] = (float)lut[768 + (int)a];
        return;
    case PFXO_OP_DESATURATE: { /* filters.rs:360-370 */
        float lum = 0.2126f * r + 0.7152f * g + 0.0722f * b;
        o[0] = lum; o[1] = lum; o[2] = lum; o[3] = a;
        return;
    }
    default: o[0] = r; o[1] = g; o[2] = b; o[3] = a;
    }
}

static inline uint8_t round_u8(float v) { return rs_f32_as_u8(rs_clampf(roundf(v), 0.0f, 255.0f)); }

/* Dense driver.  sparse_mode:
 *   PFXO_DENSE        plain `_from_flat` arithmetic on every pixel (no TiledImage effects)
 *   PFXO_FROM_FLAT    `_from_flat` + `TiledImage::from_rgba_image(&out)` (:105): output chunks whose alpha is
 *                     all zero come back as zeros (what extract_layer sees in the reference tests)
 *   PFXO_IN_PLACE     `apply_pixel_transform` (:21-42): input is first sparsified (canvas_from_image), only
 *                     populated chunks are visited; result chunks are NOT re-sparsified */
void pfxo_adjust(const uint8_t* src, uint32_t w, uint32_t h, int op, const float* params, const uint8_t* lut,
                 const uint8_t* mask, int sparse_mode, uint8_t* dst, int threads)
{
    size_t n = (size_t)w * h;
    uint32_t cxn = (w + PFXO_CHUNK - 1) / PFXO_CHUNK, cyn = (h + PFXO_CHUNK - 1) / PFXO_CHUNK;
    uint8_t* pop = NULL;
    if (sparse_mode == PFXO_IN_PLACE) {
        pop = (uint8_t*)malloc((size_t)cxn * cyn);
        pfxo_chunk_populated(src, w, h, pop);
    }
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) {
        const uint8_t* s = src + (size_t)i * 4;
        uint8_t* d = dst + (size_t)i * 4;
        if (pop && !pop[o_chunk_index((uint32_t)(i % w), (uint32_t)(i / w), w)]) { memset(d, 0, 4); continue; }
        if (mask && mask[i] == 0) { memcpy(d, s, 4); continue; }
        float o[4];
        px_fn(op, params, lut, (float)s[0], (float)s[1], (float)s[2], (float)s[3], o);
        d[0] = round_u8(o[0]); d[1] = round_u8(o[1]); d[2] = round_u8(o[2]); d[3] = round_u8(o[3]);
    }
    if (sparse_mode == PFXO_FROM_FLAT) {
        uint8_t* tmp = (uint8_t*)malloc(n * 4);
        memcpy(tmp, dst, n * 4);
        pfxo_tiled_roundtrip(tmp, w, h, dst);
        free(tmp);
    }
    free(pop);
}

/* ---------- Rhai-inline flavour: scripting.rs:869-1075 (in place, alpha untouched, truncating) ---------- */
static inline uint8_t trunc_u8(float v) { return rs_f32_as_u8(v); }

void pfxo_rhai_adjust(uint8_t* px, size_t n_px, int op, const float* p)
{
    uint8_t lut[256];
    if (op == PFXO_RHAI_LEVELS) pfxo_rhai_levels_lut(p[0], p[1], p[2], lut);
    for (size_t i = 0; i < n_px; ++i) {
        uint8_t* q = px + i * 4;
        switch (op) {
        case PFXO_RHAI_INVERT: q[0] = 255 - q[0]; q[1] = 255 - q[1]; q[2] = 255 - q[2]; break; /* :870-881 */
        case PFXO_RHAI_DESATURATE: { /* :884-898 */
            uint32_t gray = ((uint32_t)q[0] * 299 + (uint32_t)q[1] * 587 + (uint32_t)q[2] * 114) / 1000;
            q[0] = q[1] = q[2] = (uint8_t)gray;
            break;
        }
        case PFXO_RHAI_SEPIA: { /* :901-918 */
            float r = q[0], g = q[1], b = q[2];
            q[0] = trunc_u8(fminf(r * 0.393f + g * 0.769f + b * 0.189f, 255.0f));
            q[1] = trunc_u8(fminf(r * 0.349f + g * 0.686f + b * 0.168f, 255.0f));
            q[2] = trunc_u8(fminf(r * 0.272f + g * 0.534f + b * 0.131f, 255.0f));
            break;
        }
        case PFXO_RHAI_SEPIA_STRENGTH: { /* :921-941; p[0] already clamped to 0..1 by the caller in f64 */
            float strength = p[0], inv = 1.0f - strength;
            float r = q[0], g = q[1], b = q[2];
            float sr = fminf(r * 0.393f + g * 0.769f + b * 0.189f, 255.0f);
            float sg = fminf(r * 0.349f + g * 0.686f + b * 0.168f, 255.0f);
            float sb = fminf(r * 0.272f + g * 0.534f + b * 0.131f, 255.0f);
            q[0] = trunc_u8(r * inv + sr * strength);
            q[1] = trunc_u8(g * inv + sg * strength);
            q[2] = trunc_u8(b * inv + sb * strength);
            break;
        }
        case PFXO_RHAI_BRIGHTNESS_CONTRAST: { /* :944-965 */
            float bright = p[0], contrast = p[1];
            float factor = (259.0f * (contrast + 255.0f)) / (255.0f * (259.0f - contrast));
            for (int c = 0; c < 3; ++c)
                q[c] = trunc_u8(rs_clampf(factor * ((float)q[c] + bright - 128.0f) + 128.0f, 0.0f, 255.0f));
            break;
        }
        case PFXO_RHAI_HSL: { /* :968-1034 */
            float hue_shift = p[0], sat_factor = 1.0f + p[1] / 100.0f, light_offset = p[2] * 255.0f / 100.0f;
            float r = (float)q[0] / 255.0f, g = (float)q[1] / 255.0f, b = (float)q[2] / 255.0f;
            float cmax = fmaxf(fmaxf(r, g), b), cmin = fminf(fminf(r, g), b);
            float l = (cmax + cmin) / 2.0f;
            float h, s;
            if (fabsf(cmax - cmin) < 1e-10f) { h = 0.0f; s = 0.0f; }
            else {
                float d = cmax - cmin;
                s = (l > 0.5f) ? d / (2.0f - cmax - cmin) : d / (cmax + cmin);
                float hh;
                if (fabsf(cmax - r) < 1e-10f) hh = (g - b) / d + ((g < b) ? 6.0f : 0.0f);
                else if (fabsf(cmax - g) < 1e-10f) hh = (b - r) / d + 2.0f;
                else hh = (r - g) / d + 4.0f;
                h = hh / 6.0f;
            }
            float nh = h + hue_shift / 360.0f;
            { /* f32::rem_euclid(1.0): r = x % 1.0; if r < 0 { r + 1.0 } else { r } */
                float rr = fmodf(nh, 1.0f);
                nh = (rr < 0.0f) ? rr + 1.0f : rr;
            }
            float ns = rs_clampf(s * sat_factor, 0.0f, 1.0f);
            float nr, ng, nb;
            if (fabsf(ns) < 1e-10f) { nr = l; ng = l; nb = l; }
            else {
                float qq = (l < 0.5f) ? l * (1.0f + ns) : l + ns - l * ns;
                float pp = 2.0f * l - qq;
                nr = hue_to_rgb(pp, qq, nh + 1.0f / 3.0f);
                ng = hue_to_rgb(pp, qq, nh);
                nb = hue_to_rgb(pp, qq, nh - 1.0f / 3.0f);
            }
            q[0] = trunc_u8(rs_clampf(nr * 255.0f + light_offset, 0.0f, 255.0f));
            q[1] = trunc_u8(rs_clampf(ng * 255.0f + light_offset, 0.0f, 255.0f));
            q[2] = trunc_u8(rs_clampf(nb * 255.0f + light_offset, 0.0f, 255.0f));
            break;
        }
        case PFXO_RHAI_EXPOSURE: { /* :1037-1047 */
            float gain = powf(2.0f, p[0]);
            for (int c = 0; c < 3; ++c) q[c] = trunc_u8(rs_clampf((float)q[c] * gain, 0.0f, 255.0f));
            break;
        }
        case PFXO_RHAI_LEVELS: /* :1050-1071 */
            q[0] = lut[q[0]]; q[1] = lut[q[1]]; q[2] = lut[q[2]];
            break;
        default: break;
        }
    }
}
