/*
 * o_composite.c — oracle restatement of PaintFE's CPU compositor.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows, line by line:
 *   src/canvas/canvas_state.rs:1246-1422  blend_pixel_static
 *   src/canvas/canvas_state.rs:1425-1505  per-channel helpers
 *   src/canvas/canvas_state.rs:505-698    composite_viewport (viewport=None, no preview layer)
 *   src/canvas/layers.rs:276-325          AdjustmentLayerData::apply_to_pixel(_with_opacity)
 *   src/canvas/tiled_image.rs:50-104      chunk sparsity rule
 */
#include "o_common.h"

/* ---- helpers canvas_state.rs:1425-1505 (all f32, literals f32) ---- */
static inline float overlay_channel(float base, float top)
{
    if (base < 0.5f) return 2.0f * base * top;
    return 1.0f - 2.0f * (1.0f - base) * (1.0f - top);
}
static inline float color_burn_channel(float base, float top)
{
    if (top == 0.0f) return 0.0f;
    return fmaxf(1.0f - (1.0f - base) / top, 0.0f);
}
static inline float color_dodge_channel(float base, float top)
{
    if (top >= 1.0f) return 1.0f;
    return fminf(base / (1.0f - top), 1.0f);
}
static inline float reflect_channel(float base, float top)
{
    if (top >= 1.0f) return 1.0f;
    return fminf(base * base / (1.0f - top), 1.0f);
}
static inline float soft_light_channel(float base, float top)
{
    if (top <= 0.5f) return base - (1.0f - 2.0f * top) * base * (1.0f - base);
    float d;
    if (base <= 0.25f) d = ((16.0f * base - 12.0f) * base + 4.0f) * base;
    else d = sqrtf(base);
    return base + (2.0f * top - 1.0f) * (d - base);
}
static inline float divide_channel(float base, float top)
{
    if (top <= 0.0f) return 1.0f;
    return fminf(base / top, 1.0f);
}
static inline float vivid_light_channel(float base, float top)
{
    if (top <= 0.5f) {
        float t2 = 2.0f * top;
        if (t2 <= 0.0f) return 0.0f;
        return fmaxf(1.0f - (1.0f - base) / t2, 0.0f);
    } else {
        float t2 = 2.0f * (top - 0.5f);
        if (t2 >= 1.0f) return 1.0f;
        return fminf(base / (1.0f - t2), 1.0f);
    }
}
static inline float pin_light_channel(float base, float top)
{
    if (top <= 0.5f) return fminf(base, 2.0f * top);
    return fmaxf(base, 2.0f * (top - 0.5f));
}

static inline float blend_fn(int mode, float b, float t)
{
    switch (mode) { /* canvas_state.rs:1304-1405 */
    case 0: return t;                                             /* Normal */
    case 1: return b * t;                                         /* Multiply */
    case 2: return 1.0f - (1.0f - b) * (1.0f - t);                /* Screen */
    case 3: return fminf(b + t, 1.0f);                            /* Additive */
    case 4: return reflect_channel(b, t);                         /* Reflect */
    case 5: return reflect_channel(t, b);                         /* Glow */
    case 6: return color_burn_channel(b, t);                      /* ColorBurn */
    case 7: return color_dodge_channel(b, t);                     /* ColorDodge */
    case 8: return overlay_channel(b, t);                         /* Overlay */
    case 9: return fabsf(b - t);                                  /* Difference */
    case 10: return 1.0f - fabsf(1.0f - b - t);                   /* Negation */
    case 11: return fmaxf(b, t);                                  /* Lighten */
    case 12: return fminf(b, t);                                  /* Darken */
    case 15: return overlay_channel(t, b);                        /* HardLight */
    case 16: return soft_light_channel(b, t);                     /* SoftLight */
    case 17: return b + t - 2.0f * b * t;                         /* Exclusion */
    case 18: return fmaxf(b - t, 0.0f);                           /* Subtract */
    case 19: return divide_channel(b, t);                         /* Divide */
    case 20: return fmaxf(b + t - 1.0f, 0.0f);                    /* LinearBurn */
    case 21: return vivid_light_channel(b, t);                    /* VividLight */
    case 22: return rs_clampf(b + 2.0f * t - 1.0f, 0.0f, 1.0f);   /* LinearLight */
    case 23: return pin_light_channel(b, t);                      /* PinLight */
    case 24: return (b + t >= 1.0f) ? 1.0f : 0.0f;                /* HardMix */
    default: return t; /* unknown ids decode to Normal: layers.rs:156-185 from_u8 */
    }
}

/* canvas_state.rs:1246-1422 */
void pfxo_blend_pixel(const uint8_t base[4], const uint8_t top[4], int mode, float opacity, uint8_t out[4])
{
    if (mode < 0 || mode > 24) mode = 0;
    if (top[3] == 0) { /* :1253 */
        out[0] = base[0]; out[1] = base[1]; out[2] = base[2]; out[3] = base[3];
        return;
    }
    if (mode == 0 && opacity >= 1.0f && top[3] == 255) { /* :1258 */
        out[0] = top[0]; out[1] = top[1]; out[2] = top[2]; out[3] = top[3];
        return;
    }
    opacity = rs_clampf(opacity, 0.0f, 1.0f); /* :1262 */

    float base_r = (float)base[0] / 255.0f, base_g = (float)base[1] / 255.0f;
    float base_b = (float)base[2] / 255.0f, base_a = (float)base[3] / 255.0f;
    float top_r = (float)top[0] / 255.0f, top_g = (float)top[1] / 255.0f;
    float top_b = (float)top[2] / 255.0f;
    float top_a = ((float)top[3] / 255.0f) * opacity;

    if (mode == 14) { /* Overwrite :1275 */
        out[0] = rs_f32_as_u8(top_r * 255.0f);
        out[1] = rs_f32_as_u8(top_g * 255.0f);
        out[2] = rs_f32_as_u8(top_b * 255.0f);
        out[3] = rs_f32_as_u8(top_a * 255.0f);
        return;
    }
    if (mode == 13) { /* Xor :1283 */
        float xor_a = base_a * (1.0f - top_a) + top_a * (1.0f - base_a);
        if (xor_a == 0.0f) { out[0] = out[1] = out[2] = out[3] = 0; return; }
        float xr = (base_r * base_a * (1.0f - top_a) + top_r * top_a * (1.0f - base_a)) / xor_a;
        float xg = (base_g * base_a * (1.0f - top_a) + top_g * top_a * (1.0f - base_a)) / xor_a;
        float xb = (base_b * base_a * (1.0f - top_a) + top_b * top_a * (1.0f - base_a)) / xor_a;
        out[0] = rs_f32_as_u8(rs_clampf(xr * 255.0f, 0.0f, 255.0f));
        out[1] = rs_f32_as_u8(rs_clampf(xg * 255.0f, 0.0f, 255.0f));
        out[2] = rs_f32_as_u8(rs_clampf(xb * 255.0f, 0.0f, 255.0f));
        out[3] = rs_f32_as_u8(rs_clampf(xor_a * 255.0f, 0.0f, 255.0f));
        return;
    }

    float r = blend_fn(mode, base_r, top_r);
    float g = blend_fn(mode, base_g, top_g);
    float b = blend_fn(mode, base_b, top_b);

    float out_a = top_a + base_a * (1.0f - top_a); /* :1407 */
    if (out_a == 0.0f) { out[0] = out[1] = out[2] = out[3] = 0; return; }
    float out_r = (r * top_a + base_r * base_a * (1.0f - top_a)) / out_a;
    float out_g = (g * top_a + base_g * base_a * (1.0f - top_a)) / out_a;
    float out_b = (b * top_a + base_b * base_a * (1.0f - top_a)) / out_a;
    out[0] = rs_f32_as_u8(rs_clampf(out_r * 255.0f, 0.0f, 255.0f));
    out[1] = rs_f32_as_u8(rs_clampf(out_g * 255.0f, 0.0f, 255.0f));
    out[2] = rs_f32_as_u8(rs_clampf(out_b * 255.0f, 0.0f, 255.0f));
    out[3] = rs_f32_as_u8(rs_clampf(out_a * 255.0f, 0.0f, 255.0f));
}

/* layers.rs:276-312 */
static void adj_apply_to_pixel(const pfxo_layer* L, const uint8_t p[4], uint8_t o[4])
{
    uint8_t r = p[0], g = p[1], b = p[2], a = p[3];
    switch (L->kind) {
    case PFXO_ADJ_EXPOSURE: {
        float gain = powf(2.0f, L->adj[0]);
        o[0] = rs_f32_as_u8(rs_clampf((float)r * gain, 0.0f, 255.0f));
        o[1] = rs_f32_as_u8(rs_clampf((float)g * gain, 0.0f, 255.0f));
        o[2] = rs_f32_as_u8(rs_clampf((float)b * gain, 0.0f, 255.0f));
        o[3] = a;
        break;
    }
    case PFXO_ADJ_BRIGHTNESS_CONTRAST: {
        float brightness = L->adj[0], contrast = L->adj[1];
        float factor = (259.0f * (contrast + 255.0f)) / (255.0f * (259.0f - contrast));
        o[0] = rs_f32_as_u8(rs_clampf(factor * ((float)r + brightness - 128.0f) + 128.0f, 0.0f, 255.0f));
        o[1] = rs_f32_as_u8(rs_clampf(factor * ((float)g + brightness - 128.0f) + 128.0f, 0.0f, 255.0f));
        o[2] = rs_f32_as_u8(rs_clampf(factor * ((float)b + brightness - 128.0f) + 128.0f, 0.0f, 255.0f));
        o[3] = a;
        break;
    }
    case PFXO_ADJ_INVERT:
        o[0] = 255 - r; o[1] = 255 - g; o[2] = 255 - b; o[3] = a;
        break;
    case PFXO_ADJ_CHANNEL_MIXER: {
        float s0 = (float)r, s1 = (float)g, s2 = (float)b, s3 = (float)a;
        for (int c = 0; c < 4; ++c) {
            const float* m = &L->adj[c * 4];
            o[c] = rs_f32_as_u8(rs_clampf(s0 * m[0] + s1 * m[1] + s2 * m[2] + s3 * m[3], 0.0f, 255.0f));
        }
        break;
    }
    default:
        o[0] = r; o[1] = g; o[2] = b; o[3] = a;
    }
}

/* layers.rs:314-325 */
static void adj_apply_with_opacity(const pfxo_layer* L, uint8_t p[4])
{
    uint8_t adj[4];
    adj_apply_to_pixel(L, p, adj);
    float t = rs_clampf(L->opacity, 0.0f, 1.0f);
    float inv = 1.0f - t;
    for (int c = 0; c < 4; ++c)
        p[c] = rs_f32_as_u8(roundf((float)p[c] * inv + (float)adj[c] * t));
}

/* canvas_state.rs:505-698 with viewport=None and preview_layer=None */
void pfxo_composite(const pfxo_layer* layers, int n_layers, uint32_t w, uint32_t h, uint8_t* dst, int threads)
{
    pfxo_composite_preview(layers, n_layers, w, h, NULL, dst, threads);
}

/* :621-658: the tool preview is folded into the active layer's pixel before masking and compositing */
static void apply_preview(uint8_t top[4], const uint8_t pp[4], const pfxo_preview* pv)
{
    if (pv->replaces_layer) { memcpy(top, pp, 4); return; }
    if (pp[3] == 0) return;
    int mode = pv->blend_mode > 24 ? 0 : pv->blend_mode;
    if (pv->is_eraser) {
        float mask_strength = (float)pp[3] / 255.0f, current_a = (float)top[3] / 255.0f;
        float new_a = fmaxf(current_a * (1.0f - mask_strength), 0.0f);
        top[3] = rs_f32_as_u8(new_a * 255.0f);
    } else if (mode == 14 /* Overwrite */ || mode == 13 /* Xor */) {
        uint8_t ow[4];
        pfxo_blend_pixel(top, pp, mode, 1.0f, ow);
        float cov = (float)pp[3] / 255.0f, inv = 1.0f - cov;
        for (int c = 0; c < 4; ++c) top[c] = rs_f32_as_u8((float)top[c] * inv + (float)ow[c] * cov + 0.5f);
    } else {
        uint8_t o[4];
        pfxo_blend_pixel(top, pp, mode, 1.0f, o);
        memcpy(top, o, 4);
    }
}

/* Reference-faithful tail (canvas_state.rs:565-695): the rayon closure returns each chunk's pixels, `.collect()` gathers them and
 * ONE thread then writes them back with a put_pixel per pixel.  0 (default) = write-back inside the parallel loop (the fair,
 * fully parallel variant the parity tests use: same bytes), 1 = collect + serial write-back, for bench.py's `cpu_baseline.faithful`. */
static int g_serial_writeback = 0;
void pfxo_set_serial_writeback(int on) { g_serial_writeback = on; }

/* canvas_state.rs:505-698 with viewport=None */
void pfxo_composite_preview(const pfxo_layer* layers, int n_layers, uint32_t w, uint32_t h, const pfxo_preview* pv, uint8_t* dst, int threads)
{
    const uint32_t cxn = (w + PFXO_CHUNK - 1) / PFXO_CHUNK, cyn = (h + PFXO_CHUNK - 1) / PFXO_CHUNK;
    const size_t n_chunks = (size_t)cxn * cyn;
    memset(dst, 0, (size_t)w * h * 4); /* :506 RgbaImage::new -> zeroed */
    if (pv && !pv->pixels) pv = NULL;

    /* per-layer chunk population = TiledImage::from_rgba_image's has_content (tiled_image.rs:81-95) */
    uint8_t** pop = (uint8_t**)calloc((size_t)n_layers, sizeof(uint8_t*));
    uint8_t* active = (uint8_t*)calloc(n_chunks, 1);
    uint8_t* pv_pop = NULL;
    for (int li = 0; li < n_layers; ++li) {
        if (!layers[li].pixels) continue;
        pop[li] = (uint8_t*)malloc(n_chunks);
        pfxo_chunk_populated(layers[li].pixels, w, h, pop[li]);
        if (layers[li].visible) /* :530-540 active_chunks = union over visible layers */
            for (size_t i = 0; i < n_chunks; ++i) active[i] |= pop[li][i];
    }
    if (pv) { /* :541-548 the preview's chunk keys join the active set */
        pv_pop = (uint8_t*)malloc(n_chunks);
        if (pv->chunk_present) memcpy(pv_pop, pv->chunk_present, n_chunks);
        else pfxo_chunk_populated(pv->pixels, w, h, pv_pop);
        for (size_t i = 0; i < n_chunks; ++i) active[i] |= (uint8_t)(pv_pop[i] != 0);
    }

    uint8_t* collected = g_serial_writeback ? (uint8_t*)malloc(n_chunks * (size_t)PFXO_CHUNK * PFXO_CHUNK * 4) : NULL; /* chunk_results, :565-684 */
    o_set_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
    for (long ci = 0; ci < (long)n_chunks; ++ci) { /* :565 par_iter over active chunks */
        if (!active[ci]) continue;
        uint32_t cx = (uint32_t)ci % cxn, cy = (uint32_t)ci / cxn;
        uint32_t bx = cx * PFXO_CHUNK, by = cy * PFXO_CHUNK;
        uint32_t cw = (w - bx < PFXO_CHUNK) ? w - bx : PFXO_CHUNK;
        uint32_t ch = (h - by < PFXO_CHUNK) ? h - by : PFXO_CHUNK;
        uint8_t acc[PFXO_CHUNK * PFXO_CHUNK * 4]; /* :573 */
        memset(acc, 0, sizeof acc);

        for (int li = 0; li < n_layers; ++li) { /* :575 */
            const pfxo_layer* L = &layers[li];
            if (!L->visible) continue;
            if (L->kind != PFXO_LAYER_RASTER) { /* :579-584 */
                for (uint32_t i = 0; i < cw * ch; ++i) adj_apply_with_opacity(L, &acc[i * 4]);
                continue;
            }
            int has_preview = pv && li == pv->active_layer && pv_pop[ci]; /* :593-597 */
            int has_layer = L->pixels && pop[li][ci];
            if (!has_layer && !has_preview) continue; /* :600 */
            int mode = L->blend_mode > 24 ? 0 : L->blend_mode;
            int opaque_overwrite = (mode == 0) && (L->opacity >= 1.0f) && !has_preview; /* :605 */
            for (uint32_t ly = 0; ly < ch; ++ly) {
                for (uint32_t lx = 0; lx < cw; ++lx) {
                    size_t gi = (size_t)(by + ly) * w + (bx + lx);
                    uint8_t top[4] = {0, 0, 0, 0};
                    if (has_layer) memcpy(top, &L->pixels[gi * 4], 4);
                    if (has_preview) apply_preview(top, &pv->pixels[gi * 4], pv);
                    if (L->mask) { /* :660-665 */
                        uint32_t conceal = L->mask[gi];
                        if (conceal > 0) top[3] = (uint8_t)(((uint32_t)top[3] * (255u - conceal)) / 255u);
                    }
                    uint8_t* px = &acc[(ly * cw + lx) * 4];
                    if (opaque_overwrite && top[3] == 255) { /* :667 */
                        memcpy(px, top, 4);
                    } else {
                        uint8_t o[4];
                        pfxo_blend_pixel(px, top, mode, L->opacity, o);
                        memcpy(px, o, 4);
                    }
                }
            }
        }
        if (collected) { memcpy(collected + (size_t)ci * sizeof acc, acc, sizeof acc); continue; }
        for (uint32_t ly = 0; ly < ch; ++ly) /* :686-695 (parallel here; the reference's tail is serial) */
            memcpy(&dst[((size_t)(by + ly) * w + bx) * 4], &acc[(size_t)ly * cw * 4], (size_t)cw * 4);
    }
    if (collected) { /* :686-695: serial, one put_pixel per pixel */
        for (size_t ci = 0; ci < n_chunks; ++ci) {
            if (!active[ci]) continue;
            const uint32_t cx = (uint32_t)(ci % cxn), cy = (uint32_t)(ci / cxn), bx = cx * PFXO_CHUNK, by = cy * PFXO_CHUNK;
            const uint32_t cw = (w - bx < PFXO_CHUNK) ? w - bx : PFXO_CHUNK, ch = (h - by < PFXO_CHUNK) ? h - by : PFXO_CHUNK;
            const uint8_t* px = collected + ci * (size_t)PFXO_CHUNK * PFXO_CHUNK * 4;
            for (uint32_t ly = 0; ly < ch; ++ly)
                for (uint32_t lx = 0; lx < cw; ++lx) {
                    uint8_t* d = &dst[((size_t)(by + ly) * w + bx + lx) * 4];
                    const uint8_t* q = &px[((size_t)ly * cw + lx) * 4];
                    d[0] = q[0]; d[1] = q[1]; d[2] = q[2]; d[3] = q[3];
                }
        }
        free(collected);
    }

    for (int li = 0; li < n_layers; ++li) free(pop[li]);
    free(pop);
    free(active);
    free(pv_pop);
}

void pfxo_flatten_stack(const uint8_t* stack, int n_layers, const uint8_t* modes, const float* opacities,
                        uint32_t w, uint32_t h, uint8_t* dst, int threads)
{
    pfxo_layer* L = (pfxo_layer*)calloc((size_t)n_layers, sizeof(pfxo_layer));
    for (int i = 0; i < n_layers; ++i) {
        L[i].pixels = stack + (size_t)i * w * h * 4;
        L[i].opacity = opacities[i];
        L[i].blend_mode = modes[i];
        L[i].visible = 1;
        L[i].kind = PFXO_LAYER_RASTER;
    }
    pfxo_composite(L, n_layers, w, h, dst, threads);
    free(L);
}
