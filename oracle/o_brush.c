/*
 * o_brush.c — oracle restatement of the brush stamp loop (round tip and image tips, with scatter / colour jitter).
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:27-50    rebuild_brush_lut
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:54-82    compute_brush_alpha
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:135-400  draw_circle_no_dirty
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:404-528  rebuild_tip_mask
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:533-760  draw_image_tip_no_dirty
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:846-857  stamp_hash
 * The 23 tool goldens pin the round tip; scatter, jitter and image tips have no reference golden (parity unpinned beyond the
 * restatement; product and oracle are compared bit for bit).
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:762-835  draw_line_no_dirty
 *   src/ui/panels/tools/behavior/raster/bezier_commit.rs:103-225 commit_bezier_to_layer / commit_eraser_to_layer
 *   src/ui/panels/tools/state.rs:133-157                         ToolProperties::default
 * The target ("preview") image is a dense w*h RGBA8 buffer; TiledImage chunk creation is invisible to results.
 */
#include "o_common.h"


/* :54-82 */
float pfxo_brush_alpha(float dist, float radius, float hardness, int anti_aliased)
{
    if (radius <= 0.0f) return 0.0f;
    float safe_hardness = rs_clampf(hardness, 0.0f, 1.0f);
    float t = rs_clampf(dist / radius, 0.0f, 1.0f);
    float falloff = t * t * (3.0f - 2.0f * t);
    float material_alpha = 1.0f + (safe_hardness - 1.0f) * falloff;
    float coverage;
    if (anti_aliased) {
        float edge0 = radius + 0.5f, edge1 = radius - 0.5f;
        if (dist <= edge1) coverage = 1.0f;
        else if (dist >= edge0) coverage = 0.0f;
        else {
            float x = rs_clampf((dist - edge0) / (edge1 - edge0), 0.0f, 1.0f);
            coverage = x * x * (3.0f - 2.0f * x);
        }
    } else coverage = (dist <= radius) ? 1.0f : 0.0f;
    return material_alpha * coverage;
}

/* :27-50 */
void pfxo_brush_lut(float size, float hardness, int anti_aliased, uint8_t lut[256])
{
    float radius = size / 2.0f;
    if (radius < 0.001f) { memset(lut, 0, 256); return; }
    for (int i = 0; i < 256; ++i) {
        float t_sq = (float)i / 255.0f;
        float dist = sqrtf(t_sq) * radius;
        float alpha = pfxo_brush_alpha(dist, radius, hardness, anti_aliased);
        lut[i] = rs_f32_as_u8(fminf(roundf(alpha * 255.0f), 255.0f));
    }
}

/* :846-857 */
static uint32_t stamp_hash(float x, float y, uint32_t counter)
{
    uint32_t ix = rs_f32_as_u32(x * 100.0f), iy = rs_f32_as_u32(y * 100.0f);
    uint32_t h = ix * 374761393u + iy * 668265263u + counter * 1013904223u;
    h ^= h >> 13;
    h *= 1274126177u;
    h ^= h >> 16;
    return h;
}

/* :179-193 / :552-565 scatter: the stamp centre moves by up to scatter * diameter */
static void scatter_pos(const pfxo_brush* b, const pfxo_brush_dyn* d, float px, float py, float* cx, float* cy)
{
    *cx = px; *cy = py;
    if (d && d->scatter > 0.01f) {
        float diam = b->size; /* pressure_size() without pen pressure */
        float h1 = (float)stamp_hash(px, py, d->stamp_counter) / 4294967295.0f;
        float h2 = (float)stamp_hash(py, px, d->stamp_counter + 99991u) / 4294967295.0f;
        *cx = px + (h1 * 2.0f - 1.0f) * d->scatter * diam;
        *cy = py + (h2 * 2.0f - 1.0f) * d->scatter * diam;
    }
}

/* :223-256 / :599-632 per-stamp colour with HSL jitter */
static void stamp_color(const pfxo_brush* b, const pfxo_brush_dyn* d, float px, float py, uint8_t rgb[3])
{
    float src_r = b->color[0], src_g = b->color[1], src_b = b->color[2];
    rgb[0] = rs_f32_as_u8(src_r * 255.0f); rgb[1] = rs_f32_as_u8(src_g * 255.0f); rgb[2] = rs_f32_as_u8(src_b * 255.0f);
    if (!d || !(d->hue_jitter > 0.01f || d->brightness_jitter > 0.01f)) return;
    float h, s, l, nr, ng, nb;
    pfxo_rgb_to_hsl(src_r, src_g, src_b, &h, &s, &l);
    if (d->hue_jitter > 0.01f) {
        float hh = (float)stamp_hash(px + 0.1f, py + 0.2f, d->stamp_counter + 777u) / 4294967295.0f;
        float v = h + (hh * 2.0f - 1.0f) * d->hue_jitter * 0.5f;
        h = v - truncf(v); /* fract() */
        if (h < 0.0f) h += 1.0f;
    }
    if (d->brightness_jitter > 0.01f) {
        float bh = (float)stamp_hash(px + 0.3f, py + 0.4f, d->stamp_counter + 555u) / 4294967295.0f;
        l = rs_clampf(l + (bh * 2.0f - 1.0f) * d->brightness_jitter * 0.5f, 0.0f, 1.0f);
    }
    pfxo_hsl_to_rgb(h, s, l, &nr, &ng, &nb);
    rgb[0] = rs_f32_as_u8(nr * 255.0f); rgb[1] = rs_f32_as_u8(ng * 255.0f); rgb[2] = rs_f32_as_u8(nb * 255.0f);
}

/* rebuild_tip_mask :404-528: bilinear rescale of the tip's source mask to ceil(size), hardness contrast, box anti-alias passes.
 * Returns the side of the square written to `out` (capacity >= ceil(size)^2), 0 when there is no source. */
uint32_t pfxo_brush_tip_rescale(const uint8_t* src, uint32_t src_size, float brush_size, float hardness, uint8_t* out)
{
    if (!src || src_size == 0) return 0;
    uint32_t dst_size = rs_f32_as_u32(ceilf(brush_size));
    if (dst_size < 1) dst_size = 1;
    float scale = (float)src_size / (float)dst_size;
    for (uint32_t dy = 0; dy < dst_size; ++dy)
        for (uint32_t dx = 0; dx < dst_size; ++dx) {
            float sx = (float)dx * scale, sy = (float)dy * scale;
            uint32_t sx0 = rs_f32_as_u32(floorf(sx)), sy0 = rs_f32_as_u32(floorf(sy));
            uint32_t sx1 = sx0 + 1 < src_size - 1 ? sx0 + 1 : src_size - 1, sy1 = sy0 + 1 < src_size - 1 ? sy0 + 1 : src_size - 1;
            float fx = sx - (float)sx0, fy = sy - (float)sy0;
            float v00 = src[sy0 * src_size + sx0], v10 = src[sy0 * src_size + sx1], v01 = src[sy1 * src_size + sx0], v11 = src[sy1 * src_size + sx1];
            float top = v00 * (1.0f - fx) + v10 * fx, bot = v01 * (1.0f - fx) + v11 * fx;
            float val = top * (1.0f - fy) + bot * fy;
            out[dy * dst_size + dx] = rs_f32_as_u8(fminf(roundf(val), 255.0f));
        }
    if (hardness < 0.99f) {
        float threshold = (1.0f - hardness) * 0.6f, range = 1.0f - threshold;
        for (uint32_t i = 0; i < dst_size * dst_size; ++i) {
            float norm = (float)out[i] / 255.0f;
            float adj = rs_clampf((norm - threshold) / range, 0.0f, 1.0f);
            out[i] = rs_f32_as_u8(roundf(adj * 255.0f));
        }
    }
    if (dst_size < src_size && dst_size >= 3) {
        float ratio = (float)src_size / (float)dst_size;
        int passes = ratio > 4.0f ? 2 : (ratio > 1.5f ? 1 : 0);
        uint8_t* tmp = (uint8_t*)malloc((size_t)dst_size * dst_size);
        for (int p = 0; p < passes; ++p) {
            memcpy(tmp, out, (size_t)dst_size * dst_size);
            for (uint32_t y = 0; y < dst_size; ++y)
                for (uint32_t x = 0; x < dst_size; ++x) {
                    uint32_t sum = out[y * dst_size + x], count = 1;
                    if (x > 0) { sum += out[y * dst_size + x - 1]; ++count; }
                    if (x + 1 < dst_size) { sum += out[y * dst_size + x + 1]; ++count; }
                    tmp[y * dst_size + x] = (uint8_t)(sum / count);
                }
            for (uint32_t y = 0; y < dst_size; ++y)
                for (uint32_t x = 0; x < dst_size; ++x) {
                    uint32_t sum = tmp[y * dst_size + x], count = 1;
                    if (y > 0) { sum += tmp[(y - 1) * dst_size + x]; ++count; }
                    if (y + 1 < dst_size) { sum += tmp[(y + 1) * dst_size + x]; ++count; }
                    out[y * dst_size + x] = (uint8_t)(sum / count);
                }
        }
        free(tmp);
    }
    return dst_size;
}

/* draw_image_tip_no_dirty :533-760 */
static void stamp_image_tip(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, const pfxo_brush_dyn* d, float px, float py,
                            float rotation_deg, const uint8_t* selection)
{
    uint32_t mask_size = d->tip_mask_size;
    const uint8_t* mask = d->tip_mask;
    if (mask_size == 0 || !mask) return;
    float cx, cy;
    scatter_pos(b, d, px, py, &cx, &cy);
    float half = (float)mask_size / 2.0f;
    int rotated = fabsf(rotation_deg) > 0.01f;
    float cos_a = 1.0f, sin_a = 0.0f;
    if (rotated) {
        float rad = -(rotation_deg * (3.14159265358979323846f / 180.0f));
        cos_a = cosf(rad); sin_a = sinf(rad);
    }
    float effective_half = rotated ? half * 1.41421356237309504880f : half;
    uint32_t min_x = rs_f32_as_u32(fmaxf(cx - effective_half, 0.0f)), min_y = rs_f32_as_u32(fmaxf(cy - effective_half, 0.0f));
    uint32_t wm1 = width ? width - 1 : 0, hm1 = height ? height - 1 : 0;
    uint32_t max_x = rs_f32_as_u32(cx + effective_half), max_y = rs_f32_as_u32(cy + effective_half);
    if (max_x > wm1) max_x = wm1;
    if (max_y > hm1) max_y = hm1;
    if (min_x > max_x || min_y > max_y) return;
    uint8_t rgb[3];
    stamp_color(b, d, px, py, rgb);
    float src_a = b->color[3];
    for (uint32_t gy = min_y; gy <= max_y; ++gy)
        for (uint32_t gx = min_x; gx <= max_x; ++gx) {
            if (selection && selection[(size_t)gy * width + gx] == 0) continue;
            float rel_x = (float)gx - cx, rel_y = (float)gy - cy;
            uint8_t geom_u8;
            if (rotated) {
                float rot_x = rel_x * cos_a - rel_y * sin_a + half, rot_y = rel_x * sin_a + rel_y * cos_a + half;
                if (rot_x < -0.5f || rot_y < -0.5f || rot_x >= (float)mask_size - 0.5f || rot_y >= (float)mask_size - 0.5f) continue;
                float sx = fmaxf(rot_x, 0.0f), sy = fmaxf(rot_y, 0.0f);
                uint32_t sx0 = rs_f32_as_u32(floorf(sx)), sy0 = rs_f32_as_u32(floorf(sy));
                uint32_t sx1 = sx0 + 1 < mask_size - 1 ? sx0 + 1 : mask_size - 1, sy1 = sy0 + 1 < mask_size - 1 ? sy0 + 1 : mask_size - 1;
                float fx = sx - (float)sx0, fy = sy - (float)sy0;
                float v00 = mask[sy0 * mask_size + sx0], v10 = mask[sy0 * mask_size + sx1], v01 = mask[sy1 * mask_size + sx0], v11 = mask[sy1 * mask_size + sx1];
                float top = v00 * (1.0f - fx) + v10 * fx, bot = v01 * (1.0f - fx) + v11 * fx;
                geom_u8 = rs_f32_as_u8(fminf(roundf(top * (1.0f - fy) + bot * fy), 255.0f));
            } else {
                int32_t mx = rs_f32_as_i32(roundf(rel_x + half)), my = rs_f32_as_i32(roundf(rel_y + half));
                if (mx < 0 || my < 0 || mx >= (int32_t)mask_size || my >= (int32_t)mask_size) continue;
                geom_u8 = mask[(uint32_t)my * mask_size + (uint32_t)mx];
            }
            if (geom_u8 == 0) continue;
            float geom_alpha = (float)geom_u8 / 255.0f;
            uint8_t* p = img + ((size_t)gy * width + gx) * 4;
            if (b->is_eraser) {
                float erase_strength = geom_alpha * src_a * b->flow;
                if (erase_strength < 0.01f) continue;
                if (erase_strength > (float)p[3] / 255.0f) { p[0] = p[1] = p[2] = 0; p[3] = rs_f32_as_u8(erase_strength * 255.0f); }
            } else {
                uint8_t a8 = rs_f32_as_u8(geom_alpha * src_a * b->flow * 255.0f);
                if (a8 >= p[3]) { p[0] = rgb[0]; p[1] = rgb[1]; p[2] = rgb[2]; p[3] = a8; }
            }
        }
}

static void stamp_circle(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, float cx, float cy, const uint8_t rgb[3],
                         const uint8_t* selection);

/* draw_circle_no_dirty :135-400 with the brush dynamics (scatter, colour jitter, image tips) */
void pfxo_brush_stamp_ex(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, const pfxo_brush_dyn* d, float px, float py,
                         const uint8_t* selection)
{
    if (d && d->tip_mask) { /* :148-177 */
        float rotation_deg = d->tip_rotation;
        if (d->tip_random_rotation) {
            float lo = d->tip_rotation_lo, range = d->tip_rotation_hi - lo;
            rotation_deg = fabsf(range) < 0.01f ? lo : lo + (float)(stamp_hash(px, py, d->stamp_counter) % 10000u) / 10000.0f * range;
        }
        stamp_image_tip(img, width, height, b, d, px, py, rotation_deg, selection);
        return;
    }
    float cx, cy;
    uint8_t rgb[3];
    scatter_pos(b, d, px, py, &cx, &cy);
    stamp_color(b, d, px, py, rgb);
    stamp_circle(img, width, height, b, cx, cy, rgb, selection);
}

void pfxo_brush_stamp(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, float cx, float cy, const uint8_t* selection)
{
    pfxo_brush_stamp_ex(img, width, height, b, NULL, cx, cy, selection);
}

/* :194-400 */
static void stamp_circle(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, float cx, float cy, const uint8_t rgb[3],
                         const uint8_t* selection)
{
    float radius = b->size / 2.0f;
    float radius_sq = radius * radius;
    if (radius_sq < 0.001f) return;
    float draw_radius = b->anti_aliased ? radius + 0.5f : radius;
    float draw_radius_sq = draw_radius * draw_radius;
    int use_direct_alpha = draw_radius > radius;
    float inv_radius_sq = 1.0f / radius_sq;

    uint32_t min_x = rs_f32_as_u32(fmaxf(floorf(cx - draw_radius), 0.0f));
    uint32_t max_x = rs_f32_as_u32(ceilf(cx + draw_radius));
    uint32_t wm1 = width ? width - 1 : 0, hm1 = height ? height - 1 : 0;
    if (max_x > wm1) max_x = wm1;
    uint32_t min_y = rs_f32_as_u32(fmaxf(floorf(cy - draw_radius), 0.0f));
    uint32_t max_y = rs_f32_as_u32(ceilf(cy + draw_radius));
    if (max_y > hm1) max_y = hm1;
    if (min_x > max_x || min_y > max_y) return;

    float src_a = b->color[3];
    uint8_t src_r8 = rgb[0], src_g8 = rgb[1], src_b8 = rgb[2]; /* :223-256: truncated colour bytes, jittered per stamp */
    uint8_t lut[256];
    if (!use_direct_alpha) pfxo_brush_lut(b->size, b->hardness, b->anti_aliased, lut);

    for (uint32_t gy = min_y; gy <= max_y; ++gy) {
        float dy = (float)gy - cy;
        float dy_sq = dy * dy;
        for (uint32_t gx = min_x; gx <= max_x; ++gx) {
            if (selection && selection[(size_t)gy * width + gx] == 0) continue;
            float dx = (float)gx - cx;
            float dist_sq = dx * dx + dy_sq;
            if (dist_sq > draw_radius_sq) continue;
            uint8_t geom_alpha_u8;
            if (use_direct_alpha)
                geom_alpha_u8 = rs_f32_as_u8(fminf(
                    roundf(pfxo_brush_alpha(sqrtf(dist_sq), radius, b->hardness, b->anti_aliased) * 255.0f), 255.0f));
            else
                geom_alpha_u8 = lut[rs_f32_as_u32(fminf(dist_sq * inv_radius_sq * 255.0f, 255.0f))];
            if (geom_alpha_u8 == 0) continue;
            float geom_alpha = (float)geom_alpha_u8 / 255.0f;
            uint8_t* px = img + ((size_t)gy * width + gx) * 4;
            if (b->is_eraser) {
                float erase_strength = geom_alpha * src_a * b->flow;
                if (erase_strength < 0.01f) continue;
                float old_mask = (float)px[3] / 255.0f;
                if (erase_strength > old_mask) {
                    px[0] = 0; px[1] = 0; px[2] = 0;
                    px[3] = rs_f32_as_u8(erase_strength * 255.0f);
                }
            } else {
                float brush_alpha = geom_alpha * src_a * b->flow;
                if (brush_alpha < 0.01f) continue;
                if (b->mode == PFXO_BRUSH_NORMAL) {
                    uint8_t brush_alpha_u8 = rs_f32_as_u8(brush_alpha * 255.0f);
                    if (brush_alpha_u8 >= px[3]) { /* max-alpha stamping :367 */
                        px[0] = src_r8; px[1] = src_g8; px[2] = src_b8; px[3] = brush_alpha_u8;
                    }
                } else {
                    float old_r = (float)px[0] / 255.0f, old_g = (float)px[1] / 255.0f, old_b = (float)px[2] / 255.0f;
                    float h, s, l, nr, ng, nb;
                    pfxo_rgb_to_hsl(old_r, old_g, old_b, &h, &s, &l);
                    float strength = brush_alpha * 0.5f;
                    if (b->mode == PFXO_BRUSH_DODGE) l = rs_clampf(l + strength, 0.0f, 1.0f);
                    else if (b->mode == PFXO_BRUSH_BURN) l = rs_clampf(l - strength, 0.0f, 1.0f);
                    else if (b->mode == PFXO_BRUSH_SPONGE) s = rs_clampf(s - strength, 0.0f, 1.0f);
                    pfxo_hsl_to_rgb(h, s, l, &nr, &ng, &nb);
                    px[0] = rs_f32_as_u8(nr * 255.0f);
                    px[1] = rs_f32_as_u8(ng * 255.0f);
                    px[2] = rs_f32_as_u8(nb * 255.0f);
                }
            }
        }
    }
}

/* :762-835 (circle tip: step = 1.0) */
int pfxo_brush_line_points(float x0, float y0, float x1, float y1, uint32_t width, uint32_t height,
                           float* out_xy, int cap)
{
    float dx = x1 - x0, dy = y1 - y0;
    float distance = sqrtf(dx * dx + dy * dy);
    int n = 0;
    if (distance < 0.1f) {
        if (x0 >= 0.0f && rs_f32_as_u32(x0) < width && y0 >= 0.0f && rs_f32_as_u32(y0) < height) {
            if (n < cap) { out_xy[0] = x0; out_xy[1] = y0; }
            n = 1;
        }
        return n;
    }
    float step = 1.0f;
    size_t steps = (size_t)rs_f32_as_u32(ceilf(distance / step));
    for (size_t i = 0; i <= steps; ++i) {
        float t = (float)i / (float)steps;
        float x = x0 + dx * t, y = y0 + dy * t;
        if (x >= 0.0f && rs_f32_as_u32(x) < width && y >= 0.0f && rs_f32_as_u32(y) < height) {
            if (n < cap) { out_xy[n * 2] = x; out_xy[n * 2 + 1] = y; }
            ++n;
        }
    }
    return n;
}

void pfxo_brush_line(uint8_t* img, uint32_t w, uint32_t h, const pfxo_brush* b,
                     float x0, float y0, float x1, float y1, const uint8_t* selection)
{
    int n = pfxo_brush_line_points(x0, y0, x1, y1, w, h, NULL, 0);
    if (n <= 0) return;
    float* pts = (float*)malloc(sizeof(float) * 2 * (size_t)n);
    pfxo_brush_line_points(x0, y0, x1, y1, w, h, pts, n);
    for (int i = 0; i < n; ++i) pfxo_brush_stamp(img, w, h, b, pts[i * 2], pts[i * 2 + 1], selection);
    free(pts);
}

/* bezier_commit.rs:103-161 */
void pfxo_brush_commit(uint8_t* layer, const uint8_t* preview, uint32_t w, uint32_t h, int mode, const uint8_t* selection)
{
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (selection && selection[i] == 0) continue;
        const uint8_t* pp = preview + i * 4;
        if (pp[3] > 0) {
            uint8_t o[4];
            pfxo_blend_pixel(layer + i * 4, pp, mode, 1.0f, o);
            memcpy(layer + i * 4, o, 4);
        }
    }
}

/* bezier_commit.rs:166-225 */
void pfxo_eraser_commit(uint8_t* layer, const uint8_t* preview, uint32_t w, uint32_t h, const uint8_t* selection)
{
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (selection && selection[i] == 0) continue;
        const uint8_t* mp = preview + i * 4;
        if (mp[3] > 0) {
            float mask_strength = (float)mp[3] / 255.0f;
            float current_a = (float)layer[i * 4 + 3] / 255.0f;
            float new_a = fmaxf(current_a * (1.0f - mask_strength), 0.0f);
            layer[i * 4 + 3] = rs_f32_as_u8(new_a * 255.0f);
        }
    }
}
