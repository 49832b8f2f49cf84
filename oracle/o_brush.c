/*
 * o_brush.c — oracle restatement of the round-tip brush stamp loop.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:27-50    rebuild_brush_lut
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:54-82    compute_brush_alpha
 *   src/ui/panels/tools/behavior/raster/brush_render.rs:135-400  draw_circle_no_dirty (circle tip, no scatter/jitter)
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

/* :135-400 */
void pfxo_brush_stamp(uint8_t* img, uint32_t width, uint32_t height, const pfxo_brush* b, float cx, float cy,
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

    float src_r = b->color[0], src_g = b->color[1], src_b = b->color[2], src_a = b->color[3];
    uint8_t src_r8 = rs_f32_as_u8(src_r * 255.0f), src_g8 = rs_f32_as_u8(src_g * 255.0f),
            src_b8 = rs_f32_as_u8(src_b * 255.0f);
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
