// k_brush.hip — the round-tip brush stamp loop over a preview image.
//
// Reference: draw_circle_no_dirty src/ui/panels/tools/behavior/raster/brush_render.rs:135-400 (circle tip, no
// scatter / colour jitter), rebuild_brush_lut :27-50, compute_brush_alpha :54-82.
// The reference stamps serially, stamp after stamp, each stamp looping over its bounding box.  Stamps only ever
// read and write the pixel they are positioned on, so the loop nest is interchanged: one lane per pixel of the
// stroke's bounding box walks the stamp list IN ORDER.  That keeps every order-dependent mode (max-alpha Normal
// with `>=` ties, eraser `>`, Dodge/Burn/Sponge read-modify-write) bit-identical while exposing width x height
// parallelism.  Stamp centres are wave-uniform (scalar loads).
#include "k_brush_math.h"
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

struct hsl3 { float h, s, l; };
// src/ops/adjustments.rs:944-1012 (same restatement as k_pointwise.hip)
PFX_DEV hsl3 rgb_to_hsl(float r, float g, float b)
{
    const float mx = __builtin_fmaxf(__builtin_fmaxf(r, g), b), mn = __builtin_fminf(__builtin_fminf(r, g), b);
    const float l = (mx + mn) / 2.0f;
    if (__builtin_fabsf(mx - mn) < 1e-6f) return {0.0f, 0.0f, l};
    const float d = mx - mn;
    const float s = (l > 0.5f) ? d / (2.0f - mx - mn) : d / (mx + mn);
    float h;
    if (__builtin_fabsf(mx - r) < 1e-6f) { h = (g - b) / d; if (h < 0.0f) h += 6.0f; h = h / 6.0f; }
    else if (__builtin_fabsf(mx - g) < 1e-6f) h = ((b - r) / d + 2.0f) / 6.0f;
    else h = ((r - g) / d + 4.0f) / 6.0f;
    return {h, s, l};
}
PFX_DEV float hue_to_rgb(float p, float q, float t)
{
    if (t < 0.0f) t += 1.0f;
    if (t > 1.0f) t -= 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 1.0f / 2.0f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}
PFX_DEV uint32_t rs_f32_as_u32(float v) { return (v > 0.0f) ? ((v >= 4294967296.0f) ? 0xffffffffu : (uint32_t)v) : 0u; }

__global__ __launch_bounds__(256) void brush_kernel(uint32_t* __restrict__ target, uint32_t w, uint32_t h, pfxk_brush B,
                                                    const float2* __restrict__ pts, uint32_t n_pts,
                                                    const uint8_t* __restrict__ lut, const uint8_t* __restrict__ selection,
                                                    int bx0, int by0, int bx1, int by1)
{
    const int gx = bx0 + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int gy = by0 + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (gx > bx1 || gy > by1) return;
    const size_t i = (size_t)gy * w + (size_t)gx;
    if (selection && selection[i] == 0) return; // :312-320
    uint32_t px = target[i];
    const uint32_t px_in = px;
    const uint32_t wm1 = w ? w - 1u : 0u, hm1 = h ? h - 1u : 0u;
    for (uint32_t k = 0; k < n_pts; ++k) {
        const float2 c = pts[k];
        // stamp bounding box exactly as the reference computes it (:209-215)
        const uint32_t min_x = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(c.x - B.draw_radius), 0.0f));
        const uint32_t max_x = min(rs_f32_as_u32(__builtin_ceilf(c.x + B.draw_radius)), wm1);
        const uint32_t min_y = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(c.y - B.draw_radius), 0.0f));
        const uint32_t max_y = min(rs_f32_as_u32(__builtin_ceilf(c.y + B.draw_radius)), hm1);
        if ((uint32_t)gx < min_x || (uint32_t)gx > max_x || (uint32_t)gy < min_y || (uint32_t)gy > max_y) continue;
        const float dy = (float)gy - c.y, dx = (float)gx - c.x;
        const float dist_sq = dx * dx + dy * dy;
        if (dist_sq > B.draw_radius_sq) continue;
        uint32_t geom_u8;
        if (B.use_direct_alpha) {
            const float a = pfx_brush_alpha(__builtin_sqrtf(dist_sq), B.radius, B.hardness, B.anti_aliased != 0);
            geom_u8 = (uint32_t)__builtin_fminf(__builtin_roundf(a * 255.0f), 255.0f); // :330-333
        } else {
            geom_u8 = lut[rs_f32_as_u32(__builtin_fminf(dist_sq * B.inv_radius_sq * 255.0f, 255.0f))]; // :335-336
        }
        if (geom_u8 == 0u) continue;
        const float geom_alpha = div255((float)geom_u8);
        if (B.is_eraser) { // :345-356
            const float erase_strength = geom_alpha * B.src_a * B.flow;
            if (erase_strength < 0.01f) continue;
            const float old_mask = div255((float)(px >> 24));
            if (erase_strength > old_mask) px = (uint32_t)trunc_u8f(erase_strength * 255.0f) << 24;
        } else {
            const float brush_alpha = geom_alpha * B.src_a * B.flow;
            if (brush_alpha < 0.01f) continue;
            if (B.mode == 0) { // Normal: max-alpha stamping, ties overwrite (:363-373)
                const uint32_t a8 = (uint32_t)trunc_u8f(brush_alpha * 255.0f);
                if (a8 >= (px >> 24)) px = B.rgb8 | (a8 << 24);
            } else { // Dodge / Burn / Sponge (:374-393)
                hsl3 c3 = rgb_to_hsl(div255(ubyte0(px)), div255(ubyte1(px)), div255(ubyte2(px)));
                const float strength = brush_alpha * 0.5f;
                if (B.mode == 1) c3.l = rs_clamp(c3.l + strength, 0.0f, 1.0f);
                else if (B.mode == 2) c3.l = rs_clamp(c3.l - strength, 0.0f, 1.0f);
                else if (B.mode == 3) c3.s = rs_clamp(c3.s - strength, 0.0f, 1.0f);
                float nr, ng, nb;
                if (__builtin_fabsf(c3.s) < 1e-6f) { nr = ng = nb = c3.l; }
                else {
                    const float q = (c3.l < 0.5f) ? c3.l * (1.0f + c3.s) : c3.l + c3.s - c3.l * c3.s;
                    const float p = 2.0f * c3.l - q;
                    nr = hue_to_rgb(p, q, c3.h + 1.0f / 3.0f);
                    ng = hue_to_rgb(p, q, c3.h);
                    nb = hue_to_rgb(p, q, c3.h - 1.0f / 3.0f);
                }
                px = (px & 0xff000000u) | (uint32_t)trunc_u8f(nr * 255.0f) | ((uint32_t)trunc_u8f(ng * 255.0f) << 8) |
                     ((uint32_t)trunc_u8f(nb * 255.0f) << 16);
            }
        }
    }
    if (px != px_in) target[i] = px;
}

} // namespace

extern "C" hipError_t pfxk_brush_stamps(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                                        const float* d_points_xy, uint32_t n_points, const uint8_t* d_lut256,
                                        const uint8_t* d_selection, int bx0, int by0, int bx1, int by1)
{
    if (n_points == 0 || bx1 < bx0 || by1 < by0) return hipSuccess;
    dim3 g((uint32_t)(bx1 - bx0 + 64) / 64u, (uint32_t)(by1 - by0 + 4) / 4u);
    brush_kernel<<<g, 256, 0, s>>>((uint32_t*)d_target, w, h, *B, (const float2*)d_points_xy, n_points, d_lut256,
                                   d_selection, bx0, by0, bx1, by1);
    return hipGetLastError();
}
