// k_brush.hip — the brush stamp loop over a preview image: round tip and image tips.
//
// Reference: draw_circle_no_dirty src/ui/panels/tools/behavior/raster/brush_render.rs:135-400, draw_image_tip_no_dirty
// :533-760, rebuild_brush_lut :27-50, compute_brush_alpha :54-82.  Scatter, colour jitter and tip rotation are per-stamp
// quantities the reference computes on the CPU before touching pixels; here the host prologue (pfx_api.cpp) does the same and
// hands the kernel a list of prepared stamps (centre, colour bytes, inverse-rotation cos / sin).
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

PFX_DEV int32_t rs_f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)v;
}

// geometry of one image-tip stamp at pixel (gx, gy): draw_image_tip_no_dirty :584-727.  Returns false when the stamp does not
// touch the pixel; otherwise geom_u8 = the tip mask's coverage there (nearest texel, or bilinear under inverse rotation).
PFX_DEV bool tip_coverage(const pfxk_brush& B, const pfxk_stamp& S, const uint8_t* __restrict__ mask, int gx, int gy, uint32_t wm1, uint32_t hm1,
                          uint32_t& geom_u8)
{
    const uint32_t ms = B.tip_size;
    const float half = (float)ms / 2.0f;
    const float eh = S.rotated ? half * 1.41421356237309504880f : half;
    const uint32_t min_x = rs_f32_as_u32(__builtin_fmaxf(S.cx - eh, 0.0f)), min_y = rs_f32_as_u32(__builtin_fmaxf(S.cy - eh, 0.0f));
    const uint32_t max_x = min(rs_f32_as_u32(S.cx + eh), wm1), max_y = min(rs_f32_as_u32(S.cy + eh), hm1);
    if ((uint32_t)gx < min_x || (uint32_t)gx > max_x || (uint32_t)gy < min_y || (uint32_t)gy > max_y) return false;
    const float rel_x = (float)gx - S.cx, rel_y = (float)gy - S.cy;
    if (S.rotated) {
        const float rot_x = rel_x * S.cos_a - rel_y * S.sin_a + half, rot_y = rel_x * S.sin_a + rel_y * S.cos_a + half;
        if (rot_x < -0.5f || rot_y < -0.5f || rot_x >= (float)ms - 0.5f || rot_y >= (float)ms - 0.5f) return false;
        const float sx = __builtin_fmaxf(rot_x, 0.0f), sy = __builtin_fmaxf(rot_y, 0.0f);
        const uint32_t sx0 = rs_f32_as_u32(__builtin_floorf(sx)), sy0 = rs_f32_as_u32(__builtin_floorf(sy));
        const uint32_t sx1 = min(sx0 + 1u, ms - 1u), sy1 = min(sy0 + 1u, ms - 1u);
        const float fx = sx - (float)sx0, fy = sy - (float)sy0;
        const float v00 = (float)mask[sy0 * ms + sx0], v10 = (float)mask[sy0 * ms + sx1], v01 = (float)mask[sy1 * ms + sx0], v11 = (float)mask[sy1 * ms + sx1];
        const float top = v00 * (1.0f - fx) + v10 * fx, bot = v01 * (1.0f - fx) + v11 * fx;
        geom_u8 = (uint32_t)__builtin_fminf(__builtin_roundf(top * (1.0f - fy) + bot * fy), 255.0f);
    } else {
        const int32_t mx = rs_f32_as_i32(__builtin_roundf(rel_x + half)), my = rs_f32_as_i32(__builtin_roundf(rel_y + half));
        if (mx < 0 || my < 0 || mx >= (int32_t)ms || my >= (int32_t)ms) return false;
        geom_u8 = mask[(uint32_t)my * ms + (uint32_t)mx];
    }
    return true;
}

// BINNED: the host has dealt the stamps to the 64 x 64 chunks their bounding boxes touch (pfx_api.cpp: brush_bin_stamps — TiledImage's chunk grid); a workgroup
// takes four rows of one ACTIVE chunk (`chunks[blockIdx.x]` = chunk x, chunk y, first entry, entries) and walks that chunk's stamp list `bins` (indices into
// `stamps`, in stroke order).  The work is then proportional to the painted area, not to the stroke's bounding box times its length: a diagonal stroke of 6 501
// stamps across an 8K preview layer 2.07 -> see profiles/r05_tuning.md.  Every per-pixel test is the unbinned kernel's, so the result is bit-identical.
template <bool BINNED>
__global__ __launch_bounds__(256) void brush_kernel(uint32_t* __restrict__ target, uint32_t w, uint32_t h, pfxk_brush B,
                                                    const pfxk_stamp* __restrict__ stamps, uint32_t n_pts_all,
                                                    const uint8_t* __restrict__ lut, const uint8_t* __restrict__ tip_mask,
                                                    const uint8_t* __restrict__ selection, int bx0, int by0, int bx1, int by1,
                                                    const uint4* __restrict__ chunks, const uint32_t* __restrict__ bins)
{
    // a wave covers one 64-pixel row segment: gy and the segment's x range are wave-uniform
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t n_pts = n_pts_all;
    if constexpr (BINNED) {
        const uint4 ch = chunks[blockIdx.x];
        bx0 = (int)(ch.x * 64u); by0 = (int)(ch.y * 64u);
        bx1 = min(bx0 + 63, (int)w - 1); by1 = min(by0 + 63, (int)h - 1);
        bins += ch.z; n_pts = ch.w;
    }
    const int gx0 = BINNED ? bx0 : bx0 + (int)(blockIdx.x * 64u), gx = gx0 + (int)lane;
    const int gy = by0 + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (gy > by1 || gx0 > bx1) return; // whole wave
    const size_t i = (size_t)gy * w + (size_t)min(gx, bx1);
    bool active = gx <= bx1;
    if (active && selection && selection[i] == 0) active = false; // :312-320
    uint32_t px = active ? target[i] : 0u;
    const uint32_t px_in = px;
    const uint32_t wm1 = w ? w - 1u : 0u, hm1 = h ? h - 1u : 0u;
    const uint32_t seg_lo = (uint32_t)gx0, seg_hi = (uint32_t)min(gx0 + 63, bx1);
    auto stamp_px = [&](uint32_t k) {
        const pfxk_stamp S = stamps[k]; // uniform -> scalar loads
        if (B.tip_size) { // image tip (:533-760): max-alpha stamping or eraser only, no brush modes, no 0.01 cut-off for paint
            uint32_t g8;
            if (!tip_coverage(B, S, tip_mask, gx, gy, wm1, hm1, g8) || g8 == 0u) return;
            const float geom_alpha = div255((float)g8);
            if (B.is_eraser) {
                const float erase_strength = geom_alpha * B.src_a * B.flow;
                if (erase_strength < 0.01f) return;
                if (erase_strength > div255((float)(px >> 24))) px = (uint32_t)trunc_u8f(erase_strength * 255.0f) << 24;
            } else {
                const uint32_t a8 = (uint32_t)trunc_u8f(geom_alpha * B.src_a * B.flow * 255.0f);
                if (a8 >= (px >> 24)) px = S.rgb8 | (a8 << 24);
            }
            return;
        }
        const float2 c = make_float2(S.cx, S.cy);
        // stamp bounding box exactly as the reference computes it (:209-215)
        const uint32_t min_x = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(c.x - B.draw_radius), 0.0f));
        const uint32_t max_x = min(rs_f32_as_u32(__builtin_ceilf(c.x + B.draw_radius)), wm1);
        const uint32_t min_y = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(c.y - B.draw_radius), 0.0f));
        const uint32_t max_y = min(rs_f32_as_u32(__builtin_ceilf(c.y + B.draw_radius)), hm1);
        if ((uint32_t)gx < min_x || (uint32_t)gx > max_x || (uint32_t)gy < min_y || (uint32_t)gy > max_y) return;
        const float dy = (float)gy - c.y, dx = (float)gx - c.x;
        const float dist_sq = dx * dx + dy * dy;
        if (dist_sq > B.draw_radius_sq) return;
        uint32_t geom_u8;
        if (B.use_direct_alpha) {
            const float a = pfx_brush_alpha(__builtin_sqrtf(dist_sq), B.radius, B.hardness, B.anti_aliased != 0);
            geom_u8 = (uint32_t)__builtin_fminf(__builtin_roundf(a * 255.0f), 255.0f); // :330-333
        } else {
            geom_u8 = lut[rs_f32_as_u32(__builtin_fminf(dist_sq * B.inv_radius_sq * 255.0f, 255.0f))]; // :335-336
        }
        if (geom_u8 == 0u) return;
        const float geom_alpha = div255((float)geom_u8);
        if (B.is_eraser) { // :345-356
            const float erase_strength = geom_alpha * B.src_a * B.flow;
            if (erase_strength < 0.01f) return;
            const float old_mask = div255((float)(px >> 24));
            if (erase_strength > old_mask) px = (uint32_t)trunc_u8f(erase_strength * 255.0f) << 24;
        } else {
            const float brush_alpha = geom_alpha * B.src_a * B.flow;
            if (brush_alpha < 0.01f) return;
            if (B.mode == 0) { // Normal: max-alpha stamping, ties overwrite (:363-373)
                const uint32_t a8 = (uint32_t)trunc_u8f(brush_alpha * 255.0f);
                if (a8 >= (px >> 24)) px = S.rgb8 | (a8 << 24);
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
    };
    // Stamp culling, 64 stamps at a time: lane j tests stamp k0 + j's bounding box — the very box the per-pixel test uses — against the
    // wave's row segment; only the stamps some lane of the wave can see are walked, in order.  (A long diagonal stroke's bounding box is
    // mostly empty and a row segment sees a few per cent of the stamps: without this every lane walked all of them.)
    for (uint32_t k0 = 0; k0 < n_pts; k0 += 64u) {
        const uint32_t kk = k0 + lane;
        bool seen = false;
        uint32_t sidx = kk;                                   // BINNED: entry kk of the chunk's list names stamp bins[kk]
        if constexpr (BINNED) sidx = kk < n_pts ? bins[kk] : 0u;
        if (kk < n_pts) {
            const float cx = stamps[sidx].cx, cy = stamps[sidx].cy;
            uint32_t min_x, max_x, min_y, max_y;
            if (B.tip_size) { // tip_coverage's box
                const float half = (float)B.tip_size / 2.0f, eh = stamps[sidx].rotated ? half * 1.41421356237309504880f : half;
                min_x = rs_f32_as_u32(__builtin_fmaxf(cx - eh, 0.0f)); min_y = rs_f32_as_u32(__builtin_fmaxf(cy - eh, 0.0f));
                max_x = min(rs_f32_as_u32(cx + eh), wm1); max_y = min(rs_f32_as_u32(cy + eh), hm1);
            } else {          // :209-215
                min_x = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(cx - B.draw_radius), 0.0f));
                max_x = min(rs_f32_as_u32(__builtin_ceilf(cx + B.draw_radius)), wm1);
                min_y = rs_f32_as_u32(__builtin_fmaxf(__builtin_floorf(cy - B.draw_radius), 0.0f));
                max_y = min(rs_f32_as_u32(__builtin_ceilf(cy + B.draw_radius)), hm1);
            }
            seen = !(seg_hi < min_x || seg_lo > max_x || (uint32_t)gy < min_y || (uint32_t)gy > max_y);
        }
        unsigned long long m = __ballot(seen);
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            const uint32_t k = BINNED ? (uint32_t)__builtin_amdgcn_readlane((int)sidx, (int)j) : k0 + j;
            if (active) stamp_px(k);
        }
    }
    if (px != px_in) target[i] = px;
}

} // namespace

extern "C" hipError_t pfxk_brush_stamps(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                                        const pfxk_stamp* d_stamps, uint32_t n_points, const uint8_t* d_lut256, const uint8_t* d_tip_mask,
                                        const uint8_t* d_selection, int bx0, int by0, int bx1, int by1)
{
    if (n_points == 0 || bx1 < bx0 || by1 < by0) return hipSuccess;
    dim3 g((uint32_t)(bx1 - bx0 + 64) / 64u, (uint32_t)(by1 - by0 + 4) / 4u);
    brush_kernel<false><<<g, 256, 0, s>>>((uint32_t*)d_target, w, h, *B, d_stamps, n_points, d_lut256, d_tip_mask, d_selection, bx0, by0, bx1, by1, nullptr, nullptr);
    return hipGetLastError();
}

// the same stamps dealt to chunks: d_chunks = n_chunks x {chunk x, chunk y, first entry of d_bins, entries}, d_bins = stamp indices in stroke order per chunk
extern "C" hipError_t pfxk_brush_stamps_binned(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                                               const pfxk_stamp* d_stamps, uint32_t n_points, const uint8_t* d_lut256, const uint8_t* d_tip_mask,
                                               const uint8_t* d_selection, const uint32_t* d_chunks, uint32_t n_chunks, const uint32_t* d_bins)
{
    if (n_points == 0 || n_chunks == 0) return hipSuccess;
    brush_kernel<true><<<dim3(n_chunks, 16), 256, 0, s>>>((uint32_t*)d_target, w, h, *B, d_stamps, n_points, d_lut256, d_tip_mask, d_selection, 0, 0, 0, 0,
                                                         (const uint4*)d_chunks, d_bins);
    return hipGetLastError();
}
