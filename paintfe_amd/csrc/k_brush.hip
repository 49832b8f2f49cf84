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

// geometry of one image-tip stamp at pixel (gx, gy): draw_image_tip_no_dirty :584-727 — the caller has already tested the stamp's box (S.x0 .. S.y1, :584-587).
// Returns false when the stamp does not touch the pixel; otherwise geom_u8 = the tip mask's coverage there (nearest texel, or bilinear under inverse rotation).
PFX_DEV bool tip_coverage(const pfxk_brush& B, const pfxk_stamp& S, const uint8_t* __restrict__ mask, int gx, int gy, uint32_t& geom_u8)
{
    const uint32_t ms = B.tip_size;
    const float half = (float)ms / 2.0f;
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

// compute_brush_alpha (brush_render.rs:54-82; k_brush_math.h is the host's form) with its two divisions by wave-uniform denominators — the radius and
// edge1 - edge0 — as prepared reciprocals (k_common.h: rdiv, correctly rounded, identical to `/`; numerators here are 0 or >= 2^-24 in magnitude, below 2^20).
// The caller guarantees radius > 0 (brush_prepare skips radius^2 < 0.001).
struct brush_alpha_k { rdiv radius, edge; float hard_m1, edge0, edge1; };
PFX_DEV brush_alpha_k brush_alpha_prepare(float radius, float hardness)
{
    const float edge0 = radius + 0.5f, edge1 = radius - 0.5f;
    return {rdiv_prepare(radius), rdiv_prepare(edge1 - edge0), rs_clamp(hardness, 0.0f, 1.0f) - 1.0f, edge0, edge1};
}
template <bool AA> PFX_DEV float brush_alpha_dev(const brush_alpha_k& K, float dist, float radius)
{
    const float t = rs_clamp(rdiv_apply(K.radius, dist), 0.0f, 1.0f);
    const float falloff = t * t * (3.0f - 2.0f * t);
    const float material_alpha = 1.0f + K.hard_m1 * falloff;
    float coverage;
    if constexpr (AA) {
        if (dist <= K.edge1) coverage = 1.0f;
        else if (dist >= K.edge0) coverage = 0.0f;
        else {
            const float x = rs_clamp(rdiv_apply(K.edge, dist - K.edge0), 0.0f, 1.0f);
            coverage = x * x * (3.0f - 2.0f * x);
        }
    } else coverage = (dist <= radius) ? 1.0f : 0.0f;
    return material_alpha * coverage;
}

// KIND: what a stamp does to a pixel — wave-uniform for the whole launch, so each kind is its own straight-line kernel body
enum : int { BK_PAINT = 0, BK_ERASER = 1, BK_TONE = 2 /* Dodge / Burn / Sponge */, BK_TIP = 3 /* image tip: paint or eraser */ };

// BINNED: the host has dealt the stamps to the 64 x 64 chunks their bounding boxes touch (pfx_api.cpp: pfx_brush_stamps_ex_dev — TiledImage's chunk grid); a
// workgroup takes four rows of one ACTIVE chunk (`chunks[blockIdx.x]` = chunk x, chunk y, first entry, entries) and walks that chunk's stamp list `bins`
// (indices into `stamps`, in stroke order).  The work is then proportional to the painted area, not to the stroke's bounding box times its length.  Every
// per-pixel test is the unbinned kernel's, so the result is bit-identical.  A stamp's pixel box (S.x0 .. S.y1: :209-215 for the round tip, :584-587 for an image
// tip) is computed once by the host prologue with the reference's float operations.
template <bool BINNED, int KIND, bool DIRECT>
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
    const uint32_t seg_lo = (uint32_t)gx0, seg_hi = (uint32_t)min(gx0 + 63, bx1);
    const brush_alpha_k AK = brush_alpha_prepare(B.radius, B.hardness);
    const float gain = B.src_a * B.flow;
    (void)AK; (void)gain;
    auto stamp_px = [&](const pfxk_stamp& S) {   // S is wave-uniform (broadcast from the lane that culled it)
        if ((uint32_t)gx < S.x0 || (uint32_t)gx > S.x1 || (uint32_t)gy < S.y0 || (uint32_t)gy > S.y1) return;   // the stamp's box (host prologue)
        if constexpr (KIND == BK_TIP) { // image tip (:533-760): max-alpha stamping or eraser only, no brush modes, no 0.01 cut-off for paint
            uint32_t g8;
            if (!tip_coverage(B, S, tip_mask, gx, gy, g8) || g8 == 0u) return;
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
        } else {
            const float dy = (float)gy - S.cy, dx = (float)gx - S.cx;
            const float dist_sq = dx * dx + dy * dy;
            if (dist_sq > B.draw_radius_sq) return;
            uint32_t geom_u8;
            if constexpr (DIRECT) {   // anti-aliased: the alpha straight from the distance (:330-333); DIRECT <=> draw_radius > radius <=> anti_aliased
                const float a = brush_alpha_dev<true>(AK, __builtin_sqrtf(dist_sq), B.radius);
                geom_u8 = (uint32_t)__builtin_fminf(__builtin_roundf(a * 255.0f), 255.0f);
            } else {
                geom_u8 = lut[rs_f32_as_u32(__builtin_fminf(dist_sq * B.inv_radius_sq * 255.0f, 255.0f))]; // :335-336
            }
            if (geom_u8 == 0u) return;
            const float geom_alpha = div255((float)geom_u8);
            if constexpr (KIND == BK_ERASER) { // :345-356
                const float erase_strength = geom_alpha * B.src_a * B.flow;
                if (erase_strength < 0.01f) return;
                const float old_mask = div255((float)(px >> 24));
                if (erase_strength > old_mask) px = (uint32_t)trunc_u8f(erase_strength * 255.0f) << 24;
            } else {
                const float brush_alpha = geom_alpha * B.src_a * B.flow;
                if (brush_alpha < 0.01f) return;
                if constexpr (KIND == BK_PAINT) { // Normal: max-alpha stamping, ties overwrite (:363-373)
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
        }
    };
    // Stamp culling, 64 stamps at a time: lane j tests stamp k0 + j's box — the very box the per-pixel test uses — against the wave's row segment; only the
    // stamps some lane of the wave can see are walked, in order.  (A long diagonal stroke's bounding box is mostly empty and a row segment sees a few per cent
    // of the stamps: without this every lane walked all of them.)
    for (uint32_t k0 = 0; k0 < n_pts; k0 += 64u) {
        const uint32_t kk = k0 + lane;
        bool seen = false;
        uint32_t sidx = kk;                                   // BINNED: entry kk of the chunk's list names stamp bins[kk]
        if constexpr (BINNED) sidx = kk < n_pts ? bins[kk] : 0u;
        // Short strokes (the bounding-box launch: a mouse segment's few dozen stamps, where a wave has nothing else to hide latency behind): the lane keeps the
        // stamp it tested and the walk broadcasts a visible stamp's fields from that lane (v_readlane) instead of fetching the stamp again through the scalar
        // cache — 0.035 -> 0.028 ms for a 60-stamp segment.  Long (binned) strokes keep the scalar loads: their waves overlap each other's latency and the eight
        // broadcasts per stamp cost more than they save (6 501-stamp stroke 0.214 against 0.250 ms, same box).
        constexpr bool BCAST = !BINNED;
        uint4 q0 = make_uint4(0u, 0u, 0u, 0u), bx = make_uint4(1u, 0u, 1u, 0u);
        float2 rot = make_float2(1.0f, 0.0f);
        if (kk < n_pts) {
            const uint4* sp = reinterpret_cast<const uint4*>(&stamps[sidx]);
            bx = sp[1];
            if constexpr (BCAST) { q0 = sp[0]; if constexpr (KIND == BK_TIP) rot = *reinterpret_cast<const float2*>(sp + 2); }
            seen = !(seg_hi < bx.x || seg_lo > bx.y || (uint32_t)gy < bx.z || (uint32_t)gy > bx.w);
        }
        unsigned long long m = __ballot(seen);
        while (m) {
            const int j = (int)__builtin_ctzll(m);
            m &= m - 1ull;
            if constexpr (BCAST) {
                auto bc = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, j); };
                pfxk_stamp S;
                S.cx = __builtin_bit_cast(float, bc(q0.x)); S.cy = __builtin_bit_cast(float, bc(q0.y)); S.rgb8 = bc(q0.z);
                S.x0 = bc(bx.x); S.x1 = bc(bx.y); S.y0 = bc(bx.z); S.y1 = bc(bx.w);
                if constexpr (KIND == BK_TIP) {
                    S.rotated = bc(q0.w);
                    S.cos_a = __builtin_bit_cast(float, bc(__builtin_bit_cast(uint32_t, rot.x))); S.sin_a = __builtin_bit_cast(float, bc(__builtin_bit_cast(uint32_t, rot.y)));
                } else { S.rotated = 0u; S.cos_a = 1.0f; S.sin_a = 0.0f; }
                if (active) stamp_px(S);
            } else {
                if (active) stamp_px(stamps[(uint32_t)__builtin_amdgcn_readlane((int)sidx, j)]);   // uniform index -> scalar loads
            }
        }
    }
    if (px != px_in) target[i] = px;
}

template <bool BINNED>
hipError_t launch_brush(hipStream_t s, dim3 g, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B, const pfxk_stamp* d_stamps, uint32_t n_points,
                        const uint8_t* d_lut256, const uint8_t* d_tip_mask, const uint8_t* d_selection, int bx0, int by0, int bx1, int by1, const uint32_t* d_chunks,
                        const uint32_t* d_bins)
{
#define PFX_BRUSH_GO(KIND, DIRECT) brush_kernel<BINNED, KIND, DIRECT><<<g, 256, 0, s>>>((uint32_t*)d_target, w, h, *B, d_stamps, n_points, d_lut256, d_tip_mask, \
                                                                                   d_selection, bx0, by0, bx1, by1, (const uint4*)d_chunks, d_bins)
    if (B->tip_size) PFX_BRUSH_GO(BK_TIP, false);
    else if (B->use_direct_alpha) {
        if (B->is_eraser) PFX_BRUSH_GO(BK_ERASER, true); else if (B->mode == 0) PFX_BRUSH_GO(BK_PAINT, true); else PFX_BRUSH_GO(BK_TONE, true);
    } else {
        if (B->is_eraser) PFX_BRUSH_GO(BK_ERASER, false); else if (B->mode == 0) PFX_BRUSH_GO(BK_PAINT, false); else PFX_BRUSH_GO(BK_TONE, false);
    }
#undef PFX_BRUSH_GO
    return hipGetLastError();
}

} // namespace

extern "C" hipError_t pfxk_brush_stamps(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                                        const pfxk_stamp* d_stamps, uint32_t n_points, const uint8_t* d_lut256, const uint8_t* d_tip_mask,
                                        const uint8_t* d_selection, int bx0, int by0, int bx1, int by1)
{
    if (n_points == 0 || bx1 < bx0 || by1 < by0) return hipSuccess;
    const dim3 g((uint32_t)(bx1 - bx0 + 64) / 64u, (uint32_t)(by1 - by0 + 4) / 4u);
    return launch_brush<false>(s, g, d_target, w, h, B, d_stamps, n_points, d_lut256, d_tip_mask, d_selection, bx0, by0, bx1, by1, nullptr, nullptr);
}

// the same stamps dealt to chunks: d_chunks = n_chunks x {chunk x, chunk y, first entry of d_bins, entries}, d_bins = stamp indices in stroke order per chunk
extern "C" hipError_t pfxk_brush_stamps_binned(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                                               const pfxk_stamp* d_stamps, uint32_t n_points, const uint8_t* d_lut256, const uint8_t* d_tip_mask,
                                               const uint8_t* d_selection, const uint32_t* d_chunks, uint32_t n_chunks, const uint32_t* d_bins)
{
    if (n_points == 0 || n_chunks == 0) return hipSuccess;
    return launch_brush<true>(s, dim3(n_chunks, 16), d_target, w, h, B, d_stamps, n_points, d_lut256, d_tip_mask, d_selection, 0, 0, 0, 0, d_chunks, d_bins);
}
