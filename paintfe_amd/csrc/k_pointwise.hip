// k_pointwise.hip — the pointwise adjustment bank, both CPU numeric flavours, as one chunk-tiled streaming kernel.
//
// Reference (ops::adjustments flavour: f32, `.round().clamp(0,255) as u8`, selection aware, chunk-sparse):
//   src/ops/adjustments.rs:21-108 drivers; :115-140 invert/sepia; :265-416 B/C, HSL, exposure, highlights/shadows;
//   :465-511 levels LUT apply; :517-526 temperature/tint; :584-631 curves LUT apply; :944-1012 HSL helpers;
//   :1240-1441 threshold, posterize, colour balance, gradient map, b&w, vibrance; src/ops/filters.rs:321-378 desaturate.
// Reference (Rhai-inline flavour: truncating `as u8`, alpha untouched, selection ignored): src/ops/scripting.rs:869-1075.
//
// Design: HBM-bound (4 B read + 4 B written per pixel).  One workgroup owns one 64x64 TiledImage chunk
// (ref: src/canvas/defs.rs:7), a lane owns 4 rows x 4 consecutive pixels (four 16-byte loads/stores), so the
// TiledImage sparsity rules — "only populated chunks are visited" (adjustments.rs:29, tiled_image.rs:905-932) and
// "result chunks with no alpha are dropped" (adjustments.rs:105, tiled_image.rs:81-95) — are a single
// workgroup-wide OR of alpha with no extra pass over memory.  256-entry LUTs are staged in LDS.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

struct hsl3 { float h, s, l; };
struct rgb3 { float r, g, b; };

// adjustments.rs:944-974 (EPS = 1e-6).  Only ever called on k / 255 values (k a byte, div255 is exact RN): two of them are equal or
// at least 1/255 apart, so the reference's `(a - b).abs() < 1e-6` tests are plain equality tests here.
PFX_DEV hsl3 rgb_to_hsl(float r, float g, float b)
{
    const float mx = __builtin_fmaxf(__builtin_fmaxf(r, g), b);
    const float mn = __builtin_fminf(__builtin_fminf(r, g), b);
    const float l = (mx + mn) / 2.0f;
    // Branch-free on purpose (the branches of the reference are per-pixel data: a wave takes all of them, and every divergent
    // branch costs scalar exec-mask traffic on top): the same operations in the same order, operands selected before each
    // division, results selected at the end.  Lanes of the grey case divide 0 by 0; that NaN is never selected.
    const bool gray = mx == mn;
    // Divisions via k_common.h:rdiv (bit-identical to '/'): inputs are k/255, so d = max-min >= 1/255, the saturation
    // denominators are >= 1/255 and every numerator is 0 or >= 1/255 in magnitude — all in the normal range.
    const float d = mx - mn;
    const float s = fdiv_fast(d, (l > 0.5f) ? (2.0f - mx - mn) : (mx + mn)); // selecting the operand == selecting the quotient
    const rdiv kd = rdiv_prepare(d), k6 = rdiv_prepare(6.0f);
    const bool is_r = mx == r, is_g = mx == g;
    const float n_rb = pin(is_g ? (b - r) : (r - g));
    const float sector = rdiv_apply(kd, is_r ? pin(g - b) : n_rb);
    // red sector: `if h < 0 { h += 6 }` (adding +0.0 otherwise leaves the value as it is); green: + 2; blue: + 4
    const float offset = is_r ? ((sector < 0.0f) ? 6.0f : 0.0f) : (is_g ? 2.0f : 4.0f);
    const float h = rdiv_apply(k6, sector + offset);
    return {gray ? 0.0f : h, gray ? 0.0f : s, l};
}
// adjustments.rs:995-1012 / scripting.rs hue_to_rgb for a t that is already wrapped into [0, 1].  The reference's two ramps
//   t < 1/6:        p + ((q - p) * 6) * t            1/2 <= t < 2/3:  p + ((q - p) * (2/3 - t)) * 6
// are ONE multiply-multiply-add with the operands picked first ((q - p) * x) * y, x = 6 or 2/3 - t, y = t or 6: the same operations on
// the same values in the same order as whichever ramp the reference evaluates; its if-chain is the three selects at the end.
PFX_DEV float hue_seg(float p, float q, float qmp, float t)
{
    const bool up = t < 1.0f / 6.0f, is_q = t < 1.0f / 2.0f, down = t < 2.0f / 3.0f;
    const float x = up ? 6.0f : (2.0f / 3.0f - t);
    const float y = up ? t : 6.0f;
    const float ramp = pin(p + (qmp * x) * y);
    float v = down ? ramp : p;
    v = is_q ? q : v;
    return up ? ramp : v;
}
// adjustments.rs:976-993.  h is in [0, 1] at every call site (fract() + 1 if negative; rem_euclid(1.0); rgb_to_hsl's own h), so of
// hue_to_rgb's two wrap tests `t < 0 -> t + 1`, `t > 1 -> t - 1` only one can fire per channel: h + 1/3 lies in [1/3, 4/3], h itself
// needs none, h - 1/3 lies in [-1/3, 2/3] (and a wrapped value never trips the other test).
template <bool RHAI>
PFX_DEV rgb3 hsl_to_rgb(float h, float s, float l)
{
    const bool gray = __builtin_fabsf(s) < (RHAI ? 1e-10f : 1e-6f);
    const float q_lo = pin(l * (1.0f + s)), q_hi = pin(l + s - l * s);
    const float q = (l < 0.5f) ? q_lo : q_hi;
    const float p = 2.0f * l - q;
    const float qmp = q - p;
    float tr = h + 1.0f / 3.0f, tb = h - 1.0f / 3.0f;
    tr = (tr > 1.0f) ? tr - 1.0f : tr;
    tb = (tb < 0.0f) ? tb + 1.0f : tb;
    const float r = hue_seg(p, q, qmp, tr), g = hue_seg(p, q, qmp, h), b = hue_seg(p, q, qmp, tb);
    return {gray ? l : r, gray ? l : g, gray ? l : b};
}
PFX_DEV float lum709(float r, float g, float b) { return 0.2126f * r + 0.7152f * g + 0.0722f * b; }

// One pixel of the ops::adjustments flavour: (r,g,b,a) in 0..255 as f32 -> unrounded f32 (the closure of
// apply_pixel_transform).  P = parameter block prepared by the host (pfx_api.cpp:prepare_adjust).
template <int OP>
PFX_DEV void adjust_px(const pfxk_params& P, const uint8_t* __restrict__ lut, float r, float g, float b, float a,
                       float (&o)[4])
{
    o[3] = a;
    if constexpr (OP == PFXK_OP_INVERT) { o[0] = 255.0f - r; o[1] = 255.0f - g; o[2] = 255.0f - b; }
    else if constexpr (OP == PFXK_OP_INVERT_ALPHA) { o[0] = r; o[1] = g; o[2] = b; o[3] = 255.0f - a; }
    else if constexpr (OP == PFXK_OP_SEPIA) {
        o[0] = __builtin_fminf(0.393f * r + 0.769f * g + 0.189f * b, 255.0f);
        o[1] = __builtin_fminf(0.349f * r + 0.686f * g + 0.168f * b, 255.0f);
        o[2] = __builtin_fminf(0.272f * r + 0.534f * g + 0.131f * b, 255.0f);
    } else if constexpr (OP == PFXK_OP_BRIGHTNESS_CONTRAST) { // p0 = brightness, p1 = factor
        o[0] = P.p[1] * (r + P.p[0] - 128.0f) + 128.0f;
        o[1] = P.p[1] * (g + P.p[0] - 128.0f) + 128.0f;
        o[2] = P.p[1] * (b + P.p[0] - 128.0f) + 128.0f;
    } else if constexpr (OP == PFXK_OP_HSL) { // p0 = hue_shift/360, p1 = sat_factor, p2 = light_offset
        const hsl3 c = rgb_to_hsl(div255(r), div255(g), div255(b)); // r, g, b are byte values: div255 == r / 255.0
        float nh = c.h + P.p[0];
        nh = nh - __builtin_truncf(nh); // f32::fract
        if (nh < 0.0f) nh = nh + 1.0f;
        const float ns = rs_clamp(c.s * P.p[1], 0.0f, 1.0f);
        const rgb3 n = hsl_to_rgb<false>(nh, ns, c.l);
        o[0] = n.r * 255.0f + P.p[2]; o[1] = n.g * 255.0f + P.p[2]; o[2] = n.b * 255.0f + P.p[2];
    } else if constexpr (OP == PFXK_OP_EXPOSURE) { // p0 = gain
        o[0] = r * P.p[0]; o[1] = g * P.p[0]; o[2] = b * P.p[0];
    } else if constexpr (OP == PFXK_OP_HIGHLIGHTS_SHADOWS) { // p0 = shadow_amt, p1 = highlight_amt
        const float lum = lum709(r, g, b) / 255.0f;
        const float sw = (1.0f - lum) * (1.0f - lum);
        const float hw = lum * lum;
        const float adj = sw * P.p[0] * 128.0f + hw * P.p[1] * 128.0f;
        o[0] = r + adj; o[1] = g + adj; o[2] = b + adj;
    } else if constexpr (OP == PFXK_OP_TEMPERATURE_TINT) { // p0 = temp_shift, p1 = tint_shift
        o[0] = r + P.p[0]; o[1] = g - P.p[1] * 0.5f; o[2] = b - P.p[0];
    } else if constexpr (OP == PFXK_OP_THRESHOLD) {
        const float v = (lum709(r, g, b) >= P.p[0]) ? 255.0f : 0.0f;
        o[0] = v; o[1] = v; o[2] = v;
    } else if constexpr (OP == PFXK_OP_POSTERIZE) { // p0 = factor (levels.max(2) as f32)
        const float fm1 = P.p[0] - 1.0f;
        o[0] = __builtin_roundf(r / 255.0f * fm1) / fm1 * 255.0f;
        o[1] = __builtin_roundf(g / 255.0f * fm1) / fm1 * 255.0f;
        o[2] = __builtin_roundf(b / 255.0f * fm1) / fm1 * 255.0f;
    } else if constexpr (OP == PFXK_OP_COLOR_BALANCE) { // p0..8 = shadows, midtones, highlights
        const float lum = lum709(r, g, b) / 255.0f;
        const float sw0 = __builtin_fmaxf(1.0f - lum * 2.0f, 0.0f), hw0 = __builtin_fmaxf(lum * 2.0f - 1.0f, 0.0f);
        const float sw = sw0 * sw0, hw = hw0 * hw0;
        const float mw = __builtin_fmaxf(1.0f - sw - hw, 0.0f);
        o[0] = r + (sw * P.p[0] + mw * P.p[3] + hw * P.p[6]) * 1.28f;
        o[1] = g + (sw * P.p[1] + mw * P.p[4] + hw * P.p[7]) * 1.28f;
        o[2] = b + (sw * P.p[2] + mw * P.p[5] + hw * P.p[8]) * 1.28f;
    } else if constexpr (OP == PFXK_OP_GRADIENT_MAP) { // lut = 256 x RGBA
        const float lf = lum709(r, g, b);
        uint32_t li = (uint32_t)__builtin_fminf(__builtin_fmaxf(lf, 0.0f), 255.0f); // `as usize`.min(255)
        o[0] = (float)lut[li * 4 + 0]; o[1] = (float)lut[li * 4 + 1]; o[2] = (float)lut[li * 4 + 2];
    } else if constexpr (OP == PFXK_OP_BLACK_AND_WHITE) {
        const float v = rs_clamp((r * P.p[0] + g * P.p[1] + b * P.p[2]) / 100.0f, 0.0f, 255.0f);
        o[0] = v; o[1] = v; o[2] = v;
    } else if constexpr (OP == PFXK_OP_VIBRANCE) { // p0 = amount / 100
        const float v = P.p[0];
        const hsl3 c = rgb_to_hsl(div255(r), div255(g), div255(b)); // r, g, b are byte values: div255 == r / 255.0
        const float boost = (v >= 0.0f) ? v * ((1.0f - c.s) * (1.0f - c.s)) : v * (c.s * c.s);
        const float ns = rs_clamp(c.s + boost, 0.0f, 1.0f);
        const rgb3 n = hsl_to_rgb<false>(c.h, ns, c.l);
        o[0] = n.r * 255.0f; o[1] = n.g * 255.0f; o[2] = n.b * 255.0f;
    } else if constexpr (OP == PFXK_OP_LUT_RGBA) { // lut = R[256] G[256] B[256] A[256]
        o[0] = (float)lut[(uint32_t)r]; o[1] = (float)lut[256u + (uint32_t)g]; o[2] = (float)lut[512u + (uint32_t)b];
        o[3] = (float)lut[768u + (uint32_t)a];
    } else if constexpr (OP == PFXK_OP_DESATURATE) {
        const float l = lum709(r, g, b);
        o[0] = l; o[1] = l; o[2] = l;
    } else { o[0] = r; o[1] = g; o[2] = b; }
}

// One pixel of the Rhai-inline flavour: returns final integer-valued channel values (alpha untouched).
template <int OP>
PFX_DEV void rhai_px(const pfxk_params& P, const uint8_t* __restrict__ lut, uint32_t px, float (&o)[4])
{
    const float r = ubyte0(px), g = ubyte1(px), b = ubyte2(px);
    o[3] = ubyte3(px);
    if constexpr (OP == PFXK_RHAI_INVERT) { o[0] = 255.0f - r; o[1] = 255.0f - g; o[2] = 255.0f - b; }
    else if constexpr (OP == PFXK_RHAI_DESATURATE) { // integer (299r + 587g + 114b) / 1000, scripting.rs:891
        const uint32_t gray = ((px & 0xffu) * 299u + ((px >> 8) & 0xffu) * 587u + ((px >> 16) & 0xffu) * 114u) / 1000u;
        o[0] = o[1] = o[2] = (float)gray;
    } else if constexpr (OP == PFXK_RHAI_SEPIA) {
        o[0] = trunc_u8f(__builtin_fminf(r * 0.393f + g * 0.769f + b * 0.189f, 255.0f));
        o[1] = trunc_u8f(__builtin_fminf(r * 0.349f + g * 0.686f + b * 0.168f, 255.0f));
        o[2] = trunc_u8f(__builtin_fminf(r * 0.272f + g * 0.534f + b * 0.131f, 255.0f));
    } else if constexpr (OP == PFXK_RHAI_SEPIA_STRENGTH) { // p0 = strength, p1 = 1 - strength
        const float sr = __builtin_fminf(r * 0.393f + g * 0.769f + b * 0.189f, 255.0f);
        const float sg = __builtin_fminf(r * 0.349f + g * 0.686f + b * 0.168f, 255.0f);
        const float sb = __builtin_fminf(r * 0.272f + g * 0.534f + b * 0.131f, 255.0f);
        o[0] = trunc_u8f(r * P.p[1] + sr * P.p[0]);
        o[1] = trunc_u8f(g * P.p[1] + sg * P.p[0]);
        o[2] = trunc_u8f(b * P.p[1] + sb * P.p[0]);
    } else if constexpr (OP == PFXK_RHAI_BRIGHTNESS_CONTRAST) { // p0 = bright, p1 = factor
        o[0] = quant255(P.p[1] * (r + P.p[0] - 128.0f) + 128.0f);
        o[1] = quant255(P.p[1] * (g + P.p[0] - 128.0f) + 128.0f);
        o[2] = quant255(P.p[1] * (b + P.p[0] - 128.0f) + 128.0f);
    } else if constexpr (OP == PFXK_RHAI_HSL) { // p0 = hue_shift/360, p1 = sat_factor, p2 = light_offset
        const float rn = div255(r), gn = div255(g), bn = div255(b);
        const float cmax = __builtin_fmaxf(__builtin_fmaxf(rn, gn), bn), cmin = __builtin_fminf(__builtin_fminf(rn, gn), bn);
        const float l = (cmax + cmin) / 2.0f;
        float h = 0.0f, s = 0.0f;
        if (!(__builtin_fabsf(cmax - cmin) < 1e-10f)) { // k/255 inputs: distinct values differ by >= 1/255 (same ranges as rgb_to_hsl)
            const float d = cmax - cmin;
            s = fdiv_fast(d, (l > 0.5f) ? (2.0f - cmax - cmin) : (cmax + cmin));
            const rdiv kd = rdiv_prepare(d);
            float hh;
            if (__builtin_fabsf(cmax - rn) < 1e-10f) hh = rdiv_apply(kd, gn - bn) + ((gn < bn) ? 6.0f : 0.0f);
            else if (__builtin_fabsf(cmax - gn) < 1e-10f) hh = rdiv_apply(kd, bn - rn) + 2.0f;
            else hh = rdiv_apply(kd, rn - gn) + 4.0f;
            h = fdiv_fast(hh, 6.0f);
        }
        float nh = h + P.p[0];
        { // f32::rem_euclid(1.0)
            const float rr = nh - __builtin_truncf(nh); // fmod(x, 1.0) == x - trunc(x), exact
            nh = (rr < 0.0f) ? rr + 1.0f : rr;
        }
        const float ns = rs_clamp(s * P.p[1], 0.0f, 1.0f);
        const rgb3 n = hsl_to_rgb<true>(nh, ns, l);
        o[0] = quant255(n.r * 255.0f + P.p[2]);
        o[1] = quant255(n.g * 255.0f + P.p[2]);
        o[2] = quant255(n.b * 255.0f + P.p[2]);
    } else if constexpr (OP == PFXK_RHAI_EXPOSURE) { // p0 = gain
        o[0] = quant255(r * P.p[0]); o[1] = quant255(g * P.p[0]); o[2] = quant255(b * P.p[0]);
    } else if constexpr (OP == PFXK_RHAI_LEVELS) { // lut[256], applied to r,g,b
        o[0] = (float)lut[px & 0xffu]; o[1] = (float)lut[(px >> 8) & 0xffu]; o[2] = (float)lut[(px >> 16) & 0xffu];
    } else { o[0] = r; o[1] = g; o[2] = b; }
}

template <int OP, bool RHAI>
PFX_DEV uint32_t apply_px(const pfxk_params& P, const uint8_t* __restrict__ lut, uint32_t px)
{
    float o[4];
    if constexpr (RHAI) {
        rhai_px<OP>(P, lut, px, o);
        return pack_rgba(o[0], o[1], o[2], o[3]);
    } else {
        adjust_px<OP>(P, lut, ubyte0(px), ubyte1(px), ubyte2(px), ubyte3(px), o);
        if constexpr (OP == PFXK_OP_INVERT_ALPHA || OP == PFXK_OP_LUT_RGBA) return pack_round_rgba(o[0], o[1], o[2], o[3]);
        else if ((OP == PFXK_OP_HSL || OP == PFXK_OP_VIBRANCE) && P.p[11] != 0.0f) { // wave-uniform: the host found every parameter finite
            // hsl_to_rgb returns finite values for finite h, s, l (sums and products of numbers in [0, 2]; the grey lanes' 0 / 0 never leaves
            // rgb_to_hsl): no +inf to keep away from the tie bit, so the v_med3 of round_tie_prep (a half-rate instruction) is not needed
            auto tie = [](float v) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) | 1u); };
            uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32(tie(o[0]), 0, px);
            q = __builtin_amdgcn_cvt_pk_u8_f32(tie(o[1]), 1, q);
            return __builtin_amdgcn_cvt_pk_u8_f32(tie(o[2]), 2, q);
        } else { // alpha passes through untouched: round three channels, keep the byte
            uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[0]), 0, px);
            q = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[1]), 1, q);
            return __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[2]), 2, q);
        }
    }
}

// grid = one block per 64x64 chunk; lane = (cg = tid % 16 -> 4 px, rows tid/16 + {0,16,32,48})
template <int OP, bool RHAI>
__global__ __launch_bounds__(256) void pointwise_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        const uint8_t* __restrict__ mask, const uint8_t* __restrict__ lut_g,
                                                        pfxk_params P, int sparse_mode, uint32_t w, uint32_t h)
{
    __shared__ uint8_t lut[1024];
    constexpr bool USES_LUT = RHAI ? (OP == PFXK_RHAI_LEVELS) : (OP == PFXK_OP_GRADIENT_MAP || OP == PFXK_OP_LUT_RGBA);
    if constexpr (USES_LUT) {
        reinterpret_cast<uint32_t*>(lut)[threadIdx.x] = reinterpret_cast<const uint32_t*>(lut_g)[threadIdx.x];
        __syncthreads();
    }
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const uint32_t bx = (bid % cxn) * 64u, by = (bid / cxn) * 64u;
    const uint32_t cg = threadIdx.x & 15u, r0 = threadIdx.x >> 4;
    const uint32_t x = bx + cg * 4u;
    const bool vec = (w & 3u) == 0u; // rows 16-byte aligned -> one dwordx4 per row
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);

    if (!mask && vec && bx + 64u <= w && by + 64u <= h) {
        // interior chunk without a selection mask (every chunk of an 8K frame but the bottom row's): no per-pixel bounds or mask
        // logic, so the 16 pixels of a lane are straight-line code — the general path below spends ~5 exec-mask branches per pixel
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const uint4*>(s32 + (size_t)(by + r0 + 16u * k) * w + x);
        int any_in = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) any_in |= (int)((v[k].x | v[k].y | v[k].z | v[k].w) >> 24);
        const bool live = sparse_mode == 2 ? __syncthreads_or(any_in) != 0 : true; // IN_PLACE: an unpopulated chunk is never visited
        int any_out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (live) {
                v[k].x = apply_px<OP, RHAI>(P, lut, v[k].x); v[k].y = apply_px<OP, RHAI>(P, lut, v[k].y);
                v[k].z = apply_px<OP, RHAI>(P, lut, v[k].z); v[k].w = apply_px<OP, RHAI>(P, lut, v[k].w);
            } else v[k] = make_uint4(0u, 0u, 0u, 0u);
            any_out |= (int)((v[k].x | v[k].y | v[k].z | v[k].w) >> 24);
        }
        const bool drop = sparse_mode == 1 ? __syncthreads_or(any_out) == 0 : false; // FROM_FLAT: a chunk whose alpha is all zero is dropped
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4*>(d32 + (size_t)(by + r0 + 16u * k) * w + x) = drop ? make_uint4(0u, 0u, 0u, 0u) : v[k];
        return;
    }

    uint32_t in[4][4], out[4][4];
    bool ok[4][4];
    int any_in = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = by + r0 + 16u * k;
        const size_t rowoff = (size_t)y * w + x;
#pragma unroll
        for (int p = 0; p < 4; ++p) { ok[k][p] = (y < h) && (x + p < w); in[k][p] = 0u; }
        if (y < h && x < w) {
            if (vec) {
                const uint4 v = *reinterpret_cast<const uint4*>(s32 + rowoff);
                in[k][0] = v.x; in[k][1] = v.y; in[k][2] = v.z; in[k][3] = v.w;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) if (ok[k][p]) in[k][p] = s32[rowoff + p];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) any_in |= (int)(in[k][p] >> 24);
    }
    bool live = true;
    if (sparse_mode == 2) live = __syncthreads_or(any_in) != 0; // IN_PLACE: unpopulated chunk is never visited

    int any_out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = by + r0 + 16u * k;
        uint32_t m4 = 0x01010101u;
        if (mask && y < h && x < w) {
            const size_t mo = (size_t)y * w + x;
            if (vec) m4 = *reinterpret_cast<const uint32_t*>(mask + mo);
            else { m4 = 0; for (int p = 0; p < 4; ++p) if (ok[k][p]) m4 |= (uint32_t)mask[mo + p] << (8 * p); }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool sel = ((m4 >> (8 * p)) & 0xffu) != 0u;
            uint32_t v = 0u;
            if (live) v = sel ? apply_px<OP, RHAI>(P, lut, in[k][p]) : in[k][p];
            out[k][p] = ok[k][p] ? v : 0u;
            any_out |= (int)(out[k][p] >> 24);
        }
    }
    if (sparse_mode == 1) { // FROM_FLAT: from_rgba_image(&out) drops a chunk whose alpha is all zero
        if (__syncthreads_or(any_out) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int p = 0; p < 4; ++p) out[k][p] = 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = by + r0 + 16u * k;
        if (!(y < h && x < w)) continue;
        const size_t rowoff = (size_t)y * w + x;
        if (vec) *reinterpret_cast<uint4*>(d32 + rowoff) = make_uint4(out[k][0], out[k][1], out[k][2], out[k][3]);
        else {
#pragma unroll
            for (int p = 0; p < 4; ++p) if (ok[k][p]) d32[rowoff + p] = out[k][p];
        }
    }
}

template <int OP, bool RHAI>
hipError_t launch(hipStream_t s, const uint8_t* src, uint8_t* dst, const uint8_t* mask, const uint8_t* lut,
                  const pfxk_params& P, int sparse_mode, uint32_t w, uint32_t h)
{
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    pointwise_kernel<OP, RHAI><<<nchunks, 256, 0, s>>>(src, dst, mask, lut, P, sparse_mode, w, h);
    return hipGetLastError();
}

} // namespace

extern "C" hipError_t pfxk_adjust(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask,
                                  const uint8_t* d_lut, int op, const pfxk_params* P, int sparse_mode, uint32_t w,
                                  uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    switch (op) {
#define PFX_CASE(OP) case OP: return launch<OP, false>(s, d_src, d_dst, d_mask, d_lut, *P, sparse_mode, w, h);
        PFX_CASE(PFXK_OP_INVERT) PFX_CASE(PFXK_OP_INVERT_ALPHA) PFX_CASE(PFXK_OP_SEPIA)
        PFX_CASE(PFXK_OP_BRIGHTNESS_CONTRAST) PFX_CASE(PFXK_OP_HSL) PFX_CASE(PFXK_OP_EXPOSURE)
        PFX_CASE(PFXK_OP_HIGHLIGHTS_SHADOWS) PFX_CASE(PFXK_OP_TEMPERATURE_TINT) PFX_CASE(PFXK_OP_THRESHOLD)
        PFX_CASE(PFXK_OP_POSTERIZE) PFX_CASE(PFXK_OP_COLOR_BALANCE) PFX_CASE(PFXK_OP_GRADIENT_MAP)
        PFX_CASE(PFXK_OP_BLACK_AND_WHITE) PFX_CASE(PFXK_OP_VIBRANCE) PFX_CASE(PFXK_OP_LUT_RGBA)
        PFX_CASE(PFXK_OP_DESATURATE)
#undef PFX_CASE
    default: return hipErrorInvalidValue;
    }
}

extern "C" hipError_t pfxk_rhai_adjust(hipStream_t s, uint8_t* d_px, const uint8_t* d_lut, int op,
                                       const pfxk_params* P, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    switch (op) {
#define PFX_CASE(OP) case OP: return launch<OP, true>(s, d_px, d_px, nullptr, d_lut, *P, 0, w, h);
        PFX_CASE(PFXK_RHAI_INVERT) PFX_CASE(PFXK_RHAI_DESATURATE) PFX_CASE(PFXK_RHAI_SEPIA)
        PFX_CASE(PFXK_RHAI_SEPIA_STRENGTH) PFX_CASE(PFXK_RHAI_BRIGHTNESS_CONTRAST) PFX_CASE(PFXK_RHAI_HSL)
        PFX_CASE(PFXK_RHAI_EXPOSURE) PFX_CASE(PFXK_RHAI_LEVELS)
#undef PFX_CASE
    default: return hipErrorInvalidValue;
    }
}
