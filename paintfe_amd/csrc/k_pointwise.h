// k_pointwise.h — the per-pixel functions of the pointwise adjustment bank (both CPU numeric flavours), shared by the pointwise kernels (k_pointwise.hip)
// and by the kernels that run a chain of these ops as an epilogue of their store (k_gauss_exact.hip).  References: k_pointwise.hip's header.
#pragma once
#include "k_common.h"
#include "pfx_kernels.h"

namespace pfxk {
namespace pw {

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


// ---- chains (round 6, pfx_chain_dev): N pixels of a lane through the ops of a chain, one op at a time (an op is wave-uniform: its dispatch is a scalar
// branch, its body the same straight-line code the single-op kernels run; every op re-quantises to u8, so the result equals launching the ops one by one).
// `luts`: LDS (or global) copy of the chain's lookup tables, 1024 bytes per slot.
template <int N, int OP, bool RHAI>
PFX_DEV void chain_step(const pfxk_params& P, const uint8_t* __restrict__ lut, uint32_t (&px)[N])
{
#pragma unroll
    for (int k = 0; k < N; ++k) px[k] = apply_px<OP, RHAI>(P, lut, px[k]);
}
// The chain is always the kernel's FIRST parameter and is read through the kernarg segment (constant address space, wave-uniform addresses: scalar loads into
// SGPRs).  Indexing the by-value parameter itself with the run-time op index makes hipcc copy the whole struct to scratch and read the parameters back per lane
// through the vector memory path — in the matrix-core Gaussian those reads share vmcnt with the source prefetch and serialise it (measured: 0.78 ms against 0.22).
typedef const pfxk_chain __attribute__((address_space(4)))* chain_kptr;
PFX_DEV chain_kptr chain_in_kernarg() { return (chain_kptr)__builtin_amdgcn_kernarg_segment_ptr(); }
template <int N>
PFX_DEV void chain_apply(chain_kptr C, const uint8_t* __restrict__ luts, uint32_t (&px)[N])
{
    const uint32_t n = C->n;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t code = C->op[i];
        // the parameter block is fetched INSIDE each case: only the fields that op reads are loaded (dead loads go), so a chain costs a handful of scalar
        // registers in the kernels that host it instead of twelve
        auto params = [&]() { pfxk_params P;
#pragma unroll
            for (int k = 0; k < 12; ++k) P.p[k] = C->P[i].p[k];
            return P; };
        switch (code) {
#define PFX_CA(OP) case OP: chain_step<N, OP, false>(params(), luts + 1024u * C->lut_slot[i], px); break;
#define PFX_CR(OP) case PFXK_CHAIN_RHAI | OP: chain_step<N, OP, true>(params(), luts + 1024u * C->lut_slot[i], px); break;
            PFX_CA(PFXK_OP_INVERT) PFX_CA(PFXK_OP_INVERT_ALPHA) PFX_CA(PFXK_OP_SEPIA) PFX_CA(PFXK_OP_BRIGHTNESS_CONTRAST) PFX_CA(PFXK_OP_HSL)
            PFX_CA(PFXK_OP_EXPOSURE) PFX_CA(PFXK_OP_HIGHLIGHTS_SHADOWS) PFX_CA(PFXK_OP_TEMPERATURE_TINT) PFX_CA(PFXK_OP_THRESHOLD)
            PFX_CA(PFXK_OP_POSTERIZE) PFX_CA(PFXK_OP_COLOR_BALANCE) PFX_CA(PFXK_OP_GRADIENT_MAP) PFX_CA(PFXK_OP_BLACK_AND_WHITE)
            PFX_CA(PFXK_OP_VIBRANCE) PFX_CA(PFXK_OP_LUT_RGBA) PFX_CA(PFXK_OP_DESATURATE)
            PFX_CR(PFXK_RHAI_INVERT) PFX_CR(PFXK_RHAI_DESATURATE) PFX_CR(PFXK_RHAI_SEPIA) PFX_CR(PFXK_RHAI_SEPIA_STRENGTH)
            PFX_CR(PFXK_RHAI_BRIGHTNESS_CONTRAST) PFX_CR(PFXK_RHAI_HSL) PFX_CR(PFXK_RHAI_EXPOSURE) PFX_CR(PFXK_RHAI_LEVELS)
#undef PFX_CA
#undef PFX_CR
        default: break;
        }
    }
}

} // namespace pw
} // namespace pfxk
