// k_flatten.hip — the layer compositor: CanvasState::composite() as ONE streaming gfx950 kernel.
//
// Reference: src/canvas/canvas_state.rs:505-698 (composite_viewport) and :1246-1505 (blend_pixel_static and the
// per-channel helpers); adjustment layers src/canvas/layers.rs:276-325; live mask :660-665.
//
// Design (streaming, no LDS, no MFMA):
//   * one lane owns 4 consecutive pixels: every layer is read with one 16-byte load per lane (1 KiB per wave
//     instruction, fully coalesced), the result is written with one 16-byte store;
//   * the accumulator ("pixels[idx]" in the reference, a u8 RGBA re-quantised after every layer) lives in
//     registers for the whole layer stack as integer-valued floats — nothing but the N layer reads and the one
//     result write touches HBM: 4*N + 4 bytes per pixel, the algorithmic minimum;
//   * the blend mode is uniform per layer, so the 25-way dispatch is a scalar branch outside the pixel code and
//     each mode is its own straight-line specialisation;
//   * arithmetic is the reference's f32 sequence, operation for operation, without FMA contraction
//     (this file is compiled with -ffp-contract=off); u8/255 uses the proved 2-op form (k_common.h:div255);
//   * the three `/ out_a` of a pixel share one refined reciprocal (rdiv below): the exact operation sequence
//     hipcc emits for an IEEE f32 divide, with the per-denominator part hoisted — bit-identical to `/`.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

enum : uint32_t {
    M_NORMAL = 0, M_MULTIPLY, M_SCREEN, M_ADDITIVE, M_REFLECT, M_GLOW, M_COLOR_BURN, M_COLOR_DODGE, M_OVERLAY,
    M_DIFFERENCE, M_NEGATION, M_LIGHTEN, M_DARKEN, M_XOR, M_OVERWRITE, M_HARD_LIGHT, M_SOFT_LIGHT, M_EXCLUSION,
    M_SUBTRACT, M_DIVIDE, M_LINEAR_BURN, M_VIVID_LIGHT, M_LINEAR_LIGHT, M_PIN_LIGHT, M_HARD_MIX
};

// ---- correctly rounded division with a shared denominator -------------------------------------------------
// hipcc lowers `n / d` (f32, IEEE) to: div_scale x2, rcp, 2 FMAs refining the reciprocal, mul + 4 FMAs refining the
// quotient, div_fmas, div_fixup.  div_scale / div_fmas scaling / div_fixup only act when an operand or the quotient
// is denormal, huge, zero-denominator or NaN.  For this kernel's operands (numerators in [0, ~1], denominators in
// [2^-48, 1]: guaranteed by the host, which selects the FAST=false instantiation when a layer opacity is a positive
// value below 2^-40) they are identities, so the sequence below produces the same bits with the reciprocal part
// computed once per denominator.  tests/test_gpu_parity.py::test_fast_division_matches_ieee checks 2^28 operand
// pairs against `/` on the device; every golden / oracle parity test runs through this path.
// (rdiv / rdiv_prepare / rdiv_apply live in k_common.h)
template <bool FAST> PFX_DEV float fdiv(float n, float d)
{
    if constexpr (FAST) return rdiv_apply(rdiv_prepare(d), n);
    else return n / d;
}

// ---- canvas_state.rs:1425-1505 ----
PFX_DEV float overlay_channel(float base, float top)
{
    return (base < 0.5f) ? 2.0f * base * top : 1.0f - 2.0f * (1.0f - base) * (1.0f - top);
}
template <bool F> PFX_DEV float color_burn_channel(float base, float top)
{
    return (top == 0.0f) ? 0.0f : __builtin_fmaxf(1.0f - fdiv<F>(1.0f - base, top), 0.0f);
}
template <bool F> PFX_DEV float color_dodge_channel(float base, float top)
{
    return (top >= 1.0f) ? 1.0f : __builtin_fminf(fdiv<F>(base, 1.0f - top), 1.0f);
}
template <bool F> PFX_DEV float reflect_channel(float base, float top)
{
    return (top >= 1.0f) ? 1.0f : __builtin_fminf(fdiv<F>(base * base, 1.0f - top), 1.0f);
}
PFX_DEV float soft_light_channel(float base, float top)
{
    if (top <= 0.5f) return base - (1.0f - 2.0f * top) * base * (1.0f - base);
    float d = (base <= 0.25f) ? ((16.0f * base - 12.0f) * base + 4.0f) * base : __builtin_sqrtf(base);
    return base + (2.0f * top - 1.0f) * (d - base);
}
template <bool F> PFX_DEV float divide_channel(float base, float top)
{
    return (top <= 0.0f) ? 1.0f : __builtin_fminf(fdiv<F>(base, top), 1.0f);
}
template <bool F> PFX_DEV float vivid_light_channel(float base, float top)
{
    // canvas_state.rs:1479-1497.  Both branches divide; the operands are selected first so that a lane pays for one
    // division (the branch is per-lane data, so "both sides" is what a divergent wave would execute anyway).
    const bool lo = (top <= 0.5f);
    const float t2 = lo ? 2.0f * top : 2.0f * (top - 0.5f);
    const float n = lo ? 1.0f - base : base;
    const float d = lo ? t2 : 1.0f - t2;
    const float q = fdiv<F>(n, d);
    const float burn = (t2 <= 0.0f) ? 0.0f : __builtin_fmaxf(1.0f - q, 0.0f);
    const float dodge = (t2 >= 1.0f) ? 1.0f : __builtin_fminf(q, 1.0f);
    return lo ? burn : dodge;
}
PFX_DEV float pin_light_channel(float base, float top)
{
    return (top <= 0.5f) ? __builtin_fminf(base, 2.0f * top) : __builtin_fmaxf(base, 2.0f * (top - 0.5f));
}

template <uint32_t M, bool F>
PFX_DEV float blend_fn(float b, float t)
{
    if constexpr (M == M_NORMAL) return t;
    else if constexpr (M == M_MULTIPLY) return b * t;
    else if constexpr (M == M_SCREEN) return 1.0f - (1.0f - b) * (1.0f - t);
    else if constexpr (M == M_ADDITIVE) return __builtin_fminf(b + t, 1.0f);
    else if constexpr (M == M_REFLECT) return reflect_channel<F>(b, t);
    else if constexpr (M == M_GLOW) return reflect_channel<F>(t, b);
    else if constexpr (M == M_COLOR_BURN) return color_burn_channel<F>(b, t);
    else if constexpr (M == M_COLOR_DODGE) return color_dodge_channel<F>(b, t);
    else if constexpr (M == M_OVERLAY) return overlay_channel(b, t);
    else if constexpr (M == M_DIFFERENCE) return __builtin_fabsf(b - t);
    else if constexpr (M == M_NEGATION) return 1.0f - __builtin_fabsf(1.0f - b - t);
    else if constexpr (M == M_LIGHTEN) return __builtin_fmaxf(b, t);
    else if constexpr (M == M_DARKEN) return __builtin_fminf(b, t);
    else if constexpr (M == M_HARD_LIGHT) return overlay_channel(t, b);
    else if constexpr (M == M_SOFT_LIGHT) return soft_light_channel(b, t);
    else if constexpr (M == M_EXCLUSION) return b + t - 2.0f * b * t;
    else if constexpr (M == M_SUBTRACT) return __builtin_fmaxf(b - t, 0.0f);
    else if constexpr (M == M_DIVIDE) return divide_channel<F>(b, t);
    else if constexpr (M == M_LINEAR_BURN) return __builtin_fmaxf(b + t - 1.0f, 0.0f);
    else if constexpr (M == M_VIVID_LIGHT) return vivid_light_channel<F>(b, t);
    else if constexpr (M == M_LINEAR_LIGHT) return rs_clamp(b + 2.0f * t - 1.0f, 0.0f, 1.0f);
    else if constexpr (M == M_PIN_LIGHT) return pin_light_channel(b, t);
    else if constexpr (M == M_HARD_MIX) return (b + t >= 1.0f) ? 1.0f : 0.0f;
    else return t;
}

// Rust `(v * 255.0).clamp(0.0, 255.0) as u8`, kept as an integer-valued float.  A -0.0 result is harmless: div255(-0.0)
// is +0.0 and (uint32_t)(-0.0f) is 0.
// CLAMP=false drops the clamp where it is provably the identity: every quotient q = n/d of blend_pixel_static with
// d > 0 satisfies 0 <= q <= 1 + 3 ulp (all 23 separable blend functions return values in [0, 1] in f32 — checked
// function by function in DESIGN.md §flatten — so 0 <= n <= d(1 + 2 ulp)); then 0 <= q*255 < 255.001 and
// trunc() alone yields the clamped value.  The FAST=false instantiation keeps the clamp.
template <bool CLAMP>
PFX_DEV float q255(float v)
{
    if constexpr (CLAMP) return __builtin_truncf(__builtin_fminf(__builtin_fmaxf(v * 255.0f, 0.0f), 255.0f));
    else return __builtin_truncf(v * 255.0f);
}

// One blend_pixel_static (canvas_state.rs:1246-1422).  `acc` = base as integer-valued floats (r,g,b,a);
// `top` = packed RGBA8 of the layer pixel (alpha already masked); `opacity_raw` = layer.opacity as stored,
// `opc` = opacity.clamp(0,1).
// Branch-free on purpose: per-lane early-outs diverge on real data (a wave almost never agrees), so the early
// returns of the reference become selects at the end; the discarded lanes may hold NaN/Inf (0/0), never stored.
// OB ("opaque base", only with F): the caller has established wave-wide that acc alpha == 255.  Then base_a = 1.0 and
// out_a = fl(top_a + fl(1 - top_a)) is exactly 1.0 for every f32 top_a in [0, 1] (top_a >= 0.5: 1 - top_a is exact; below,
// fl(1 - top_a) is off by at most 2^-25, and 1 +- 2^-25 rounds to 1.0, ties to even), base_c * 1.0 = base_c and n / 1.0 = n:
// the division, the alpha products and the alpha re-quantisation drop out with identical bits.  Typical documents (an opaque
// background under everything) run this path for every layer.  tests: test_flatten_opaque_base_path_bitexact.
// OB == 2: in addition the whole wave's top pixels are opaque and the layer opacity is >= 1 (a photo or texture layer with a
// blend mode at 100 %): top_a = div255(255) * 1.0 = 1.0, 1 - top_a = 0, so n = f * 1.0 + base * 0.0 = f and the pixel is
// `(f(base, top) * 255) as u8` with alpha 255; Normal is the reference's own early-out (:1258), the top pixel itself.
template <uint32_t M, bool F, int OB = 0>
PFX_DEV void blend_px(float (&acc)[4], uint32_t top, float opacity_raw, float opc)
{
    if constexpr (OB == 2 && F && M != M_XOR && M != M_OVERWRITE) {
        const float t0 = ubyte0(top), t1 = ubyte1(top), t2 = ubyte2(top);
        if constexpr (M == M_NORMAL) { acc[0] = t0; acc[1] = t1; acc[2] = t2; }
        else {
            const float r = blend_fn<M, F>(div255(acc[0]), div255(t0));
            const float g = blend_fn<M, F>(div255(acc[1]), div255(t1));
            const float b = blend_fn<M, F>(div255(acc[2]), div255(t2));
            acc[0] = q255<false>(r); acc[1] = q255<false>(g); acc[2] = q255<false>(b);
        }
        acc[3] = 255.0f;
        return;
    }
    const uint32_t ta8 = top >> 24;
    const bool skip = (ta8 == 0u);                                     // :1253  -> keep base
    const float t0 = ubyte0(top), t1 = ubyte1(top), t2 = ubyte2(top), t3 = (float)ta8;
    const float top_r = div255(t0), top_g = div255(t1), top_b = div255(t2);
    const float top_a = div255(t3) * opc;                              // :1272
    float o0, o1, o2, o3;
    constexpr bool CL = !F; // the FAST instantiation runs only when every layer opacity clamps into [2^-40, 1]
    if constexpr (M == M_OVERWRITE) {                                  // :1275 (`as u8` without clamp == with clamp)
        o0 = q255<CL>(top_r); o1 = q255<CL>(top_g); o2 = q255<CL>(top_b); o3 = q255<CL>(top_a);
    } else {
        constexpr bool UNIT = OB != 0 && F && M != M_XOR; // out_a == 1.0 exactly, see above
        const float base_r = div255(acc[0]), base_g = div255(acc[1]), base_b = div255(acc[2]);
        const float base_a = (OB != 0 && F) ? 1.0f : div255(acc[3]);
        const float ita = 1.0f - top_a;
        float den, nr, ng, nb;
        if constexpr (UNIT) {
            const float r = blend_fn<M, F>(base_r, top_r);
            const float g = blend_fn<M, F>(base_g, top_g);
            const float b = blend_fn<M, F>(base_b, top_b);
            den = 1.0f;
            nr = r * top_a + base_r * ita;
            ng = g * top_a + base_g * ita;
            nb = b * top_a + base_b * ita;
        } else if constexpr (M == M_XOR) {                                    // :1283
            const float iba = 1.0f - base_a;
            den = base_a * ita + top_a * iba;
            nr = base_r * base_a * ita + top_r * top_a * iba;
            ng = base_g * base_a * ita + top_g * top_a * iba;
            nb = base_b * base_a * ita + top_b * top_a * iba;
        } else {
            const float r = blend_fn<M, F>(base_r, top_r);
            const float g = blend_fn<M, F>(base_g, top_g);
            const float b = blend_fn<M, F>(base_b, top_b);
            den = top_a + base_a * ita;                                // :1407
            nr = r * top_a + base_r * base_a * ita;                    // :1412
            ng = g * top_a + base_g * base_a * ita;
            nb = b * top_a + base_b * base_a * ita;
        }
        float qr, qg, qb;
        if constexpr (UNIT) { qr = nr; qg = ng; qb = nb; }
        else if constexpr (F) { const rdiv k = rdiv_prepare(den); qr = rdiv_apply(k, nr); qg = rdiv_apply(k, ng); qb = rdiv_apply(k, nb); }
        else { qr = nr / den; qg = ng / den; qb = nb / den; }
        o0 = q255<CL>(qr); o1 = q255<CL>(qg); o2 = q255<CL>(qb); o3 = UNIT ? 255.0f : q255<CL>(den);
        // :1285 / :1408 `den == 0 -> (0,0,0,0)`.  With opacity > 0 (FAST precondition) a non-skipped pixel has
        // top_a > 0, hence out_a = top_a + base_a*(1-top_a) > 0: the check can only fire for Xor (both opaque).
        if constexpr (!F || M == M_XOR) {
            const bool zero = (den == 0.0f);
            o0 = zero ? 0.0f : o0; o1 = zero ? 0.0f : o1; o2 = zero ? 0.0f : o2; o3 = zero ? 0.0f : o3;
        }
    }
    if constexpr (M == M_NORMAL) {
        if (opacity_raw >= 1.0f) {                                     // uniform; :1258 opaque overwrite
            const bool opaque = (ta8 == 255u);
            o0 = opaque ? t0 : o0; o1 = opaque ? t1 : o1; o2 = opaque ? t2 : o2; o3 = opaque ? 255.0f : o3;
        }
    }
    acc[0] = skip ? acc[0] : o0; acc[1] = skip ? acc[1] : o1; acc[2] = skip ? acc[2] : o2; acc[3] = skip ? acc[3] : o3;
}

template <uint32_t M, bool F, int PX, int OB>
PFX_DEV void blendN(float (&acc)[PX][4], const uint32_t (&top)[PX], float opacity_raw, float opc)
{
#pragma unroll
    for (int p = 0; p < PX; ++p) blend_px<M, F, OB>(acc[p], top[p], opacity_raw, opc);
}

template <bool F, int PX = 4, int OB = 0>
PFX_DEV void blend4_dispatch(uint32_t mode, float (&acc)[PX][4], const uint32_t (&top)[PX], float opacity_raw, float opc)
{
    switch (mode) { // wave-uniform: one scalar branch per layer
#define PFX_CASE(M) case M: blendN<M, F, PX, OB>(acc, top, opacity_raw, opc); break;
        PFX_CASE(M_NORMAL) PFX_CASE(M_MULTIPLY) PFX_CASE(M_SCREEN) PFX_CASE(M_ADDITIVE) PFX_CASE(M_REFLECT)
        PFX_CASE(M_GLOW) PFX_CASE(M_COLOR_BURN) PFX_CASE(M_COLOR_DODGE) PFX_CASE(M_OVERLAY) PFX_CASE(M_DIFFERENCE)
        PFX_CASE(M_NEGATION) PFX_CASE(M_LIGHTEN) PFX_CASE(M_DARKEN) PFX_CASE(M_XOR) PFX_CASE(M_OVERWRITE)
        PFX_CASE(M_HARD_LIGHT) PFX_CASE(M_SOFT_LIGHT) PFX_CASE(M_EXCLUSION) PFX_CASE(M_SUBTRACT) PFX_CASE(M_DIVIDE)
        PFX_CASE(M_LINEAR_BURN) PFX_CASE(M_VIVID_LIGHT) PFX_CASE(M_LINEAR_LIGHT) PFX_CASE(M_PIN_LIGHT)
        PFX_CASE(M_HARD_MIX)
#undef PFX_CASE
    default: blendN<M_NORMAL, F, PX, OB>(acc, top, opacity_raw, opc); break; // BlendMode::from_u8 fallback, layers.rs:183
    }
}

// the streaming kernels' per-layer entry: picks the opaque-base specialisation when the whole wave's accumulators are opaque
template <int PX>
PFX_DEV void blend_layer_fast(uint32_t mode, float (&acc)[PX][4], const uint32_t (&top)[PX], float opacity_raw)
{
    const float opc = rs_clamp(opacity_raw, 0.0f, 1.0f);
    bool ob = true, ot = opacity_raw >= 1.0f;
#pragma unroll
    for (int p = 0; p < PX; ++p) { ob = ob && (acc[p][3] == 255.0f); ot = ot && (top[p] >> 24) == 255u; }
    if (__all(ob)) {
        if (__all(ot)) blend4_dispatch<true, PX, 2>(mode, acc, top, opacity_raw, opc);
        else blend4_dispatch<true, PX, 1>(mode, acc, top, opacity_raw, opc);
    } else blend4_dispatch<true, PX, 0>(mode, acc, top, opacity_raw, opc);
}

// live layer mask: top.a = (a * (255 - conceal)) / 255, integer (canvas_state.rs:660-665)
PFX_DEV uint32_t apply_conceal(uint32_t px, uint32_t conceal)
{
    if (conceal == 0u) return px;
    uint32_t a = ((px >> 24) * (255u - conceal)) / 255u;
    return (px & 0x00ffffffu) | (a << 24);
}

// AdjustmentLayerData::apply_to_pixel_with_opacity (layers.rs:276-325) on one accumulator pixel
PFX_DEV void adjust_px(float (&p)[4], uint32_t kind, const float* __restrict__ adj, float opacity)
{
    float o[4] = {p[0], p[1], p[2], p[3]};
    switch (kind) {
    case PFXK_ADJ_EXPOSURE: { // adj[0] = gain = powf(2, ev), computed on the host with glibc like the reference
        const float gain = adj[0];
        o[0] = quant255(p[0] * gain); o[1] = quant255(p[1] * gain); o[2] = quant255(p[2] * gain);
        break;
    }
    case PFXK_ADJ_BRIGHTNESS_CONTRAST: { // adj[0]=brightness, adj[1]=factor (host-computed)
        const float br = adj[0], factor = adj[1];
        o[0] = quant255(factor * (p[0] + br - 128.0f) + 128.0f);
        o[1] = quant255(factor * (p[1] + br - 128.0f) + 128.0f);
        o[2] = quant255(factor * (p[2] + br - 128.0f) + 128.0f);
        break;
    }
    case PFXK_ADJ_INVERT: o[0] = 255.0f - p[0]; o[1] = 255.0f - p[1]; o[2] = 255.0f - p[2]; break;
    case PFXK_ADJ_CHANNEL_MIXER:
#pragma unroll
        for (int c = 0; c < 4; ++c)
            o[c] = quant255(p[0] * adj[c * 4 + 0] + p[1] * adj[c * 4 + 1] + p[2] * adj[c * 4 + 2] + p[3] * adj[c * 4 + 3]);
        break;
    default: break;
    }
    const float t = rs_clamp(opacity, 0.0f, 1.0f), inv = 1.0f - t;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = round_u8f(p[c] * inv + o[c] * t); // `.round() as u8`
}

// Tool preview folded into the active layer's pixel before masking / compositing (canvas_state.rs:621-658): `top` is the layer
// pixel, `pp` the preview pixel.  Uses the plain-divide instantiation: this path is interactive-only and tiny.
PFX_DEV uint32_t preview_apply(uint32_t top, uint32_t pp, const pfxk_preview& PV)
{
    if (PV.replaces) return pp;                                        // :623
    if ((pp >> 24) == 0u) return top;                                  // :625
    if (PV.is_eraser) {                                                // :626-631
        const float mask_strength = div255(ubyte3(pp)), current_a = div255(ubyte3(top));
        const float new_a = __builtin_fmaxf(current_a * (1.0f - mask_strength), 0.0f);
        return (top & 0x00ffffffu) | ((uint32_t)quant255(new_a * 255.0f) << 24);
    }
    float b[1][4] = {{ubyte0(top), ubyte1(top), ubyte2(top), ubyte3(top)}};
    const uint32_t t1[1] = {pp};
    blend4_dispatch<false, 1>(PV.mode, b, t1, 1.0f, 1.0f);             // blend_pixel_static(top, pp, preview_blend, 1.0)
    if (PV.mode == M_OVERWRITE || PV.mode == M_XOR) {                  // :632-653 coverage-weighted lerp
        const float cov = div255(ubyte3(pp)), inv = 1.0f - cov;
        return pack_rgba(quant255(ubyte0(top) * inv + b[0][0] * cov + 0.5f), quant255(ubyte1(top) * inv + b[0][1] * cov + 0.5f),
                         quant255(ubyte2(top) * inv + b[0][2] * cov + 0.5f), quant255(ubyte3(top) * inv + b[0][3] * cov + 0.5f));
    }
    return pack_rgba(b[0][0], b[0][1], b[0][2], b[0][3]);
}

template <bool GENERAL, bool F>
__global__ __launch_bounds__(256) void flatten_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                      const float* __restrict__ adj_table,
                                                      const uint8_t* __restrict__ chunk_active, uint32_t w, uint32_t h,
                                                      uint8_t* __restrict__ dst, const pfxk_preview PV, const pfxk_region RG)
{
    // GENERAL + RG.rw: only the dirty rectangle is composited, into a compact rw x rh destination (composite_dirty_readback,
    // src/gpu/renderer.rs:588); quads then run along the rectangle's rows
    const bool region = GENERAL && RG.rw != 0u;
    const uint32_t qpr = region ? (RG.rw + 3u) / 4u : 0u;
    const size_t n_quads = region ? (size_t)qpr * RG.rh : ((size_t)w * h + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * blockDim.x) {
        size_t p0 = q * 4, n_px = (size_t)w * h, o0 = q * 4; // n_px: exclusive bound of valid pixel indices for this quad
        if (region) {
            const uint32_t ry = (uint32_t)(q / qpr), qx = (uint32_t)(q - (size_t)ry * qpr);
            p0 = (size_t)(RG.y0 + ry) * w + RG.x0 + qx * 4u;
            n_px = (size_t)(RG.y0 + ry) * w + RG.x0 + RG.rw;
            o0 = (size_t)ry * RG.rw + qx * 4u;
        }
        const bool full = p0 + 4 <= n_px && !(region && (((p0 | o0) & 3u) != 0u)); // 16-byte loads / stores need aligned quads
        float acc[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f; // :573

        bool act[4] = {true, true, true, true};
        if (GENERAL && chunk_active) { // adjustment layers only touch chunks populated in some visible layer (:529-550)
            const uint32_t cxn = (w + 63u) / 64u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                size_t pi = p0 + p;
                if (pi >= n_px) pi = n_px - 1;
                uint32_t y = (uint32_t)(pi / w), x = (uint32_t)(pi - (size_t)y * w);
                act[p] = chunk_active[(y >> 6) * cxn + (x >> 6)] != 0;
            }
        }

        if (!GENERAL && full && n_layers > 0) {
            // raster-only stack: software-pipelined — layer k+1's descriptor (scalar load) and 16-byte pixel quad
            // are requested before layer k is blended, so the ~100 VALU ops of a blend cover the HBM latency
            pfxk_layer_desc L = layers[0];
            uint4 v = *reinterpret_cast<const uint4*>(L.pixels + p0 * 4);
            for (uint32_t li = 0; li < n_layers; ++li) {
                pfxk_layer_desc Ln = L;
                uint4 vn = v;
                if (li + 1 < n_layers) {
                    Ln = layers[li + 1];
                    vn = *reinterpret_cast<const uint4*>(Ln.pixels + p0 * 4);
                }
                const uint32_t top[4] = {v.x, v.y, v.z, v.w};
                // wave-level early-out: a wave whose 256 pixels are all transparent in this layer (sparse layers of real
                // documents; the TiledImage analogue is a missing chunk, canvas_state.rs:600) skips the blend entirely
                if (__any(((v.x | v.y | v.z | v.w) >> 24) != 0u)) {
                    if constexpr (F) blend_layer_fast<4>(L.mode, acc, top, L.opacity);
                    else blend4_dispatch<F>(L.mode, acc, top, L.opacity, rs_clamp(L.opacity, 0.0f, 1.0f));
                }
                L = Ln;
                v = vn;
            }
        } else
        for (uint32_t li = 0; li < n_layers; ++li) {
            const pfxk_layer_desc L = layers[li]; // uniform -> scalar loads
            if (GENERAL && L.kind != PFXK_LAYER_RASTER) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (act[p]) adjust_px(acc[p], L.kind, adj_table + L.adj_off, L.opacity);
                continue;
            }
            uint32_t top[4];
            if (full) {
                const uint4 v = *reinterpret_cast<const uint4*>(L.pixels + p0 * 4);
                top[0] = v.x; top[1] = v.y; top[2] = v.z; top[3] = v.w;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    top[p] = (p0 + p < n_px) ? reinterpret_cast<const uint32_t*>(L.pixels)[p0 + p] : 0u;
            }
            if (GENERAL && PV.pixels && li == PV.active_pos) { // :593-597,621-658
                const uint32_t cxn = (w + 63u) / 64u;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const size_t pi = p0 + p;
                    if (pi < n_px) {
                        const uint32_t y = (uint32_t)(pi / w), x = (uint32_t)(pi - (size_t)y * w);
                        const uint32_t ci = (y >> 6) * cxn + (x >> 6);
                        if (PV.chunk_present[ci]) {
                            // a layer chunk that does not exist reads as (0,0,0,0), colour included (:613-617)
                            const uint32_t lp = PV.layer_chunk_present[ci] ? top[p] : 0u;
                            top[p] = preview_apply(lp, reinterpret_cast<const uint32_t*>(PV.pixels)[pi], PV);
                        }
                    }
                }
            }
            if (GENERAL && L.mask) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (p0 + p < n_px) top[p] = apply_conceal(top[p], L.mask[p0 + p]);
            }
            if constexpr (F) blend_layer_fast<4>(L.mode, acc, top, L.opacity); // lanes past the image edge hold alpha 0: they only disable the opaque path
            else blend4_dispatch<F>(L.mode, acc, top, L.opacity, rs_clamp(L.opacity, 0.0f, 1.0f));
        }

        uint32_t out[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) out[p] = pack_rgba(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        if (full) {
            *reinterpret_cast<uint4*>(dst + o0 * 4) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p0 + p < n_px) reinterpret_cast<uint32_t*>(dst)[o0 + p] = out[p];
        }
    }
}

// Raster-only, FAST-precondition stacks whose pixel count is a multiple of PX: the streaming core without the general
// kernel's tail/mask/adjustment handling.  PX pixels per lane (PX*4-byte loads), MINW = waves/SIMD the register
// allocator must allow.  Variants are selected with pfx_tune("flatten_variant", v) for measurement; all produce
// identical results (same blend_px).
template <int PX> struct px_vec;
template <> struct px_vec<4> { using type = uint4; };
template <> struct px_vec<2> { using type = uint2; };
template <> struct px_vec<1> { using type = uint32_t; };

template <int PX, int MINW>
__global__ __launch_bounds__(256, MINW) void flatten_fast_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                                 size_t n_groups, uint8_t* __restrict__ dst)
{
    using V = typename px_vec<PX>::type;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_groups; q += (size_t)gridDim.x * blockDim.x) {
        float acc[PX][4];
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f;
        pfxk_layer_desc L = layers[0];
        V v = reinterpret_cast<const V*>(L.pixels)[q];
        for (uint32_t li = 0; li < n_layers; ++li) {
            pfxk_layer_desc Ln = L;
            V vn = v;
            if (li + 1 < n_layers) { // prefetch layer li+1 before blending layer li
                Ln = layers[li + 1];
                vn = reinterpret_cast<const V*>(Ln.pixels)[q];
            }
            uint32_t top[PX];
            __builtin_memcpy(top, &v, sizeof top);
            uint32_t any_a = 0;
#pragma unroll
            for (int p = 0; p < PX; ++p) any_a |= top[p];
            if (__any((any_a >> 24) != 0u)) blend_layer_fast<PX>(L.mode, acc, top, L.opacity);
            L = Ln;
            v = vn;
        }
        uint32_t out[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) out[p] = pack_rgba(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        V o;
        __builtin_memcpy(&o, out, sizeof out);
        reinterpret_cast<V*>(dst)[q] = o;
    }
}

int g_flatten_variant = 0; // tuning knob (pfxk_flatten_set_variant)

// chunk activity = union over visible raster layers of "chunk has any alpha != 0" (canvas_state.rs:529-550)
__global__ __launch_bounds__(256) void chunk_active_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                           uint32_t w, uint32_t h, uint8_t* __restrict__ chunk_active,
                                                           const uint8_t* __restrict__ preview_present)
{
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t cx = blockIdx.x % cxn, cy = blockIdx.x / cxn;
    const uint32_t bx = cx * 64u, by = cy * 64u;
    const uint32_t cw = min(64u, w - bx), ch = min(64u, h - by);
    int any = preview_present ? (preview_present[blockIdx.x] != 0) : 0; // the preview's chunk keys join the set (:541-548)
    for (uint32_t li = 0; li < n_layers && !any; ++li) {
        const pfxk_layer_desc L = layers[li];
        if (L.kind != PFXK_LAYER_RASTER || !L.pixels) continue;
        int mine = 0;
        for (uint32_t i = threadIdx.x; i < cw * ch; i += blockDim.x) {
            uint32_t lx = i % cw, ly = i / cw;
            mine |= L.pixels[((size_t)(by + ly) * w + bx + lx) * 4 + 3] != 0;
        }
        any = __syncthreads_or(mine);
    }
    if (threadIdx.x == 0) chunk_active[blockIdx.x] = (uint8_t)(any != 0);
}

// dst[i] = blend_pixel_static(base[i], top[i], mode, opacity) — element-wise form (spot checks)
template <bool F>
__global__ __launch_bounds__(256) void blend_arrays_kernel(const uint32_t* __restrict__ base, const uint32_t* __restrict__ top,
                                                           uint32_t* __restrict__ dst, size_t n, uint32_t mode, float opacity)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = base[i];
        float acc[4][4];
        acc[0][0] = ubyte0(b); acc[0][1] = ubyte1(b); acc[0][2] = ubyte2(b); acc[0][3] = ubyte3(b);
#pragma unroll
        for (int p = 1; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f;
        const uint32_t t[4] = {top[i], 0u, 0u, 0u};
        blend4_dispatch<F>(mode, acc, t, opacity, rs_clamp(opacity, 0.0f, 1.0f));
        dst[i] = pack_rgba(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    }
}

// Stroke commit (ref: src/ui/panels/tools/behavior/raster/bezier_commit.rs:103-225):
//   brush : layer = blend_pixel_static(layer, preview, mode, 1.0) where preview.a > 0 and selection != 0
//   eraser: layer.a = ((a/255) * (1 - m/255)).max(0) * 255 as u8 where m = preview.a > 0
__global__ __launch_bounds__(256) void brush_commit_kernel(uint32_t* __restrict__ layer, const uint32_t* __restrict__ preview,
                                                           const uint8_t* __restrict__ selection, size_t n, uint32_t mode,
                                                           int is_eraser)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (selection && selection[i] == 0) continue;
        const uint32_t pp = preview[i];
        if ((pp >> 24) == 0u) continue;
        const uint32_t lp = layer[i];
        if (is_eraser) {
            const float mask_strength = div255(ubyte3(pp));
            const float current_a = div255(ubyte3(lp));
            const float new_a = __builtin_fmaxf(current_a * (1.0f - mask_strength), 0.0f);
            layer[i] = (lp & 0x00ffffffu) | ((uint32_t)trunc_u8f(new_a * 255.0f) << 24);
        } else {
            float acc[4][4];
            acc[0][0] = ubyte0(lp); acc[0][1] = ubyte1(lp); acc[0][2] = ubyte2(lp); acc[0][3] = ubyte3(lp);
#pragma unroll
            for (int p = 1; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f;
            const uint32_t t[4] = {pp, 0u, 0u, 0u};
            blend4_dispatch<true>(mode, acc, t, 1.0f, 1.0f);
            layer[i] = pack_rgba(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
        }
    }
}

// device-side check of rdiv against the compiler's IEEE divide: out[0] += number of mismatching pairs
__global__ __launch_bounds__(256) void rdiv_check_kernel(uint64_t seed, uint32_t iters, unsigned long long* out)
{
    uint64_t s = seed + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t a = (uint32_t)(s >> 33), b = (uint32_t)(s >> 7);
        // numerator in [0, 2), denominator in [2^-48, 1]: random mantissas, exponents in the kernel's operand range
        const float n = ((a & 0xC0u) == 0u) ? 0.0f : __builtin_bit_cast(float, ((a >> 8) & 0x007fffffu) | ((127u - (a & 31u)) << 23));
        const float d = __builtin_bit_cast(float, (b & 0x007fffffu) | ((127u - ((b >> 23) % 49u)) << 23));
        const float q_ref = n / d;
        const float q_fast = rdiv_apply(rdiv_prepare(d), n);
        bad += (__builtin_bit_cast(uint32_t, q_ref) != __builtin_bit_cast(uint32_t, q_fast));
    }
    if (bad) atomicAdd(out, bad);
}

} // namespace

extern "C" void pfxk_flatten_set_variant(int v) { g_flatten_variant = v; }

extern "C" hipError_t pfxk_rdiv_check(hipStream_t s, uint64_t seed, uint32_t blocks, uint32_t iters, unsigned long long* d_out)
{
    rdiv_check_kernel<<<blocks, 256, 0, s>>>(seed, iters, d_out);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_blend_arrays(hipStream_t s, const uint8_t* d_base, const uint8_t* d_top, uint8_t* d_dst,
                                        size_t n_px, uint32_t mode, float opacity, int fast_div)
{
    if (n_px == 0) return hipSuccess;
    size_t blocks = (n_px + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (fast_div)
        blend_arrays_kernel<true><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_base, (const uint32_t*)d_top,
                                                                   (uint32_t*)d_dst, n_px, mode, opacity);
    else
        blend_arrays_kernel<false><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_base, (const uint32_t*)d_top,
                                                                    (uint32_t*)d_dst, n_px, mode, opacity);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_brush_commit(hipStream_t s, uint8_t* d_layer, const uint8_t* d_preview,
                                        const uint8_t* d_selection, uint32_t w, uint32_t h, uint32_t mode, int is_eraser)
{
    const size_t n = (size_t)w * h;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    brush_commit_kernel<<<(uint32_t)blocks, 256, 0, s>>>((uint32_t*)d_layer, (const uint32_t*)d_preview, d_selection, n, mode,
                                                         is_eraser);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_flatten(hipStream_t stream, const pfxk_layer_desc* d_layers, uint32_t n_layers,
                                   const float* d_adj_table, int general, int fast_div, uint8_t* d_chunk_active,
                                   uint32_t w, uint32_t h, uint8_t* d_dst, const pfxk_preview* preview, const pfxk_region* region)
{
    size_t n_quads = ((size_t)w * h + 3) / 4;
    if (n_quads == 0) return hipSuccess;
    pfxk_preview PV{};
    if (preview && preview->pixels) { PV = *preview; general = 1; }
    pfxk_region RG{};
    if (region && region->rw && region->rh) { RG = *region; general = 1; n_quads = (size_t)((RG.rw + 3u) / 4u) * RG.rh; }
    if (general && d_chunk_active) {
        const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
        chunk_active_kernel<<<nchunks, 256, 0, stream>>>(d_layers, n_layers, w, h, d_chunk_active, PV.pixels ? PV.chunk_present : nullptr);
    }
    const uint32_t block = 256;
    const size_t cap = 256u * 8u * 4u; // 256 CUs x 8 blocks, x4 waves of grid-stride work granularity
    const size_t n_px = (size_t)w * h;
    if (!general && fast_div && n_layers > 0 && (n_px & 3u) == 0 && g_flatten_variant > 0) {
        auto grid = [&](size_t groups) { size_t b = (groups + block - 1) / block; return (uint32_t)(b > cap ? cap : b); };
        switch (g_flatten_variant) {
        case 1: flatten_fast_kernel<4, 1><<<grid(n_px / 4), block, 0, stream>>>(d_layers, n_layers, n_px / 4, d_dst); break;
        case 2: flatten_fast_kernel<4, 8><<<grid(n_px / 4), block, 0, stream>>>(d_layers, n_layers, n_px / 4, d_dst); break;
        case 3: flatten_fast_kernel<2, 1><<<grid(n_px / 2), block, 0, stream>>>(d_layers, n_layers, n_px / 2, d_dst); break;
        case 4: flatten_fast_kernel<2, 8><<<grid(n_px / 2), block, 0, stream>>>(d_layers, n_layers, n_px / 2, d_dst); break;
        case 5: flatten_fast_kernel<1, 8><<<grid(n_px), block, 0, stream>>>(d_layers, n_layers, n_px, d_dst); break;
        default: flatten_fast_kernel<4, 4><<<grid(n_px / 4), block, 0, stream>>>(d_layers, n_layers, n_px / 4, d_dst); break;
        }
        return hipGetLastError();
    }
    size_t blocks = (n_quads + block - 1) / block;
    if (blocks > cap) blocks = cap;
    const uint32_t g = (uint32_t)blocks;
#define PFX_LAUNCH(G, F) flatten_kernel<G, F><<<g, block, 0, stream>>>(d_layers, n_layers, d_adj_table, (G) ? d_chunk_active : nullptr, w, h, d_dst, PV, RG)
    if (general) { if (fast_div) PFX_LAUNCH(true, true); else PFX_LAUNCH(true, false); }
    else         { if (fast_div) PFX_LAUNCH(false, true); else PFX_LAUNCH(false, false); }
#undef PFX_LAUNCH
    return hipGetLastError();
}
