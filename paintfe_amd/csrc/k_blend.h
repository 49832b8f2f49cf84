// k_blend.h — CanvasState::blend_pixel_static (src/canvas/canvas_state.rs:1246-1505) as branch-free gfx950 device code.
// Shared by the compositor kernels (k_flatten.hip), the stroke commit and the per-mode instruction table tool
// (tools/isa_table.py -> profiles/rNN_blend_isa.md).
#pragma once
#include "k_common.h"

namespace pfxk {

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
    return (top == 0.0f) ? 0.0f : clamp01(1.0f - fdiv<F>(1.0f - base, top)); // 1 - q <= 1
}
template <bool F> PFX_DEV float color_dodge_channel(float base, float top)
{
    return (top >= 1.0f) ? 1.0f : clamp01(fdiv<F>(base, 1.0f - top)); // q >= 0
}
template <bool F> PFX_DEV float reflect_channel(float base, float top)
{
    return (top >= 1.0f) ? 1.0f : clamp01(fdiv<F>(base * base, 1.0f - top));
}
// v_max_f32 / v_min_f32 as ONE instruction: fmaxf / fminf make hipcc canonicalise both operands first (two more v_max x, x) because
// it cannot prove that values read from memory are not signalling NaNs; the blend operands are bytes / 255.
PFX_DEV float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
PFX_DEV float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// three-operand forms: min / max instructions issue in 4 cycles whatever their operand count (profiles/r03_valu_rates.txt)
PFX_DEV float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PFX_DEV float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <int PX> PFX_DEV float alpha_min(const float (&v)[PX][4])
{
    if constexpr (PX == 3) return vmin3(v[0][3], v[1][3], v[2][3]);
    else { float m = v[0][3]; for (int p = 1; p < PX; ++p) m = vmin(m, v[p][3]); return m; }
}
template <int PX> PFX_DEV float alpha_max(const float (&v)[PX][4])
{
    if constexpr (PX == 3) return vmax3(v[0][3], v[1][3], v[2][3]);
    else { float m = v[0][3]; for (int p = 1; p < PX; ++p) m = vmax(m, v[p][3]); return m; }
}

// Correctly rounded sqrt for a normal positive argument: the core of the sequence hipcc emits for sqrtf
// (-fhip-fp32-correctly-rounded-divide-sqrt) — hardware estimate, its two neighbours, two exact residuals, two selects — without the
// scaling of tiny arguments and the zero / inf / NaN pass-through around it, which are identities for an argument in (0.25, 1].
PFX_DEV float sqrt_normal(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) - 1u), sp = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

PFX_DEV float soft_light_channel(float base, float top)
{
    if (top <= 0.5f) return base - (1.0f - 2.0f * top) * base * (1.0f - base);
    // base <= 0.25 takes the polynomial; the sqrt lane value for such a base is discarded (and sqrt_normal(0) is a harmless 0 or NaN)
    float d = (base <= 0.25f) ? ((16.0f * base - 12.0f) * base + 4.0f) * base : sqrt_normal(base);
    return base + (2.0f * top - 1.0f) * (d - base);
}
// Soft Light's d(base) depends on the accumulator channel alone, and that is one of 256 values RN(k / 255): the streaming compositor
// keeps the 256 results in LDS (1 KB per workgroup, filled by the kernel's own lanes with the very expression above, so the bits are
// those the lanes would compute) and replaces v_sqrt + refinement + polynomial + select — about 50 issue cycles per channel — by
// one fma, one and, one ds_read_b32.  bn * 1020 is 4k(1 +- 2^-24), so fl(bn * 1020 + 2^23) is exactly 2^23 + 4k: the byte offset.
__shared__ float s_soft_d[256];
PFX_DEV void soft_d_fill(uint32_t tid, uint32_t nthreads)
{
    for (uint32_t k = tid; k < 256u; k += nthreads) {
        const float base = div255((float)k);
        s_soft_d[k] = (base <= 0.25f) ? ((16.0f * base - 12.0f) * base + 4.0f) * base : sqrt_normal(base);
    }
}
PFX_DEV float soft_light_channel_lds(float base, float top)
{
    const uint32_t off = __builtin_bit_cast(uint32_t, __builtin_fmaf(base, 1020.0f, 8388608.0f)) & 0x3fcu;
    const float d = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s_soft_d) + off);
    if (top <= 0.5f) return base - (1.0f - 2.0f * top) * base * (1.0f - base);
    return base + (2.0f * top - 1.0f) * (d - base);
}
template <bool F> PFX_DEV float divide_channel(float base, float top)
{
    return (top <= 0.0f) ? 1.0f : clamp01(fdiv<F>(base, top));
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
    const float burn = (t2 <= 0.0f) ? 0.0f : clamp01(1.0f - q);
    const float dodge = (t2 >= 1.0f) ? 1.0f : clamp01(q);
    return lo ? burn : dodge;
}
PFX_DEV float pin_light_channel(float base, float top)
{
    return (top <= 0.5f) ? vmin(base, 2.0f * top) : vmax(base, 2.0f * (top - 0.5f));
}

template <uint32_t M, bool F>
PFX_DEV float blend_fn(float b, float t)
{
    if constexpr (M == M_NORMAL) return t;
    else if constexpr (M == M_MULTIPLY) return b * t;
    else if constexpr (M == M_SCREEN) return 1.0f - (1.0f - b) * (1.0f - t);
    else if constexpr (M == M_ADDITIVE) return clamp01(b + t);               // min(b + t, 1), b + t >= 0
    else if constexpr (M == M_REFLECT) return reflect_channel<F>(b, t);
    else if constexpr (M == M_GLOW) return reflect_channel<F>(t, b);
    else if constexpr (M == M_COLOR_BURN) return color_burn_channel<F>(b, t);
    else if constexpr (M == M_COLOR_DODGE) return color_dodge_channel<F>(b, t);
    else if constexpr (M == M_OVERLAY) return overlay_channel(b, t);
    else if constexpr (M == M_DIFFERENCE) return __builtin_fabsf(b - t);
    else if constexpr (M == M_NEGATION) return 1.0f - __builtin_fabsf(1.0f - b - t);
    else if constexpr (M == M_LIGHTEN) return vmax(b, t);
    else if constexpr (M == M_DARKEN) return vmin(b, t);
    else if constexpr (M == M_HARD_LIGHT) return overlay_channel(t, b);
    else if constexpr (M == M_SOFT_LIGHT) return soft_light_channel(b, t);
    else if constexpr (M == M_EXCLUSION) return b + t - 2.0f * b * t;
    else if constexpr (M == M_SUBTRACT) return clamp01(b - t);               // max(b - t, 0), b - t <= 1
    else if constexpr (M == M_DIVIDE) return divide_channel<F>(b, t);
    else if constexpr (M == M_LINEAR_BURN) return clamp01(b + t - 1.0f);
    else if constexpr (M == M_VIVID_LIGHT) return vivid_light_channel<F>(b, t);
    else if constexpr (M == M_LINEAR_LIGHT) return clamp01(b + 2.0f * t - 1.0f);
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

// ---- normalised-accumulator form (the streaming compositor, k_flatten.hip:flatten_stream_kernel) -----------------------------
// Same arithmetic as blend_px<M, true, OB>, different representation: the layer pixel arrives as four f32 already equal to
// RN(byte / 255) — gfx950's typed buffer load (buffer_load_format_xyzw, 8_8_8_8 UNORM) performs exactly that conversion in the
// texture path (tools/lab/typed_load.hip checks all 256 byte values on every channel against `/ 255.0f`) — and the u8
// accumulator of the reference (canvas_state.rs:573) is held as bn = RN(k / 255), the value the next layer would compute from
// the stored byte k anyway.  Per layer-pixel this removes the four v_cvt_f32_ubyteN and eight div255 operations of the layer
// pixel; re-quantisation `k = (q * 255) as u8` followed by `k / 255` of the next blend is requant() below.
PFX_DEV float requant(float q) { return div255(__builtin_truncf(q * 255.0f)); }

// blend function of the normalised form: Soft Light through the LDS table (the kernel must have run soft_d_fill + a barrier)
template <uint32_t M> PFX_DEV float blend_fn_nx(float b, float t)
{
    if constexpr (M == M_SOFT_LIGHT) return soft_light_channel_lds(b, t);
    else return blend_fn<M, true>(b, t);
}

template <uint32_t M, int OB>
PFX_DEV void blend_nx(float (&acc)[4], const float (&top)[4], float opacity_raw, float opc)
{
    constexpr bool F = true;
    if constexpr (OB == 2 && M != M_XOR && M != M_OVERWRITE) { // opaque accumulator, opaque layer pixel, opacity >= 1 (wave-uniform)
        if constexpr (M == M_NORMAL) { acc[0] = top[0]; acc[1] = top[1]; acc[2] = top[2]; }
        else {
            const float r = blend_fn_nx<M>(acc[0], top[0]);
            const float g = blend_fn_nx<M>(acc[1], top[1]);
            const float b = blend_fn_nx<M>(acc[2], top[2]);
            acc[0] = requant(r); acc[1] = requant(g); acc[2] = requant(b);
        }
        acc[3] = 1.0f;
        return;
    }
    const bool skip = (top[3] == 0.0f);                                // :1253  -> keep base
    const float top_r = top[0], top_g = top[1], top_b = top[2];
    const float top_a = top[3] * opc;                                  // :1272
    float o0, o1, o2, o3;
    if constexpr (M == M_OVERWRITE) {                                  // :1275
        o0 = requant(top_r); o1 = requant(top_g); o2 = requant(top_b); o3 = requant(top_a);
    } else {
        constexpr bool UNIT = OB != 0 && M != M_XOR;                   // out_a == 1.0 exactly (see blend_px)
        const float base_r = acc[0], base_g = acc[1], base_b = acc[2];
        const float base_a = (OB != 0) ? 1.0f : acc[3];
        const float ita = 1.0f - top_a;
        float den, nr, ng, nb;
        if constexpr (UNIT) {
            const float r = blend_fn_nx<M>(base_r, top_r);
            const float g = blend_fn_nx<M>(base_g, top_g);
            const float b = blend_fn_nx<M>(base_b, top_b);
            den = 1.0f;
            nr = r * top_a + base_r * ita;
            ng = g * top_a + base_g * ita;
            nb = b * top_a + base_b * ita;
        } else if constexpr (M == M_XOR) {                             // :1283
            const float iba = 1.0f - base_a;
            den = base_a * ita + top_a * iba;
            nr = base_r * base_a * ita + top_r * top_a * iba;
            ng = base_g * base_a * ita + top_g * top_a * iba;
            nb = base_b * base_a * ita + top_b * top_a * iba;
        } else {
            const float r = blend_fn_nx<M>(base_r, top_r);
            const float g = blend_fn_nx<M>(base_g, top_g);
            const float b = blend_fn_nx<M>(base_b, top_b);
            den = top_a + base_a * ita;                                // :1407
            nr = r * top_a + base_r * base_a * ita;                    // :1412
            ng = g * top_a + base_g * base_a * ita;
            nb = b * top_a + base_b * base_a * ita;
        }
        if constexpr (UNIT) { o0 = requant(nr); o1 = requant(ng); o2 = requant(nb); o3 = 1.0f; }
        else {
            const rdiv k = rdiv_prepare(den);
            o0 = requant(rdiv_apply(k, nr)); o1 = requant(rdiv_apply(k, ng)); o2 = requant(rdiv_apply(k, nb)); o3 = requant(den);
        }
        if constexpr (M == M_XOR) {                                    // :1285 out_a == 0 -> (0,0,0,0)
            const bool zero = (den == 0.0f);
            o0 = zero ? 0.0f : o0; o1 = zero ? 0.0f : o1; o2 = zero ? 0.0f : o2; o3 = zero ? 0.0f : o3;
        }
    }
    // :1258, Normal at opacity >= 1 over an opaque layer pixel returns `top`.  No select is needed for it in this representation: top_a =
    // 1.0 * 1.0, ita = 0, so n = top_c * 1 + x * 0 = top_c, den = 1 + base_a * 0 = 1, n / 1 = n (rdiv: y = 1, residual 0) and
    // requant(RN(k / 255)) = RN(k / 255) for all 256 k (tests/test_host_logic.py::test_requant_is_identity_on_byte_values): the general
    // formula already yields the layer pixel, bit for bit.
    (void)opacity_raw;
    if constexpr (OB != 0 && M != M_XOR && M != M_OVERWRITE) {
        // Opaque accumulator: the :1253 early-out needs no select either.  A transparent layer pixel has top_a = 0 * opc = 0, ita = 1, so
        // n = r * 0 + base_c * 1 = base_c for every finite r (all 23 blend functions return finite values in [0, 1], rare sides included),
        // and requant(base_c) = base_c as above; alpha stays 1.0 in this path anyway.
        (void)skip;
        acc[0] = o0; acc[1] = o1; acc[2] = o2;
    } else {
        acc[0] = skip ? acc[0] : o0; acc[1] = skip ? acc[1] : o1; acc[2] = skip ? acc[2] : o2; acc[3] = skip ? acc[3] : o3;
    }
}

#ifndef PFX_XSKIP
#define PFX_XSKIP 0
#endif
template <uint32_t M, int PX, int OB>
PFX_DEV void blendN_nx(float (&acc)[PX][4], const float (&top)[PX][4], float opacity_raw, float opc)
{
#pragma unroll
    for (int p = 0; p < PX; ++p) {
#if PFX_XSKIP
        // lanes whose layer pixel is transparent (:1253) sit the pixel's blend out under EXEC instead of computing it and discarding the
        // result: the kernel runs at the chip's power limit, so an idle lane is clock the others get
        if (OB == 2 || top[p][3] != 0.0f) blend_nx<M, OB>(acc[p], top[p], opacity_raw, opc);
#else
        blend_nx<M, OB>(acc[p], top[p], opacity_raw, opc);
#endif
    }
}

template <int PX, int OB>
PFX_DEV void blend_nx_dispatch(uint32_t mode, float (&acc)[PX][4], const float (&top)[PX][4], float opacity_raw, float opc)
{
    switch (mode) { // wave-uniform: one scalar branch per layer
#define PFX_CASE(M) case M: blendN_nx<M, PX, OB>(acc, top, opacity_raw, opc); break;
        PFX_CASE(M_NORMAL) PFX_CASE(M_MULTIPLY) PFX_CASE(M_SCREEN) PFX_CASE(M_ADDITIVE) PFX_CASE(M_REFLECT)
        PFX_CASE(M_GLOW) PFX_CASE(M_COLOR_BURN) PFX_CASE(M_COLOR_DODGE) PFX_CASE(M_OVERLAY) PFX_CASE(M_DIFFERENCE)
        PFX_CASE(M_NEGATION) PFX_CASE(M_LIGHTEN) PFX_CASE(M_DARKEN) PFX_CASE(M_XOR) PFX_CASE(M_OVERWRITE)
        PFX_CASE(M_HARD_LIGHT) PFX_CASE(M_SOFT_LIGHT) PFX_CASE(M_EXCLUSION) PFX_CASE(M_SUBTRACT) PFX_CASE(M_DIVIDE)
        PFX_CASE(M_LINEAR_BURN) PFX_CASE(M_VIVID_LIGHT) PFX_CASE(M_LINEAR_LIGHT) PFX_CASE(M_PIN_LIGHT)
        PFX_CASE(M_HARD_MIX)
#undef PFX_CASE
    default: blendN_nx<M_NORMAL, PX, OB>(acc, top, opacity_raw, opc); break; // BlendMode::from_u8 fallback, layers.rs:183
    }
}

// one layer on PX pixels per lane: picks the opaque-accumulator specialisations wave-wide.  `opc` = the layer opacity clamped to [0, 1],
// prepared by the host (pfxk_layer_desc::adj_off of a raster layer): clamping a wave-uniform value on the VALU cost two half-rate
// instructions per wave and layer, and the raw value is only ever compared with 1.0 — `raw >= 1` <=> `clamp(raw) >= 1`.
template <int PX>
PFX_DEV void blend_layer_nx(uint32_t mode, float (&acc)[PX][4], const float (&top)[PX][4], float opc)
{
    const float amin = alpha_min<PX>(acc);
    if (__all(amin == 1.0f)) {
        const float tmin = alpha_min<PX>(top);
        if (opc >= 1.0f && __all(tmin == 1.0f)) blend_nx_dispatch<PX, 2>(mode, acc, top, opc, opc);
        else blend_nx_dispatch<PX, 1>(mode, acc, top, opc, opc);
    } else blend_nx_dispatch<PX, 0>(mode, acc, top, opc, opc);
}

// ---- per-group classes (k_flatten.hip: flatten_srt_kernel) ----
// The class-sorting compositor deals a unit's opaque accumulators to the leading 64-pixel groups (pixel p of every lane = group p).  LEAD = how many
// leading groups are opaque wave-wide: those take the OB = 1 arithmetic, the rest the general one.  (LEAD == PX is blend_nx_dispatch<PX, 1 | 2>.)
// One dispatch per value of LEAD: a scalar branch per group INSIDE each mode's case (every (mode, class) body once, 40 % less code) was measured
// too — 93 VGPRs instead of 80 and 7 % more executed VALU instructions, 4.6 % slower than round 3's kernel (profiles/r04_tuning.md).
template <uint32_t M, int PX, int LEAD>
PFX_DEV void blendN_nx_lead(float (&acc)[PX][4], const float (&top)[PX][4], float opacity_raw, float opc)
{
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (p < LEAD) blend_nx<M, 1>(acc[p], top[p], opacity_raw, opc);
        else blend_nx<M, 0>(acc[p], top[p], opacity_raw, opc);
    }
}
template <int PX, int LEAD>
PFX_DEV void blend_nx_dispatch_lead(uint32_t mode, float (&acc)[PX][4], const float (&top)[PX][4], float opacity_raw, float opc)
{
    switch (mode) { // wave-uniform: one scalar branch per layer
#define PFX_CASE(M) case M: blendN_nx_lead<M, PX, LEAD>(acc, top, opacity_raw, opc); break;
        PFX_CASE(M_NORMAL) PFX_CASE(M_MULTIPLY) PFX_CASE(M_SCREEN) PFX_CASE(M_ADDITIVE) PFX_CASE(M_REFLECT)
        PFX_CASE(M_GLOW) PFX_CASE(M_COLOR_BURN) PFX_CASE(M_COLOR_DODGE) PFX_CASE(M_OVERLAY) PFX_CASE(M_DIFFERENCE)
        PFX_CASE(M_NEGATION) PFX_CASE(M_LIGHTEN) PFX_CASE(M_DARKEN) PFX_CASE(M_XOR) PFX_CASE(M_OVERWRITE)
        PFX_CASE(M_HARD_LIGHT) PFX_CASE(M_SOFT_LIGHT) PFX_CASE(M_EXCLUSION) PFX_CASE(M_SUBTRACT) PFX_CASE(M_DIVIDE)
        PFX_CASE(M_LINEAR_BURN) PFX_CASE(M_VIVID_LIGHT) PFX_CASE(M_LINEAR_LIGHT) PFX_CASE(M_PIN_LIGHT)
        PFX_CASE(M_HARD_MIX)
#undef PFX_CASE
    default: blendN_nx_lead<M_NORMAL, PX, LEAD>(acc, top, opacity_raw, opc); break; // BlendMode::from_u8 fallback, layers.rs:183
    }
}
// `lead` = leading groups known to be opaque wave-wide (k_flatten.hip: srt_layers keeps the count up to date: an opaque accumulator stays opaque under
// every mode but Xor and Overwrite, so the count is only re-taken behind those, at re-deal attempts and at the start of a pass; a stale-low count is
// merely conservative).  ONE switch over (class variant, mode): with a dispatch per variant the variants' results met in different registers and the
// copies that reconcile them were 7 % of the executed VALU instructions (profiles/r04_tuning.md).
#if !defined(PFX_ONE_SWITCH)
#define PFX_ONE_SWITCH 1
#endif
template <int PX>
PFX_DEV void blend_layer_nx_groups(uint32_t mode, float (&acc)[PX][4], const float (&top)[PX][4], float opc, uint32_t lead)
{
    static_assert(PX == 2 || PX == 3, "two or three groups");
#if PFX_ONE_SWITCH
    uint32_t v = lead;                                   // 0 .. PX - 1: that many leading opaque groups
    if (lead == (uint32_t)PX) {
        const float tmin = alpha_min<PX>(top);
        v = (opc >= 1.0f && __all(tmin == 1.0f)) ? 4u : 3u; // 3: every group opaque, 4: and an opaque layer at 100 %
    }
    const uint32_t m = mode > 24u ? 0u : mode;           // BlendMode::from_u8 fallback, layers.rs:183
    switch (v * 32u + m) {
#define PFX_CASE(M) case 0u * 32u + M: blendN_nx<M, PX, 0>(acc, top, opc, opc); break; \
                    case 1u * 32u + M: blendN_nx_lead<M, PX, 1>(acc, top, opc, opc); break; \
                    case 2u * 32u + M: blendN_nx_lead<M, PX, (PX == 3 ? 2 : 1)>(acc, top, opc, opc); break; \
                    case 3u * 32u + M: blendN_nx<M, PX, 1>(acc, top, opc, opc); break; \
                    case 4u * 32u + M: blendN_nx<M, PX, 2>(acc, top, opc, opc); break;
        PFX_CASE(M_NORMAL) PFX_CASE(M_MULTIPLY) PFX_CASE(M_SCREEN) PFX_CASE(M_ADDITIVE) PFX_CASE(M_REFLECT)
        PFX_CASE(M_GLOW) PFX_CASE(M_COLOR_BURN) PFX_CASE(M_COLOR_DODGE) PFX_CASE(M_OVERLAY) PFX_CASE(M_DIFFERENCE)
        PFX_CASE(M_NEGATION) PFX_CASE(M_LIGHTEN) PFX_CASE(M_DARKEN) PFX_CASE(M_XOR) PFX_CASE(M_OVERWRITE)
        PFX_CASE(M_HARD_LIGHT) PFX_CASE(M_SOFT_LIGHT) PFX_CASE(M_EXCLUSION) PFX_CASE(M_SUBTRACT) PFX_CASE(M_DIVIDE)
        PFX_CASE(M_LINEAR_BURN) PFX_CASE(M_VIVID_LIGHT) PFX_CASE(M_LINEAR_LIGHT) PFX_CASE(M_PIN_LIGHT)
        PFX_CASE(M_HARD_MIX)
#undef PFX_CASE
    default: break;
    }
#else
    if (lead == 0u) blend_nx_dispatch<PX, 0>(mode, acc, top, opc, opc);
    else if (lead == (uint32_t)PX) {
        const float tmin = alpha_min<PX>(top);
        if (opc >= 1.0f && __all(tmin == 1.0f)) blend_nx_dispatch<PX, 2>(mode, acc, top, opc, opc);
        else blend_nx_dispatch<PX, 1>(mode, acc, top, opc, opc);
    } else if (PX == 3 && lead == 2u) blend_nx_dispatch_lead<PX, (PX == 3 ? 2 : 1)>(mode, acc, top, opc, opc);
    else blend_nx_dispatch_lead<PX, 1>(mode, acc, top, opc, opc);
#endif
}
// ---- two-stage form (round 5) ----
// The one-switch form above instantiates every (class variant, mode) pair: 125 bodies, 266 KB of code for three pixels per lane, a compare tree eight levels deep,
// and — what costs issue slots — a 125-way join whose twelve accumulator registers the register allocator no longer coalesces: every layer step ended in 12 to 24
// v_mov_b32 (7 - 13 % of the executed VALU instructions, profiles/r05_tuning.md).  blend_pixel_static factors: the blend function f(base_c, top_c) of the mode
// (:1302-1405) depends on no alpha, and the compositing around it (:1407-1421) on no mode.  Stage A switches over the mode and leaves f for the 3 PX colour
// channels in fresh registers (nothing to merge: the values are born in the cases); stage B switches over the class variant and updates the accumulators in
// place.  23 + 5 small bodies; the same IEEE operations on the same operands in the same order per pixel as blend_nx — only the instruction schedule changes.
// Xor and Overwrite have no f and keep their blend_nx bodies (a two-way branch in front).
template <int OB>
PFX_DEV void comp_nx(float (&acc)[4], const float (&top)[4], const float (&f)[3], float opc)
{
    if constexpr (OB == 2) { // opaque accumulator, opaque layer pixel, opacity >= 1 (wave-uniform).  Normal: requant(top_c) == top_c for byte values
        acc[0] = requant(f[0]); acc[1] = requant(f[1]); acc[2] = requant(f[2]); acc[3] = 1.0f;
        return;
    }
    const float top_a = top[3] * opc;                                  // :1272
    const float ita = 1.0f - top_a;
    if constexpr (OB == 1) {                                           // out_a == 1.0 exactly (see blend_px); no select for a transparent layer pixel (see blend_nx)
        acc[0] = requant(f[0] * top_a + acc[0] * ita);
        acc[1] = requant(f[1] * top_a + acc[1] * ita);
        acc[2] = requant(f[2] * top_a + acc[2] * ita);
    } else {
        const bool skip = (top[3] == 0.0f);                            // :1253  -> keep base
        const float base_a = acc[3];
        const float den = top_a + base_a * ita;                        // :1407
        const float nr = f[0] * top_a + acc[0] * base_a * ita;         // :1412
        const float ng = f[1] * top_a + acc[1] * base_a * ita;
        const float nb = f[2] * top_a + acc[2] * base_a * ita;
        const rdiv k = rdiv_prepare(den);
        const float o0 = requant(rdiv_apply(k, nr)), o1 = requant(rdiv_apply(k, ng)), o2 = requant(rdiv_apply(k, nb)), o3 = requant(den);
        acc[0] = skip ? acc[0] : o0; acc[1] = skip ? acc[1] : o1; acc[2] = skip ? acc[2] : o2; acc[3] = skip ? acc[3] : o3;
    }
}
template <uint32_t M, int PX>
PFX_DEV void fN_nx(float (&f)[PX][3], const float (&acc)[PX][4], const float (&top)[PX][4])
{
#pragma unroll
    for (int p = 0; p < PX; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[p][c] = blend_fn_nx<M>(acc[p][c], top[p][c]);
}
template <int PX, int LEAD, int OBL, int OBR> // pixels [0, LEAD) composite with OBL, the rest with OBR
PFX_DEV void compN_nx(float (&acc)[PX][4], const float (&top)[PX][4], const float (&f)[PX][3], float opc)
{
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        if (p < LEAD) comp_nx<OBL>(acc[p], top[p], f[p], opc);
        else comp_nx<OBR>(acc[p], top[p], f[p], opc);
    }
}
// `v`: 0 .. PX - 1 = that many leading opaque groups, 3 = every group opaque, 4 = and the layer's pixels are opaque at 100 % (blend_layer_nx_groups)
template <int PX>
PFX_DEV void blend_two_stage(uint32_t m, uint32_t v, float (&acc)[PX][4], const float (&top)[PX][4], float opc)
{
    if (m == M_XOR || m == M_OVERWRITE) {
        switch (v * 2u + (m == M_XOR ? 1u : 0u)) {
#define PFX_CASE2(V, BODY) case V * 2u: { constexpr uint32_t M = M_OVERWRITE; BODY; } break; case V * 2u + 1u: { constexpr uint32_t M = M_XOR; BODY; } break;
            PFX_CASE2(0u, (blendN_nx<M, PX, 0>(acc, top, opc, opc)))
            PFX_CASE2(1u, (blendN_nx_lead<M, PX, 1>(acc, top, opc, opc)))
            PFX_CASE2(2u, (blendN_nx_lead<M, PX, (PX == 3 ? 2 : 1)>(acc, top, opc, opc)))
            PFX_CASE2(3u, (blendN_nx<M, PX, 1>(acc, top, opc, opc)))
            PFX_CASE2(4u, (blendN_nx<M, PX, 2>(acc, top, opc, opc)))
#undef PFX_CASE2
        default: break;
        }
        return;
    }
    float f[PX][3];
    switch (m) {
#define PFX_CASE(M) case M: fN_nx<M, PX>(f, acc, top); break;
        PFX_CASE(M_MULTIPLY) PFX_CASE(M_SCREEN) PFX_CASE(M_ADDITIVE) PFX_CASE(M_REFLECT)
        PFX_CASE(M_GLOW) PFX_CASE(M_COLOR_BURN) PFX_CASE(M_COLOR_DODGE) PFX_CASE(M_OVERLAY) PFX_CASE(M_DIFFERENCE)
        PFX_CASE(M_NEGATION) PFX_CASE(M_LIGHTEN) PFX_CASE(M_DARKEN)
        PFX_CASE(M_HARD_LIGHT) PFX_CASE(M_SOFT_LIGHT) PFX_CASE(M_EXCLUSION) PFX_CASE(M_SUBTRACT) PFX_CASE(M_DIVIDE)
        PFX_CASE(M_LINEAR_BURN) PFX_CASE(M_VIVID_LIGHT) PFX_CASE(M_LINEAR_LIGHT) PFX_CASE(M_PIN_LIGHT)
        PFX_CASE(M_HARD_MIX)
#undef PFX_CASE
    default: fN_nx<M_NORMAL, PX>(f, acc, top); break;   // Normal, and BlendMode::from_u8's fallback (layers.rs:183)
    }
    switch (v) {
    case 0u: compN_nx<PX, 0, 0, 0>(acc, top, f, opc); break;
    case 1u: compN_nx<PX, 1, 1, 0>(acc, top, f, opc); break;
    case 2u: compN_nx<PX, (PX == 3 ? 2 : 1), 1, 0>(acc, top, f, opc); break;
    case 3u: compN_nx<PX, PX, 1, 1>(acc, top, f, opc); break;
    default: compN_nx<PX, PX, 2, 2>(acc, top, f, opc); break;
    }
}
#if !defined(PFX_TWO_STAGE)
#define PFX_TWO_STAGE 1
#endif
// one layer on PX groups (PX = 1: the early groups' one-pixel-per-lane pass), `lead` as in blend_layer_nx_groups
template <int PX>
PFX_DEV void blend_layer_nx_two_stage(uint32_t mode, float (&acc)[PX][4], const float (&top)[PX][4], float opc, uint32_t lead)
{
    uint32_t v = lead;
    if (lead == (uint32_t)PX) {
        const float tmin = alpha_min<PX>(top);
        v = (opc >= 1.0f && __all(tmin == 1.0f)) ? 4u : 3u;
    }
    blend_two_stage<PX>(mode > 24u ? 0u : mode, v, acc, top, opc);
}

// number of leading groups whose accumulators are all opaque (three compares and scalar counting)
template <int PX>
PFX_DEV uint32_t count_lead(const float (&acc)[PX][4])
{
    uint32_t lead = 0u;
    bool run = true;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        run = run && __popcll(__ballot(acc[p][3] == 1.0f)) == 64;
        lead += run ? 1u : 0u;
    }
    return lead;
}

} // namespace pfxk
