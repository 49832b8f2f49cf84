// k_effects2.hip — the rest of the effect bank (the Rhai Effect API's apply_noise / reduce_noise /
// crystallize / bulge / twist / vignette / halftone / ink / oil_painting plus the dialog-only render/glitch effects).
//
// Reference (file:line next to each pixel function): src/ops/effects.rs:53-161 (apply_per_pixel, sample_bilinear,
// hash_f32) and src/ops/effects/{blur,distort,noise,stylize,render,glitch,artistic,contours}.rs.
//
// One template kernel, one pixel per lane, 64x4 pixel tiles (a wave = 64 consecutive pixels of a row: coalesced 256 B
// loads/stores); the gathers (bilinear taps, windows) are served by L1/L2.  All are 4 B read + 4 B written per pixel
// algorithmically.  Every f32 expression is evaluated in the reference's association order with no contraction.
// Parity classes: integer / hash / sqrt / divide effects are bit-exact; the four that call libm per pixel (twist:
// sin+cos, gaussian noise: ln+cos, reduce_noise: exp, vignette: powf) evaluate the function in f64 and round once to
// f32, which equals glibc's f32 routine except where glibc itself is not correctly rounded (< 1 % of arguments, each
// moving an output byte with probability ~1e-5): +-1 LSB class, goldens reproduced with tolerance 0.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

PFX_DEV int rs_i32(float v) // `as i32`
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
PFX_DEV uint32_t rs_u32(float v) // `as u32`
{
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
PFX_DEV int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
PFX_DEV uint32_t pack_round(float r, float g, float b, float a) { return pack_round_rgba(r, g, b, a); }

// effects.rs:143-161
PFX_DEV uint32_t hash_u32(uint32_t x)
{
    x *= 0x9E3779B9u; x ^= x >> 16;
    x *= 0x85EBCA6Bu; x ^= x >> 13;
    x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
PFX_DEV float hash_f32(uint32_t x, uint32_t y, uint32_t seed)
{
    const uint32_t h = hash_u32(x * 374761393u + y * 668265263u + seed);
    return (float)(h & 0x00FFFFFFu) / 16777216.0f; // power of two: exact
}

// Value noise on the integer lattice (hashed corner values, quintic fade, bilinear blend), and its octave sum.  References: perlin_noise_2d
// src/ops/effects/noise.rs:53-71, turbulence src/ops/effects/distort.rs:229-246 — their operation order, which is the parity contract.
PFX_DEV float fade5(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
PFX_DEV float blend_to(float from, float to, float t) { return from + t * (to - from); }
PFX_DEV float perlin_noise_2d(float x, float y, uint32_t seed)
{
    const int cx = rs_i32(__builtin_floorf(x)), cy = rs_i32(__builtin_floorf(y));   // lattice cell
    const float wx = fade5(x - (float)cx), wy = fade5(y - (float)cy);
    const uint32_t ux = (uint32_t)cx, uy = (uint32_t)cy;
    const float lower = blend_to(hash_f32(ux, uy, seed), hash_f32(ux + 1u, uy, seed), wx);
    const float upper = blend_to(hash_f32(ux, uy + 1u, seed), hash_f32(ux + 1u, uy + 1u, seed), wx);
    return blend_to(lower, upper, wy);
}
PFX_DEV float turbulence_2d(float x, float y, uint32_t seed, uint32_t octaves, float roughness)
{
    float sum = 0.0f, weight = 1.0f, scale = 1.0f, weights = 0.0f;
    for (uint32_t o = 0; o < octaves; ++o, weight *= roughness, scale *= 2.0f) {
        sum += perlin_noise_2d(x * scale, y * scale, seed + o * 1000u) * weight;
        weights += weight;
    }
    return weights > 0.0f ? sum / weights : 0.0f;
}

// effects.rs:108-140 (clamp-to-edge, weight form p00(1-dx)(1-dy)+...; NOT the warp's bilinear)
PFX_DEV uint32_t texel(const uint32_t* __restrict__ img, int w, int h, int x, int y)
{
    return img[(size_t)clampi(y, 0, h - 1) * w + clampi(x, 0, w - 1)];
}
PFX_DEV uint32_t sample_bilinear_round(const uint32_t* __restrict__ img, int w, int h, float fx, float fy)
{
    const int x0 = rs_i32(__builtin_floorf(fx)), y0 = rs_i32(__builtin_floorf(fy));
    const int x1 = (int)((uint32_t)x0 + 1u), y1 = (int)((uint32_t)y0 + 1u);
    const float dx = fx - (float)x0, dy = fy - (float)y0;
    const uint32_t p00 = texel(img, w, h, x0, y0), p10 = texel(img, w, h, x1, y0), p01 = texel(img, w, h, x0, y1), p11 = texel(img, w, h, x1, y1);
    const float ax = 1.0f - dx, ay = 1.0f - dy;
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = (float)((p00 >> (8 * c)) & 0xffu), b = (float)((p10 >> (8 * c)) & 0xffu);
        const float cc = (float)((p01 >> (8 * c)) & 0xffu), d = (float)((p11 >> (8 * c)) & 0xffu);
        o[c] = a * ax * ay + b * dx * ay + cc * ax * dy + d * dx * dy;
    }
    return pack_round(o[0], o[1], o[2], o[3]);
}

PFX_DEV float rem_euclid(float a, float b) // f32::rem_euclid
{
    const float r = fmodf(a, b);
    return r < 0.0f ? r + __builtin_fabsf(b) : r;
}

// glibc's f32 routines evaluated through f64 (see the file header)
PFX_DEV float libm_cos(float x) { return (float)cos((double)x); }
PFX_DEV float libm_sin(float x) { return (float)sin((double)x); }
// expf the way glibc >= 2.27 computes it (sysdeps/ieee754/flt-32/e_expf.c, S. Nagy's algorithm): x * 32/ln2 split into an integer
// k and a remainder r in [-1/2, 1/2], 2^(k/32) from a 32-entry table of doubles with the exponent added into the bit pattern, a
// degree-3 polynomial in r, everything in f64, one rounding to f32 at the end.  Same table (2^(i/32) correctly rounded, minus
// i << 47), same coefficients, FMA-contracted like the variant glibc selects on FMA-capable x86 CPUs — so the weights of the
// bilateral filter are glibc's bits at about a fifth of the instructions of a full-precision f64 exp().
__device__ const unsigned long long EXP2F_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
PFX_DEV float libm_exp(float x)
{
    if (!(x >= -0x1.9fe368p6f)) return (x != x) ? x : 0.0f; // below: the result underflows to +0 (NaN propagates)
    if (x > 0x1.62e42ep6f) return __builtin_inff();           // above: overflow
    const double InvLn2N = 0x1.71547652b82fep+0 * 32.0, SHIFT = 0x1.8p52;
    const double C0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0, C1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0, C2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    const double z = InvLn2N * (double)x;
    double kd = z + SHIFT; // round to nearest integer, ties to even, in the low mantissa bits
    const unsigned long long ki = __builtin_bit_cast(unsigned long long, kd);
    kd -= SHIFT;
    const double r = z - kd;
    const unsigned long long t = EXP2F_TAB[ki & 31u] + (ki << 47);
    const double sc = __builtin_bit_cast(double, t);
    const double p = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double y = __builtin_fma(C2, r, 1.0);
    y = __builtin_fma(p, r2, y);
    return (float)(y * sc);
}
PFX_DEV float libm_log(float x) { return (float)log((double)x); }

// ---------------------------------------------------------------------------------------------------------------
template <int FX>
PFX_DEV uint32_t fx_pixel(const uint32_t* __restrict__ src, uint32_t s, int x, int y, int w, int h, const pfxk_fx_params& P)
{
    const float r = ubyte0(s), g = ubyte1(s), b = ubyte2(s), a = ubyte3(s);
    if constexpr (FX == PFXK_FX2_ZOOM) { // blur.rs:381-422; f: cx cy s inv_n max_dist tint*255[4] tint_strength; i0: n
        const float cx = P.f[0], cy = P.f[1], st = P.f[2];
        const int n = P.i[0];
        const float dx = (float)x - cx, dy = (float)y - cy;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f; // sums of integers < 2^24: exact in any order
        // samples in groups of four: the four gathers are issued (from the last sample's position past the end: always valid) before any is summed — with
        // one load per trip of the run-time-count loop every sample was a memory round trip of its own
        for (int i = 0; i < n; i += 4) {
            uint32_t p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ik = min(i + k, n - 1);
                const float t = 1.0f - st * ((float)ik / (float)(n - 1));
                const int sx = clampi(rs_i32(__builtin_roundf(cx + dx * t)), 0, w - 1);
                const int sy = clampi(rs_i32(__builtin_roundf(cy + dy * t)), 0, h - 1);
                p[k] = src[(size_t)sy * w + sx];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i + k < n) { s0 += ubyte0(p[k]); s1 += ubyte1(p[k]); s2 += ubyte2(p[k]); s3 += ubyte3(p[k]); }   // uniform
        }
        float v[4] = {s0 * P.f[3], s1 * P.f[3], s2 * P.f[3], s3 * P.f[3]};
        if (P.f[9] > 0.001f) {
            const float dist = __builtin_sqrtf(dx * dx + dy * dy);
            const float t = __builtin_fmaxf(1.0f - dist / P.f[4], 0.0f) * P.f[9];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = v[c] + (P.f[5 + c] - v[c]) * t;
        }
        return pack_round(v[0], v[1], v[2], v[3]);
    } else if constexpr (FX == PFXK_FX2_DENTS) { // distort.rs:268-309; f: inv_scale amount scale roughness; u0 seed; i: oct pinch wrap
        const float fx = (float)x * P.f[0], fy = (float)y * P.f[0];
        float nx = turbulence_2d(fx, fy, P.u[0], (uint32_t)P.i[0], P.f[3]) * 2.0f - 1.0f;
        float ny = turbulence_2d(fx, fy, P.u[0] + 9999u, (uint32_t)P.i[0], P.f[3]) * 2.0f - 1.0f;
        if (P.i[1]) {
            const float cx = (float)w * 0.5f, cy = (float)h * 0.5f;
            const float dx = (float)x - cx, dy = (float)y - cy;
            const float dist = __builtin_fmaxf(__builtin_sqrtf(dx * dx + dy * dy), 1.0f);
            const float factor = (1.0f - dist / __builtin_fmaxf(cx, cy)) * 0.5f;
            nx = nx + dx / dist * factor;
            ny = ny + dy / dist * factor;
        }
        float sx = (float)x + nx * P.f[1] * P.f[2];
        float sy = (float)y + ny * P.f[1] * P.f[2];
        if (P.i[2]) { sx = rem_euclid(sx, (float)w); sy = rem_euclid(sy, (float)h); }
        return sample_bilinear_round(src, w, h, sx, sy);
    } else if constexpr (FX == PFXK_FX2_BULGE) { // distort.rs:413-436; f: cx cy max_r strength amount
        const float dx = (float)x - P.f[0], dy = (float)y - P.f[1];
        const float dist = __builtin_sqrtf(dx * dx + dy * dy);
        const float norm = __builtin_fminf(dist / P.f[2], 1.0f);
        if (norm >= 1.0f) return s;
        const float falloff = 1.0f - norm;
        const float factor = P.f[4] > 0.0f ? 1.0f - falloff * P.f[3] * 0.5f : (P.f[4] < 0.0f ? 1.0f + falloff * P.f[3] * 0.5f : 1.0f);
        return sample_bilinear_round(src, w, h, P.f[0] + dx * factor, P.f[1] + dy * factor);
    } else if constexpr (FX == PFXK_FX2_TWIST) { // distort.rs:479-492; f: cx cy max_r twist_amount
        const float dx = (float)x - P.f[0], dy = (float)y - P.f[1];
        const float dist = __builtin_sqrtf(dx * dx + dy * dy);
        const float norm = dist / P.f[2];
        const float rotation = P.f[3] * (1.0f - norm);
        const float cos_r = libm_cos(rotation), sin_r = libm_sin(rotation);
        return sample_bilinear_round(src, w, h, P.f[0] + dx * cos_r - dy * sin_r, P.f[1] + dx * sin_r + dy * cos_r);
    } else if constexpr (FX == PFXK_FX2_NOISE) { // noise.rs:86-142; f: inv_scale strength; i: type mono oct; u0 seed
        const float sx = (float)x * P.f[0], sy = (float)y * P.f[0];
        const uint32_t qx = rs_u32(__builtin_floorf(sx)), qy = rs_u32(__builtin_floorf(sy));
        const float strength = P.f[1];
        const uint32_t seed = P.u[0];
        float nr, ng, nb;
        if (P.i[1]) {
            float nv;
            if (P.i[0] == 0) nv = hash_f32(qx, qy, seed) * 2.0f - 1.0f;
            else if (P.i[0] == 1) {
                const float u1 = __builtin_fmaxf(hash_f32(qx, qy, seed), 0.0001f);
                const float u2 = hash_f32(qx, qy, seed + 7u);
                nv = __builtin_sqrtf(-2.0f * libm_log(u1)) * libm_cos(2.0f * 3.14159265358979323846f * u2) * 0.33f;
            } else nv = turbulence_2d(sx, sy, seed, (uint32_t)P.i[2], 0.5f) * 2.0f - 1.0f;
            nr = ng = nb = nv * strength;
        } else if (P.i[0] == 2) {
            nr = (turbulence_2d(sx, sy, seed, (uint32_t)P.i[2], 0.5f) * 2.0f - 1.0f) * strength;
            ng = (turbulence_2d(sx, sy, seed + 1u, (uint32_t)P.i[2], 0.5f) * 2.0f - 1.0f) * strength;
            nb = (turbulence_2d(sx, sy, seed + 2u, (uint32_t)P.i[2], 0.5f) * 2.0f - 1.0f) * strength;
        } else { // uniform and gaussian share the colour branch (noise.rs:114-138)
            nr = (hash_f32(qx, qy, seed) * 2.0f - 1.0f) * strength;
            ng = (hash_f32(qx, qy, seed + 1u) * 2.0f - 1.0f) * strength;
            nb = (hash_f32(qx, qy, seed + 2u) * 2.0f - 1.0f) * strength;
        }
        return pack_round(r + nr, g + ng, b + nb, a);
    } else if constexpr (FX == PFXK_FX2_REDUCE_NOISE) { // noise.rs:211-256; f: 2*sigma_s^2, 2*sigma_r^2+0.001; i0: r
        const int rad = P.i[0];
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, wsum = 0.f;
        // both divisors are per-call constants: the refined reciprocals are shared by all taps (k_common.h:rdiv, bit-identical to
        // '/' for these operands: divisors >= 0.001, integer-valued numerators <= 195075)
        const rdiv k_spatial = rdiv_prepare(P.f[0]), k_range = rdiv_prepare(P.f[1]);
        for (int dy = -rad; dy <= rad; ++dy) {
            const uint32_t* row = src + (size_t)clampi(y + dy, 0, h - 1) * w;
            for (int dx = -rad; dx <= rad; ++dx) {
                const uint32_t p = row[clampi(x + dx, 0, w - 1)];
                const float pr = ubyte0(p), pg = ubyte1(p), pb = ubyte2(p), pa = ubyte3(p);
                const float spatial = rdiv_apply(k_spatial, (float)(dx * dx + dy * dy));
                const float dr = r - pr, dg = g - pg, db = b - pb;
                const float range = rdiv_apply(k_range, dr * dr + dg * dg + db * db);
                const float wt = libm_exp(-spatial - range);
                s0 += pr * wt; s1 += pg * wt; s2 += pb * wt; s3 += pa * wt;
                wsum += wt;
            }
        }
        if (!(wsum > 0.0f)) return s;
        const float inv = 1.0f / wsum;
        return pack_round(s0 * inv, s1 * inv, s2 * inv, s3 * inv);
    } else if constexpr (FX == PFXK_FX2_VIGNETTE) { // stylize.rs:183-190; f: cx cy max_dist soft amount
        const float dx = (float)x - P.f[0], dy = (float)y - P.f[1];
        const float dist = __builtin_sqrtf(dx * dx + dy * dy) / P.f[2];
        const float q = __builtin_fminf(dist / P.f[3], 1.0f);
        const float pw = (float)((double)q * (double)q); // powf(q, 2.0): the f64 product is exact, one rounding
        const float vf = rs_clamp(1.0f - (P.f[4] * pw), 0.0f, 1.0f);
        return pack_round(r * vf, g * vf, b * vf, a);
    } else if constexpr (FX == PFXK_FX2_HALFTONE) { // stylize.rs:254-276; f: ds cos_a sin_a; i0 shape
        const float lum = (0.2126f * r + 0.7152f * g + 0.0722f * b) / 255.0f;
        const float fx = (float)x * P.f[1] + (float)y * P.f[2];
        const float fy = -((float)x) * P.f[2] + (float)y * P.f[1];
        const float qx = fx / P.f[0], qy = fy / P.f[0];
        const float cx = __builtin_fabsf(qx - __builtin_truncf(qx)) - 0.5f, cy = __builtin_fabsf(qy - __builtin_truncf(qy)) - 0.5f;
        float th;
        if (P.i[0] == 0) th = __builtin_sqrtf(cx * cx + cy * cy) * 2.0f;
        else if (P.i[0] == 1) th = __builtin_fmaxf(__builtin_fabsf(cx), __builtin_fabsf(cy)) * 2.0f;
        else if (P.i[0] == 2) th = __builtin_fabsf(cx) + __builtin_fabsf(cy);
        else th = __builtin_fabsf(cy) * 2.0f;
        return (th < lum ? 0x00ffffffu : 0u) | (s & 0xff000000u);
    } else if constexpr (FX == PFXK_FX2_GRID) { // render.rs:66-91; u: cw ch lw color; i0 style; f0 opacity
        const bool draw = P.i[0] == 0 ? (((uint32_t)x % P.u[0]) < P.u[2] || ((uint32_t)y % P.u[1]) < P.u[2])
                                      : ((((uint32_t)x / P.u[0]) + ((uint32_t)y / P.u[1])) % 2u == 0u);
        if (!draw) return s;
        const float t = P.f[0], it = 1.0f - t;
        return pack_round(r * it + ubyte0(P.u[3]) * t, g * it + ubyte1(P.u[3]) * t, b * it + ubyte2(P.u[3]) * t, a * it + ubyte3(P.u[3]) * t);
    } else if constexpr (FX == PFXK_FX2_BORDER) { // render.rs:148-160; u0 border_w, u1 color
        const uint32_t bw = P.u[0], xu = (uint32_t)x, yu = (uint32_t)y;
        return (xu < bw || yu < bw || xu >= (uint32_t)w - bw || yu >= (uint32_t)h - bw) ? P.u[1] : s;
    } else if constexpr (FX == PFXK_FX2_SHADOW) { // render.rs:325-344; aux0: blurred alpha image (RGBA, channel 0 used); f0 opacity; u0 color
        // u1 != 0: aux0 is the one-channel plane (round 6: the blur runs on the plane, a quarter of the arithmetic); else the RGBA (a, a, a, a) image, channel 0 read
        const uint32_t bl = P.u[1] ? (uint32_t)((const uint8_t*)P.aux0)[(size_t)y * w + x] : ((const uint32_t*)P.aux0)[(size_t)y * w + x];
        const float shadow_a = div255(ubyte0(bl)) * P.f[0] * div255(ubyte3(P.u[0]));
        const float src_a = div255(a);
        const float out_a = src_a + shadow_a * (1.0f - src_a);
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float shadow_c = div255((float)((P.u[0] >> (8 * c)) & 0xffu)), src_c = div255((float)((s >> (8 * c)) & 0xffu));
            const float oc = out_a > 0.0f ? (src_c * src_a + shadow_c * shadow_a * (1.0f - src_a)) / out_a : 0.0f;
            o[c] = oc * 255.0f;
        }
        return pack_round(o[0], o[1], o[2], out_a * 255.0f);
    } else if constexpr (FX == PFXK_FX2_OUTLINE) { // render.rs:449-567; f0 radius; i: sr mode aa; u0 color
        const int sr = P.i[0];
        // min d^2 over the window to a filled (alpha > 0) and to an empty texel; the reference's pruned scan returns the same minimum
        int best_f = 0x7fffffff, best_e = 0x7fffffff;
        if (P.aux0 != nullptr) {
            // the "filled" bit of every pixel comes from a bit plane (alpha_bits_kernel: one bit per pixel, rows padded with 16 zero bits on either
            // side): a window row is one funnel shift, and its nearest filled / empty texel is a count of trailing / leading zeros on either side of
            // the centre — per window ROW what the loop below does per window ELEMENT (search radii up to 15: 31-bit rows)
            const uint32_t* bits = static_cast<const uint32_t*>(P.aux0);
            const int stride = P.i[3], p0 = x + 16 - sr;
            const uint32_t wmask = (2u << (2 * sr)) - 1u;                       // 2 sr + 1 bits
            const int lo = max(0, sr - x), hi = min(2 * sr, sr + (w - 1 - x)); // window columns inside the image (render.rs:472-475)
            const uint32_t valid = ((2u << hi) - 1u) & ~((1u << lo) - 1u);
            auto nearest = [&](uint32_t W, int dy, int& best) {
                const uint32_t right = W >> sr, left = W << (31 - sr);          // bit 0 / bit 31 = the centre column
                const int dr = right ? __builtin_ctz(right) : 64, dl = left ? __builtin_clz(left) : 64;
                const int dx = min(dr, dl);
                if (dx <= sr) best = min(best, dx * dx + dy * dy);
            };
            for (int dy = -sr; dy <= sr; ++dy) {
                const int sy = y + dy;
                if (sy < 0 || sy >= h) continue;
                const uint32_t* row = bits + (size_t)sy * stride + (p0 >> 5);
                const uint32_t Wf = __builtin_amdgcn_alignbit(row[1], row[0], (uint32_t)(p0 & 31)) & wmask;
                nearest(Wf, dy, best_f);
                nearest(~Wf & valid, dy, best_e);
            }
        } else
        for (int dy = -sr; dy <= sr; ++dy) {
            const int sy = y + dy;
            if (sy < 0 || sy >= h) continue;
            for (int dx = -sr; dx <= sr; ++dx) {
                const int sx = x + dx;
                if (sx < 0 || sx >= w) continue;
                const int d = dx * dx + dy * dy;
                if ((src[(size_t)sy * w + sx] >> 24) != 0u) best_f = min(best_f, d);
                else best_e = min(best_e, d);
            }
        }
        const float radius = P.f[0];
        auto shell = [&](float distance) {
            if (P.i[2]) {
                const float t = rs_clamp((radius + 0.5f - distance) / 1.0f, 0.0f, 1.0f);
                return t * t * (3.0f - 2.0f * t);
            }
            return distance <= radius ? 1.0f : 0.0f;
        };
        const float src_a = div255(a);
        float outside_cov = best_f != 0x7fffffff ? shell(__builtin_fmaxf(__builtin_sqrtf((float)best_f) - 1.0f, 0.0f)) : 0.0f;
        outside_cov = outside_cov * (1.0f - src_a);
        float inside_cov = best_e != 0x7fffffff ? shell(__builtin_sqrtf((float)best_e)) : 0.0f;
        inside_cov = inside_cov * src_a;
        const float under_cov = P.i[1] == 1 ? 0.0f : outside_cov, over_cov = P.i[1] == 0 ? 0.0f : inside_cov;
        const float ca = div255(ubyte3(P.u[0]));
        const float a_under = ca * under_cov, a_over = ca * over_cov;
        float comp[3] = {div255(r), div255(g), div255(b)};
        float comp_a = src_a;
        const float col[3] = {div255(ubyte0(P.u[0])), div255(ubyte1(P.u[0])), div255(ubyte2(P.u[0]))};
        if (a_under > 0.0f) {
            const float out_a = comp_a + a_under * (1.0f - comp_a);
            if (out_a > 0.0f) {
#pragma unroll
                for (int c = 0; c < 3; ++c) comp[c] = (comp[c] * comp_a + col[c] * a_under * (1.0f - comp_a)) / out_a;
            }
            comp_a = out_a;
        }
        if (a_over > 0.0f) {
            const float out_a = a_over + comp_a * (1.0f - a_over);
            if (out_a > 0.0f) {
#pragma unroll
                for (int c = 0; c < 3; ++c) comp[c] = (col[c] * a_over + comp[c] * comp_a * (1.0f - a_over)) / out_a;
            }
            comp_a = out_a;
        }
        // `(v.clamp(0,1) * 255).round() as u8`
        return pack_round(rs_clamp(comp[0], 0.f, 1.f) * 255.0f, rs_clamp(comp[1], 0.f, 1.f) * 255.0f, rs_clamp(comp[2], 0.f, 1.f) * 255.0f,
                          rs_clamp(comp_a, 0.f, 1.f) * 255.0f);
    } else if constexpr (FX == PFXK_FX2_PIXEL_DRAG) { // glitch.rs:73-95; f: dx_dir dy_dir dist amount/100; u0 seed
        if (hash_f32((uint32_t)y, 0u, P.u[0]) > P.f[3]) return s;
        const int drag = rs_i32(hash_f32((uint32_t)y, 1u, P.u[0]) * P.f[2]);
        const int sx = clampi(rs_i32(__builtin_roundf((float)x - (float)drag * P.f[0])), 0, w - 1);
        const int sy = clampi(rs_i32(__builtin_roundf((float)y - (float)drag * P.f[1])), 0, h - 1);
        return src[(size_t)sy * w + sx];
    } else if constexpr (FX == PFXK_FX2_RGB_DISPLACE) { // glitch.rs:178-191; i: rx ry gx gy bx by
        const uint32_t pr = src[(size_t)clampi(y + P.i[1], 0, h - 1) * w + clampi(x + P.i[0], 0, w - 1)];
        const uint32_t pg = src[(size_t)clampi(y + P.i[3], 0, h - 1) * w + clampi(x + P.i[2], 0, w - 1)];
        const uint32_t pb = src[(size_t)clampi(y + P.i[5], 0, h - 1) * w + clampi(x + P.i[4], 0, w - 1)];
        return (pr & 0xffu) | (pg & 0xff00u) | (pb & 0xff0000u) | (s & 0xff000000u);
    } else if constexpr (FX == PFXK_FX2_INK) { // artistic.rs:68-94; f: edge_strength threshold
        float l[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const uint32_t p = texel(src, w, h, x + i - 1, y + j - 1);
                l[j][i] = 0.2126f * ubyte0(p) + 0.7152f * ubyte1(p) + 0.0722f * ubyte2(p);
            }
        const float gx = -l[0][0] - 2.0f * l[1][0] - l[2][0] + l[0][2] + 2.0f * l[1][2] + l[2][2];
        const float gy = -l[0][0] - 2.0f * l[0][1] - l[0][2] + l[2][0] + 2.0f * l[2][1] + l[2][2];
        const float edge = __builtin_sqrtf(gx * gx + gy * gy) * P.f[0] / 100.0f;
        return (edge > P.f[1] ? 0u : 0x00ffffffu) | (s & 0xff000000u);
    } else if constexpr (FX == PFXK_FX2_COLOR_FILTER) { // artistic.rs:279-308; f: fc[3] intensity; i0 mode
        const float in[3] = {r, g, b};
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sv = div255(in[c]), f = P.f[c];
            float bl;
            if (P.i[0] == 0) bl = sv * f;
            else if (P.i[0] == 1) bl = 1.0f - (1.0f - sv) * (1.0f - f);
            else if (P.i[0] == 2) bl = sv < 0.5f ? 2.0f * sv * f : 1.0f - 2.0f * (1.0f - sv) * (1.0f - f);
            else bl = f < 0.5f ? sv - (1.0f - 2.0f * f) * sv * (1.0f - sv) : sv + (2.0f * f - 1.0f) * (__builtin_sqrtf(sv) - sv);
            o[c] = (sv * (1.0f - P.f[3]) + bl * P.f[3]) * 255.0f;
        }
        return pack_round(o[0], o[1], o[2], a);
    } else if constexpr (FX == PFXK_FX2_CONTOURS) { // contours.rs:84-111; f: inv_scale freq edge la blend lc[3]; u0 seed; i0 oct
        const float nv = turbulence_2d((float)x * P.f[0], (float)y * P.f[0], P.u[0], (uint32_t)P.i[0], 0.5f);
        const float level = nv * P.f[1];
        const float dtc = __builtin_fabsf(level - __builtin_roundf(level)) / P.f[1];
        const float edge = P.f[2];
        const float line_alpha = dtc < edge ? 1.0f : (dtc < edge * 2.0f ? 1.0f - (dtc - edge) / edge : 0.0f);
        const float alpha = line_alpha * P.f[3] * P.f[4];
        const float ia = 1.0f - alpha;
        return pack_round(r * ia + P.f[5] * alpha, g * ia + P.f[6] * alpha, b * ia + P.f[7] * alpha, a);
    }
    return s;
}

// one bit per pixel: alpha != 0.  Row layout: bit (x + 16) of the row's bit string (16 zero bits of padding on either side, one spare dword)
__global__ __launch_bounds__(256) void alpha_bits_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ bits, int w, int h, int stride)
{
    const int lane = threadIdx.x & 63, wv = blockIdx.x * 4 + (threadIdx.x >> 6), y = blockIdx.y;
    if (2 * wv >= stride) return;
    const int x = wv * 64 + lane - 16;
    const bool filled = x >= 0 && x < w && (src[(size_t)y * w + x] >> 24) != 0u;
    const unsigned long long m = __ballot(filled);
    if (lane < 2 && 2 * wv + lane < stride) bits[(size_t)y * stride + 2 * wv + lane] = (uint32_t)(m >> (32 * lane));
}

template <int FX>
__global__ __launch_bounds__(256) void fx_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                 const uint8_t* __restrict__ mask, const pfxk_fx_params P, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    const uint32_t s = src[oi];
    if (mask && mask[oi] == 0) { dst[oi] = s; return; } // every effect leaves unselected pixels as they were
    dst[oi] = fx_pixel<FX>(src, s, x, y, w, h, P);
}

// ---- crystallize (distort.rs:26-169): seeds built on the host; accumulate -> average -> assign ------------------
PFX_DEV int nearest_seed(const float2* __restrict__ seeds, int cells_x, int cells_y, float cs, int x, int y)
{
    const int gcx = rs_i32((float)x / cs), gcy = rs_i32((float)y / cs);
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    float best = 3.40282347e+38f;
    int best_idx = 0;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int nx = gcx + dx, ny = gcy + dy;
            if (nx < 0 || ny < 0 || nx >= cells_x || ny >= cells_y) continue;
            const int idx = ny * cells_x + nx;
            const float2 sp = seeds[idx];
            const float d = (px - sp.x) * (px - sp.x) + (py - sp.y) * (py - sp.y);
            if (d < best) { best = d; best_idx = idx; }
        }
    return best_idx;
}

// acc: per cell 5 x u64 {sum r, g, b, a, count}.  Integer sums == the reference's f64 sums of integers (exact below 2^53).
// A wave holds 64 consecutive pixels of a row, which fall into a handful of cells: the runs of lanes with the same cell are summed with one
// wave scan (r | b << 16 and g | a << 16, the sums stay below 64 * 255) and the first lane of each run adds it to the block's LDS table.
// (Round 2 had every lane add its own pixel to the table: 64 lanes on the same five words serialised, 78 % of the kernel's LDS cycles were
// bank conflicts.)
__global__ __launch_bounds__(256) void crystal_accum_kernel(const uint32_t* __restrict__ src, const float2* __restrict__ seeds,
                                                            unsigned long long* __restrict__ acc, int cells_x, int cells_y, float cs,
                                                            int w, int h, int tile_rows)
{
    // the block's four waves (a 64 x 4 tile) meet in an LDS hash table — one entry per (wave, cell), a few dozen inserts per block — and
    // the table is flushed with one set of global u64 atomics per (block, cell)
    constexpr int SLOTS = 512; // >= 2 x the 256 keys a block can insert: open addressing always finds a slot
    __shared__ int keys[SLOTS];
    __shared__ uint32_t vals[SLOTS][5];
    for (int i = threadIdx.x; i < SLOTS; i += 256) { keys[i] = -1; vals[i][0] = vals[i][1] = vals[i][2] = vals[i][3] = vals[i][4] = 0u; }
    __syncthreads();
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    for (int it = 0; it < tile_rows / 4; ++it) { // 64 x tile_rows pixels per block: the global atomics per pixel fall with the tile's area / perimeter
    const int y = blockIdx.y * tile_rows + it * 4 + (threadIdx.x >> 6);
    const bool live = x < w && y < h;
    uint32_t p = 0;
    int cell = -1;
    if (live) {
        p = src[(size_t)y * w + x];
        cell = nearest_seed(seeds, cells_x, cells_y, cs, x, y);
    }
    // A cell meets a row in one or a few runs of consecutive lanes.  One inclusive scan of the wave's packed channels serves every run of the
    // row: the first lane of each run takes the difference of the scan at the run's two ends (two ds_bpermute per value) and adds the run to
    // the block's table — all runs of the wave at once, no loop over cells.
    const int lane = (int)(threadIdx.x & 63);
    uint32_t pre_rb = live ? (p & 0x00ff00ffu) : 0u, pre_ga = live ? ((p >> 8) & 0x00ff00ffu) : 0u; // 64 * 255 < 2^16: the fields do not carry
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t a = (uint32_t)__shfl_up((int)pre_rb, off, 64), b = (uint32_t)__shfl_up((int)pre_ga, off, 64);
        if (lane >= off) { pre_rb += a; pre_ga += b; }
    }
    const int prev_cell = __shfl_up(cell, 1, 64);
    const bool first = lane == 0 || prev_cell != cell;
    const unsigned long long firsts = __ballot(first);
    const unsigned long long above = lane == 63 ? 0ull : (firsts >> (lane + 1));
    const int last = above ? lane + __builtin_ctzll(above) : 63;            // last lane of the run this lane would lead
    const uint32_t end_rb = (uint32_t)__shfl((int)pre_rb, last, 64), end_ga = (uint32_t)__shfl((int)pre_ga, last, 64);
    const uint32_t beg_rb = (uint32_t)__shfl_up((int)pre_rb, 1, 64), beg_ga = (uint32_t)__shfl_up((int)pre_ga, 1, 64);
    if (first && cell >= 0) {
        const uint32_t rb = end_rb - (lane ? beg_rb : 0u), ga = end_ga - (lane ? beg_ga : 0u);
        int slot = (int)(((uint32_t)cell * 2654435761u) >> 23); // 9 bits
        for (;;) {
            const int prev = atomicCAS(&keys[slot], -1, cell);
            if (prev == -1 || prev == cell) break;
            slot = (slot + 1) & (SLOTS - 1);
        }
        atomicAdd(&vals[slot][0], rb & 0xffffu); atomicAdd(&vals[slot][1], ga & 0xffffu);
        atomicAdd(&vals[slot][2], rb >> 16);     atomicAdd(&vals[slot][3], ga >> 16);
        atomicAdd(&vals[slot][4], (uint32_t)(last - lane + 1));
    }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SLOTS; i += 256) {
        const int c = keys[i];
        if (c < 0) continue;
        unsigned long long* a = acc + (size_t)c * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) atomicAdd(a + k, (unsigned long long)vals[i][k]);
    }
}

__global__ __launch_bounds__(256) void crystal_avg_kernel(const unsigned long long* __restrict__ acc, uint32_t* __restrict__ avg, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long cnt = acc[(size_t)i * 5 + 4];
    uint32_t out = 0;
    if (cnt > 0) {
        const double inv = 1.0 / (double)(uint32_t)cnt; // counts[i] is u32 in the reference
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double v = round((double)acc[(size_t)i * 5 + c] * inv);
            v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
            out |= (uint32_t)v << (8 * c);
        }
    }
    avg[i] = out;
}

__global__ __launch_bounds__(256) void crystal_assign_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                             const uint8_t* __restrict__ mask, const float2* __restrict__ seeds,
                                                             const uint32_t* __restrict__ avg, int cells_x, int cells_y, float cs, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    if (mask && mask[oi] == 0) { dst[oi] = src[oi]; return; }
    dst[oi] = avg[nearest_seed(seeds, cells_x, cells_y, cs, x, y)];
}

// ---- oil painting (artistic.rs:123-215): per-lane intensity histogram in LDS ------------------------------------
// bins[levels][lanes] u64, one word per bin: count << 51 | sum_b << 34 | sum_g << 17 | sum_r (count <= 441, each sum <= 112455 < 2^17),
// so a window element is ONE fire-and-forget ds_add_u64 (no read-modify-write round trip in the dependency chain; the
// [level][lane] layout keeps a wave's 64 adds on distinct banks whatever the levels are).
// A lane owns a column and walks OIL_ROWS rows down: the histogram persists, a step adds the window's new bottom row and removes the row that
// left it (2 (2r+1) updates instead of (2r+1)^2 — and a source pixel's level is computed twice, not (2r+1)^2 times).  Subtracting the packed
// word that was added restores every field (the 64-bit adds form a group; no field ever goes below what the window holds).
constexpr int OIL_ROWS = 32;
__global__ void oil_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint8_t* __restrict__ mask, int radius,
                           int levels, int w, int h)
{
    extern __shared__ unsigned long long obins[];
    const int lanes = blockDim.x, t = threadIdx.x;
    const int x = blockIdx.x * lanes + t, y0 = blockIdx.y * OIL_ROWS;
    if (x >= w) return;
    unsigned long long* b0 = obins + t;
    for (int k = 0; k < levels; ++k) b0[k * lanes] = 0ull;
    auto row_update = [&](int yy, bool add) {
        const uint32_t* row = src + (size_t)clampi(yy, 0, h - 1) * w;
        for (int dx = -radius; dx <= radius; ++dx) {
            const uint32_t p = row[clampi(x + dx, 0, w - 1)];
            const uint32_t pr = p & 0xffu, pg = (p >> 8) & 0xffu, pb = (p >> 16) & 0xffu;
            uint32_t k = (pr + pg + pb) / 3u * (uint32_t)levels / 256u;
            k = min(k, (uint32_t)levels - 1u);
            const unsigned long long word = (1ull << 51) | ((unsigned long long)pb << 34) | ((unsigned long long)pg << 17) | (unsigned long long)pr;
            atomicAdd(b0 + k * lanes, add ? word : 0ull - word);
        }
    };
    for (int dy = -radius; dy < radius; ++dy) row_update(y0 + dy, true);
    const int y1 = min(y0 + OIL_ROWS, h);
    for (int y = y0; y < y1; ++y) {
        row_update(y + radius, true);
        const size_t oi = (size_t)y * w + x;
        const uint32_t s = src[oi];
        uint32_t out = s;
        if (!(mask && mask[oi] == 0)) {
            uint32_t max_count = 0;
            unsigned long long best = 0ull;
            for (int k = 0; k < levels; ++k) {
                const unsigned long long v = b0[k * lanes];
                const uint32_t c = (uint32_t)(v >> 51);
                if (c > max_count) { max_count = c; best = v; } // first level with the largest count (artistic.rs:186-195)
            }
            out = s & 0xff000000u;
            if (max_count > 0) {
                const uint32_t sr = (uint32_t)best & 0x1ffffu, sg = (uint32_t)(best >> 17) & 0x1ffffu, sb = (uint32_t)(best >> 34) & 0x1ffffu;
                out |= (sr / max_count) | ((sg / max_count) << 8) | ((sb / max_count) << 16);
            }
        }
        dst[oi] = out;
        row_update(y - radius, false);
    }
}

// ---- drop shadow helpers (render.rs:233-301): offset alpha plane, separable max "widen", expand to RGBA ---------
__global__ __launch_bounds__(256) void shadow_alpha_kernel(const uint32_t* __restrict__ src, uint8_t* __restrict__ plane, int ox, int oy,
                                                           int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int sx = x - ox, sy = y - oy;
    plane[(size_t)y * w + x] = (sx >= 0 && sx < w && sy >= 0 && sy < h) ? (uint8_t)(src[(size_t)sy * w + sx] >> 24) : (uint8_t)0;
}
template <bool VERT>
__global__ __launch_bounds__(256) void plane_max_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int r, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t m = 0;
    if constexpr (VERT) {
        for (int sy = max(y - r, 0); sy <= min(y + r, h - 1); ++sy) m = max(m, (uint32_t)in[(size_t)sy * w + x]);
    } else {
        for (int sx = max(x - r, 0); sx <= min(x + r, w - 1); ++sx) m = max(m, (uint32_t)in[(size_t)y * w + sx]);
    }
    out[(size_t)y * w + x] = (uint8_t)m;
}
__global__ __launch_bounds__(256) void plane_expand_kernel(const uint8_t* __restrict__ plane, uint32_t* __restrict__ rgba, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) rgba[i] = (uint32_t)plane[i] * 0x01010101u;
}

dim3 tile_grid(uint32_t w, uint32_t h) { return dim3((w + 63) / 64, (h + 3) / 4); }

} // namespace

#define FX_CASE(ID) \
    case ID: fx_kernel<ID><<<tile_grid(w, h), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, *P, (int)w, (int)h); break;

extern "C" hipError_t pfxk_fx(hipStream_t s, int fx, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const pfxk_fx_params* P,
                              uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    switch (fx) {
        FX_CASE(PFXK_FX2_ZOOM) FX_CASE(PFXK_FX2_DENTS) FX_CASE(PFXK_FX2_BULGE) FX_CASE(PFXK_FX2_TWIST) FX_CASE(PFXK_FX2_NOISE)
        FX_CASE(PFXK_FX2_REDUCE_NOISE) FX_CASE(PFXK_FX2_VIGNETTE) FX_CASE(PFXK_FX2_HALFTONE) FX_CASE(PFXK_FX2_GRID) FX_CASE(PFXK_FX2_BORDER)
        FX_CASE(PFXK_FX2_SHADOW) FX_CASE(PFXK_FX2_OUTLINE) FX_CASE(PFXK_FX2_PIXEL_DRAG) FX_CASE(PFXK_FX2_RGB_DISPLACE) FX_CASE(PFXK_FX2_INK)
        FX_CASE(PFXK_FX2_COLOR_FILTER) FX_CASE(PFXK_FX2_CONTOURS)
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

extern "C" uint32_t pfxk_alpha_bits_stride(uint32_t w) { return 2u * ((w + 32u + 63u) / 64u) + 1u; } // dwords per row
extern "C" hipError_t pfxk_alpha_bits(hipStream_t s, const uint8_t* d_src, uint32_t* d_bits, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    const int stride = (int)pfxk_alpha_bits_stride(w);
    alpha_bits_kernel<<<dim3((uint32_t)((stride / 2 + 1 + 3) / 4), h), 256, 0, s>>>((const uint32_t*)d_src, d_bits, (int)w, (int)h, stride);
    return hipGetLastError();
}

// d_acc: cells*5 u64 (zeroed here), d_avg: cells u32
extern "C" hipError_t pfxk_crystallize(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const float* d_seeds_xy,
                                       unsigned long long* d_acc, uint32_t* d_avg, int cells_x, int cells_y, float cs, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    const int n = cells_x * cells_y;
    hipError_t e = hipMemsetAsync(d_acc, 0, (size_t)n * 5 * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    const int tile_rows = cs >= 8.0f ? 64 : 4; // distinct cells per block stay below half the table: (64 / 8 + 2)^2 = 100, or one per pixel of 64 x 4
    crystal_accum_kernel<<<dim3((w + 63) / 64, (h + tile_rows - 1) / tile_rows), 256, 0, s>>>((const uint32_t*)d_src, (const float2*)d_seeds_xy, d_acc, cells_x, cells_y, cs, (int)w, (int)h, tile_rows);
    crystal_avg_kernel<<<(n + 255) / 256, 256, 0, s>>>(d_acc, d_avg, n);
    crystal_assign_kernel<<<tile_grid(w, h), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (const float2*)d_seeds_xy, d_avg, cells_x,
                                                          cells_y, cs, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_oil_painting(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int radius, int levels,
                                        uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    const int lanes = levels <= 32 ? 256 : (levels <= 64 ? 128 : 64); // 8 B per level per lane, <= 64 KiB of LDS per block
    const size_t lds = (size_t)lanes * levels * 8;
    oil_kernel<<<dim3((w + lanes - 1) / lanes, (h + OIL_ROWS - 1) / OIL_ROWS), lanes, lds, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, radius, levels, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_shadow_alpha(hipStream_t s, const uint8_t* d_src, uint8_t* d_plane_a, uint8_t* d_plane_b, uint8_t* d_rgba, int ox, int oy,
                                        int spread, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    shadow_alpha_kernel<<<tile_grid(w, h), 256, 0, s>>>((const uint32_t*)d_src, d_plane_a, ox, oy, (int)w, (int)h);
    if (spread > 0) {
        plane_max_kernel<false><<<tile_grid(w, h), 256, 0, s>>>(d_plane_a, d_plane_b, spread, (int)w, (int)h);
        plane_max_kernel<true><<<tile_grid(w, h), 256, 0, s>>>(d_plane_b, d_plane_a, spread, (int)w, (int)h);
    }
    if (d_rgba) {   // NULL: the caller blurs and composites the PLANE (d_plane_a holds it)
        const size_t n = (size_t)w * h;
        size_t blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        plane_expand_kernel<<<(uint32_t)blocks, 256, 0, s>>>(d_plane_a, (uint32_t*)d_rgba, n);
    }
    return hipGetLastError();
}
