// k_resize.hip — resize_image: `image::imageops::resize` as the reference calls it (src/ops/transform.rs:347-359,
// src/ops/scripting.rs:749-770; algorithm of the `image` crate 0.25.9, see pfx_resize.cpp for the restated weight tables).
//
// Two separable passes like the crate: vertical into an f32 RGBA image, then horizontal with clamp + round-half-away.
// Each output sample accumulates `t += v * w` in source order with one rounding per operation (no FMA): bit-exact class.
// Algorithmic bytes: 4 B/px of source read + 4 B/px of result written; the f32 intermediate (16 B per w x nh sample, written and
// read once) is the non-algorithmic traffic the crate's structure implies.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

// one lane per (x, oy); oy uniform per block -> window bounds and weights come through scalar loads
__global__ __launch_bounds__(256) void resize_v_kernel(const uint32_t* __restrict__ src, float4* __restrict__ tmp, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ count, const uint32_t* __restrict__ off, const float* __restrict__ wts,
                                                       int w)
{
    const int x = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (x >= w) return;
    const uint32_t l = left[oy], n = count[oy];
    const float* wp = wts + off[oy];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    const uint32_t* p = src + (size_t)l * w + x;
    for (uint32_t i = 0; i < n; ++i, p += w) {
        const uint32_t v = *p;
        const float wt = wp[i];
        t0 += ubyte0(v) * wt; t1 += ubyte1(v) * wt; t2 += ubyte2(v) * wt; t3 += ubyte3(v) * wt;
    }
    tmp[(size_t)oy * w + x] = make_float4(t0, t1, t2, t3);
}

__global__ __launch_bounds__(256) void resize_h_kernel(const float4* __restrict__ tmp, uint32_t* __restrict__ dst, const uint32_t* __restrict__ left,
                                                       const uint32_t* __restrict__ count, const uint32_t* __restrict__ off, const float* __restrict__ wts,
                                                       int w, int nw, int nh)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= nw || oy >= nh) return;
    const uint32_t l = left[ox], n = count[ox];
    const float* wp = wts + off[ox];
    const float4* p = tmp + (size_t)oy * w + l;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    for (uint32_t i = 0; i < n; ++i) {
        const float4 v = p[i];
        const float wt = wp[i];
        t0 += v.x * wt; t1 += v.y * wt; t2 += v.z * wt; t3 += v.w * wt;
    }
    // NumCast::from(FloatNearest(clamp(t, 0, 255))): round half away from zero
    dst[(size_t)oy * nw + ox] = pack_round_rgba(t0, t1, t2, t3);
}

// Both passes in one kernel: a block owns RZ_TOX x RZ_TOY output pixels, runs the vertical pass for the source columns its
// horizontal taps reach (h_left[first] .. h_left[last] + h_count[last], both monotone in the output column) into an f32 RGBA tile in
// LDS, then the horizontal pass out of it.  The same operations in the same order as the two kernels above — bit-identical — without
// the 16 B/sample intermediate in HBM (3/4 of the two-pass traffic at 8K -> 4K).
#ifndef PFX_RZ_ROWS_PER_WAVE
#define PFX_RZ_ROWS_PER_WAVE 1
#endif
#ifndef PFX_RZ_TOX
#define PFX_RZ_TOX 64
#endif
constexpr int RZ_TOX = PFX_RZ_TOX, RZ_TOY = 8;   // (128 output columns measured slower: 0.143 against 0.104 ms at 8K -> 4K bilinear)
// Taps are taken four at a time: the four loads of a group are issued from clamped (always valid) indices before any of them is used, and the
// accumulation — `t += v * w` in source order, one rounding per operation — skips the ones past the window under a block-uniform test.  With one load per
// trip of a run-time-count loop every tap was a full memory (or LDS) round trip of its own (8K -> 4K bilinear 0.131 ms, 0.158 of the HBM roofline).
__global__ __launch_bounds__(256) void resize_fused_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint32_t* __restrict__ v_left,
                                                           const uint32_t* __restrict__ v_count, const uint32_t* __restrict__ v_off, const float* __restrict__ v_wts,
                                                           const uint32_t* __restrict__ h_left, const uint32_t* __restrict__ h_count, const uint32_t* __restrict__ h_off,
                                                           const float* __restrict__ h_wts, int w, int nw, int nh, int span_max)
{
    extern __shared__ float4 rz_tile[]; // [RZ_TOY][span_max]
    const int ox0 = blockIdx.x * RZ_TOX, oy0 = blockIdx.y * RZ_TOY, ox_last = min(ox0 + RZ_TOX, nw) - 1;
    const int L = (int)h_left[ox0], span = (int)(h_left[ox_last] + h_count[ox_last]) - L;
#if PFX_RZ_ROWS_PER_WAVE
    // a wave per output row (two rows each): oy stays wave-uniform — window bounds and weights come through scalar loads — and the four waves work on four
    // rows at once instead of half the block idling over one row's ~130 source columns
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), x_first = (int)(threadIdx.x & 63u), x_step = 64;
    for (int ry = wv; ry < RZ_TOY; ry += 4) {
#else
    const int x_first = (int)threadIdx.x, x_step = 256;
    for (int ry = 0; ry < RZ_TOY; ++ry) { // oy uniform: window bounds and weights come through scalar loads
#endif
        const int oy = oy0 + ry;
        if (oy >= nh) break;
        const uint32_t l = v_left[oy], n = v_count[oy];
        const float* wp = v_wts + v_off[oy];
        for (int xs = x_first; xs < span; xs += x_step) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            const uint32_t* p = src + (size_t)l * w + L + xs;
            for (uint32_t i = 0; i < n; i += 4) {
                uint32_t v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) v[k] = p[(size_t)min(i + k, n - 1u) * w];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (i + k < n) {   // uniform
                        const float wt = wp[i + k];
                        t0 += ubyte0(v[k]) * wt; t1 += ubyte1(v[k]) * wt; t2 += ubyte2(v[k]) * wt; t3 += ubyte3(v[k]) * wt;
                    }
            }
            rz_tile[ry * span_max + xs] = make_float4(t0, t1, t2, t3);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RZ_TOX * RZ_TOY / 256; ++k) {
        const int idx = threadIdx.x + 256 * k, ox = ox0 + (idx & (RZ_TOX - 1)), ry = idx / RZ_TOX, oy = oy0 + ry;
        if (ox >= nw || oy >= nh) continue;
        const uint32_t n = h_count[ox];
        const float* wp = h_wts + h_off[ox];
        const float4* p = rz_tile + ry * span_max + ((int)h_left[ox] - L);
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        for (uint32_t i = 0; i < n; i += 4) {   // n varies by column (per lane): the guard below is a select per tap, the LDS reads are unconditional
            float4 v[4]; float wt[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) { const uint32_t j = min(i + q, n - 1u); v[q] = p[j]; wt[q] = wp[j]; }
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q)
                if (i + q < n) { t0 += v[q].x * wt[q]; t1 += v[q].y * wt[q]; t2 += v[q].z * wt[q]; t3 += v[q].w * wt[q]; }
        }
        dst[(size_t)oy * nw + ox] = pack_round_rgba(t0, t1, t2, t3); // NumCast::from(FloatNearest(clamp(t, 0, 255))): round half away from zero
    }
}

// ---- layer affine / perspective resampler (apply_affine, src/ops/transform.rs:826-946) --------------------------------------
// Inverse homography hi[9] from the host; per pixel: u, v in canvas-centred coordinates, projective divide, bilinear against a
// transparent outside (lerp form a + (b - a) * t, `.round().clamp(0,255) as u8`) or nearest.  Pixels that map outside stay 0.
PFX_DEV int rs_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
__global__ __launch_bounds__(256) void affine_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const pfxk_affine_params P, int src_w,
                                                     int src_h, int cw, int ch)
{
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= cw || dy >= ch) return;
    uint32_t out = 0u;
    const float v = ((float)dy - P.cy - P.off_y) * P.inv_scale;
    const float base_sx = P.hi[1] * v + P.hi[2], base_sy = P.hi[4] * v + P.hi[5], base_sw = P.hi[7] * v + P.hi[8];
    const float u = ((float)dx - P.cx - P.off_x) * P.inv_scale;
    const float w = P.hi[6] * u + base_sw;
    if (!(__builtin_fabsf(w) < 1e-8f)) {
        const float inv_w = 1.0f / w;
        const float sx = (P.hi[0] * u + base_sx) * inv_w + P.cx;
        const float sy = (P.hi[3] * u + base_sy) * inv_w + P.cy;
        if (P.nearest) {
            const int nx = rs_i32(__builtin_roundf(sx)), ny = rs_i32(__builtin_roundf(sy));
            if (nx >= 0 && ny >= 0 && nx < src_w && ny < src_h) out = src[(size_t)ny * src_w + nx];
        } else {
            const int x0 = rs_i32(__builtin_floorf(sx)), y0 = rs_i32(__builtin_floorf(sy));
            if (!(x0 < -1 || y0 < -1 || x0 >= src_w || y0 >= src_h)) {
                const float fx = sx - (float)x0, fy = sy - (float)y0;
                auto sample = [&](int px, int py) -> uint32_t { return (px < 0 || py < 0 || px >= src_w || py >= src_h) ? 0u : src[(size_t)py * src_w + px]; };
                const uint32_t tl = sample(x0, y0), tr = sample(x0 + 1, y0), bl = sample(x0, y0 + 1), br = sample(x0 + 1, y0 + 1);
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = (float)((tl >> (8 * c)) & 0xffu), b = (float)((tr >> (8 * c)) & 0xffu);
                    const float cc = (float)((bl >> (8 * c)) & 0xffu), d = (float)((br >> (8 * c)) & 0xffu);
                    const float top = a + (b - a) * fx, bot = cc + (d - cc) * fx;
                    o[c] = top + (bot - top) * fy;
                }
                out = pack_round_rgba(o[0], o[1], o[2], o[3]);
            }
        }
    }
    dst[(size_t)dy * cw + dx] = out;
}

} // namespace

extern "C" hipError_t pfxk_affine(hipStream_t s, const uint8_t* d_src, uint32_t sw, uint32_t sh, uint8_t* d_dst, uint32_t cw, uint32_t ch,
                                  const pfxk_affine_params* P)
{
    if (cw == 0 || ch == 0) return hipSuccess;
    affine_kernel<<<dim3((cw + 63) / 64, (ch + 3) / 4), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, *P, (int)sw, (int)sh, (int)cw, (int)ch);
    return hipGetLastError();
}

// tables: v_* index by output row (nh entries), h_* by output column (nw entries).  span_max = the widest source-column range a
// 64-column output tile reaches (pfx_resize.cpp); 0 or a tile beyond 64 KiB of LDS takes the two-pass path through d_tmp.
extern "C" int pfxk_resize_tile_cols(void) { return RZ_TOX; }
extern "C" hipError_t pfxk_resize(hipStream_t s, const uint8_t* d_src, float* d_tmp, uint8_t* d_dst, const uint32_t* v_left, const uint32_t* v_count,
                                  const uint32_t* v_off, const float* v_wts, const uint32_t* h_left, const uint32_t* h_count, const uint32_t* h_off,
                                  const float* h_wts, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t span_max)
{
    (void)h;
    if (w == 0 || nw == 0 || nh == 0) return hipSuccess;
    const size_t lds = (size_t)span_max * RZ_TOY * sizeof(float4);
    if (span_max != 0 && lds <= 64u * 1024u) {
        hipError_t e = grant_lds_for((const void*)resize_fused_kernel, lds);
        if (e) return e;
        resize_fused_kernel<<<dim3((nw + RZ_TOX - 1) / RZ_TOX, (nh + RZ_TOY - 1) / RZ_TOY), 256, lds, s>>>(
            (const uint32_t*)d_src, (uint32_t*)d_dst, v_left, v_count, v_off, v_wts, h_left, h_count, h_off, h_wts, (int)w, (int)nw, (int)nh, (int)span_max);
        return hipGetLastError();
    }
    resize_v_kernel<<<dim3((w + 255) / 256, nh), 256, 0, s>>>((const uint32_t*)d_src, (float4*)d_tmp, v_left, v_count, v_off, v_wts, (int)w);
    resize_h_kernel<<<dim3((nw + 63) / 64, (nh + 3) / 4), 256, 0, s>>>((const float4*)d_tmp, (uint32_t*)d_dst, h_left, h_count, h_off, h_wts, (int)w, (int)nw,
                                                                       (int)nh);
    return hipGetLastError();
}
