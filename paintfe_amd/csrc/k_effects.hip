// k_effects.hip — effects that reuse the hot-path kernels: sharpen / glow (Gaussian + a two-input
// pointwise pass), bokeh (equal-weight disc) blur, motion blur.
//
// Reference: glow_core src/ops/effects/stylize.rs:26-70, sharpen_core :96-143,
//            bokeh_blur_core src/ops/effects/blur.rs:22-115, motion_blur_core :144-210.
// All are 4 B read (+ window) / 4 B written per pixel; bokeh and motion are gathers served by L1/L2 (the window of
// neighbouring lanes overlaps almost entirely).  Bit-exact classes: bokeh (integer sums, one f32 scale) and motion
// (f32 sums of integers < 2^24 are exact in any order); sharpen / glow inherit the Gaussian's class.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

// dst = mask ? op(src, blur) : src ; alpha always from src
template <int OP>
__global__ __launch_bounds__(256) void combine_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ blur,
                                                      const uint8_t* __restrict__ mask, uint32_t* __restrict__ dst, size_t n,
                                                      float p0)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t s = src[i];
        if (mask && mask[i] == 0) { dst[i] = s; continue; }
        const uint32_t b = blur[i];
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float sv = (float)((s >> (8 * c)) & 0xffu), bv = (float)((b >> (8 * c)) & 0xffu);
            if constexpr (OP == PFXK_FX_SHARPEN) { // stylize.rs:135-137: s + amount * (s - b)
                o[c] = sv + p0 * (sv - bv);
            } else {                               // stylize.rs:60-64: screen blend 1 - (1 - s)(1 - b*intensity)
                const float sn = div255(sv), bn = div255(bv);
                const float result = 1.0f - (1.0f - sn) * (1.0f - bn * p0);
                o[c] = result * 255.0f;
            }
        }
        // `.round().clamp(0, 255) as u8` of the three colour channels written over the source pixel's bytes: its alpha stays
        uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[0]), 0, s);
        px = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[1]), 1, px);
        dst[i] = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(o[2]), 2, px);
    }
}

// spans: (dy, half-width) pairs of the disc rows (built on the host like blur.rs:35-46)
__global__ __launch_bounds__(256) void bokeh_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                    const uint8_t* __restrict__ mask, const int2* __restrict__ spans, int n_spans,
                                                    float inv_count, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    if (mask && mask[oi] == 0) { dst[oi] = src[oi]; return; }
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0; // sums fit 32 bits: (2r+1)^2 * 255 < 2^32 for r < 2000
    for (int k = 0; k < n_spans; ++k) {
        const int2 sp = spans[k]; // uniform -> scalar load
        const uint32_t* row = src + (size_t)min(max(y + sp.x, 0), h - 1) * w;
        for (int dx = -sp.y; dx <= sp.y; ++dx) {
            const uint32_t p = row[min(max(x + dx, 0), w - 1)];
            t0 += p & 0xffu; t1 += (p >> 8) & 0xffu; t2 += (p >> 16) & 0xffu; t3 += p >> 24;
        }
    }
    // `totals[c] as f32 * inv_count` (u64 -> f32 conversion rounds to nearest, same as u32 -> f32 for these magnitudes)
    dst[oi] = pack_round_rgba((float)t0 * inv_count, (float)t1 * inv_count, (float)t2 * inv_count, (float)t3 * inv_count);
}

PFX_DEV int rs_round_i32(float v) // `.round() as i32`
{
    const float r = __builtin_roundf(v);
    if (r != r) return 0;
    if (r >= 2147483648.0f) return 2147483647;
    if (r <= -2147483648.0f) return (-2147483647 - 1);
    return (int)r;
}

__global__ __launch_bounds__(256) void motion_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                     const uint8_t* __restrict__ mask, int steps, float dx, float dy,
                                                     float inv_steps, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    if (mask && mask[oi] == 0) { dst[oi] = src[oi]; return; }
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int i = -steps; i <= steps; ++i) { // blur.rs:190-201
        const int sx = min(max(rs_round_i32((float)x + (float)i * dx), 0), w - 1);
        const int sy = min(max(rs_round_i32((float)y + (float)i * dy), 0), h - 1);
        const uint32_t p = src[(size_t)sy * w + sx];
        s0 += ubyte0(p); s1 += ubyte1(p); s2 += ubyte2(p); s3 += ubyte3(p);
    }
    dst[oi] = pack_round_rgba(s0 * inv_steps, s1 * inv_steps, s2 * inv_steps, s3 * inv_steps);
}

} // namespace

extern "C" hipError_t pfxk_combine(hipStream_t s, const uint8_t* d_src, const uint8_t* d_blur, const uint8_t* d_mask,
                                   uint8_t* d_dst, uint32_t w, uint32_t h, int op, float p0)
{
    const size_t n = (size_t)w * h;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (op == PFXK_FX_SHARPEN)
        combine_kernel<PFXK_FX_SHARPEN><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_src, (const uint32_t*)d_blur, d_mask, (uint32_t*)d_dst, n, p0);
    else
        combine_kernel<PFXK_FX_GLOW><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_src, (const uint32_t*)d_blur, d_mask, (uint32_t*)d_dst, n, p0);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_bokeh(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask,
                                 const int32_t* d_spans_dy_hw, int n_spans, float inv_count, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    dim3 g((w + 63) / 64, (h + 3) / 4);
    bokeh_kernel<<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (const int2*)d_spans_dy_hw, n_spans, inv_count, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_motion(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int steps,
                                  float dx, float dy, float inv_steps, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    dim3 g((w + 63) / 64, (h + 3) / 4);
    motion_kernel<<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, steps, dx, dy, inv_steps, (int)w, (int)h);
    return hipGetLastError();
}
