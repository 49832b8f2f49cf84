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
    dst[(size_t)oy * nw + ox] = pack_rgba(round_u8f(t0), round_u8f(t1), round_u8f(t2), round_u8f(t3));
}

} // namespace

// tables: v_* index by output row (nh entries), h_* by output column (nw entries)
extern "C" hipError_t pfxk_resize(hipStream_t s, const uint8_t* d_src, float* d_tmp, uint8_t* d_dst, const uint32_t* v_left, const uint32_t* v_count,
                                  const uint32_t* v_off, const float* v_wts, const uint32_t* h_left, const uint32_t* h_count, const uint32_t* h_off,
                                  const float* h_wts, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh)
{
    (void)h;
    if (w == 0 || nw == 0 || nh == 0) return hipSuccess;
    resize_v_kernel<<<dim3((w + 255) / 256, nh), 256, 0, s>>>((const uint32_t*)d_src, (float4*)d_tmp, v_left, v_count, v_off, v_wts, (int)w);
    resize_h_kernel<<<dim3((nw + 63) / 64, (nh + 3) / 4), 256, 0, s>>>((const float4*)d_tmp, (uint32_t*)d_dst, h_left, h_count, h_off, h_wts, (int)w, (int)nw,
                                                                       (int)nh);
    return hipGetLastError();
}
