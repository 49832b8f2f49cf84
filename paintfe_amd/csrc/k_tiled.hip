// k_tiled.hip — TiledImage's sparsity rule and small chunk-level helpers.
//
// Reference: src/canvas/tiled_image.rs:50-104 (from_rgba_image keeps a 64x64 chunk iff some alpha != 0),
// :271-293 (to_rgba_image: missing chunk -> zeros); src/ops/filters.rs:186-200 (selection copy-back);
// src/ops/adjustments.rs:144-205 (auto-levels min/max over selected, non-transparent pixels).
// All HBM-bound single-pass streaming kernels: one workgroup per 64x64 chunk, 16-byte accesses when rows are aligned.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

// lane layout shared with k_pointwise.hip: cg = tid % 16 -> 4 px, rows tid/16 + {0,16,32,48}
template <bool WRITE_IMAGE>
__global__ __launch_bounds__(256) void chunk_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                    uint8_t* __restrict__ populated, uint32_t w, uint32_t h)
{
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t bx = (blockIdx.x % cxn) * 64u, by = (blockIdx.x / cxn) * 64u;
    const uint32_t cg = threadIdx.x & 15u, r0 = threadIdx.x >> 4;
    const uint32_t x = bx + cg * 4u;
    const bool vec = (w & 3u) == 0u;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    uint32_t in[4][4];
    int any = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t y = by + r0 + 16u * k;
#pragma unroll
        for (int p = 0; p < 4; ++p) in[k][p] = 0u;
        if (y < h && x < w) {
            const size_t off = (size_t)y * w + x;
            if (vec) {
                const uint4 v = *reinterpret_cast<const uint4*>(s32 + off);
                in[k][0] = v.x; in[k][1] = v.y; in[k][2] = v.z; in[k][3] = v.w;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p) if (x + p < w) in[k][p] = s32[off + p];
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) any |= (int)(in[k][p] >> 24);
    }
    const int pop = __syncthreads_or(any);
    if (populated && threadIdx.x == 0) populated[blockIdx.x] = (uint8_t)(pop != 0);
    if constexpr (WRITE_IMAGE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t y = by + r0 + 16u * k;
            if (!(y < h && x < w)) continue;
            const size_t off = (size_t)y * w + x;
            if (!pop) { in[k][0] = in[k][1] = in[k][2] = in[k][3] = 0u; }
            if (vec) *reinterpret_cast<uint4*>(d32 + off) = make_uint4(in[k][0], in[k][1], in[k][2], in[k][3]);
            else {
#pragma unroll
                for (int p = 0; p < 4; ++p) if (x + p < w) d32[off + p] = in[k][p];
            }
        }
    }
}

// TiledImage <-> flat image.  One workgroup per 64x64 chunk of the canvas; slot[c] = index of the chunk in the packed array
// (64*64 px each, edge chunks zero-padded) or 0xffffffff = the TiledImage has no such chunk.
//   IMPORT: to_rgba_image (tiled_image.rs:271-293)  flat <- chunk pixels, zeros where no chunk exists
//   export: from_rgba_image (:50-104)               chunk <- flat pixels, zero-padded past the canvas edge
template <bool IMPORT>
__global__ __launch_bounds__(256) void chunk_copy_kernel(uint32_t* __restrict__ flat, uint32_t* __restrict__ packed,
                                                         const uint32_t* __restrict__ slot, uint32_t w, uint32_t h)
{
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t bx = (blockIdx.x % cxn) * 64u, by = (blockIdx.x / cxn) * 64u;
    const uint32_t sl = slot[blockIdx.x];
    if (!IMPORT && sl == 0xffffffffu) return;
    const uint32_t cg = threadIdx.x & 15u, r0 = threadIdx.x >> 4;
    const uint32_t x = bx + cg * 4u;
    const bool vec = (w & 3u) == 0u;
    uint32_t* chunk = packed + (size_t)sl * 4096u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t ly = r0 + 16u * k, y = by + ly;
        uint4* cp = reinterpret_cast<uint4*>(chunk + ly * 64u + cg * 4u);
        if constexpr (IMPORT) {
            if (!(y < h && x < w)) continue;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (sl != 0xffffffffu) v = *cp;
            const size_t off = (size_t)y * w + x;
            if (vec) *reinterpret_cast<uint4*>(flat + off) = v;
            else {
                const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int p = 0; p < 4; ++p) if (x + p < w) flat[off + p] = e[p];
            }
        } else {
            uint32_t e[4] = {0u, 0u, 0u, 0u};
            if (y < h && x < w) {
                const size_t off = (size_t)y * w + x;
                if (vec) { const uint4 v = *reinterpret_cast<const uint4*>(flat + off); e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
                else {
#pragma unroll
                    for (int p = 0; p < 4; ++p) if (x + p < w) e[p] = flat[off + p];
                }
            }
            *cp = make_uint4(e[0], e[1], e[2], e[3]);
        }
    }
}

__global__ __launch_bounds__(256) void select_kernel(const uint32_t* __restrict__ src, const uint32_t* __restrict__ fx,
                                                     const uint8_t* __restrict__ mask, uint32_t* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (mask[i] > 0) ? fx[i] : src[i];
}

__global__ __launch_bounds__(256) void minmax_kernel(const uint32_t* __restrict__ src, const uint8_t* __restrict__ mask,
                                                     size_t n, uint32_t* __restrict__ out6)
{
    uint32_t mn[3] = {255u, 255u, 255u}, mx[3] = {0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (mask && mask[i] == 0) continue;
        const uint32_t px = src[i];
        if ((px >> 24) == 0u) continue; // skip fully transparent (adjustments.rs:186)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t v = (px >> (8 * c)) & 0xffu;
            mn[c] = min(mn[c], v);
            mx[c] = max(mx[c], v);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) { // 64-wide wave reduction
            mn[c] = min(mn[c], (uint32_t)__shfl_xor((int)mn[c], off, 64));
            mx[c] = max(mx[c], (uint32_t)__shfl_xor((int)mx[c], off, 64));
        }
    }
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            atomicMin(&out6[c * 2 + 0], mn[c]);
            atomicMax(&out6[c * 2 + 1], mx[c]);
        }
    }
}

__global__ void minmax_init_kernel(uint32_t* out6)
{
    if (threadIdx.x < 6) out6[threadIdx.x] = (threadIdx.x & 1u) ? 0u : 255u;
}

} // namespace

extern "C" hipError_t pfxk_chunk_populated(hipStream_t s, const uint8_t* d_src, uint32_t w, uint32_t h, uint8_t* d_populated)
{
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    chunk_kernel<false><<<nchunks, 256, 0, s>>>(d_src, nullptr, d_populated, w, h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_tiled_roundtrip(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    chunk_kernel<true><<<nchunks, 256, 0, s>>>(d_src, d_dst, nullptr, w, h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_chunks_import(hipStream_t s, const uint8_t* d_packed, const uint32_t* d_slot, uint32_t w, uint32_t h, uint8_t* d_flat)
{
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    chunk_copy_kernel<true><<<nchunks, 256, 0, s>>>((uint32_t*)d_flat, (uint32_t*)const_cast<uint8_t*>(d_packed), d_slot, w, h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_chunks_export(hipStream_t s, const uint8_t* d_flat, const uint32_t* d_slot, uint32_t w, uint32_t h, uint8_t* d_packed)
{
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    chunk_copy_kernel<false><<<nchunks, 256, 0, s>>>((uint32_t*)const_cast<uint8_t*>(d_flat), (uint32_t*)d_packed, d_slot, w, h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_select_by_mask(hipStream_t s, const uint8_t* d_src, const uint8_t* d_fx, const uint8_t* d_mask,
                                          uint8_t* d_dst, uint32_t w, uint32_t h)
{
    const size_t n = (size_t)w * h;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    select_kernel<<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_src, (const uint32_t*)d_fx, d_mask, (uint32_t*)d_dst, n);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_minmax_rgb(hipStream_t s, const uint8_t* d_src, const uint8_t* d_mask, uint32_t w, uint32_t h,
                                      uint32_t* d_out6)
{
    const size_t n = (size_t)w * h;
    minmax_init_kernel<<<1, 64, 0, s>>>(d_out6);
    if (n == 0) return hipGetLastError();
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    minmax_kernel<<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_src, d_mask, n, d_out6);
    return hipGetLastError();
}
