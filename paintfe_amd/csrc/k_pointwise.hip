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
#include "k_pointwise.h"

using namespace pfxk;
using namespace pfxk::pw;

namespace {

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

// A chain of pointwise ops in ONE pass (pfx_chain_dev): the same chunk tiling, dense mode, no selection.  A lane's 16 pixels go through the ops one op at a time
// (k_pointwise.h: chain_apply); tables of the chain's LUT ops are staged in LDS once per workgroup.
__global__ __launch_bounds__(256) void pointwise_chain_kernel(const pfxk_chain C_arg /* first: read through the kernarg segment (k_pointwise.h) */, const uint8_t* __restrict__ src,
                                                              uint8_t* __restrict__ dst, const uint8_t* __restrict__ luts_g, uint32_t w, uint32_t h)
{
    const chain_kptr C = chain_in_kernarg();
    __shared__ uint8_t luts[PFXK_CHAIN_LUTS * 1024];
    if (C->n_luts) {
        for (uint32_t i = threadIdx.x; i < C->n_luts * 256u; i += 256u) reinterpret_cast<uint32_t*>(luts)[i] = reinterpret_cast<const uint32_t*>(luts_g)[i];
        __syncthreads();
    }
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const uint32_t bx = (bid % cxn) * 64u, by = (bid / cxn) * 64u;
    const uint32_t cg = threadIdx.x & 15u, r0 = threadIdx.x >> 4;
    const uint32_t x = bx + cg * 4u;
    const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d32 = reinterpret_cast<uint32_t*>(dst);
    uint32_t px[16];
    if ((w & 3u) == 0u && bx + 64u <= w && by + 64u <= h) {   // interior chunk, rows 16-byte aligned: four 16-byte loads and stores per lane
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint4 v = *reinterpret_cast<const uint4*>(s32 + (size_t)(by + r0 + 16u * k) * w + x);
            px[4 * k] = v.x; px[4 * k + 1] = v.y; px[4 * k + 2] = v.z; px[4 * k + 3] = v.w;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {   // a row's four pixels at a time: 16 at once cost 206 registers (two waves per SIMD on an HBM-bound kernel)
            uint32_t q[4] = {px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]};
            chain_apply<4>(C, luts, q);
            px[4 * k] = q[0]; px[4 * k + 1] = q[1]; px[4 * k + 2] = q[2]; px[4 * k + 3] = q[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            *reinterpret_cast<uint4*>(d32 + (size_t)(by + r0 + 16u * k) * w + x) = make_uint4(px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t y = by + r0 + 16u * k;
            px[4 * k + p] = (y < h && x + p < w) ? s32[(size_t)y * w + x + p] : 0u;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t q[4] = {px[4 * k], px[4 * k + 1], px[4 * k + 2], px[4 * k + 3]};
        chain_apply<4>(C, luts, q);
        px[4 * k] = q[0]; px[4 * k + 1] = q[1]; px[4 * k + 2] = q[2]; px[4 * k + 3] = q[3];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t y = by + r0 + 16u * k;
            if (y < h && x + p < w) d32[(size_t)y * w + x + p] = px[4 * k + p];
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

extern "C" hipError_t pfxk_pointwise_chain(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_luts, const pfxk_chain* C, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0 || C->n == 0) return hipSuccess;
    if (C->n > PFXK_CHAIN_MAX || C->n_luts > PFXK_CHAIN_LUTS) return hipErrorInvalidValue;
    const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
    pointwise_chain_kernel<<<nchunks, 256, 0, s>>>(*C, d_src, d_dst, d_luts, w, h);
    return hipGetLastError();
}
