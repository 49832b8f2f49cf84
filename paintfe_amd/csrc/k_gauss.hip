// k_gauss.hip — separable Gaussian blur: parallel_gaussian_blur (ref: src/ops/filters.rs:242-316).
//
// Reference semantics kept: kernel radius ceil(3*sigma) built on the host (filters.rs:214-234), clamp-to-edge,
// all four channels blurred straight (not premultiplied), horizontal pass u8 -> f32, vertical pass f32 -> u8 with
// round-half-away, accumulation in ascending tap order.  EXACT=true evaluates `acc += src * kv` as a separate
// multiply and add (bit-exact with the CPU path); EXACT=false fuses them (fmaf), which changes a result only
// when the f32 sum sits within ~1e-5 of an x.5 boundary (+-1 LSB class).
//
// This filter is VALU/LDS-bound, not HBM-bound (97 taps x 2 passes x 4 channels = 776 MAC per pixel at sigma=16
// against 8 algorithmic bytes), so the design goal is MACs per LDS byte:
//   H pass: a lane owns 4 consecutive outputs; the row segment (+2r halo) is staged once in LDS as f32x4
//           (u8 -> f32 converted once per pixel, not once per tap); every ds_read_b128 feeds 16 MACs.
//           LDS rows are padded by one 16-byte slot per 4 pixels so the lane stride (4 px = 64 B) spreads over
//           all 64 banks (k_gauss.hip:lds_slot).
//   V pass: a lane owns 8 consecutive rows of one column; a 16-column x (128+2r)-row f32x4 tile is staged in LDS,
//           every ds_read_b128 feeds 32 MACs.
//   Tap weights are wave-uniform: they arrive through scalar loads, no VGPR or LDS traffic.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

constexpr int H_PX = 4;          // outputs per lane, horizontal
constexpr int H_THREADS = 256;
constexpr int H_TILE = H_PX * H_THREADS; // 1024 px per block
constexpr int V_PY = 8;          // outputs per lane, vertical
constexpr int V_TX = 16;         // columns per block
constexpr int V_YG = 16;         // row groups per block
constexpr int V_TILE_ROWS = V_PY * V_YG; // 128 output rows per block

PFX_DEV int lds_slot(int i) { return i + (i >> 2); } // one pad slot per 4 pixels

template <bool EXACT>
PFX_DEV void mac4(float4& acc, const float4 p, const float wv)
{
    if constexpr (EXACT) { // separate rounding of the product and the sum, like the reference
        acc.x = acc.x + p.x * wv; acc.y = acc.y + p.y * wv; acc.z = acc.z + p.z * wv; acc.w = acc.w + p.w * wv;
    } else {
        acc.x = __builtin_fmaf(p.x, wv, acc.x); acc.y = __builtin_fmaf(p.y, wv, acc.y);
        acc.z = __builtin_fmaf(p.z, wv, acc.z); acc.w = __builtin_fmaf(p.w, wv, acc.w);
    }
}

// wts points at tap 0 of a zero-padded array: wts[-8..-1] = 0 and wts[klen..klen+7] = 0.
template <bool EXACT>
__global__ __launch_bounds__(H_THREADS) void gauss_h_kernel(const uint8_t* __restrict__ src, float4* __restrict__ tmp,
                                                           const float* __restrict__ wts, int radius, int w, int h)
{
    extern __shared__ float4 tile[]; // lds_slot(H_TILE + 2r) entries
    const int y = blockIdx.y;
    const int x_tile = blockIdx.x * H_TILE;
    // +H_PX: the last lanes read up to H_PX-1 entries past their window (zero weight, but must be finite)
    const int n_in = min(H_TILE, w - x_tile) + 2 * radius + H_PX;
    const uint32_t* row = reinterpret_cast<const uint32_t*>(src) + (size_t)y * w;
    for (int i = threadIdx.x; i < n_in; i += H_THREADS) {
        int sx = min(max(x_tile - radius + i, 0), w - 1); // clamp-to-edge (filters.rs:268-270)
        uint32_t px = row[sx];
        tile[lds_slot(i)] = make_float4(ubyte0(px), ubyte1(px), ubyte2(px), ubyte3(px));
    }
    __syncthreads();

    const int x0 = x_tile + threadIdx.x * H_PX;
    if (x0 >= w) return;
    float4 acc[H_PX];
#pragma unroll
    for (int o = 0; o < H_PX; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int klen = 2 * radius + 1;
    const int base = threadIdx.x * H_PX;
    // input j (relative to this lane's window start) is tap (j - o) of output o
    for (int j = 0; j < klen + H_PX - 1; ++j) {
        const float4 p = tile[lds_slot(base + j)];
#pragma unroll
        for (int o = 0; o < H_PX; ++o) mac4<EXACT>(acc[o], p, wts[j - o]);
    }
    float4* out = tmp + (size_t)y * w + x0;
#pragma unroll
    for (int o = 0; o < H_PX; ++o)
        if (x0 + o < w) out[o] = acc[o];
}

template <bool EXACT>
__global__ __launch_bounds__(V_TX* V_YG) void gauss_v_kernel(const float4* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                            const float* __restrict__ wts, int radius, int w, int h)
{
    extern __shared__ float4 tile[]; // (V_TILE_ROWS + 2r) x V_TX
    const int x_tile = blockIdx.x * V_TX;
    const int y_tile = blockIdx.y * V_TILE_ROWS;
    const int rows_out = min(V_TILE_ROWS, h - y_tile);
    const int n_rows = rows_out + 2 * radius;
    const int tid = threadIdx.x;
    const int lx = tid % V_TX, yg = tid / V_TX;
    const int cols = min(V_TX, w - x_tile);
    for (int i = tid; i < n_rows * V_TX; i += V_TX * V_YG) {
        int r = i / V_TX, c = i % V_TX;
        int sy = min(max(y_tile - radius + r, 0), h - 1); // clamp-to-edge (filters.rs:296-298)
        int sx = x_tile + min(c, cols - 1);
        tile[i] = tmp[(size_t)sy * w + sx];
    }
    __syncthreads();

    const int y0 = y_tile + yg * V_PY;
    if (lx >= cols || y0 >= h) return;
    float4 acc[V_PY];
#pragma unroll
    for (int o = 0; o < V_PY; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int klen = 2 * radius + 1;
    const int last = min(klen + V_PY - 1, n_rows - yg * V_PY); // rows beyond the tile only feed outputs >= h
    for (int j = 0; j < last; ++j) {
        const float4 p = tile[(yg * V_PY + j) * V_TX + lx];
#pragma unroll
        for (int o = 0; o < V_PY; ++o) mac4<EXACT>(acc[o], p, wts[j - o]);
    }
    const int x = x_tile + lx;
#pragma unroll
    for (int o = 0; o < V_PY; ++o) {
        if (y0 + o < h) {
            const float4 a = acc[o];
            reinterpret_cast<uint32_t*>(dst)[(size_t)(y0 + o) * w + x] =
                pack_rgba(round_u8f(a.x), round_u8f(a.y), round_u8f(a.z), round_u8f(a.w)); // filters.rs:308-311
        }
    }
}

} // namespace

// LDS bound of the vertical tile: (128 + 2r) rows x 16 columns x 16 B <= 160 KiB
extern "C" int pfxk_gauss_max_radius(void) { return 256; }

template <bool EXACT>
static hipError_t launch_h(hipStream_t stream, const uint8_t* d_src, float4* tmp, const float* wts, int radius, uint32_t w, uint32_t h)
{
    const int h_entries = H_TILE + 2 * radius + H_PX;
    const size_t lds_h = (size_t)(h_entries + (h_entries >> 2) + 1) * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)gauss_h_kernel<EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
    if (e) return e;
    dim3 gh((w + H_TILE - 1) / H_TILE, h);
    gauss_h_kernel<EXACT><<<gh, H_THREADS, lds_h, stream>>>(d_src, tmp, wts, radius, (int)w, (int)h);
    return hipGetLastError();
}

template <bool EXACT>
static hipError_t launch_v(hipStream_t stream, const float4* tmp, uint8_t* d_dst, const float* wts, int radius, uint32_t w, uint32_t h)
{
    const size_t lds_v = (size_t)(V_TILE_ROWS + 2 * radius) * V_TX * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)gauss_v_kernel<EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);
    if (e) return e;
    dim3 gv((w + V_TX - 1) / V_TX, (h + V_TILE_ROWS - 1) / V_TILE_ROWS);
    gauss_v_kernel<EXACT><<<gv, V_TX * V_YG, lds_v, stream>>>(tmp, d_dst, wts, radius, (int)w, (int)h);
    return hipGetLastError();
}

// horizontal pass: u8 -> f32 intermediate (w*h*16 bytes)
extern "C" hipError_t pfxk_gauss_h(hipStream_t stream, const uint8_t* d_src, float* d_tmp, const float* d_wts_tap0, int radius,
                                   uint32_t w, uint32_t h, int exact)
{
    if (w == 0 || h == 0) return hipSuccess;
    return exact ? launch_h<true>(stream, d_src, reinterpret_cast<float4*>(d_tmp), d_wts_tap0, radius, w, h)
                 : launch_h<false>(stream, d_src, reinterpret_cast<float4*>(d_tmp), d_wts_tap0, radius, w, h);
}

// vertical pass: f32 intermediate -> u8 (round half away from zero)
extern "C" hipError_t pfxk_gauss_v(hipStream_t stream, const float* d_tmp, uint8_t* d_dst, const float* d_wts_tap0, int radius,
                                   uint32_t w, uint32_t h, int exact)
{
    if (w == 0 || h == 0) return hipSuccess;
    return exact ? launch_v<true>(stream, reinterpret_cast<const float4*>(d_tmp), d_dst, d_wts_tap0, radius, w, h)
                 : launch_v<false>(stream, reinterpret_cast<const float4*>(d_tmp), d_dst, d_wts_tap0, radius, w, h);
}
