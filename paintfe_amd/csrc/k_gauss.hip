// k_gauss.hip — separable Gaussian blur: parallel_gaussian_blur (ref: src/ops/filters.rs:242-316).
//
// Reference semantics kept: kernel radius ceil(3*sigma) built on the host (filters.rs:214-234), clamp-to-edge,
// all four channels blurred straight (not premultiplied), horizontal pass u8 -> f32, vertical pass f32 -> u8 with
// round-half-away, accumulation in ascending tap order.  EXACT=true evaluates `acc += src * kv` as a separate
// multiply and add (bit-exact with the CPU path); EXACT=false fuses them (fmaf), which changes a result only
// when the f32 sum sits within ~1e-5 of an x.5 boundary (+-1 LSB class).
//
// This filter is VALU-bound, not HBM-bound (97 taps x 2 passes x 4 channels = 776 MAC per pixel at sigma=16
// against 8 algorithmic bytes), so the design goal is MACs per issued instruction and per LDS byte:
//   H pass: a lane owns 4 consecutive outputs; the row segment (+2r halo) is staged once in LDS as f32x4
//           (u8 -> f32 converted once per pixel, not once per tap); every ds_read_b128 feeds 16 MACs.
//           LDS rows are padded by one 16-byte slot per 4 pixels so the lane stride (4 px = 64 B) spreads over
//           all 64 banks; inputs are consumed in groups of 4 so the padded address is `base + 80*g + 16*i`
//           (one VALU add per 64 MACs, immediate offsets on the ds_reads).
//   V pass: a lane owns 8 consecutive rows of one column; a 16-column x (128+2r)-row f32x4 tile is staged in LDS,
//           every ds_read_b128 feeds 32 MACs.
//   Both inner loops are software-pipelined: the LDS reads and the scalar weight loads of step g+1 are issued
//   before the MACs of step g, so the wave's own MACs cover the LDS latency (occupancy is LDS-limited).
//   Tap weights are wave-uniform: they arrive through scalar loads, no VGPR or LDS traffic.
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

constexpr int H_PX = 4;          // outputs per lane, horizontal
constexpr int H_THREADS = 256;
constexpr int H_TILE = H_PX * H_THREADS; // 1024 px per block
constexpr int H_SLACK = 12;      // tile entries past the last needed input (group padding + one prefetched group)
constexpr int V_SLACK = 4;       // rows past the last needed input (group padding)
constexpr int W_PAD = 16;        // zeros on both sides of the weight array

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

// wts points at tap 0 of a zero-padded array: wts[-W_PAD..-1] = 0 and wts[klen..klen+W_PAD-1] = 0.
template <bool EXACT>
__global__ __launch_bounds__(H_THREADS) void gauss_h_kernel(const uint8_t* __restrict__ src, float4* __restrict__ tmp,
                                                           const float* __restrict__ wts, int radius, int w, int h)
{
    extern __shared__ float4 tile[]; // lds_slot(H_TILE + 2r + H_SLACK) entries
    const int y = blockIdx.y;
    const int x_tile = blockIdx.x * H_TILE;
    const int n_in = min(H_TILE, w - x_tile) + 2 * radius + H_SLACK; // entries past the window carry zero weight
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
    const int groups = (klen + H_PX - 1 + 3) >> 2; // inputs 0 .. klen+H_PX-2, padded to a multiple of 4
    // input j (relative to this lane's window start) is tap (j - o) of output o; j = 4g + i
    const float4* p = tile + lds_slot(threadIdx.x * H_PX); // 4-aligned start: slot(base + 4g + i) = slot(base) + 5g + i
    float4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];
    for (int g = 0; g < groups; ++g) {
        p += 5;
        const float4 b0 = p[0], b1 = p[1], b2 = p[2], b3 = p[3]; // prefetch group g+1 (within H_SLACK)
        const float* wg = wts + 4 * g;                           // 7 uniform taps: wg[-3] .. wg[3]
        const float wm3 = wg[-3], wm2 = wg[-2], wm1 = wg[-1], w0 = wg[0], w1 = wg[1], w2 = wg[2], w3 = wg[3];
        mac4<EXACT>(acc[0], a0, w0);  mac4<EXACT>(acc[1], a0, wm1); mac4<EXACT>(acc[2], a0, wm2); mac4<EXACT>(acc[3], a0, wm3);
        mac4<EXACT>(acc[0], a1, w1);  mac4<EXACT>(acc[1], a1, w0);  mac4<EXACT>(acc[2], a1, wm1); mac4<EXACT>(acc[3], a1, wm2);
        mac4<EXACT>(acc[0], a2, w2);  mac4<EXACT>(acc[1], a2, w1);  mac4<EXACT>(acc[2], a2, w0);  mac4<EXACT>(acc[3], a2, wm1);
        mac4<EXACT>(acc[0], a3, w3);  mac4<EXACT>(acc[1], a3, w2);  mac4<EXACT>(acc[2], a3, w1);  mac4<EXACT>(acc[3], a3, w0);
        a0 = b0; a1 = b1; a2 = b2; a3 = b3;
    }
    float4* out = tmp + (size_t)y * w + x0;
#pragma unroll
    for (int o = 0; o < H_PX; ++o)
        if (x0 + o < w) out[o] = acc[o];
}

// Vertical pass.  TX columns x (4*YG) output rows per block, a lane owns 4 consecutive rows of one column (every
// ds_read_b128 feeds 16 MACs, like the H pass).  RS = LDS row stride in float4 (>= TX; chosen so that the 16-lane
// ds_read_b128 groups hit 16 distinct 16-byte slots: see tools/lds_stride_search.py).  The halo (2r rows) dominates
// the tile, so blocks are tall (4*YG = 256 rows) and narrow: LDS per wave stays small and 4+ waves per SIMD cover
// the LDS and staging latency.
template <bool EXACT, int TX, int YG, int RS>
__global__ __launch_bounds__(TX* YG) void gauss_v_kernel(const float4* __restrict__ tmp, uint8_t* __restrict__ dst,
                                                         const float* __restrict__ wts, int radius, int w, int h)
{
    constexpr int PY = 4, NT = TX * YG, ROWS = PY * YG;
    extern __shared__ float4 tile[]; // (ROWS + 2r + V_SLACK) x RS
    const int x_tile = blockIdx.x * TX;
    const int y_tile = blockIdx.y * ROWS;
    const int n_rows = ROWS + 2 * radius + V_SLACK; // always the full tile: rows past the image clamp to h-1
    const int tid = threadIdx.x;
    const int lx = tid % TX, yg = tid / TX;
    const int cols = min(TX, w - x_tile);
    const int total = n_rows * TX;
    auto src_of = [&](int i) -> const float4* {
        i = min(i, total - 1);
        const int r = i / TX, c = i % TX;
        const int sy = min(max(y_tile - radius + r, 0), h - 1); // clamp-to-edge (filters.rs:296-298)
        return tmp + (size_t)sy * w + x_tile + min(c, cols - 1);
    };
    auto put = [&](int i, const float4 v) {
        if (i < total) tile[(i / TX) * RS + (i % TX)] = v;
    };
    for (int i0 = tid; i0 < total; i0 += NT * 4) { // 4 independent 16-byte loads in flight per lane
        const float4 v0 = *src_of(i0), v1 = *src_of(i0 + NT), v2 = *src_of(i0 + 2 * NT), v3 = *src_of(i0 + 3 * NT);
        put(i0, v0); put(i0 + NT, v1); put(i0 + 2 * NT, v2); put(i0 + 3 * NT, v3);
    }
    __syncthreads();

    const int y0 = y_tile + yg * PY;
    if (lx >= cols || y0 >= h) return;
    float4 acc[PY];
#pragma unroll
    for (int o = 0; o < PY; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int klen = 2 * radius + 1;
    const int groups = (klen + PY - 1 + 3) >> 2; // input rows 0 .. klen+PY-2, padded to a multiple of 4
    // input row j (relative to this lane's first row) is tap (j - o) of output o; j = 4g + i
    const float4* p = tile + (yg * PY) * RS + lx;
    for (int g = 0; g < groups; ++g) {
        const float4 a0 = p[0], a1 = p[RS], a2 = p[2 * RS], a3 = p[3 * RS];
        p += 4 * RS;
        const float* wg = wts + 4 * g; // 7 uniform taps: wg[-3] .. wg[3]
        const float wm3 = wg[-3], wm2 = wg[-2], wm1 = wg[-1], w0 = wg[0], w1 = wg[1], w2 = wg[2], w3 = wg[3];
        mac4<EXACT>(acc[0], a0, w0);  mac4<EXACT>(acc[1], a0, wm1); mac4<EXACT>(acc[2], a0, wm2); mac4<EXACT>(acc[3], a0, wm3);
        mac4<EXACT>(acc[0], a1, w1);  mac4<EXACT>(acc[1], a1, w0);  mac4<EXACT>(acc[2], a1, wm1); mac4<EXACT>(acc[3], a1, wm2);
        mac4<EXACT>(acc[0], a2, w2);  mac4<EXACT>(acc[1], a2, w1);  mac4<EXACT>(acc[2], a2, w0);  mac4<EXACT>(acc[3], a2, wm1);
        mac4<EXACT>(acc[0], a3, w3);  mac4<EXACT>(acc[1], a3, w2);  mac4<EXACT>(acc[2], a3, w1);  mac4<EXACT>(acc[3], a3, w0);
    }
    const int x = x_tile + lx;
#pragma unroll
    for (int o = 0; o < PY; ++o) {
        if (y0 + o < h) {
            const float4 a = acc[o];
            reinterpret_cast<uint32_t*>(dst)[(size_t)(y0 + o) * w + x] =
                pack_rgba(round_u8f(a.x), round_u8f(a.y), round_u8f(a.z), round_u8f(a.w)); // filters.rs:308-311
        }
    }
}

template <bool EXACT>
hipError_t launch_h(hipStream_t stream, const uint8_t* d_src, float4* tmp, const float* wts, int radius, uint32_t w, uint32_t h)
{
    const int h_entries = H_TILE + 2 * radius + H_SLACK;
    const size_t lds_h = (size_t)(h_entries + (h_entries >> 2) + 1) * sizeof(float4);
    hipError_t e = hipFuncSetAttribute((const void*)gauss_h_kernel<EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_h);
    if (e) return e;
    dim3 gh((w + H_TILE - 1) / H_TILE, h);
    gauss_h_kernel<EXACT><<<gh, H_THREADS, lds_h, stream>>>(d_src, tmp, wts, radius, (int)w, (int)h);
    return hipGetLastError();
}

template <bool EXACT, int TX, int YG, int RS>
hipError_t launch_v_cfg(hipStream_t stream, const float4* tmp, uint8_t* d_dst, const float* wts, int radius, uint32_t w, uint32_t h)
{
    constexpr int ROWS = 4 * YG;
    const size_t lds_v = (size_t)(ROWS + 2 * radius + V_SLACK) * RS * sizeof(float4);
    if (lds_v > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void*)gauss_v_kernel<EXACT, TX, YG, RS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v);
    if (e) return e;
    dim3 gv((w + TX - 1) / TX, (h + ROWS - 1) / ROWS);
    gauss_v_kernel<EXACT, TX, YG, RS><<<gv, TX * YG, lds_v, stream>>>(tmp, d_dst, wts, radius, (int)w, (int)h);
    return hipGetLastError();
}

int g_v_cfg = 0; // tuning knob (pfxk_gauss_set_v_config); 0 is the shipped configuration

template <bool EXACT>
hipError_t launch_v(hipStream_t stream, const float4* tmp, uint8_t* d_dst, const float* wts, int radius, uint32_t w, uint32_t h)
{
    int cfg = g_v_cfg;
    if (radius > 380) cfg = 2; // narrowest tile for huge radii (LDS bound)
    switch (cfg) {
    case 1: return launch_v_cfg<EXACT, 16, 64, 16>(stream, tmp, d_dst, wts, radius, w, h);
    case 2: return launch_v_cfg<EXACT, 4, 64, 5>(stream, tmp, d_dst, wts, radius, w, h);
    case 3: return launch_v_cfg<EXACT, 8, 64, 10>(stream, tmp, d_dst, wts, radius, w, h);
    // shipped: 8 columns x 128 rows, 256 threads.  Measured at 8K, sigma=16 (profiles/r01_tuning.md): 0.344 ms vs
    // 0.368 (8x256 rows), 0.350 (4x256), 0.446 (16x256)
    default: return launch_v_cfg<EXACT, 8, 32, 10>(stream, tmp, d_dst, wts, radius, w, h);
    }
}

} // namespace

// LDS bounds: H tile (1024 + 2r + 12) x 20 B and the narrowest V tile (256 + 2r + 4) rows x 5 x 16 B <= 160 KiB
extern "C" int pfxk_gauss_max_radius(void) { return 850; }
extern "C" void pfxk_gauss_set_v_config(int cfg) { g_v_cfg = cfg; }
extern "C" int pfxk_gauss_weight_pad(void) { return W_PAD; }

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
