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
#include <type_traits>

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

unsigned long long* g_dbg_buf = nullptr; // development: phase timestamps of gauss_strip_kernel (pfxk_gauss_set_dbg_buf)
int g_v_cfg = 0; // tuning knob (pfxk_gauss_set_v_config); 0 is the shipped configuration

template <bool EXACT>
hipError_t launch_v(hipStream_t stream, const float4* tmp, uint8_t* d_dst, const float* wts, int radius, uint32_t w, uint32_t h)
{
    int cfg = g_v_cfg & 0xff;
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


// ---- fused H+V Gaussian on the matrix cores (default mode, radius <= 48) -------------------------------------------------------
// Both passes are banded-Toeplitz products  D[m][n] = sum_k A[m][k] * T[k][n],  T[k][n] = w[k - n + r]  — a genuine contraction,
// so they run on v_mfma_f32_32x32x16_f16 (16x the f32 FMA rate) while the VALU only converts and packs:
//   * operands are split so that every product is exact in the f32 accumulator: a u8 sample is exact in f16; a weight is
//     scaled by a power of two S and split w*S = w1 + w2 (two f16, 22 significant bits); the f32 horizontal result h is split
//     h = h1 + h2 the same way.  H pass: p*w1 + p*w2.  V pass: h1*w1 + (h1*w2 + h2*w1); the dropped h2*w2 is < 2^-22 of the sum.
//     Large and small terms accumulate in separate accumulators.  The result differs from the CPU path's f32 mul/add chain by
//     rounding noise of the same order as the FMA-contracted VALU kernels (+-1 LSB class, tests assert it).
//   * one workgroup (8 waves) produces an R x 32 output tile: the H pass computes (R + 2*R8) rows x 32 columns straight from the
//     RGBA8 source (each wave 8 rows x 4 channels at a time: A = 32 (channel, row) lines x 16 source columns converted in registers, T fragments
//     resident in VGPRs for the whole kernel) and leaves them in LDS as f16 pairs, transposed so that a column's rows are
//     contiguous; the V pass reads its A fragments from there with ds_read_b128 (A = 8 columns x 4 channels, k = source rows),
//     rounds, packs RGBA and stages the tile in LDS for 128-byte row stores.  The f32 intermediate never touches HBM:
//     8 algorithmic bytes per pixel are the kernel's only HBM traffic (+ halo re-reads served by L2).
//   * accumulation order inside the MFMA is the hardware's; the contraction index is only ever paired A-slot with B-slot, so
//     the kernel relies on nothing but the documented C/D map (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
typedef _Float16 pfx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pfx_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pfx_f16x2 __attribute__((ext_vector_type(2)));
typedef float pfx_f32x16 __attribute__((ext_vector_type(16)));
// v_cvt_pkrtz_f16_f32: two f32 -> packed f16, round toward zero (exact for the integers 0..255)
PFX_DEV pfx_f16x2 pkrtz(float a, float b) { return __builtin_bit_cast(pfx_f16x2, __builtin_amdgcn_cvt_pkrtz(a, b)); }

constexpr int GM_COLS = 32;             // output columns per tile (one MFMA N block)
constexpr int GM_MAXR = 48;             // largest radius (sigma <= 16)
constexpr int GM_ROWS = 224;            // LDS rows of H results per tile: R + 2 * R8 <= 224
constexpr int GM_YP = 232;              // row pitch (f16) of one (part, channel, column) line: 224 + 8
constexpr int GM_PLANE = GM_COLS * GM_YP + 32; // (part, channel) plane stride in f16, +64 bytes: with the 29-slot column pitch (mod 16 = 13)
                                               // the V pass's ds_read_b128 lane groups (4 columns x 4 channels) hit 16 distinct 16-byte slots
constexpr int GM_OUT_PITCH = 33;        // dwords per staged output row
constexpr int GM_MAX_R = 192;           // largest R (small radii)
constexpr int GM_WOFF = 48;             // wsplit[GM_WOFF + t] = tap t; zeros elsewhere
constexpr int GM_WLEN = 192;            // entries per weight part
constexpr size_t GM_LDS = (size_t)8 * GM_PLANE * 2 + (size_t)GM_MAX_R * GM_OUT_PITCH * 4;

// RGBA8 -> four u8 planes (plane pitch = plane_stride bytes): the matrix-core kernel's A fragments are 8 consecutive samples of
// ONE channel, so it reads each source byte exactly once per tile (interleaved RGBA would be fetched by four lanes each).
__global__ __launch_bounds__(256) void gauss_planarize_kernel(const uint32_t* __restrict__ src, uint8_t* __restrict__ planes, size_t n_px,
                                                              size_t plane_stride)
{
    const size_t n4 = n_px >> 2;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(src)[q];
        // byte k of the four pixels -> one dword per plane
        const uint32_t lo01 = __builtin_amdgcn_perm(v.y, v.x, 0x05010400u), hi01 = __builtin_amdgcn_perm(v.y, v.x, 0x07030602u);
        const uint32_t lo23 = __builtin_amdgcn_perm(v.w, v.z, 0x05010400u), hi23 = __builtin_amdgcn_perm(v.w, v.z, 0x07030602u);
        // lo01 = [g1 g0 r1 r0] (bytes 3..0), hi01 = [a1 a0 b1 b0]
        reinterpret_cast<uint32_t*>(planes)[q] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);                        // r3 r2 r1 r0
        reinterpret_cast<uint32_t*>(planes + plane_stride)[q] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);          // g
        reinterpret_cast<uint32_t*>(planes + 2 * plane_stride)[q] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);      // b
        reinterpret_cast<uint32_t*>(planes + 3 * plane_stride)[q] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);      // a
    }
    for (size_t p = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_px; p += (size_t)gridDim.x * blockDim.x) {
        const uint32_t v = src[p];
        planes[p] = (uint8_t)v; planes[plane_stride + p] = (uint8_t)(v >> 8);
        planes[2 * plane_stride + p] = (uint8_t)(v >> 16); planes[3 * plane_stride + p] = (uint8_t)(v >> 24);
    }
}

// FAST: every tile of the launch reads its source window with 16-byte loads (window inside the image, 16-byte aligned rows);
// otherwise samples are fetched one by one with clamp-to-edge.  A launch covers the tile columns [col_a, col_a + n_a) followed by
// [col_b, col_b + tiles_x - n_a): the host sends the interior columns to the FAST instantiation and the border columns to the other.
template <bool FAST, int NKB>
__global__ __launch_bounds__(512, 1) void gauss_mfma_kernel(const uint8_t* __restrict__ planes, size_t plane_stride, uint8_t* __restrict__ dst,
                                                            const uint16_t* __restrict__ wsplit, int w, int h, int r, int R8,
                                                            int R, float inv_scale2, float bias_c, int tiles_x, int n_tiles, int col_a, int n_a, int col_b, int y_phase, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_lds[];
    _Float16* HR = reinterpret_cast<_Float16*>(gm_lds);                           // [part][c][x][GM_YP]
    uint32_t* OUT = reinterpret_cast<uint32_t*>(gm_lds + (size_t)8 * GM_PLANE * 2); // [R][GM_OUT_PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;

    // Toeplitz fragments.  The 16 NKB samples of a window are dealt to the two lane halves as two contiguous runs (half hh owns
    // samples [8 NKB hh, 8 NKB (hh + 1)), K block kb takes 8 of each run): an A lane then reads ONE contiguous run per slot
    // (whole cache lines, 16-byte loads), and since the MFMA only ever pairs A slot (hh, j) with B slot (hh, j) any such dealing is
    // valid as long as T follows it:  B_kb[(hh, j)][n] = tap(8 NKB hh + 8 kb + j - n - (R8 - r)).  Resident for the whole kernel.
    pfx_f16x8 B1[NKB], B2[NKB];
    {
        const _Float16* w1 = reinterpret_cast<const _Float16*>(wsplit) + GM_WOFF;
        const _Float16* w2 = w1 + GM_WLEN;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int t0 = 8 * NKB * hh + 8 * kb - i - (R8 - r); // in [-46, 127]
#pragma unroll
            for (int j = 0; j < 8; ++j) { B1[kb][j] = w1[t0 + j]; B2[kb][j] = w2[t0 + j]; }
        }
    }
    const int nrows = R + 2 * R8;
    const int nnb = R >> 5;

    // H-pass work of a wave: slots k = 0..3 of every tile, slot k = unit u = wave + 8k = 8 rows x 4 channels (A row m = c * 8 + row).
    // raw[k][kb] holds the 8 source samples of K block kb of slot k of the CURRENT tile; as soon as a block is converted to its
    // f16 fragment the same registers are refilled with the same block of the NEXT tile, so every global load has a whole tile of
    // work to land in — HBM latency is not exposed although only 2 waves share a SIMD.
    // No branches in this pass (the waitcnt bookkeeping stays exact): a slot past the tile's last unit recomputes the last unit
    // (identical values to identical LDS addresses), the refill past the last tile re-reads the last tile.
    uint32_t raw[4][NKB][2]; // [slot][K block]: slot k of the NEXT tile is requested once slot k of this tile is converted
    const int n_units = nrows >> 3; // H units of 8 rows x 4 channels (<= 28)
    struct hsrc { const uint8_t* line; int xs; };
    auto tile_x0 = [&](int tile) { const int ci = tile % tiles_x; return (ci < n_a ? col_a + ci : col_b + (ci - n_a)) * GM_COLS; };
    auto locate = [&](int tile, int k) -> hsrc {
        hsrc s;
        const int u = min(wave + 8 * k, n_units - 1), tl = min(tile, n_tiles - 1);
        const int x0 = tile_x0(tl), y0 = (tl / tiles_x) * R - y_phase;
        s.xs = x0 - R8 + 8 * NKB * hh;                                               // first sample of this lane's run
        const int ysrc = min(max(y0 - R8 + u * 8 + (i & 7), 0), h - 1);              // clamp-to-edge (filters.rs:296-298)
        s.line = planes + (size_t)(i >> 3) * plane_stride + (size_t)ysrc * w;       // A row m = i = channel * 8 + row
        return s;
    };
    auto fetch = [&](const hsrc& s, int k) { // the 8 NKB consecutive samples of this lane's run
        if constexpr (FAST) {
#pragma unroll
            for (int q = 0; q < NKB / 2; ++q) {
                const uint4 v = *reinterpret_cast<const uint4*>(s.line + s.xs + 16 * q);
                raw[k][2 * q][0] = v.x; raw[k][2 * q][1] = v.y; raw[k][2 * q + 1][0] = v.z; raw[k][2 * q + 1][1] = v.w;
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                uint32_t b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = s.line[min(max(s.xs + 8 * kb + j, 0), w - 1)]; // clamp-to-edge (filters.rs:268-270)
                raw[k][kb][0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                raw[k][kb][1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            }
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        fetch(locate(blockIdx.x, k), k);
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int x0 = tile_x0(tile), y0 = (tile / tiles_x) * R - y_phase; // tile rows start at multiples of R of the WHOLE image

        // ---- H pass ----
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dbg & 2) continue;
            const int u = min(wave + 8 * k, n_units - 1);
            // u8 -> f16 without arithmetic: the half with bit pattern 0x6400 | b is exactly 1024 + b, so one v_perm_b32 per two
            // samples builds the fragment; the constant 1024 * sum(T[.][n]) it adds to every output is the accumulator's start value
            pfx_f16x8 fr[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint32_t d0 = raw[k][kb][0], d1 = raw[k][kb][1];
                uint32_t q[4];
                q[0] = __builtin_amdgcn_perm(d0, 0x64646464u, 0x00050004u); // [0x64 b1 0x64 b0]
                q[1] = __builtin_amdgcn_perm(d0, 0x64646464u, 0x00070006u); // [0x64 b3 0x64 b2]
                q[2] = __builtin_amdgcn_perm(d1, 0x64646464u, 0x00050004u);
                q[3] = __builtin_amdgcn_perm(d1, 0x64646464u, 0x00070006u);
                fr[kb] = __builtin_bit_cast(pfx_f16x8, q);
            }
            __builtin_amdgcn_sched_barrier(0); // keep the refill HERE (the scheduler would sink it next to its use, one tile later)
            if (!(dbg & 1)) fetch(locate(tile + (int)gridDim.x, k), k);
            __builtin_amdgcn_sched_barrier(0);
            pfx_f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = -bias_c;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B1[kb], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B2[kb], acc, 0, 0, 0);
            }
            // D[m][x]: lane holds x = i; reg q -> m = (q & 3) + 8 (q >> 2) + 4 hh = c * 8 + row_local with c = q >> 2,
            // row_local = (q & 3) + 4 hh: regs 4c .. 4c+3 are four consecutive rows of channel c.  The value is S * h (S = the
            // weights' power-of-two scale, S * 255 < 65504); it is stored as hi + lo, hi = the top 11 significant bits.
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pfx_f16x4 h1, h2;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float va = acc[4 * g + e], vb = acc[4 * g + e + 1];
                    const pfx_f16x2 hi = pkrtz(va, vb);
                    const pfx_f16x2 lo = pkrtz(va - (float)hi[0], vb - (float)hi[1]);
                    h1[e] = hi[0]; h1[e + 1] = hi[1]; h2[e] = lo[0]; h2[e + 1] = lo[1];
                }
                *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(0 * 4 + g) * GM_PLANE + i * GM_YP + u * 8 + 4 * hh) = h1;
                *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(1 * 4 + g) * GM_PLANE + i * GM_YP + u * 8 + 4 * hh) = h2;
            }
        }
        __syncthreads();

        // ---- V pass: unit = (8-column block xb, 32-row output block nb); A row m = i = xl * 4 + c ----
        for (int u = wave; u < 4 * nnb && !(dbg & 4); u += 8) {
            const int xb = u & 3, nb = u >> 2;
            const int xl = i >> 2, c = i & 3;
            const _Float16* a1p = HR + (size_t)(0 * 4 + c) * GM_PLANE + (8 * xb + xl) * GM_YP + 32 * nb + 8 * NKB * hh;
            const _Float16* a2p = HR + (size_t)(1 * 4 + c) * GM_PLANE + (8 * xb + xl) * GM_YP + 32 * nb + 8 * NKB * hh;
            pfx_f32x16 accA, accB;
#pragma unroll
            for (int q = 0; q < 16; ++q) { accA[q] = 0.0f; accB[q] = 0.0f; }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const pfx_f16x8 a1 = *reinterpret_cast<const pfx_f16x8*>(a1p + 8 * kb);
                const pfx_f16x8 a2 = *reinterpret_cast<const pfx_f16x8*>(a2p + 8 * kb);
                accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B1[kb], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B2[kb], accB, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, B1[kb], accB, 0, 0, 0);
            }
            // D[m][n]: n = output row i of block nb; reg q -> m = (q & 3) + 8 (q >> 2) + 4 hh = xl' * 4 + c' with c' = q & 3,
            // xl' = 2 (q >> 2) + hh: regs 4g .. 4g+3 are the RGBA of pixel (8 xb + 2 g + hh, 32 nb + i)
            uint32_t* orow = OUT + (32 * nb + i) * GM_OUT_PITCH + 8 * xb + hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // `.round().clamp(0, 255) as u8` (filters.rs:308-311) of a non-negative sum: trunc(v + 0.5) with the final scale
                // fused in; v_cvt_u32_f32 saturates negatives (rounding noise around 0) to 0
                uint32_t px[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    px[e] = min((uint32_t)__builtin_fmaf(accA[4 * g + e] + accB[4 * g + e], inv_scale2, 0.5f), 255u);
                orow[2 * g] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
            }
        }
        __syncthreads();

        // ---- store: R rows x 32 pixels, 128 contiguous bytes per row ----
        for (int idx = tid; idx < R * GM_COLS && !(dbg & 8); idx += 512) {
            const int rr = idx >> 5, cc = idx & 31;
            if (y0 + rr >= 0 && y0 + rr < h && x0 + cc < w)
                reinterpret_cast<uint32_t*>(dst)[(size_t)(y0 + rr) * w + x0 + cc] = OUT[rr * GM_OUT_PITCH + cc];
        }
        // the next tile's H pass writes HR (not OUT); OUT is rewritten only after the next tile's first barrier
    }
}


// ---- the same two matrix passes as a column-strip walk: no recomputed rows -------------------------------------------------------
// gauss_mfma_kernel recomputes the horizontal pass for the 2 * R8 halo rows of every R-row tile (1.75x at sigma = 16).  Here a
// workgroup owns a 32-column strip and walks down it in steps of 32 rows with the horizontal results in an LDS RING of
// 16 NKB + 32 rows: waves 0-3 produce the 32 new rows of step i (one 8-row x 4-channel unit each) while waves 4-7 run the
// vertical pass of step i - NKB / 2 on the 16 NKB rows already in the ring (one 8-column block each) — disjoint ring slots, one
// barrier per step, both roles share a SIMD pairwise so one wave's conversions and packing overlap the other's MFMAs.  The
// previous step's output tile is written to global memory by the producer waves (128-byte row segments, double-buffered in LDS).
// A strip is cut into `n_seg` row segments so that a launch has about two workgroups per CU; a segment pays NKB / 2 - 1 producer
// steps of run-in.  Arithmetic per output is identical to gauss_mfma_kernel's (same fragments, same K-block grouping relative to
// the 32-row output block, blocks on the whole image's 32-row grid).
constexpr int GS_DEPTH = 3; // register sets of source samples in flight per producer lane
template <bool FAST, int NKB>
__global__ __launch_bounds__(512, 1) void gauss_strip_kernel(const uint8_t* __restrict__ planes, size_t plane_stride, uint8_t* __restrict__ dst,
                                                             const uint16_t* __restrict__ wsplit, int w, int h, int r, int R8, float inv_scale2,
                                                             float bias_c, int n_cols, int y_phase, int n_steps, int steps_per_seg, int dbg, unsigned long long* __restrict__ dbg_buf)
{
    constexpr int RING = 16 * NKB + 32, YP = RING + 8, PLANE = GM_COLS * YP + 32, HALF = NKB / 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_lds[];
    _Float16* HR = reinterpret_cast<_Float16*>(gm_lds);                            // [part][c][x][YP], rows = ring slots
    uint32_t* OUT = reinterpret_cast<uint32_t*>(gm_lds + (size_t)8 * PLANE * 2);    // [2][32][GM_OUT_PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const bool producer = wave < 4;

    pfx_f16x8 B1[NKB], B2[NKB]; // Toeplitz fragments, as in gauss_mfma_kernel
    {
        const _Float16* w1 = reinterpret_cast<const _Float16*>(wsplit) + GM_WOFF;
        const _Float16* w2 = w1 + GM_WLEN;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int t0 = 8 * NKB * hh + 8 * kb - i - (R8 - r);
#pragma unroll
            for (int j = 0; j < 8; ++j) { B1[kb][j] = w1[t0 + j]; B2[kb][j] = w2[t0 + j]; }
        }
        // settle the fragments here: otherwise the waitcnt bookkeeping treats them as "possibly still loading" at their first use in
        // every loop iteration and drains the wave's prefetches / stores there (vmcnt(3) .. vmcnt(0) in front of the MFMAs)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { asm volatile("" : "+v"(B1[kb])); asm volatile("" : "+v"(B2[kb])); }
    }
    const int ci = blockIdx.x % n_cols, seg = blockIdx.x / n_cols;
    const int x0 = ci * GM_COLS;
    const int t_first = seg * steps_per_seg, t_last = min(t_first + steps_per_seg, n_steps); // output blocks [t_first, t_last)
    if (t_first >= t_last) return;
    const int nst = t_last - t_first;
    const int a0 = 32 * t_first - y_phase - R8; // image row of ring slot 0 (producer step 0)

    // producer state: the 8 NKB samples of this lane's run for the next GS_DEPTH - 1 producer steps (a ring of register sets,
    // requested GS_DEPTH - 1 steps ahead: one step is ~1.5k cycles, HBM latency under load is several thousand)
    uint32_t raw[GS_DEPTH][NKB][2];
    const uint8_t* plane_c = planes + (size_t)(i >> 3) * plane_stride; // A row m = i = channel * 8 + row
    const int xs = x0 - R8 + 8 * NKB * hh;
    // FAST (rows 16-byte aligned, w % 16 == 0): the run is fetched as 16-byte pieces.  A piece is either wholly inside the row or
    // wholly outside it (clamp-to-edge, filters.rs:268-270: every sample of it is then the row's first / last byte), so border
    // strips cost one clamp of the piece address and a byte broadcast; interior strips (wave-uniform test) skip even that.
    const int n_hsteps = nst + HALF - 1; // producer steps 0 .. n_hsteps - 1; consumer step v needs producer steps v .. v + HALF - 1
    auto walk = [&](auto borderc) { // BORDER: the strip's source window leaves the image; chosen once per workgroup, so the loop
    constexpr bool BORDER = decltype(borderc)::value; // body below has no data-dependent branch around its loads (exact vmcnt waits)
    auto fetch = [&](auto bufc, int hs_req) {
        constexpr int BUF = decltype(bufc)::value;
        const int hs = min(hs_req, n_hsteps - 1);                                    // past the end: re-read the last step (unused)
        const int ysrc = min(max(a0 + 32 * hs + 8 * wave + (i & 7), 0), h - 1);      // clamp-to-edge (filters.rs:296-298)
        const uint8_t* line = plane_c + (size_t)ysrc * w;
        if constexpr (FAST) {
#pragma unroll
            for (int q = 0; q < NKB / 2; ++q) { // BORDER: the piece address is clamped here, its bytes are replaced when they are consumed
                const int xp = xs + 16 * q, xc = BORDER ? min(max(xp, 0), w - 16) : xp;
                const uint4 v = *reinterpret_cast<const uint4*>(line + xc);
                raw[BUF][2 * q][0] = v.x; raw[BUF][2 * q][1] = v.y; raw[BUF][2 * q + 1][0] = v.z; raw[BUF][2 * q + 1][1] = v.w;
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                uint32_t b[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = line[min(max(xs + 8 * kb + j, 0), w - 1)]; // clamp-to-edge (filters.rs:268-270)
                raw[BUF][kb][0] = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
                raw[BUF][kb][1] = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
            }
        }
    };

    auto stamp = [&](int it, int slot) { // development: s_memtime at phase boundaries of iterations 10..13, block 0, waves 0 and 4
        if ((dbg & 16) && blockIdx.x == 0 && (wave == 0 || wave == 4) && lane == 0 && it >= 10 && it < 14)
            dbg_buf[((wave >> 2) * 4 + (it - 10)) * 8 + slot] = __builtin_readcyclecounter();
    };
    // The two roles run separate loops with the same number of barriers.  The producer loop is unrolled GS_DEPTH times with the
    // refill of register set BUF unconditional and straight after its conversion: the loads then write their final registers and
    // every vmcnt wait is exact (a refill inside a conditional ends up as load-to-temporary + s_waitcnt vmcnt(0) + copy).
    const int last = nst + HALF, n_iter = ((last + GS_DEPTH) / GS_DEPTH) * GS_DEPTH; // iterations 0 .. n_iter - 1 (the surplus ones only synchronise)
    if (producer) {
        auto produce = [&](auto bufc, int it) {
            constexpr int BUF = decltype(bufc)::value;
            stamp(it, 0);
            pfx_f16x8 fr[NKB];
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                uint32_t d0 = raw[BUF][kb][0], d1 = raw[BUF][kb][1];
                if constexpr (BORDER && FAST) { // a 16-byte piece left / right of the row is the row's first / last byte throughout
                    const int xp = xs + 16 * (kb >> 1);
                    const uint32_t first = __builtin_amdgcn_perm(0u, raw[BUF][kb & ~1][0], 0x00000000u), lastb = __builtin_amdgcn_perm(0u, raw[BUF][kb | 1][1], 0x03030303u);
                    d0 = xp < 0 ? first : (xp >= w ? lastb : d0);
                    d1 = xp < 0 ? first : (xp >= w ? lastb : d1);
                }
                uint32_t q[4];
                q[0] = __builtin_amdgcn_perm(d0, 0x64646464u, 0x00050004u); // 0x6400 | byte = 1024 + byte as f16
                q[1] = __builtin_amdgcn_perm(d0, 0x64646464u, 0x00070006u);
                q[2] = __builtin_amdgcn_perm(d1, 0x64646464u, 0x00050004u);
                q[3] = __builtin_amdgcn_perm(d1, 0x64646464u, 0x00070006u);
                fr[kb] = __builtin_bit_cast(pfx_f16x8, q);
            }
            stamp(it, 1);
            __builtin_amdgcn_sched_barrier(0); // keep the refill HERE: the scheduler would sink it next to its use, steps later
            if (!(dbg & 8)) fetch(std::integral_constant<int, BUF>{}, it + GS_DEPTH);
            __builtin_amdgcn_sched_barrier(0);
            stamp(it, 2);
            if (it < n_hsteps && !(dbg & 2)) { // horizontal pass of 32 new rows into ring slots [32 it mod RING, +32)
                // two independent accumulator chains, summed in the epilogue
                pfx_f32x16 acc, acc2;
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[q] = -bias_c; acc2[q] = 0.0f; }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B1[kb], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B2[kb], acc2, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] += acc2[q];
                stamp(it, 3);
                const int ro = (32 * it) % RING + 8 * wave + 4 * hh; // regs 4c .. 4c+3 = four consecutive rows of channel c at column i
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pfx_f16x4 h1, h2;
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const float va = acc[4 * g + e], vb = acc[4 * g + e + 1];
                        const pfx_f16x2 hi = pkrtz(va, vb);
                        const pfx_f16x2 lo = pkrtz(va - (float)hi[0], vb - (float)hi[1]);
                        h1[e] = hi[0]; h1[e + 1] = hi[1]; h2[e] = lo[0]; h2[e + 1] = lo[1];
                    }
                    *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(0 * 4 + g) * PLANE + i * YP + ro) = h1;
                    *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(1 * 4 + g) * PLANE + i * YP + ro) = h2;
                }
            }
            // write the output block the consumers finished in the previous iteration (the barrier has passed): 128-byte row segments.
            // Measured per iteration: a consumer wave needs ~3900 cycles for reads + 24 MFMAs + rounding / packing, a producer wave
            // ~2200 for conversion + refill + 16 MFMAs + split / pack — the store belongs to the lighter role.
            {
                const int vp = it - 1 - HALF;
                if (vp >= 0 && vp < nst && !(dbg & 1)) {
                    const uint32_t* ob = OUT + (vp & 1) * 32 * GM_OUT_PITCH;
                    const int yb = 32 * (t_first + vp) - y_phase;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int idx = tid + 256 * q, rr = idx >> 5, cc = idx & 31;
                        if (yb + rr >= 0 && yb + rr < h && x0 + cc < w)
                            reinterpret_cast<uint32_t*>(dst)[(size_t)(yb + rr) * w + x0 + cc] = ob[rr * GM_OUT_PITCH + cc];
                    }
                }
            }
            stamp(it, 4);
            __syncthreads();
            stamp(it, 5);
        };
        fetch(std::integral_constant<int, 0>{}, 0);
        if constexpr (GS_DEPTH > 1) fetch(std::integral_constant<int, 1>{}, 1);
        if constexpr (GS_DEPTH > 2) fetch(std::integral_constant<int, 2>{}, 2);
        if constexpr (GS_DEPTH > 3) fetch(std::integral_constant<int, 3>{}, 3);
        for (int it = 0; it < n_iter; it += GS_DEPTH) {
            produce(std::integral_constant<int, 0>{}, it);
            if constexpr (GS_DEPTH > 1) produce(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (GS_DEPTH > 2) produce(std::integral_constant<int, 2>{}, it + 2);
            if constexpr (GS_DEPTH > 3) produce(std::integral_constant<int, 3>{}, it + 3);
        }
    } else {
        for (int it = 0; it < n_iter; ++it) {
            stamp(it, 0);
            // vertical pass of output block v on ring slots [32 v mod RING, + 16 NKB); A row m = i = xl * 4 + c
            const int v = it - HALF;
            if (v >= 0 && v < nst && !(dbg & 4)) {
                const int xb = wave - 4, xl = i >> 2, c = i & 3;
                const _Float16* a1p = HR + (size_t)(0 * 4 + c) * PLANE + (8 * xb + xl) * YP;
                const _Float16* a2p = HR + (size_t)(1 * 4 + c) * PLANE + (8 * xb + xl) * YP;
                const int rb = (32 * v) % RING + 8 * NKB * hh;
                stamp(it, 1);
                pfx_f32x16 accA, accB, accC; // three independent chains
#pragma unroll
                for (int q = 0; q < 16; ++q) { accA[q] = 0.0f; accB[q] = 0.0f; accC[q] = 0.0f; }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    int ro = rb + 8 * kb;
                    ro = ro >= RING ? ro - RING : ro;
                    const pfx_f16x8 a1 = *reinterpret_cast<const pfx_f16x8*>(a1p + ro);
                    const pfx_f16x8 a2 = *reinterpret_cast<const pfx_f16x8*>(a2p + ro);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B1[kb], accA, 0, 0, 0);
                    accB = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B2[kb], accB, 0, 0, 0);
                    accC = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, B1[kb], accC, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) accB[q] += accC[q];
                stamp(it, 3);
                uint32_t* orow = OUT + (v & 1) * 32 * GM_OUT_PITCH + i * GM_OUT_PITCH + 8 * xb + hh; // regs 4g..4g+3 = RGBA of (8 xb + 2 g + hh, row i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t px[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        px[e] = min((uint32_t)__builtin_fmaf(accA[4 * g + e] + accB[4 * g + e], inv_scale2, 0.5f), 255u); // filters.rs:308-311
                    orow[2 * g] = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
                }
            }
            stamp(it, 4);
            __syncthreads();
            stamp(it, 5);
        }
    }
    }; // walk
    const bool interior = __builtin_amdgcn_readfirstlane((x0 - R8 >= 0 && x0 - R8 + 16 * NKB <= w) ? 1 : 0) != 0;
    if (interior) walk(std::false_type{}); else walk(std::true_type{});
}

} // namespace

// LDS bounds: H tile (1024 + 2r + 12) x 20 B and the narrowest V tile (256 + 2r + 4) rows x 5 x 16 B <= 160 KiB
extern "C" int pfxk_gauss_max_radius(void) { return 850; }
extern "C" void pfxk_gauss_set_v_config(int cfg) { g_v_cfg = cfg; }
extern "C" void pfxk_gauss_set_dbg_buf(unsigned long long* p) { g_dbg_buf = p; }
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

// fused matrix-core Gaussian (default mode): d_wsplit = 2 x GM_WLEN f16 (pfxk_gauss_mfma_weights layout)
extern "C" int pfxk_gauss_mfma_max_radius(void) { return GM_MAXR; }
extern "C" int pfxk_gauss_mfma_wlen(void) { return GM_WLEN; }
extern "C" int pfxk_gauss_mfma_woff(void) { return GM_WOFF; }
extern "C" size_t pfxk_gauss_mfma_scratch_bytes(uint32_t w, uint32_t h) { return 4 * ((((size_t)w * h) + 255) & ~(size_t)255); }
extern "C" hipError_t pfxk_gauss_mfma(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, uint8_t* d_planes, const uint16_t* d_wsplit,
                                      int radius, float inv_scale2, float bias_c, uint32_t w, uint32_t h, uint32_t first_row, int n_cus)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius < 1 || radius > GM_MAXR) return hipErrorInvalidValue;
    const int R8 = (radius + 15) & ~15;                    // window start x0 - R8: 16-byte aligned runs
    const int nkb = (GM_COLS + R8 + radius + 15) / 16;      // 4, 6 or 8
    int R = (GM_ROWS - 2 * R8) & ~31;
    if (R > GM_MAX_R) R = GM_MAX_R;
    // `first_row` = index of the buffer's row 0 in the whole image when the buffer is a band of it: tiles are laid out on the whole
    // image's grid, so every output sees the same K-block grouping (the same f32 summation order) as in a whole-image call
    const int y_phase = (int)(first_row % (uint32_t)R);
    const int tiles_x = ((int)w + GM_COLS - 1) / GM_COLS, tiles_y = ((int)h + y_phase + R - 1) / R;
    // interior tile columns: source window [x0 - R8, x0 - R8 + 16 nkb) inside the image, rows 16-byte aligned
    const size_t n_px = (size_t)w * h, plane_stride = (n_px + 255) & ~(size_t)255;
    {
        size_t blocks = (n_px / 4 + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        if (blocks == 0) blocks = 1;
        gauss_planarize_kernel<<<(uint32_t)blocks, 256, 0, stream>>>(reinterpret_cast<const uint32_t*>(d_src), d_planes, n_px, plane_stride);
    }
    const bool aligned = ((uintptr_t)d_planes & 15u) == 0 && (w & 15u) == 0; // 16-byte fragment loads
    int c_lo = (R8 + GM_COLS - 1) / GM_COLS;                             // first column with x0 - R8 >= 0
    int c_hi = ((int)w - 16 * nkb + R8) / GM_COLS + 1;                   // one past the last column with x0 - R8 + 16 nkb <= w
    if (!aligned || (int)w - 16 * nkb + R8 < 0 || c_hi <= c_lo) { c_lo = 0; c_hi = 0; }
    if (c_hi > tiles_x) c_hi = tiles_x;
    if (!(g_v_cfg & 0x100)) { // shipped: column-strip walk (tune gauss_v_cfg = 256 selects the tile kernel for A/B)
        const int y_ph = (int)(first_row % 32u);
        const int n_steps = ((int)h + y_ph + 31) / 32;
        const bool fast = ((uintptr_t)d_planes & 15u) == 0 && (w & 15u) == 0 && w >= 16;
        hipError_t errs = hipSuccess;
        auto launch_s = [&](auto nkb_c) {
            constexpr int NK = decltype(nkb_c)::value;
            constexpr int RING = 16 * NK + 32;
            const size_t lds = (size_t)8 * (GM_COLS * (RING + 8) + 32) * 2 + (size_t)2 * 32 * GM_OUT_PITCH * 4;
            // cut every strip into n_seg row segments so that the launch has just under two workgroups per CU (one is resident per CU,
            // LDS-bound; measured at 8K: 1 / 2 / 3 segments per strip = 0.359 / 0.340 / 0.341 ms); a segment pays NK/2 - 1 run-in steps
            int n_seg = (int)((18L * n_cus / 10 + tiles_x - 1) / tiles_x);
            if (g_v_cfg & 0xff) n_seg = g_v_cfg & 0xff; // tuning override
            if (n_seg < 1) n_seg = 1;
            int per = (n_steps + n_seg - 1) / n_seg;
            if (per < 4) per = n_steps < 4 ? n_steps : 4;
            n_seg = (n_steps + per - 1) / per;
            const int grid = tiles_x * n_seg;
            if (fast) {
                errs = hipFuncSetAttribute((const void*)gauss_strip_kernel<true, NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (errs) return;
                gauss_strip_kernel<true, NK><<<grid, 512, lds, stream>>>(d_planes, plane_stride, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles_x, y_ph, n_steps, per, g_v_cfg >> 9, g_dbg_buf);
            } else {
                errs = hipFuncSetAttribute((const void*)gauss_strip_kernel<false, NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (errs) return;
                gauss_strip_kernel<false, NK><<<grid, 512, lds, stream>>>(d_planes, plane_stride, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles_x, y_ph, n_steps, per, g_v_cfg >> 9, g_dbg_buf);
            }
        };
        switch (nkb) {
        case 4: launch_s(std::integral_constant<int, 4>{}); break;
        case 6: launch_s(std::integral_constant<int, 6>{}); break;
        default: launch_s(std::integral_constant<int, 8>{}); break;
        }
        if (errs) return errs;
        return hipGetLastError();
    }
    const int n_int = c_hi - c_lo, n_brd = tiles_x - n_int;
    hipError_t err = hipSuccess;
    auto launch = [&](auto fast, auto nkb_c) {
        constexpr bool F = decltype(fast)::value;
        constexpr int NK = decltype(nkb_c)::value;
        const int cols = F ? n_int : n_brd;
        if (cols <= 0) return;
        err = hipFuncSetAttribute((const void*)gauss_mfma_kernel<F, NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GM_LDS);
        if (err) return;
        const int n_tiles = cols * tiles_y, grid = n_tiles < n_cus ? n_tiles : n_cus; // persistent: one workgroup per CU (LDS-bound)
        if (F) gauss_mfma_kernel<F, NK><<<grid, 512, GM_LDS, stream>>>(d_planes, plane_stride, d_dst, d_wsplit, (int)w, (int)h, radius, R8, R, inv_scale2, bias_c, cols, n_tiles, c_lo, cols, 0, y_phase, g_v_cfg >> 9);
        else   gauss_mfma_kernel<F, NK><<<grid, 512, GM_LDS, stream>>>(d_planes, plane_stride, d_dst, d_wsplit, (int)w, (int)h, radius, R8, R, inv_scale2, bias_c, cols, n_tiles, 0, c_lo, c_hi, y_phase, g_v_cfg >> 9);
    };
    auto both = [&](auto nkb_c) {
        launch(std::true_type{}, nkb_c);
        if (!err) launch(std::false_type{}, nkb_c);
    };
    switch (nkb) { // = R8 / 8 + 2
    case 4: both(std::integral_constant<int, 4>{}); break;
    case 6: both(std::integral_constant<int, 6>{}); break;
    default: both(std::integral_constant<int, 8>{}); break;
    }
    if (err) return err;
    return hipGetLastError();
}
