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
#include <atomic>
#include <algorithm>
#include "pfx_kernels.h"
#include "k_pointwise.h"

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
// ds_read_b128 groups hit 16 distinct 16-byte slots; found by an exhaustive stride search in round 1).  The halo (2r rows) dominates
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
                pack_round_rgba(a.x, a.y, a.z, a.w); // filters.rs:308-311
        }
    }
}

template <bool EXACT>
hipError_t launch_h(hipStream_t stream, const uint8_t* d_src, float4* tmp, const float* wts, int radius, uint32_t w, uint32_t h)
{
    const int h_entries = H_TILE + 2 * radius + H_SLACK;
    const size_t lds_h = (size_t)(h_entries + (h_entries >> 2) + 1) * sizeof(float4);
    hipError_t e = grant_lds_for((const void*)gauss_h_kernel<EXACT>, lds_h);
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
    hipError_t e = grant_lds_for((const void*)gauss_v_kernel<EXACT, TX, YG, RS>, lds_v);
    if (e) return e;
    dim3 gv((w + TX - 1) / TX, (h + ROWS - 1) / ROWS);
    gauss_v_kernel<EXACT, TX, YG, RS><<<gv, TX * YG, lds_v, stream>>>(tmp, d_dst, wts, radius, (int)w, (int)h);
    return hipGetLastError();
}

unsigned long long* g_dbg_buf = nullptr; // development: phase timestamps of gauss_strip_kernel (pfxk_gauss_set_dbg_buf)
int g_v_cfg = 0; // tuning knob (pfxk_gauss_set_v_config); 0 is the shipped configuration
int g_mfma_seg = 0; // tuning knob (pfxk_gauss_set_mfma_segments): row segments per strip of the matrix-core kernel, 0 = automatic
// f16 pieces per weight / per horizontal result in the matrix-core kernel (pfxk_gauss_set_mfma_parts).  Shipped: ONE piece per weight, two per
// horizontal result (round 4, tools/gauss_parts.py at 8K against the bit-exact mode): sigma 16 0.177 -> 0.138 ms, sigma 4 0.119 -> 0.090, sigma 24
// 0.243 -> 0.190; channels that differ (all by 1): uniform noise 4.9e-5 -> 3.6e-4, photograph-like ramps + noise 5.0e-5 -> 4.8e-5, smooth ramps
// 1e-7 -> 8e-9.  One piece for both (pfx_tune "gauss_parts" = 11) runs sigma 16 in 0.114 ms but rounds the horizontal result to 11 bits — 0.125 LSB
// steps at the bright end: 2 % of a smooth ramp's channels come out one off; kept as a measured variant, not shipped.
int g_mfma_cols64 = 6;   // 6 and 8 K blocks (sigma 5.4 .. 16) ship on it: 8K sigma 8 0.094 -> 0.090, sigma 16 0.122 -> 0.111; 4 K blocks measure equal (0.075) and stay on two 32-column workgroups per CU.  pfxk_gauss_set_mfma_cols64 (pfx_tune "gauss_cols64"): bit 0 / 1 / 2 = launches with 4 / 6 / 8 K blocks on the 64-column kernel (else the 32-column one); identical results
std::atomic<int> g_mfma_wp{1}, g_mfma_hp{2};   // process-wide development knobs read by batch workers' threads

template <bool EXACT>
hipError_t launch_v(hipStream_t stream, const float4* tmp, uint8_t* d_dst, const float* wts, int radius, uint32_t w, uint32_t h)
{
    int cfg = g_v_cfg & 0xff;
    // The tile's halo (2r rows) is reloaded by every tile: beyond the matrix-core kernel's radii (sigma 16 .. 100 in the reference's advanced
    // dialog) taller tiles pay — measured at 8K: sigma 50 1.06 -> 0.92 ms with 16 x 256, sigma 64 2.2 -> 1.1 with 8 x 512, sigma 100 3.3 -> 2.0 with 8 x 256
    if (cfg == 0 && radius > 110) cfg = radius <= 160 ? 1 : (radius <= 240 ? 4 : 3); // 16 x 256, 8 x 512, 8 x 256 (what still fits 160 KB of LDS)
    if (radius > 340) cfg = 0;
    if (radius > 380) cfg = 2; // narrowest tile for huge radii (LDS bound)
    switch (cfg) {
    case 1: return launch_v_cfg<EXACT, 16, 64, 16>(stream, tmp, d_dst, wts, radius, w, h);
    case 2: return launch_v_cfg<EXACT, 4, 64, 5>(stream, tmp, d_dst, wts, radius, w, h);
    case 3: return launch_v_cfg<EXACT, 8, 64, 10>(stream, tmp, d_dst, wts, radius, w, h);
    case 4: return launch_v_cfg<EXACT, 8, 128, 10>(stream, tmp, d_dst, wts, radius, w, h);
    // shipped: 8 columns x 128 rows, 256 threads.  Measured at 8K, sigma=16 (profiles/r01_tuning.md): 0.344 ms vs
    // 0.368 (8x256 rows), 0.350 (4x256), 0.446 (16x256)
    default: return launch_v_cfg<EXACT, 8, 32, 10>(stream, tmp, d_dst, wts, radius, w, h);
    }
}


// ---- fused H+V Gaussian on the matrix cores (default mode, radius <= 48) -------------------------------------------------------
// Both passes are banded-Toeplitz products  D[m][n] = sum_k A[m][k] * T[k][n],  T[k][n] = w[k - n + r]  — a genuine contraction,
// so they run on v_mfma_f32_32x32x16_f16 (16x the f32 FMA rate) while the VALU only converts and packs:
//   * operands are split so that every product is exact in the f32 accumulator: a u8 sample is exact in f16; a weight is
//     scaled by S = 256 and split w*S = w1 + w2 (two f16, 22 significant bits); the f32 horizontal result h (kept scaled, S*h <
//     65504) is split S*h = h1 + h2 the same way.  H pass: p*w1 + p*w2.  V pass: h1*w1 + h1*w2 + h2*w1; the dropped h2*w2 is < 2^-22
//     of the sum.  The result differs from the CPU path's f32 mul/add chain by rounding noise only (+-1 LSB class, tests and
//     bench.py assert it; at 8K ~1e-4 of the channels differ).
//     WP = 1 (shipped since round 4): the weight is ONE f16, w * S rounded to 11 bits and the table nudged so that its sum is that of the exact
//     weights (pfx_host_gaussian_split_f16): H pass p*w1', V pass h1*w1' + h2*w1' — 3 MFMAs per K block and output tile instead of 5.  The filter
//     applied is a Gaussian whose taps are off by <= 2^-12 relative with zero sum: a flat image blurs to itself, an adversarial one moves by
//     at most sum|delta| * 255 / S < 0.1 LSB per pass, so the +-1 LSB bound holds by construction (rates: the g_mfma_wp comment above).
//   * u8 -> f16 without arithmetic: the half with bit pattern 0x6400 | b is exactly 1024 + b, so one v_perm_b32 per two samples
//     builds a fragment; the constant 1024 * sum(T[.][n]) it adds to every output is the H accumulator's start value.
//   * the 16 NKB samples of a window are dealt to the MFMA's K slots as runs of 8: slot (lane half hh, K block kb) holds samples
//     [16 kb + 8 hh, +8).  The MFMA only ever pairs A slot (hh, j) with B slot (hh, j), so any such dealing is valid as long as T
//     follows it:  B_kb[(hh, j)][n] = tap(16 kb + 8 hh + j - n - (R8 - r)).  This one makes K block kb cover window rows
//     [16 kb, 16 kb + 16): the 32 newest rows of a vertical window are its last two K blocks (used below).  The kernel relies on
//     nothing but the documented C/D map (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)).
//
// Column-strip walk: a workgroup owns a 32-column strip and walks down it in steps of 32 rows with the horizontal results in an
// LDS ring of two 32-row steps (f16 pairs, transposed so that a column's rows are contiguous; rounds 2-3 kept the whole 16 NKB-row window there).  Waves 0-3 ("producers") compute the
// 32 new rows of a step (8 rows x 4 channels each, NKB MFMAs with one-piece weights); waves 4-7 ("consumers") run the vertical pass of a
// 32-row output block on its 16 NKB-row window (8 columns x 4 channels each, 2 NKB MFMAs) — the window's fragments stay in the consumers'
// registers from block to block, only the 32 newest rows come from the ring; disjoint ring halves, one barrier per iteration, each SIMD hosts
// one wave of either role per workgroup, and up to 8 K blocks two workgroups share a CU.  No row is computed twice, the f32 intermediate never touches HBM: 8 algorithmic
// bytes per pixel are the kernel's only HBM traffic (+ the x-halo re-reads, served by L2).
//   * the two waves of a SIMD share its matrix pipe (32 cycles per MFMA, 40 MFMAs per iteration) and a wave issues a VALU
//     instruction only every ~6 cycles, so an iteration is arranged in ANTI-PHASE (s_memtime timeline: tools/gauss_timeline.py):
//     first the consumers multiply — the fragments of all but the last two K blocks are in registers already, the
//     last two (the rows finished before the barrier) arrive under those MFMAs — while the producers split and store the rows
//     whose MFMAs they issued BEFORE the barrier, de-interleave the next step and build its fragments; then the producers issue
//     their MFMAs (and go to the barrier without waiting for them) while the consumers round and pack.  Step s is multiplied in
//     iteration s, reaches the ring in s + 1; block v runs in iteration v + NKB / 2 + 1 and leaves in the one after.
//   * a producer lane fetches 2 NKB consecutive RGBA pixels of one row (16-byte loads, whole cache lines across the wave), requested
//     GS_DEPTH steps ahead into a ring of register sets (the refill is unconditional and straight after the set's last use: loads
//     write their final registers, every vmcnt wait is exact).  Before the MFMAs the wave de-interleaves the four channels through a
//     private 4.6 KB LDS patch (v_perm gathers, ds_write_b128 / ds_read_b64, no barrier: one wave, in-order LDS) so that A row
//     m = channel * 8 + row holds runs of 8 consecutive samples of ONE channel.
//   * the previous output block leaves through LDS as 128-byte row segments, 16 bytes per consumer lane, between the consumers'
//     MFMA groups;
//   * a strip is cut into row segments so that the launch has just under two workgroups per CU; a segment pays NKB / 2
//     iterations of run-in; output blocks lie on the whole image's 32-row grid (y_phase), so a band of a sharded document gets
//     the same K-block grouping — the same f32 summation order — as the whole image.
typedef _Float16 pfx_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pfx_f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pfx_f16x2 __attribute__((ext_vector_type(2)));
typedef float pfx_f32x16 __attribute__((ext_vector_type(16)));
// v_cvt_pkrtz_f16_f32: two f32 -> packed f16, round toward zero
PFX_DEV pfx_f16x2 pkrtz(float a, float b) { return __builtin_bit_cast(pfx_f16x2, __builtin_amdgcn_cvt_pkrtz(a, b)); }

constexpr int GM_COLS = 32;             // output columns per strip (one MFMA N block)
constexpr int GM_MAXR = 80;             // largest radius (sigma <= 26.6: 12 K blocks; the ring + patches then fill 155 of the 160 KB of LDS)
constexpr int GM_OUT_PITCH = 36;        // dwords per staged output row (16-byte aligned rows: the block leaves as ds_read_b128)
constexpr int GM_WOFF = 48;             // wsplit[GM_WOFF + t] = tap t; zeros elsewhere
constexpr int GM_WLEN = 256;            // entries per weight part
#ifndef PFX_GAUSS_TWO_WG8
#define PFX_GAUSS_TWO_WG8 1   // development A/B: 0 = the 8-K-block build keeps B1 in registers and 3 source steps in flight (148 registers: one workgroup per CU)
#endif
constexpr int GS_DEPTH = 3;             // register sets of source pixels in flight per producer lane (2 in the 8-K-block build, whose register budget is 128: see gauss_strip_kernel)
constexpr int gs_xrow(int nkb) { return 16 * (nkb > 8 ? nkb : 8) + 16; } // bytes per (channel, row) line of the de-interleave patch: 16 NKB samples + 16 (bank spread)

inline size_t gauss_strip_lds_bytes(int nkb, int hp)
{
    const int ring = 64; // two 32-row steps (the vertical window itself lives in the consumers' registers)
    return (size_t)4 * hp * (GM_COLS * (ring + 8) + 32) * 2 + (size_t)2 * 32 * GM_OUT_PITCH * 4 + (size_t)4 * 4 * 8 * gs_xrow(nkb) + (size_t)nkb * 64 * 16; // the last term (B1 fragments) is used by the 8-K-block one-piece build only
}

// DBG: the development instantiation (switchable parts, s_memtime stamps); the shipped one has none of those branches — a dozen
// skipped-over stamps per iteration were 10 % of the kernel
// CHAIN (round 6, pfx_chain_dev): pfxk_chain = the consumers put every blurred pixel through a chain of table-free pointwise ops between rounding and staging
// (k_pointwise.h: chain_apply on the u8 pixel — what the ops would read from a buffer): `Gaussian -> HSL` is one launch, the blurred image never reaches memory.
struct gs_no_chain {};
// workgroups per CU the build is compiled for (x 2 = waves per SIMD): two up to 8 K blocks with one-piece weights
template <class CHAIN> constexpr int gs_wg_per_cu(int nkb, int wp) { return (nkb <= (PFX_GAUSS_TWO_WG8 ? 8 : 6) && wp == 1) ? 2 : 1; }
template <bool FAST, int NKB, bool DBG, int WP = 2, int HP = 2, class CHAIN = gs_no_chain>
__global__ __launch_bounds__(512, 2 * gs_wg_per_cu<CHAIN>(NKB, WP)) void gauss_strip_kernel(const CHAIN chain_arg /* first: read through the kernarg segment (k_pointwise.h) */,
                                                             const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const uint16_t* __restrict__ wsplit, int w, int h, int r, int R8, float inv_scale2,
                                                             float bias_c, int n_cols, int y_phase, int n_steps, int steps_per_seg, int dbg_arg,
                                                             unsigned long long* __restrict__ dbg_buf)
{
    const pw::chain_kptr chain = pw::chain_in_kernarg();   // meaningful in the CHAIN builds only
    constexpr int GS_XROW = gs_xrow(NKB);
    // ring = two steps of 32 rows: the one the consumers load this iteration (finished before the last barrier) and the one the producers are storing
    constexpr int RING = 64, YP = RING + 8, PLANE = GM_COLS * YP + 32, HALF = NKB / 2, PPL = 2 * NKB; // PPL: pixels per producer lane
    // The 8-K-block build (sigma 10.7 .. 16) is the one that needs help to fit two workgroups per CU (128 registers per wave): its B1 fragments live in LDS and
    // it keeps 2 source steps in flight instead of 3.  4 and 6 K blocks fit as they are; 10 and 12 (96+ registers of window fragments) stay at one workgroup.
    constexpr bool B_LDS = PFX_GAUSS_TWO_WG8 && NKB == 8 && WP == 1;
    constexpr int GSD = B_LDS ? 2 : GS_DEPTH;
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_lds[];
    _Float16* HR = reinterpret_cast<_Float16*>(gm_lds);                            // [part][c][x][YP], rows = ring slots
    uint32_t* OUT = reinterpret_cast<uint32_t*>(gm_lds + (size_t)4 * HP * PLANE * 2); // [2][32][GM_OUT_PITCH]
    uint8_t* XP = gm_lds + (size_t)4 * HP * PLANE * 2 + (size_t)2 * 32 * GM_OUT_PITCH * 4; // [producer wave][c][row][GS_XROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const bool producer = wave < 4;

    // Toeplitz fragments.  B1 (the one-piece weights, or the high piece) lives in LDS — 1 KB per K block, the same for every wave (a lane's fragment depends on
    // its position in the wave alone) — and is read in front of the MFMA that multiplies with it: 4 NKB registers per wave freed, which with the two-step ring
    // is what lets two workgroups share a CU.  B2 (two-piece weights only: pfx_tune "gauss_parts" = 22) stays in registers.
    pfx_f16x8* const BL = reinterpret_cast<pfx_f16x8*>(gm_lds + (size_t)4 * HP * PLANE * 2 + (size_t)2 * 32 * GM_OUT_PITCH * 4 + (size_t)4 * 4 * 8 * GS_XROW);
    pfx_f16x8 B1r[B_LDS ? 1 : NKB], B2[NKB];
    {
        // WP == 1: the third table, every weight ONE f16 (pfx_host_gaussian_split_f16); B2 is then never used
        const _Float16* w1 = reinterpret_cast<const _Float16*>(wsplit) + GM_WOFF + (WP == 1 ? 2 * GM_WLEN : 0);
        const _Float16* w2 = reinterpret_cast<const _Float16*>(wsplit) + GM_WOFF + GM_WLEN;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int t0 = 16 * kb + 8 * hh - i - (R8 - r); // in [-46, 16 NKB - 8]
            pfx_f16x8 b1;
#pragma unroll
            for (int j = 0; j < 8; ++j) { b1[j] = w1[t0 + j]; B2[kb][j] = w2[t0 + j]; }
            if constexpr (B_LDS) { if (wave == 0) BL[kb * 64 + lane] = b1; }
            else B1r[kb] = b1;
        }
        // settle the fragments here: otherwise the waitcnt bookkeeping treats them as "possibly still loading" at their first use in
        // every loop iteration and drains the wave's prefetches / stores there (vmcnt(3) .. vmcnt(0) in front of the MFMAs)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { asm volatile("" : "+v"(B2[kb])); if constexpr (!B_LDS) asm volatile("" : "+v"(B1r[kb])); }
        if constexpr (B_LDS) __syncthreads();
    }
    auto B1 = [&](int kb) -> pfx_f16x8 { if constexpr (B_LDS) return BL[kb * 64 + lane]; else return B1r[kb]; };
    // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): the swizzle hands every XCD a run of NEIGHBOURING strips, whose
    // source windows overlap by 3/4 (a 32-column strip reads 16 NKB columns), so the overlap is fetched into one L2 instead of up to 8
    const int bid = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int ci = bid % n_cols, seg = bid / n_cols;
    const int x0 = ci * GM_COLS;
    const int t_first = seg * steps_per_seg, t_last = min(t_first + steps_per_seg, n_steps); // output blocks [t_first, t_last)
    if (t_first >= t_last) return;
    const int nst = t_last - t_first;
    const int a0 = 32 * t_first - y_phase - R8; // image row of ring slot 0 (producer step 0)
    const int n_hsteps = nst + HALF - 1;        // producer steps 0 .. n_hsteps - 1; consumer step v needs producer steps v .. v + HALF - 1
    // a step's MFMAs are issued in iteration `step`, its rows reach the ring in step + 1; block v runs in iteration v + HALF + 1 and leaves in
    // v + HALF + 2
    const int last = nst + HALF + 1, n_iter = ((last + GSD) / GSD) * GSD; // iterations 0 .. n_iter - 1 (surplus ones only synchronise)

    // development: s_memtime at phase boundaries of iterations 10..13, one interior block, waves 0 and 4.  The reads are not waited for
    // where they are issued (that would drain the wave's LDS queue and distort the phase); flush_stamps() waits once per iteration.
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp0 = 0, tp1 = 0;
    const int dbg = DBG ? dbg_arg : 0; // a compile-time zero in the shipped instantiation: every `dbg &` test folds away
    const bool stamping = DBG && (dbg & 48) && bid == 8 && (wave == 0 || wave == 4);
    auto stamp = [&](int it, int slot) {
        if constexpr (!DBG) return;
        if (dbg & 32) { // period probe: iteration starts 10 and 40 only, one flush at the end
            if (stamping && slot == 0 && it == 10) asm volatile("s_memtime %0" : "=s"(tp0));
            if (stamping && slot == 0 && it == 40) asm volatile("s_memtime %0" : "=s"(tp1));
            return;
        }
        if (stamping && it >= 10 && it < 14) asm volatile("s_memtime %0" : "=s"(ts[slot]));
    };
    auto flush_stamps = [&](int it) {
        if constexpr (!DBG) return;
        if (dbg & 32) {
            if (stamping && it == 40) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) { dbg_buf[(wave >> 2) * 32] = tp0; dbg_buf[(wave >> 2) * 32 + 1] = tp1; }
            }
            return;
        }
        if (stamping && it >= 10 && it < 14) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) dbg_buf[((wave >> 2) * 4 + (it - 10)) * 8 + k] = ts[k];
            }
        }
    };

    // CHAIN builds: the consumers put their four pixels of a block through the chain between rounding and staging.  (Spreading the pointwise arithmetic over all
    // eight waves at store time was built too: the 24-way op dispatch then has ten inline sites — hipcc leaves it out of line, and a call inside this loop costs a
    // closure in scratch whose reloads share vmcnt with the source prefetch: 0.78 ms against 0.22, profiles/r06_tuning.md.)
    constexpr bool CH = !std::is_same<CHAIN, gs_no_chain>::value;
    if (producer) {
        // BORDER: the strip's source window leaves the image; chosen once per workgroup so that the loop has no data-dependent
        // branch around its loads.  A 4-pixel piece is wholly inside or wholly outside the row (w % 4 == 0 in the FAST
        // instantiation): outside pieces are fetched from the clamped address and replaced by the edge pixel when consumed.
        auto walk = [&](auto borderc) {
            constexpr bool BORDER = decltype(borderc)::value;
            // fetch role of a lane: row fr = lane >> 3 of the wave's 8 rows, pixels [fs * PPL, (fs + 1) * PPL) of the 16 NKB window
            const int frow = lane >> 3, fs = lane & 7;
            const int fx = x0 - R8 + fs * PPL;
            uint32_t raw[GSD][PPL];
            auto fetch = [&](auto bufc, int hs_req) {
                constexpr int BUF = decltype(bufc)::value;
                const int hs = min(hs_req, n_hsteps - 1);                                   // past the end: re-read the last step (unused)
                const int ysrc = min(max(a0 + 32 * hs + 8 * wave + frow, 0), h - 1);        // clamp-to-edge (filters.rs:296-298)
                const uint32_t* line = reinterpret_cast<const uint32_t*>(src) + (size_t)ysrc * w;
                if constexpr (FAST) {
#pragma unroll
                    for (int q = 0; q < PPL / 4; ++q) {
                        const int xp = fx + 4 * q, xc = BORDER ? min(max(xp, 0), w - 4) : xp;
                        const uint4 v = *reinterpret_cast<const uint4*>(line + xc);
                        raw[BUF][4 * q] = v.x; raw[BUF][4 * q + 1] = v.y; raw[BUF][4 * q + 2] = v.z; raw[BUF][4 * q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < PPL; ++q) raw[BUF][q] = line[min(max(fx + q, 0), w - 1)]; // clamp-to-edge (filters.rs:268-270)
                }
            };
            uint8_t* xp_w = XP + (size_t)wave * 4 * 8 * GS_XROW;
            pfx_f32x16 acc, acc2; // two independent accumulator chains of one step; they live across the barrier (summed in the epilogue)
            pfx_f32x16 neg_bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) neg_bias[q] = -bias_c;
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[q] = 0.0f; acc2[q] = 0.0f; }
            // One producer iteration, arranged so that its VALU phases face the consumers' MFMA phase and vice versa: (0) split and
            // store the 32 rows whose MFMAs were issued before the last barrier (step it - 1) — the consumers are multiplying; (1) - (2)
            // de-interleave step it, refill that register set, build the fragments; (3) issue step it's MFMAs and go to the barrier
            // without waiting for them — the consumers are rounding and packing by then.
            auto produce = [&](auto bufc, int it) {
                constexpr int BUF = decltype(bufc)::value;
                stamp(it, 0);
                if (it >= 1 && it - 1 < n_hsteps && !(dbg & 2)) {
                    // D[m][x]: lane holds column x = i; reg q -> m = (q & 3) + 8 (q >> 2) + 4 hh = c * 8 + row with c = q >> 2: regs 4c .. 4c+3
                    // are four consecutive rows of channel c.  The value S * h goes to ring slots [32 (it - 1) mod RING, +32) as hi + lo,
                    // hi = its top 11 significant bits.
                    const int ro = (32 * (it - 1)) % RING + 8 * wave + 4 * hh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        pfx_f16x4 h1, h2;
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const float va = WP == 2 ? acc[4 * g + e] + acc2[4 * g + e] : acc[4 * g + e];
                            const float vb = WP == 2 ? acc[4 * g + e + 1] + acc2[4 * g + e + 1] : acc[4 * g + e + 1];
                            if constexpr (HP == 2) {
                                const pfx_f16x2 hi = pkrtz(va, vb);
                                const pfx_f16x2 lo = pkrtz(va - (float)hi[0], vb - (float)hi[1]);
                                h1[e] = hi[0]; h1[e + 1] = hi[1]; h2[e] = lo[0]; h2[e + 1] = lo[1];
                            } else { h1[e] = (_Float16)va; h1[e + 1] = (_Float16)vb; }   // one piece: round to nearest (no bias)
                        }
                        *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(0 * 4 + g) * PLANE + i * YP + ro) = h1;
                        if constexpr (HP == 2) *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(1 * 4 + g) * PLANE + i * YP + ro) = h2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                stamp(it, 1);
                // (1) de-interleave: four pixels (r g b a) x 4 -> one dword per channel; a lane's PPL samples of a channel are contiguous
                // in the wave's LDS patch [c][row][sample], so they leave as 16-byte stores
                {
                    uint32_t pl[4][PPL / 4];
#pragma unroll
                    for (int q = 0; q < PPL / 4; ++q) {
                        uint32_t p0 = raw[BUF][4 * q], p1 = raw[BUF][4 * q + 1], p2 = raw[BUF][4 * q + 2], p3 = raw[BUF][4 * q + 3];
                        if constexpr (BORDER && FAST) { // the piece lies left / right of the row: every pixel of it is the row's first / last one
                            const int xp = fx + 4 * q;
                            const uint32_t e = xp < 0 ? p0 : p3;
                            const bool out = xp < 0 || xp >= w;
                            p0 = out ? e : p0; p1 = out ? e : p1; p2 = out ? e : p2; p3 = out ? e : p3;
                        }
                        const uint32_t lo01 = __builtin_amdgcn_perm(p1, p0, 0x05010400u), hi01 = __builtin_amdgcn_perm(p1, p0, 0x07030602u); // [g1 g0 r1 r0], [a1 a0 b1 b0]
                        const uint32_t lo23 = __builtin_amdgcn_perm(p3, p2, 0x05010400u), hi23 = __builtin_amdgcn_perm(p3, p2, 0x07030602u);
                        pl[0][q] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u); // r3 r2 r1 r0
                        pl[1][q] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u); // g
                        pl[2][q] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u); // b
                        pl[3][q] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u); // a
                    }
                    const uint32_t off = (uint32_t)(frow * GS_XROW + fs * PPL);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint8_t* d = xp_w + c * 8 * GS_XROW + off;
                        if constexpr (PPL == 16) *reinterpret_cast<uint4*>(d) = make_uint4(pl[c][0], pl[c][1], pl[c][2], pl[c][3]);
                        else if constexpr (PPL == 8) *reinterpret_cast<uint2*>(d) = make_uint2(pl[c][0], pl[c][1]);
                        else {
#pragma unroll
                            for (int q = 0; q < PPL / 4; ++q) reinterpret_cast<uint32_t*>(d)[q] = pl[c][q];
                        }
                    }
                }
                stamp(it, 2);
                __builtin_amdgcn_sched_barrier(0); // keep the refill HERE: the scheduler would sink it next to its use, steps later
                if (!(dbg & 8)) fetch(std::integral_constant<int, BUF>{}, it + GSD);
                __builtin_amdgcn_sched_barrier(0);
                // (2) A fragments: row m = i = channel * 8 + row, K slot (hh, kb) holds samples [16 kb + 8 hh, +8); 0x6400 | byte = 1024 + byte
                pfx_f16x8 fr[NKB];
                {
                    const uint8_t* mine = xp_w + (size_t)(i >> 3) * 8 * GS_XROW + (i & 7) * GS_XROW + 8 * hh;
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        const uint2 d = *reinterpret_cast<const uint2*>(mine + 16 * kb); // 2-way bank conflict (all addresses 8 mod 16 or 0 mod 16)
                        uint32_t qa[4];
                        qa[0] = __builtin_amdgcn_perm(d.x, 0x64646464u, 0x00050004u); qa[1] = __builtin_amdgcn_perm(d.x, 0x64646464u, 0x00070006u);
                        qa[2] = __builtin_amdgcn_perm(d.y, 0x64646464u, 0x00050004u); qa[3] = __builtin_amdgcn_perm(d.y, 0x64646464u, 0x00070006u);
                        fr[kb] = __builtin_bit_cast(pfx_f16x8, qa);
                    }
                }
                stamp(it, 3);
                if (it < n_hsteps && !(dbg & 2)) { // horizontal pass of 32 new rows
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B1(kb), kb ? acc : neg_bias, 0, 0, 0);
                        if constexpr (WP == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], B2[kb], kb ? acc2 : pfx_f32x16{}, 0, 0, 0);
                    }
                }
                stamp(it, 4);
                stamp(it, 6);
                __syncthreads();
                stamp(it, 7);
                flush_stamps(it);
            };
            fetch(std::integral_constant<int, 0>{}, 0);
            if constexpr (GSD > 1) fetch(std::integral_constant<int, 1>{}, 1);
            if constexpr (GSD > 2) fetch(std::integral_constant<int, 2>{}, 2);
            for (int it = 0; it < n_iter; it += GSD) {
                produce(std::integral_constant<int, 0>{}, it);
                if constexpr (GSD > 1) produce(std::integral_constant<int, 1>{}, it + 1);
                if constexpr (GSD > 2) produce(std::integral_constant<int, 2>{}, it + 2);
            }
        };
        const bool interior = __builtin_amdgcn_readfirstlane((x0 - R8 >= 0 && x0 - R8 + 16 * NKB <= w) ? 1 : 0) != 0;
        if (interior) walk(std::false_type{}); else walk(std::true_type{});
    } else {
        // Consumer iteration `it` runs block v = it - HALF - 1 on window rows [32 v, 32 v + 16 NKB) of the strip.  K slot (hh, kb) holds window rows
        // [16 kb + 8 hh, +8), so only the last two K blocks touch the 32 rows the producers finished in the previous iteration.  (1) request the
        // previous block's packed pixels (one 16-byte LDS read per lane) and those two K blocks; (2) MFMAs of the other K blocks, whose fragments are
        // in registers from earlier iterations; (3) the previous block goes to memory as 128-byte row segments (256 lanes x 16 bytes); (4) the last
        // two K blocks' MFMAs; (5) round, pack, stage block v.  Every LDS address is valid in every iteration, so no load sits behind a branch.
        constexpr int EARLY = NKB - 2;
        const int xb = wave - 4, xl = i >> 2, c = i & 3;
        const _Float16* a1p = HR + (size_t)(0 * 4 + c) * PLANE + (8 * xb + xl) * YP + 8 * hh;
        const _Float16* a2p = HR + (size_t)((HP - 1) * 4 + c) * PLANE + (8 * xb + xl) * YP + 8 * hh;
        const int st_t = tid - 256, st_rr = st_t >> 3, st_cg = 4 * (st_t & 7);
        // Round 4: the window's fragments STAY in registers.  Block v + 1's window is block v's moved down by 32 rows: K blocks 2 .. NKB - 1 of v are K blocks
        // 0 .. NKB - 3 of v + 1, the same rows of the same columns — the same fragment.  So a fragment lives in slot g mod NKB (g = 2 v + kb, its K block's
        // index down the strip), an iteration loads only the two K blocks whose rows the producers finished before the last barrier, and K block kb of
        // block v multiplies slot (2 v + kb) mod NKB with B[kb]: the loop is unrolled over v mod NKB / 2 so that every slot index is static.  (Rounds 2-3
        // re-read all NKB fragments of a block from the ring: 16 of the consumers' 20 LDS reads per iteration, and the reason the ring had to hold the
        // whole window.)
        pfx_f16x8 f1[NKB], f2[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { f1[kb] = pfx_f16x8{}; f2[kb] = pfx_f16x8{}; }
        auto consume = [&](auto phasec, int it) {
            constexpr int PH = decltype(phasec)::value;   // v mod HALF
            stamp(it, 0);
            const int v = it - HALF - 1, vp = v - 1;
            const bool active = v >= 0 && v < nst && !(dbg & 4);
            const int st_y = 32 * (t_first + vp) - y_phase + st_rr;
            const bool do_store = vp >= 0 && vp < nst && !(dbg & 1) && st_y >= 0 && st_y < h && (!FAST || x0 + st_cg < w);
            const uint4 ov = *reinterpret_cast<const uint4*>(OUT + (vp & 1) * 32 * GM_OUT_PITCH + st_rr * GM_OUT_PITCH + st_cg);
            // the two K blocks finished before the barrier (every LDS address is valid in every iteration: the ring offset just keeps turning)
#pragma unroll
            for (int kb = EARLY; kb < NKB; ++kb) {
                constexpr int dummy = 0; (void)dummy;
                const int ro = 32 * (it & 1) + 16 * (kb - EARLY);   // step it - 2 (stored during iteration it - 1) sits in ring half (it & 1)
                const int slot = (2 * PH + kb) % NKB;
                f1[slot] = *reinterpret_cast<const pfx_f16x8*>(a1p + ro);
                if constexpr (HP == 2) f2[slot] = *reinterpret_cast<const pfx_f16x8*>(a2p + ro);
            }
            __builtin_amdgcn_sched_barrier(0);
            stamp(it, 1);
            // chain A = h1 * w1; chain X = h1 * w2 + h2 * w1 (the two small terms share an accumulator, A's MFMA sits between them)
            pfx_f32x16 accA, accX;
            if (active) {
#pragma unroll
                for (int kb = 0; kb < EARLY; ++kb) {
                    const int slot = (2 * PH + kb) % NKB;
                    if constexpr (WP == 2) accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], B2[kb], kb ? accX : pfx_f32x16{}, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], B1(kb), kb ? accA : pfx_f32x16{}, 0, 0, 0);
                    if constexpr (HP == 2) accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[slot], B1(kb), (kb || WP == 2) ? accX : pfx_f32x16{}, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            stamp(it, 2);
            if (do_store) {
                uint32_t* drow = reinterpret_cast<uint32_t*>(dst) + (size_t)st_y * w + x0 + st_cg;
                if constexpr (FAST) *reinterpret_cast<uint4*>(drow) = ov; // w % 4 == 0, dst 16-byte aligned: the piece is wholly inside the row
                else {
                    const uint32_t o4[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (x0 + st_cg + q < w) drow[q] = o4[q];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
#pragma unroll
                for (int kb = EARLY; kb < NKB; ++kb) {
                    const int slot = (2 * PH + kb) % NKB;
                    if constexpr (WP == 2) accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], B2[kb], accX, 0, 0, 0);
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], B1(kb), accA, 0, 0, 0);
                    if constexpr (HP == 2) accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[slot], B1(kb), accX, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            stamp(it, 3);
            stamp(it, 4);
            if (active) {
                // D[m][n]: n = output row i of the block; reg q -> m = (q & 3) + 8 (q >> 2) + 4 hh = xl' * 4 + c' with c' = q & 3,
                // xl' = 2 (q >> 2) + hh: regs 4g .. 4g+3 are the RGBA of pixel (8 xb + 2 g + hh, row i)
                uint32_t* orow = OUT + (v & 1) * 32 * GM_OUT_PITCH + i * GM_OUT_PITCH + 8 * xb + hh;
                uint32_t px4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // `.round().clamp(0, 255) as u8` (filters.rs:308-311) with the final scale fused in: v_cvt_pk_u8_f32 rounds to nearest
                    // (ties to even — an exact .5 is where it can differ from the CPU's half-away rounding, inside the +-1 LSB that
                    // the f32 summation order already costs) and saturates to [0, 255]
                    uint32_t px = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) px = __builtin_amdgcn_cvt_pk_u8_f32(((WP == 2 || HP == 2) ? accA[4 * g + e] + accX[4 * g + e] : accA[4 * g + e]) * inv_scale2, e, px);
                    px4[g] = px;
                }
                if constexpr (CH) pw::chain_apply<4>(chain, nullptr, px4);
#pragma unroll
                for (int g = 0; g < 4; ++g) orow[2 * g] = px4[g];
            }
            stamp(it, 6);
            __syncthreads();
            stamp(it, 7);
            flush_stamps(it);
        };
        // iteration `it` runs block v = it - HALF - 1, whose phase v mod HALF is (it + HALF - 1) mod HALF: static in a loop unrolled HALF times
        auto phase_call = [&](auto qc, int it) {
            constexpr int Q = decltype(qc)::value;
            if (it < n_iter) consume(std::integral_constant<int, (Q + HALF - 1) % HALF>{}, it);
        };
        for (int it = 0; it < n_iter; it += HALF) {
            phase_call(std::integral_constant<int, 0>{}, it);
            phase_call(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (HALF > 2) phase_call(std::integral_constant<int, 2 % HALF>{}, it + 2);
            if constexpr (HALF > 3) phase_call(std::integral_constant<int, 3 % HALF>{}, it + 3);
            if constexpr (HALF > 4) phase_call(std::integral_constant<int, 4 % HALF>{}, it + 4);
            if constexpr (HALF > 5) phase_call(std::integral_constant<int, 5 % HALF>{}, it + 5);
        }
    }
}


// ---- 64-column strips (round 6) -----------------------------------------------------------------------------------------------------------
// The same walk with a workgroup owning 64 columns: four producer waves + EIGHT consumer waves (768 threads, one workgroup per CU).  A producer row's window is
// 16 (NKB + 2) samples for 64 outputs instead of 2 x 16 NKB for two 32-column strips: the fetch, de-interleave and fragment work per output column drops by
// (NKB + 2) / (2 NKB) (10 / 16 at sigma 16) while the MFMA count per column stays what it was — the second 32-column block of the strip multiplies the SAME A fragments,
// two K blocks further on, with the SAME Toeplitz fragments (B for block 1 at K block kb is block 0's at kb - 2: 32 columns are two K blocks).  Every output column
// therefore sees the grouping of its taps into K blocks and their accumulation order that the 32-column kernel gives it: results are bit-identical to
// gauss_strip_kernel<true, NKB, false, 1, 2> (tests/test_gpu_parity.py compares the two), band identity included.  Consumers are the 32-column kernel's, eight of them.
#ifndef PFX_G6_BREG
#define PFX_G6_BREG 1  // the consumers keep their Toeplitz fragments in registers (0: read from LDS in front of every MFMA, like the producers): -1.5 .. -3 % (profiles/r06_tuning.md)
#endif
#ifndef PFX_G6_GSD
#define PFX_G6_GSD 2   // source steps in flight per producer lane (development A/B: 3)
#endif
#ifndef PFX_G6_SWZ
#define PFX_G6_SWZ 1   // bank-conflict-free LDS layouts (round 6; tools/lds_bank_sim.py reproduces the counters: 41 % of this kernel's LDS cycles were conflicts): the ring's and the
#endif                 // window's 8-byte halves swapped for every second group of eight rows, staging rows of odd pitch read back as dwords; 0 = the layouts before
constexpr int G6_COLS = 64, G6_OUT_PITCH = PFX_G6_SWZ ? 65 : 68, G6_T = 768;
constexpr int g6_xrow(int nkp) { return 16 * nkp + 16; }
inline size_t gauss_strip64_lds_bytes(int nkb)
{
    return (size_t)4 * 2 * (G6_COLS * 72 + 32) * 2 + (size_t)2 * 32 * G6_OUT_PITCH * 4 + (size_t)4 * 4 * 8 * g6_xrow(nkb + 2) + (size_t)nkb * 64 * 16;
}
template <int NKB, class CHAIN = gs_no_chain>
__global__ __launch_bounds__(G6_T) void gauss_strip64_kernel(const CHAIN chain_arg /* first: read through the kernarg segment (k_pointwise.h) */, const uint8_t* __restrict__ src,
                                                             uint8_t* __restrict__ dst, const uint16_t* __restrict__ wsplit, int w, int h,
                                                             int r, int R8, float inv_scale2, float bias_c, int n_cols, int y_phase, int n_steps, int steps_per_seg)
{
    const pw::chain_kptr chain = pw::chain_in_kernarg();   // meaningful in the CHAIN builds only
    constexpr bool CH64 = !std::is_same<CHAIN, gs_no_chain>::value;
    static_assert(NKB == 4 || NKB == 6 || NKB == 8, "a producer lane fetches 16 pixels of the window row (+ 4 of the last 32 when the window has 160)");
    constexpr int NKP = NKB + 2, GS_XROW = g6_xrow(NKP);
    constexpr int QN = NKP == 10 ? 5 : 4;   // 16-byte pieces a producer lane fetches per row: 8 lanes x 16 pixels cover 128; a 160-pixel window adds one quad per lane, a 96-pixel one idles two lanes
    constexpr int RING = 64, YP = RING + 8, PLANE = G6_COLS * YP + 32, HALF = NKB / 2, GSD = PFX_G6_GSD;
    extern __shared__ __attribute__((aligned(16))) uint8_t gm_lds[];
    constexpr size_t RING_BYTES = (size_t)4 * 2 * PLANE * 2;
    _Float16* HR = reinterpret_cast<_Float16*>(gm_lds);                                          // [part][c][x][YP]
    uint32_t* OUT = reinterpret_cast<uint32_t*>(gm_lds + RING_BYTES);                             // [2][32][G6_OUT_PITCH]
    uint8_t* XP = gm_lds + RING_BYTES + (size_t)2 * 32 * G6_OUT_PITCH * 4;                        // [producer wave][c][row][GS_XROW]
    pfx_f16x8* const BL = reinterpret_cast<pfx_f16x8*>(XP + (size_t)4 * 4 * 8 * GS_XROW);         // Toeplitz fragments, one per K block and lane
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, hh = lane >> 5;
    const bool producer = wave < 4;
    if (wave == 0) {
        const _Float16* w1 = reinterpret_cast<const _Float16*>(wsplit) + GM_WOFF + 2 * GM_WLEN;   // the one-piece table
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const int t0 = 16 * kb + 8 * hh - i - (R8 - r);
            pfx_f16x8 b1;
#pragma unroll
            for (int j = 0; j < 8; ++j) b1[j] = w1[t0 + j];
            BL[kb * 64 + lane] = b1;
        }
    }
    __syncthreads();
    auto B1 = [&](int kb) -> pfx_f16x8 { return BL[kb * 64 + lane]; };
    const int bid = (int)xcd_swizzle(blockIdx.x, gridDim.x);
    const int ci = bid % n_cols, seg = bid / n_cols;
    const int x0 = ci * G6_COLS;
    const int t_first = seg * steps_per_seg, t_last = min(t_first + steps_per_seg, n_steps);
    if (t_first >= t_last) return;
    const int nst = t_last - t_first;
    const int a0 = 32 * t_first - y_phase - R8;
    const int n_hsteps = nst + HALF - 1;
    const int last = nst + HALF + 1, n_iter = ((last + GSD) / GSD) * GSD;

    if (producer) {
        auto walk = [&](auto borderc) {
            constexpr bool BORDER = decltype(borderc)::value;
            // fetch role of a lane: row lane >> 3 of the wave's 8; pixels [16 fs, 16 fs + 16) of the 160-pixel window and the quad [128 + 4 fs, +4)
            const int frow = lane >> 3, fs = lane & 7;
            const int fx = x0 - R8;
            auto piece_x = [&](int q) { return q < 4 ? fx + 16 * fs + 4 * q : fx + 128 + 4 * fs; };
            uint32_t raw[GSD][4 * QN];
            const bool in_window = 16 * fs < 16 * NKP;   // NKB = 4: lanes 6 and 7 of a row fetch pixels beyond the 96-pixel window (loaded from clamped addresses, never stored)
            auto fetch = [&](auto bufc, int hs_req) {
                constexpr int BUF = decltype(bufc)::value;
                const int hs = min(hs_req, n_hsteps - 1);
                const int ysrc = min(max(a0 + 32 * hs + 8 * wave + frow, 0), h - 1);
                const uint32_t* line = reinterpret_cast<const uint32_t*>(src) + (size_t)ysrc * w;
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    const int xp = piece_x(q), xc = (BORDER || NKP < 8) ? min(max(xp, 0), w - 4) : xp;
                    const uint4 v = *reinterpret_cast<const uint4*>(line + xc);
                    raw[BUF][4 * q] = v.x; raw[BUF][4 * q + 1] = v.y; raw[BUF][4 * q + 2] = v.z; raw[BUF][4 * q + 3] = v.w;
                }
            };
            uint8_t* xp_w = XP + (size_t)wave * 4 * 8 * GS_XROW;
            pfx_f32x16 acc[2], neg_bias;
#pragma unroll
            for (int q = 0; q < 16; ++q) { neg_bias[q] = -bias_c; acc[0][q] = 0.0f; acc[1][q] = 0.0f; }
            auto produce = [&](auto bufc, int it) {
                constexpr int BUF = decltype(bufc)::value;
                if (it >= 1 && it - 1 < n_hsteps) {   // (0) split and store the 32 rows x 64 columns whose MFMAs were issued before the last barrier
                    // PFX_G6_SWZ: columns 8 .. 15 (mod 16) keep the two 4-row halves of every 8 rows swapped — the sixteen lanes an 8-byte store serves together
                    // are columns i .. i + 15, 144 bytes apart: i and i + 8 met on one bank pair; the consumer wave that owns such columns swaps them back
                    const int ro = (32 * (it - 1)) % RING + 8 * wave + 4 * (PFX_G6_SWZ ? (hh ^ ((i >> 3) & 1)) : hh);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            pfx_f16x4 h1, h2;
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const float va = acc[nb][4 * g + e], vb = acc[nb][4 * g + e + 1];
                                const pfx_f16x2 hi = pkrtz(va, vb);
                                const pfx_f16x2 lo = pkrtz(va - (float)hi[0], vb - (float)hi[1]);
                                h1[e] = hi[0]; h1[e + 1] = hi[1]; h2[e] = lo[0]; h2[e + 1] = lo[1];
                            }
                            *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(0 * 4 + g) * PLANE + (32 * nb + i) * YP + ro) = h1;
                            *reinterpret_cast<pfx_f16x4*>(HR + (size_t)(1 * 4 + g) * PLANE + (32 * nb + i) * YP + ro) = h2;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                {   // (1) de-interleave: a lane's 16 + 4 samples of a channel leave as one 16-byte and one 4-byte store
                    uint32_t pl[4][QN];
#pragma unroll
                    for (int q = 0; q < QN; ++q) {
                        uint32_t p0 = raw[BUF][4 * q], p1 = raw[BUF][4 * q + 1], p2 = raw[BUF][4 * q + 2], p3 = raw[BUF][4 * q + 3];
                        if constexpr (BORDER) {
                            const int xp = piece_x(q);
                            const uint32_t e = xp < 0 ? p0 : p3;
                            const bool out = xp < 0 || xp >= w;
                            p0 = out ? e : p0; p1 = out ? e : p1; p2 = out ? e : p2; p3 = out ? e : p3;
                        }
                        const uint32_t lo01 = __builtin_amdgcn_perm(p1, p0, 0x05010400u), hi01 = __builtin_amdgcn_perm(p1, p0, 0x07030602u);
                        const uint32_t lo23 = __builtin_amdgcn_perm(p3, p2, 0x05010400u), hi23 = __builtin_amdgcn_perm(p3, p2, 0x07030602u);
                        pl[0][q] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
                        pl[1][q] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
                        pl[2][q] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
                        pl[3][q] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        uint8_t* d = xp_w + c * 8 * GS_XROW + frow * GS_XROW;
                        // PFX_G6_SWZ: window rows 16 .. 31 (channels 2, 3) keep the two 8-sample halves of every K block swapped: an 8-byte fragment read serves
                        // lanes 0 .. 31 together, rows 176 bytes apart — rows m and m + 16 met on one bank pair
                        const bool sw = PFX_G6_SWZ && c >= 2;
                        if (NKP >= 8 || in_window) *reinterpret_cast<uint4*>(d + 16 * fs) = sw ? make_uint4(pl[c][2], pl[c][3], pl[c][0], pl[c][1]) : make_uint4(pl[c][0], pl[c][1], pl[c][2], pl[c][3]);
                        if constexpr (QN == 5) *reinterpret_cast<uint32_t*>(d + 128 + ((4 * fs) ^ (sw ? 8 : 0))) = pl[c][4];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                fetch(std::integral_constant<int, BUF>{}, it + GSD);
                __builtin_amdgcn_sched_barrier(0);
                pfx_f16x8 fr[NKP];   // (2) A fragments of the 160-sample window: row m = channel * 8 + row, K slot (hh, kb) = samples [16 kb + 8 hh, +8)
                {
                    const uint8_t* mine = xp_w + (size_t)(i >> 3) * 8 * GS_XROW + (i & 7) * GS_XROW + 8 * (PFX_G6_SWZ ? (hh ^ (i >> 4)) : hh);
#pragma unroll
                    for (int kb = 0; kb < NKP; ++kb) {
                        const uint2 d = *reinterpret_cast<const uint2*>(mine + 16 * kb);
                        uint32_t qa[4];
                        qa[0] = __builtin_amdgcn_perm(d.x, 0x64646464u, 0x00050004u); qa[1] = __builtin_amdgcn_perm(d.x, 0x64646464u, 0x00070006u);
                        qa[2] = __builtin_amdgcn_perm(d.y, 0x64646464u, 0x00050004u); qa[3] = __builtin_amdgcn_perm(d.y, 0x64646464u, 0x00070006u);
                        fr[kb] = __builtin_bit_cast(pfx_f16x8, qa);
                    }
                }
                if (it < n_hsteps) {   // (3) column block nb multiplies window K blocks 2 nb .. 2 nb + NKB - 1 with Toeplitz fragments 0 .. NKB - 1: two independent chains
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb) {
                        const pfx_f16x8 b = B1(kb);
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb], b, kb ? acc[0] : neg_bias, 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[kb + 2], b, kb ? acc[1] : neg_bias, 0, 0, 0);
                    }
                }
                __syncthreads();
            };
            fetch(std::integral_constant<int, 0>{}, 0);
            fetch(std::integral_constant<int, 1>{}, 1);
            if constexpr (GSD > 2) fetch(std::integral_constant<int, 2 % GSD>{}, 2);
            for (int it = 0; it < n_iter; it += GSD) {
                produce(std::integral_constant<int, 0>{}, it);
                produce(std::integral_constant<int, 1>{}, it + 1);
                if constexpr (GSD > 2) produce(std::integral_constant<int, 2 % GSD>{}, it + 2);
            }
        };
        const bool interior = __builtin_amdgcn_readfirstlane((x0 - R8 >= 0 && x0 - R8 + 16 * NKP <= w) ? 1 : 0) != 0;
        if (interior) walk(std::false_type{}); else walk(std::true_type{});
    } else {
        // the 32-column kernel's consumer, eight waves of it: wave 4 + xb owns columns [8 xb, 8 xb + 8) of the strip
        constexpr int EARLY = NKB - 2;
        const int xb = wave - 4, xl = i >> 2, c = i & 3;
        const _Float16* a1p = HR + (size_t)(0 * 4 + c) * PLANE + (8 * xb + xl) * YP + 8 * hh;
        const _Float16* a2p = HR + (size_t)(1 * 4 + c) * PLANE + (8 * xb + xl) * YP + 8 * hh;
        // staging read-back: PFX_G6_SWZ — rows of 65 dwords (the packed-pixel stores of lanes 0 .. 31 are 32 rows of one column: an odd pitch spreads them over the
        // banks), read back as four dwords by (row 4 xb + (lane >> 3 & 3), columns 4 (lane & 7) + 32 (lane >> 5) .. + 3): lanes 0 .. 31 of a dword read hit 32 banks,
        // eight consecutive lanes store 128 contiguous bytes of an output row
        const int st_t = tid - 256;
        const int st_rr = PFX_G6_SWZ ? 4 * xb + ((lane >> 3) & 3) : st_t >> 4, st_cg = PFX_G6_SWZ ? 4 * (lane & 7) + 32 * hh : 4 * (st_t & 15);
        const bool swap_halves = PFX_G6_SWZ && (xb & 1);   // this wave's columns were stored with their 4-row halves swapped (producer step (0)); wave-uniform
        auto ring_frag = [&](const _Float16* p) -> pfx_f16x8 {
            const pfx_f16x8 v = *reinterpret_cast<const pfx_f16x8*>(p);
            return swap_halves ? __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3) : v;
        };
        pfx_f16x8 f1[NKB], f2[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { f1[kb] = pfx_f16x8{}; f2[kb] = pfx_f16x8{}; }
#if PFX_G6_BREG
        pfx_f16x8 Bc[NKB];   // the consumers' Toeplitz fragments in registers (three waves per SIMD leave 168): 16 LDS reads fewer per iteration and wave
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) { Bc[kb] = BL[kb * 64 + lane]; asm volatile("" : "+v"(Bc[kb])); }
        auto BC = [&](int kb) -> pfx_f16x8 { return Bc[kb]; };
#else
        auto BC = [&](int kb) -> pfx_f16x8 { return B1(kb); };
#endif
        auto consume = [&](auto phasec, int it) {
            constexpr int PH = decltype(phasec)::value;
            const int v = it - HALF - 1, vp = v - 1;
            const bool active = v >= 0 && v < nst;
            const int st_y = 32 * (t_first + vp) - y_phase + st_rr;
            const bool do_store = vp >= 0 && vp < nst && st_y >= 0 && st_y < h && x0 + st_cg < w;
            uint4 ov;
            {
                const uint32_t* op = OUT + (vp & 1) * 32 * G6_OUT_PITCH + st_rr * G6_OUT_PITCH + st_cg;
                if constexpr (PFX_G6_SWZ) { ov.x = op[0]; ov.y = op[1]; ov.z = op[2]; ov.w = op[3]; }
                else ov = *reinterpret_cast<const uint4*>(op);
            }
#pragma unroll
            for (int kb = EARLY; kb < NKB; ++kb) {
                const int ro = 32 * (it & 1) + 16 * (kb - EARLY);
                const int slot = (2 * PH + kb) % NKB;
                f1[slot] = ring_frag(a1p + ro);
                f2[slot] = ring_frag(a2p + ro);
            }
            __builtin_amdgcn_sched_barrier(0);
            pfx_f32x16 accA, accX;
            if (active) {
#pragma unroll
                for (int kb = 0; kb < EARLY; ++kb) {
                    const int slot = (2 * PH + kb) % NKB;
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], BC(kb), kb ? accA : pfx_f32x16{}, 0, 0, 0);
                    accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[slot], BC(kb), kb ? accX : pfx_f32x16{}, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (do_store) *reinterpret_cast<uint4*>(reinterpret_cast<uint32_t*>(dst) + (size_t)st_y * w + x0 + st_cg) = ov;
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
#pragma unroll
                for (int kb = EARLY; kb < NKB; ++kb) {
                    const int slot = (2 * PH + kb) % NKB;
                    accA = __builtin_amdgcn_mfma_f32_32x32x16_f16(f1[slot], BC(kb), accA, 0, 0, 0);
                    accX = __builtin_amdgcn_mfma_f32_32x32x16_f16(f2[slot], BC(kb), accX, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                uint32_t* orow = OUT + (v & 1) * 32 * G6_OUT_PITCH + i * G6_OUT_PITCH + 8 * xb + hh;
                uint32_t px4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t px = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) px = __builtin_amdgcn_cvt_pk_u8_f32((accA[4 * g + e] + accX[4 * g + e]) * inv_scale2, e, px);
                    px4[g] = px;
                }
                if constexpr (CH64) pw::chain_apply<4>(chain, nullptr, px4);   // a chain's light pointwise ops, between rounding and staging (pfx_chain_dev)
#pragma unroll
                for (int g = 0; g < 4; ++g) orow[2 * g] = px4[g];
            }
            __syncthreads();
        };
        auto phase_call = [&](auto qc, int it) {
            constexpr int Q = decltype(qc)::value;
            if (it < n_iter) consume(std::integral_constant<int, (Q + HALF - 1) % HALF>{}, it);
        };
        for (int it = 0; it < n_iter; it += HALF) {
            phase_call(std::integral_constant<int, 0>{}, it);
            phase_call(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (HALF > 2) phase_call(std::integral_constant<int, 2 % HALF>{}, it + 2);
            if constexpr (HALF > 3) phase_call(std::integral_constant<int, 3 % HALF>{}, it + 3);
        }
    }
}

} // namespace

// LDS bounds: H tile (1024 + 2r + 12) x 20 B and the narrowest V tile (256 + 2r + 4) rows x 5 x 16 B <= 160 KiB
extern "C" int pfxk_gauss_max_radius(void) { return 850; }
extern "C" void pfxk_gauss_set_v_config(int cfg) { g_v_cfg = cfg; }
extern "C" void pfxk_gauss_set_mfma_segments(int n) { g_mfma_seg = n > 0 && n < 256 ? n : 0; }
extern "C" void pfxk_gauss_set_mfma_parts(int wp, int hp) { g_mfma_wp = wp == 1 ? 1 : 2; g_mfma_hp = (hp == 1 && wp == 1) ? 1 : 2; }
extern "C" void pfxk_gauss_set_mfma_cols64(int mask) { g_mfma_cols64 = mask & 7; }
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
static hipError_t launch_gauss_mfma(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const uint16_t* d_wsplit,
                                    int radius, float inv_scale2, float bias_c, float bias_single, uint32_t w, uint32_t h, uint32_t first_row, int n_cus, const pfxk_chain* chain)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius < 1 || radius > GM_MAXR) return hipErrorInvalidValue;
    const int wp = g_mfma_wp, hp = g_mfma_hp;
    if (wp == 1) bias_c = bias_single;
    const int R8 = (radius + 15) & ~15;                    // window start x0 - R8: pieces of 4 pixels stay 16-byte aligned
    const int nkb = ((GM_COLS + R8 + radius + 15) / 16 + 1) & ~1; // 4, 6, 8, 10 or 12
    const int tiles_x = ((int)w + GM_COLS - 1) / GM_COLS;
    // `first_row` = index of the buffer's row 0 in the whole image when the buffer is a band of it: the 32-row output blocks lie on
    // the whole image's grid, so every output sees the same K-block grouping (the same f32 summation order) as in a whole-image call
    const int y_ph = (int)(first_row % 32u);
    const int n_steps = ((int)h + y_ph + 31) / 32;
    const bool fast = (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) == 0 && (w & 3u) == 0 && w >= 4;
    hipError_t errs = hipSuccess;
    auto launch_s = [&](auto nkb_c) {
        constexpr int NK = decltype(nkb_c)::value;
        const int hp_k = (fast && wp == 1 && hp == 1) ? 1 : 2;   // the instantiation chosen below: only the aligned one-piece build has the HP = 1 variant
        const size_t lds = gauss_strip_lds_bytes(NK, hp_k);
        // cut every strip into n_seg row segments so that the launch is ONE round of resident workgroups (LDS-bound: two per CU with 4 K blocks,
        // one from 6 up): every workgroup starts at once and none waits for a second round, and a segment pays NK/2 - 1 run-in steps.  Measured
        // (tools/lab/gauss_seg.py, sigma 16): 8K 1 / 2 / 3 segments = 0.166 / 0.175 / 0.184 ms, 4K 0.080 / 0.053 / 0.070, 1080p 4 segments 0.023
        // against 0.033 for the 7 the previous rule ("just under two workgroups per CU") chose
        // workgroups per CU: LDS (one ring + patches: two fit) and registers (the kernel is built for 4 waves per SIMD up to 8 K blocks, 2 beyond)
        const int by_regs = chain ? gs_wg_per_cu<pfxk_chain>(NK, wp) : gs_wg_per_cu<gs_no_chain>(NK, wp);
        const int resident = n_cus * (int)std::min<size_t>((size_t)by_regs, std::max<size_t>(1, (size_t)160 * 1024 / lds));
        int n_seg = resident / tiles_x;
        if (g_mfma_seg > 0) n_seg = g_mfma_seg; // tuning override (its own key: "gauss_v_cfg" only configures the VALU vertical pass)
        if (n_seg < 1) n_seg = 1;
        int per = (n_steps + n_seg - 1) / n_seg;
        if (per < 4) per = n_steps < 4 ? n_steps : 4;
        n_seg = (n_steps + per - 1) / per;
        const int grid = tiles_x * n_seg;
        const int dbg = g_v_cfg >> 9;
        auto go = [&](auto kern) {
            errs = grant_lds_for((const void*)kern, lds);
            if (errs) return;
            kern<<<grid, 512, lds, stream>>>(gs_no_chain{}, d_src, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles_x, y_ph, n_steps, per, dbg, g_dbg_buf);
        };
        if (chain) {   // the chained form exists for the shipped configuration only (aligned buffers, one-piece weights, two-piece horizontal result)
            if (!(fast && wp == 1 && hp == 2)) { errs = hipErrorNotSupported; return; }
            static lds_grant grant;
            errs = grant_lds(grant, (const void*)gauss_strip_kernel<true, NK, false, 1, 2, pfxk_chain>, lds);
            if (errs) return;
            gauss_strip_kernel<true, NK, false, 1, 2, pfxk_chain><<<grid, 512, lds, stream>>>(*chain, d_src, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles_x, y_ph,
                                                                                             n_steps, per, 0, nullptr);
            return;
        }
        if (fast && dbg && wp == 1 && hp == 2) go(gauss_strip_kernel<true, NK, true, 1, 2>); // development (tools/gauss_dbg.py, tools/gauss_timeline.py)
        else if (fast && dbg) go(gauss_strip_kernel<true, NK, true>);
        else if (fast && wp == 1 && hp == 2) go(gauss_strip_kernel<true, NK, false, 1, 2>);
        else if (fast && wp == 1 && hp == 1) go(gauss_strip_kernel<true, NK, false, 1, 1>);
        else if (fast) go(gauss_strip_kernel<true, NK, false>);
        else if (wp == 1) go(gauss_strip_kernel<false, NK, false, 1, 2>); // unaligned buffers / widths: the same tables and bias as the fast instantiation
        else go(gauss_strip_kernel<false, NK, false>);
    };
    if (nkb <= 8 && fast && wp == 1 && hp == 2 && !(g_v_cfg >> 9) && ((g_mfma_cols64 >> (nkb / 2 - 2)) & 1)) {
        // 64-column strips (up to 8 K blocks, sigma <= 16): one workgroup of twelve waves per CU; bit-identical to the 32-column kernel
        const int tiles64 = ((int)w + G6_COLS - 1) / G6_COLS;
        const size_t lds = gauss_strip64_lds_bytes(nkb);
        int n_seg = std::max(1, n_cus / tiles64);
        if (g_mfma_seg > 0) n_seg = g_mfma_seg;
        int per = (n_steps + n_seg - 1) / n_seg;
        if (per < 4) per = n_steps < 4 ? n_steps : 4;
        n_seg = (n_steps + per - 1) / per;
        auto go64 = [&](auto nkc) -> hipError_t {
            constexpr int NK = decltype(nkc)::value;
            static lds_grant grant, grant_c;
            if (chain) {
                hipError_t e = grant_lds(grant_c, (const void*)gauss_strip64_kernel<NK, pfxk_chain>, lds);
                if (e) return e;
                gauss_strip64_kernel<NK, pfxk_chain><<<tiles64 * n_seg, G6_T, lds, stream>>>(*chain, d_src, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles64, y_ph,
                                                                                            n_steps, per);
                return hipGetLastError();
            }
            hipError_t e = grant_lds(grant, (const void*)gauss_strip64_kernel<NK>, lds);
            if (e) return e;
            gauss_strip64_kernel<NK><<<tiles64 * n_seg, G6_T, lds, stream>>>(gs_no_chain{}, d_src, d_dst, d_wsplit, (int)w, (int)h, radius, R8, inv_scale2, bias_c, tiles64, y_ph, n_steps, per);
            return hipGetLastError();
        };
        return nkb == 8 ? go64(std::integral_constant<int, 8>{}) : nkb == 6 ? go64(std::integral_constant<int, 6>{}) : go64(std::integral_constant<int, 4>{});
    }
    switch (nkb) {
    case 4: launch_s(std::integral_constant<int, 4>{}); break;
    case 6: launch_s(std::integral_constant<int, 6>{}); break;
    case 8: launch_s(std::integral_constant<int, 8>{}); break;
    case 10: launch_s(std::integral_constant<int, 10>{}); break;
    default: launch_s(std::integral_constant<int, 12>{}); break;
    }
    if (errs) return errs;
    return hipGetLastError();
}

extern "C" hipError_t pfxk_gauss_mfma(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const uint16_t* d_wsplit,
                                      int radius, float inv_scale2, float bias_c, float bias_single, uint32_t w, uint32_t h, uint32_t first_row, int n_cus)
{
    return launch_gauss_mfma(stream, d_src, d_dst, d_wsplit, radius, inv_scale2, bias_c, bias_single, w, h, first_row, n_cus, nullptr);
}
// the same with a chain of table-free pointwise ops applied to every blurred pixel before it is stored; hipErrorNotSupported when the buffers / width are not
// 16-byte aligned or a development piece count is selected (the caller then runs the two launches)
extern "C" hipError_t pfxk_gauss_mfma_chain(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const uint16_t* d_wsplit, int radius, float inv_scale2, float bias_c,
                                            float bias_single, uint32_t w, uint32_t h, uint32_t first_row, int n_cus, const pfxk_chain* chain)
{
    if (!chain || chain->n == 0 || chain->n > PFXK_CHAIN_MAX || chain->n_luts != 0) return hipErrorInvalidValue;
    return launch_gauss_mfma(stream, d_src, d_dst, d_wsplit, radius, inv_scale2, bias_c, bias_single, w, h, first_row, n_cus, chain);
}
