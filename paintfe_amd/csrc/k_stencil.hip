// k_stencil.hip — the integer stencil filters: box blur, median, pixelate.  All bit-exact classes.
//
// Reference: box_blur_core src/ops/effects/blur.rs:233-318 (separable sliding window, u8 intermediate,
//            `((sum + d/2) / d) as u8` on both passes, clamp-to-edge, selection applied in the vertical pass);
//            median_core src/ops/effects/noise.rs:357-410 (per-channel element len/2 of the sorted clamped window);
//            pixelate_core src/ops/effects/distort.rs:333-373 (block-centre nearest sample).
// Integer sums are order-independent, so the reference's serial sliding sums become direct window sums for the
// first output of a lane and a 2-term slide for the following ones.
#include "k_common.h"
#include <type_traits>
#include "k_median25_net.h"
#include "k_median_shared_net.h"
#include "k_median_xlane_net.h"
#include "pfx_kernels.h"

using namespace pfxk;

namespace {

// ---------------------------------------------------------------- box blur
constexpr int BX_THREADS = 256;
constexpr int BX_PX = 8;                       // consecutive outputs per lane
constexpr int BX_TILE = BX_THREADS * BX_PX;    // 2048

struct u4 { uint32_t c[4]; };
PFX_DEV void add_px(u4& s, uint32_t px) { s.c[0] += px & 0xffu; s.c[1] += (px >> 8) & 0xffu; s.c[2] += (px >> 16) & 0xffu; s.c[3] += px >> 24; }
PFX_DEV void sub_px(u4& s, uint32_t px) { s.c[0] -= px & 0xffu; s.c[1] -= (px >> 8) & 0xffu; s.c[2] -= (px >> 16) & 0xffu; s.c[3] -= px >> 24; }
// (sum + d/2) / d with a host-built reciprocal: exact while sum * d < 2^32 (255 * d^2 < 2^32 <=> d < 4104)
PFX_DEV uint32_t div_round(uint32_t sum, uint32_t half, uint32_t magic) { return __umulhi(sum + half, magic); }
PFX_DEV uint32_t avg_px(const u4& s, uint32_t half, uint32_t magic)
{
    return div_round(s.c[0], half, magic) | (div_round(s.c[1], half, magic) << 8) |
           (div_round(s.c[2], half, magic) << 16) | (div_round(s.c[3], half, magic) << 24);
}

// LDS index of tile element i: one pad word per 32, so that lanes PX (a power of two) elements apart start in different banks —
// unpadded, the 64 lanes' windows (stride 8 words) met in 4 of the 32 banks: 16-way conflicts on every read, 3x the kernel's time
PFX_DEV int bx_pad(int i) { return i + (i >> 5); }
template <int PX>
__global__ __launch_bounds__(BX_THREADS) void box_h_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                          int r, uint32_t half, uint32_t magic, int w, int h)
{
    extern __shared__ uint32_t row_tile[]; // bx_pad(BX_THREADS * PX + 2r)
    constexpr int TILE = BX_THREADS * PX;
    const int y = blockIdx.y, x_tile = blockIdx.x * TILE;
    const int n_in = min(TILE, w - x_tile) + 2 * r;
    const uint32_t* row = src + (size_t)y * w;
    for (int i = threadIdx.x; i < n_in; i += BX_THREADS) row_tile[bx_pad(i)] = row[min(max(x_tile - r + i, 0), w - 1)];
    __syncthreads();
    const int lx0 = threadIdx.x * PX, x0 = x_tile + lx0;
    // results leave through a second LDS tile: a lane's PX outputs are consecutive words, so direct stores would touch 64 sectors per
    // instruction, 4 bytes each (measured: the kernel ran at a third of the copy rate on its write transactions alone)
    uint32_t* const out_tile = row_tile + bx_pad(TILE + 2 * r) + 1;
    if (x0 < w) {
        u4 s = {{0, 0, 0, 0}};
        for (int k = 0; k <= 2 * r; ++k) add_px(s, row_tile[bx_pad(lx0 + k)]);
#pragma unroll 8
        for (int o = 0; o < PX; ++o) {
            out_tile[bx_pad(lx0 + o)] = avg_px(s, half, magic);
            sub_px(s, row_tile[bx_pad(lx0 + o)]);                       // past the row's end these read clamped or stale words: never stored
            add_px(s, row_tile[bx_pad(min(lx0 + o + 2 * r + 1, n_in - 1))]);
        }
    }
    __syncthreads();
    const int n_out = min(TILE, w - x_tile);
    uint32_t* const out = dst + (size_t)y * w + x_tile;
    for (int i = threadIdx.x; i < n_out; i += BX_THREADS) out[i] = out_tile[bx_pad(i)];
}

// Horizontal pass for large radii: window sums as differences of a per-channel prefix sum over the row tile, so an output costs two LDS reads per channel
// whatever the radius (box_h_kernel's sliding window pays (2r + 1 + 2 PX) / PX reads per output: 14 at r = 48).  A lane sums a contiguous chunk of the staged
// tile, the 256 chunk sums are scanned (wave shuffles + one LDS hop across the four waves), the lane writes its chunk's exclusive prefixes (one plane per
// channel, padded like the tile), and output o = P[o + 2r + 1] - P[o] — integer sums, order-independent: bit-identical to the sliding window (blur.rs:262-276).
constexpr int BXP_TILE = 1024;
PFX_DEV int bxp_words(int r) { return bx_pad(BXP_TILE + 2 * r + 1) + 1; }
__global__ __launch_bounds__(BX_THREADS) void box_h_prefix_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                 int r, uint32_t half, uint32_t magic, int w, int h)
{
    extern __shared__ uint32_t sm[];
    const int words = bxp_words(r);
    uint32_t* const raw = sm;                  // staged pixels
    uint32_t* const P = sm + words;            // P[c * words + bx_pad(i)] = sum over inputs [0, i) of channel c
    uint32_t* const wt = sm + 5 * words;       // [4 waves][4 channels]
    const int tid = threadIdx.x, y = blockIdx.y, x_tile = blockIdx.x * BXP_TILE;
    const int n_out = min(BXP_TILE, w - x_tile), n_in = n_out + 2 * r;
    const uint32_t* row = src + (size_t)y * w;
    {   // all of a lane's loads are requested before the first LDS store (a load per loop trip is a chain of memory round trips: profiles/r04_tuning.md)
        constexpr int MAXT = 12;
        uint32_t v[MAXT];
#pragma unroll
        for (int k = 0; k < MAXT; ++k)
            if (k * BX_THREADS < n_in) v[k] = row[min(max(x_tile - r + min(tid + k * BX_THREADS, n_in - 1), 0), w - 1)];
#pragma unroll
        for (int k = 0; k < MAXT; ++k)
            if (k * BX_THREADS < n_in && tid + k * BX_THREADS < n_in) raw[bx_pad(tid + k * BX_THREADS)] = v[k];
        for (int i = tid + MAXT * BX_THREADS; i < n_in; i += BX_THREADS) raw[bx_pad(i)] = row[min(max(x_tile - r + i, 0), w - 1)];
    }
    __syncthreads();
    const int per = (n_in + BX_THREADS - 1) / BX_THREADS;
    const int i0 = min(tid * per, n_in), i1 = min(i0 + per, n_in);
    u4 s = {{0, 0, 0, 0}};
    for (int i = i0; i < i1; ++i) add_px(s, raw[bx_pad(i)]);
    u4 inc = s;                                // inclusive scan over the wave's 64 chunk sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc.c[c], d, 64);
            if ((tid & 63) >= d) inc.c[c] += t;
        }
    }
    if ((tid & 63) == 63) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wt[(tid >> 6) * 4 + c] = inc.c[c];
    }
    __syncthreads();
    u4 run;                                    // exclusive prefix of this lane's chunk
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t base = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) base += wt[wv * 4 + c];
        run.c[c] = base + inc.c[c] - s.c[c];
    }
    for (int i = i0; i < i1; ++i) {
        const int pi = bx_pad(i);
#pragma unroll
        for (int c = 0; c < 4; ++c) P[c * words + pi] = run.c[c];
        add_px(run, raw[pi]);
    }
    if (i1 == n_in && i0 < n_in) {             // the one lane whose chunk ends the tile also writes the total
        const int pi = bx_pad(n_in);
#pragma unroll
        for (int c = 0; c < 4; ++c) P[c * words + pi] = run.c[c];
    }
    __syncthreads();
    uint32_t* const out = dst + (size_t)y * w + x_tile;
    for (int o = tid; o < n_out; o += BX_THREADS) {
        const int a = bx_pad(o), b = bx_pad(o + 2 * r + 1);
        u4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v.c[c] = P[c * words + b] - P[c * words + a];
        out[o] = avg_px(v, half, magic);
    }
}

// Vertical pass: a lane owns PY consecutive rows of one column.  The rows entering and leaving the window are requested eight outputs ahead
// (16 loads in flight per lane): with the loads inside the per-output loop every iteration waited out a memory round trip
constexpr int BV_U = 8;
template <int PY>
__global__ __launch_bounds__(256) void box_v_kernel(const uint32_t* __restrict__ hbuf, const uint32_t* __restrict__ src,
                                                    const uint8_t* __restrict__ mask, uint32_t* __restrict__ dst, int r,
                                                    uint32_t half, uint32_t magic, int w, int h)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * PY;
    if (x >= w || y0 >= h) return;
    const uint32_t* const col = hbuf + x;
    auto at = [&](int y) { return col[(size_t)min(max(y, 0), h - 1) * w]; };
    u4 s = {{0, 0, 0, 0}};
#pragma unroll 8
    for (int k = -r; k <= r; ++k) add_px(s, at(y0 + k));
    for (int o = 0; o < PY; o += BV_U) {
        if (y0 + o >= h) break;
        uint32_t in[BV_U], out[BV_U];
#pragma unroll
        for (int j = 0; j < BV_U; ++j) { in[j] = at(y0 + o + j + r + 1); out[j] = at(y0 + o + j - r); }
#pragma unroll
        for (int j = 0; j < BV_U; ++j) {
            const int y = y0 + o + j;
            if (y < h) {
                const size_t i = (size_t)y * w + x;
                dst[i] = (mask && mask[i] == 0) ? src[i] : avg_px(s, half, magic); // blur.rs:296-303
            }
            sub_px(s, out[j]);
            add_px(s, in[j]);
        }
    }
}

// Both passes in one kernel for small radii: a block owns a BF_T x BF_T output tile, stages the (BF_T + 2r)^2 source pixels it reaches
// (clamp-to-edge, blur.rs:258,290) in LDS, runs the horizontal pass for the tile's BF_T + 2r rows into a second LDS tile — the u8
// intermediate of the reference, rounded exactly as in box_h_kernel — and the vertical pass out of that.  Same integer arithmetic,
// bit-identical to the two kernels above; the 4 B/px intermediate never reaches HBM and the halo (1.2x at r = 3) is recomputed.
constexpr int BF_T = 64, BF_RUN = 8, BF_MAXR = 4; // r >= 5: the two passes (0.145 ms at 8K) beat the fused tile (0.16 .. 0.17 ms, 1.3x halo)
// R is a template parameter (1 .. BF_MAXR): the tile geometry is a compile-time constant — the staging loop's index split is a multiply-shift, not a
// run-time division, and the window loops are straight-line LDS reads at static offsets.  All of a lane's ~20 tile elements are loaded before the first is
// stored: with one load per trip of the run-time-count loop every element was a memory round trip of its own (8K, r = 3: 0.123 ms).
template <int R>
__global__ __launch_bounds__(256) void box_fused_kernel(const uint32_t* __restrict__ src, const uint8_t* __restrict__ mask, uint32_t* __restrict__ dst,
                                                        uint32_t half, uint32_t magic, int w, int h)
{
    extern __shared__ uint32_t bf_lds[];
    constexpr int side = BF_T + 2 * R;             // source tile is side x side, horizontal results are side rows x BF_T
    constexpr int sp = side | 1;                   // odd row pitches: the runs of 8 lanes working on consecutive rows start in different banks
    constexpr int HP = BF_T + 1;                   // (even pitches put them in 16 of 32 banks on the reads and ONE bank per column on the writes)
    uint32_t* s_src = bf_lds;                      // [side][sp]
    uint32_t* s_h = bf_lds + side * sp;            // [side][HP]
    const int bx = blockIdx.x * BF_T, by = blockIdx.y * BF_T;
    {
        constexpr int NLD = (side * side + 255) / 256;   // 18 .. 21 elements per lane: ALL of a lane's loads are in flight before its first LDS store
        uint32_t v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {            // elements past the tile's end read a clamped (valid) address and are not stored
            const int i = (int)threadIdx.x + 256 * k, ty = i / side, tx = i - ty * side;
            v[k] = src[(size_t)min(max(by - R + ty, 0), h - 1) * w + min(max(bx - R + tx, 0), w - 1)];
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = (int)threadIdx.x + 256 * k, ty = i / side, tx = i - ty * side;
            if (i < side * side) s_src[ty * sp + tx] = v[k];
        }
    }
    __syncthreads();
    // horizontal: runs of BF_RUN outputs, sliding window (blur.rs:262-276)
    for (int run = threadIdx.x; run < side * (BF_T / BF_RUN); run += 256) {
        const int ty = run / (BF_T / BF_RUN), x0 = (run - ty * (BF_T / BF_RUN)) * BF_RUN;
        const uint32_t* row = s_src + ty * sp + x0; // window of output x0 + o = row[o .. o + 2r]
        uint32_t px[BF_RUN + 2 * R];
#pragma unroll
        for (int k = 0; k < BF_RUN + 2 * R; ++k) px[k] = row[k];
        u4 s = {{0, 0, 0, 0}};
#pragma unroll
        for (int k = 0; k <= 2 * R; ++k) add_px(s, px[k]);
#pragma unroll
        for (int o = 0; o < BF_RUN; ++o) {
            s_h[ty * HP + x0 + o] = avg_px(s, half, magic);
            if (o + 1 < BF_RUN) { sub_px(s, px[o]); add_px(s, px[o + 2 * R + 1]); }
        }
    }
    __syncthreads();
    // vertical: one run of BF_T / 4 rows per lane (blur.rs:294-312)
    const int lx = threadIdx.x & 63, y0 = (threadIdx.x >> 6) * (BF_T / 4);
    const int x = bx + lx;
    if (x >= w) return;
    u4 s = {{0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k <= 2 * R; ++k) add_px(s, s_h[(y0 + k) * HP + lx]);
#pragma unroll 4
    for (int o = 0; o < BF_T / 4; ++o) {
        const int y = by + y0 + o;
        if (y >= h) break;
        const size_t i = (size_t)y * w + x;
        dst[i] = (mask && mask[i] == 0) ? src[i] : avg_px(s, half, magic); // blur.rs:296-303
        sub_px(s, s_h[(y0 + o) * HP + lx]);
        add_px(s, s_h[(y0 + o + 2 * R + 1 < side ? y0 + o + 2 * R + 1 : side - 1) * HP + lx]);
    }
}

// Both passes in one kernel for radii 5 .. BS_MAXR (round 5): a column-strip walk.  A workgroup owns BS_W output columns of a segment of rows and walks down in
// blocks of BS_RB rows:
//  (1) horizontal pass (blur.rs:262-276), one wave per source row, no staging: a lane loads 4 consecutive pixels of the row's BS_W + 2r columns straight from
//      global memory (16 bytes per lane), the wave scans them (DPP: row_shr 1/2/4/8, row_bcast 15/31) into per-pixel exclusive prefix sums P — two channels
//      per register as 16-bit lanes: a 256-pixel row tile sums to <= 65 280 — and output x is (P[x + 2r + 1] - P[x] + d/2) / d: a cost per pixel that does not
//      depend on the radius.  The block's BS_RB rows of the reference's u8 intermediate go into an LDS RING of 2r + 1 + BS_RB rows;
//  (2) vertical pass (blur.rs:294-312): one running column sum per lane — add the row entering the window, emit the output row r rows behind, subtract the row
//      leaving, both read from the ring.
// The 4 B/px intermediate never reaches HBM (the two-pass path moves 16 B/px for 8 algorithmic ones).  Recomputed: 2r rows of run-in per segment and the
// strips' 2r-column halos; the halo columns are read by the neighbouring strip's workgroup too, so the launcher puts neighbouring strips on the SAME XCD (block b
// runs on XCD b % 8) where the second read is an L2 hit.  Integer sums are order-independent: bit-identical to box_h_kernel + box_v_kernel.
constexpr int BS_W = 128, BS_RB = 16, BS_MAXR = 60, BS_RP = BS_W + 1;
__host__ __device__ inline int bs_pp(int r) { return 2 * ((BS_W + 2 * r + 1 + 3) & ~3); }   // words per row of prefix pairs: BS_W + 2r entries + the total's slot, rounded to the lanes' 4-entry groups
PFX_DEV uint32_t bs_pair(uint32_t px, uint32_t sel) { return __builtin_amdgcn_perm(0u, px, sel); }            // two channels of a pixel as 16-bit lanes
constexpr uint32_t BS_SEL_RG = 0x0c010c00u, BS_SEL_BA = 0x0c030c02u;
PFX_DEV uint32_t bs_wave_scan(uint32_t v)   // inclusive add-scan over the 64 lanes
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
PFX_DEV void bs_wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// ((sum + d/2) / d) as u8 for an ODD d, written into byte `b` of `acc`: with d/2 = (d - 1)/2 the quotient is sum/d rounded to nearest, and a tie cannot occur
// (the fractional part of sum/d is a multiple of 1/d, never 1/2): the nearest integer stays nearest under any error below 1/(2d), and float(sum) * RN(1/d) is off
// by < 255.5 * 2^-23 — so the conversion's round-to-nearest (v_cvt_pk_u8_f32) yields the reference's integer division exactly (sum <= 65535 is exact in f32).
// Three full-rate instructions per channel and no packing arithmetic, where v_mul_hi_u32 alone is a quarter-rate one (tests: every radius against the oracle).
PFX_DEV uint32_t bs_avg_into(uint32_t sum16, float inv_d, uint32_t b, uint32_t acc) { return __builtin_amdgcn_cvt_pk_u8_f32((float)sum16 * inv_d, b, acc); }
} // namespace
typedef int bs_v4i __attribute__((ext_vector_type(4)));
__device__ bs_v4i bs_buffer_load_v4i32(bs_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ int bs_buffer_load_i32(bs_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void bs_buffer_store_i32(int data, bs_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
namespace {
PFX_DEV bs_v4i bs_rsrc(const void* base, uint32_t bytes)
{
    const uint64_t a = (uint64_t)base;
    bs_v4i r; r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)(0xFACu | (7u << 12) | (4u << 15)); // untyped dword access
    return r;
}
template <bool MASK>
__global__ __launch_bounds__(256) void box_strip_kernel(const uint32_t* __restrict__ src, const uint8_t* __restrict__ mask, uint32_t* __restrict__ dst, int r,
                                                        float inv_d, int w, int h, int seg_rows, int nseg, int strips)
{
    extern __shared__ uint32_t bs_lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware item order: XCD k owns a contiguous group of strips, and its blocks walk that group strip by strip within a segment
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int s_lo = strips * xcd / 8, s_hi = strips * (xcd + 1) / 8, sg = s_hi - s_lo;
    if (sg == 0 || j >= sg * nseg) return;
    const int seg = j / sg, strip = s_lo + (j - seg * sg);
    const int x0 = strip * BS_W, y0 = seg * seg_rows, y1 = min(y0 + seg_rows, h);
    if (y0 >= h) return;
    const int RR = 2 * r + 1 + BS_RB, d = 2 * r + 1, sw = BS_W + 2 * r;
    uint32_t* const s_ring = bs_lds;                                  // [RR][BS_RP]: intermediate row v lives in slot (v - v_begin) mod RR
    const int PP = bs_pp(r);
    uint2* const Pw = reinterpret_cast<uint2*>(bs_lds + ((RR * BS_RP + 3) & ~3) + wave * (2 * PP)); // this wave's prefix pairs {R|G, B|A} for two rows, 16-byte aligned
    const int v_begin = y0 - r, v_end = y1 + r;                       // intermediate ("virtual") rows this segment needs; row v is the H pass of source row clamp(v)
    const uint32_t bytes = (uint32_t)w * (uint32_t)h * 4u;
    const bs_v4i rs_src = bs_rsrc(src, bytes), rs_dst = bs_rsrc(dst, bytes);
    // horizontal-pass role: wave = 4 rows of the block, lane = 4 consecutive staged columns
    const int lx = x0 - r + 4 * lane;
    const bool interior = x0 - r >= 0 && x0 - r + sw <= w;            // the strip's staged columns lie inside the row (lanes past them read what follows: never used)
    auto load_rows = [&](int vb, bs_v4i (&px)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row_off = __builtin_amdgcn_readfirstlane(min(max(vb + wave * 4 + q, 0), h - 1) * w * 4);
            if (interior) px[q] = bs_buffer_load_v4i32(rs_src, lx * 4, row_off, 0);
            else {
                px[q].x = bs_buffer_load_i32(rs_src, min(max(lx + 0, 0), w - 1) * 4, row_off, 0); px[q].y = bs_buffer_load_i32(rs_src, min(max(lx + 1, 0), w - 1) * 4, row_off, 0);
                px[q].z = bs_buffer_load_i32(rs_src, min(max(lx + 2, 0), w - 1) * 4, row_off, 0); px[q].w = bs_buffer_load_i32(rs_src, min(max(lx + 3, 0), w - 1) * 4, row_off, 0);
            }
        }
    };
    // vertical-pass role: lane = (column, channel pair)
    const int col = tid >> 1, x = x0 + col;
    const uint32_t vsel = (tid & 1) ? BS_SEL_BA : BS_SEL_RG, vb0 = (tid & 1) ? 2u : 0u;
    const int vx = (x < w && (tid & 1) == 0) ? x * 4 : (int)0x7fffffff;   // the odd lane of a column and columns past the row store out of range: dropped by the buffer check
    uint32_t vsum = 0u;
    int base = 0;                                                     // ring slot of the block's first row
    bs_v4i cur[4], nxt[4];
    load_rows(v_begin, nxt);
    for (int vb = v_begin; vb < v_end; vb += BS_RB) {
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
        // (1) horizontal: two rows at a time (two independent scans in flight)
#pragma unroll
        for (int q = 0; q < 4; q += 2) {
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                const bs_v4i c = cur[q + qq];
                uint32_t e[2][3], ex[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const uint32_t sel = pl ? BS_SEL_BA : BS_SEL_RG;
                    e[pl][0] = bs_pair((uint32_t)c.x, sel); e[pl][1] = e[pl][0] + bs_pair((uint32_t)c.y, sel); e[pl][2] = e[pl][1] + bs_pair((uint32_t)c.z, sel);
                    const uint32_t tot = e[pl][2] + bs_pair((uint32_t)c.w, sel);
                    ex[pl] = bs_wave_scan(tot) - tot;             // sum of the pixels left of this lane's four
                }
                if (4 * lane <= sw) {                             // entries 0 .. sw (lanes past them hold pixels no window reaches)
                    uint4* out = reinterpret_cast<uint4*>(Pw + qq * (PP / 2) + 4 * lane);
                    out[0] = make_uint4(ex[0], ex[1], ex[0] + e[0][0], ex[1] + e[1][0]);
                    out[1] = make_uint4(ex[0] + e[0][1], ex[1] + e[1][1], ex[0] + e[0][2], ex[1] + e[1][2]);
                }
            }
            bs_wave_lds_sync();
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                int slot = base + wave * 4 + q + qq; slot = slot >= RR ? slot - RR : slot;
                const uint2* P = Pw + qq * (PP / 2);
#pragma unroll
                for (int hx = 0; hx < 2; ++hx) {
                    const int ox = lane + 64 * hx;                // window of output ox = staged columns [ox, ox + 2r]
                    const uint2 lo = P[ox], hi = P[ox + d];
                    const uint32_t s0 = hi.x - lo.x, s1 = hi.y - lo.y;
                    uint32_t o = bs_avg_into(s0 & 0xffffu, inv_d, 0u, 0u);
                    o = bs_avg_into(s0 >> 16, inv_d, 1u, o); o = bs_avg_into(s1 & 0xffffu, inv_d, 2u, o); o = bs_avg_into(s1 >> 16, inv_d, 3u, o);
                    s_ring[slot * BS_RP + ox] = o;
                }
            }
            bs_wave_lds_sync();                                   // the next pair of rows overwrites the planes
        }
        __syncthreads();
        if (vb + BS_RB < v_end) load_rows(vb + BS_RB, nxt);       // the next block's rows travel under the vertical pass
        // (2) vertical: all of the block's ring reads are requested before the running sum walks through them
        {
            uint32_t rin[BS_RB], rout[BS_RB];
#pragma unroll
            for (int k = 0; k < BS_RB; ++k) {
                int sn = base + k; sn = sn >= RR ? sn - RR : sn;
                int so = sn - 2 * r; so = so < 0 ? so + RR : so;  // row v - 2r = y - r leaves the window behind output y = v - r
                rin[k] = s_ring[sn * BS_RP + col];
                rout[k] = s_ring[so * BS_RP + col];               // (a slot not yet written during the run-in: read, never used)
            }
#pragma unroll
            for (int k = 0; k < BS_RB; ++k) {
                const int v = vb + k;
                if (v < v_end) {
                    vsum += bs_pair(rin[k], vsel);
                    if (v >= y0 + r) {                            // the window of output row y = v - r is complete
                        const int row_off = (v - r) * w * 4;
                        uint32_t mine = bs_avg_into(vsum & 0xffffu, inv_d, vb0, 0u);
                        mine = bs_avg_into(vsum >> 16, inv_d, vb0 + 1u, mine);
                        uint32_t o = mine | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]: the column's other channel pair
                        if constexpr (MASK) {                     // blur.rs:296-303: unselected pixels keep the source
                            if (vx != (int)0x7fffffff && mask[(size_t)(v - r) * w + x] == 0) o = (uint32_t)bs_buffer_load_i32(rs_src, vx, row_off, 0);
                        }
                        bs_buffer_store_i32((int)o, rs_dst, vx, row_off, 0);
                        vsum -= bs_pair(rout[k], vsel);
                    }
                }
            }
        }
        base += BS_RB; base = base >= RR ? base - RR : base;
        __syncthreads();                                          // the next block's horizontal pass writes ring slots this vertical pass has read
    }
}

// ---------------------------------------------------------------- median
constexpr int MD_TX = 32, MD_TY = 8; // outputs per block: one per lane

// Radii 4..7: per-channel binary search on the value, MSB first (largest t with count(x >= t) >= need).  The tile is staged in
// LDS already split into 16-bit lanes — R,B in one word, G,A in the other — so that one packed add tests two channels:
// (x + 256 - t) has bit 8 set  <=>  x >= t  (x, t in 0..255).  The flags are accumulated where they stand (bit 8 of each
// lane; a row's <= 49 flags sum to < 2^16, so lanes cannot carry into each other; rows are folded down once each) and two
// elements share one v_add3.
__global__ __launch_bounds__(MD_TX* MD_TY) void median_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                              const uint8_t* __restrict__ mask, int r, int w, int h)
{
    extern __shared__ uint2 win_tile[]; // (MD_TY + 2r) x (MD_TX + 2r), {R,B | G,A}
    const int tw = MD_TX + 2 * r, th = MD_TY + 2 * r;
    const int bx = blockIdx.x * MD_TX, by = blockIdx.y * MD_TY;
    for (int i = threadIdx.x; i < tw * th; i += MD_TX * MD_TY) {
        const int ty = i / tw, tx = i - ty * tw;
        const int sx = min(max(bx - r + tx, 0), w - 1), sy = min(max(by - r + ty, 0), h - 1); // noise.rs:389-392
        const uint32_t px = src[(size_t)sy * w + sx];
        win_tile[i] = make_uint2(px & 0x00ff00ffu, (px >> 8) & 0x00ff00ffu);
    }
    __syncthreads();
    const int lx = threadIdx.x % MD_TX, ly = threadIdx.x / MD_TX;
    const int x = bx + lx, y = by + ly;
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    if (mask && mask[oi] == 0) { dst[oi] = src[oi]; return; }
    const int side = 2 * r + 1;
    const uint32_t n = (uint32_t)(side * side);
    const uint32_t need = n - n / 2; // elements >= the median (element len/2 of the ascending sort)
    uint32_t te = 0, to = 0; // thresholds: even bytes (R,B) and odd bytes (G,A) in 16-bit lanes
    for (int bit = 7; bit >= 0; --bit) {
        const uint32_t cand_e = te | (0x00010001u << bit), cand_o = to | (0x00010001u << bit);
        const uint32_t tce = 0x01000100u - cand_e, tco = 0x01000100u - cand_o;
        uint32_t ce = 0, co = 0;
        for (int dy = 0; dy < side; ++dy) {
            const uint2* rowp = win_tile + (ly + dy) * tw + lx;
            uint32_t re = 0, ro = 0; // one row's flags, left at bit 8 of each lane (<= 49 of them)
            int dx = 0;
            for (; dx + 1 < side; dx += 2) {
                const uint2 a = rowp[dx], b = rowp[dx + 1];
                re += ((a.x + tce) & 0x01000100u) + ((b.x + tce) & 0x01000100u);
                ro += ((a.y + tco) & 0x01000100u) + ((b.y + tco) & 0x01000100u);
            }
            const uint2 a = rowp[dx]; // side is odd
            re += (a.x + tce) & 0x01000100u;
            ro += (a.y + tco) & 0x01000100u;
            ce += (re >> 8) & 0x00ff00ffu; // window totals up to 49^2 = 2401 per 16-bit lane
            co += (ro >> 8) & 0x00ff00ffu;
        }
        const uint32_t b = 1u << bit;
        if ((ce & 0xffffu) >= need) te |= b;
        if ((ce >> 16) >= need) te |= b << 16;
        if ((co & 0xffffu) >= need) to |= b;
        if ((co >> 16) >= need) to |= b << 16;
    }
    dst[oi] = (te & 0x00ff00ffu) | ((to & 0x00ff00ffu) << 8);
}

// The same search with FOUR horizontally adjacent pixels per lane: median_kernel above reads the whole window from LDS once per bit — 8 x (2r+1)^2
// ds_read_b64 per pixel, which is what bounds it (r = 7: 1 800 reads, 28.8 k LDS cycles per four waves against 20 k VALU cycles each) — while the
// windows of neighbouring pixels share all but one column.  A lane walks the 2r + 4 columns its four pixels reach and tests every element
// it loads against the thresholds of the pixels whose window holds it: 0.3 x the LDS reads per pixel, the same arithmetic per pixel.
constexpr int MQ_P = 4, MQ_TX = 32 * MQ_P, MQ_TY = 8;
__global__ __launch_bounds__(256) void median_search4_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                             const uint8_t* __restrict__ mask, int r, int w, int h)
{
    extern __shared__ uint2 win_tile[]; // (MQ_TY + 2r) x (MQ_TX + 2r), {R,B | G,A}
    const int tw = MQ_TX + 2 * r, th = MQ_TY + 2 * r;
    const int bx = blockIdx.x * MQ_TX, by = blockIdx.y * MQ_TY;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        const int sx = min(max(bx - r + tx, 0), w - 1), sy = min(max(by - r + ty, 0), h - 1); // noise.rs:389-392
        const uint32_t px = src[(size_t)sy * w + sx];
        win_tile[i] = make_uint2(px & 0x00ff00ffu, (px >> 8) & 0x00ff00ffu);
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const int x0 = bx + lx * MQ_P, y = by + ly;
    if (x0 >= w || y >= h) return;
    const int side = 2 * r + 1;
    const uint32_t n = (uint32_t)(side * side);
    const uint32_t need = n - n / 2; // elements >= the median (element len/2 of the ascending sort)
    constexpr uint32_t M = 0x01000100u;
    uint32_t te[MQ_P], to[MQ_P];
#pragma unroll
    for (int p = 0; p < MQ_P; ++p) te[p] = to[p] = 0u;
    for (int bit = 7; bit >= 0; --bit) {
        uint32_t tce[MQ_P], tco[MQ_P], ce[MQ_P], co[MQ_P];
#pragma unroll
        for (int p = 0; p < MQ_P; ++p) {
            tce[p] = M - (te[p] | (0x00010001u << bit));
            tco[p] = M - (to[p] | (0x00010001u << bit));
            ce[p] = co[p] = 0u;
        }
        for (int dy = 0; dy < side; ++dy) {
            const uint2* rowp = win_tile + (ly + dy) * tw + lx * MQ_P; // rowp[dx]: column x0 - r + dx; pixel p's window is dx in [p, p + side)
            uint32_t re[MQ_P], ro[MQ_P]; // one row's flags, left at bit 8 of each 16-bit lane (<= 49 of them)
#pragma unroll
            for (int p = 0; p < MQ_P; ++p) re[p] = ro[p] = 0u;
#pragma unroll
            for (int dx = 0; dx < MQ_P - 1; ++dx) { // leading columns: only the pixels p <= dx
                const uint2 a = rowp[dx];
#pragma unroll
                for (int p = 0; p <= dx; ++p) { re[p] += (a.x + tce[p]) & M; ro[p] += (a.y + tco[p]) & M; }
            }
            for (int dx = MQ_P - 1; dx < side; ++dx) { // every pixel's window holds these
                const uint2 a = rowp[dx];
#pragma unroll
                for (int p = 0; p < MQ_P; ++p) { re[p] += (a.x + tce[p]) & M; ro[p] += (a.y + tco[p]) & M; }
            }
#pragma unroll
            for (int k = 0; k < MQ_P - 1; ++k) { // trailing columns side + k: only the pixels p > k
                const uint2 a = rowp[side + k];
#pragma unroll
                for (int p = k + 1; p < MQ_P; ++p) { re[p] += (a.x + tce[p]) & M; ro[p] += (a.y + tco[p]) & M; }
            }
#pragma unroll
            for (int p = 0; p < MQ_P; ++p) { ce[p] += (re[p] >> 8) & 0x00ff00ffu; co[p] += (ro[p] >> 8) & 0x00ff00ffu; } // totals up to 49^2 per 16-bit lane
        }
        const uint32_t b = 1u << bit;
#pragma unroll
        for (int p = 0; p < MQ_P; ++p) {
            if ((ce[p] & 0xffffu) >= need) te[p] |= b;
            if ((ce[p] >> 16) >= need) te[p] |= b << 16;
            if ((co[p] & 0xffffu) >= need) to[p] |= b;
            if ((co[p] >> 16) >= need) to[p] |= b << 16;
        }
    }
#pragma unroll
    for (int p = 0; p < MQ_P; ++p) {
        if (x0 + p >= w) break;
        const size_t oi = (size_t)y * w + x0 + p;
        const uint32_t out = (te[p] & 0x00ff00ffu) | ((to[p] & 0x00ff00ffu) << 8);
        dst[oi] = (mask && mask[oi] == 0) ? src[oi] : out;
    }
}

// 3x3 median (the radius most callers use) without a search: sort each 3-pixel column once with v_min3 / v_med3 / v_max3, then
// median9 = med3(max of the three column minima, med of the column medians, min of the column maxima).  A lane produces 4
// adjacent pixels from 6 sorted columns, so a column sort is shared by up to three windows: ~55 VALU ops per pixel against
// ~1300 for the generic search, which turns the filter from VALU-bound into a streaming kernel.  Same integers, same result.
PFX_DEV uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PFX_DEV uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
PFX_DEV uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// VEC (w % 4 == 0): one 16-byte load per row and lane; the two edge columns come from the neighbouring lanes' registers
template <bool VEC>
__global__ __launch_bounds__(256) void median3_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                      const uint8_t* __restrict__ mask, int w, int h)
{
    const int lane = threadIdx.x & 63;
    const int x0 = (blockIdx.x * 64 + lane) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= w || y >= h) return;
    uint32_t p[3][6]; // rows y-1..y+1 (clamped), columns x0-1..x0+4 (clamped): noise.rs:389-392
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const uint32_t* row = src + (size_t)min(max(y + j - 1, 0), h - 1) * w;
        if constexpr (VEC) {
            // No load behind a per-lane condition (round 4: with the wave's two edge pixels fetched under `if (lane == 0)` / `if (lane == 63)` hipcc waited
            // vmcnt(0) between the rows — three dependent round trips per output row).  Every lane issues ONE more dword load instead: even lanes the pixel
            // left of the wave's span, odd lanes the pixel right of it (two distinct addresses per wave); lane 0 (even) and lane 63 (odd) then hold
            // their own edge, everyone else takes the neighbour's register.  A lane at the image's edge clamps to its own v.x / v.w (w % 4 == 0).
            const int wx0 = x0 - 4 * lane;
            const uint4 v = *reinterpret_cast<const uint4*>(row + x0);
            const uint32_t edge = row[(lane & 1) ? min(wx0 + 256, w - 1) : max(wx0 - 1, 0)];
            p[j][1] = v.x; p[j][2] = v.y; p[j][3] = v.z; p[j][4] = v.w;
            uint32_t left = __shfl_up(v.w, 1), right = __shfl_down(v.x, 1);
            left = lane == 0 ? edge : left;
            right = lane == 63 ? edge : right;
            right = (x0 + 4 >= w) ? v.w : right;   // last lane of the row: column x0 + 4 clamps to w - 1 = x0 + 3
            p[j][0] = left; p[j][5] = right;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) p[j][i] = row[min(max(x0 + i - 1, 0), w - 1)];
        }
    }
    uint32_t out[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint32_t lo[6], mid[6], hi[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint32_t a = (p[0][i] >> (8 * c)) & 0xffu, b = (p[1][i] >> (8 * c)) & 0xffu, d = (p[2][i] >> (8 * c)) & 0xffu;
            lo[i] = umin3(a, b, d); mid[i] = umed3(a, b, d); hi[i] = umax3(a, b, d);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            out[j] |= umed3(umax3(lo[j], lo[j + 1], lo[j + 2]), umed3(mid[j], mid[j + 1], mid[j + 2]), umin3(hi[j], hi[j + 1], hi[j + 2])) << (8 * c);
    }
    const size_t o0 = (size_t)y * w + x0;
    if constexpr (VEC) {
        if (mask) {
#pragma unroll
            for (int j = 0; j < 4; ++j) out[j] = mask[o0 + j] == 0 ? p[1][j + 1] : out[j];
        }
        *reinterpret_cast<uint4*>(dst + o0) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x0 + j >= w) break;
            dst[o0 + j] = (mask && mask[o0 + j] == 0) ? p[1][j + 1] : out[j];
        }
    }
}

// 5x5 median: a 113-comparator selection network (k_median25_net.h, generated and exhaustively verified by
// tools/gen_median_net.py) on packed 16-bit lanes — R,B in one register pair and G,A in another, so one v_pk_min_u16 /
// v_pk_max_u16 pair exchanges two channels at once: ~520 VALU ops per pixel against ~2800 for the generic search.
// 7x7 (R = 3): the same with the 313-comparator median-of-49 network (49 x 2 packed registers per lane).
typedef unsigned short pfx_us2 __attribute__((ext_vector_type(2)));
template <int R>
__global__ __launch_bounds__(MD_TX* MD_TY) void median_net_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                                  const uint8_t* __restrict__ mask, int w, int h)
{
    constexpr int r = R, side = 2 * R + 1, N = side * side, tw = MD_TX + 2 * r, th = MD_TY + 2 * r;
    __shared__ uint32_t tile[tw * th];
    const int bx = blockIdx.x * MD_TX, by = blockIdx.y * MD_TY;
    for (int i = threadIdx.x; i < tw * th; i += MD_TX * MD_TY) {
        const int ty = i / tw, tx = i - ty * tw;
        tile[i] = src[(size_t)min(max(by - r + ty, 0), h - 1) * w + min(max(bx - r + tx, 0), w - 1)]; // noise.rs:389-392
    }
    __syncthreads();
    const int lx = threadIdx.x % MD_TX, ly = threadIdx.x / MD_TX;
    const int x = bx + lx, y = by + ly;
    if (x >= w || y >= h) return;
    const size_t oi = (size_t)y * w + x;
    if (mask && mask[oi] == 0) { dst[oi] = tile[(ly + r) * tw + lx + r]; return; }
    pfx_us2 e[N], o[N];
#pragma unroll
    for (int dy = 0; dy < side; ++dy)
#pragma unroll
        for (int dx = 0; dx < side; ++dx) {
            const uint32_t p = tile[(ly + dy) * tw + lx + dx];
            e[dy * side + dx] = __builtin_bit_cast(pfx_us2, p & 0x00ff00ffu);
            o[dy * side + dx] = __builtin_bit_cast(pfx_us2, (p >> 8) & 0x00ff00ffu);
        }
#define PFX_CE(i, j)                                                                                          \
    { const pfx_us2 a = e[i], b = e[j]; e[i] = __builtin_elementwise_min(a, b); e[j] = __builtin_elementwise_max(a, b); \
      const pfx_us2 c = o[i], d = o[j]; o[i] = __builtin_elementwise_min(c, d); o[j] = __builtin_elementwise_max(c, d); }
    if constexpr (R == 2) { PFX_MEDIAN25_NET(PFX_CE) }
    else { PFX_MEDIAN49_NET(PFX_CE) }
#undef PFX_CE
    dst[oi] = __builtin_bit_cast(uint32_t, e[N / 2]) | (__builtin_bit_cast(uint32_t, o[N / 2]) << 8);
}

// 5x5 / 7x7 / 9x9 median, four adjacent windows per lane (k_median_shared_net.h, tools/gen_median_shared.py): every column is sorted once
// and serves up to four windows, the sorted run two neighbouring windows share is merged once, and a window's median is picked out of
// `shared run U one more column` — 83.5 (r = 2) / 199.5 (r = 3) min / max operations per window and channel pair instead of 226 / 626
// for the single-window selection networks above.  Same integers, same element len/2 of the ascending sort (noise.rs:398-404).
// A block is 4 rows x 64 lanes x 4 pixels; the (4 + 2r) x (256 + 2r) source tile is staged in LDS with the reference's edge clamp.
constexpr int MS_W = 256, MS_H = 4;
// DIRECT (w % 4 == 0, 16-byte aligned images — every real document): a lane fetches its 4 + 2r columns of a row as three aligned
// 16-byte quads straight from global memory (the quads left and right of its own overlap the neighbouring lanes' — L1 serves them),
// edge quads replicate the border pixel (noise.rs:389-392); no LDS tile, no barrier, all 3 (2r+1) loads in flight at once.
// Otherwise the (4 + 2r) x (256 + 2r) source tile of the block is staged in LDS with the same clamp.
template <int R, bool DIRECT>
__global__ __launch_bounds__(256) void median_shared_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                            const uint8_t* __restrict__ mask, int w, int h)
{
    constexpr int S = 2 * R + 1, NC = 4 + 2 * R, TW = MS_W + 2 * R, TH = MS_H + 2 * R;
    __shared__ uint32_t tile[DIRECT ? 1 : TH * TW];
    const int bx = blockIdx.x * MS_W, by = blockIdx.y * MS_H;
    const int lane = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int x0 = bx + lane * 4, y = by + ly;
    uint32_t px[S][NC]; // rows y-R .. y+R, columns x0-R .. x0+3+R
    if constexpr (DIRECT) {
        if (x0 >= w || y >= h) return;
        const bool first = x0 == 0, last = x0 + 4 >= w;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const uint32_t* row = src + (size_t)min(max(y + k - R, 0), h - 1) * w;
            const uint4 v = *reinterpret_cast<const uint4*>(row + x0);
            uint4 l = *reinterpret_cast<const uint4*>(row + (first ? 0 : x0 - 4));
            uint4 q = *reinterpret_cast<const uint4*>(row + (last ? x0 : x0 + 4));
            if (first) l = make_uint4(v.x, v.x, v.x, v.x);
            if (last) q = make_uint4(v.w, v.w, v.w, v.w);
            const uint32_t a[12] = {l.x, l.y, l.z, l.w, v.x, v.y, v.z, v.w, q.x, q.y, q.z, q.w};
#pragma unroll
            for (int c = 0; c < NC; ++c) px[k][c] = a[c + 4 - R];
        }
    } else {
        for (int i = threadIdx.x; i < TW * TH; i += 256) {
            const int ty = i / TW, tx = i - ty * TW;
            tile[i] = src[(size_t)min(max(by - R + ty, 0), h - 1) * w + min(max(bx - R + tx, 0), w - 1)]; // noise.rs:389-392
        }
        __syncthreads();
        if (x0 >= w || y >= h) return;
#pragma unroll
        for (int k = 0; k < S; ++k)
#pragma unroll
            for (int c = 0; c < NC; ++c) px[k][c] = tile[(ly + k) * TW + lane * 4 + c];
    }
    uint32_t out[4] = {0u, 0u, 0u, 0u};
#define PFX_MS_MIN(a, b) __builtin_elementwise_min(a, b)
#define PFX_MS_MAX(a, b) __builtin_elementwise_max(a, b)
    { // R,B in 16-bit lanes: one v_pk_min_u16 / v_pk_max_u16 handles two channels
#define PFX_MS_IN(c, k) __builtin_bit_cast(pfx_us2, px[k][c] & 0x00ff00ffu)
#define PFX_MS_OUT(j, v) out[j] = __builtin_bit_cast(uint32_t, v)
        if constexpr (R == 2) { PFX_MEDIAN_SHARED_R2(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
        else if constexpr (R == 3) { PFX_MEDIAN_SHARED_R3(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
        else { PFX_MEDIAN_SHARED_R4(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
#undef PFX_MS_IN
#undef PFX_MS_OUT
    }
    { // G,A
#define PFX_MS_IN(c, k) __builtin_bit_cast(pfx_us2, (px[k][c] >> 8) & 0x00ff00ffu)
#define PFX_MS_OUT(j, v) out[j] |= __builtin_bit_cast(uint32_t, v) << 8
        if constexpr (R == 2) { PFX_MEDIAN_SHARED_R2(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
        else if constexpr (R == 3) { PFX_MEDIAN_SHARED_R3(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
        else { PFX_MEDIAN_SHARED_R4(pfx_us2, PFX_MS_IN, PFX_MS_MIN, PFX_MS_MAX, PFX_MS_OUT) }
#undef PFX_MS_IN
#undef PFX_MS_OUT
    }
#undef PFX_MS_MIN
#undef PFX_MS_MAX
    const size_t o0 = (size_t)y * w + x0;
    if (mask) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < w && mask[o0 + j] == 0) out[j] = px[R][R + j];
    }
    if constexpr (DIRECT) *reinterpret_cast<uint4*>(dst + o0) = make_uint4(out[0], out[1], out[2], out[3]);
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < w) dst[o0 + j] = out[j];
    }
}

// 5x5 median with the sorted columns shared ACROSS LANES (round 6; k_median_xlane_net.h, tools/gen_median_xlane.py).  median_shared_kernel's lane sorts all eight
// columns its four windows touch; four of them are its neighbours' own columns and are sorted there too.  Here a lane loads and sorts only its own four columns
// (one 16-byte load per row instead of three), takes its neighbours' sorted columns and the merged pair that straddles the lane boundary with wave shifts
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1, one full-rate move per register) — 236 packed min / max + 25 moves per lane and channel pair instead of 334 min / max.
// Lanes 0 and 63 of a wave are halo lanes: they compute for their neighbours and store nothing (a wave row covers 62 x 4 = 248 output pixels).  Same integers, same
// element len/2 of the ascending sort (noise.rs:398-404); edge columns replicate the border pixel (noise.rs:389-392) because a lane left / right of the image loads
// the clamped pixel four times.
constexpr int MX_LANES_OUT = 62, MX_W = MX_LANES_OUT * 4, MX_H = 4;
// (mov_dpp, not update_dpp with a zero `old`: that form costs a v_mov_b32 of the zero in front of every shift; what lanes 0 / 63 receive is never used)
PFX_DEV pfx_us2 mx_shr(pfx_us2 v) { return __builtin_bit_cast(pfx_us2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true)); }   // from lane - 1
PFX_DEV pfx_us2 mx_shl(pfx_us2 v) { return __builtin_bit_cast(pfx_us2, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true)); }   // from lane + 1
// ROWS = 2: a lane produces TWO vertically adjacent rows of four windows from six rows of its own columns — rows 1 .. 4 of a column belong to both windows and are
// sorted once, the fifth row is merged in per window row (13 comparators per column and two rows instead of 18; 54 + 6 operations per window instead of 59 + 6)
#ifndef PFX_MX_WAVES
#define PFX_MX_WAVES 0   // development A/B: waves per SIMD the kernel is compiled for (0 = the compiler's choice: 90 registers for one row, 221 for two)
#endif
#if PFX_MX_WAVES
#define PFX_MX_ATTR __attribute__((amdgpu_waves_per_eu(PFX_MX_WAVES, PFX_MX_WAVES)))
#else
#define PFX_MX_ATTR
#endif
// R = 3 (7x7, one row per lane): the same scheme — a lane's own pairs (c3 c4), (c5 c6) are merged once and handed to both neighbours, 130.5 + 14 operations per window
// against the per-lane network's 199.5.
template <bool DIRECT, int ROWS, int R = 2>
__global__ __launch_bounds__(256) PFX_MX_ATTR void median_xlane2_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint8_t* __restrict__ mask, int w, int h)
{
    static_assert(R == 2 || (R == 3 && ROWS == 1), "generated networks: PFX_MEDIAN_XLANE_R2, _R2_ROWS2, _R3");
    constexpr int S = 2 * R + ROWS;                               // rows of own pixels a lane holds
    const int lane = threadIdx.x & 63;
    const int y = ((int)blockIdx.y * MX_H + (int)(threadIdx.x >> 6)) * ROWS;
    if (y >= h) return;                                           // whole wave: every lane of a live wave stays active (the shifts read its neighbours)
    const int x0 = (int)blockIdx.x * MX_W + 4 * (lane - 1);       // own columns x0 .. x0 + 3; lane 0 sits left of the wave's outputs, lane 63 right of them
    const int xw = (int)blockIdx.x * MX_W;
    const bool edge_wave = xw - 4 < 0 || xw + 4 * 62 + 4 > w;   // lane 0's or lane 63's quad leaves the row
    uint32_t px[S][4];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const uint32_t* row = src + (size_t)min(max(y + k - R, 0), h - 1) * w;
        if constexpr (DIRECT) {   // w % 4 == 0, rows 16-byte aligned: the quad is wholly inside the row or wholly outside it
            if (!edge_wave) {     // wave-uniform: every lane's quad lies inside the row (all waves but a row's first and last)
                const uint4 v = *reinterpret_cast<const uint4*>(row + x0);
                px[k][0] = v.x; px[k][1] = v.y; px[k][2] = v.z; px[k][3] = v.w;
            } else {
                const uint4 v = *reinterpret_cast<const uint4*>(row + min(max(x0, 0), w - 4));
                const bool left = x0 < 0, right = x0 >= w;
                px[k][0] = right ? v.w : v.x;
                px[k][1] = left ? v.x : (right ? v.w : v.y);
                px[k][2] = left ? v.x : (right ? v.w : v.z);
                px[k][3] = left ? v.x : v.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) px[k][c] = row[min(max(x0 + c, 0), w - 1)];
        }
    }
    uint32_t out[4 * ROWS];
#pragma unroll
    for (int j = 0; j < 4 * ROWS; ++j) out[j] = 0u;
#define PFX_MX_MIN(a, b) __builtin_elementwise_min(a, b)
#define PFX_MX_MAX(a, b) __builtin_elementwise_max(a, b)
    {   // R, B in 16-bit lanes
#define PFX_MX_IN(c, k) __builtin_bit_cast(pfx_us2, px[k][c] & 0x00ff00ffu)
#define PFX_MX_OUT(j, v) out[j] = __builtin_bit_cast(uint32_t, v)
        if constexpr (R == 3) { PFX_MEDIAN_XLANE_R3(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
        else if constexpr (ROWS == 1) { PFX_MEDIAN_XLANE_R2(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
        else { PFX_MEDIAN_XLANE_R2_ROWS2(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
#undef PFX_MX_IN
#undef PFX_MX_OUT
    }
    {   // G, A
#define PFX_MX_IN(c, k) __builtin_bit_cast(pfx_us2, (px[k][c] >> 8) & 0x00ff00ffu)
#define PFX_MX_OUT(j, v) out[j] |= __builtin_bit_cast(uint32_t, v) << 8
        if constexpr (R == 3) { PFX_MEDIAN_XLANE_R3(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
        else if constexpr (ROWS == 1) { PFX_MEDIAN_XLANE_R2(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
        else { PFX_MEDIAN_XLANE_R2_ROWS2(pfx_us2, PFX_MX_IN, PFX_MX_MIN, PFX_MX_MAX, mx_shr, mx_shl, PFX_MX_OUT) }
#undef PFX_MX_IN
#undef PFX_MX_OUT
    }
#undef PFX_MX_MIN
#undef PFX_MX_MAX
    if (lane == 0 || lane == 63 || x0 >= w) return;               // halo lanes and lanes right of the image: nothing to store (x0 >= 0 for lane >= 1)
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        if (y + rr >= h) break;
        const size_t o0 = (size_t)(y + rr) * w + x0;
        if (mask) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < w && mask[o0 + j] == 0) out[4 * rr + j] = px[R + rr][j];
        }
        if constexpr (DIRECT) *reinterpret_cast<uint4*>(dst + o0) = make_uint4(out[4 * rr], out[4 * rr + 1], out[4 * rr + 2], out[4 * rr + 3]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x0 + j < w) dst[o0 + j] = out[4 * rr + j];
        }
    }
}

// ---------------------------------------------------------------- pixelate
__global__ __launch_bounds__(256) void pixelate_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                       const uint8_t* __restrict__ mask, uint32_t bs, uint32_t w, uint32_t h)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * w + x;
    if (mask && mask[i] == 0) { dst[i] = src[i]; return; }
    const uint32_t sx = min((x / bs) * bs + bs / 2u, w - 1u), sy = min((y / bs) * bs + bs / 2u, h - 1u);
    dst[i] = src[(size_t)sy * w + sx];
}

// Four pixels per lane, one 16-byte store (no selection, rows 16-byte aligned): the block row is wave-uniform, the block column is divided out once per
// lane and carried over the four pixels — 8K block 8: 0.051 -> 0.038 ms.  The op writes every pixel and reads 1 / bs^2 of them.
__global__ __launch_bounds__(256) void pixelate4_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t bs, uint32_t w, uint32_t h)
{
    const uint32_t x = (blockIdx.x * 64u + (threadIdx.x & 63u)) * 4u;
    const uint32_t y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= w || y >= h) return;
    const uint32_t sy = min((y / bs) * bs + bs / 2u, h - 1u);
    const uint32_t* row = src + (size_t)sy * w;
    uint32_t q = x / bs, r = x - q * bs, v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = row[min(q * bs + bs / 2u, w - 1u)];
        if (++r == bs) { r = 0u; ++q; }
    }
    *reinterpret_cast<uint4*>(dst + (size_t)y * w + x) = make_uint4(v[0], v[1], v[2], v[3]);
}

} // namespace

int g_median_search1 = 0; // pfxk_median_set_search1: the value search with one pixel per lane (the pre-sharing kernel)
extern "C" void pfxk_median_set_search1(int on) { g_median_search1 = on; }
int g_median_xlane = 1;  // pfxk_median_set_xlane (pfx_tune "median_xlane"): bits 0-1: radius 2 on the cross-lane network, one (1) or two (2) rows per lane, or on median_shared_kernel (0); bit 2: radius 3 on it too
extern "C" void pfxk_median_set_xlane(int on) { g_median_xlane = on; }
extern "C" int pfxk_median_get_xlane(void) { return g_median_xlane; }
int g_median_single = 0; // pfxk_median_set_single: the one-window-per-lane networks for radii 2 and 3
extern "C" void pfxk_median_set_single(int on) { g_median_single = on; }
// radii from which a lane takes 16 columns instead of 8 / 64 rows instead of 16 (128 rows from twice that radius on).  Round-4 sweep of 3 x 4 shapes per radius
// (tools/lab/box_sweep.py, profiles/r04_box_sweep.txt): 8K r = 5 / 9 / 16 / 24 0.132 / 0.145 / 0.157 / 0.171 -> 0.128 / 0.139 / 0.151 / 0.165 ms; r >= 48 unchanged
int g_box_px_switch = 12, g_box_py_switch = 20;
extern "C" void pfxk_box_set_switch(int px, int py) { if (px >= 0) g_box_px_switch = px; if (py >= 0) g_box_py_switch = py; }
// radii from which the horizontal pass uses prefix sums (pfx_tune "box_prefix_from"; 0 = never).  8K, tools/lab/box_prefix_ab.py: the prefix pass costs ~0.115 ms whatever the
// radius (three barriers, a 6-step scan), the sliding window 0.06 ms at r = 9 and 0.15 at r = 100: r = 48 0.201 / 0.221 ms, r = 100 0.276 / 0.237, r = 300 0.570 / 0.339
int g_box_prefix_from = 72;
extern "C" void pfxk_box_set_prefix_from(int r) { g_box_prefix_from = r; }
inline int bxp_words_host(int r) { const int i = BXP_TILE + 2 * r + 1; return i + (i >> 5) + 1; }
int g_box_px_force = 0, g_box_py_force = 0; // development sweep (pfx_tune "box_px" / "box_py"): 0 = by radius
extern "C" void pfxk_box_set_force(int px, int py) { if (px >= 0) g_box_px_force = px; if (py >= 0) g_box_py_force = py; }
int g_box_two_pass = 0; // pfxk_box_set_two_pass: keep the u8 intermediate in HBM (the pre-fusion path; A/B and parity tests)
extern "C" void pfxk_box_set_two_pass(int on) { g_box_two_pass = on; }
int g_box_strip = 2, g_box_strip_fill = 100, g_box_strip_nseg = 0; // pfxk_box_set_strip: 2 (default) = the fused strip walk for radii 1 .. BS_MAXR (8K r = 1 .. 4: 0.085-0.098 ms against the 64 x 64 tile kernel's 0.095-0.102),
                                                                    // 1 = the tile kernel up to BF_MAXR and the strip walk above, 0 = tile kernel / two passes; chip fill in % of one round of workgroups; forced segment count
extern "C" void pfxk_box_set_strip(int on, int fill, int nseg) { if (on >= 0) g_box_strip = on; if (fill > 0) g_box_strip_fill = fill; if (nseg >= 0) g_box_strip_nseg = nseg; }
extern "C" hipError_t pfxk_box_blur(hipStream_t s, const uint8_t* d_src, uint8_t* d_tmp, uint8_t* d_dst,
                                    const uint8_t* d_mask, int radius, uint32_t w, uint32_t h, int force_two_pass)
{
    if (w == 0 || h == 0) return hipSuccess;
    const uint32_t d = (uint32_t)(2 * radius + 1);
    if (d >= 4096u) return hipErrorInvalidValue;
    const uint32_t magic = (uint32_t)((0x100000000ull / d) + 1ull), half = d / 2u;
    if (radius <= BF_MAXR && g_box_strip != 2 && g_box_two_pass == 0 && !force_two_pass && d_src != d_dst) { // small radii: both passes in one kernel (the halo recomputation stays below 1.6x)
        const int side = BF_T + 2 * radius;
        const size_t lds = (size_t)(side * (side | 1) + side * (BF_T + 1)) * 4;
        auto go = [&](auto rc) -> hipError_t {
            constexpr int R = decltype(rc)::value;
            static lds_grant grant;
            hipError_t e = grant_lds(grant, (const void*)box_fused_kernel<R>, lds);
            if (e) return e;
            box_fused_kernel<R><<<dim3((w + BF_T - 1) / BF_T, (h + BF_T - 1) / BF_T), 256, lds, s>>>((const uint32_t*)d_src, d_mask, (uint32_t*)d_dst, half, magic, (int)w, (int)h);
            return hipGetLastError();
        };
        static_assert(BF_MAXR == 4, "box_fused_kernel is instantiated for radii 1 .. 4");
        switch (radius) {
        case 1: return go(std::integral_constant<int, 1>{});
        case 2: return go(std::integral_constant<int, 2>{});
        case 3: return go(std::integral_constant<int, 3>{});
        case 4: return go(std::integral_constant<int, 4>{});
        default: break;   // radius 0 never reaches the kernels (pfx_api.cpp copies); fall through to the two-pass path
        }
    }
    if ((radius > BF_MAXR || g_box_strip == 2) && radius >= 1 && radius <= BS_MAXR && g_box_strip && g_box_two_pass == 0 && !force_two_pass && d_src != d_dst && (uint64_t)w * h < (1ull << 29)) { // fused strip walk, no intermediate in HBM
        const int RR = 2 * radius + 1 + BS_RB;
        const size_t lds = (size_t)(((RR * BS_RP + 3) & ~3) + 4 * 2 * bs_pp(radius)) * 4;   // the ring, then two rows of prefix pairs per wave on a 16-byte boundary
        const int strips = (int)((w + BS_W - 1) / BS_W);
        // One round of workgroups per XCD: an XCD owns ceil(strips / 8) strips at most and holds 32 CUs x (workgroups the LDS footprint allows) at a time; a
        // second, partly filled round would double the launch's time.  Segments are at least 6r rows (the 2r rows of run-in stay below a third).
        const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / lds));
        const int sg_max = (strips + 7) / 8;
        int nseg = std::max(1, 32 * wg_per_cu * g_box_strip_fill / 100 / sg_max);
        nseg = std::min(nseg, std::max(1, (int)h / std::max(64, 6 * radius)));
        if (g_box_strip_nseg > 0) nseg = g_box_strip_nseg;
        const int seg_rows = ((int)h + nseg - 1) / nseg;
        nseg = ((int)h + seg_rows - 1) / seg_rows;
        const float inv_d = 1.0f / (float)d;
        auto go = [&](auto mk) -> hipError_t {
            constexpr bool MK = decltype(mk)::value;
            static lds_grant grant;
            hipError_t e = grant_lds(grant, (const void*)box_strip_kernel<MK>, lds);
            if (e) return e;
            box_strip_kernel<MK><<<dim3(8u * (uint32_t)(sg_max * nseg)), 256, lds, s>>>((const uint32_t*)d_src, d_mask, (uint32_t*)d_dst, radius, inv_d, (int)w, (int)h,
                                                                                      seg_rows, nseg, strips);
            return hipGetLastError();
        };
        return d_mask ? go(std::true_type{}) : go(std::false_type{});
    }
    // outputs per lane grow with the radius: a lane's first window costs 2r + 1 reads whatever it is followed by
    auto launch_h = [&](auto px_c) -> hipError_t {
        constexpr int PX = decltype(px_c)::value;
        const int n = BX_THREADS * PX + 2 * radius, n2 = BX_THREADS * PX;
        const size_t lds = (size_t)((n + (n >> 5) + 1) + (n2 + (n2 >> 5) + 1)) * 4;
        hipError_t e = grant_lds_for((const void*)box_h_kernel<PX>, lds);
        if (e) return e;
        box_h_kernel<PX><<<dim3((w + BX_THREADS * PX - 1) / (BX_THREADS * PX), h), BX_THREADS, lds, s>>>((const uint32_t*)d_src, (uint32_t*)d_tmp, radius, half, magic, (int)w, (int)h);
        return hipGetLastError();
    };
    const int px = g_box_px_force ? g_box_px_force : (radius < g_box_px_switch ? 8 : 16);
    if (g_box_prefix_from > 0 && radius >= g_box_prefix_from && g_box_px_force == 0) {
        const size_t lds = ((size_t)5 * bxp_words_host(radius) + 16) * 4;
        if (lds <= 160u * 1024u) {
            hipError_t e0 = grant_lds_for((const void*)box_h_prefix_kernel, lds);
            if (e0) return e0;
            box_h_prefix_kernel<<<dim3((w + BXP_TILE - 1) / BXP_TILE, h), BX_THREADS, lds, s>>>((const uint32_t*)d_src, (uint32_t*)d_tmp, radius, half, magic, (int)w, (int)h);
            e0 = hipGetLastError();
            if (e0) return e0;
            goto vertical;
        }
    }
    {
    hipError_t e = px == 4 ? launch_h(std::integral_constant<int, 4>{}) : (px == 8 ? launch_h(std::integral_constant<int, 8>{}) : launch_h(std::integral_constant<int, 16>{}));
    if (e) return e;
    }
vertical:
#define PFX_BV(PY) box_v_kernel<PY><<<dim3((w + 63) / 64, (h + 4 * PY - 1) / (4 * PY)), 256, 0, s>>>((const uint32_t*)d_tmp, (const uint32_t*)d_src, d_mask, (uint32_t*)d_dst, radius, half, magic, (int)w, (int)h)
    const int py = g_box_py_force ? g_box_py_force : (radius < g_box_py_switch ? 16 : (radius < 2 * g_box_py_switch ? 64 : 128));
    if (py == 16) PFX_BV(16); else if (py == 32) PFX_BV(32); else if (py == 64) PFX_BV(64); else PFX_BV(128);
#undef PFX_BV
    return hipGetLastError();
}

// Large radii (beyond the tile kernels' LDS): Huang's sliding histogram.  One lane owns a run of MH_RUN consecutive pixels of one
// row and a private 4 x 256-bin histogram in LDS ([bin][lane] u16: lanes never share a counter); sliding one pixel to the right
// removes the window's left column and adds its right one (2 (2r+1) updates), the per-channel median follows by walking a few
// bins from the previous position.  Same result as the reference's sort (`sorted[len / 2]`, noise.rs:357-410), any radius whose
// window count fits the 16-bit counters (r <= 127).
constexpr int MH_RUN = 32;
__global__ __launch_bounds__(64) void median_hist_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                        const uint8_t* __restrict__ mask, int r, int w, int h)
{
    extern __shared__ uint16_t mh_hist[]; // [c][bin][lane]
    const int lane = threadIdx.x;
    const int runs_x = (w + MH_RUN - 1) / MH_RUN;
    const int run = blockIdx.x * 64 + lane;
    const bool live = run < runs_x * h;
    const int y = live ? run / runs_x : 0, xs = live ? (run % runs_x) * MH_RUN : 0;
    for (int i = lane; i < 4 * 256 * 64; i += 64) mh_hist[i] = 0;
    __syncthreads();
    if (!live) return;
    auto bin = [&](int c, uint32_t v) -> uint16_t& { return mh_hist[((c * 256) + (int)v) * 64 + lane]; };
    const int side = 2 * r + 1, th = (side * side) / 2;
    int med[4] = {0, 0, 0, 0}, lt[4] = {0, 0, 0, 0}; // lt[c] = window elements of channel c below med[c]
    auto add = [&](uint32_t px, int sgn) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t v = (px >> (8 * c)) & 0xffu;
            bin(c, v) = (uint16_t)(bin(c, v) + sgn);
            if ((int)v < med[c]) lt[c] += sgn;
        }
    };
    auto column = [&](int x, int sgn) {
        const int cx = min(max(x, 0), w - 1);
        for (int dy = -r; dy <= r; ++dy) add(src[(size_t)min(max(y + dy, 0), h - 1) * w + cx], sgn);
    };
    for (int dx = -r; dx <= r; ++dx) column(xs + dx, +1);
    const int xe = min(xs + MH_RUN, w);
    for (int x = xs; x < xe; ++x) {
        uint32_t out = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int m = med[c], l = lt[c];
            while (l > th) { --m; l -= bin(c, (uint32_t)m); }
            while (l + (int)bin(c, (uint32_t)m) <= th) { l += bin(c, (uint32_t)m); ++m; }
            med[c] = m; lt[c] = l;
            out |= (uint32_t)m << (8 * c);
        }
        const size_t gi = (size_t)y * w + x;
        dst[gi] = (mask && mask[gi] == 0) ? src[gi] : out;
        if (x + 1 < xe) { column(x - r, -1); column(x + r + 1, +1); }
    }
}

extern "C" hipError_t pfxk_median(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int radius,
                                  uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius <= 1) { // 3x3: min3/med3/max3 network
        const dim3 g((w + 255) / 256, (h + 3) / 4);
        if ((w & 3u) == 0 && (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) == 0) median3_kernel<true><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        else median3_kernel<false><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        return hipGetLastError();
    }
    if (radius == 3 && (g_median_xlane & 4) && !g_median_single) { // 7x7 on the cross-lane network (pfx_tune "median_xlane" bit 2; pfx_api.cpp routes r = 3 here only then)
        const bool direct = (w & 3u) == 0 && w >= 4 && (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) == 0;
        const dim3 g((w + MX_W - 1) / MX_W, (h + MX_H - 1) / MX_H);
        if (direct) median_xlane2_kernel<true, 1, 3><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        else median_xlane2_kernel<false, 1, 3><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        return hipGetLastError();
    }
    if (radius == 2 && (g_median_xlane & 3) && !g_median_single) { // 5x5: four windows per lane, sorted columns shared across lanes
        const bool direct = (w & 3u) == 0 && w >= 4 && (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) == 0;
        if ((g_median_xlane & 3) == 2) {   // two rows per lane
            const dim3 g((w + MX_W - 1) / MX_W, (h + 2 * MX_H - 1) / (2 * MX_H));
            if (direct) median_xlane2_kernel<true, 2><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
            else median_xlane2_kernel<false, 2><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        } else {
            const dim3 g((w + MX_W - 1) / MX_W, (h + MX_H - 1) / MX_H);
            if (direct) median_xlane2_kernel<true, 1><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
            else median_xlane2_kernel<false, 1><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        }
        return hipGetLastError();
    }
    if (radius >= 2 && radius <= 4 && !g_median_single) { // 5x5 / 7x7 / 9x9: four windows per lane on shared sorted columns
        const dim3 g((w + MS_W - 1) / MS_W, (h + MS_H - 1) / MS_H);
        const bool direct = (w & 3u) == 0 && (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) == 0;
#define PFX_MS(R, D) median_shared_kernel<R, D><<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h)
        if (radius == 2) { if (direct) PFX_MS(2, true); else PFX_MS(2, false); }
        else if (radius == 3) { if (direct) PFX_MS(3, true); else PFX_MS(3, false); }
        else { if (direct) PFX_MS(4, true); else PFX_MS(4, false); }
#undef PFX_MS
        return hipGetLastError();
    }
    if (radius == 2 || radius == 3) { // the single-window selection networks (pfx_tune "median_single": A/B and parity of the two paths)
        const dim3 g((w + MD_TX - 1) / MD_TX, (h + MD_TY - 1) / MD_TY);
        if (radius == 2) median_net_kernel<2><<<g, MD_TX * MD_TY, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        else median_net_kernel<3><<<g, MD_TX * MD_TY, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, (int)w, (int)h);
        return hipGetLastError();
    }
    if (radius > PFXK_MEDIAN_TILE_MAX_RADIUS) { // sliding histogram
        const size_t lds_h = (size_t)4 * 256 * 64 * sizeof(uint16_t);
        hipError_t eh = grant_lds_for((const void*)median_hist_kernel, lds_h);
        if (eh) return eh;
        const size_t runs = (size_t)((w + MH_RUN - 1) / MH_RUN) * h;
        median_hist_kernel<<<(uint32_t)((runs + 63) / 64), 64, lds_h, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, radius, (int)w, (int)h);
        return hipGetLastError();
    }
    if (!g_median_search1) { // four pixels per lane: a third of the LDS reads per pixel
        const size_t lds4 = (size_t)(MQ_TX + 2 * radius) * (MQ_TY + 2 * radius) * 8;
        hipError_t e4 = grant_lds_for((const void*)median_search4_kernel, lds4);
        if (e4) return e4;
        median_search4_kernel<<<dim3((w + MQ_TX - 1) / MQ_TX, (h + MQ_TY - 1) / MQ_TY), 256, lds4, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, radius, (int)w, (int)h);
        return hipGetLastError();
    }
    const size_t lds = (size_t)(MD_TX + 2 * radius) * (MD_TY + 2 * radius) * 8;
    hipError_t e = grant_lds_for((const void*)median_kernel, lds);
    if (e) return e;
    dim3 g((w + MD_TX - 1) / MD_TX, (h + MD_TY - 1) / MD_TY);
    median_kernel<<<g, MD_TX * MD_TY, lds, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, radius, (int)w, (int)h);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_pixelate(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, uint32_t bs,
                                    uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (!d_mask && (w & 3u) == 0u && ((uintptr_t)d_dst & 15u) == 0u) {
        pixelate4_kernel<<<dim3((w / 4u + 63u) / 64u, (h + 3) / 4), 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, bs, w, h);
        return hipGetLastError();
    }
    dim3 g((w + 63) / 64, (h + 3) / 4);
    pixelate_kernel<<<g, 256, 0, s>>>((const uint32_t*)d_src, (uint32_t*)d_dst, d_mask, bs, w, h);
    return hipGetLastError();
}
