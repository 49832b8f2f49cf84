// k_median_bits.hip — median_core (src/ops/effects/noise.rs:357-410) for radii 3..8 as a bit-sliced radix select.
//
// The reference sorts the (2r+1)^2 clamped window of every channel and takes element len/2.  The value search of k_stencil.hip
// (median_search4_kernel) finds the same element with 8 threshold counts over the whole window: 8 x 225 byte compares per channel
// at r = 7.  Here the window is held as BIT PLANES instead of bytes: plane b of a window is one bit per element, a (2r+1)-bit field per
// window row, two to four rows to a register — 8 registers per plane at r = 7 — and the rank-k element is selected from the most
// significant plane down:
//     zeros = cand & ~plane_b;  c0 = popcount(zeros);  k < c0 ? (cand = zeros, bit b = 0) : (k -= c0, cand &= plane_b, bit b = 1)
// i.e. 4 instructions per REGISTER (32 window elements) and plane: ~25 per plane at r = 7 instead of 225 byte compares.
// A lane owns one channel of one image column and walks down: the window loses its top row and gains a bottom row, which is one
// (2r+1)-bit field per plane — eight funnel shifts out of the image row's bit planes — written over the slot of the row that left
// (the slot order inside the registers is irrelevant to a popcount).
//
// The bit planes of the whole image are produced by a pre-pass (median_planes_kernel): a 32 x 32 bit transpose per half wave, rows padded by
// r replicated pixels on either side (the reference's clamp-to-edge, noise.rs:383), stored inverted (the select wants ~plane).
// Scratch: 128 bytes per row and 32 padded columns, ~ the image size.
#include "k_common.h"
#include "pfx_kernels.h"
#include <type_traits>

typedef int mb_v4i __attribute__((ext_vector_type(4)));
__device__ void mb_buffer_store_i8(uint8_t data, mb_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i8");
__device__ uint8_t mb_buffer_load_i8(mb_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i8");

namespace {
using namespace pfxk;

constexpr int MB_COLS = 16;   // columns per wave (x 4 channels = 64 lanes)
constexpr int MB_ROWS = 32;   // output rows per wave

__host__ __device__ inline uint32_t mb_dwords(uint32_t w, int r) { return (w + 2u * (uint32_t)r + 31u) / 32u + 1u; } // + the funnel shift's high dword

// planes[((y * 4 + c) * ND + j) * 8 + b]: bit t = NOT bit b of channel c of pixel (clamp(32 j + t - r), y)
constexpr int MP_ROWS = 4; // rows per wave: that many loads in flight (one pixel per lane and load; the kernel is a pure stream)
__global__ __launch_bounds__(256) void median_planes_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ planes, int r, int w, int h,
                                                           uint32_t nd)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = blockIdx.x * 4u + (threadIdx.x >> 6); // 64 padded positions = dword columns 2 wv, 2 wv + 1
    const uint32_t y0 = blockIdx.y * MP_ROWS;
    if (2u * wv >= nd) return;
    const int x = min(max((int)(wv * 64u + lane) - r, 0), w - 1);
    uint32_t v[MP_ROWS];
#pragma unroll
    for (int i = 0; i < MP_ROWS; ++i) v[i] = ~src[(size_t)min(y0 + i, (uint32_t)h - 1u) * w + x];
    const uint32_t hh = lane >> 5, c = (lane >> 3) & 3u, b = lane & 7u, j = 2u * wv + hh;
    // 32 x 32 bit-matrix transpose inside each half of the wave (lane t holds pixel t's 32 bits -> lane i holds bit i of 32 pixels): five
    // butterfly stages, each exchanging with lane ^ j (ds_swizzle, no memory) the off-diagonal j x j blocks; a rotation brings the partner's
    // block under the mask on either side of the exchange
#define PFX_TR_STAGE(J, M)                                                                                       \
    {                                                                                                            \
        const uint32_t other = (uint32_t)__builtin_amdgcn_ds_swizzle((int)out, ((J) << 10) | 0x1f);              \
        const bool lo = (lane & (J)) == 0u;                                                                      \
        const uint32_t rot = __builtin_amdgcn_alignbit(other, other, lo ? 32u - (J) : (uint32_t)(J));            \
        const uint32_t keep = lo ? (M) : ~(M);                                                                   \
        out = (out & keep) | (rot & ~keep);                                                                      \
    }
#pragma unroll
    for (int i = 0; i < MP_ROWS; ++i) {
        uint32_t out = v[i];
        PFX_TR_STAGE(16u, 0x0000ffffu) PFX_TR_STAGE(8u, 0x00ff00ffu) PFX_TR_STAGE(4u, 0x0f0f0f0fu) PFX_TR_STAGE(2u, 0x33333333u) PFX_TR_STAGE(1u, 0x55555555u)
        if (j < nd && y0 + i < (uint32_t)h) planes[(((size_t)(y0 + i) * 4u + c) * nd + j) * 8u + b] = out;
    }
#undef PFX_TR_STAGE
}

template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int R> struct mb_geom {
    static constexpr int S = 2 * R + 1;                                  // window side: rows, and bits per row
    // r = 8 (the reference dialog's maximum, ui/dialogs/effects/noise.rs): a 17-bit row does not pack two to a register — its first 16 columns
    // do, and the 17th column of all 17 rows gets a register of its own (the order of the elements is irrelevant to a popcount)
    static constexpr bool XCOL = S > 16;
    static constexpr int MW = XCOL ? 16 : S;                             // bits of a row that go into the packed fields
    static constexpr int RPR = MW <= 5 ? 6 : MW <= 7 ? 4 : MW <= 9 ? 3 : 2; // rows per register
    static constexpr int FO = 32 / RPR;                                  // field pitch inside a register (>= MW)
    static constexpr int NRM = (S + RPR - 1) / RPR;                      // registers of packed fields per plane
    static constexpr int NR = NRM + (XCOL ? 1 : 0);                      // registers per plane
    static constexpr uint32_t FM = MW == 32 ? ~0u : (1u << MW) - 1u;
    static constexpr uint32_t cand_init(int reg)
    {
        if (reg >= NRM) return (1u << S) - 1u;                           // the extra column: one bit per row
        uint32_t m = 0;
        for (int f = 0; f < RPR; ++f) if (reg * RPR + f < S) m |= FM << (f * FO);
        return m;
    }
};

template <int R>
__global__ __launch_bounds__(256) void median_bits_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ planes, uint8_t* __restrict__ dst,
                                                         const uint8_t* __restrict__ mask, int w, int h, uint32_t nd)
{
    using G = mb_geom<R>;
    constexpr int S = G::S, NR = G::NR;
    constexpr int NROW = MB_ROWS + 2 * R;   // plane rows a block walks through
    // the block's slice of the bit planes: [row][channel][dword 0..2][plane] — 64 columns + 2r bits span three dwords of a plane row.  Staged once
    // (one exposed memory latency per block instead of one per row), read back 64 contiguous bytes per lane and row.
    __shared__ uint4 s_pl[NROW * 24];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const int x0 = (int)blockIdx.x * 4 * MB_COLS;
    const int x = x0 + (int)(wid * MB_COLS + (lane >> 2));
    const uint32_t c = lane & 3u;
    const int y0 = (int)blockIdx.y * MB_ROWS, y1 = min(y0 + MB_ROWS, h);
    {
        const uint32_t jq = (uint32_t)x0 >> 5;                       // first dword column (x0 is a multiple of 64)
        const uint32_t row_q = nd * 8u;                              // uint4 per image row: 4 channels x nd dwords x 8 planes / 4
        const uint4* const pq = reinterpret_cast<const uint4*>(planes);
        for (uint32_t i = threadIdx.x; i < (uint32_t)NROW * 24u; i += 256u) {
            const uint32_t row = i / 24u, rem = i - row * 24u, cc = rem / 6u, q = rem - cc * 6u; // q: 16-byte piece of the 96-byte run
            const uint32_t yy = (uint32_t)min(max(y0 - R + (int)row, 0), h - 1);
            const uint32_t dj = min(jq + (q >> 1), nd - 1u);          // past the row's last dword only for columns >= w
            s_pl[i] = pq[(size_t)yy * row_q + ((size_t)cc * nd + dj) * 2u + (q & 1u)];
        }
    }
    __syncthreads();
    if ((int)(x0 + wid * MB_COLS) >= w) return;
    const int xc = min(x, w - 1);
    const uint32_t sh = (uint32_t)xc & 31u;
    const uint4* const lcol = s_pl + c * 6u + (((uint32_t)(xc - x0) >> 5) * 2u); // + row * 24

    uint32_t np[8][NR];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int j = 0; j < NR; ++j) np[b][j] = 0u;

    uint32_t f[8]; // the row being inserted: its S-bit field per plane (bits above S are garbage the candidate mask never selects)
    auto fetch = [&](int lrow) __attribute__((always_inline)) { // lrow: row relative to y0 - R
        const uint4* p = lcol + lrow * 24;
        const uint4 l0 = p[0], l1 = p[1], h0 = p[2], h1 = p[3];
        f[0] = __builtin_amdgcn_alignbit(h0.x, l0.x, sh); f[1] = __builtin_amdgcn_alignbit(h0.y, l0.y, sh);
        f[2] = __builtin_amdgcn_alignbit(h0.z, l0.z, sh); f[3] = __builtin_amdgcn_alignbit(h0.w, l0.w, sh);
        f[4] = __builtin_amdgcn_alignbit(h1.x, l1.x, sh); f[5] = __builtin_amdgcn_alignbit(h1.y, l1.y, sh);
        f[6] = __builtin_amdgcn_alignbit(h1.z, l1.z, sh); f[7] = __builtin_amdgcn_alignbit(h1.w, l1.w, sh);
    };
    // the slot of a row is static: the row loop is unrolled S times (a run-time slot index costs a branch tree whose merges copy every
    // plane register: measured as many v_mov as the select has instructions)
    auto insert = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int s = decltype(slot_c)::value, reg = s / G::RPR, off = (s % G::RPR) * G::FO;
        constexpr uint32_t fm = G::FM << off;
        // (np & ~fm) | (f << off & fm) as one full-rate v_bitop3 (v_bfi issues at half rate); unused register bits stay 0
#pragma unroll
        for (int b = 0; b < 8; ++b) np[b][reg] = __builtin_amdgcn_bitop3_b32(np[b][reg], f[b] << off, fm, 0xd8);
        if constexpr (G::XCOL) { // bit 16 of the row's field -> bit s of the column register
#pragma unroll
            for (int b = 0; b < 8; ++b)
                np[b][NR - 1] = __builtin_amdgcn_bitop3_b32(np[b][NR - 1], s < 16 ? f[b] >> (16 - s) : f[b] << (s - 16), 1u << s, 0xd8);
        }
    };
    // The select's instruction sequence is pinned with inline asm: per register and plane one v_and (zeros among the candidates), one
    // accumulating v_bcnt, one v_bitop3 (cand &= plane ^ ones).  Left to itself hipcc re-derives the candidate sets from the planes with
    // extra xor / bitop3 pairs and sums the counts with half-rate v_add3 (measured: 12 issue cycles per register and plane instead of 8).
    auto select = [&]() __attribute__((always_inline)) -> uint32_t {
        uint32_t cand[NR];
        uint32_t k = (uint32_t)(S * S) / 2u, res = 0u; // sorted[len / 2], noise.rs:401
#pragma unroll
        for (int b = 7; b >= 0; --b) {
            uint32_t cnt0 = 0u, cnt1 = 0u, z;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                // every element is a candidate of the top plane, and the planes hold 0 outside the window's fields
                if (b == 7) { if (j & 1) cnt1 += (uint32_t)__builtin_popcount(np[7][j]); else cnt0 += (uint32_t)__builtin_popcount(np[7][j]); }
                else if (j == 0) asm("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, 0" : "=v"(cnt0), "=&v"(z) : "v"(cand[j]), "v"(np[b][j]));
                else if (j == 1) asm("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, 0" : "=v"(cnt1), "=&v"(z) : "v"(cand[j]), "v"(np[b][j]));
                else if (j & 1) asm("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt1), "=&v"(z) : "v"(cand[j]), "v"(np[b][j]));
                else asm("v_and_b32 %1, %2, %3\n\tv_bcnt_u32_b32 %0, %1, %0" : "+v"(cnt0), "=&v"(z) : "v"(cand[j]), "v"(np[b][j]));
            }
            const uint32_t c0 = cnt0 + cnt1;              // candidates whose bit b is 0
            const uint32_t d = k - c0;
            uint32_t zm;                                   // k < c0 (both below 2^8): the rank-k candidate has bit b clear -> all ones
            asm("v_ashrrev_i32 %0, 31, %1" : "=v"(zm) : "v"(d));
            if (b > 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) {             // cand & ~(plane ^ zm): keep the zeros (zm) or the ones
                    if (b == 7) cand[j] = __builtin_amdgcn_bitop3_b32(G::cand_init(j), np[7][j], zm, 0x90);
                    else asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(cand[j]) : "v"(np[b][j]), "v"(zm));
                }
                k = min(k, d);                             // k < c0: d wrapped around and k stays
            }
            res = __builtin_amdgcn_bitop3_b32(res, zm, 1u << b, 0xf2); // res | (~zm & bit)
        }
        return res;
    };
    auto emit = [&](int y, uint32_t res) __attribute__((always_inline)) {
        if (x < w) {
            const size_t gi = (size_t)y * w + x;
            uint8_t o = (uint8_t)res;
            if (mask && mask[gi] == 0) o = src[gi * 4u + c];
            dst[gi * 4u + c] = o;
        }
    };

    static_for<0, S - 1>([&](auto sc) __attribute__((always_inline)) { // the first window's rows but the last: slots 0 .. S-2
        fetch((int)decltype(sc)::value);
        insert(sc);
    });
    for (int y = y0; y < y1; y += S) {
        bool done = false;
        static_for<0, S>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            if (!done && y + u < y1) {
                fetch(y - y0 + u + 2 * R);
                insert(std::integral_constant<int, (S - 1 + u) % S>{});
                emit(y + u, select());
            } else done = true;
        });
    }
}


// ---- two adjacent columns per lane (radii 2..7) ----
// The windows of columns x and x + 1 share 2r of their 2r + 1 columns.  A lane of this kernel owns one channel of a column PAIR: its plane registers
// hold (2r + 2)-bit fields (the union of the two windows' columns) and both selects run on the same registers, each under its own candidate mask
// (bits 0..2r of every field for x, bits 1..2r+1 for x + 1).  Fetching a row's fields from LDS and inserting them — as many instructions as before —
// is paid once per two results, and so is the LDS traffic.  The field widths still pack as the single-column kernel's do (r = 3: four 8-bit fields
// per register, r = 4: three of 10, r = 5..7: two of 16), so a select costs what it did; radius 8 (18-bit fields) stays on the kernel above.
// The selects of the pair run one after the other in a two-trip loop (interleaving them buys nothing: tools/lab/select_chain.hip) so that the
// code, unrolled over the 2r + 1 row slots, stays the size of the single-column kernel's.
__device__ __forceinline__ mb_v4i mb_rsrc(const void* base, uint32_t bytes)
{
    const unsigned long long a = (unsigned long long)base;
    mb_v4i r; r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)(0xFACu | (7u << 12) | (4u << 15)); return r;
}

constexpr int MB2_COLS = 128; // columns per block: 4 waves x 16 pairs
#ifndef PFX_MB2_ROWS
#define PFX_MB2_ROWS 32
#endif
#ifndef PFX_MB2_KSEL
#define PFX_MB2_KSEL 1       // the remaining rank by a bitop3 select (full rate) instead of v_min_u32
#endif

template <int R> struct mb2_geom {
    static constexpr int S = 2 * R + 1;
    static constexpr int FW = S + 1;                                          // field: the columns of both windows
    static constexpr int RPR = FW <= 6 ? 5 : FW <= 8 ? 4 : FW <= 10 ? 3 : 2;  // rows per register (FW <= 16)
    static constexpr int FO = 32 / RPR;
    static constexpr int NR = (S + RPR - 1) / RPR;
    static constexpr uint32_t FM = (1u << FW) - 1u;
    static constexpr uint32_t cand_init(int reg)                              // window of column x; << 1 for x + 1
    {
        uint32_t m = 0;
        for (int f = 0; f < RPR; ++f) if (reg * RPR + f < S) m |= ((1u << S) - 1u) << (f * FO);
        return m;
    }
};

template <int R, bool MASK>
__global__ __launch_bounds__(256) void median_bits2_kernel(const uint8_t* __restrict__ src, const uint32_t* __restrict__ planes, uint8_t* __restrict__ dst,
                                                          const uint8_t* __restrict__ mask, int w, int h, uint32_t nd)
{
    using G = mb2_geom<R>;
    constexpr int S = G::S, NR = G::NR, ROWS = PFX_MB2_ROWS;
    constexpr int NROW = ROWS + 2 * R;
    // [row][channel][dword 0..4][plane]: 128 columns + 2r + 1 bits span five dwords of a plane row (160 bytes per row and channel)
    __shared__ uint4 s_pl[NROW * 40];
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const int x0 = (int)blockIdx.x * MB2_COLS;
    const int xa = x0 + (int)(wid * 32u + (lane >> 2) * 2u);
    const uint32_t c = lane & 3u;
    const int y0 = (int)blockIdx.y * ROWS, y1 = min(y0 + ROWS, h);
    {
        const uint32_t jq = (uint32_t)x0 >> 5;
        const uint32_t row_q = nd * 8u;                              // uint4 per image row
        const uint4* const pq = reinterpret_cast<const uint4*>(planes);
        for (uint32_t i = threadIdx.x; i < (uint32_t)NROW * 40u; i += 256u) {
            const uint32_t row = i / 40u, rem = i - row * 40u, cc = rem / 10u, q = rem - cc * 10u; // q: 16-byte piece of the 160-byte run
            const uint32_t yy = (uint32_t)min(max(y0 - R + (int)row, 0), h - 1);
            const uint32_t dj = min(jq + (q >> 1), nd - 1u);          // past the row's last dword only for columns >= w
            s_pl[i] = pq[(size_t)yy * row_q + ((size_t)cc * nd + dj) * 2u + (q & 1u)];
        }
    }
    __syncthreads();
    if ((int)(x0 + wid * 32u) >= w) return;
    const uint32_t sh = (uint32_t)xa & 31u;
    const uint4* const lcol = s_pl + c * 10u + (((uint32_t)(xa - x0) >> 5) * 2u); // + row * 40

    uint32_t np[8][NR];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int j = 0; j < NR; ++j) np[b][j] = 0u;

    uint32_t f[8];
    auto fetch = [&](int lrow) __attribute__((always_inline)) {
        const uint4* p = lcol + lrow * 40;
        const uint4 l0 = p[0], l1 = p[1], h0 = p[2], h1 = p[3];
        f[0] = __builtin_amdgcn_alignbit(h0.x, l0.x, sh); f[1] = __builtin_amdgcn_alignbit(h0.y, l0.y, sh);
        f[2] = __builtin_amdgcn_alignbit(h0.z, l0.z, sh); f[3] = __builtin_amdgcn_alignbit(h0.w, l0.w, sh);
        f[4] = __builtin_amdgcn_alignbit(h1.x, l1.x, sh); f[5] = __builtin_amdgcn_alignbit(h1.y, l1.y, sh);
        f[6] = __builtin_amdgcn_alignbit(h1.z, l1.z, sh); f[7] = __builtin_amdgcn_alignbit(h1.w, l1.w, sh);
    };
    auto insert = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int s = decltype(slot_c)::value, reg = s / G::RPR, off = (s % G::RPR) * G::FO;
        constexpr uint32_t fm = G::FM << off;
#pragma unroll
        for (int b = 0; b < 8; ++b) np[b][reg] = __builtin_amdgcn_bitop3_b32(np[b][reg], f[b] << off, fm, 0xd8);
    };
    auto select = [&](uint32_t win) __attribute__((always_inline)) -> uint32_t {
        uint32_t cand[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) cand[j] = G::cand_init(j) << win;
        uint32_t k = (uint32_t)(S * S) / 2u, res = 0u; // sorted[len / 2], noise.rs:401
#pragma unroll
        for (int b = 7; b >= 0; --b) {
            uint32_t z[NR], cnt0 = 0u, cnt1 = 0u;
#pragma unroll
            for (int j = 0; j < NR; ++j) asm("v_and_b32 %0, %1, %2" : "=v"(z[j]) : "v"(cand[j]), "v"(np[b][j]));
#pragma unroll
            for (int j = 0; j < NR; ++j) {   // two accumulation chains from four registers on
                if (NR >= 4 && (j & 1)) asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(cnt1) : "v"(z[j]), "v"(cnt1));
                else asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(cnt0) : "v"(z[j]), "v"(cnt0));
            }
            const uint32_t c0 = NR >= 4 ? cnt0 + cnt1 : cnt0;   // candidates whose bit b is 0
            const uint32_t d = k - c0;
            uint32_t zm;                                        // k < c0: the rank-k candidate has bit b clear -> all ones
            asm("v_ashrrev_i32 %0, 31, %1" : "=v"(zm) : "v"(d));
            if (b > 0) {
#pragma unroll
                for (int j = 0; j < NR; ++j) asm("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x90" : "+v"(cand[j]) : "v"(np[b][j]), "v"(zm)); // cand & ~(plane ^ zm)
#if PFX_MB2_KSEL
                k = __builtin_amdgcn_bitop3_b32(k, d, zm, 0xe4);   // zm ? k : d
#else
                k = min(k, d);
#endif
            }
            res = __builtin_amdgcn_bitop3_b32(res, zm, 1u << b, 0xf2); // res | (~zm & bit)
        }
        return res;
    };
    const mb_v4i rs_dst = mb_rsrc(dst, (uint32_t)w * (uint32_t)h * 4u);
    const mb_v4i rs_src = mb_rsrc(src, (uint32_t)w * (uint32_t)h * 4u);
    const mb_v4i rs_msk = mb_rsrc(mask, MASK ? (uint32_t)w * (uint32_t)h : 0u);
    // byte offsets inside the image row; a column past the right edge gets an offset no buffer holds (the store is dropped)
    const int vo0 = xa < w ? xa * 4 + (int)c : (int)0x7fffffff, vo1 = xa + 1 < w ? (xa + 1) * 4 + (int)c : (int)0x7fffffff;
    auto emit = [&](int y, uint32_t win, uint32_t res) __attribute__((always_inline)) {
        const int vo = win ? vo1 : vo0;
        uint8_t o = (uint8_t)res;
        if constexpr (MASK) {
            const uint8_t m = mb_buffer_load_i8(rs_msk, vo >> 2, y * w, 0), sv = mb_buffer_load_i8(rs_src, vo, y * w * 4, 0);
            if (m == 0) o = sv;
        }
        mb_buffer_store_i8(o, rs_dst, vo, y * w * 4, 0);
    };

    static_for<0, S - 1>([&](auto sc) __attribute__((always_inline)) {
        fetch((int)decltype(sc)::value);
        insert(sc);
    });
    for (int y = y0; y < y1; y += S) {
        bool done = false;
        static_for<0, S>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            if (!done && y + u < y1) {
                fetch(y - y0 + u + 2 * R);
                insert(std::integral_constant<int, (S - 1 + u) % S>{});
#pragma unroll 1
                for (uint32_t win = 0; win < 2u; ++win) emit(y + u, win, select(win));
            } else done = true;
        });
    }
}

int g_mb_pair = 1; // pfxk_median_bits_set_pair: radii 2..7 on the column-pair kernel
} // namespace

extern "C" void pfxk_median_bits_set_pair(int on) { g_mb_pair = on; }

extern "C" size_t pfxk_median_bits_scratch(int radius, uint32_t w, uint32_t h)
{
    return (size_t)h * 4u * mb_dwords(w, radius) * 8u * sizeof(uint32_t);
}

extern "C" hipError_t pfxk_median_bits(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, uint32_t* d_planes, int radius,
                                       uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius < 2 || radius > 8) return hipErrorInvalidValue;
    const uint32_t nd = mb_dwords(w, radius);
    median_planes_kernel<<<dim3((nd + 7u) / 8u, (h + MP_ROWS - 1) / MP_ROWS), 256, 0, s>>>((const uint32_t*)d_src, d_planes, radius, (int)w, (int)h, nd);
    if (g_mb_pair && radius <= 7) {
        const dim3 g2((w + MB2_COLS - 1) / MB2_COLS, (h + PFX_MB2_ROWS - 1) / PFX_MB2_ROWS);
#define PFX_MB2(R) case R: if (d_mask) median_bits2_kernel<R, true><<<g2, 256, 0, s>>>(d_src, d_planes, d_dst, d_mask, (int)w, (int)h, nd); \
                           else median_bits2_kernel<R, false><<<g2, 256, 0, s>>>(d_src, d_planes, d_dst, d_mask, (int)w, (int)h, nd); break;
        switch (radius) { PFX_MB2(2) PFX_MB2(3) PFX_MB2(4) PFX_MB2(5) PFX_MB2(6) PFX_MB2(7) }
#undef PFX_MB2
        return hipGetLastError();
    }
    const dim3 g((w + 4 * MB_COLS - 1) / (4 * MB_COLS), (h + MB_ROWS - 1) / MB_ROWS);
#define PFX_MB(R) case R: median_bits_kernel<R><<<g, 256, 0, s>>>(d_src, d_planes, d_dst, d_mask, (int)w, (int)h, nd); break;
    switch (radius) { PFX_MB(2) PFX_MB(3) PFX_MB(4) PFX_MB(5) PFX_MB(6) PFX_MB(7) PFX_MB(8) }
#undef PFX_MB
    return hipGetLastError();
}
