// k_gauss_exact.hip — the bit-exact Gaussian (pfx_ctx_set_exact; filters.rs:214-316 with separately rounded products and sums) at small radii: both passes
// in one kernel.  A file of its own because it is built WITHOUT the SLP vectoriser (Makefile): packed v_pk_mul_f32 / v_pk_add_f32 run at half rate on gfx950 and
// this kernel is nothing but multiplies and adds; k_gauss.hip's matrix-core kernel keeps SLP.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include <algorithm>
#include "k_common.h"
#include "pfx_kernels.h"
#include "k_pointwise.h"

using namespace pfxk;

namespace {

// An empty volatile asm that "rewrites" the sixteen accumulators and clobbers memory: arithmetic feeding them cannot sink below it, loads cannot rise above it.
PFX_DEV void gf_pin(float4 (&a)[4])
{
    asm volatile("" : "+v"(a[0].x), "+v"(a[0].y), "+v"(a[0].z), "+v"(a[0].w), "+v"(a[1].x), "+v"(a[1].y), "+v"(a[1].z), "+v"(a[1].w),
                      "+v"(a[2].x), "+v"(a[2].y), "+v"(a[2].z), "+v"(a[2].w), "+v"(a[3].x), "+v"(a[3].y), "+v"(a[3].z), "+v"(a[3].w) :: "memory");
}
PFX_DEV void mac4x(float4& acc, const float4 p, const float wv)   // separate rounding of the product and the sum, like the reference (k_gauss.hip: mac4<true>)
{
    acc.x = acc.x + p.x * wv; acc.y = acc.y + p.y * wv; acc.z = acc.z + p.z * wv; acc.w = acc.w + p.w * wv;
}

// ---- bit-exact mode, radii 1 .. GF_MAXR: both passes in one kernel (round 5) ----
// Since round 5 the composite effects (sharpen, glow, drop shadow) and the batch pipeline run the bit-exact Gaussian by default, and at their radii the two
// kernels above are bound by the f32 x 4 intermediate they exchange through HBM (4 + 16 bytes per pixel and pass: 0.40 ms at 8K whatever the radius).  Here a
// workgroup owns GF_W output columns of a segment of rows and walks down GF_RB rows at a time (the shape of k_stencil.hip's box_strip_kernel): the block's source
// rows are staged in LDS as bytes, the horizontal pass writes f32 x 4 rows into an LDS RING of 2r + 1 + GF_RB rows, the vertical pass reads its taps from the
// ring.  Every output's sums run over the taps in the reference's order with separately rounded products and sums (mac4<true>; the register blocking pads with
// zero-weight taps, which add +0.0): bit-identical to k_gauss.hip's gauss_h_kernel<true> + gauss_v_kernel<true> and to filters.rs:214-316.  HBM traffic: the 8 algorithmic
// bytes per pixel plus the strips' halo columns (L2 hits: neighbouring strips share an XCD) and 2r rows of run-in per segment.
constexpr int GF_W = 64, GF_RB = 32, GF_T = 512, GF_MAXR = 16, GF_XPAD = 8;   // 8 waves per workgroup: twice the waves per byte of LDS ring of a 16-row block
PFX_DEV int gf_swz(int x) { return (x & 3) * 16 + (x >> 2); }   // ring position of column x: a lane's four outputs land 16 slots apart, so the 16-byte stores of a wave are contiguous
// The vertical pass's lane -> column map is the swizzle's inverse, so that lane i reads ring position i: a wave's sixteen-byte reads of a ring row are 1 KB in lane
// order and each quarter-wave (what the LDS serves per pass at this width) covers all 64 banks once.  With column = lane, lanes 0 .. 3 read positions 0, 16, 32, 48 —
// 256 bytes apart, the same four banks: 64 % of this kernel's LDS cycles were bank conflicts (rocprofv3 SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, profiles/r06_tuning.md).
// A wave still writes 64 adjacent pixels of an output row, permuted among its lanes.
#ifndef PFX_GF_VMAP
#define PFX_GF_VMAP 1
#endif
PFX_DEV int gf_vcol(int lane) { return PFX_GF_VMAP ? ((lane & 15) * 4 + (lane >> 4)) : lane; }   // gf_swz(gf_vcol(i)) == i
PFX_DEV float4 gf_px(uint32_t px) { return make_float4(ubyte0(px), ubyte1(px), ubyte2(px), ubyte3(px)); }
__host__ __device__ constexpr int gf_src_pitch(int r) { const int n = GF_W + 2 * r + GF_XPAD; return n + ((5 - (n & 3)) & 3); }   // = 1 (mod 4): the four rows of a wave start in different banks
// R is a template parameter: the tap loops are straight-line code over exactly the 2R + 1 taps of each of a lane's four outputs (the generic kernels' register
// blocking pads every output to a multiple of four taps with zero weights: 40 % of the multiply-adds at R = 3), the weights sit in scalar registers.
// EPI: what leaves the vertical pass — 0 the blurred pixel; 1 / 2 the sharpen / glow combine of the SOURCE pixel with it (stylize.rs:135-137 / :60-64, k_effects.hip's
// combine_kernel word for word: the blurred value enters rounded to u8, as it would from a buffer; alpha from the source; an unselected pixel keeps the source)
// EPI 3 (round 6, pfx_chain_dev): the blurred pixel goes through a chain of pointwise ops before it is stored (k_pointwise.h: chain_apply on the rounded u8 pixel —
// what the ops would read from a buffer), so `Gaussian -> HSL -> ...` is one launch and the blurred image never exists in memory; the chain's tables sit behind the
// staged source rows in LDS.
struct gf_no_chain {};
template <int R, int EPI, class CHAIN = gf_no_chain>
__global__ __launch_bounds__(GF_T) void gauss_fused_exact_kernel(const CHAIN chain_arg /* first: the chain is read through the kernarg segment (k_pointwise.h) */,
                                                                const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const float* __restrict__ wts,
                                                                int w, int h, int seg_rows, int nseg, int strips, const uint8_t* __restrict__ mask, float p0,
                                                                const uint8_t* __restrict__ luts_g)
{
    const pw::chain_kptr chain = pw::chain_in_kernarg();   // meaningful for EPI 3 only
    extern __shared__ __attribute__((aligned(16))) uint8_t gf_lds[];
    const int tid = (int)threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);   // XCD k owns a contiguous group of strips (block b runs on XCD b % 8)
    const int s_lo = strips * xcd / 8, s_hi = strips * (xcd + 1) / 8, sg = s_hi - s_lo;
    if (sg == 0 || j >= sg * nseg) return;
    const int seg = j / sg, strip = s_lo + (j - seg * sg);
    const int x0 = strip * GF_W, y0 = seg * seg_rows, y1 = min(y0 + seg_rows, h);
    if (y0 >= h) return;
    constexpr int r = R, RR = 2 * R + 1 + GF_RB, KLEN = 2 * R + 1, NIN = KLEN + 3, NG = (NIN + 3) / 4, sp = gf_src_pitch(R), n_in = GF_W + 2 * R + GF_XPAD;
    float4* const ring = reinterpret_cast<float4*>(gf_lds);                              // [RR][GF_W]: intermediate row v lives in slot (v - v_begin) mod RR
    uint32_t* const s_src = reinterpret_cast<uint32_t*>(gf_lds + (size_t)RR * GF_W * 16);  // [GF_RB][sp]
    const int v_begin = y0 - r, v_end = y1 + r;
    const uint8_t* const s_luts = gf_lds + (size_t)RR * GF_W * 16 + (size_t)GF_RB * sp * 4;   // EPI 3: the chain's tables (n_luts x 1024 bytes)
    if constexpr (EPI == 3) {
        for (uint32_t i = (uint32_t)tid; i < chain->n_luts * 256u; i += (uint32_t)GF_T)
            reinterpret_cast<uint32_t*>(gf_lds + (size_t)RR * GF_W * 16 + (size_t)GF_RB * sp * 4)[i] = reinterpret_cast<const uint32_t*>(luts_g)[i];
        // the first barrier of the block loop below orders these stores before any vertical pass reads them
    }
    float wt[KLEN];
#pragma unroll
    for (int k = 0; k < KLEN; ++k) wt[k] = wts[k];                                        // uniform: scalar registers
    // horizontal role: lane = (row of the block, run of 4 outputs)
    const int hrow = tid >> 4, hrun = tid & 15;
    // vertical role: lane = (column, group of 4 output rows); the group index is the wave index, so ring rows are wave-uniform
    const int vlane = tid & 63, col = gf_vcol(vlane), vg = __builtin_amdgcn_readfirstlane(tid >> 6), x = x0 + col, scol = gf_swz(col);
    // (1) staging of GF_RB source rows x n_in columns (clamp-to-edge, filters.rs:268-270 / 296-298): a block's rows are requested under the previous block's
    // vertical pass and stored behind it
    constexpr int PER = (GF_RB * n_in + GF_T - 1) / GF_T;
    uint32_t stg[PER];
    auto stage_load = [&](int vb) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = min(tid + GF_T * k, GF_RB * n_in - 1), ty = i / n_in, tx = i - ty * n_in;
            stg[k] = src[(size_t)min(max(vb + ty, 0), h - 1) * w + min(max(x0 - r + tx, 0), w - 1)];
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + GF_T * k, ty = i / n_in, tx = i - ty * n_in;
            if (i < GF_RB * n_in) s_src[ty * sp + tx] = stg[k];
        }
    };
    stage_load(v_begin);
    stage_store();
    int base = 0;
    for (int vb = v_begin; vb < v_end; vb += GF_RB) {
        __syncthreads();
        {   // (2) horizontal: input i (relative to the lane's first output) is tap i - o of output o; taps in the reference's order (filters.rs:262-276)
            const uint32_t* p = s_src + hrow * sp + 4 * hrun;
            float4 acc[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
            // inputs in groups of four, the next group's LDS reads issued before the current group's arithmetic and nothing else moved across (an unrolled
            // loop left to the scheduler hoists every read: 186 registers at R = 12)
            uint32_t cur[4], nxt[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = p[k];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) nxt[k] = p[4 * (g + 1) + k];   // (the staged row holds GF_XPAD spare columns: reads past NIN stay inside it)
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 4 * g + k;
                    if (i < NIN) {
                        const float4 a = gf_px(cur[k]);
#pragma unroll
                        for (int o = 0; o < 4; ++o)
                            if (i - o >= 0 && i - o < KLEN) mac4x(acc[o], a, wt[i - o]);
                    }
                }
                gf_pin(acc);   // this group's arithmetic ends here and the next-but-one group's LDS reads start behind it: neither optimiser nor scheduler may regroup them
#pragma unroll
                for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
            }
            int slot = base + hrow; slot = slot >= RR ? slot - RR : slot;
            float4* out = ring + slot * GF_W + hrun;    // gf_swz(4 hrun + o) = 16 o + hrun
#pragma unroll
            for (int o = 0; o < 4; ++o) out[16 * o] = acc[o];
        }
        __syncthreads();
        const bool more = vb + GF_RB < v_end;
        if (more) stage_load(vb + GF_RB);
        {   // (3) vertical: this block completes output rows [vb - r, vb - r + GF_RB); the lane's four are yo .. yo + 3, their taps ring rows yo - r .. yo + 3 + r
            const int yo = vb - r + 4 * vg;
            if (yo + 3 >= y0 && yo < y1 && x < w) {
                int slot = base + 4 * vg - 2 * r;                                          // ring slot of row yo - r
                slot = slot < 0 ? slot + RR : (slot >= RR ? slot - RR : slot);
                float4 acc[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 cur[4], nxt[4];
                auto rows = [&](float4 (&d)[4], int first) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (first + k < NIN) { d[k] = ring[slot * GF_W + scol]; slot = slot + 1 >= RR ? 0 : slot + 1; }
                    }
                };
                rows(cur, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) rows(nxt, 4 * (g + 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * g + k;
                        if (i < NIN) {
#pragma unroll
                            for (int o = 0; o < 4; ++o)
                                if (i - o >= 0 && i - o < KLEN) mac4x(acc[o], cur[k], wt[i - o]);
                        }
                    }
                    gf_pin(acc);   // this group's arithmetic ends here and the next-but-one group's LDS reads start behind it: neither optimiser nor scheduler may regroup them
#pragma unroll
                    for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
                }
                if constexpr (EPI == 3) {
                    uint32_t b4[4];
#pragma unroll
                    for (int o = 0; o < 4; ++o) b4[o] = pack_round_rgba(acc[o].x, acc[o].y, acc[o].z, acc[o].w); // filters.rs:308-311
                    pw::chain_apply<4>(chain, s_luts, b4);
#pragma unroll
                    for (int o = 0; o < 4; ++o)
                        if (yo + o >= y0 && yo + o < y1) dst[(size_t)(yo + o) * w + x] = b4[o];
                } else
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (yo + o >= y0 && yo + o < y1) {
                        const size_t oi = (size_t)(yo + o) * w + x;
                        const uint32_t b = pack_round_rgba(acc[o].x, acc[o].y, acc[o].z, acc[o].w); // filters.rs:308-311
                        if constexpr (EPI == 0) dst[oi] = b;
                        else {
                            const uint32_t sp = src[oi];
                            uint32_t px = sp;
                            if (!(mask && mask[oi] == 0)) {
                                float e[3];
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    const float sv = (float)((sp >> (8 * c)) & 0xffu), bv = (float)((b >> (8 * c)) & 0xffu);
                                    if constexpr (EPI == 1) e[c] = sv + p0 * (sv - bv);
                                    else { const float sn = div255(sv), bn = div255(bv); e[c] = (1.0f - (1.0f - sn) * (1.0f - bn * p0)) * 255.0f; }
                                }
                                px = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(e[0]), 0, sp);
                                px = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(e[1]), 1, px);
                                px = __builtin_amdgcn_cvt_pk_u8_f32(round_tie_prep(e[2]), 2, px);
                            }
                            dst[oi] = px;
                        }
                    }
            }
        }
        if (more) stage_store();   // s_src was last read before the barrier above; the next horizontal pass waits for the barrier at the loop's top
        base += GF_RB; base = base >= RR ? base - RR : base;
        // the next block's ring writes come behind that barrier too, and land in slots this vertical pass does not read (RR = 2r + 1 + GF_RB)
    }
}


// ---- the same for a ONE-CHANNEL image (round 6): what the drop shadow blurs (render.rs:291-301 expands its alpha plane to (a, a, a, a) and blurs all four channels; only
// one is read back).  A "pixel" of the walk is a QUAD of four adjacent plane bytes: the vertical pass is the kernel above's with the four columns in the four float lanes,
// the horizontal pass gives a lane sixteen adjacent output columns (four quads) whose taps come from 16 + 2R staged bytes — per output the same products and sums in the same
// order as one channel of the RGBA kernel (bit-identical: tests/test_gpu_effects.py compares the shadow with the oracle at tolerance 0), a quarter of the arithmetic and of the
// traffic.  The plane's width must be a multiple of four (rows are then dword-aligned and a quad never crosses a row).
template <int R>
__global__ __launch_bounds__(GF_T) void gauss_plane_exact_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const float* __restrict__ wts, int w, int h,
                                                                 int seg_rows, int nseg, int strips)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t gf_lds[];
    const int tid = (int)threadIdx.x;
    const int xcd = (int)(blockIdx.x & 7u), j = (int)(blockIdx.x >> 3);
    const int s_lo = strips * xcd / 8, s_hi = strips * (xcd + 1) / 8, sg = s_hi - s_lo;
    if (sg == 0 || j >= sg * nseg) return;
    const int seg = j / sg, strip = s_lo + (j - seg * sg);
    const int wq = w >> 2;                                      // quads per row
    const int xq0 = strip * GF_W, y0 = seg * seg_rows, y1 = min(y0 + seg_rows, h);
    if (y0 >= h) return;
    constexpr int r = R, RR = 2 * R + 1 + GF_RB, KLEN = 2 * R + 1, R4 = (R + 3) & ~3, OFF = R4 - R;   // the staged row starts R4 bytes (whole dwords) left of the strip
    constexpr int ND = (4 * GF_W + 2 * R4) / 4, SPD = ND + 1 + ((4 - ((ND + 1) & 3)) & 3) + 1;        // staged dwords per row; pitch = 1 (mod 4): rows start in different banks
    float4* const ring = reinterpret_cast<float4*>(gf_lds);                                // [RR][GF_W] quads
    uint32_t* const s_src = reinterpret_cast<uint32_t*>(gf_lds + (size_t)RR * GF_W * 16);    // [GF_RB][SPD]
    const int v_begin = y0 - r, v_end = y1 + r;
    float wt[KLEN];
#pragma unroll
    for (int k = 0; k < KLEN; ++k) wt[k] = wts[k];
    const int hrow = tid >> 4, hrun = tid & 15;
    const int vlane = tid & 63, col = gf_vcol(vlane), vg = __builtin_amdgcn_readfirstlane(tid >> 6), xq = xq0 + col, scol = gf_swz(col);
    constexpr int PER = (GF_RB * ND + GF_T - 1) / GF_T;
    uint32_t stg[PER];
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(src);
    auto stage_load = [&](int vb) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = min(tid + GF_T * k, GF_RB * ND - 1), ty = i / ND, td = i - ty * ND;
            const int yy = min(max(vb + ty, 0), h - 1), xb = 4 * xq0 - R4 + 4 * td;         // first byte column of this dword
            if (xb >= 0 && xb + 4 <= w) stg[k] = src32[((size_t)yy * w + xb) >> 2];           // wholly inside the row
            else {                                                                          // clamp-to-edge per byte (filters.rs:268-270)
                const uint8_t* row = src + (size_t)yy * w;
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) v |= (uint32_t)row[min(max(xb + b, 0), w - 1)] << (8 * b);
                stg[k] = v;
            }
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = tid + GF_T * k, ty = i / ND, td = i - ty * ND;
            if (i < GF_RB * ND) s_src[ty * SPD + td] = stg[k];
        }
    };
    stage_load(v_begin);
    stage_store();
    int base = 0;
    for (int vb = v_begin; vb < v_end; vb += GF_RB) {
        __syncthreads();
        {   // horizontal: the lane's outputs are byte columns 16 hrun .. 16 hrun + 15 of the strip; input byte i (relative to output 0's leftmost tap) is tap i - o of output o
            const uint32_t* p = s_src + hrow * SPD + 4 * hrun;                                // dword holding staged byte 16 hrun (= output 0's column - R4)
            float acc[16];
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = 0.0f;
            constexpr int NIN = 16 + 2 * R, NDW = (OFF + NIN + 3) / 4;
#pragma unroll
            for (int d = 0; d < NDW; ++d) {
                const uint32_t v = p[d];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int i = 4 * d + b - OFF;
                    if (i >= 0 && i < NIN) {
                        const float a = (float)((v >> (8 * b)) & 0xffu);
#pragma unroll
                        for (int o = 0; o < 16; ++o)
                            if (i - o >= 0 && i - o < KLEN) acc[o] = acc[o] + a * wt[i - o];
                    }
                }
                if ((d & 3) == 3) {   // keep the loop from being reordered into one huge block (register pressure): every 16 input bytes the accumulators are pinned
#pragma unroll
                    for (int o = 0; o < 16; ++o) asm volatile("" : "+v"(acc[o]));
                }
            }
            int slot = base + hrow; slot = slot >= RR ? slot - RR : slot;
            float4* out = ring + slot * GF_W + hrun;
#pragma unroll
            for (int q = 0; q < 4; ++q) out[16 * q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
        __syncthreads();
        const bool more = vb + GF_RB < v_end;
        if (more) stage_load(vb + GF_RB);
        {   // vertical: the RGBA kernel's pass, a quad's four columns in the four lanes of a float4
            const int yo = vb - r + 4 * vg;
            if (yo + 3 >= y0 && yo < y1 && xq < wq) {
                int slot = base + 4 * vg - 2 * r;
                slot = slot < 0 ? slot + RR : (slot >= RR ? slot - RR : slot);
                constexpr int NIN = KLEN + 3, NG = (NIN + 3) / 4;
                float4 acc[4];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
                float4 cur[4], nxt[4];
                auto rows = [&](float4 (&d)[4], int first) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (first + k < NIN) { d[k] = ring[slot * GF_W + scol]; slot = slot + 1 >= RR ? 0 : slot + 1; }
                    }
                };
                rows(cur, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) rows(nxt, 4 * (g + 1));
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * g + k;
                        if (i < NIN) {
#pragma unroll
                            for (int o = 0; o < 4; ++o)
                                if (i - o >= 0 && i - o < KLEN) mac4x(acc[o], cur[k], wt[i - o]);
                        }
                    }
                    gf_pin(acc);
#pragma unroll
                    for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
                }
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (yo + o >= y0 && yo + o < y1)
                        reinterpret_cast<uint32_t*>(dst)[((size_t)(yo + o) * w >> 2) + xq] = pack_round_rgba(acc[o].x, acc[o].y, acc[o].z, acc[o].w);
            }
        }
        if (more) stage_store();
        base += GF_RB; base = base >= RR ? base - RR : base;
    }
}

} // namespace

int g_fused_exact = 1; // pfxk_gauss_set_fused_exact (pfx_tune "gauss_fused_exact"): 0 = the bit-exact mode always through the two kernels (A/B, parity tests)
extern "C" void pfxk_gauss_set_fused_exact(int on) { g_fused_exact = on; }
extern "C" int pfxk_gauss_fused_exact_max_radius(void) { return g_fused_exact ? GF_MAXR : 0; }
static hipError_t launch_fused_exact(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h,
                                     int epilogue /* 0 blur, 1 sharpen, 2 glow, 3 chain */, float p0, const uint8_t* d_mask, const pfxk_chain* chain, const uint8_t* d_luts)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius < 1 || radius > GF_MAXR) return hipErrorInvalidValue;
    if (epilogue == 3 && (!chain || chain->n > PFXK_CHAIN_MAX || chain->n_luts > PFXK_CHAIN_LUTS || (chain->n_luts && !d_luts))) return hipErrorInvalidValue;
    const int RR = 2 * radius + 1 + GF_RB;
    const size_t lds = (size_t)RR * GF_W * 16 + (size_t)GF_RB * gf_src_pitch(radius) * 4 + (epilogue == 3 ? (size_t)chain->n_luts * 1024 : 0);
    const int strips = (int)((w + GF_W - 1) / GF_W);
    // one round of workgroups per XCD (32 CUs x what the LDS footprint allows), segments of at least 8r + 64 rows (the 2r rows of run-in stay small)
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / lds));
    const int sg_max = (strips + 7) / 8;
    int nseg = std::max(1, 32 * wg_per_cu / sg_max);
    nseg = std::min(nseg, std::max(1, (int)h / (8 * radius + 64)));
    const int seg_rows = ((int)h + nseg - 1) / nseg;
    nseg = ((int)h + seg_rows - 1) / seg_rows;
    const dim3 grid(8u * (uint32_t)(sg_max * nseg));
    auto go2 = [&](auto rc, auto ec) -> hipError_t {
        constexpr int R = decltype(rc)::value, E = decltype(ec)::value;
        static lds_grant grant;   // one per (R, E) instantiation
        if constexpr (E == 3) {
            hipError_t e = grant_lds(grant, (const void*)gauss_fused_exact_kernel<R, 3, pfxk_chain>, lds);
            if (e) return e;
            gauss_fused_exact_kernel<R, 3, pfxk_chain><<<grid, GF_T, lds, stream>>>(*chain, (const uint32_t*)d_src, (uint32_t*)d_dst, d_wts_tap0, (int)w, (int)h, seg_rows, nseg, strips,
                                                                                    nullptr, 0.0f, d_luts);
        } else {
            hipError_t e = grant_lds(grant, (const void*)gauss_fused_exact_kernel<R, E>, lds);
            if (e) return e;
            gauss_fused_exact_kernel<R, E><<<grid, GF_T, lds, stream>>>(gf_no_chain{}, (const uint32_t*)d_src, (uint32_t*)d_dst, d_wts_tap0, (int)w, (int)h, seg_rows, nseg, strips, d_mask, p0,
                                                                        nullptr);
        }
        return hipGetLastError();
    };
    auto go = [&](auto rc) -> hipError_t {
        if (epilogue == 1) return go2(rc, std::integral_constant<int, 1>{});
        if (epilogue == 2) return go2(rc, std::integral_constant<int, 2>{});
        if (epilogue == 3) return go2(rc, std::integral_constant<int, 3>{});
        return go2(rc, std::integral_constant<int, 0>{});
    };
    static_assert(GF_MAXR == 16, "gauss_fused_exact_kernel is instantiated for radii 1 .. 16");
    switch (radius) {
#define PFX_GF(R) case R: return go(std::integral_constant<int, R>{});
        PFX_GF(1) PFX_GF(2) PFX_GF(3) PFX_GF(4) PFX_GF(5) PFX_GF(6) PFX_GF(7) PFX_GF(8) PFX_GF(9) PFX_GF(10) PFX_GF(11) PFX_GF(12) PFX_GF(13) PFX_GF(14) PFX_GF(15) PFX_GF(16)
#undef PFX_GF
    default: return hipErrorInvalidValue;
    }
}
extern "C" hipError_t pfxk_gauss_fused_exact(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h,
                                             int epilogue /* 0 blur, 1 sharpen, 2 glow */, float p0, const uint8_t* d_mask)
{
    if (epilogue < 0 || epilogue > 2) return hipErrorInvalidValue;
    return launch_fused_exact(stream, d_src, d_dst, d_wts_tap0, radius, w, h, epilogue, p0, d_mask, nullptr, nullptr);
}
extern "C" hipError_t pfxk_gauss_fused_exact_chain(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h,
                                                   const pfxk_chain* C, const uint8_t* d_luts)
{
    return launch_fused_exact(stream, d_src, d_dst, d_wts_tap0, radius, w, h, 3, 0.0f, nullptr, C, d_luts);
}

// one-channel form (the drop shadow's alpha plane): w % 4 == 0, tight rows; bit-identical, element by element, to one channel of pfxk_gauss_fused_exact on (a, a, a, a)
extern "C" hipError_t pfxk_gauss_plane_exact(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0) return hipSuccess;
    if (radius < 1 || radius > GF_MAXR || (w & 3u) != 0 || ((uintptr_t)d_src & 3u) || ((uintptr_t)d_dst & 3u)) return hipErrorInvalidValue;
    const int RR = 2 * radius + 1 + GF_RB, R4 = (radius + 3) & ~3, ND = (4 * GF_W + 2 * R4) / 4, SPD = ND + 1 + ((4 - ((ND + 1) & 3)) & 3) + 1;
    const size_t lds = (size_t)RR * GF_W * 16 + (size_t)GF_RB * SPD * 4;
    const int wq = (int)(w >> 2), strips = (wq + GF_W - 1) / GF_W;
    const int wg_per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / lds));
    const int sg_max = (strips + 7) / 8;
    int nseg = std::max(1, 32 * wg_per_cu / sg_max);
    nseg = std::min(nseg, std::max(1, (int)h / (8 * radius + 64)));
    const int seg_rows = ((int)h + nseg - 1) / nseg;
    nseg = ((int)h + seg_rows - 1) / seg_rows;
    const dim3 grid(8u * (uint32_t)(sg_max * nseg));
    auto go = [&](auto rc) -> hipError_t {
        constexpr int R = decltype(rc)::value;
        static lds_grant grant;
        hipError_t e = grant_lds(grant, (const void*)gauss_plane_exact_kernel<R>, lds);
        if (e) return e;
        gauss_plane_exact_kernel<R><<<grid, GF_T, lds, stream>>>(d_src, d_dst, d_wts_tap0, (int)w, (int)h, seg_rows, nseg, strips);
        return hipGetLastError();
    };
    switch (radius) {
#define PFX_GP(R) case R: return go(std::integral_constant<int, R>{});
        PFX_GP(1) PFX_GP(2) PFX_GP(3) PFX_GP(4) PFX_GP(5) PFX_GP(6) PFX_GP(7) PFX_GP(8) PFX_GP(9) PFX_GP(10) PFX_GP(11) PFX_GP(12) PFX_GP(13) PFX_GP(14) PFX_GP(15) PFX_GP(16)
#undef PFX_GP
    default: return hipErrorInvalidValue;
    }
}
