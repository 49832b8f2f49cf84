// k_warp.hip — Liquify displacement warp and Catmull-Rom mesh warp (gather kernels, +-1 LSB class; in practice
// bit-exact because the f32 expressions are evaluated in the reference's order without FMA contraction).
//
// Reference: warp_displacement_full src/ops/transform.rs:1288-1345 (bilinear, texels outside the source are 0,
//            output left transparent when floor(sx) < -1 || floor(sy) < -1 || >= size);
//            catmull_rom_weights :1558-1567, catmull_rom_surface :1589-1646,
//            generate_displacement_from_mesh :1670-1705 / _fast :1712-1740, warp_mesh_catmull_rom :1743-1761.
// Design: one lane per output pixel, a wave covers 64 consecutive x of one row, so the four bilinear taps of
// neighbouring lanes fall in the same or adjacent 128-byte lines for any smooth field (L1/L2-served gather; the
// source is read roughly once from HBM).  The fused mesh warp evaluates the two bicubic surfaces in registers from
// control points staged in LDS and never materialises the 8 B/px displacement field.
#include "k_common.h"
#include <type_traits>
#include "pfx_kernels.h"

using namespace pfxk;

// LLVM buffer intrinsic hipcc has no __builtin for (declared outside the anonymous namespace: an external symbol)
typedef int pfx_w_v4i __attribute__((ext_vector_type(4)));
typedef int pfx_w_v2i __attribute__((ext_vector_type(2)));
__device__ pfx_w_v2i pfx_w_buffer_load_v2i32(pfx_w_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");

namespace {

// Source images below 4 GiB are sampled through a raw buffer resource and 32-bit byte offsets: the two 64-bit address computations of a pixel's row
// pairs (v_mad_u64_u32, v_lshl_add_u64 x 2, ...) become one 24-bit multiply-add and a shift each — the fused mesh warp is bound by VALU issue.
struct warp_src { const uint32_t* p; pfx_w_v4i rs; };
PFX_DEV warp_src make_warp_src(const uint32_t* src, int32_t src_w, int32_t src_h)
{
    const uint64_t a = (uint64_t)src;
    warp_src S;
    S.p = src;
    S.rs.x = (int)(uint32_t)a;
    S.rs.y = (int)((uint32_t)(a >> 32) & 0xffffu);
    S.rs.z = (int)((uint32_t)src_w * (uint32_t)src_h * 4u);
    S.rs.w = (int)(0xFACu | (7u << 12) | (4u << 15));   // untyped dword access, stride 0: offsets and num_records are bytes
    return S;
}

// warp_displacement_full's sampler (:1288-1345: bilinear, lerp form a + (b - a) * t, texels outside the source are 0, output
// transparent when floor(sx) < -1 || floor(sy) < -1 || >= size), split at the memory boundary so that a lane can have the taps of
// several pixels in flight at once, and trimmed where the hardware or the value range makes a step free:
//  * v_cvt_i32_f32 saturates and maps NaN to 0 — exactly Rust's `as i32` — so the explicit range tests go (inline asm keeps
//    the compiler from treating an out-of-range conversion as undefined);
//  * when every lane of the wave samples strictly inside the source, the four per-tap bounds tests (and their exec-mask
//    branches) are skipped: one wave-uniform branch instead;
//  * the lerp of bytes with weights in [0, 1) is finite, so `.round().clamp(0, 255) as u8` is k_common.h's 2-instruction-per-channel form.
PFX_DEV int32_t cvt_i32_sat(float v)
{
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// Round 4: the four loads sit in straight-line code.  With the taps behind the interior / border branch (round 3) hipcc's wait-count bookkeeping gave
// up at the join and put s_waitcnt vmcnt(0) in front of every later pixel's address arithmetic — the eight rows of a batch became eight dependent
// round trips.  Now every tap is loaded from a clamped (always valid) address and the texels that lie outside are zeroed afterwards from a 4-bit
// mask; the wave-uniform interior test only skips the clamps and the mask (no load behind a branch, the waits are counted exactly).
struct bilinear_taps { uint32_t tl, tr, bl, br; float fx, fy; uint32_t m; bool all_in; }; // m: bit 0 tl, 1 tr, 2 bl, 3 br inside the source (0: output transparent); bits 4 / 5: pair loads at the left / right border
// all_in: every lane of the wave samples strictly inside the source (wave-uniform, lives in a scalar register): m is then not even materialised
// PAIR (compile-time, chosen by the launcher: src_w >= 2): a run-time test here would put the loads behind a branch again
template <bool PAIR, bool BUF32 = false>
PFX_DEV bilinear_taps bilinear_fetch(const warp_src& S, int32_t src_w, int32_t src_h, float x, float y, float ddx, float ddy)
{
    const uint32_t* __restrict__ src = S.p;
    bilinear_taps T;
    const float sx = x - ddx, sy = y - ddy;
    const int32_t x0 = cvt_i32_sat(__builtin_floorf(sx)), y0 = cvt_i32_sat(__builtin_floorf(sy));
    T.fx = sx - (float)x0;
    T.fy = sy - (float)y0;
    // strictly inside: 0 <= x0 < w - 1 as ONE unsigned compare per axis; a NaN coordinate converts to texel 0 with NaN weights, so such a lane must not
    // count as interior (its wave would skip the test that makes it transparent): one unordered compare covers both coordinates (an infinite one
    // saturates x0 / y0 and fails the range test)
    // as lane masks straight from the compares (hip's __all() and the ballot builtin first materialise the predicate in a VGPR): 36 = unsigned <, 7 = ordered
    const uint64_t interior = __builtin_amdgcn_uicmp((uint32_t)x0, (uint32_t)(src_w - 1), 36) & __builtin_amdgcn_uicmp((uint32_t)y0, (uint32_t)(src_h - 1), 36) &
                              __builtin_amdgcn_fcmpf(sx, sy, 7);
    T.all_in = interior == __builtin_amdgcn_ballot_w64(true);
    int32_t xa = x0, xb = x0 + 1, ya = y0, yb = y0 + 1;
    int32_t xp = x0;                         // PAIR: first texel of the 8-byte pair
    if (!T.all_in) {
        // :1310; a NaN coordinate converts to texel 0 with NaN weights: every channel would be `NaN as u8` = 0, the same as "outside"
        const bool ok = !(x0 < -1 || y0 < -1 || x0 >= src_w || y0 >= src_h) && T.fx == T.fx && T.fy == T.fy;
        const uint32_t mx = (x0 >= 0 ? 5u : 0u) | (x0 + 1 < src_w ? 10u : 0u), my = (y0 >= 0 ? 3u : 0u) | (y0 + 1 < src_h ? 12u : 0u);
        T.m = ok ? (mx & my) : 0u;
        if (!ok) { T.fx = 0.0f; T.fy = 0.0f; }   // all four texels read as 0: the lerp of zeros with finite weights is the transparent pixel
        xa = min(max(xa, 0), src_w - 1); xb = min(max(xb, 0), src_w - 1);
        ya = min(max(ya, 0), src_h - 1); yb = min(max(yb, 0), src_h - 1);
        if constexpr (PAIR) {
            // the pair starts at clamp(x0, 0, w - 2); at the left / right border the texel that exists sits in the other half of the pair (the one
            // that does not is masked); which half is settled in bilinear_finish (bits 4 / 5 of m): nothing here waits for the loads
            xp = min(max(x0, 0), src_w - 2);
            T.m |= (x0 < xp ? 16u : 0u) | (x0 > xp ? 32u : 0u);
        }
    }
    if constexpr (PAIR) {
        // the two texels of a row as ONE 8-byte load (the address unit is what bounds these kernels: half the instructions)
        uint2 pa, pb;
        if constexpr (BUF32) {   // rows and columns of an image are below 2^24: v_mad_u32_u24 is exact and full rate
            // interior waves: the lower row is the upper one's offset plus the pitch, which rides in the load's scalar offset (no second address)
            const uint32_t oa = (__umul24((uint32_t)ya, (uint32_t)src_w) + (uint32_t)xp) << 2;
            uint32_t ob = oa, sb = (uint32_t)src_w << 2;
            if (!T.all_in) {
                ob = (__umul24((uint32_t)yb, (uint32_t)src_w) + (uint32_t)xp) << 2; sb = 0u;
                asm volatile("" : "+v"(ob));   // stays on the border path (if-conversion would compute it for every wave and select)
            }
            const pfx_w_v2i va = pfx_w_buffer_load_v2i32(S.rs, (int)oa, 0, 0);
            const pfx_w_v2i vb = pfx_w_buffer_load_v2i32(S.rs, (int)ob, (int)sb, 0);
            pa = make_uint2((uint32_t)va.x, (uint32_t)va.y); pb = make_uint2((uint32_t)vb.x, (uint32_t)vb.y);
        } else {
            const uint32_t* ra = src + (size_t)(uint32_t)ya * (uint32_t)src_w;
            const uint32_t* rb = src + (size_t)(uint32_t)yb * (uint32_t)src_w;
            pa = *reinterpret_cast<const uint2*>(ra + (uint32_t)xp); pb = *reinterpret_cast<const uint2*>(rb + (uint32_t)xp);
        }
        T.tl = pa.x; T.tr = pa.y; T.bl = pb.x; T.br = pb.y;
    } else {
        const uint32_t* ra = src + (size_t)(uint32_t)ya * (uint32_t)src_w;
        const uint32_t* rb = src + (size_t)(uint32_t)yb * (uint32_t)src_w;
        T.tl = ra[(uint32_t)xa]; T.tr = ra[(uint32_t)xb]; T.bl = rb[(uint32_t)xa]; T.br = rb[(uint32_t)xb];
    }
    return T;
}
PFX_DEV uint32_t bilinear_finish(const bilinear_taps& T)
{
    uint32_t tl = T.tl, tr = T.tr, bl = T.bl, br = T.br;
    if (!T.all_in) { // texels outside the source are 0 (:1318-1331)
        if (T.m & 16u) { tr = tl; br = bl; }        // x0 == -1: the pair started at texel 0, which is tr / br
        else if (T.m & 32u) { tl = tr; bl = br; }   // x0 == w - 1: the pair ended at the last texel, which is tl / bl
        tl = (T.m & 1u) ? tl : 0u; tr = (T.m & 2u) ? tr : 0u; bl = (T.m & 4u) ? bl : 0u; br = (T.m & 8u) ? br : 0u;
    }
    float o[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // pinned as floats: left alone, hipcc turns `(float)b - (float)a` into an SDWA byte subtract + v_cvt_f32_i32 — exact, but an SDWA form costs the SIMD
        // 4.75 cycles in the mix against 2 for v_cvt_f32_ubyteN and 2 for v_sub_f32 (profiles/r04_valu_rates.txt)
        const float ftl = pin((float)((tl >> (8 * c)) & 0xffu)), ftr = pin((float)((tr >> (8 * c)) & 0xffu));
        const float fbl = pin((float)((bl >> (8 * c)) & 0xffu)), fbr = pin((float)((br >> (8 * c)) & 0xffu));
        const float top = ftl + (ftr - ftl) * T.fx; // :1337-1339
        const float bot = fbl + (fbr - fbl) * T.fx;
        o[c] = top + (bot - top) * T.fy;
    }
    return pack_round_rgba_finite(o[0], o[1], o[2], o[3]); // finite, within [-eps, 255 + eps]: `.round().clamp(0, 255) as u8` in 8 instructions
}

constexpr uint32_t WARP_YR = 4; // rows per lane: the field entries of all of them, then all 16 taps, are requested before any is consumed

// One lane per output column, WARP_YR rows: a pixel needs two dependent memory round trips (its field entry, then the four taps the
// entry points at), so with one pixel per lane the kernel is bound by latency x occupancy (Little's law: 0.60 ms at 16K for 2.1 GB
// = 3.5 TB/s); four pixels' worth of requests in flight per lane move it towards the HBM rate.
// `y_off`: index of the buffers' row 0 in the whole output when `disp` / `dst` are a band of it (a document sharded by rows, SURVEY 8e: the source is
// replicated, every member warps its band of the output); 0 for a whole image.
template <bool PAIR, bool BUF32>
__global__ __launch_bounds__(256) void warp_disp_kernel(const uint32_t* __restrict__ src, int32_t sw, int32_t sh,
                                                        const float2* __restrict__ disp, uint32_t w, uint32_t h,
                                                        uint32_t* __restrict__ dst, uint32_t y_off)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y0 = (blockIdx.y * 4u + (threadIdx.x >> 6)) * WARP_YR;
    if (x >= w || y0 >= h) return;
    float2 d[WARP_YR];
#pragma unroll
    for (uint32_t k = 0; k < WARP_YR; ++k) d[k] = disp[(size_t)min(y0 + k, h - 1u) * w + x]; // rows past the end re-read the last one (unused)
    const warp_src S = make_warp_src(src, sw, sh);
    bilinear_taps taps[WARP_YR];
#pragma unroll
    for (uint32_t k = 0; k < WARP_YR; ++k) taps[k] = bilinear_fetch<PAIR, BUF32>(S, sw, sh, (float)x, (float)(min(y0 + k, h - 1u) + y_off), d[k].x, d[k].y);
#pragma unroll
    for (uint32_t k = 0; k < WARP_YR; ++k)
        if (y0 + k < h) dst[(size_t)(y0 + k) * w + x] = bilinear_finish(taps[k]);
}

PFX_DEV void cr_weights(float t, float (&wt)[4]) // :1558-1567
{
    const float t2 = t * t, t3 = t2 * t;
    wt[0] = -0.5f * t3 + t2 - 0.5f * t;
    wt[1] = 1.5f * t3 - 2.5f * t2 + 1.0f;
    wt[2] = -1.5f * t3 + 2.0f * t2 + 0.5f * t;
    wt[3] = 0.5f * t3 - 0.5f * t2;
}

// :1589-1646; pts = (rows+1) x (cols+1) xy pairs (LDS or global)
PFX_DEV float2 cr_surface(const float2* __restrict__ pts, uint32_t cols, uint32_t rows, float u_global, float v_global)
{
    const uint32_t ppr = cols + 1u, num_rows = rows + 1u;
    const float col_f = rs_clamp(u_global, 0.0f, (float)cols - 0.0001f);
    const float row_f = rs_clamp(v_global, 0.0f, (float)rows - 0.0001f);
    const uint32_t ci = min((uint32_t)col_f, cols - 1u), ri = min((uint32_t)row_f, rows - 1u);
    const float u_local = col_f - (float)ci, v_local = row_f - (float)ri;
    float wv[4], wu[4];
    cr_weights(v_local, wv);
    cr_weights(u_local, wu);
    const uint32_t rv[4] = {ri == 0u ? 0u : ri - 1u, ri, min(ri + 1u, num_rows - 1u), min(ri + 2u, num_rows - 1u)};
    const uint32_t cu[4] = {ci == 0u ? 0u : ci - 1u, ci, min(ci + 1u, ppr - 1u), min(ci + 2u, ppr - 1u)};
    float rx[4], ry[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2* base = pts + rv[j] * ppr;
        const float2 p0 = base[cu[0]], p1 = base[cu[1]], p2 = base[cu[2]], p3 = base[cu[3]];
        rx[j] = wu[0] * p0.x + wu[1] * p1.x + wu[2] * p2.x + wu[3] * p3.x;
        ry[j] = wu[0] * p0.y + wu[1] * p1.y + wu[2] * p2.y + wu[3] * p3.y;
    }
    return make_float2(wv[0] * rx[0] + wv[1] * rx[1] + wv[2] * rx[2] + wv[3] * rx[3],
                       wv[0] * ry[0] + wv[1] * ry[1] + wv[2] * ry[2] + wv[3] * ry[3]);
}

// The same surface with its u-dependent half cached: a lane walks down a column (u fixed), so the per-control-row partial sums
// rx[j], ry[j] (56 of the ~107 operations of an evaluation) only change when the pixel row enters another mesh cell row — a
// wave-uniform event, since a wave shares y.  Same operations in the same order per result: bit-identical to cr_surface.
struct cr_column {
    float wu[4];
    uint32_t cu[4];
    float rx[4], ry[4];
    uint32_t ri_cached;
};
PFX_DEV void cr_column_init(cr_column& C, uint32_t cols, float u_global)
{
    const uint32_t ppr = cols + 1u;
    const float col_f = rs_clamp(u_global, 0.0f, (float)cols - 0.0001f);
    const uint32_t ci = min((uint32_t)col_f, cols - 1u);
    cr_weights(col_f - (float)ci, C.wu);
    C.cu[0] = ci == 0u ? 0u : ci - 1u; C.cu[1] = ci; C.cu[2] = min(ci + 1u, ppr - 1u); C.cu[3] = min(ci + 2u, ppr - 1u);
    C.ri_cached = 0xffffffffu;
}
// the v-dependent half of an evaluation (shared by both surfaces of a pixel, and by all 64 pixels of a wave's row)
struct cr_row { uint32_t ri; float wv[4]; };
PFX_DEV cr_row cr_row_of(uint32_t rows, float v_global)
{
    cr_row R;
    const float row_f = rs_clamp(v_global, 0.0f, (float)rows - 0.0001f);
    R.ri = min((uint32_t)row_f, rows - 1u);
    cr_weights(row_f - (float)R.ri, R.wv);
    return R;
}
// the per-control-row partial sums of cell row `ri` (the u-dependent half)
PFX_DEV void cr_column_fill(cr_column& C, const float2* __restrict__ pts, uint32_t cols, uint32_t rows, uint32_t ri)
{
    const uint32_t ppr = cols + 1u, num_rows = rows + 1u;
    C.ri_cached = ri;
    const uint32_t rv[4] = {ri == 0u ? 0u : ri - 1u, ri, min(ri + 1u, num_rows - 1u), min(ri + 2u, num_rows - 1u)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2* base = pts + rv[j] * ppr;
        const float2 p0 = base[C.cu[0]], p1 = base[C.cu[1]], p2 = base[C.cu[2]], p3 = base[C.cu[3]];
        C.rx[j] = C.wu[0] * p0.x + C.wu[1] * p1.x + C.wu[2] * p2.x + C.wu[3] * p3.x;
        C.ry[j] = C.wu[0] * p0.y + C.wu[1] * p1.y + C.wu[2] * p2.y + C.wu[3] * p3.y;
    }
}
// FILLED: the caller has filled the sums for this row's cell row (no test)
template <bool FILLED = false>
PFX_DEV float2 cr_column_eval(cr_column& C, const float2* __restrict__ pts, uint32_t cols, uint32_t rows, const cr_row& R)
{
    const float (&wv)[4] = R.wv;
    if constexpr (!FILLED)
        if (R.ri != C.ri_cached) cr_column_fill(C, pts, cols, rows, R.ri); // uniform across the wave
    return make_float2(wv[0] * C.rx[0] + wv[1] * C.rx[1] + wv[2] * C.rx[2] + wv[3] * C.rx[3],
                       wv[0] * C.ry[0] + wv[1] * C.ry[1] + wv[2] * C.ry[2] + wv[3] * C.ry[3]);
}

constexpr uint32_t MESH_LDS_PTS = 2048; // control points per grid staged in LDS (2 grids x 16 KiB)
#ifndef PFX_MESH_YR
#define PFX_MESH_YR 8
#endif
constexpr uint32_t MESH_YR = PFX_MESH_YR; // rows per batch: their taps are all requested before any is consumed
constexpr uint32_t MESH_YB = 32 / MESH_YR; // batches walked by one lane (a block covers 64 x 128 pixels): the u-dependent half of the surface —
                                        // column weights, 2 x 56 operations + 64 LDS reads of per-control-row sums — is set up once per walk

// The displacement field alone (generate_displacement_from_mesh; the fused field + gather is mesh_roll_kernel below).  IN_LDS: control points staged
// in LDS (the normal case: a 6x6 grid is 49 points); the pointer's address space is then known at compile time (ds_read, not flat_load).
template <bool IN_LDS>
__global__ __launch_bounds__(256) void mesh_kernel(const float2* __restrict__ g_orig, const float2* __restrict__ g_def, uint32_t cols, uint32_t rows,
                                                   uint32_t w, uint32_t h, float2* __restrict__ disp, uint32_t y_off, uint32_t h_full)
{
    // h = rows of `disp`; they are rows [y_off, y_off + h) of an h_full-row image (y_off = 0, h_full = h: the whole image)
    // control points in dynamic LDS, 16 bytes per point: a static 2 x 16 KB pair capped the kernel at five workgroups per CU whatever its registers
    extern __shared__ __attribute__((aligned(16))) uint8_t mesh_lds[];
    float2* const s_orig = reinterpret_cast<float2*>(mesh_lds);
    float2* const s_def = s_orig + (IN_LDS ? (cols + 1u) * (rows + 1u) : 0u);
    if constexpr (IN_LDS) {
        const uint32_t npts = (cols + 1u) * (rows + 1u);
        for (uint32_t i = threadIdx.x; i < npts; i += 256u) {
            if (g_orig) s_orig[i] = g_orig[i];
            s_def[i] = g_def[i];
        }
        __syncthreads();
    }
    const uint32_t x_lane = blockIdx.x * 64u + (threadIdx.x & 63u), y_walk = (blockIdx.y * 4u + (threadIdx.x >> 6)) * (MESH_YR * MESH_YB);
    if (y_walk >= h) return;               // whole wave
    const bool x_valid = x_lane < w;       // lanes past the right edge stay alive: the row halves below are exchanged by lane index
    const uint32_t x = x_valid ? x_lane : w - 1u;
    // :1687-1688; operands in [0.5, 2^15]: k_common.h:fdiv_fast is bit-identical to '/'
    const float u = fdiv_fast((float)x + 0.5f, (float)w) * (float)cols;
    cr_column cd, co;
    cr_column_init(cd, cols, u);
    if (g_orig) cr_column_init(co, cols, u);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t b = 0; b < MESH_YB; ++b) {
        const uint32_t y_first = y_walk + b * MESH_YR;
        if (y_first >= h) break;           // whole wave
        // a wave shares its rows: lane k (< MESH_YR) evaluates row k's v-dependent half once, everyone reads it back through
        // v_readlane (scalar operands from then on) instead of recomputing ~30 operations per pixel
        cr_row mine = cr_row_of(rows, fdiv_fast((float)(y_first + y_off + (lane & (MESH_YR - 1u))) + 0.5f, (float)h_full) * (float)rows);
        const uint32_t y_end = min(y_first + MESH_YR, h);
#pragma unroll
        for (uint32_t k = 0; k < MESH_YR; ++k) {
            const uint32_t y = min(y_first + k, y_end - 1u); // rows past the end repeat the last one (never stored): no branch around the loads
            cr_row R;
            const int kl = (int)(y - y_first);                  // the lane that evaluated this row's v-dependent half
            R.ri = (uint32_t)__builtin_amdgcn_readlane((int)mine.ri, kl);
#pragma unroll
            for (int j = 0; j < 4; ++j) R.wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.wv[j]), kl));
            float2 d, o;
            if constexpr (IN_LDS) d = cr_column_eval(cd, s_def, cols, rows, R);
            else d = cr_column_eval(cd, g_def, cols, rows, R);
            if (g_orig) {
                if constexpr (IN_LDS) o = cr_column_eval(co, s_orig, cols, rows, R);
                else o = cr_column_eval(co, g_orig, cols, rows, R);
            } else o = make_float2((float)x + 0.5f, (float)(y + y_off) + 0.5f); // _fast: uniform original grid is the identity (:1735-1736)
            const float ddx = d.x - o.x, ddy = d.y - o.y;
            const size_t i = (size_t)y * w + x;
            if (x_valid && y_first + k < y_end) disp[i] = make_float2(ddx, ddy);
        }
    }
}


#ifndef PFX_MESH_ROLL
#define PFX_MESH_ROLL 2
#endif
#ifndef PFX_MESH_WALK
#define PFX_MESH_WALK 48
#endif
#ifndef PFX_MESH_WX
#define PFX_MESH_WX 4
#endif
// rows walked by a lane (<= 64: lane k evaluates row k's v-dependent half) and waves of a workgroup side by side in x.  16K, three alternations on one box
// (profiles/r04_tuning.md): 32 x 1 0.447 ms, 16 x 1 0.48, 8 x 1 0.52, 32 x 4 0.434, 48 x 4 0.422, 64 x 4 0.427 — a workgroup writes 1 KB runs of a row, and the
// u-dependent half of the surface is set up once per 48 rows
constexpr uint32_t MESH_WALK = PFX_MESH_WALK, MESH_WX = PFX_MESH_WX;
static_assert(MESH_WALK <= 64 && (MESH_WX == 1 || MESH_WX == 2 || MESH_WX == 4), "mesh_roll_kernel: lane k evaluates row k; 4 waves per workgroup");
// Rolling form of the fused warp (round 4): D rows' taps in flight all the time — row k + D is requested as soon as row k has been interpolated —
// instead of batches of eight requested together and then consumed together; 7 D registers of taps instead of 56, so more waves fit.
template <bool IN_LDS, int D, bool PAIR, bool BUF32>
__global__ __launch_bounds__(256) void mesh_roll_kernel(const uint32_t* __restrict__ src, const float2* __restrict__ g_orig, const float2* __restrict__ g_def,
                                                        uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint32_t* __restrict__ dst, uint32_t y_off,
                                                        uint32_t h_full, uint32_t gx, uint32_t gy)
{
    constexpr uint32_t WALK = MESH_WALK, WX = MESH_WX; // rows per lane; a workgroup covers (64 WX) x (WALK 4 / WX) pixels
    // Tile of this workgroup.  gx != 0: a one-dimensional launch with an XCD-aware order (round 5) — block b runs on XCD b % 8 and each XCD has its own L2, so
    // XCD k takes the k-th eighth of the tile rows and walks it in raster order: the workgroups resident on an XCD at one time are neighbours, and the source
    // texels their footprints share (the smooth field moves a tile's window by a fraction of its size) are fetched into that L2 once instead of into several.
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (gx != 0u) {
        const uint32_t xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
        const uint32_t r0 = gy * xcd / 8u, r1 = gy * (xcd + 1u) / 8u;
        if (j >= (r1 - r0) * gx) return;
        by = r0 + j / gx; bx = j - (j / gx) * gx;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t mesh_lds[];
    float2* const s_orig = reinterpret_cast<float2*>(mesh_lds);
    float2* const s_def = s_orig + (IN_LDS ? (cols + 1u) * (rows + 1u) : 0u);
    if constexpr (IN_LDS) {
        const uint32_t npts = (cols + 1u) * (rows + 1u);
        for (uint32_t i = threadIdx.x; i < npts; i += 256u) {
            if (g_orig) s_orig[i] = g_orig[i];
            s_def[i] = g_def[i];
        }
        __syncthreads();
    }
    const float2* p_def = IN_LDS ? s_def : g_def;
    const float2* p_orig = IN_LDS ? s_orig : g_orig;
    // the wave's first row as a scalar: row counters, the rows' v_readlane indices and the end-of-image tests then stay on the scalar unit
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lane = threadIdx.x & 63u, x_lane = (bx * WX + wave % WX) * 64u + lane;
    const uint32_t y_walk = (by * (4u / WX) + wave / WX) * WALK;
    if (y_walk >= h) return;               // whole wave
    const bool x_valid = x_lane < w;
    const uint32_t x = x_valid ? x_lane : w - 1u;
    const float u = fdiv_fast((float)x + 0.5f, (float)w) * (float)cols; // :1687-1688
    cr_column cd, co;
    cr_column_init(cd, cols, u);
    if (g_orig) cr_column_init(co, cols, u);
    const uint32_t n_rows = min(WALK, h - y_walk);
    // lane k (< 32) evaluates row k's v-dependent half once (mesh_kernel)
    const cr_row mine = cr_row_of(rows, fdiv_fast((float)(y_walk + y_off + (lane & (WALK <= 32u ? 31u : 63u))) + 0.5f, (float)h_full) * (float)rows);
    const warp_src S = make_warp_src(src, (int32_t)w, (int32_t)h_full);
    // ONE_CELL_ROW: all rows of the walk lie in one row of mesh cells (all but one walk in rows-of-cells / 32): the per-control-row sums are loop
    // invariants — no cache test per row, and none of the register copies the compiler puts at that test's join
    auto walk = [&](auto one_cell_row) {
        constexpr bool ONE = decltype(one_cell_row)::value;
        auto fetch = [&](uint32_t k) {         // rows past the end repeat the last one (never stored): no branch around the loads
            const uint32_t kk = min(k, n_rows - 1u), y = y_walk + kk;
            cr_row R;
            R.ri = ONE ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)mine.ri, (int)kk);   // unused under ONE
#pragma unroll
            for (int j = 0; j < 4; ++j) R.wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine.wv[j]), (int)kk));
            const float2 d = cr_column_eval<ONE>(cd, p_def, cols, rows, R);
            float2 o;
            if (g_orig) o = cr_column_eval<ONE>(co, p_orig, cols, rows, R);
            else o = make_float2((float)x + 0.5f, (float)(y + y_off) + 0.5f);
            return bilinear_fetch<PAIR, BUF32>(S, (int32_t)w, (int32_t)h_full, (float)x, (float)(y + y_off), d.x - o.x, d.y - o.y);
        };
        bilinear_taps taps[D];
#pragma unroll
        for (int j = 0; j < D; ++j) taps[j] = fetch((uint32_t)j);
        for (uint32_t k0 = 0; k0 < n_rows; k0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const uint32_t k = k0 + j;
                const uint32_t px = bilinear_finish(taps[j]);
                if (x_valid && k < n_rows) dst[(size_t)(y_walk + k) * w + x] = px;
                taps[j] = fetch(k + D);
            }
        }
    };
    const uint32_t ri_first = (uint32_t)__builtin_amdgcn_readlane((int)mine.ri, 0), ri_last = (uint32_t)__builtin_amdgcn_readlane((int)mine.ri, (int)(n_rows - 1u));
    if (ri_first == ri_last) {
        cr_column_fill(cd, p_def, cols, rows, ri_first);
        if (g_orig) cr_column_fill(co, p_orig, cols, rows, ri_first);
        walk(std::true_type{});
    } else walk(std::false_type{});
}

} // namespace

extern "C" hipError_t pfxk_warp_displacement(hipStream_t s, const uint8_t* d_src, uint32_t sw, uint32_t sh,
                                             const float* d_disp, uint32_t w, uint32_t h, uint8_t* d_dst, uint32_t first_row)
{
    if (w == 0 || h == 0) return hipSuccess;
    dim3 g((w + 63) / 64, (h + 4 * WARP_YR - 1) / (4 * WARP_YR));
    const bool buf32 = sw >= 2u && sw < (1u << 24) && sh < (1u << 24) && (uint64_t)sw * sh * 4u < (1ull << 32);   // 32-bit byte offsets through a buffer resource
#define PFX_WD(P, B) warp_disp_kernel<P, B><<<g, 256, 0, s>>>((const uint32_t*)d_src, (int32_t)sw, (int32_t)sh, (const float2*)d_disp, w, h, (uint32_t*)d_dst, first_row)
    if (buf32) PFX_WD(true, true); else if (sw >= 2u) PFX_WD(true, false); else PFX_WD(false, false);
#undef PFX_WD
    return hipGetLastError();
}

// ---- DisplacementField brushes on a device-resident field (apply_push / expand / contract / twirl, transform.rs:1051-1200) ----
// The reference stamps dabs one after another; here one lane owns a pixel of the stroke's bounding box and walks the dab list in
// order, accumulating in registers (same per-pixel order of the `+=`), one read and one write of the field per launch.
// exp() is evaluated in f64 and rounded once (see k_effects2.hip's header): the field may differ from the CPU path in the last
// ulp of a weight, which stays far below the warp's +-1 LSB.
// CHUNKED: the launch runs over the 64 x 64 chunks some dab's box touches (`chunks[blockIdx.x]` = chunk x | chunk y << 16, built by the host) instead of the
// dabs' common bounding box, which for dabs spread over a large field is mostly untouched
template <bool CHUNKED>
__global__ __launch_bounds__(256) void disp_brush_kernel(float2* __restrict__ disp, uint32_t w, const pfxk_disp_dab* __restrict__ dabs, uint32_t n,
                                                         int bx0, int by0, int bx1, int by1, const uint32_t* __restrict__ chunks)
{
    if constexpr (CHUNKED) {   // the chunk, clipped to the dabs' bounding box (exclusive upper bounds)
        const uint32_t ch = chunks[blockIdx.x];
        const int cx0 = (int)((ch & 0xffffu) * 64u), cy0 = (int)((ch >> 16) * 64u);
        bx1 = min(bx1, cx0 + 64); by1 = min(by1, cy0 + 64); bx0 = max(bx0, cx0); by0 = max(by0, cy0);
    }
    // a wave covers one 64-pixel row segment (py and the segment's x range are wave-uniform); dabs are culled against it 64 at a time
    // with the reference's own loop bounds, so a lane only walks the dabs its wave can see — in order, like the reference's `+=`
    const uint32_t lane = threadIdx.x & 63u;
    const int px0 = bx0 + (CHUNKED ? 0 : (int)(blockIdx.x * 64u)), px = px0 + (int)lane, py = by0 + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (py >= by1 || px0 >= bx1) return; // whole wave
    const bool active = px < bx1;
    float2* p = disp + (size_t)py * w + (size_t)min(px, bx1 - 1);
    // the field is read when the wave meets its first dab and written back only by the lanes a dab reached: the dabs' common bounding box is mostly
    // untouched field (32 dabs of radius 400 spread over a 16K field: 0.36 -> 0.20 ms), and `x += 0` would not even be a no-op for -0.0
    float2 d = make_float2(0.0f, 0.0f);
    bool loaded = false, touched = false;
    const int seg_hi = min(px0 + 64, bx1); // exclusive
    auto dab_px = [&](uint32_t k) {
        const pfxk_disp_dab D = dabs[k]; // uniform -> scalar loads
        if (px < D.x0 || px >= D.x1 || py < D.y0 || py >= D.y1) return; // the reference's loop bounds (:1066-1069)
        const float dx = (float)px - D.cx, dy = (float)py - D.cy;
        const float dist_sq = dx * dx + dy * dy;
        if (dist_sq > D.r * D.r) return;
        touched = true;
        if (D.mode == 0) {        // push :1051-1085
            const float weight = (float)exp((double)(-dist_sq / D.sigma_sq_2)) * D.strength;
            d.x += D.delta_x * weight;
            d.y += D.delta_y * weight;
        } else if (D.mode == 1) { // expand :1087-1120
            const float dist = __builtin_fmaxf(__builtin_sqrtf(dist_sq), 0.001f);
            const float t = dist / D.r;
            const float weight = (1.0f - t) * (1.0f - t) * D.strength * 3.0f;
            d.x += dx / dist * weight;
            d.y += dy / dist * weight;
        } else if (D.mode == 2) { // contract :1122-1155
            const float dist = __builtin_fmaxf(__builtin_sqrtf(dist_sq), 0.001f);
            const float weight = (float)exp((double)(-dist_sq / D.sigma_sq_2)) * D.strength;
            d.x += -dx / dist * weight * 2.0f;
            d.y += -dy / dist * weight * 2.0f;
        } else {                  // twirl cw (3) / ccw (4) :1157-1200
            const float weight = (float)exp((double)(-dist_sq / D.sigma_sq_2)) * D.strength * (D.mode == 3 ? 1.0f : -1.0f);
            d.x += -dy * weight * 0.1f;
            d.y += dx * weight * 0.1f;
        }
    };
    for (uint32_t k0 = 0; k0 < n; k0 += 64u) {
        const uint32_t kk = k0 + lane;
        bool seen = false;
        if (kk < n) seen = !(seg_hi <= dabs[kk].x0 || px0 >= dabs[kk].x1 || py < dabs[kk].y0 || py >= dabs[kk].y1);
        unsigned long long m = __ballot(seen);
        if (m && !loaded) { loaded = true; if (active) d = *p; }   // wave-uniform
        while (m) {
            const uint32_t k = k0 + (uint32_t)__builtin_ctzll(m);
            m &= m - 1ull;
            if (active) dab_px(k);
        }
    }
    if (active && touched) *p = d;
}

extern "C" hipError_t pfxk_disp_brushes(hipStream_t s, float* d_disp, uint32_t w, uint32_t h, const pfxk_disp_dab* d_dabs, uint32_t n, int bx0, int by0,
                                        int bx1, int by1)
{
    (void)h;
    if (n == 0 || bx1 <= bx0 || by1 <= by0) return hipSuccess;
    dim3 g((uint32_t)(bx1 - bx0 + 63) / 64, (uint32_t)(by1 - by0 + 3) / 4);
    disp_brush_kernel<false><<<g, 256, 0, s>>>((float2*)d_disp, w, d_dabs, n, bx0, by0, bx1, by1, nullptr);
    return hipGetLastError();
}

// the same over a list of 64 x 64 chunks (chunk x | chunk y << 16) instead of the whole bounding box
extern "C" hipError_t pfxk_disp_brushes_chunked(hipStream_t s, float* d_disp, uint32_t w, uint32_t h, const pfxk_disp_dab* d_dabs, uint32_t n, int bx0, int by0,
                                                int bx1, int by1, const uint32_t* d_chunks, uint32_t n_chunks)
{
    (void)h;
    if (n == 0 || n_chunks == 0 || bx1 <= bx0 || by1 <= by0) return hipSuccess;
    disp_brush_kernel<true><<<dim3(n_chunks, 16), 256, 0, s>>>((float2*)d_disp, w, d_dabs, n, bx0, by0, bx1, by1, d_chunks);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_mesh_displacement(hipStream_t s, const float* d_orig, const float* d_def, uint32_t cols,
                                             uint32_t rows, uint32_t w, uint32_t h, float* d_disp)
{
    if (w == 0 || h == 0) return hipSuccess;
    dim3 g((w + 63) / 64, (h + 4 * MESH_YR * MESH_YB - 1) / (4 * MESH_YR * MESH_YB));
    if ((cols + 1u) * (rows + 1u) <= MESH_LDS_PTS)
        mesh_kernel<true><<<g, 256, (size_t)(cols + 1u) * (rows + 1u) * 16u, s>>>((const float2*)d_orig, (const float2*)d_def, cols, rows, w, h, (float2*)d_disp, 0u, h);
    else
        mesh_kernel<false><<<g, 256, 0, s>>>((const float2*)d_orig, (const float2*)d_def, cols, rows, w, h, (float2*)d_disp, 0u, h);
    return hipGetLastError();
}

int g_mesh_xcd = 1; // pfxk_warp_set_mesh_xcd (pfx_tune "mesh_xcd"): XCD-aware tile order of the fused mesh warp
extern "C" void pfxk_warp_set_mesh_xcd(int on) { g_mesh_xcd = on; }
// band form: d_dst holds rows [first_row, first_row + h) of an h_full-row result; d_src is the whole w x h_full source
extern "C" hipError_t pfxk_warp_mesh(hipStream_t s, const uint8_t* d_src, const float* d_orig, const float* d_def,
                                     uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint8_t* d_dst, uint32_t first_row, uint32_t h_full)
{
    if (w == 0 || h == 0) return hipSuccess;
    dim3 g((w + 64 * MESH_WX - 1) / (64 * MESH_WX), (h + MESH_WALK * (4 / MESH_WX) - 1) / (MESH_WALK * (4 / MESH_WX)));
    // rows in flight per lane: 2 measured best at 16K (profiles/r04_tuning.md: 0.565 ms against 0.573-0.59 for 3 / 4, 0.63 for 6 and for round 3's batches of 8)
    constexpr int ROLL = PFX_MESH_ROLL;
    const size_t lds = (size_t)(cols + 1u) * (rows + 1u) * 16u;
    const bool in_lds = (cols + 1u) * (rows + 1u) <= MESH_LDS_PTS;
    const bool buf32 = w >= 2u && w < (1u << 24) && h_full < (1u << 24) && (uint64_t)w * h_full * 4u < (1ull << 32);
    uint32_t gx = 0u, gy = 0u;
    if (g_mesh_xcd && g.y >= 16u) {   // XCD-aware tile order (mesh_roll_kernel); small launches keep the plain 2-D grid
        gx = g.x; gy = g.y;
        g = dim3(8u * ((gy + 7u) / 8u) * gx, 1u);
    }
#define PFX_ROLL(L, P, B) mesh_roll_kernel<L, ROLL, P, B><<<g, 256, (L) ? lds : 0, s>>>((const uint32_t*)d_src, (const float2*)d_orig, (const float2*)d_def, cols, rows, w, h, (uint32_t*)d_dst, first_row, h_full, gx, gy)
    if (in_lds) { if (buf32) PFX_ROLL(true, true, true); else if (w >= 2u) PFX_ROLL(true, true, false); else PFX_ROLL(true, false, false); }
    else { if (buf32) PFX_ROLL(false, true, true); else if (w >= 2u) PFX_ROLL(false, true, false); else PFX_ROLL(false, false, false); }
#undef PFX_ROLL
    return hipGetLastError();
}
