// k_flatten.hip — the layer compositor: CanvasState::composite() as ONE streaming gfx950 kernel.
//
// Reference: src/canvas/canvas_state.rs:505-698 (composite_viewport) and :1246-1505 (blend_pixel_static and the
// per-channel helpers); adjustment layers src/canvas/layers.rs:276-325; live mask :660-665.
//
// Design (streaming, no LDS, no MFMA):
//   * one lane owns 4 consecutive pixels: every layer is read with one 16-byte load per lane (1 KiB per wave
//     instruction, fully coalesced), the result is written with one 16-byte store;
//   * the accumulator ("pixels[idx]" in the reference, a u8 RGBA re-quantised after every layer) lives in
//     registers for the whole layer stack as integer-valued floats — nothing but the N layer reads and the one
//     result write touches HBM: 4*N + 4 bytes per pixel, the algorithmic minimum;
//   * the blend mode is uniform per layer, so the 25-way dispatch is a scalar branch outside the pixel code and
//     each mode is its own straight-line specialisation;
//   * arithmetic is the reference's f32 sequence, operation for operation, without FMA contraction
//     (this file is compiled with -ffp-contract=off); u8/255 uses the proved 2-op form (k_common.h:div255);
//   * the three `/ out_a` of a pixel share one refined reciprocal (rdiv below): the exact operation sequence
//     hipcc emits for an IEEE f32 divide, with the per-denominator part hoisted — bit-identical to `/`.
#include <algorithm>
#include <atomic>
#include <type_traits>
#include "k_common.h"
#include "pfx_kernels.h"

using namespace pfxk;

#include "k_blend.h"

// LLVM buffer intrinsics hipcc has no __builtin for (declared outside the anonymous namespace: they are external symbols)
typedef float pfx_v4f __attribute__((ext_vector_type(4)));
typedef int pfx_v4i __attribute__((ext_vector_type(4)));
__device__ pfx_v4f pfx_buffer_load_format_v4f32(pfx_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ float pfx_buffer_load_format_f32(pfx_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.f32");
__device__ void pfx_buffer_store_i32(int data, pfx_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ void pfx_buffer_store_format_v4f32(pfx_v4f data, pfx_v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.format.v4f32");


namespace {


// live layer mask: top.a = (a * (255 - conceal)) / 255, integer (canvas_state.rs:660-665)
PFX_DEV uint32_t apply_conceal(uint32_t px, uint32_t conceal)
{
    if (conceal == 0u) return px;
    uint32_t a = ((px >> 24) * (255u - conceal)) / 255u;
    return (px & 0x00ffffffu) | (a << 24);
}

// AdjustmentLayerData::apply_to_pixel_with_opacity (layers.rs:276-325) on one accumulator pixel
PFX_DEV void adjust_px(float (&p)[4], uint32_t kind, const float* __restrict__ adj, float opacity)
{
    float o[4] = {p[0], p[1], p[2], p[3]};
    switch (kind) {
    case PFXK_ADJ_EXPOSURE: { // adj[0] = gain = powf(2, ev), computed on the host with glibc like the reference
        const float gain = adj[0];
        o[0] = quant255(p[0] * gain); o[1] = quant255(p[1] * gain); o[2] = quant255(p[2] * gain);
        break;
    }
    case PFXK_ADJ_BRIGHTNESS_CONTRAST: { // adj[0]=brightness, adj[1]=factor (host-computed)
        const float br = adj[0], factor = adj[1];
        o[0] = quant255(factor * (p[0] + br - 128.0f) + 128.0f);
        o[1] = quant255(factor * (p[1] + br - 128.0f) + 128.0f);
        o[2] = quant255(factor * (p[2] + br - 128.0f) + 128.0f);
        break;
    }
    case PFXK_ADJ_INVERT: o[0] = 255.0f - p[0]; o[1] = 255.0f - p[1]; o[2] = 255.0f - p[2]; break;
    case PFXK_ADJ_CHANNEL_MIXER:
#pragma unroll
        for (int c = 0; c < 4; ++c)
            o[c] = quant255(p[0] * adj[c * 4 + 0] + p[1] * adj[c * 4 + 1] + p[2] * adj[c * 4 + 2] + p[3] * adj[c * 4 + 3]);
        break;
    default: break;
    }
    const float t = rs_clamp(opacity, 0.0f, 1.0f), inv = 1.0f - t;
#pragma unroll
    for (int c = 0; c < 4; ++c) p[c] = round_u8f(p[c] * inv + o[c] * t); // `.round() as u8`
}

// Tool preview folded into the active layer's pixel before masking / compositing (canvas_state.rs:621-658): `top` is the layer
// pixel, `pp` the preview pixel.  Uses the plain-divide instantiation: this path is interactive-only and tiny.
PFX_DEV uint32_t preview_apply(uint32_t top, uint32_t pp, const pfxk_preview& PV)
{
    if (PV.replaces) return pp;                                        // :623
    if ((pp >> 24) == 0u) return top;                                  // :625
    if (PV.is_eraser) {                                                // :626-631
        const float mask_strength = div255(ubyte3(pp)), current_a = div255(ubyte3(top));
        const float new_a = __builtin_fmaxf(current_a * (1.0f - mask_strength), 0.0f);
        return (top & 0x00ffffffu) | ((uint32_t)quant255(new_a * 255.0f) << 24);
    }
    float b[1][4] = {{ubyte0(top), ubyte1(top), ubyte2(top), ubyte3(top)}};
    const uint32_t t1[1] = {pp};
    blend4_dispatch<false, 1>(PV.mode, b, t1, 1.0f, 1.0f);             // blend_pixel_static(top, pp, preview_blend, 1.0)
    if (PV.mode == M_OVERWRITE || PV.mode == M_XOR) {                  // :632-653 coverage-weighted lerp
        const float cov = div255(ubyte3(pp)), inv = 1.0f - cov;
        return pack_rgba(quant255(ubyte0(top) * inv + b[0][0] * cov + 0.5f), quant255(ubyte1(top) * inv + b[0][1] * cov + 0.5f),
                         quant255(ubyte2(top) * inv + b[0][2] * cov + 0.5f), quant255(ubyte3(top) * inv + b[0][3] * cov + 0.5f));
    }
    return pack_rgba(b[0][0], b[0][1], b[0][2], b[0][3]);
}

template <bool GENERAL, bool F>
__global__ __launch_bounds__(256) void flatten_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                      const float* __restrict__ adj_table,
                                                      const uint8_t* __restrict__ chunk_active, uint32_t w, uint32_t h,
                                                      uint8_t* __restrict__ dst, const pfxk_preview PV, const pfxk_region RG)
{
    // GENERAL + RG.rw: only the dirty rectangle is composited, into a compact rw x rh destination (composite_dirty_readback,
    // src/gpu/renderer.rs:588); quads then run along the rectangle's rows
    const bool region = GENERAL && RG.rw != 0u;
    const uint32_t qpr = region ? (RG.rw + 3u) / 4u : 0u;
    const size_t n_quads = region ? (size_t)qpr * RG.rh : ((size_t)w * h + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * blockDim.x) {
        size_t p0 = q * 4, n_px = (size_t)w * h, o0 = q * 4; // n_px: exclusive bound of valid pixel indices for this quad
        if (region) {
            const uint32_t ry = (uint32_t)(q / qpr), qx = (uint32_t)(q - (size_t)ry * qpr);
            p0 = (size_t)(RG.y0 + ry) * w + RG.x0 + qx * 4u;
            n_px = (size_t)(RG.y0 + ry) * w + RG.x0 + RG.rw;
            o0 = (size_t)ry * RG.rw + qx * 4u;
        }
        const bool full = p0 + 4 <= n_px && !(region && (((p0 | o0) & 3u) != 0u)); // 16-byte loads / stores need aligned quads
        float acc[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f; // :573

        bool act[4] = {true, true, true, true};
        if (GENERAL && chunk_active) { // adjustment layers only touch chunks populated in some visible layer (:529-550)
            const uint32_t cxn = (w + 63u) / 64u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                size_t pi = p0 + p;
                if (pi >= n_px) pi = n_px - 1;
                uint32_t y = (uint32_t)(pi / w), x = (uint32_t)(pi - (size_t)y * w);
                act[p] = chunk_active[(y >> 6) * cxn + (x >> 6)] != 0;
            }
        }

        if (!GENERAL && full && n_layers > 0) {
            // raster-only stack: software-pipelined — layer k+1's descriptor (scalar load) and 16-byte pixel quad
            // are requested before layer k is blended, so the ~100 VALU ops of a blend cover the HBM latency
            pfxk_layer_desc L = layers[0];
            uint4 v = *reinterpret_cast<const uint4*>(L.pixels + p0 * 4);
            for (uint32_t li = 0; li < n_layers; ++li) {
                pfxk_layer_desc Ln = L;
                uint4 vn = v;
                if (li + 1 < n_layers) {
                    Ln = layers[li + 1];
                    vn = *reinterpret_cast<const uint4*>(Ln.pixels + p0 * 4);
                }
                const uint32_t top[4] = {v.x, v.y, v.z, v.w};
                // wave-level early-out: a wave whose 256 pixels are all transparent in this layer (sparse layers of real
                // documents; the TiledImage analogue is a missing chunk, canvas_state.rs:600) skips the blend entirely
                if (__any(((v.x | v.y | v.z | v.w) >> 24) != 0u)) {
                    if constexpr (F) blend_layer_fast<4>(L.mode, acc, top, L.opacity);
                    else blend4_dispatch<F>(L.mode, acc, top, L.opacity, rs_clamp(L.opacity, 0.0f, 1.0f));
                }
                L = Ln;
                v = vn;
            }
        } else
        for (uint32_t li = 0; li < n_layers; ++li) {
            const pfxk_layer_desc L = layers[li]; // uniform -> scalar loads
            if (GENERAL && L.kind != PFXK_LAYER_RASTER) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (act[p]) adjust_px(acc[p], L.kind, adj_table + L.adj_off, L.opacity);
                continue;
            }
            uint32_t top[4];
            if (full) {
                const uint4 v = *reinterpret_cast<const uint4*>(L.pixels + p0 * 4);
                top[0] = v.x; top[1] = v.y; top[2] = v.z; top[3] = v.w;
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    top[p] = (p0 + p < n_px) ? reinterpret_cast<const uint32_t*>(L.pixels)[p0 + p] : 0u;
            }
            if (GENERAL && PV.pixels && li == PV.active_pos) { // :593-597,621-658
                const uint32_t cxn = (w + 63u) / 64u;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const size_t pi = p0 + p;
                    if (pi < n_px) {
                        const uint32_t y = (uint32_t)(pi / w), x = (uint32_t)(pi - (size_t)y * w);
                        const uint32_t ci = (y >> 6) * cxn + (x >> 6);
                        if (PV.chunk_present[ci]) {
                            // a layer chunk that does not exist reads as (0,0,0,0), colour included (:613-617)
                            const uint32_t lp = PV.layer_chunk_present[ci] ? top[p] : 0u;
                            top[p] = preview_apply(lp, reinterpret_cast<const uint32_t*>(PV.pixels)[pi], PV);
                        }
                    }
                }
            }
            if (GENERAL && L.mask) {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (p0 + p < n_px) top[p] = apply_conceal(top[p], L.mask[p0 + p]);
            }
            if constexpr (F) blend_layer_fast<4>(L.mode, acc, top, L.opacity); // lanes past the image edge hold alpha 0: they only disable the opaque path
            else blend4_dispatch<F>(L.mode, acc, top, L.opacity, rs_clamp(L.opacity, 0.0f, 1.0f));
        }

        uint32_t out[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) out[p] = pack_rgba(acc[p][0], acc[p][1], acc[p][2], acc[p][3]);
        if (full) {
            *reinterpret_cast<uint4*>(dst + o0 * 4) = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (p0 + p < n_px) reinterpret_cast<uint32_t*>(dst)[o0 + p] = out[p];
        }
    }
}

// ---- the streaming compositor: raster-only stacks under the FAST precondition (every document without live masks,
// adjustment layers or a tool preview; BASELINE's 8K x 32 configuration) ----------------------------------------------------------
//   * a wave owns PX groups of 64 consecutive pixels; lane l handles pixel 64*j + l of group j, so every memory instruction of
//     the wave covers 256 contiguous bytes;
//   * a layer pixel is fetched with ONE typed buffer load (buffer_load_format_xyzw through an 8_8_8_8 UNORM resource): the
//     texture path delivers the four f32 RN(byte / 255) that blend_pixel_static starts from (k_blend.h:blend_nx), and its range
//     check returns (0,0,0,0) — "layer pixel transparent, keep the accumulator" — past the end of the image, so there is no
//     tail code; the result is written with range-checked dword buffer stores;
//   * the accumulator stays in registers as RN(k / 255) for the whole stack; layer k+1's PX pixels are in flight while layer k
//     is blended (two register sets, the loop is unrolled by two so no copies are needed);
//   * HBM traffic is the algorithmic minimum, 4 bytes per layer-pixel in and 4 bytes per pixel out.
// gfx9-family buffer resource (V#), stride 0 => offsets and num_records are bytes
enum : uint32_t {
    PFX_RSRC_UNORM8X4 = 0xFACu | (0u << 12) | (10u << 15), // dst_sel = x,y,z,w; num_format UNORM; data_format 8_8_8_8
    PFX_RSRC_RAW32 = 0xFACu | (7u << 12) | (4u << 15),     // num_format FLOAT, data_format 32 (untyped dword access)
    PFX_RSRC_ALPHA8 = 0xFAFu | (0u << 12) | (10u << 15)    // 8_8_8_8 UNORM with dst_sel_x = A: buffer_load_format_x returns alpha / 255
};
PFX_DEV pfx_v4i make_rsrc(const void* base, uint32_t bytes, uint32_t word3)
{
    const uint64_t a = (uint64_t)base;
    pfx_v4i r;
    r.x = (int)(uint32_t)a;
    r.y = (int)((uint32_t)(a >> 32) & 0xffffu);
    r.z = (int)bytes;
    r.w = (int)word3;
    return r;
}

// the same for a pointer that is known to be a canonical device address (bits 48 .. 63 clear: word 1's stride / swizzle fields stay zero without the mask)
PFX_DEV pfx_v4i make_rsrc_canonical(const void* base, uint32_t bytes, uint32_t word3)
{
    const uint64_t a = (uint64_t)base;
    pfx_v4i r;
    r.x = (int)(uint32_t)a;
    r.y = (int)(uint32_t)(a >> 32);
    r.z = (int)bytes;
    r.w = (int)word3;
    return r;
}

template <int PX>
PFX_DEV void stream_fetch(float (&t)[PX][4], const uint8_t* pixels, uint32_t bytes, int voff)
{
    const pfx_v4i rs = make_rsrc(pixels, bytes, PFX_RSRC_UNORM8X4);
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const pfx_v4f v = pfx_buffer_load_format_v4f32(rs, voff + j * 256, 0, 0);
        t[j][0] = v.x; t[j][1] = v.y; t[j][2] = v.z; t[j][3] = v.w;
    }
}

#ifndef PFX_STREAM_TWO_STAGE
#define PFX_STREAM_TWO_STAGE 1   // round 5, 9-layer stacks per mode on one box: -1 .. -8 % (Xor, Vivid Light, Overlay, Hard Light, Glow most), none slower (tools/lab/ab_ops_libs.sh)
#endif
template <int PX>
PFX_DEV void stream_layer(float (&acc)[PX][4], const float (&t)[PX][4], uint32_t mode, float opacity)
{
    const float amax = alpha_max<PX>(t);
    // a wave whose 64*PX pixels are all transparent in this layer (sparse layers of real documents; the TiledImage analogue
    // is a missing chunk, canvas_state.rs:600) skips the blend entirely
#if PFX_STREAM_TWO_STAGE
    if (__any(amax != 0.0f)) {
        const float amin = alpha_min<PX>(acc);   // every accumulator opaque: the class variants 3 / 4 of the two-stage form, else the general one
        blend_layer_nx_two_stage<PX>(mode, acc, t, opacity, __all(amin == 1.0f) ? (uint32_t)PX : 0u);
    }
#else
    if (__any(amax != 0.0f)) blend_layer_nx<PX>(mode, acc, t, opacity);
#endif
}

template <int PX, int NB, int MINW>
__global__ __launch_bounds__(256, MINW) void flatten_stream_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                                   uint32_t n_px, uint8_t* __restrict__ dst,
                                                                   const uint8_t* __restrict__ chunk_start, uint32_t w)
{
    // NB register sets hold the layer pixels of NB consecutive layers: while layer k is blended, layers k+1 .. k+NB-1 are in
    // flight (the typed load lands 16 bytes of registers per 4 bytes read, so the in-flight volume a CU needs to cover HBM
    // latency — ~32 KB — is bought with registers: PX * NB * 4 VGPRs per lane)
    static_assert(NB == 2 || NB == 3, "two or three layer register sets");
    soft_d_fill(threadIdx.x, 256u);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t n_waves = gridDim.x * 4u;
    const uint32_t n_tiles = (n_px + 64u * PX - 1u) / (64u * PX);
    const uint32_t bytes = n_px * 4u;
    const pfx_v4i rs_dst = make_rsrc(dst, bytes, PFX_RSRC_RAW32);
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const int voff = (int)((tile * (64u * PX) + lane) * 4u);
        float acc[PX][4], tA[PX][4], tB[PX][4], tC[NB == 3 ? PX : 1][4];
        uint32_t mA = 0, mB = 0, mC = 0;
        uint32_t oA = 0, oB = 0, oC = 0; // opacity bits: kept integer so the loop-carried copies stay in SGPRs
#pragma unroll
        for (int j = 0; j < PX; ++j) acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0f; // :573
        // Every fetch is unconditional (past the last layer it re-reads the last one, unused): with no load inside a conditional the
        // compiler's s_waitcnt bookkeeping is exact — vmcnt(PX * (NB - 1)) in front of a blend, NB - 1 layers genuinely in flight —
        // instead of the vmcnt(0) it falls back to when a prefetch sits behind a branch.  The descriptor (scalar loads) of the layer
        // fetched NEXT is requested one stage ahead, so no fetch waits for it either.
        const uint32_t last = n_layers - 1u;
        // Layers of the layer store carry per-chunk alpha summaries (pfxk_chunk_alpha_flags): `chunk_start[c]` is the topmost layer that
        // resets every pixel of 64 x 64 chunk c (an opaque Normal layer at 100 %, canvas_state.rs:1258, or an Overwrite layer without a
        // transparent pixel there, :1275) — whatever is below it cannot show.  The tile starts at the lowest such layer of the chunks it
        // touches: a photo layer over a stack costs nothing below the photo, and no pixel is read to find that out.
        uint32_t first = 0u;
        if (chunk_start) {
            const uint32_t cxn = (w + 63u) / 64u;
            uint32_t s = 255u;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const uint32_t p = tile * (64u * PX) + 64u * j + lane;
                if (p < n_px) { const uint32_t y = p / w, x = p - y * w; s = min(s, (uint32_t)chunk_start[(y >> 6) * cxn + (x >> 6)]); }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s = min(s, (uint32_t)__shfl_xor((int)s, off));
            first = min((uint32_t)__builtin_amdgcn_readfirstlane((int)s), last);
        }
        const uint8_t* npx = layers[first].pixels;
        uint32_t nmode = layers[first].mode;
        uint32_t nop = layers[first].adj_off; // raster layers: bits of the clamped opacity (pfx_kernels.h)
#define PFX_FETCH(T, M, O, K) { M = nmode; O = nop; stream_fetch<PX>(T, npx, bytes, voff); \
                                const uint32_t kn = ((K) + 1u < last) ? (K) + 1u : last; \
                                npx = layers[kn].pixels; nmode = layers[kn].mode; nop = layers[kn].adj_off; }
#define PFX_LAYER(T, M, O, K) if ((K) < n_layers) stream_layer<PX>(acc, T, M, __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(O)));
        PFX_FETCH(tA, mA, oA, first)
        if constexpr (NB == 2) {
            for (uint32_t li = first; li < n_layers; li += 2) {
                PFX_FETCH(tB, mB, oB, li + 1)
                PFX_LAYER(tA, mA, oA, li)
                PFX_FETCH(tA, mA, oA, li + 2)
                PFX_LAYER(tB, mB, oB, li + 1)
            }
        } else {
            PFX_FETCH(tB, mB, oB, first + 1u)
            for (uint32_t li = first; li < n_layers; li += 3) { // at the top: layers li (A) and li + 1 (B) are in flight or landed
                PFX_FETCH(tC, mC, oC, li + 2)
                PFX_LAYER(tA, mA, oA, li)
                PFX_FETCH(tA, mA, oA, li + 3)
                PFX_LAYER(tB, mB, oB, li + 1)
                PFX_FETCH(tB, mB, oB, li + 4)
                PFX_LAYER(tC, mC, oC, li + 2)
            }
        }
#undef PFX_LAYER
#undef PFX_FETCH
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            // bn = RN(k / 255)  =>  bn * 255 = k (1 + e), |e| < 2^-23: adding 0.5 and truncating recovers k
            const uint32_t px = (uint32_t)(acc[j][0] * 255.0f + 0.5f) | ((uint32_t)(acc[j][1] * 255.0f + 0.5f) << 8) |
                                ((uint32_t)(acc[j][2] * 255.0f + 0.5f) << 16) | ((uint32_t)(acc[j][3] * 255.0f + 0.5f) << 24);
            pfx_buffer_store_i32((int)px, rs_dst, voff + j * 256, 0, 0);
        }
    }
}

// ---- dead-layer elimination (canvas_state.rs:1253-1281) -------------------------------------------------------------------------
// blend_pixel_static has two results that do not depend on `base`: an Overwrite layer wherever its alpha is non-zero (:1275-1281, any
// opacity) and a Normal layer at opacity >= 1 wherever its alpha is 255 (:1258).  For a pixel, every layer below the topmost such
// "reset" layer L* is dead: whatever the accumulator holds is replaced at L*.  Skipping dead layers is bit-exact by construction;
// running some of them anyway is harmless, so every decision below only has to be conservative.
//
// The host lists up to 4 candidate layers (mode / opacity only, topmost first kept); the kernel finds L* per pixel from the
// candidates' alpha.  Real documents are spatially coherent (a photo layer covers a region): a 192-pixel unit whose pixels agree
// simply starts its layer loop at min L*.  Fine-grained mixtures (BASELINE's S2 stack: an Overwrite layer at depth 14 whose alpha
// is non-zero on a random 75 % of the pixels) need compaction, or every wave would still walk every layer for its unlucky lanes:
//   * a wave owns a contiguous stream of U units and a private LDS slice — no workgroup barrier anywhere;
//   * classifying a unit appends its EARLY pixels (L* below the unit's split layer r) to a FIFO of pixel offsets (ballot + mbcnt
//     prefix: a stable partition, the queue stays in address order so a gathered wave load touches a few neighbouring lines);
//   * whenever the queue holds 192 entries the wave runs layers [start, r) on them with full lanes (typed buffer loads with per-lane
//     offsets), and parks the accumulators as RGBA8 — the reference's own accumulator type — in an LDS ring at the pixels' natural slots;
//   * a unit whose early pixels are all through runs layers [r, n) in natural order (coalesced loads and stores), accumulators
//     picked up from the ring; the carry-over queue is what keeps the compacted rounds full whatever the early fraction is.
// HBM traffic: the candidates' alpha is read once more (+4 bytes per pixel and candidate actually inspected).
// work counters of the last launches (pfxk_flatten_dle_stats): [0] compacted rounds, [1] pixels in them, [2] layers x rounds,
// [3] natural units, [4] layers x natural units, [5] candidate alpha reads (units), [6] units that used the queue
__device__ unsigned long long g_dle_stats[16];

#ifndef PFX_DLE_SGPR_ATTR
#define PFX_DLE_SGPR_ATTR
#endif
struct dle_sched { uint32_t wavesA, UA, wavesB, UB, UC; }; // waves [0, wavesA) own UA units each, the next wavesB UB, the rest UC

template <int PX, int NB, int AUX = 0>
PFX_DEV void dle_layers(float (&acc)[PX][4], const pfxk_layer_desc* __restrict__ layers, uint32_t lb, uint32_t le, uint32_t bytes,
                        const int (&voff)[PX], bool noblend)
{
    // NB register sets rotate (the loop is unrolled NB times, so set indices are compile-time): NB - 1 layers are in flight while one is
    // blended; every fetch is unconditional (past the item's end it re-reads the last layer) so that the s_waitcnt counts stay exact
    float t[NB][PX][4];
    uint32_t m[NB], o[NB];
    const uint32_t last = le - 1u;
    const uint8_t* npx = layers[lb].pixels;
    uint32_t nmode = layers[lb].mode;
    uint32_t nop = layers[lb].adj_off; // raster layers: bits of the clamped opacity (pfx_kernels.h)
    auto fetch = [&](auto SET, uint32_t K) {
        constexpr int S = decltype(SET)::value;
        m[S] = nmode; o[S] = nop;
        const pfx_v4i rs = make_rsrc(npx, bytes, PFX_RSRC_UNORM8X4);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const pfx_v4f v = pfx_buffer_load_format_v4f32(rs, voff[j], 0, AUX);
            t[S][j][0] = v.x; t[S][j][1] = v.y; t[S][j][2] = v.z; t[S][j][3] = v.w;
        }
        const uint32_t kn = (K + 1u < last) ? K + 1u : last;
        npx = layers[kn].pixels; nmode = layers[kn].mode; nop = layers[kn].adj_off;
    };
    auto blend = [&](auto SET, uint32_t K) {
        constexpr int S = decltype(SET)::value;
        if (K < le) {
            if (noblend) { // diagnostic (pfx_tune "dle_stats" = 2): the load stream without the arithmetic — results are garbage
#pragma unroll
                for (int j = 0; j < PX; ++j) { acc[j][0] += t[S][j][0]; acc[j][1] += t[S][j][1]; acc[j][2] += t[S][j][2]; acc[j][3] = t[S][j][3]; }
            } else stream_layer<PX>(acc, t[S], m[S], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(o[S])));
        }
    };
    if constexpr (NB == 3) {
        fetch(std::integral_constant<int, 0>{}, lb);
        fetch(std::integral_constant<int, 1>{}, lb + 1u);
        for (uint32_t li = lb; li < le; li += 3) {
            fetch(std::integral_constant<int, 2>{}, li + 2); blend(std::integral_constant<int, 0>{}, li);
            fetch(std::integral_constant<int, 0>{}, li + 3); blend(std::integral_constant<int, 1>{}, li + 1);
            fetch(std::integral_constant<int, 1>{}, li + 4); blend(std::integral_constant<int, 2>{}, li + 2);
        }
    } else if constexpr (NB == 2) {
        fetch(std::integral_constant<int, 0>{}, lb);
        for (uint32_t li = lb; li < le; li += 2) {
            fetch(std::integral_constant<int, 1>{}, li + 1); blend(std::integral_constant<int, 0>{}, li);
            fetch(std::integral_constant<int, 0>{}, li + 2); blend(std::integral_constant<int, 1>{}, li + 1);
        }
    } else {
        static_assert(NB == 4, "two, three or four register sets");
        fetch(std::integral_constant<int, 0>{}, lb);
        fetch(std::integral_constant<int, 1>{}, lb + 1u);
        fetch(std::integral_constant<int, 2>{}, lb + 2u);
        for (uint32_t li = lb; li < le; li += 4) {
            fetch(std::integral_constant<int, 3>{}, li + 3); blend(std::integral_constant<int, 0>{}, li);
            fetch(std::integral_constant<int, 0>{}, li + 4); blend(std::integral_constant<int, 1>{}, li + 1);
            fetch(std::integral_constant<int, 1>{}, li + 5); blend(std::integral_constant<int, 2>{}, li + 2);
            fetch(std::integral_constant<int, 2>{}, li + 6); blend(std::integral_constant<int, 3>{}, li + 3);
        }
    }
}

PFX_DEV void wave_lds_sync()
{
    // LDS operations of one wave execute in order; this only keeps the compiler from moving them across the hand-over
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int PX, int RING_LOG2, int WPB, int NB = 3, int AUX = 0>
__global__ __launch_bounds__(64 * WPB) PFX_DLE_SGPR_ATTR void flatten_dle_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                                uint32_t n_px, uint8_t* __restrict__ dst, const pfxk_dle_cands C,
                                                                const dle_sched SC)
{
    constexpr uint32_t UPX = 64u * PX, QCAP = PX == 2 ? 256u : 512u, RING = 1u << RING_LOG2, R = RING / UPX, NREC = 16u;
    static_assert(R >= 2 && R <= NREC, "ring depth");
    __shared__ uint32_t s_acc[WPB][RING];   // parked accumulators (RGBA8) at the pixels' natural slots (offset mod RING), R units deep
    __shared__ uint16_t s_q[WPB][QCAP];     // FIFO of early pixels (offset from the wave's first pixel)
    __shared__ uint32_t s_rec[WPB][NREC][2]; // per unit in flight: {first layer of its natural pass, queue tail after its append}
    soft_d_fill(threadIdx.x, 64u * WPB);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x * (uint32_t)WPB + wid);
    const uint32_t units_total = (n_px + UPX - 1u) / UPX;
    // stream lengths shrink towards the end of the launch (workgroups start in index order): long streams waste the least on their
    // final partial round, short ones keep the tail of the launch short
    uint32_t u0, U;
    if (gw < SC.wavesA) { U = SC.UA; u0 = gw * SC.UA; }
    else if (gw < SC.wavesA + SC.wavesB) { U = SC.UB; u0 = SC.wavesA * SC.UA + (gw - SC.wavesA) * SC.UB; }
    else { U = SC.UC; u0 = SC.wavesA * SC.UA + SC.wavesB * SC.UB + (gw - SC.wavesA - SC.wavesB) * SC.UC; }
    if (u0 >= units_total) return;
    const uint32_t nu = min(U, units_total - u0);
    const uint32_t base_px = u0 * UPX;
    const uint32_t bytes = n_px * 4u;
    const pfx_v4i rs_dst = make_rsrc(dst, bytes, PFX_RSRC_RAW32);
    uint32_t* const acc_ring = s_acc[wid];
    uint16_t* const q = s_q[wid];
    uint32_t (*const rec)[2] = s_rec[wid];

    uint32_t cls_next = 0, nat_next = 0;          // units classified / finished so far (relative to u0)
    uint32_t q_head = 0, q_tail = 0;              // monotone counters; entry k lives at q[k % QCAP]
    uint32_t q_r = 0, q_start = 0;                // layers [q_start, q_r) are what the queued pixels still need
    uint32_t st_rounds = 0, st_rpx = 0, st_rlay = 0, st_nlay = 0, st_reads = 0, st_cunits = 0;
    uint32_t probe_fail = 0, skip_left = 0;      // classification back-off (below)
    for (;;) {
        const uint32_t q_cnt = q_tail - q_head;
        bool run_queue = q_cnt >= UPX, run_nat = false;
        uint32_t nat_start = 0;
        if (!run_queue && nat_next < cls_next) {
            const uint32_t slot = nat_next % NREC;
            const uint32_t need = __builtin_amdgcn_readfirstlane(rec[slot][1]);
            // `need` = queue position after this unit's entries; consumed when head has passed it (counters do not wrap: < 2^32 px)
            if (need <= q_head) { run_nat = true; nat_start = __builtin_amdgcn_readfirstlane(rec[slot][0]); }
        }
        if (!run_queue && !run_nat) {
            if (cls_next < nu && cls_next - nat_next < R) {
                // ---- classify unit cls_next ----
                const uint32_t u = cls_next;
                const uint32_t o0 = u * UPX + lane;
                uint32_t cls[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) cls[j] = 0u;
                // Reading a candidate's alpha costs as much HBM traffic as a layer: where the reset layers do not pay (say Normal layers at
                // 100 % whose pixels are mostly translucent) the stream stops asking — after two units in a row that saved nothing the next
                // 14 start at layer 0 unexamined, then two more are probed
                const bool probe = skip_left == 0u;
                if (!probe) skip_left -= 1u;
                bool done = !probe;
#pragma unroll
                for (int i = 3; i >= 0; --i) {
                    if ((uint32_t)i < C.n && !done) {
                        const pfx_v4i ra = make_rsrc(layers[C.layer[i]].pixels, bytes, PFX_RSRC_ALPHA8);
                        bool all_found = true;
#pragma unroll
                        for (int j = 0; j < PX; ++j) {
                            const float a = pfx_buffer_load_format_f32(ra, (int)((base_px + o0 + 64u * j) * 4u), 0, 0);
                            const bool hit = C.kind[i] ? (a == 1.0f) : (a != 0.0f);
                            cls[j] = (cls[j] == 0u && hit) ? (uint32_t)(i + 1) : cls[j];
                            all_found = all_found && cls[j] != 0u;
                        }
                        done = __all(all_found);
                        st_reads += 1u;
                    }
                }
                // cnt[i] = pixels of the unit whose reset class is >= i (cnt[0] = all); cmin = the class every pixel reaches
                uint32_t cnt[5] = {UPX, 0u, 0u, 0u, 0u};
#pragma unroll
                for (int i = 1; i <= 4; ++i)
#pragma unroll
                    for (int j = 0; j < PX; ++j) cnt[i] += (uint32_t)__popcll(__ballot(cls[j] >= (uint32_t)i));
                uint32_t cmin = 0;
#pragma unroll
                for (int i = 1; i <= 4; ++i) if (cnt[i] == UPX) cmin = (uint32_t)i;
                uint32_t lay[5] = {0u, C.layer[0], C.layer[1], C.layer[2], C.layer[3]};
                uint32_t s_u = 0;
#pragma unroll
                for (int i = 1; i <= 4; ++i) if (cmin == (uint32_t)i) s_u = lay[i];
                // split class: the candidate that saves the most layer-pixels, if at least 30 % of the unit skip something
                uint32_t best = 0, best_sav = 0, r = s_u;
#pragma unroll
                for (int i = 1; i <= 4; ++i) {
                    if ((uint32_t)i > cmin && (uint32_t)i <= C.n && cnt[i] * 10u >= UPX * 3u) {
                        const uint32_t sav = cnt[i] * (lay[i] - s_u);
                        if (sav > best_sav) { best_sav = sav; best = (uint32_t)i; r = lay[i]; }
                    }
                }
                if (best != 0u && q_cnt != 0u && q_r != r) { best = 0u; r = s_u; } // one split layer in the queue at a time
                if (probe) {
                    if (best != 0u || s_u != 0u) probe_fail = 0u;
                    else if (++probe_fail >= 2u) { probe_fail = 0u; skip_left = 14u; }
                }
                // natural slots start from the reference's initial accumulator (0,0,0,0), canvas_state.rs:573
#pragma unroll
                for (int j = 0; j < PX; ++j) acc_ring[(o0 + 64u * j) % RING] = 0u;
                if (best != 0u) {
                    st_cunits += 1u;
                    if (q_cnt == 0u) q_start = s_u; else q_start = min(q_start, s_u);
                    q_r = r;
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const bool early = cls[j] < best;
                        const uint64_t m = __ballot(early);
                        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (early) q[(q_tail + pre) % QCAP] = (uint16_t)(o0 + 64u * j);
                        q_tail += (uint32_t)__popcll(m);
                    }
                }
                if (lane == 0) { rec[u % NREC][0] = r; rec[u % NREC][1] = best != 0u ? q_tail : 0u; }
                wave_lds_sync();
                cls_next = u + 1u;
                continue;
            }
            if (q_cnt == 0u) {                    // everything classified, run and stored
                if (lane == 0 && (C.stats & 1u)) {
                    atomicAdd(&g_dle_stats[0], st_rounds); atomicAdd(&g_dle_stats[1], st_rpx); atomicAdd(&g_dle_stats[2], st_rlay);
                    atomicAdd(&g_dle_stats[3], nat_next); atomicAdd(&g_dle_stats[4], st_nlay); atomicAdd(&g_dle_stats[5], st_reads);
                    atomicAdd(&g_dle_stats[6], st_cunits);
                }
                break;
            }
            run_queue = true;                     // flush a partial round: the ring is full or the stream has ended
        }

        int voff[PX];
        float acc[PX][4];
        uint32_t lb, le;
        if (run_queue) {
            const uint32_t m = min(q_cnt, UPX);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const uint32_t k = 64u * j + lane;
                const uint32_t o = (uint32_t)q[(q_head + k) % QCAP];
                voff[j] = k < m ? (int)((base_px + o) * 4u) : (int)bytes; // past the end: the load returns (0,0,0,0), "transparent"
                acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0f;
            }
            lb = q_start; le = q_r;
            q_head += m;
            st_rounds += 1u; st_rpx += m; st_rlay += le - lb;
        } else {
            const uint32_t o0 = nat_next * UPX + lane;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                voff[j] = (int)((base_px + o0 + 64u * j) * 4u);
                const uint32_t p = acc_ring[(o0 + 64u * j) % RING];
                acc[j][0] = div255(ubyte0(p)); acc[j][1] = div255(ubyte1(p)); acc[j][2] = div255(ubyte2(p)); acc[j][3] = div255(ubyte3(p));
            }
            lb = nat_start; le = n_layers;
            st_nlay += le - lb;
        }
        dle_layers<PX, NB, AUX>(acc, layers, lb, le, bytes, voff, (C.stats & 2u) != 0u);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            // bn = RN(k / 255)  =>  bn * 255 = k (1 + e), |e| < 2^-23: adding 0.5 and truncating recovers k
            const uint32_t px = (uint32_t)(acc[j][0] * 255.0f + 0.5f) | ((uint32_t)(acc[j][1] * 255.0f + 0.5f) << 8) |
                                ((uint32_t)(acc[j][2] * 255.0f + 0.5f) << 16) | ((uint32_t)(acc[j][3] * 255.0f + 0.5f) << 24);
            if (run_queue) { if (voff[j] != (int)bytes) acc_ring[((uint32_t)voff[j] / 4u - base_px) % RING] = px; }
            else pfx_buffer_store_i32((int)px, rs_dst, voff[j], 0, 0);
        }
        if (run_queue) wave_lds_sync();
        else nat_next += 1u;
    }
}


#ifndef PFX_SRT_GROUPS
#define PFX_SRT_GROUPS 1   // development A/B (tools/build_variant.sh): 0 = the per-unit class test of round 3
#endif
#ifndef PFX_SRT_REDEAL
#define PFX_SRT_REDEAL 1   // development A/B: 0 = no re-deal code in the layer loop
#endif
#ifndef PFX_SRT_TYPED_STORE
#define PFX_SRT_TYPED_STORE 0 // development A/B: 1 = the result leaves through buffer_store_format_xyzw (float -> UNORM8 in the texture path, no pack arithmetic):
                              // 1.120-1.124 ms against 1.106-1.112 for four v_cvt_pk_u8_f32 and a dword store, two alternations on one box (profiles/r04_tuning.md)
#endif
#ifndef PFX_SRT_NATSTORE
#define PFX_SRT_NATSTORE 1 // development A/B: 0 = a dealt unit is stored in dealt order
#endif
template <int PX>
PFX_DEV void stream_layer_groups(float (&acc)[PX][4], const float (&t)[PX][4], uint32_t mode, float opacity, uint32_t lead)
{
    const float amax = alpha_max<PX>(t);
#if PFX_TWO_STAGE
    if (__any(amax != 0.0f)) blend_layer_nx_two_stage<PX>(mode, acc, t, opacity, lead);
#elif PFX_SRT_GROUPS
    if (__any(amax != 0.0f)) blend_layer_nx_groups<PX>(mode, acc, t, opacity, lead);
#else
    if (__any(amax != 0.0f)) blend_layer_nx<PX>(mode, acc, t, opacity);
#endif
}

// dle_layers for the class-sorting kernel (two register sets): blends layers [lb, le) and, in front of layers s1, s1 + seg, ..., re-deals the unit's
// pixels to the lanes when that completes another wave-uniform opaque group.  A re-deal moves the accumulators, the pixel offsets and the one layer
// that is already in flight (requested with the old offsets) through the LDS tile; the next request uses the new offsets.
// Scalar work per layer is kept short — the CU's one scalar unit serves all 24 waves, and round 4's counters show it more than half busy: the descriptor
// pointer advances instead of being indexed, nothing clamps it (the host appends PFXK_DESC_PAD copies of the last descriptor: pfx_api.cpp:build_stack; a pass reads up to descriptor le + 2; what is
// fetched through them or beyond `le` is never blended), the whole 32-byte descriptor comes with one scalar load, the resource needs no mask.
// NOBLEND (diagnostic, pfx_tune "dle_stats" = 2): the load stream without the arithmetic — results are garbage.
// TR (diagnostic build, pfx_tune "dle_stats" = 4): wave clock spent in front of each blend waiting for the layer's pixels (an explicit s_waitcnt bracketed by
// s_memtime) and inside the blends, accumulated into tr[0] / tr[1]
template <int PX, bool NOBLEND = false, bool TR = false>
PFX_DEV void srt_layers(float (&acc)[PX][4], const pfxk_layer_desc* __restrict__ layers, uint32_t lb, uint32_t le, uint32_t bytes, int (&voff)[PX],
                        uint32_t s1, uint32_t seg, float4* s_x, uint32_t* s_v, uint32_t& st_moves, unsigned long long* tr = nullptr)
{
    float t[2][PX][4];
    uint32_t m[2], o[2];
    const pfxk_layer_desc* nptr = layers + lb;
    pfxk_layer_desc nd = *nptr;    // raster layers: adj_off = bits of the clamped opacity (pfx_kernels.h)
    uint32_t next_attempt = lb < s1 ? s1 : lb + 1u;
    uint32_t lead = 0u;            // leading groups known to be opaque wave-wide
    bool recount = true;           // ... to be re-taken in front of the next blend (start of the pass; behind Xor / Overwrite, which can lower alpha)
    auto fetch = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        m[S] = nd.mode; o[S] = nd.adj_off;
        const pfx_v4i rs = make_rsrc_canonical(nd.pixels, bytes, PFX_RSRC_UNORM8X4);
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const pfx_v4f v = pfx_buffer_load_format_v4f32(rs, voff[j], 0, 0);
            t[S][j][0] = v.x; t[S][j][1] = v.y; t[S][j][2] = v.z; t[S][j][3] = v.w;
        }
        nptr += 1;
        nd = *nptr;
    };
    auto blend = [&](auto SET, uint32_t K) {
        constexpr int S = decltype(SET)::value;
        if (K < le) {
            unsigned long long c0 = 0, c1 = 0;
            if constexpr (TR) {
                c0 = __builtin_amdgcn_s_memtime();
                if constexpr (PX == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if constexpr (PX == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                c1 = __builtin_amdgcn_s_memtime();
                tr[0] += c1 - c0;
            }
            if constexpr (NOBLEND) {
#pragma unroll
                for (int j = 0; j < PX; ++j) { acc[j][0] += t[S][j][0]; acc[j][1] += t[S][j][1]; acc[j][2] += t[S][j][2]; acc[j][3] = t[S][j][3]; }
            } else if constexpr (PX == 1 && !PFX_TWO_STAGE) {
                stream_layer<1>(acc, t[S], m[S], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(o[S])));
            } else {
                if (recount) { lead = count_lead<PX>(acc); recount = false; }
                stream_layer_groups<PX>(acc, t[S], m[S], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(o[S])), lead);
                recount = m[S] == M_XOR || m[S] == M_OVERWRITE;
            }
            if constexpr (TR) {
                float sink = 0.0f;                      // the blend's results must exist before the clock is read
#pragma unroll
                for (int j = 0; j < PX; ++j) sink += acc[j][0] + acc[j][3];
                asm volatile("" :: "v"(sink));
                tr[1] += __builtin_amdgcn_s_memtime() - c1;
            }
        }
    };
    auto redeal = [&](auto SET, uint32_t K) {          // in front of layer K, whose pixels are in flight in set SET
        constexpr int S = decltype(SET)::value;
#if !PFX_SRT_REDEAL
        return;
#endif
        if constexpr (PX == 1 || NOBLEND) return;
        if (K != next_attempt || K >= le) return;
        next_attempt = K + seg;
        uint64_t mo[PX];
        uint32_t co[PX], total_o = 0u, now = 0u;
        bool run = true;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            mo[j] = __ballot(acc[j][3] == 1.0f);
            co[j] = (uint32_t)__popcll(mo[j]);
            total_o += co[j];
            run = run && co[j] == 64u;
            now += run ? 1u : 0u;
        }
        lead = now; recount = false;                   // the count is a by-product of the attempt
        if (total_o / 64u <= now) return;              // grouping the opaque accumulators would not complete another leading group
        lead = total_o / 64u;
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); // recomputed: not worth a register across the blends
        uint32_t slot[PX], pre_o = 0u, pre_n = total_o;
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const uint32_t rank_o = __builtin_amdgcn_mbcnt_hi((uint32_t)(mo[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mo[j], 0u));
            slot[j] = acc[j][3] == 1.0f ? pre_o + rank_o : pre_n + (lane - rank_o);
            pre_o += co[j]; pre_n += 64u - co[j];
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) { s_x[slot[j]] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]); s_v[slot[j]] = (uint32_t)voff[j]; }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const float4 a = s_x[64u * j + lane];
            acc[j][0] = a.x; acc[j][1] = a.y; acc[j][2] = a.z; acc[j][3] = a.w;
            voff[j] = (int)s_v[64u * j + lane];
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < PX; ++j) s_x[slot[j]] = make_float4(t[S][j][0], t[S][j][1], t[S][j][2], t[S][j][3]);
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const float4 a = s_x[64u * j + lane];
            t[S][j][0] = a.x; t[S][j][1] = a.y; t[S][j][2] = a.z; t[S][j][3] = a.w;
        }
        wave_lds_sync();
        st_moves += 1u;
    };
    fetch(std::integral_constant<int, 0>{});
    for (uint32_t li = lb; li < le; li += 2) {
        fetch(std::integral_constant<int, 1>{}); blend(std::integral_constant<int, 0>{}, li); redeal(std::integral_constant<int, 1>{}, li + 1);
        fetch(std::integral_constant<int, 0>{}); blend(std::integral_constant<int, 1>{}, li + 1); redeal(std::integral_constant<int, 0>{}, li + 2);
    }
}

// The early groups' pass (one pixel per lane, layers [lb, le) below the split): a set is four registers here, so NB sets keep NB - 1 layers in flight for the
// price of two of the natural pass's.  The pass is short on arithmetic (a third of a natural step) and long on memory time — the group's lanes sit on a random
// quarter of the unit's pixels, so every layer's load still touches all of the unit's cache lines — i.e. it is paced by how many loads a wave has in flight.
// Descriptors are read ahead without clamping like srt_layers: up to descriptor le + 2 NB - 2 (the host pads the table: PFXK_DESC_PAD).
template <int NB, bool NOBLEND = false, bool TR = false>
PFX_DEV void srt_early(float (&acc)[1][4], const pfxk_layer_desc* __restrict__ layers, uint32_t lb, uint32_t le, uint32_t bytes, int voff, unsigned long long* tr = nullptr)
{
    static_assert(NB >= 2 && 2 * NB - 2 < PFXK_DESC_PAD, "descriptor padding"); // valid indices end at n + PFXK_DESC_PAD - 1
    float t[NB][1][4];
    uint32_t m[NB], o[NB];
    const pfxk_layer_desc* nptr = layers + lb;
    pfxk_layer_desc nd = *nptr;
    uint32_t lead = 0u;
    bool recount = true;
    auto fetch = [&](auto SET) {
        constexpr int S = decltype(SET)::value;
        m[S] = nd.mode; o[S] = nd.adj_off;
        const pfx_v4i rs = make_rsrc_canonical(nd.pixels, bytes, PFX_RSRC_UNORM8X4);
        const pfx_v4f v = pfx_buffer_load_format_v4f32(rs, voff, 0, 0);
        t[S][0][0] = v.x; t[S][0][1] = v.y; t[S][0][2] = v.z; t[S][0][3] = v.w;
        nptr += 1;
        nd = *nptr;
    };
    auto blend = [&](auto SET, uint32_t K) {
        constexpr int S = decltype(SET)::value;
        if (K < le) {
            unsigned long long c0 = 0, c1 = 0;
            if constexpr (TR) {
                c0 = __builtin_amdgcn_s_memtime();
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NB - 1) : "memory");
                c1 = __builtin_amdgcn_s_memtime();
                tr[0] += c1 - c0;
            }
            if constexpr (NOBLEND) { acc[0][0] += t[S][0][0]; acc[0][1] += t[S][0][1]; acc[0][2] += t[S][0][2]; acc[0][3] = t[S][0][3]; }
            else {
                if (recount) { lead = count_lead<1>(acc); recount = false; }
                stream_layer_groups<1>(acc, t[S], m[S], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(o[S])), lead);
                recount = true; // one compare: cheaper than tracking which modes can change a single group's class
            }
            if constexpr (TR) {
                const float sink = acc[0][0] + acc[0][3];
                asm volatile("" :: "v"(sink));
                tr[1] += __builtin_amdgcn_s_memtime() - c1;
            }
        }
    };
    auto seq = [&](auto self, auto I, uint32_t li) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < NB) {
            fetch(std::integral_constant<int, (i + NB - 1) % NB>{});
            blend(std::integral_constant<int, i>{}, li + (uint32_t)i);
            self(self, std::integral_constant<int, i + 1>{}, li);
        }
    };
    auto pre = [&](auto self, auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < NB - 1) { fetch(std::integral_constant<int, i>{}); self(self, std::integral_constant<int, i + 1>{}); }
    };
    pre(pre, std::integral_constant<int, 0>{});
    for (uint32_t li = lb; li < le; li += (uint32_t)NB) seq(seq, std::integral_constant<int, 0>{}, li);
}
// Round 6: the same pass on RAW dword loads (one register per layer in flight, 2 / 3 / 5 / 7 layers in flight, converted on consumption) was built
// bit-exact and measured 2 ... 10 % SLOWER (profiles/r06_early_raw_ab.txt, profiles/r06_tuning.md): the waves wait less, the launch takes longer — the 12 conversion
// instructions per early pixel cost more than the deeper queue buys.  Code removed; typed loads stay.
#ifndef PFX_EARLY_NB
#define PFX_EARLY_NB 2   // 8K x 32 layers (S2), one box: 2 sets 0.990-1.010 ms, 4 sets 1.013-1.018, 6 / 8 sets 1.04-1.05, srt_layers<1> 1.044-1.048 (profiles/r05_tuning.md)
#endif

// ---- class sorting inside a unit (round 4) -------------------------------------------------------------------------------------------------------
// Two per-pixel properties decide how much a layer costs a pixel: whether the layer is dead for it (below its topmost reset layer: dead-layer
// elimination, above) and whether its accumulator is opaque (out_a == 1: no division, no base-alpha products, no alpha re-quantisation, no select for
// a transparent layer pixel — about 60 % of the general blend: k_blend.h, OB = 1).  Both only pay when a whole 64-lane group agrees, and on per-pixel-random
// alpha (BASELINE's S2) no group of consecutive pixels ever does.  This kernel re-deals the pixels of ONE unit (64 PX consecutive pixels) to the lanes so
// that groups agree — a stable partition by ballot + mbcnt ranks; accumulators and pixel offsets move through a 3.75 KB LDS tile as 16-byte and 4-byte
// items, no arithmetic.  Lane l, group j then holds SOME pixel of the unit: the wave's loads touch the same cache lines in another lane order, which
// the texture path serves at 98 % of the lane-order rate (tools/lab/perm_load.hip: 4.62 against 4.72 TB/s), whereas gathering across four units — what
// round 3's FIFO of early pixels does, and what a FIFO-per-class version of this kernel did with twice the HBM traffic — runs at 2.55 TB/s.
//   * early pixels first: a unit whose pixels disagree about their reset layer is dealt with the EARLY pixels (reset below the split layer r) in the
//     leading group(s); layers [start, r) then run on those groups alone (one group at a time through the one-pixel-per-lane layer loop), the other
//     groups' pixels start at r from (0,0,0,0) as before.  No queue across units, no parked accumulators, no partially filled rounds;
//   * opaque accumulators first: in front of layers s1, s1 + seg, ... of the natural pass (srt_layers) the unit is re-dealt when that completes
//     another wave-uniform opaque group; opaque accumulators stay opaque under every mode but Xor and Overwrite, so at most PX re-deals per unit.
// Every decision is a performance hint: a pixel runs each of its layers exactly once, in order, with arithmetic chosen by a per-group test of the
// actual accumulators.  The unit goes back to lane order and leaves as four v_cvt_pk_u8_f32 and a dword store per pixel (the typed format
// store, float -> UNORM8 in the texture path, measured 1 % slower: PFX_SRT_TYPED_STORE; the context's one-time check of the texture path's conversions,
// pfxk_unorm_store_check, still gates this kernel — it reads every layer through the same conversions).
struct dle_plan { uint32_t s1, seg; }; // re-deal attempts in front of layers s1, s1 + seg, s1 + 2 seg, ... (seg == 0: none)

// 63 VGPRs and no spills once the uniform regions are left unstructurized (Makefile: FLAGS_k_flatten): the eighth wave per SIMD
#ifndef PFX_SRT_PRIO
#define PFX_SRT_PRIO 0   // development A/B: bits 0-1 = wave priority in the early passes, bits 2-3 = in the natural pass
#endif
#ifndef PFX_SRT_WAVES
#define PFX_SRT_WAVES 8
#endif
#if PFX_SRT_WAVES == 8
#define PFX_SRT_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))
#elif PFX_SRT_WAVES == 7
#define PFX_SRT_ATTR __attribute__((amdgpu_waves_per_eu(7, 7)))
#else
#define PFX_SRT_ATTR
#endif
template <int PX, bool NOBLEND = false, bool TR = false>
__global__ __launch_bounds__(64) PFX_SRT_ATTR void flatten_srt_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers, uint32_t n_px,
                                                                         uint8_t* __restrict__ dst, const pfxk_dle_cands C, const dle_sched SC,
                                                                         const dle_plan P)
{
    constexpr uint32_t UPX = 64u * PX;
    __shared__ float4 s_x[UPX];             // re-deal tile: accumulators (or the layer in flight) ...
    __shared__ uint32_t s_v[UPX];           // ... and pixel byte offsets, at their new slots
    soft_d_fill(threadIdx.x, 64u);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gw = __builtin_amdgcn_readfirstlane(blockIdx.x);
    const uint32_t units_total = (n_px + UPX - 1u) / UPX;
    uint32_t u0, U;
    if (gw < SC.wavesA) { U = SC.UA; u0 = gw * SC.UA; }
    else if (gw < SC.wavesA + SC.wavesB) { U = SC.UB; u0 = SC.wavesA * SC.UA + (gw - SC.wavesA) * SC.UB; }
    else { U = SC.UC; u0 = SC.wavesA * SC.UA + SC.wavesB * SC.UB + (gw - SC.wavesA - SC.wavesB) * SC.UC; }
    if (u0 >= units_total) return;
    const uint32_t nu = min(U, units_total - u0);
    const uint32_t base_px = u0 * UPX;
    const uint32_t bytes = n_px * 4u;
#if PFX_SRT_TYPED_STORE
    const pfx_v4i rs_acc = make_rsrc(dst, bytes, PFX_RSRC_UNORM8X4);
#else
    const pfx_v4i rs_dst = make_rsrc(dst, bytes, PFX_RSRC_RAW32);
#endif
    const uint32_t s1 = P.seg != 0u ? P.s1 : 0xFFFFFFFFu;
    uint32_t st_egroups = 0, st_elay = 0, st_nlay = 0, st_reads = 0, st_cunits = 0, st_moves = 0;
    uint32_t probe_fail = 0, skip_left = 0;      // classification back-off (flatten_dle_kernel)
    // TR: [0] classification, [1] deal + early passes, [2] natural pass, [3] back to lane order + store; waits / blends of the early ([4], [5]) and natural ([6], [7]) passes
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tc0 = 0;
    const unsigned long long t_birth = TR ? __builtin_amdgcn_s_memtime() : 0ull;
    for (uint32_t u = 0; u < nu; ++u) {
        if constexpr (TR) tc0 = __builtin_amdgcn_s_memtime();
        int voff[PX];
        float acc[PX][4];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            voff[j] = (int)((base_px + u * UPX + 64u * j + lane) * 4u);
            acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0.0f;    // :573
        }
        // ---- classify: each pixel's topmost reset candidate (flatten_dle_kernel's rules) ----
        uint32_t cls[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) cls[j] = 0u;
        const bool probe = skip_left == 0u;
        if (!probe) skip_left -= 1u;
        bool done = !probe;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            if ((uint32_t)i < C.n && !done) {
                const pfx_v4i ra = make_rsrc(layers[C.layer[i]].pixels, bytes, PFX_RSRC_ALPHA8);
                bool all_found = true;
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const float a = pfx_buffer_load_format_f32(ra, voff[j], 0, 0);
                    const bool hit = C.kind[i] ? (a == 1.0f) : (a != 0.0f);
                    cls[j] = (cls[j] == 0u && hit) ? (uint32_t)(i + 1) : cls[j];
                    all_found = all_found && cls[j] != 0u;
                }
                done = __all(all_found);
                st_reads += 1u;
            }
        }
        uint32_t cn[5] = {UPX, 0u, 0u, 0u, 0u};  // cn[i] = pixels of the unit whose reset class is >= i
#pragma unroll
        for (int i = 1; i <= 4; ++i)
#pragma unroll
            for (int j = 0; j < PX; ++j) cn[i] += (uint32_t)__popcll(__ballot(cls[j] >= (uint32_t)i));
        uint32_t cmin = 0;
#pragma unroll
        for (int i = 1; i <= 4; ++i) if (cn[i] == UPX) cmin = (uint32_t)i;
        const uint32_t lay[5] = {0u, C.layer[0], C.layer[1], C.layer[2], C.layer[3]};
        uint32_t s_u = 0;
#pragma unroll
        for (int i = 1; i <= 4; ++i) if (cmin == (uint32_t)i) s_u = lay[i];
        // split class: the candidate whose early pixels (class below it) fit the fewest groups for the most layers
        uint32_t best = 0, best_sav = 0, r = s_u, eg = 0, cn_best = 0; // cn_best = cn[best], carried along: indexing cn[] with `best` put the array in
#pragma unroll                                                   // scratch, and its 32 bytes per lane and unit were written through to HBM (WRITE_SIZE 354 MB for a 133 MB frame)
        for (int i = 1; i <= 4; ++i) {
            if ((uint32_t)i > cmin && (uint32_t)i <= C.n) {
                const uint32_t g = (UPX - cn[i] + 63u) / 64u;          // groups the early pixels of this split need
                const uint32_t sav = ((uint32_t)PX - g) * (lay[i] - s_u);
                if (sav > best_sav) { best_sav = sav; best = (uint32_t)i; r = lay[i]; eg = g; cn_best = cn[i]; }
            }
        }
        if (probe) {
            if (best != 0u || s_u != 0u) probe_fail = 0u;
            else if (++probe_fail >= 2u) { probe_fail = 0u; skip_left = 14u; }
        }
        if constexpr (TR) { const unsigned long long c = __builtin_amdgcn_s_memtime(); tph[0] += c - tc0; tc0 = c; }
        if (best != 0u) {
            // ---- early pixels to the leading groups (the accumulators are all (0,0,0,0): only the offsets move), then their layers [s_u, r) ----
            st_cunits += 1u;
            uint32_t pre_e = 0u, pre_l = UPX - cn_best;
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const bool early = cls[j] < best;
                const uint64_t m = __ballot(early);
                const uint32_t rank_e = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t c = (uint32_t)__popcll(m);
                s_v[early ? pre_e + rank_e : pre_l + (lane - rank_e)] = (uint32_t)voff[j];
                pre_e += c; pre_l += 64u - c;
            }
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < PX; ++j) voff[j] = (int)s_v[64u * j + lane];
            wave_lds_sync();
            for (uint32_t g = 0; g < eg; ++g) {       // eg < PX: one group at a time through the one-pixel-per-lane loop (one call site)
                int v1[1];
                float a1[1][4] = {{0.0f, 0.0f, 0.0f, 0.0f}};
                v1[0] = voff[0];
#pragma unroll
                for (int j = 1; j < PX; ++j) v1[0] = g == (uint32_t)j ? voff[j] : v1[0];
                uint32_t none = 0u;
#if PFX_SRT_PRIO
                __builtin_amdgcn_s_setprio(PFX_SRT_PRIO & 3);
#endif
#if PFX_EARLY_NB > 0
                (void)none;
                srt_early<PFX_EARLY_NB, NOBLEND, TR>(a1, layers, s_u, r, bytes, v1[0], tph + 4);
#else
                srt_layers<1, NOBLEND, TR>(a1, layers, s_u, r, bytes, v1, 0xFFFFFFFFu, 0u, s_x, s_v, none, tph + 4);
#endif
#pragma unroll
                for (int j = 0; j < PX; ++j)
                    if (g == (uint32_t)j) { acc[j][0] = a1[0][0]; acc[j][1] = a1[0][1]; acc[j][2] = a1[0][2]; acc[j][3] = a1[0][3]; }
            }
            st_egroups += eg; st_elay += eg * (r - s_u);
        }
        // ---- natural pass: every group from r on, re-dealt by accumulator class on the way ----
        st_nlay += n_layers - r;
        const uint32_t moves_before = st_moves;
        if constexpr (TR) { const unsigned long long c = __builtin_amdgcn_s_memtime(); tph[1] += c - tc0; tc0 = c; }
#if PFX_SRT_PRIO
        __builtin_amdgcn_s_setprio((PFX_SRT_PRIO >> 2) & 3);
#endif
        srt_layers<PX, NOBLEND, TR>(acc, layers, r, n_layers, bytes, voff, s1, P.seg, s_x, s_v, st_moves, tph + 6);
        if constexpr (TR) { const unsigned long long c = __builtin_amdgcn_s_memtime(); tph[2] += c - tc0; tc0 = c; }
        // A dealt unit goes back to lane order before it is stored (three 16-byte LDS writes and reads per unit): whole-line stores instead of three
        // stores that each cover a third of every 64-byte piece.  Worth ~1 % of the step on one box (profiles/r04_tuning.md), nothing in WRITE_SIZE.
        if (PFX_SRT_NATSTORE && (best != 0u || st_moves != moves_before)) {
            const uint32_t ub = (base_px + u * UPX) * 4u;
#pragma unroll
            for (int j = 0; j < PX; ++j) s_x[((uint32_t)voff[j] - ub) >> 2] = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const float4 a = s_x[64u * j + lane];
                acc[j][0] = a.x; acc[j][1] = a.y; acc[j][2] = a.z; acc[j][3] = a.w;
                voff[j] = (int)(ub + (64u * j + lane) * 4u);
            }
            wave_lds_sync();
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
#if PFX_SRT_TYPED_STORE
            pfx_v4f v; v.x = acc[j][0]; v.y = acc[j][1]; v.z = acc[j][2]; v.w = acc[j][3];
            pfx_buffer_store_format_v4f32(v, rs_acc, voff[j], 0, 0);
#else
            // bn = RN(k / 255)  =>  bn * 255 = k (1 + e), |e| < 2^-23: v_cvt_pk_u8_f32 (round to nearest, saturating) recovers k
            uint32_t px = 0u;
#pragma unroll
            for (int c = 0; c < 4; ++c) px = __builtin_amdgcn_cvt_pk_u8_f32(acc[j][c] * 255.0f, c, px);
            pfx_buffer_store_i32((int)px, rs_dst, voff[j], 0, 0);
#endif
        }
        if constexpr (TR) { const unsigned long long c = __builtin_amdgcn_s_memtime(); tph[3] += c - tc0; tc0 = c; }
    }
    if constexpr (TR) {
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(&g_dle_stats[8 + i], tph[i]);
            atomicAdd(&g_dle_stats[0], __builtin_amdgcn_s_memtime() - t_birth); // wave lifetime (the work counters are not taken in this build)
            atomicAdd(&g_dle_stats[1], 1ull);
        }
        return;
    }
    if (lane == 0 && (C.stats & 1u)) {
        // [0] early groups run, [1] pixels in them (64 each), [2] layers x early groups, [3] units, [4] layers x units of the natural passes,
        // [5] candidate alpha reads (units), [6] units that split, [7] re-deals by accumulator class
        atomicAdd(&g_dle_stats[0], st_egroups); atomicAdd(&g_dle_stats[1], st_egroups * 64u); atomicAdd(&g_dle_stats[2], st_elay);
        atomicAdd(&g_dle_stats[3], nu); atomicAdd(&g_dle_stats[4], st_nlay); atomicAdd(&g_dle_stats[5], st_reads);
        atomicAdd(&g_dle_stats[6], st_cunits); atomicAdd(&g_dle_stats[7], st_moves);
    }
}

// Does dead-layer elimination pay on this stack?  (Stacks below the depth threshold: the class-sorting kernel wins where reset layers are spatially coherent — an
// opaque photo layer covers whole units, which then start at that layer and read nothing below — and loses on per-pixel-random alpha, where every unit is split.)
// One workgroup samples 256 units spread over the image against the topmost candidate and writes its verdict to pinned host memory, where a LATER composite of
// the same stack reads it (pfx_api.cpp: flatten_common; the decision is a performance hint, every kernel is bit-exact).
__global__ __launch_bounds__(1024) void dle_probe_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t cand_layer, uint32_t cand_kind, uint32_t n_px,
                                                         uint32_t* __restrict__ verdict_pinned, uint32_t tag)
{
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0u;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t units = (n_px + 191u) / 192u, bytes = n_px * 4u;
    const pfx_v4i ra = make_rsrc(layers[cand_layer].pixels, bytes, PFX_RSRC_ALPHA8);
    uint32_t good = 0u;
#pragma unroll 4
    for (uint32_t k = 0; k < 16u; ++k) {
        const uint32_t u = (uint32_t)(((uint64_t)(wave * 16u + k) * units) / 256u);   // 256 sample units, evenly spaced
        bool all_hit = true;
#pragma unroll
        for (uint32_t j = 0; j < 3u; ++j) {
            const uint32_t px = u * 192u + 64u * j + lane;
            const float a = pfx_buffer_load_format_f32(ra, (int)(px * 4u), 0, 0);       // past the image: 0 = "transparent"; the last unit may count as mixed
            all_hit = all_hit && (px >= n_px || (cand_kind ? (a == 1.0f) : (a != 0.0f)));
        }
        good += __all(all_hit) ? 1u : 0u;
    }
    if (lane == 0) atomicAdd(&s_cnt, good);
    __syncthreads();
    // at least half of the units start at the candidate outright: the elimination kernel skips everything below it there and pays its classification elsewhere
    if (threadIdx.x == 0) *verdict_pinned = tag | (s_cnt * 2u >= 256u ? 0x80000000u : 0u);
}

// float -> UNORM8 conversion of the typed store / UNORM8 -> float of the typed load against the arithmetic the kernels assume: for every byte
// value k and channel, storing RN(k / 255) must write k and loading k must return RN(k / 255).  out[0] += mismatches
__global__ __launch_bounds__(256) void unorm_store_check_kernel(uint8_t* __restrict__ scratch /* 1024 bytes */, unsigned long long* out)
{
    const uint32_t k = threadIdx.x;
    const float bn = div255((float)k);
    const pfx_v4i rs = make_rsrc(scratch, 1024u, PFX_RSRC_UNORM8X4);
    pfx_v4f v; v.x = bn; v.y = div255((float)(255u - k)); v.z = div255((float)((k * 7u) & 255u)); v.w = bn;
    pfx_buffer_store_format_v4f32(v, rs, (int)(k * 4u), 0, 0);
    __syncthreads();
    const uint32_t raw = reinterpret_cast<const volatile uint32_t*>(scratch)[k];
    const uint32_t want = k | ((255u - k) << 8) | (((k * 7u) & 255u) << 16) | (k << 24);
    const pfx_v4f b = pfx_buffer_load_format_v4f32(rs, (int)(k * 4u), 0, 0);
    const bool ok = raw == want && __builtin_bit_cast(uint32_t, b.x) == __builtin_bit_cast(uint32_t, v.x) &&
                    __builtin_bit_cast(uint32_t, b.y) == __builtin_bit_cast(uint32_t, v.y) &&
                    __builtin_bit_cast(uint32_t, b.z) == __builtin_bit_cast(uint32_t, v.z) && __builtin_bit_cast(uint32_t, b.w) == __builtin_bit_cast(uint32_t, v.w);
    if (!ok) atomicAdd(out, 1ull);
}

// Development knobs (pfx_tune): PROCESS-WIDE on purpose — they select among bit-identical kernel shapes for A/B runs and tests, not
// per-document behaviour; atomics because batch workers and device groups run one context per thread.
std::atomic<int> g_dle_stats_on{0}, g_dle_cfg{0}, g_dle_sched{1}, g_dle_fracA{75}, g_dle_fracB{20};
std::atomic<int> g_dle_units{0}; // units per wave (0 = default)
std::atomic<int> g_dle_kernel{0}; // 0 = class sorting inside a unit (flatten_srt_kernel), 1 = round 3's flatten_dle_kernel
std::atomic<int> g_dle_s1{-1}, g_dle_s2{-1}; // re-deal plan: first attempt this many layers above the topmost candidate (-1: 1; 0: never), then every g_dle_s2 layers (-1: 3)
std::atomic<int> g_flatten_variant{0}; // tuning knob (pfxk_flatten_set_variant): 0 = shipped (2 px x 2 sets, grid stride up to 16 layers), 1-5 = PX / register-set variants, 6 = 3 px x 3 sets (the round-2 shape), +10 = grid-stride launch, 8 = no elimination kernel, 9 = the general kernel

// per 64 x 64 chunk of a layer: bit 0 = every alpha is 255, bit 1 = no alpha is 0 (computed when a layer enters the layer store)
__global__ __launch_bounds__(256) void chunk_alpha_flags_kernel(const uint8_t* __restrict__ px, uint32_t w, uint32_t h, uint32_t cx0, uint32_t cy0,
                                                                uint32_t ncx, uint8_t* __restrict__ flags)
{
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t cx = cx0 + blockIdx.x % ncx, cy = cy0 + blockIdx.x / ncx;
    const uint32_t bx = cx * 64u, by = cy * 64u;
    const uint32_t cw = min(64u, w - bx), ch = min(64u, h - by);
    int not_opaque = 0, has_zero = 0;
    for (uint32_t i = threadIdx.x; i < cw * ch; i += blockDim.x) {
        const uint32_t a = px[((size_t)(by + i / cw) * w + bx + i % cw) * 4 + 3];
        not_opaque |= a != 255u;
        has_zero |= a == 0u;
    }
    const int any_no = __syncthreads_or(not_opaque), any_zero = __syncthreads_or(has_zero);
    if (threadIdx.x == 0) flags[cy * cxn + cx] = (uint8_t)((any_no ? 0u : 1u) | (any_zero ? 0u : 2u));
}

// start[c] = topmost layer that resets the whole chunk c (0 if none): want[k] = which flag bit layer k needs (0: never), flags[k] = its summary
__global__ __launch_bounds__(256) void chunk_start_kernel(const uint8_t* const* __restrict__ flags, const uint8_t* __restrict__ want, uint32_t n_layers,
                                                          uint32_t n_chunks, uint8_t* __restrict__ start, uint32_t* __restrict__ useful, uint32_t tag)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    uint32_t s = 0u;
    for (uint32_t k = 1; k < n_layers; ++k)
        if (want[k] && flags[k] && (flags[k][c] & want[k])) s = k;
    start[c] = (uint8_t)min(s, 254u);
    // tells the host (pinned memory, read on a LATER call) whether the table skips anything at all: a stack whose table is all zeros
    // is composited without it from then on (the lookup costs the plain kernel ~7 %)
    if (s != 0u && useful) *useful = tag;
}

// chunk activity = union over visible raster layers of "chunk has any alpha != 0" (canvas_state.rs:529-550)
__global__ __launch_bounds__(256) void chunk_active_kernel(const pfxk_layer_desc* __restrict__ layers, uint32_t n_layers,
                                                           uint32_t w, uint32_t h, uint8_t* __restrict__ chunk_active,
                                                           const uint8_t* __restrict__ preview_present)
{
    const uint32_t cxn = (w + 63u) / 64u;
    const uint32_t cx = blockIdx.x % cxn, cy = blockIdx.x / cxn;
    const uint32_t bx = cx * 64u, by = cy * 64u;
    const uint32_t cw = min(64u, w - bx), ch = min(64u, h - by);
    int any = preview_present ? (preview_present[blockIdx.x] != 0) : 0; // the preview's chunk keys join the set (:541-548)
    for (uint32_t li = 0; li < n_layers && !any; ++li) {
        const pfxk_layer_desc L = layers[li];
        if (L.kind != PFXK_LAYER_RASTER || !L.pixels) continue;
        int mine = 0;
        for (uint32_t i = threadIdx.x; i < cw * ch; i += blockDim.x) {
            uint32_t lx = i % cw, ly = i / cw;
            mine |= L.pixels[((size_t)(by + ly) * w + bx + lx) * 4 + 3] != 0;
        }
        any = __syncthreads_or(mine);
    }
    if (threadIdx.x == 0) chunk_active[blockIdx.x] = (uint8_t)(any != 0);
}

// dst[i] = blend_pixel_static(base[i], top[i], mode, opacity) — element-wise form (spot checks)
template <bool F>
__global__ __launch_bounds__(256) void blend_arrays_kernel(const uint32_t* __restrict__ base, const uint32_t* __restrict__ top,
                                                           uint32_t* __restrict__ dst, size_t n, uint32_t mode, float opacity)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t b = base[i];
        float acc[4][4];
        acc[0][0] = ubyte0(b); acc[0][1] = ubyte1(b); acc[0][2] = ubyte2(b); acc[0][3] = ubyte3(b);
#pragma unroll
        for (int p = 1; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f;
        const uint32_t t[4] = {top[i], 0u, 0u, 0u};
        blend4_dispatch<F>(mode, acc, t, opacity, rs_clamp(opacity, 0.0f, 1.0f));
        dst[i] = pack_rgba(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
    }
}

// Stroke commit (ref: src/ui/panels/tools/behavior/raster/bezier_commit.rs:103-225):
//   brush : layer = blend_pixel_static(layer, preview, mode, 1.0) where preview.a > 0 and selection != 0
//   eraser: layer.a = ((a/255) * (1 - m/255)).max(0) * 255 as u8 where m = preview.a > 0
__global__ __launch_bounds__(256) void brush_commit_kernel(uint32_t* __restrict__ layer, const uint32_t* __restrict__ preview,
                                                           const uint8_t* __restrict__ selection, size_t n, uint32_t mode,
                                                           int is_eraser)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (selection && selection[i] == 0) continue;
        const uint32_t pp = preview[i];
        if ((pp >> 24) == 0u) continue;
        const uint32_t lp = layer[i];
        if (is_eraser) {
            const float mask_strength = div255(ubyte3(pp));
            const float current_a = div255(ubyte3(lp));
            const float new_a = __builtin_fmaxf(current_a * (1.0f - mask_strength), 0.0f);
            layer[i] = (lp & 0x00ffffffu) | ((uint32_t)trunc_u8f(new_a * 255.0f) << 24);
        } else {
            float acc[4][4];
            acc[0][0] = ubyte0(lp); acc[0][1] = ubyte1(lp); acc[0][2] = ubyte2(lp); acc[0][3] = ubyte3(lp);
#pragma unroll
            for (int p = 1; p < 4; ++p) acc[p][0] = acc[p][1] = acc[p][2] = acc[p][3] = 0.0f;
            const uint32_t t[4] = {pp, 0u, 0u, 0u};
            blend4_dispatch<true>(mode, acc, t, 1.0f, 1.0f);
            layer[i] = pack_rgba(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
        }
    }
}

// device-side check of rdiv against the compiler's IEEE divide: out[0] += number of mismatching pairs
__global__ __launch_bounds__(256) void rdiv_check_kernel(uint64_t seed, uint32_t iters, unsigned long long* out)
{
    uint64_t s = seed + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const uint32_t a = (uint32_t)(s >> 33), b = (uint32_t)(s >> 7);
        // numerator in [0, 2), denominator in [2^-48, 1]: random mantissas, exponents in the kernel's operand range
        const float n = ((a & 0xC0u) == 0u) ? 0.0f : __builtin_bit_cast(float, ((a >> 8) & 0x007fffffu) | ((127u - (a & 31u)) << 23));
        const float d = __builtin_bit_cast(float, (b & 0x007fffffu) | ((127u - ((b >> 23) % 49u)) << 23));
        const float q_ref = n / d;
        const float q_fast = rdiv_apply(rdiv_prepare(d), n);
        bad += (__builtin_bit_cast(uint32_t, q_ref) != __builtin_bit_cast(uint32_t, q_fast));
    }
    if (bad) atomicAdd(out, bad);
}

// device-side check of pack_round_rgba (k_common.h) against the step-by-step `v.round().clamp(0, 255) as u8` for every f32 bit pattern:
// out[0] += mismatches among the patterns arithmetic can produce, out[1] += mismatches among signalling NaNs (it cannot)
__global__ __launch_bounds__(256) void round_pack_check_kernel(unsigned long long* out)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, bad_snan = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += nth) {
        const uint32_t bits = (uint32_t)b;
        const float v = __builtin_bit_cast(float, bits);
        const uint32_t fast = pack_round_rgba(v, 0.0f, 0.0f, 0.0f) & 0xffu;
        const uint32_t ref = (v != v) ? 0u : (uint32_t)round_u8f(v); // `NaN as u8` == 0
        if (fast != ref) {
            const bool snan = (bits & 0x7f800000u) == 0x7f800000u && (bits & 0x007fffffu) != 0u && !(bits & 0x00400000u);
            if (snan) ++bad_snan; else ++bad;
        }
    }
    if (bad) atomicAdd(out, bad);
    if (bad_snan) atomicAdd(out + 1, bad_snan);
}

} // namespace

extern "C" void pfxk_flatten_set_variant(int v) { g_flatten_variant = v; }
extern "C" hipError_t pfxk_chunk_alpha_flags(hipStream_t s, const uint8_t* d_px, uint32_t w, uint32_t h, uint32_t cx0, uint32_t cy0, uint32_t ncx, uint32_t ncy,
                                             uint8_t* d_flags)
{
    if (ncx == 0 || ncy == 0) return hipSuccess;
    chunk_alpha_flags_kernel<<<ncx * ncy, 256, 0, s>>>(d_px, w, h, cx0, cy0, ncx, d_flags);
    return hipGetLastError();
}
extern "C" hipError_t pfxk_chunk_start(hipStream_t s, const uint8_t* const* d_flag_ptrs, const uint8_t* d_want, uint32_t n_layers, uint32_t n_chunks,
                                       uint8_t* d_start, uint32_t* useful_pinned, uint32_t tag)
{
    if (n_chunks == 0) return hipSuccess;
    chunk_start_kernel<<<(n_chunks + 255) / 256, 256, 0, s>>>(d_flag_ptrs, d_want, n_layers, n_chunks, d_start, useful_pinned, tag);
    return hipGetLastError();
}
extern "C" hipError_t pfxk_flatten_dle_stats(unsigned long long* out16, int reset)
{
    hipError_t e = hipSuccess;
    if (out16) e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dle_stats), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        const unsigned long long z[16] = {};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_dle_stats), z, sizeof z);
    }
    return e;
}
extern "C" void pfxk_flatten_set_dle(int units_per_wave, int ring_log2)
{
    if (units_per_wave >= 0) g_dle_units = units_per_wave;
    (void)ring_log2;
}
extern "C" void pfxk_flatten_set_dle_dev(int stats_on, int cfg)
{
    if (stats_on >= 0) g_dle_stats_on = stats_on;
    if (cfg >= 0) g_dle_cfg = cfg;
}
extern "C" void pfxk_flatten_set_dle_sched(int sched, int fracA, int fracB)
{
    if (sched >= 0) g_dle_sched = sched;
    if (fracA >= 0 && fracA <= 100) g_dle_fracA = fracA;
    if (fracB >= 0 && fracB <= 100 - g_dle_fracA) g_dle_fracB = fracB;
}
extern "C" void pfxk_flatten_set_dle_plan(int kernel, int s1, int s2)
{
    if (kernel >= 0) g_dle_kernel = kernel;
    if (s1 >= -1) g_dle_s1 = s1;
    if (s2 >= -1) g_dle_s2 = s2;
}
extern "C" hipError_t pfxk_dle_probe(hipStream_t s, const pfxk_layer_desc* d_layers, uint32_t cand_layer, uint32_t cand_kind, uint32_t n_px, uint32_t* verdict_pinned,
                                     uint32_t tag)
{
    dle_probe_kernel<<<1, 1024, 0, s>>>(d_layers, cand_layer, cand_kind, n_px, verdict_pinned, tag);
    return hipGetLastError();
}
extern "C" hipError_t pfxk_unorm_store_check(hipStream_t s, uint8_t* d_scratch1k, unsigned long long* d_out)
{
    unorm_store_check_kernel<<<1, 256, 0, s>>>(d_scratch1k, d_out);
    return hipGetLastError();
}
extern "C" hipError_t pfxk_round_pack_check(hipStream_t s, unsigned long long* d_out)
{
    round_pack_check_kernel<<<4096, 256, 0, s>>>(d_out);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_rdiv_check(hipStream_t s, uint64_t seed, uint32_t blocks, uint32_t iters, unsigned long long* d_out)
{
    rdiv_check_kernel<<<blocks, 256, 0, s>>>(seed, iters, d_out);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_blend_arrays(hipStream_t s, const uint8_t* d_base, const uint8_t* d_top, uint8_t* d_dst,
                                        size_t n_px, uint32_t mode, float opacity, int fast_div)
{
    if (n_px == 0) return hipSuccess;
    size_t blocks = (n_px + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (fast_div)
        blend_arrays_kernel<true><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_base, (const uint32_t*)d_top,
                                                                   (uint32_t*)d_dst, n_px, mode, opacity);
    else
        blend_arrays_kernel<false><<<(uint32_t)blocks, 256, 0, s>>>((const uint32_t*)d_base, (const uint32_t*)d_top,
                                                                    (uint32_t*)d_dst, n_px, mode, opacity);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_brush_commit(hipStream_t s, uint8_t* d_layer, const uint8_t* d_preview,
                                        const uint8_t* d_selection, uint32_t w, uint32_t h, uint32_t mode, int is_eraser)
{
    const size_t n = (size_t)w * h;
    if (n == 0) return hipSuccess;
    size_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    brush_commit_kernel<<<(uint32_t)blocks, 256, 0, s>>>((uint32_t*)d_layer, (const uint32_t*)d_preview, d_selection, n, mode,
                                                         is_eraser);
    return hipGetLastError();
}

extern "C" hipError_t pfxk_flatten(hipStream_t stream, const pfxk_layer_desc* d_layers, uint32_t n_layers,
                                   const float* d_adj_table, int general, int fast_div, uint8_t* d_chunk_active,
                                   int chunk_active_ready, uint32_t w, uint32_t h, uint8_t* d_dst, const pfxk_preview* preview, const pfxk_region* region,
                                   const pfxk_dle_cands* cands, const uint8_t* d_chunk_start, int typed_store_ok, int mode_class)
{
    size_t n_quads = ((size_t)w * h + 3) / 4;
    if (n_quads == 0) return hipSuccess;
    pfxk_preview PV{};
    if (preview && preview->pixels) { PV = *preview; general = 1; }
    pfxk_region RG{};
    if (region && region->rw && region->rh) { RG = *region; general = 1; n_quads = (size_t)((RG.rw + 3u) / 4u) * RG.rh; }
    if (general && d_chunk_active && !chunk_active_ready) {
        const uint32_t nchunks = ((w + 63u) / 64u) * ((h + 63u) / 64u);
        chunk_active_kernel<<<nchunks, 256, 0, stream>>>(d_layers, n_layers, w, h, d_chunk_active, PV.pixels ? PV.chunk_present : nullptr);
    }
    const uint32_t block = 256;
    const size_t cap = 256u * 8u * 4u; // 256 CUs x 8 blocks, x4 waves of grid-stride work granularity
    const size_t n_px = (size_t)w * h;
    const int flatten_variant = g_flatten_variant, dle_cfg = g_dle_cfg, dle_units = g_dle_units, dle_sched_mode = g_dle_sched, fracA = g_dle_fracA, fracB = g_dle_fracB;
    if (!general && fast_div && n_layers > 0 && n_px < (1u << 30) && flatten_variant != 9) {
        // one 64*PX-pixel tile per wave while that stays below the cap (the dispatcher balances the tail), grid-stride beyond
        // Shipped shape (variant 0), from tools/ab_shallow.py at 8K on stacks of 2 .. 32 layers, with and without an opaque background: 2 pixels
        // per lane and 2 register sets everywhere (against 3 x 3: -2 % at 32 layers, -8 % at 9, -20 % at 4); up to 16 layers the waves also
        // walk the image with a grid stride instead of taking one tile each (a 4-layer tile is over before its launch cost is: -5 .. -10 % more)
        int variant = flatten_variant % 10;
        bool stride = flatten_variant >= 10;
        if (flatten_variant == 0 || flatten_variant == 8) {
            variant = 1; stride = n_layers <= 16;
            // Round 4, tools/lab/shallow_modes_ab.py (8K, 9 and 12 layers, every mode, S2 data; profiles/r04_shallow_modes_ab.txt) and tools/ab_shallow.py (2 .. 9 layers):
            // 4 pixels per lane pay where the blend arithmetic is light — Normal, Multiply, Additive, Difference, Lighten / Darken, Overwrite, Subtract, Linear Burn:
            // one tile per wave -5 .. -7 % from 7 layers up (Normal at 9 layers 0.312 -> 0.290 ms), grid-stride -2 .. -3 % at 5 - 6 layers and for the medium modes
            // (Screen, Overlay, Negation, Hard Light, Exclusion, Linear Light, Hard Mix); the heavy modes (Reflect .. Color Dodge, Xor, Soft Light, Divide, Vivid
            // Light, Pin Light) keep 2 pixels per lane (Vivid Light +9 % with 4), and so do stacks of up to 4 layers
            if (n_layers >= 5 && mode_class == 2) { variant = 3; stride = n_layers < 7; }
            else if (n_layers >= 5 && mode_class == 1) { variant = 3; stride = true; }
        }
        auto grid = [&](uint32_t px_per_wave) {
            size_t tiles = (n_px + px_per_wave - 1) / px_per_wave, b = (tiles + 3) / 4;
            const size_t lim = stride ? cap : (size_t)1 << 20;
            return (uint32_t)(b > lim ? lim : b);
        };
        // dead-layer elimination when the stack holds a reset layer above the bottom one (variant 8 switches it off)
        if (cands && cands->n > 0 && flatten_variant != 8) {
            // dle_cfg: 0 = 3 pixels per lane, 2 register sets (6 waves per SIMD; measured best), 1 = 2 pixels per lane, 2 = 3 pixels per lane, 3 sets (old kernel only);
            // dle_sched 0 = equal streams of dle_units
            const bool srt_kernel = g_dle_kernel == 0 && typed_store_ok;
            const uint32_t px = dle_cfg == 1 ? 2u : 3u;
            const uint32_t upx = 64u * px;
            const uint32_t units = (uint32_t)((n_px + upx - 1) / upx);
            const uint32_t umax = 65535u / upx; // queue entries are 16-bit pixel offsets
            dle_sched SC{};
            if (dle_sched_mode == 0) {
                SC.UA = SC.UB = SC.UC = std::min(dle_units > 0 ? (uint32_t)dle_units : 8u, umax);
                SC.wavesA = (units + SC.UA - 1) / SC.UA;
            } else {
                // the long streams fill the chip about once (256 CUs x 24 waves): measured best (profiles/r03_tuning.md).  No floor: a floor of 4 units left small
                // launches (a 64-row dirty rectangle, a thin band of a sharded document) with a quarter of the waves the chip holds — 8K x 64 rows 0.091 -> 0.037 ms,
                // 8K x 256 rows 0.115 -> 0.081 (tools/lab/band_units_sweep.py, profiles/r04_band_units_sweep.txt)
                // No ceiling above 8 either since the class-sorting kernel (no queue across units: a stream's length only sets how the launch drains): at 8K
                // 22 units per wave left the long streams at 0.82 of a chip-full, 8 runs 1.1 % faster (18 — a hair over one chip-full — 3 % slower),
                // 8K x 2176 rows 11 -> 8: -1.4 % (tools/lab/units_ab.py, randomised order, profiles/r04_units_ab.txt)
                // (round 3's queue kernel keeps its rule: its compacted rounds need long streams)
                const uint32_t ua_fill = ((uint32_t)((uint64_t)units * (uint32_t)fracA / 100u) + 6143u) / 6144u;
                const uint32_t ua_auto = srt_kernel ? std::min(std::max(ua_fill, 1u), 8u) : std::max(ua_fill, 4u);
                SC.UA = std::min(dle_units > 0 ? (uint32_t)dle_units : ua_auto, umax);
                SC.UB = std::max(SC.UA / 4u, 1u);
                SC.UC = 1u;
                SC.wavesA = (uint32_t)((uint64_t)units * (uint32_t)fracA / 100u) / SC.UA;
                SC.wavesB = (uint32_t)((uint64_t)units * (uint32_t)fracB / 100u) / SC.UB;
            }
            const uint32_t rest = units - std::min(units, SC.wavesA * SC.UA + SC.wavesB * SC.UB);
            const uint32_t waves = SC.wavesA + SC.wavesB + (rest + SC.UC - 1) / SC.UC;
            pfxk_dle_cands C = *cands;
            C.stats = (uint32_t)g_dle_stats_on;
            if (srt_kernel) {
                // re-deal attempts start right above the topmost candidate (below it the early pixels' rounds and the reset layer itself run) and
                // repeat every `seg` layers
                dle_plan P{};
                const int o1 = g_dle_s1, o2 = g_dle_s2;
                P.s1 = C.layer[C.n - 1u] + (o1 > 0 ? (uint32_t)o1 : 1u);
                P.seg = o2 < 0 ? 3u : (o2 == 0 ? 1000u : (uint32_t)o2);   // 0: a single attempt
                if (o1 == 0) P.seg = 0u;                                   // no attempts at all: round 3's natural pass + parking in the destination
#define PFX_ARGS <<<waves, 64, 0, stream>>>(d_layers, n_layers, (uint32_t)n_px, d_dst, C, SC, P)
                if (C.stats & 4u) flatten_srt_kernel<3, false, true> PFX_ARGS; // diagnostic: per-phase wave clocks (tools/lab/srt_phases.py)
                else if (C.stats & 2u) flatten_srt_kernel<3, true> PFX_ARGS;   // diagnostic: the load stream without the arithmetic
                else if (dle_cfg == 1) flatten_srt_kernel<2> PFX_ARGS;
                else flatten_srt_kernel<3> PFX_ARGS;
#undef PFX_ARGS
                return hipGetLastError();
            }
            // one wave per workgroup: a wave's stream is independent of its neighbours' (no barrier, private LDS slice), and a 4-wave
            // workgroup would hold its LDS and wave slots until its slowest stream ends
#define PFX_ARGS <<<waves, 64, 0, stream>>>(d_layers, n_layers, (uint32_t)n_px, d_dst, C, SC)
            if (dle_cfg == 2) flatten_dle_kernel<3, 10, 1, 3> PFX_ARGS;
            else if (dle_cfg == 1) flatten_dle_kernel<2, 9, 1, 3> PFX_ARGS;
            else flatten_dle_kernel<3, 10, 1, 2> PFX_ARGS;
#undef PFX_ARGS
            return hipGetLastError();
        }
        switch (variant) {
#define PFX_STREAM(PX, NB, MINW) flatten_stream_kernel<PX, NB, MINW><<<grid(64 * PX), block, 0, stream>>>(d_layers, n_layers, (uint32_t)n_px, d_dst, d_chunk_start, w)
        case 1: PFX_STREAM(2, 2, 1); break;
        case 2: PFX_STREAM(2, 3, 1); break;
        case 3: PFX_STREAM(4, 2, 1); break;
        case 4: PFX_STREAM(4, 3, 1); break;
        case 5: PFX_STREAM(1, 3, 1); break;
        case 6: PFX_STREAM(3, 3, 1); break;
        default: PFX_STREAM(3, 3, 1); break;
#undef PFX_STREAM
        }
        return hipGetLastError();
    }
    size_t blocks = (n_quads + block - 1) / block;
    if (blocks > cap) blocks = cap;
    const uint32_t g = (uint32_t)blocks;
#define PFX_LAUNCH(G, F) flatten_kernel<G, F><<<g, block, 0, stream>>>(d_layers, n_layers, d_adj_table, (G) ? d_chunk_active : nullptr, w, h, d_dst, PV, RG)
    if (general) { if (fast_div) PFX_LAUNCH(true, true); else PFX_LAUNCH(true, false); }
    else         { if (fast_div) PFX_LAUNCH(false, true); else PFX_LAUNCH(false, false); }
#undef PFX_LAUNCH
    return hipGetLastError();
}
