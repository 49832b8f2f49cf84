// pfx_kernels.h — launch entry points of the gfx950 kernels (internal; the public ABI is include/pfx.h).
// Every launcher enqueues on `stream`, never synchronises, and returns hipGetLastError().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PFXK_LAYER_RASTER = 0, PFXK_ADJ_EXPOSURE = 1, PFXK_ADJ_BRIGHTNESS_CONTRAST = 2, PFXK_ADJ_INVERT = 3,
       PFXK_ADJ_CHANNEL_MIXER = 4 };

// one entry of the layer stack, bottom -> top (32 bytes, read with scalar loads: uniform per wave)
typedef struct pfxk_layer_desc {
    const uint8_t* pixels; // device RGBA8, tight w*h*4 (NULL for adjustment layers)
    const uint8_t* mask;   // device w*h "conceal" bytes or NULL
    float    opacity;      // as stored (unclamped: the >= 1.0 fast path looks at the raw value)
    uint32_t mode;         // BlendMode::to_u8
    uint32_t kind;         // PFXK_LAYER_RASTER or PFXK_ADJ_*
    uint32_t adj_off;      // adjustment layers: offset (floats) of the 16 parameters in the adj table; raster layers: bit pattern of
                           // opacity.clamp(0, 1) (the streaming compositor kernels read this instead of clamping per wave and layer)
} pfxk_layer_desc;

// parameter block of the pointwise kernels (passed by value: lands in SGPRs)
typedef struct pfxk_params { float p[12]; } pfxk_params;

// ids mirror include/pfx.h (pfx_adjust_op / pfx_rhai_op); static_asserts in pfx_api.cpp keep them in sync
enum { PFXK_OP_INVERT = 0, PFXK_OP_INVERT_ALPHA, PFXK_OP_SEPIA, PFXK_OP_BRIGHTNESS_CONTRAST, PFXK_OP_HSL,
       PFXK_OP_EXPOSURE, PFXK_OP_HIGHLIGHTS_SHADOWS, PFXK_OP_TEMPERATURE_TINT, PFXK_OP_THRESHOLD, PFXK_OP_POSTERIZE,
       PFXK_OP_COLOR_BALANCE, PFXK_OP_GRADIENT_MAP, PFXK_OP_BLACK_AND_WHITE, PFXK_OP_VIBRANCE, PFXK_OP_LUT_RGBA,
       PFXK_OP_DESATURATE, PFXK_OP_COUNT };
enum { PFXK_RHAI_INVERT = 0, PFXK_RHAI_DESATURATE, PFXK_RHAI_SEPIA, PFXK_RHAI_SEPIA_STRENGTH,
       PFXK_RHAI_BRIGHTNESS_CONTRAST, PFXK_RHAI_HSL, PFXK_RHAI_EXPOSURE, PFXK_RHAI_LEVELS, PFXK_RHAI_COUNT };

// ---- k_flatten.hip ----
// fast_div: 1 = shared-reciprocal division (bit-identical to '/' for opacities that are 0 or >= 2^-40), 0 = plain '/'
// tool preview layer folded into the active layer (canvas_state.rs:541-548,593-658); all pointers are device memory
typedef struct pfxk_preview {
    const uint8_t* pixels;              // w*h*4, NULL = no preview
    const uint8_t* chunk_present;       // per 64x64 chunk: the preview TiledImage has this chunk
    const uint8_t* layer_chunk_present; // per chunk: the ACTIVE LAYER has this chunk (a missing chunk reads as (0,0,0,0))
    uint32_t active_pos;                // position of the active layer in d_layers, 0xFFFFFFFF = not in the stack (hidden)
    uint32_t mode, is_eraser, replaces; // preview_blend_mode (normalised 0..24), preview_is_eraser, preview_replaces_layer
} pfxk_preview;
// dirty rectangle: composite only [x0, x0+rw) x [y0, y0+rh) into a compact rw x rh destination (rw == 0: whole canvas)
typedef struct pfxk_region { uint32_t x0, y0, rw, rh; } pfxk_region;
// Candidate "reset" layers of a stack (positions in the descriptor array, ascending): layers whose result does not depend on what is
// below them wherever their alpha qualifies — kind 0: Overwrite, alpha != 0 (canvas_state.rs:1275-1281); kind 1: Normal at
// opacity >= 1, alpha == 255 (:1258).  The compositor skips the layers below a pixel's topmost reset (k_flatten.hip, flatten_dle_kernel).
#define PFXK_DLE_MAX 4
#define PFXK_DESC_PAD 16 /* copies of the last layer descriptor the host appends: srt_layers (k_flatten.hip) reads up to descriptor n + 2, srt_early<NB> up to n + 2 NB - 2 (valid indices end at n + PFXK_DESC_PAD - 1) */
typedef struct pfxk_dle_cands { uint32_t n; uint32_t layer[PFXK_DLE_MAX]; uint32_t kind[PFXK_DLE_MAX]; uint32_t stats; /* set by the launcher */ } pfxk_dle_cands;
hipError_t pfxk_flatten(hipStream_t stream, const pfxk_layer_desc* d_layers, uint32_t n_layers,
                        const float* d_adj_table, int general, int fast_div, uint8_t* d_chunk_active, int chunk_active_ready, uint32_t w,
                        uint32_t h, uint8_t* d_dst, const pfxk_preview* preview /* may be NULL */,
                        const pfxk_region* region /* may be NULL */, const pfxk_dle_cands* cands /* may be NULL: no elimination */,
                        const uint8_t* d_chunk_start /* may be NULL: per-chunk first layer that can show (pfxk_chunk_start) */,
                        int typed_store_ok /* the device's float -> UNORM8 typed-store conversion is verified (pfxk_unorm_store_check): the class-sorting
                                              kernel may write its result that way (k_flatten.hip: flatten_srt_kernel) */,
                        int mode_class /* arithmetic weight of the stack's blend modes: 0 heavy (or unknown), 1 medium, 2 light: picks the streaming kernel's shape */);
// per-chunk alpha summary of a stored layer (bit 0: all 255, bit 1: none 0) over the chunk rectangle [cx0, cx0+ncx) x [cy0, cy0+ncy), and the
// per-chunk start table of a stack (want[k]: 1 = Normal at opacity >= 1 needs bit 0, 2 = Overwrite needs bit 1, 0 = layer k never resets)
hipError_t pfxk_chunk_alpha_flags(hipStream_t s, const uint8_t* d_px, uint32_t w, uint32_t h, uint32_t cx0, uint32_t cy0, uint32_t ncx, uint32_t ncy,
                                  uint8_t* d_flags);
hipError_t pfxk_chunk_start(hipStream_t s, const uint8_t* const* d_flag_ptrs, const uint8_t* d_want, uint32_t n_layers, uint32_t n_chunks,
                            uint8_t* d_start, uint32_t* useful_pinned /* may be NULL: receives `tag` if any chunk starts above layer 0 */, uint32_t tag);
void       pfxk_flatten_set_dle(int units_per_wave /* 0 = default, < 0 = keep */, int ring_log2 /* 10 | 11, else keep */);
/* samples 256 units against the topmost reset candidate; *verdict_pinned = tag | 0x80000000 when at least half start at it outright (k_flatten.hip: dle_probe_kernel) */
hipError_t pfxk_dle_probe(hipStream_t s, const pfxk_layer_desc* d_layers, uint32_t cand_layer, uint32_t cand_kind, uint32_t n_px, uint32_t* verdict_pinned, uint32_t tag);
hipError_t pfxk_flatten_dle_stats(unsigned long long* out16 /* may be NULL */, int reset); // synchronises the device
void       pfxk_flatten_set_dle_dev(int stats_on /* < 0 keep */, int cfg /* < 0 keep */);
void       pfxk_flatten_set_dle_sched(int sched /* 0 equal streams, 1 shrinking */, int fracA, int fracB); // < 0 keeps
void       pfxk_flatten_set_dle_plan(int kernel /* 0 class sorting, 1 round-3 kernel; < 0 keeps */, int s1 /* first re-deal attempt, layers above the topmost candidate: -1 = 1, 0 = never; < -1 keeps */,
                                     int s2 /* layers between attempts: -1 = 3, 0 = one attempt only; < -1 keeps */);
// out[0] += byte values for which a typed UNORM8 store of RN(k / 255) does not write k or the typed load does not return RN(k / 255)
hipError_t pfxk_unorm_store_check(hipStream_t s, uint8_t* d_scratch1k, unsigned long long* d_out);
void       pfxk_flatten_set_variant(int v); // tuning knob: 0 = shipped kernel, 1.. = experimental pixels-per-lane / occupancy variants
// counts (into *d_out) operand pairs for which the shared-reciprocal division differs from the IEEE divide
hipError_t pfxk_rdiv_check(hipStream_t s, uint64_t seed, uint32_t blocks, uint32_t iters, unsigned long long* d_out);
hipError_t pfxk_round_pack_check(hipStream_t s, unsigned long long* d_out /* [2]: mismatches, signalling-NaN mismatches */);

// ---- k_gauss.hip ---- (d_wts_tap0 points at tap 0 of a device array with pfxk_gauss_weight_pad() zeros on both sides)
int        pfxk_gauss_max_radius(void);
int        pfxk_gauss_weight_pad(void);
void       pfxk_gauss_set_v_config(int cfg); // tuning knob, 0 = shipped
void       pfxk_gauss_set_mfma_segments(int n); // tuning knob of the matrix-core kernel, 0 = automatic
/* bit-exact mode, radii 1 .. pfxk_gauss_fused_exact_max_radius(): both passes in one kernel, the f32 intermediate in an LDS ring (k_gauss.hip) */
void pfxk_gauss_set_fused_exact(int on);
int pfxk_gauss_fused_exact_max_radius(void);
/* one-channel form (w % 4 == 0, tight rows, radii 1 .. pfxk_gauss_fused_exact_max_radius()): element by element what one channel of pfxk_gauss_fused_exact gives on (a, a, a, a) */
hipError_t pfxk_gauss_plane_exact(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h);
hipError_t pfxk_gauss_fused_exact(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h,
                                  int epilogue /* 0 the blur, 1 sharpen, 2 glow: combine with the source in the store (k_effects.hip: combine_kernel) */, float p0, const uint8_t* d_mask);
hipError_t pfxk_gauss_h(hipStream_t stream, const uint8_t* d_src, float* d_tmp, const float* d_wts_tap0, int radius,
                        uint32_t w, uint32_t h, int exact);
int        pfxk_gauss_mfma_max_radius(void);
int        pfxk_gauss_mfma_wlen(void);
int        pfxk_gauss_mfma_woff(void);
hipError_t pfxk_gauss_mfma(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const uint16_t* d_wsplit,
                           int radius, float inv_scale2, float bias_c, float bias_single, uint32_t w, uint32_t h, uint32_t first_row, int n_cus);
hipError_t pfxk_gauss_mfma_chain(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const uint16_t* d_wsplit, int radius, float inv_scale2, float bias_c,
                                 float bias_single, uint32_t w, uint32_t h, uint32_t first_row, int n_cus, const struct pfxk_chain* chain);
void       pfxk_gauss_set_mfma_cols64(int mask); // bit 0 / 1 / 2: launches with 4 / 6 / 8 K blocks (sigma <= 5.3 / 10.6 / 16) on 64-column strips instead of 32-column ones (bit-identical results)
void       pfxk_gauss_set_mfma_parts(int weight_parts, int h_parts); // f16 pieces per weight (2 or 1) / per horizontal result (2, or 1 with single weights)
hipError_t pfxk_gauss_v(hipStream_t stream, const float* d_tmp, uint8_t* d_dst, const float* d_wts_tap0, int radius,
                        uint32_t w, uint32_t h, int exact);

// ---- chains of pointwise ops (round 6: pfx_chain_dev) ----
// op[i] = a PFXK_OP_* id (ops::adjustments flavour) or PFXK_CHAIN_RHAI | a PFXK_RHAI_* id (Rhai-inline flavour); P[i] = its prepared parameter block;
// lut_slot[i] = which 1024-byte table of the chain's LUT buffer the op reads (ops without a table: 0).
#define PFXK_CHAIN_MAX 8
#define PFXK_CHAIN_LUTS 4
#define PFXK_CHAIN_RHAI 0x100u
typedef struct pfxk_chain {
    uint32_t n, n_luts;
    uint32_t op[PFXK_CHAIN_MAX];
    uint32_t lut_slot[PFXK_CHAIN_MAX];
    pfxk_params P[PFXK_CHAIN_MAX];
} pfxk_chain;
// every op of the chain on every pixel, one pass over memory (d_src == d_dst allowed); d_luts: n_luts x 1024 bytes
hipError_t pfxk_pointwise_chain(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_luts, const pfxk_chain* C, uint32_t w, uint32_t h);
// the bit-exact fused Gaussian with the chain applied to every blurred pixel in its store (epilogue 3 of pfxk_gauss_fused_exact)
hipError_t pfxk_gauss_fused_exact_chain(hipStream_t stream, const uint8_t* d_src, uint8_t* d_dst, const float* d_wts_tap0, int radius, uint32_t w, uint32_t h,
                                        const pfxk_chain* C, const uint8_t* d_luts);

// ---- k_pointwise.hip ---- (d_lut: 1024 readable bytes when the op uses a LUT)
hipError_t pfxk_adjust(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const uint8_t* d_lut,
                       int op, const pfxk_params* P, int sparse_mode, uint32_t w, uint32_t h);
hipError_t pfxk_rhai_adjust(hipStream_t s, uint8_t* d_px, const uint8_t* d_lut, int op, const pfxk_params* P,
                            uint32_t w, uint32_t h);

// ---- k_tiled.hip ----
hipError_t pfxk_chunk_populated(hipStream_t s, const uint8_t* d_src, uint32_t w, uint32_t h, uint8_t* d_populated);
hipError_t pfxk_tiled_roundtrip(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, uint32_t w, uint32_t h);
// TiledImage <-> flat: packed 64x64 chunks + a slot table per canvas chunk (0xffffffff = no chunk); ref: tiled_image.rs:50-104,271-293
hipError_t pfxk_chunks_import(hipStream_t s, const uint8_t* d_packed, const uint32_t* d_slot, uint32_t w, uint32_t h, uint8_t* d_flat);
hipError_t pfxk_chunks_export(hipStream_t s, const uint8_t* d_flat, const uint32_t* d_slot, uint32_t w, uint32_t h, uint8_t* d_packed);
// dst = mask ? blurred : src   (blur_with_selection's copy-back, ref: src/ops/filters.rs:186-200)
hipError_t pfxk_select_by_mask(hipStream_t s, const uint8_t* d_src, const uint8_t* d_fx, const uint8_t* d_mask,
                               uint8_t* d_dst, uint32_t w, uint32_t h);
// per-channel min/max over selected pixels with alpha != 0 -> out[6] = {minR,maxR,minG,maxG,minB,maxB} (u32)
hipError_t pfxk_minmax_rgb(hipStream_t s, const uint8_t* d_src, const uint8_t* d_mask, uint32_t w, uint32_t h,
                           uint32_t* d_out6);

// ---- k_stencil.hip ---- box blur / median / pixelate
void pfxk_box_set_two_pass(int on);
void pfxk_box_set_prefix_from(int radius); // two-pass box blur: radii from which the horizontal pass uses prefix sums (0 = never)
void pfxk_box_set_force(int px, int py); // development sweep: outputs per lane of the two passes (0 = by radius, -1: keep)
void pfxk_box_set_switch(int px_radius, int py_radius); // two-pass box blur: radii from which a lane takes 32 columns / 128 rows (-1: keep)
void pfxk_median_set_search1(int on); // value search (radii 5..24, and 4 with median_single) with one pixel per lane instead of four
void pfxk_median_set_single(int on); // radii 2, 3: one window per lane (the pre-sharing selection networks)
int pfxk_median_get_xlane(void);
void pfxk_median_set_xlane(int on); // 1 (default): radius 2 on the network whose sorted columns are shared across lanes (median_xlane2_kernel)
void pfxk_box_set_strip(int on /* -1 keep */, int fill_percent /* <= 0 keep */, int nseg /* -1 keep, 0 auto */);
hipError_t pfxk_box_blur(hipStream_t s, const uint8_t* d_src, uint8_t* d_tmp, uint8_t* d_dst, const uint8_t* d_mask,
                         int radius, uint32_t w, uint32_t h, int force_two_pass /* in-place calls: the fused kernel would race */);
#define PFXK_MEDIAN_TILE_MAX_RADIUS 24 /* beyond: sliding histograms */
hipError_t pfxk_median(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int radius,
                       uint32_t w, uint32_t h);
// ---- k_median_bits.hip ---- radii 2..7 as a bit-sliced radix select over bit planes of the image (scratch: pfxk_median_bits_scratch bytes)
size_t pfxk_median_bits_scratch(int radius, uint32_t w, uint32_t h);
void pfxk_median_bits_set_pair(int on); // 1 (default): radii 2..7 on the column-pair kernel (two adjacent columns per lane on shared plane registers)
hipError_t pfxk_median_bits(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, uint32_t* d_planes, int radius,
                            uint32_t w, uint32_t h);
hipError_t pfxk_pixelate(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, uint32_t bs,
                         uint32_t w, uint32_t h);

// ---- k_effects.hip ---- sharpen / glow combine pass, bokeh disc blur, motion blur
enum { PFXK_FX_SHARPEN = 0, PFXK_FX_GLOW = 1 };
hipError_t pfxk_combine(hipStream_t s, const uint8_t* d_src, const uint8_t* d_blur, const uint8_t* d_mask, uint8_t* d_dst,
                        uint32_t w, uint32_t h, int op, float p0);
hipError_t pfxk_bokeh(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const int32_t* d_spans_dy_hw,
                      int n_spans, float inv_count, uint32_t w, uint32_t h);
hipError_t pfxk_motion(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int steps, float dx, float dy,
                       float inv_steps, uint32_t w, uint32_t h);

// ---- k_effects2.hip ---- the rest of the effect bank: one template kernel, parameter block by value (layout per effect
// documented next to each fx_pixel branch)
typedef struct pfxk_fx_params { float f[16]; int32_t i[8]; uint32_t u[4]; const void* aux0; } pfxk_fx_params;
enum { PFXK_FX2_ZOOM = 0, PFXK_FX2_DENTS, PFXK_FX2_BULGE, PFXK_FX2_TWIST, PFXK_FX2_NOISE, PFXK_FX2_REDUCE_NOISE, PFXK_FX2_VIGNETTE,
       PFXK_FX2_HALFTONE, PFXK_FX2_GRID, PFXK_FX2_BORDER, PFXK_FX2_SHADOW, PFXK_FX2_OUTLINE, PFXK_FX2_PIXEL_DRAG, PFXK_FX2_RGB_DISPLACE,
       PFXK_FX2_INK, PFXK_FX2_COLOR_FILTER, PFXK_FX2_CONTOURS };
hipError_t pfxk_fx(hipStream_t s, int fx, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const pfxk_fx_params* P,
                   uint32_t w, uint32_t h);
// outline: one bit per pixel (alpha != 0), rows of pfxk_alpha_bits_stride(w) dwords; fx params aux0 = the plane, i[3] = the stride
uint32_t pfxk_alpha_bits_stride(uint32_t w);
hipError_t pfxk_alpha_bits(hipStream_t s, const uint8_t* d_src, uint32_t* d_bits, uint32_t w, uint32_t h);
hipError_t pfxk_crystallize(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, const float* d_seeds_xy,
                            unsigned long long* d_acc /* cells*5 */, uint32_t* d_avg /* cells */, int cells_x, int cells_y, float cs,
                            uint32_t w, uint32_t h);
hipError_t pfxk_oil_painting(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, const uint8_t* d_mask, int radius, int levels,
                             uint32_t w, uint32_t h);
// offset alpha plane -> optional separable max of radius `spread` -> aaaa RGBA image (input of the shadow's Gaussian)
hipError_t pfxk_shadow_alpha(hipStream_t s, const uint8_t* d_src, uint8_t* d_plane_a, uint8_t* d_plane_b, uint8_t* d_rgba, int ox, int oy,
                             int spread, uint32_t w, uint32_t h);

// ---- k_script.hip ---- per-pixel closure VM + the pixel-moving functions of the script transform / selection API
typedef struct pfxk_vm_args {
    const uint32_t* src;      // pre-call image (what the closure's parameters, get_pixel and get_r..a see)
    uint32_t* dst;            // result image (must not alias src)
    const uint8_t* mask;      // selection for is_selected(), or NULL
    const void* code;         // rhai::BcIns[n_code]
    const uint64_t* consts;
    unsigned long long* err;  // initialised to ~0: min over failing pixels of (row-major index << 24 | code << 16 | line)
    int n_code, n_regs, n_params;
    int n_pre;                // code[0 .. n_pre): constant loads a lane executes once, in front of its first pixel (pfx_rhai.cpp: hoist_constants); a pixel starts at n_pre
    int heavy;                // the program uses fmod / pow / sin / cos / tan / atan2 / exp / ln (selects the kernel that carries them)
    int w, h;
    int x0, y0, x1, y1;       // region processed (for_region); the rest of dst must already equal src
    uint32_t step_budget;     // bytecode steps one pixel may execute before the launch reports 'Too many operations'
} pfxk_vm_args;
hipError_t pfxk_vm_run(hipStream_t s, const pfxk_vm_args* A);
// mode: 0 flip_horizontal, 1 flip_vertical, 2 rotate180, 3 rotate90 (cw), 4 rotate270 (ccw); (w, h) = source size
hipError_t pfxk_permute(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, int mode, uint32_t w, uint32_t h);
hipError_t pfxk_recanvas(hipStream_t s, const uint8_t* d_src, uint8_t* d_dst, uint32_t ow, uint32_t oh, uint32_t nw, uint32_t nh, int off_x, int off_y,
                         uint32_t fill_rgba);
// op: 0 rect [x0,x1)x[y0,y1), 1 ellipse (centre cx,cy; squared radii), 2 invert, 3 clear
hipError_t pfxk_mask_op(hipStream_t s, uint8_t* d_mask, int op, int x0, int y0, int x1, int y1, double cx, double cy, double rx2, double ry2,
                        uint32_t w, uint32_t h);
hipError_t pfxk_fill_masked(hipStream_t s, uint8_t* d_img, const uint8_t* d_mask, uint32_t rgba, uint32_t w, uint32_t h);

// ---- k_resize.hip ---- separable resampling with per-axis window / weight tables (v_*: per output row, h_*: per output column)
hipError_t pfxk_resize(hipStream_t s, const uint8_t* d_src, float* d_tmp /* w*nh*4 f32 */, uint8_t* d_dst, const uint32_t* v_left, const uint32_t* v_count,
                       const uint32_t* v_off, const float* v_wts, const uint32_t* h_left, const uint32_t* h_count, const uint32_t* h_off, const float* h_wts,
                       uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t span_max /* 0: two passes through d_tmp */);
int pfxk_resize_tile_cols(void);

typedef struct pfxk_affine_params { float hi[9]; float cx, cy, off_x, off_y, inv_scale; int32_t nearest; } pfxk_affine_params;
hipError_t pfxk_affine(hipStream_t s, const uint8_t* d_src, uint32_t sw, uint32_t sh, uint8_t* d_dst, uint32_t canvas_w, uint32_t canvas_h,
                       const pfxk_affine_params* P);

// ---- k_warp.hip ----
// first_row: index of d_disp's / d_dst's row 0 in the whole output when they are a band of it (0: whole image)
hipError_t pfxk_warp_displacement(hipStream_t s, const uint8_t* d_src, uint32_t sw, uint32_t sh, const float* d_disp,
                                  uint32_t w, uint32_t h, uint8_t* d_dst, uint32_t first_row);
// d_pts: orig points (may be NULL => "fast" identity original) then deformed points, (cols+1)*(rows+1) xy pairs each
hipError_t pfxk_mesh_displacement(hipStream_t s, const float* d_orig, const float* d_def, uint32_t cols, uint32_t rows,
                                  uint32_t w, uint32_t h, float* d_disp);
// d_dst = rows [first_row, first_row + h) of the h_full-row result; d_src = the whole w x h_full source (first_row = 0, h_full = h: whole image)
void pfxk_warp_set_mesh_xcd(int on);
hipError_t pfxk_warp_mesh(hipStream_t s, const uint8_t* d_src, const float* d_orig, const float* d_def, uint32_t cols,
                          uint32_t rows, uint32_t w, uint32_t h, uint8_t* d_dst, uint32_t first_row, uint32_t h_full);

// DisplacementField dabs (k_warp.hip): bounds and Gaussian constants are prepared on the host like the reference's prologue
typedef struct pfxk_disp_dab { int32_t mode, x0, y0, x1, y1; float cx, cy, delta_x, delta_y, r, sigma_sq_2, strength; } pfxk_disp_dab;
hipError_t pfxk_disp_brushes(hipStream_t s, float* d_disp, uint32_t w, uint32_t h, const pfxk_disp_dab* d_dabs, uint32_t n, int bx0, int by0, int bx1,
                             int by1);
/* … over the 64 x 64 chunks a dab's box touches (d_chunks: chunk x | chunk y << 16) instead of the whole bounding box */
hipError_t pfxk_disp_brushes_chunked(hipStream_t s, float* d_disp, uint32_t w, uint32_t h, const pfxk_disp_dab* d_dabs, uint32_t n, int bx0, int by0, int bx1, int by1,
                                     const uint32_t* d_chunks, uint32_t n_chunks);

// ---- k_brush.hip ----
typedef struct pfxk_brush {
    float radius, radius_sq, draw_radius, draw_radius_sq, inv_radius_sq, hardness, flow;
    float src_r, src_g, src_b, src_a;
    uint32_t rgb8;          // (c*255) as u8, packed r | g<<8 | b<<16
    int32_t anti_aliased, use_direct_alpha, is_eraser, mode;
    uint32_t tip_size;      // side of the square image-tip mask, 0 = circle tip
} pfxk_brush;
// one stamp after the host prologue of draw_circle_no_dirty / draw_image_tip_no_dirty: scattered centre, jittered colour bytes,
// inverse-rotation cosine / sine of an image tip
// … and the stamp's pixel box [x0, x1] x [y0, y1] (brush_render.rs:209-215 / :584-587), 16-byte aligned: x0 > x1 = the stamp touches no pixel
typedef struct pfxk_stamp { float cx, cy; uint32_t rgb8; uint32_t rotated; uint32_t x0, x1, y0, y1; float cos_a, sin_a; uint32_t pad[2]; } pfxk_stamp;
hipError_t pfxk_brush_stamps(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B,
                             const pfxk_stamp* d_stamps, uint32_t n_stamps, const uint8_t* d_lut256, const uint8_t* d_tip_mask,
                             const uint8_t* d_selection, int bx0, int by0, int bx1, int by1);
/* the same with the stamps dealt to the 64 x 64 chunks their boxes touch: d_chunks = n_chunks x {chunk x, chunk y, first entry, entries}, d_bins = stamp indices (stroke order per chunk) */
hipError_t pfxk_brush_stamps_binned(hipStream_t s, uint8_t* d_target, uint32_t w, uint32_t h, const pfxk_brush* B, const pfxk_stamp* d_stamps, uint32_t n_points,
                                    const uint8_t* d_lut256, const uint8_t* d_tip_mask, const uint8_t* d_selection, const uint32_t* d_chunks, uint32_t n_chunks,
                                    const uint32_t* d_bins);
hipError_t pfxk_brush_commit(hipStream_t s, uint8_t* d_layer, const uint8_t* d_preview, const uint8_t* d_selection,
                             uint32_t w, uint32_t h, uint32_t mode, int is_eraser);
// element-wise blend_pixel_static over two pixel arrays (spot checks / stroke commit)
hipError_t pfxk_blend_arrays(hipStream_t s, const uint8_t* d_base, const uint8_t* d_top, uint8_t* d_dst, size_t n_px,
                             uint32_t mode, float opacity, int fast_div);

#ifdef __cplusplus
}
#endif
