/*
 * pfx.h — C ABI of libpfx: the MI355X (gfx950) raster pixel pipeline behind PaintFE's operator surface.
 *
 * This is the drop-in boundary.  Every entry point replaces one reference interface (cited as
 * `ref: file:line`, relative to the PaintFE checkout) and keeps its argument meaning and results:
 *   - numerics are those of the reference's **CPU** path (src/ops, src/canvas), not of its WGSL shaders:
 *     bit-exact for the compositor and the integer filters, +-1 LSB worst case (normally 0) for the
 *     float-intermediate filters;
 *   - images are tight row-major straight-alpha RGBA8 (`image::RgbaImage`), `w*h*4` bytes;
 *   - selection masks are `w*h` bytes (`image::GrayImage`), 0 = leave the pixel alone (ref: src/ops/effects.rs:34-41);
 *   - all calls are blocking with respect to the host (like the reference's `device.poll(Wait)` readback,
 *     ref: src/gpu/compute/helpers.rs:147-206) unless the name ends in `_dev`;
 *   - on any error the function returns a negative pfx_status and leaves `dst` untouched, so the caller
 *     can fall back to its CPU path (the reference's own contract, ref: src/ops/adjustments.rs:1045).
 *
 * Two tiers:
 *   host-buffer calls   (`pfx_blur_rgba`, ...)   = the literal `GpuRenderer` / `_core` signatures;
 *   device-resident calls (`..._dev`)            = same kernels on caller-owned device memory and the
 *                                                  context's HIP stream: upload once, chain ops, download once.
 *
 * Device pointers handed to `_dev` calls must be 16-byte aligned (hipMalloc / pfx_dev_alloc results and row-aligned sub-buffers
 * are): the kernels move pixels in 16-byte vectors.  No C++ exception leaves the library: entry points that parse untrusted bytes or run
 * scripts convert allocation failures into PFX_ERR_OOM.
 *
 * Arguments are checked before anything is touched: a NULL context, buffer or description, a zero-sized image, an image of more than 256 000 000 pixels
 * (the reference's document limit, src/canvas/tiled_image.rs:15-26 — the kernels index pixels in 32 bits under it), a rectangle outside its image or a
 * parameter that would set an unbounded per-pixel loop count returns PFX_ERR_INVALID / PFX_ERR_UNSUPPORTED with pfx_last_error() saying which
 * (tests/test_abi_hostile.py sweeps every prototype of this header).
 *
 * One pfx_ctx = one HIP device + one stream.  A context is not thread-safe (neither is the reference's
 * `&mut GpuRenderer`); distinct contexts are independent.  No torch / C++ types cross this boundary.
 */
#ifndef PFX_H
#define PFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFX_ABI_VERSION 1
#define PFX_CHUNK 64   /* ref: src/canvas/defs.rs:7 CHUNK_SIZE */
#define PFX_MAX_LAYERS 256 /* ref: src/io.rs:503 MAX_LAYERS */

typedef enum pfx_status {
    PFX_OK = 0,
    PFX_ERR_INVALID = -1,     /* bad argument (null pointer, zero size, unknown id) */
    PFX_ERR_NO_DEVICE = -2,   /* no usable HIP device: GpuRenderer::try_new -> None (ref: src/gpu/renderer.rs:261) */
    PFX_ERR_HIP = -3,         /* a HIP call failed; pfx_last_error() has the text */
    PFX_ERR_OOM = -4,
    PFX_ERR_UNSUPPORTED = -5, /* e.g. median radius beyond the device path: caller uses its CPU path (ref: renderer.rs:945) */
    PFX_ERR_SCRIPT = -6       /* script front-end error; see pfx_script_result */
} pfx_status;

typedef struct pfx_ctx pfx_ctx;

/* ---- context (ref: GpuRenderer::try_new / new / Drop, src/gpu/renderer.rs:249-300,969) ---- */
int         pfx_ctx_create(int device, pfx_ctx** out);
void        pfx_ctx_destroy(pfx_ctx* ctx);
const char* pfx_last_error(const pfx_ctx* ctx);          /* never NULL */
int         pfx_device_count(void);
int         pfx_abi_version(void);
/* Numeric mode for the float-intermediate stencil (Gaussian): 0 = fused multiply-add allowed (default,
 * +-1 LSB class), 1 = reference evaluation order without FMA contraction (bit-exact with the CPU path). */
int         pfx_ctx_set_exact(pfx_ctx* ctx, int exact);
void*       pfx_ctx_stream(pfx_ctx* ctx);                 /* hipStream_t of this context */
/* adopt != 0: run on the caller's hipStream_t (NULL = the device's default stream, which is what torch's default
 * stream is); adopt == 0: back to the context's own stream.  Lets `_dev` calls interleave in order with a host
 * framework's kernels and RCCL collectives without host synchronisation. */
int         pfx_ctx_set_stream(pfx_ctx* ctx, void* hip_stream, int adopt);
int         pfx_ctx_synchronize(pfx_ctx* ctx);

/* ---- device memory helpers for the `_dev` tier ---- */
int pfx_dev_alloc(pfx_ctx* ctx, size_t bytes, void** out_dev);
int pfx_dev_free(pfx_ctx* ctx, void* dev);
int pfx_dev_upload(pfx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);   /* async on ctx stream + sync */
int pfx_dev_download(pfx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int pfx_dev_memset(pfx_ctx* ctx, void* dst_dev, int value, size_t bytes);
/* Page-locked host memory for the host-buffer entry points (the reference's `&[u8]` in, `Vec<u8>` out seams, src/gpu/renderer.rs:915-947): a caller that keeps its
 * pixels in such buffers moves them at the link's rate (~55 GB/s per direction) instead of the pageable path's ~20 (8K invert_rgba: 13.9 -> ~6 ms, both directions).
 * Ordinary malloc'ed / Vec memory stays valid everywhere; this is an optimisation the caller may take. */
int pfx_host_alloc(pfx_ctx* ctx, size_t bytes, void** out_host);
int pfx_host_free(pfx_ctx* ctx, void* host);

/* ================= B1: GpuRenderer filter methods (ref: src/gpu/renderer.rs:915-947) ================= */
/* blur_rgba(&[u8], w, h, sigma) -> Vec<u8>; numerics: ops::filters::parallel_gaussian_blur (ref: src/ops/filters.rs:242-316) */
int pfx_blur_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float sigma);
/* brightness_contrast_rgba; numerics: ops::adjustments::brightness_contrast (ref: src/ops/adjustments.rs:265-294) */
int pfx_brightness_contrast_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h,
                                 float brightness, float contrast);
/* hsl_rgba(hue deg, sat, light); numerics: hue_saturation_lightness (ref: src/ops/adjustments.rs:300-347) */
int pfx_hsl_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float hue, float sat, float light);
/* invert_rgba; numerics: invert_colors (ref: src/ops/adjustments.rs:115) */
int pfx_invert_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h);
/* median_rgba(radius) -> Option<Vec<u8>>; numerics: median_core (ref: src/ops/effects/noise.rs:357-410).
 * The reference's GPU method returns None for radius > 7 and its CPU median_core has no cap; here radii up to 24 run the tile
 * kernels (selection networks / binary search) and larger ones a sliding-histogram kernel; PFX_ERR_UNSUPPORTED only beyond
 * PFX_MEDIAN_MAX_RADIUS (a 255 x 255 window, the 16-bit histogram counters' limit). */
#define PFX_MEDIAN_MAX_RADIUS 127
int pfx_median_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t radius);

/* ================= B4: pure `_core` functions with optional selection mask ================= */
/* blur_with_selection_pub (ref: src/ops/filters.rs:130-207) */
int pfx_gaussian_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float sigma,
                           const uint8_t* mask);
/* box_blur_core (ref: src/ops/effects/blur.rs:233-318) */
int pfx_box_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius,
                      const uint8_t* mask);
/* median_core (ref: src/ops/effects/noise.rs:357-410) */
int pfx_median_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t radius,
                    const uint8_t* mask);
/* pixelate_core (ref: src/ops/effects/distort.rs:333-373) */
int pfx_pixelate_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t block_size,
                      const uint8_t* mask);
/* effects that reuse the same kernels.  sharpen / glow / drop shadow feed a Gaussian into a gain, so they run the BIT-EXACT Gaussian in every
 * context (the reference's tests hold them at tolerance 0: tests/visual_filters.rs:43-55,154-165); pfx_tune(ctx, "gauss_fast_effects", 1) opts a context into the
 * default-mode Gaussian there too (faster; its +-1 LSB then enters `amount` / `intensity` times). */
/* sharpen_core(flat, amount, radius, mask) (ref: src/ops/effects/stylize.rs:96-143) */
int pfx_sharpen_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float amount, float radius,
                     const uint8_t* mask);
/* glow_core(flat, radius, intensity, mask) (ref: src/ops/effects/stylize.rs:26-70) */
int pfx_glow_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius, float intensity,
                  const uint8_t* mask);
/* bokeh_blur_core(flat, radius, mask) (ref: src/ops/effects/blur.rs:22-115) */
int pfx_bokeh_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius,
                        const uint8_t* mask);
/* motion_blur_core(flat, angle_deg, distance, mask) (ref: src/ops/effects/blur.rs:144-210) */
int pfx_motion_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float angle_deg,
                         float distance, const uint8_t* mask);

/* ---- the rest of the effect bank: every remaining `*_core(flat, params.., mask) -> RgbaImage` of src/ops/effects/ ----
 * Same convention as above (caller-allocated dst, mask = w*h bytes or NULL, 0 leaves the pixel).  Integer / hash based
 * effects are bit-exact with the CPU path; twist, gaussian noise, reduce_noise and vignette evaluate one libm function
 * per pixel and are in the +-1 LSB class.  Enumerations mirror the reference enums in declaration order. */
enum { PFX_NOISE_UNIFORM = 0, PFX_NOISE_GAUSSIAN = 1, PFX_NOISE_PERLIN = 2 };                        /* NoiseType, distort.rs:501-506 */
enum { PFX_HALFTONE_CIRCLE = 0, PFX_HALFTONE_SQUARE = 1, PFX_HALFTONE_DIAMOND = 2, PFX_HALFTONE_LINE = 3 }; /* HalftoneShape, stylize.rs:195-201 */
enum { PFX_GRID_LINES = 0, PFX_GRID_CHECKERBOARD = 1 };                                              /* GridStyle, stylize.rs:285-289 */
enum { PFX_OUTLINE_OUTSIDE = 0, PFX_OUTLINE_INSIDE = 1, PFX_OUTLINE_CENTER = 2 };                    /* OutlineMode, render.rs:353-358 */
enum { PFX_COLOR_FILTER_MULTIPLY = 0, PFX_COLOR_FILTER_SCREEN = 1, PFX_COLOR_FILTER_OVERLAY = 2, PFX_COLOR_FILTER_SOFT_LIGHT = 3 }; /* artistic.rs:219-225 */
/* zoom_blur_core (ref: src/ops/effects/blur.rs:322-427); tint_color = RGBA in 0..1, may be NULL (= no tint) */
int pfx_zoom_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float center_x, float center_y, float strength,
                       uint32_t samples, const float tint_color[4], float tint_strength, const uint8_t* mask);
/* crystallize_core (ref: src/ops/effects/distort.rs:26-169) */
int pfx_crystallize_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float cell_size, uint32_t seed, const uint8_t* mask);
/* dents_core (ref: src/ops/effects/distort.rs:248-310) */
int pfx_dents_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float scale, float amount, uint32_t seed,
                   uint32_t octaves, float roughness, int pinch, int wrap, const uint8_t* mask);
/* bulge_core_at / twist_core_at (ref: src/ops/effects/distort.rs:400-437, 464-493); bulge_core / twist_core = origin (0.5, 0.5) */
int pfx_bulge_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float amount, float origin_x, float origin_y,
                   const uint8_t* mask);
int pfx_twist_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float angle_deg, float origin_x, float origin_y,
                   const uint8_t* mask);
/* add_noise_core (ref: src/ops/effects/noise.rs:73-143) */
int pfx_add_noise_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float amount, int noise_type, int monochrome,
                       uint32_t seed, float scale, uint32_t octaves, const uint8_t* mask);
/* reduce_noise_core (ref: src/ops/effects/noise.rs:172-261) */
int pfx_reduce_noise_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float strength, uint32_t radius,
                          const uint8_t* mask);
/* vignette_core (ref: src/ops/effects/stylize.rs:170-191) */
int pfx_vignette_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float amount, float softness, const uint8_t* mask);
/* halftone_core (ref: src/ops/effects/stylize.rs:242-277) */
int pfx_halftone_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float dot_size, float angle_deg, int shape,
                      const uint8_t* mask);
/* grid_core (ref: src/ops/effects/render.rs:52-92) */
int pfx_grid_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t cell_w, uint32_t cell_h, uint32_t line_width,
                  const uint8_t color[4], int style, float opacity, const uint8_t* mask);
/* canvas_border_core (ref: src/ops/effects/render.rs:114-165) */
int pfx_canvas_border_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4],
                           const uint8_t* mask);
/* shadow_core (ref: src/ops/effects/render.rs:220-349); the Gaussian inside follows pfx_ctx_set_exact */
int pfx_shadow_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, int32_t offset_x, int32_t offset_y, float blur_radius,
                    int widen_radius, const uint8_t color[4], float opacity, const uint8_t* mask);
/* outline_core (ref: src/ops/effects/render.rs:403-572) */
int pfx_outline_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4], int mode,
                     int anti_alias, const uint8_t* mask);
/* pixel_drag_core (ref: src/ops/effects/glitch.rs:44-99) */
int pfx_pixel_drag_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t seed, float amount, uint32_t distance,
                        float direction, const uint8_t* mask);
/* rgb_displace_core(flat, r_off, g_off, b_off, mask) (ref: src/ops/effects/glitch.rs:142-196) */
int pfx_rgb_displace_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, int32_t r_dx, int32_t r_dy, int32_t g_dx,
                          int32_t g_dy, int32_t b_dx, int32_t b_dy, const uint8_t* mask);
/* ink_core (ref: src/ops/effects/artistic.rs:31-99) */
int pfx_ink_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float edge_strength, float threshold, const uint8_t* mask);
/* oil_painting_core (ref: src/ops/effects/artistic.rs:123-215) */
int pfx_oil_painting_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t radius, uint32_t levels,
                          const uint8_t* mask);
/* color_filter_core (ref: src/ops/effects/artistic.rs:266-309) */
int pfx_color_filter_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, const uint8_t filter_color[4], float intensity,
                          int mode, const uint8_t* mask);
/* contours_core (ref: src/ops/effects/contours.rs:56-112) */
int pfx_contours_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float scale, float frequency, float line_width,
                      const uint8_t line_color[4], uint32_t seed, uint32_t octaves, float blend, const uint8_t* mask);

/* ---- the pointwise adjustment bank: one entry point, op id + parameter block ----
 * ops::adjustments flavour (f32, `.round().clamp(0,255) as u8`), ref: src/ops/adjustments.rs:21-108 */
typedef enum pfx_adjust_op {
    PFX_OP_INVERT = 0,           /* :115   params: -                                   */
    PFX_OP_INVERT_ALPHA,         /* :123   params: -                                   */
    PFX_OP_SEPIA,                /* :133   params: -                                   */
    PFX_OP_BRIGHTNESS_CONTRAST,  /* :265   params: brightness, contrast                */
    PFX_OP_HSL,                  /* :300   params: hue_shift(deg), saturation, lightness */
    PFX_OP_EXPOSURE,             /* :352   params: ev  (gain = powf(2, ev) on the host) */
    PFX_OP_HIGHLIGHTS_SHADOWS,   /* :374   params: shadows, highlights                 */
    PFX_OP_TEMPERATURE_TINT,     /* :517   params: temperature, tint                   */
    PFX_OP_THRESHOLD,            /* :1240  params: level                               */
    PFX_OP_POSTERIZE,            /* :1267  params: levels (>= 2)                       */
    PFX_OP_COLOR_BALANCE,        /* :1294  params: shadows[3], midtones[3], highlights[3] */
    PFX_OP_GRADIENT_MAP,         /* :1344  lut: 256 x RGBA                             */
    PFX_OP_BLACK_AND_WHITE,      /* :1373  params: r_weight, g_weight, b_weight        */
    PFX_OP_VIBRANCE,             /* :1408  params: amount                              */
    PFX_OP_LUT_RGBA,             /* levels :465 / curves :584 / auto-levels :144; lut: R[256] G[256] B[256] A[256] */
    PFX_OP_DESATURATE,           /* filters.rs:321 (BT.709, round)                     */
    PFX_OP_COUNT
} pfx_adjust_op;

/* How TiledImage sparsity leaks into the result (ref: src/canvas/tiled_image.rs:50-104,905-932):
 *   PFX_DENSE      arithmetic on every pixel, no chunk effects;
 *   PFX_FROM_FLAT  `*_from_flat` + `TiledImage::from_rgba_image(&out)`: 64x64 chunks of the result whose alpha is
 *                  all zero read back as zeros (ref: src/ops/adjustments.rs:105);
 *   PFX_IN_PLACE   `apply_pixel_transform`: only chunks populated in the input are visited, the others read
 *                  back as zeros (ref: src/ops/adjustments.rs:21-42). */
typedef enum pfx_sparse_mode { PFX_DENSE = 0, PFX_FROM_FLAT = 1, PFX_IN_PLACE = 2 } pfx_sparse_mode;

int pfx_adjust(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, int op,
               const float* params, uint32_t n_params, const uint8_t* lut, const uint8_t* mask, int sparse_mode);

/* Rhai-inline flavour (truncating `as u8`, alpha untouched, selection ignored), ref: src/ops/scripting.rs:869-1075 */
typedef enum pfx_rhai_op {
    PFX_RHAI_INVERT = 0, PFX_RHAI_DESATURATE, PFX_RHAI_SEPIA, PFX_RHAI_SEPIA_STRENGTH,
    PFX_RHAI_BRIGHTNESS_CONTRAST, PFX_RHAI_HSL, PFX_RHAI_EXPOSURE, PFX_RHAI_LEVELS, PFX_RHAI_COUNT
} pfx_rhai_op;
int pfx_rhai_adjust(pfx_ctx* ctx, uint8_t* pixels_inout, uint32_t w, uint32_t h, int op, const float* params,
                    uint32_t n_params);

/* LUT builders (host-side in the reference as well; glibc powf/sqrtf like Rust's f32 methods on Linux) */
void pfx_build_levels_lut(float in_black, float in_white, float gamma, float out_black, float out_white,
                          uint8_t lut[256]);                                   /* ref: adjustments.rs:465-488 */
void pfx_build_curves_lut(const float* points_xy, uint32_t n_points, uint8_t lut[256]); /* ref: adjustments.rs:640-729 */
void pfx_build_stretch_lut(uint8_t min, uint8_t max, uint8_t lut[256]);        /* ref: adjustments.rs:235-256 */
/* auto_levels: device min/max reduction over selected non-transparent pixels, then stretch LUTs (ref: :144-233) */
int  pfx_auto_levels(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, const uint8_t* mask);

/* ================= B2: compositor (ref: src/gpu/renderer.rs:324-586, numerics src/canvas/canvas_state.rs:505-698) ===== */
/* ensure_layer_texture(idx, w, h, data, generation): upload skipped when (generation, w, h) unchanged */
int pfx_layer_upload(pfx_ctx* ctx, uint32_t layer_idx, uint32_t w, uint32_t h, const uint8_t* rgba, uint64_t generation);
/* update_layer_rect(idx, region, data): `data` is the tight region (rw*rh*4) */
int pfx_layer_update_rect(pfx_ctx* ctx, uint32_t layer_idx, uint32_t x, uint32_t y, uint32_t rw, uint32_t rh,
                          const uint8_t* rgba);
/* optional live layer mask ("conceal" alpha, w*h bytes; NULL removes it) (ref: src/canvas/layers.rs:606-620) */
int pfx_layer_set_mask(pfx_ctx* ctx, uint32_t layer_idx, const uint8_t* conceal);
int pfx_layer_remove(pfx_ctx* ctx, uint32_t layer_idx);
int pfx_layer_clear(pfx_ctx* ctx);
uint32_t pfx_layer_count(const pfx_ctx* ctx);
size_t   pfx_layer_memory(const pfx_ctx* ctx);     /* active_texture_memory, ref: renderer.rs:956 */

typedef enum pfx_layer_kind { /* ref: src/canvas/layers.rs:249-262 AdjustmentKind */
    PFX_LAYER_RASTER = 0, PFX_ADJ_EXPOSURE = 1, PFX_ADJ_BRIGHTNESS_CONTRAST = 2, PFX_ADJ_INVERT = 3, PFX_ADJ_CHANNEL_MIXER = 4
} pfx_layer_kind;

typedef struct pfx_layer_info { /* the `(layer_idx, opacity, visible, blend_mode_u8)` tuple, bottom -> top */
    uint32_t layer_idx;
    float    opacity;
    uint8_t  visible;
    uint8_t  blend_mode;   /* BlendMode::to_u8, ref: src/canvas/layers.rs:125-153; unknown ids = Normal */
    uint8_t  kind;         /* pfx_layer_kind; adjustment layers need no uploaded pixels */
    uint8_t  _pad;
    float    adj[16];      /* Exposure [ev]; B/C [brightness, contrast]; ChannelMixer red[4] green[4] blue[4] alpha[4] */
} pfx_layer_info;

/* composite(canvas_w, canvas_h, layer_info) -> Option<Vec<u8>>; `dst` = w*h*4 */
int pfx_composite(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers, uint8_t* dst);
/* composite_dirty_readback: composite, read back only (x, y, rw, rh) into the tight `dst_region` */
int pfx_composite_region(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers,
                         uint32_t x, uint32_t y, uint32_t rw, uint32_t rh, uint8_t* dst_region);
/* The tool preview layer (CanvasState::preview_layer + preview_blend_mode / preview_is_eraser / preview_replaces_layer,
 * ref: src/canvas/canvas_state.rs:541-548,556-560,593-658): while a stroke is in flight its pixels are folded into the ACTIVE
 * layer's pixel before that layer's mask, blend mode and opacity apply — replaced outright, used as an eraser mask, lerped by
 * coverage for Overwrite / Xor, or blended with the tool's mode.  The reference's wgpu compositor does not handle it (only
 * the CPU one does); here it rides on the same kernel.  `preview_pixels` is the preview flattened to w*h RGBA8;
 * `preview_chunk_present` (one byte per 64x64 chunk, row-major) tells which chunks the preview TiledImage holds — NULL means
 * "every chunk with a non-zero alpha", TiledImage::from_rgba_image's rule. */
typedef struct pfx_preview {
    uint32_t active_layer;   /* index into the pfx_layer_info array of the call (CanvasState::active_layer_index) */
    uint8_t  blend_mode;     /* BlendMode::to_u8 */
    uint8_t  is_eraser;
    uint8_t  replaces_layer;
    uint8_t  _pad;
} pfx_preview;
int pfx_composite_preview(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers,
                          const uint8_t* preview_pixels, const uint8_t* preview_chunk_present /* may be NULL */,
                          const pfx_preview* preview, uint8_t* dst);
/* one blend_pixel_static on the host-visible side of the ABI (kept for spot checks; runs a 1-pixel launch) */
int pfx_blend_pixels(pfx_ctx* ctx, const uint8_t* base, const uint8_t* top, uint8_t* dst, size_t n_pixels,
                     uint8_t blend_mode, float opacity);

/* ================= B3: warp pipelines ================= */
/* GpuLiquifyPipeline::warp_into (ref: src/gpu/compute/liquify.rs:176); numerics warp_displacement_full
 * (ref: src/ops/transform.rs:1288-1345).  Field and output are w*h; source is sw*sh. */
int pfx_warp_displacement(pfx_ctx* ctx, const uint8_t* src, uint32_t sw, uint32_t sh, const float* disp_xy,
                          uint32_t w, uint32_t h, uint8_t* dst);
/* The pipeline's source cache (ref: src/gpu/compute/liquify.rs:166-176): warp_into uploads the source texture once and reuses it
 * until invalidate_source; an interactive session then only sends displacement fields.  _cached fails with PFX_ERR_INVALID when
 * no source is set (or it was invalidated) and leaves dst untouched. */
int pfx_warp_set_source(pfx_ctx* ctx, const uint8_t* src, uint32_t sw, uint32_t sh);
int pfx_warp_invalidate_source(pfx_ctx* ctx);
int pfx_warp_displacement_cached(pfx_ctx* ctx, const float* disp_xy, uint32_t w, uint32_t h, uint8_t* dst);
/* generate_displacement_from_mesh (ref: src/ops/transform.rs:1670-1705); orig_pts may be NULL =>
 * generate_displacement_from_mesh_fast / GpuMeshWarpDisplacementPipeline::generate_displacement
 * (ref: src/ops/transform.rs:1712, src/gpu/compute/mesh_warp.rs:131) */
int pfx_mesh_displacement(pfx_ctx* ctx, const float* orig_pts_xy, const float* deformed_pts_xy, uint32_t cols,
                          uint32_t rows, uint32_t w, uint32_t h, float* disp_xy_out);
/* warp_mesh_catmull_rom (ref: src/ops/transform.rs:1743-1761): fused field + gather, no field in memory */
int pfx_warp_mesh_catmull_rom(pfx_ctx* ctx, const uint8_t* src, const float* orig_pts_xy, const float* deformed_pts_xy,
                              uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint8_t* dst);
/* DisplacementField::apply_push/expand/contract/twirl (ref: src/ops/transform.rs:1051-1200): host-side scatter-add
 * exactly as in the reference (serial, glibc expf). mode: 0 push, 1 expand, 2 contract, 3 twirl cw, 4 twirl ccw */
void pfx_displacement_brush(float* disp_xy, uint32_t w, uint32_t h, int mode, float cx, float cy,
                            float delta_x, float delta_y, float radius, float strength);
/* the same brushes on a DEVICE-resident field (the Liquify interactive loop: dabs accumulate into the field that
 * pfx_warp_displacement_dev then samples; no field traffic over PCIe).  A dab list is applied in order.  The Gaussian
 * falloff's exp() is evaluated on the device: weights may differ from the CPU path in the last ulp (+-1 LSB class after
 * the warp). */
typedef struct pfx_disp_dab { int32_t mode; float cx, cy, delta_x, delta_y, radius, strength; } pfx_disp_dab;
int pfx_displacement_brushes_dev(pfx_ctx* ctx, void* disp_dev /* w*h*2 f32 */, uint32_t w, uint32_t h, const pfx_disp_dab* dabs, uint32_t n_dabs);

/* ================= brush stamp loop (ref: src/ui/panels/tools/behavior/raster/brush_render.rs) ================= */
typedef enum pfx_brush_mode { PFX_BRUSH_NORMAL = 0, PFX_BRUSH_DODGE = 1, PFX_BRUSH_BURN = 2, PFX_BRUSH_SPONGE = 3 } pfx_brush_mode;
typedef struct pfx_brush { /* the ToolProperties fields the circle-tip path reads (ref: state.rs:98-157) */
    float size, hardness, flow;
    float color[4];       /* straight RGBA in 0..1 (primary/secondary_color_f32) */
    int32_t anti_aliased;
    int32_t is_eraser;
    int32_t mode;         /* pfx_brush_mode */
} pfx_brush;
/* draw_circle_no_dirty for every point of `points_xy` in order (ref: brush_render.rs:135-400) into `target_inout` */
int pfx_brush_stamps(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush,
                     const float* points_xy, uint32_t n_points, const uint8_t* selection);
/* draw_line_no_dirty (ref: brush_render.rs:762-835): dense 1-px stepping, then the stamp loop */
int pfx_brush_line(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush,
                   float x0, float y0, float x1, float y1, const uint8_t* selection);
/* Brush dynamics — the remaining ToolProperties fields draw_circle_no_dirty reads (ref: state.rs:112-128, brush_render.rs:148-256,
 * 533-760) plus ToolsPanel::stamp_counter: per-stamp scatter, hue / brightness jitter (hashed from the stamp position), and image
 * brush tips with fixed or random rotation.  `tip_mask` is brush_tip_mask: the tip's coverage image already rescaled to the
 * brush size (pfx_brush_tip_rescale does what rebuild_tip_mask does, brush_render.rs:404-528); NULL = the round tip.
 * Image tips stamp with max-alpha or erase only (the reference ignores brush->mode for them). */
typedef struct pfx_brush_dynamics {
    float    scatter, hue_jitter, brightness_jitter;
    uint32_t stamp_counter;
    const uint8_t* tip_mask;      /* tip_mask_size^2 bytes (host memory), or NULL */
    uint32_t tip_mask_size;
    float    tip_rotation;        /* degrees */
    int32_t  tip_random_rotation;
    float    tip_rotation_lo, tip_rotation_hi; /* tip_rotation_range */
} pfx_brush_dynamics;
int pfx_brush_stamps_ex(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn /* may be NULL */,
                        const float* points_xy, uint32_t n_points, const uint8_t* selection);
int pfx_brush_line_ex(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn,
                      float x0, float y0, float x1, float y1, const uint8_t* selection);
/* rebuild_tip_mask (host side, like the reference): bilinear rescale of a square source mask to ceil(brush_size), hardness contrast,
 * anti-alias box passes.  `out` needs ceil(brush_size)^2 bytes; returns that side length (0 = no source). */
uint32_t pfx_brush_tip_rescale(const uint8_t* src_mask, uint32_t src_size, float brush_size, float hardness, uint8_t* out);
/* commit_bezier_to_layer / commit_eraser_to_layer (ref: bezier_commit.rs:103-225) */
int pfx_brush_commit(pfx_ctx* ctx, uint8_t* layer_inout, const uint8_t* preview, uint32_t w, uint32_t h,
                     uint8_t blend_mode, int is_eraser, const uint8_t* selection);

/* ================= A0: TiledImage import/export rule ================= */
/* from_rgba_image + to_rgba_image: chunks with no alpha read back as zeros (ref: tiled_image.rs:50-104,271-293) */
int pfx_tiled_roundtrip(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h);
/* chunk_keys(): one byte per 64x64 chunk, row-major, 1 = populated */
int pfx_chunk_populated(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* populated);

/* ================= device-resident tier (`_dev`): same kernels, caller-owned device memory, asynchronous ========= */
/* Aliasing: `src_dev == dst_dev` (in place) is accepted by the pointwise calls (pfx_adjust_dev, pfx_lut_apply-style ops) and by
 * pfx_gaussian_blur_dev / pfx_box_blur_dev, which then run their two-pass kernels through `tmp_dev` — for the Gaussian that is the
 * f32 path, so an in-place call can differ from an out-of-place one by the +-1 LSB of the default (MFMA) mode unless the context is
 * in exact mode.  Every other call that reads a neighbourhood or gathers (median, pixelate, warps, the effect bank) returns
 * PFX_ERR_INVALID when the two images overlap; partially overlapping buffers are always refused. */
/* pfx_flatten_dev: `dst_dev` may be one of the layers (the result replaces it: every pixel is written after its last read) but must not partially
 * overlap any. */
int pfx_flatten_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const void* const* mask_ptrs_dev /* may be NULL */,
                    const pfx_layer_info* layers, uint32_t n_layers, uint32_t w, uint32_t h, void* dst_dev);
int pfx_flatten_preview_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const void* const* mask_ptrs_dev, const pfx_layer_info* layers,
                            uint32_t n_layers, uint32_t w, uint32_t h, const void* preview_dev, const void* preview_chunk_present_dev /* may be NULL */,
                            const pfx_preview* preview, void* dst_dev);
/* `tmp_dev` = w*h*16 bytes of scratch for the f32 horizontal pass (ref: filters.rs:255 buf_h); NULL = context scratch */
int pfx_gaussian_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma,
                          void* tmp_dev);
/* the same on a BAND of a taller image (rows [first_row, first_row + h) of it, halo rows included): results equal, bit for bit,
 * the rows a whole-image call produces wherever the band holds the full +-ceil(3 sigma) neighbourhood (row-band sharding) */
int pfx_gaussian_blur_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma,
                               void* tmp_dev, uint32_t first_row);
/* box blur / median of a band with its halo rows (ceil(radius) / max(radius, 1) of them on each side that is not an image edge): the
 * arithmetic is integer per pixel, so every row at least `radius` rows away from the buffer's first and last row equals the same row of
 * the whole image's filter bit for bit, wherever the band was cut; `first_row` is accepted for symmetry with the Gaussian and unused. */
int pfx_box_blur_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, const void* mask_dev,
                          void* tmp_dev, uint32_t first_row);
int pfx_median_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius, const void* mask_dev,
                        uint32_t first_row);
int pfx_box_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius,
                     const void* mask_dev, void* tmp_dev /* w*h*4 or NULL */);
int pfx_median_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius,
                   const void* mask_dev);
int pfx_pixelate_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t block_size,
                     const void* mask_dev);
int pfx_adjust_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int op,
                   const float* params, uint32_t n_params, const uint8_t* lut_host, const void* mask_dev, int sparse_mode);
int pfx_rhai_adjust_dev(pfx_ctx* ctx, void* pixels_dev, uint32_t w, uint32_t h, int op, const float* params,
                        uint32_t n_params);
/* Chains (round 6): several ops on a device-resident image in as few passes over memory as the kernels allow.  The result equals calling the single-op entry
 * points one after the other — pfx_gaussian_blur_dev / pfx_box_blur_dev (no selection) / pfx_adjust_dev (PFX_DENSE, no selection) / pfx_rhai_adjust_dev — bit for
 * bit, in the context's current Gaussian mode: every op still rounds to u8, only the trips through memory go.  A run of pointwise ops is one launch; a bit-exact
 * Gaussian of radius <= 16 followed by pointwise ops is one launch (the blurred image never exists in memory).  This is what a script's
 * `apply_gaussian_blur(4.0); apply_hsl(..); apply_invert();` (ref: src/ops/scripting.rs:869-1075 and :1095-1140, each call a full pass over the image on the CPU) and the batch
 * pipeline run through.  src_dev == dst_dev is allowed for chains without a blur. */
typedef enum pfx_chain_kind { PFX_CHAIN_ADJUST = 0 /* op = pfx_adjust_op */, PFX_CHAIN_RHAI = 1 /* op = pfx_rhai_op */, PFX_CHAIN_GAUSSIAN = 2 /* params[0] = sigma */,
                              PFX_CHAIN_BOX = 3 /* params[0] = radius */ } pfx_chain_kind;
typedef struct pfx_chain_op {
    int32_t  kind;          /* pfx_chain_kind */
    int32_t  op;            /* PFX_CHAIN_ADJUST / PFX_CHAIN_RHAI: the op id; otherwise ignored */
    uint32_t n_params;
    float    params[12];    /* as for the op's own entry point */
    const uint8_t* lut;     /* host, 1024 bytes: PFX_OP_GRADIENT_MAP / PFX_OP_LUT_RGBA; NULL otherwise */
} pfx_chain_op;
int pfx_chain_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, const pfx_chain_op* ops, uint32_t n_ops);
int pfx_warp_displacement_dev(pfx_ctx* ctx, const void* src_dev, uint32_t sw, uint32_t sh, const void* disp_dev,
                              uint32_t w, uint32_t h, void* dst_dev);
int pfx_mesh_displacement_dev(pfx_ctx* ctx, const float* orig_pts_xy, const float* deformed_pts_xy, uint32_t cols,
                              uint32_t rows, uint32_t w, uint32_t h, void* disp_dev);
int pfx_warp_mesh_catmull_rom_dev(pfx_ctx* ctx, const void* src_dev, const float* orig_pts_xy,
                                  const float* deformed_pts_xy, uint32_t cols, uint32_t rows, uint32_t w, uint32_t h,
                                  void* dst_dev);
/* Band forms of the two warps for a document sharded by rows ("replicate the source, warp bands of the output"; the CPU path's
 * `par_chunks_mut(w * 4)` rows, ref: src/ops/transform.rs:1288-1345, 1687-1761): `dst_band_dev` / `disp_band_dev` hold rows [first_row, first_row +
 * band_rows) of the output / the field, the source is whole.  Bit-identical to the same rows of the whole-image call. */
int pfx_warp_displacement_band_dev(pfx_ctx* ctx, const void* src_dev, uint32_t sw, uint32_t sh, const void* disp_band_dev, uint32_t w,
                                   uint32_t band_rows, void* dst_band_dev, uint32_t first_row);
int pfx_warp_mesh_catmull_rom_band_dev(pfx_ctx* ctx, const void* src_dev, const float* orig_pts_xy, const float* deformed_pts_xy,
                                       uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, void* dst_band_dev, uint32_t first_row,
                                       uint32_t band_rows);
int pfx_brush_stamps_dev(pfx_ctx* ctx, void* target_dev, uint32_t w, uint32_t h, const pfx_brush* brush,
                         const float* points_xy, uint32_t n_points, const void* selection_dev);
int pfx_brush_stamps_ex_dev(pfx_ctx* ctx, void* target_dev, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn,
                            const float* points_xy, uint32_t n_points, const void* selection_dev);
int pfx_tiled_roundtrip_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h);
int pfx_sharpen_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float radius,
                    const void* mask_dev);
int pfx_glow_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, float intensity,
                 const void* mask_dev);
int pfx_bokeh_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius,
                       const void* mask_dev);
int pfx_motion_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float angle_deg,
                        float distance, const void* mask_dev);
/* the rest of the effect bank on device-resident images (src_dev != dst_dev except for the purely pointwise ones) */
int pfx_zoom_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float center_x, float center_y, float strength,
                      uint32_t samples, const float tint_color[4], float tint_strength, const void* mask_dev);
int pfx_crystallize_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float cell_size, uint32_t seed,
                        const void* mask_dev);
int pfx_dents_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float scale, float amount, uint32_t seed,
                  uint32_t octaves, float roughness, int pinch, int wrap, const void* mask_dev);
int pfx_bulge_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float origin_x, float origin_y,
                  const void* mask_dev);
int pfx_twist_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float angle_deg, float origin_x, float origin_y,
                  const void* mask_dev);
int pfx_add_noise_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, int noise_type, int monochrome,
                      uint32_t seed, float scale, uint32_t octaves, const void* mask_dev);
int pfx_reduce_noise_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float strength, uint32_t radius,
                         const void* mask_dev);
int pfx_vignette_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float amount, float softness, const void* mask_dev);
int pfx_halftone_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float dot_size, float angle_deg, int shape,
                     const void* mask_dev);
int pfx_grid_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t cell_w, uint32_t cell_h, uint32_t line_width,
                 const uint8_t color[4], int style, float opacity, const void* mask_dev);
int pfx_canvas_border_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4],
                          const void* mask_dev);
int pfx_shadow_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int32_t offset_x, int32_t offset_y, float blur_radius,
                   int widen_radius, const uint8_t color[4], float opacity, const void* mask_dev);
int pfx_outline_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4], int mode,
                    int anti_alias, const void* mask_dev);
int pfx_pixel_drag_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t seed, float amount, uint32_t distance,
                       float direction, const void* mask_dev);
int pfx_rgb_displace_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int32_t r_dx, int32_t r_dy, int32_t g_dx,
                         int32_t g_dy, int32_t b_dx, int32_t b_dy, const void* mask_dev);
int pfx_ink_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float edge_strength, float threshold, const void* mask_dev);
int pfx_oil_painting_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius, uint32_t levels,
                         const void* mask_dev);
int pfx_color_filter_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, const uint8_t filter_color[4], float intensity,
                         int mode, const void* mask_dev);
int pfx_contours_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float scale, float frequency, float line_width,
                     const uint8_t line_color[4], uint32_t seed, uint32_t octaves, float blend, const void* mask_dev);

/* device self-test: compares the compositor's shared-reciprocal division with the compiler's IEEE f32 divide on
 * n_millions*1e6 random operand pairs drawn from the kernel's operand range; *mismatches must come back 0 */
int pfx_selftest_division(pfx_ctx* ctx, uint64_t seed, uint32_t n_millions, uint64_t* mismatches);
/* device self-test: the three-instruction round-and-pack (`v.round().clamp(0.0, 255.0) as u8` of every pointwise op, resampler, effect and
 * warp) against the step-by-step formula for ALL 2^32 f32 bit patterns; *mismatches must come back 0.  Signalling NaNs, which no
 * arithmetic instruction produces, are counted separately (they convert to 255 instead of 0). */
int pfx_selftest_round_pack(pfx_ctx* ctx, uint64_t* mismatches, uint64_t* signalling_nan_mismatches);

/* device self-test: the texture path's conversions the compositor relies on — a typed UNORM8 buffer store of RN(k / 255) writes the byte k and a
 * typed load of the byte k returns RN(k / 255), for all 256 k on every channel; *mismatches must come back 0 (the class-sorting compositor, which reads every
 * layer through these conversions, is only used on a device where they hold: checked once per context). */
int pfx_selftest_unorm_store(pfx_ctx* ctx, uint64_t* mismatches);

/* host-side: the f16 tap tables of the matrix-core Gaussian for `sigma` (radius 1 .. 80), as the library uploads them: three parts of 256 entries,
 * tap t at index 48 + t — [0] and [1] the two-piece split w * 256 = w1 + w2, [2] the single-piece table the default mode multiplies with (every weight
 * one f16, the table's sum nudged to the exact weights' sum).  *bias_split / *bias_single = 1024 * the tables' sums (the sample encoding's offset).
 * Returns the number of taps, or a negative PFX_ERR_* (diagnostics for tests/: the kernel's +-1 LSB bound rests on these tables). */
int pfx_gaussian_f16_tables(float sigma, uint16_t out_768[768], float* bias_split, float* bias_single);

/* development tuning knobs (kernel tile configurations); unknown keys return PFX_ERR_INVALID.  Results never change — except under "gauss_parts", which
 * selects among +-1 LSB class variants of the default-mode Gaussian (f16 pieces per weight / per horizontal result: 12 shipped, 22, 11).  The knobs that select among
 * kernel shapes of the compositor ("flatten_variant", "dle_*") are process-wide (one setting for every context), the rest per context. */
int pfx_tune(pfx_ctx* ctx, const char* key, int value);
/* work counters of the compositor's dead-layer elimination since the last reset (diagnostics for profiles/ and the tests; the call
 * synchronises the device): out[0] compacted rounds, [1] pixels in them, [2] sum of layers over rounds, [3] natural 192-pixel units,
 * [4] sum of layers over natural units, [5] candidate alpha reads (units), [6] units that used the queue, [7] reserved */
int pfx_flatten_stats(pfx_ctx* ctx, uint64_t out[8], int reset);
/* diagnostic build of the class-sorting compositor (pfx_tune "dle_stats" = 4; results stay bit-exact, the launch is slower): wave clocks (s_memtime) summed over
 * the waves of the launches since the last reset — out[0] wave lifetimes, [1] waves, [8] classification, [9] deal + early passes, [10] natural passes,
 * [11] lane order + store, [12] / [13] waiting for layer pixels / blending in the early passes, [14] / [15] the same in the natural passes.  The stand-in for an
 * instruction-level thread trace: rocprofv3 --att needs the rocprof-trace-decoder library, which this image does not hold (profiles/r05_tuning.md) */
int pfx_flatten_trace(pfx_ctx* ctx, uint64_t out[16], int reset);

/* per-launch timing of the most recent `_dev` call family, measured with HIP events on the context stream
 * (used by bench.py for the roofline line).  Enable, run, then read the accumulated milliseconds / launches. */
int pfx_timing_enable(pfx_ctx* ctx, int on);
int pfx_timing_read(pfx_ctx* ctx, const char* kernel_name, double* total_ms, uint64_t* launches);
int pfx_timing_reset(pfx_ctx* ctx);

/* ---- resize_image: imageops::resize(&flat, new_w, new_h, filter) per layer (ref: src/ops/transform.rs:347-359; script
 * function resize_image src/ops/scripting.rs:749-770).  Filters = ScriptFilterType (scripting.rs:22-37): Nearest, Bilinear
 * (Triangle), Bicubic (CatmullRom), Lanczos (Lanczos3).  The resampling algorithm is the `image` crate's (0.25.9), restated;
 * nearest / bilinear / lanczos3 are pinned by the reference's goldens, bicubic is unpinned.  Bit-exact class. */
enum { PFX_RESIZE_NEAREST = 0, PFX_RESIZE_BILINEAR = 1, PFX_RESIZE_BICUBIC = 2, PFX_RESIZE_LANCZOS3 = 3 };
int pfx_resize_image(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst /* new_w*new_h*4 */, uint32_t new_w, uint32_t new_h,
                     int filter);
int pfx_resize_image_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, uint32_t new_w, uint32_t new_h, int filter);
/* apply_affine / affine_transform_layer(state, idx, rotation_z, rotation_x, rotation_y, scale, offset) (ref: src/ops/transform.rs:750-946):
 * Rz*Ry*Rx homography about the canvas centre (angles in degrees), inverse-mapped; PFX_RESIZE_BILINEAR (the layer transform's
 * mode, against a transparent outside) or PFX_RESIZE_NEAREST.  dst is canvas_w x canvas_h; unmapped pixels are (0,0,0,0). */
int pfx_affine_transform(pfx_ctx* ctx, const uint8_t* src, uint32_t src_w, uint32_t src_h, uint8_t* dst, uint32_t canvas_w, uint32_t canvas_h,
                         float rotation_z, float rotation_x, float rotation_y, float scale, float offset_x, float offset_y, int interpolation);
int pfx_affine_transform_dev(pfx_ctx* ctx, const void* src_dev, uint32_t src_w, uint32_t src_h, void* dst_dev, uint32_t canvas_w, uint32_t canvas_h,
                             float rotation_z, float rotation_x, float rotation_y, float scale, float offset_x, float offset_y, int interpolation);

/* ================= B5/B6: script front-end and CLI (ref: src/ops/scripting.rs:1733-1821, src/cli.rs) ================= */
typedef struct pfx_script_result {
    char     error[512];      /* ScriptError::friendly_message-style text, empty on success */
    int32_t  error_line;      /* 1-based, 0 if unknown */
    int32_t  error_col;
    uint32_t ops_executed;
    char     console[2048];   /* print_line output, '\n' separated (truncated) */
} pfx_script_result;
/* execute_script_sync(source, pixels, w, h, mask) (ref: scripting.rs:1733-1821).  The script language is the Rhai subset of
 * paintfe_amd/csrc/pfx_rhai.h (let / if / loops / fn / closures, strict i64-f64 typing, checked integer arithmetic); the
 * registered host API keeps the reference's names, arity and numeric flavour (scripting.rs:323-1482).  Bulk work runs on
 * the device: effects through the kernels above, per-pixel closures (map_channels / for_each_pixel / for_region) compiled
 * to a bytecode kernel; get_pixel / set_pixel touch a host mirror of the image.  Constructs outside the subset
 * return PFX_ERR_UNSUPPORTED.  Pixels are untouched on error. */
/* fixed-size form: the script must leave the image size unchanged (PFX_ERR_UNSUPPORTED otherwise) */
int pfx_script_run(pfx_ctx* ctx, const char* source, uint8_t* pixels_inout, uint32_t w, uint32_t h,
                   const uint8_t* mask, pfx_script_result* result);
/* CanvasOpRequest (ref: scripting.rs:41-58): canvas-wide transforms the caller replays on its other layers */
enum { PFX_CANVAS_FLIP_HORIZONTAL = 0, PFX_CANVAS_FLIP_VERTICAL = 1, PFX_CANVAS_ROTATE_90CW = 2, PFX_CANVAS_ROTATE_90CCW = 3,
       PFX_CANVAS_ROTATE_180 = 4, PFX_CANVAS_RESIZE_IMAGE = 5, PFX_CANVAS_RESIZE_CANVAS = 6 };
typedef struct pfx_canvas_op {
    int32_t  kind;           /* PFX_CANVAS_* */
    uint32_t w, h;           /* RESIZE_IMAGE / RESIZE_CANVAS: new size */
    uint32_t anchor_x, anchor_y; /* RESIZE_CANVAS: 0 / 1 / 2 per axis (parse_anchor, scripting.rs:69-82); RESIZE_IMAGE: anchor_x = PFX_RESIZE_* filter */
} pfx_canvas_op;
/* the same canvas-wide transforms applied to one layer image, for replaying a pfx_canvas_op list on the other layers
 * (ref: apply_canvas_ops, scripting.rs:1640-1723; ops::transform::flip_canvas_* / rotate_canvas_* / resize_canvas, transform.rs:382-424).
 * op = PFX_CANVAS_FLIP_HORIZONTAL .. PFX_CANVAS_ROTATE_180; 90-degree rotations swap the output's width and height. */
int pfx_flip_rotate(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst, int op);
int pfx_flip_rotate_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, int op);
/* anchor 0 / 1 / 2 per axis; fill = colour of the new area (NULL = transparent) */
int pfx_resize_canvas(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst /* new_w*new_h*4 */, uint32_t new_w, uint32_t new_h,
                      uint32_t anchor_x, uint32_t anchor_y, const uint8_t fill[4]);
int pfx_resize_canvas_dev(pfx_ctx* ctx, const void* src_dev, uint32_t w, uint32_t h, void* dst_dev, uint32_t new_w, uint32_t new_h, uint32_t anchor_x,
                          uint32_t anchor_y, const uint8_t fill[4]);
/* full form: returns (result_pixels, final_w, final_h, console_output, canvas_ops) like the reference.  *out is an opaque
 * result owned by the library until pfx_script_output_free; NULL on error. */
typedef struct pfx_script_output pfx_script_output;
int pfx_script_execute(pfx_ctx* ctx, const char* source, const uint8_t* pixels, uint32_t w, uint32_t h, const uint8_t* mask,
                       pfx_script_output** out, pfx_script_result* result);
const uint8_t* pfx_script_output_pixels(const pfx_script_output* out, uint32_t* w, uint32_t* h);
uint32_t    pfx_script_output_console_lines(const pfx_script_output* out);
const char* pfx_script_output_console_line(const pfx_script_output* out, uint32_t index);
uint32_t    pfx_script_output_canvas_ops(const pfx_script_output* out, pfx_canvas_op* ops, uint32_t capacity); /* returns the count */
void        pfx_script_output_free(pfx_script_output* out);
/* language-only evaluation (no device, no image functions): syntax / typing / console checks, e.g. from an editor's
 * "compile" button (ref: compile_script, scripting.rs:1489-1508).  ctx may be NULL. */
int pfx_script_check(const char* source, uint32_t w, uint32_t h, pfx_script_result* result);
/* the `pfx` batch CLI main (same flags as src/cli.rs:43-89); returns the process exit code */
int pfx_cli_main(int argc, char** argv);
/* the CLI's PNG reader on a buffer (ref: load_image_sync -> image::open(..) -> to_rgba8, src/io.rs:693-723, called from src/cli.rs:234): every colour type and bit depth, Adam7, tRNS.
 * *rgba_out is malloc'd w*h*4 bytes, released with pfx_png_free.  Malformed input -> PFX_ERR_INVALID + message; never reads or allocates beyond what
 * the IDAT data can inflate to.  No device needed. */
int  pfx_png_decode_mem(const uint8_t* bytes, size_t n_bytes, uint8_t** rgba_out, uint32_t* w_out, uint32_t* h_out, char* err, size_t err_cap);
void pfx_png_free(uint8_t* rgba);

/* ================= N2: PFE project files and TiledImage import / export on the device =====================================
 * A .pfe file is the bincode 1.x (little-endian, fixed-width integers, u64 lengths) image of ProjectFileV0..V3
 * (ref: src/io.rs:85-208): the layer stack the compositor wants, each raster layer stored as its sparse list of 64x64 chunks.
 * pfx_project is the host-side document: size, active layer, folders, layers (name, visibility, opacity, blend mode, kind,
 * chunk list, and every V2/V3 payload kept verbatim so that re-saving loses nothing).  Parity: no .pfe fixture exists in the
 * reference tree (its tests are save->load round trips), so the byte layout is pinned by an independent Python bincode
 * restatement under tests/ — "parity unpinned" against real reference output. */
typedef struct pfx_project pfx_project;
typedef struct pfx_project_layer {
    const char* name;                /* owned by the project, valid until it is modified or freed */
    uint8_t  visible;                /* Layer::visible */
    uint8_t  effectively_visible;    /* visible and not inside a hidden folder (ref: canvas_state.rs:216-227) */
    uint8_t  blend_mode;             /* BlendMode::to_u8 as stored */
    uint8_t  layer_type;             /* 0 raster, 1 text (its rasterised chunks are used), 2 adjustment */
    float    opacity;
    int64_t  folder_id;              /* -1 = none */
    uint32_t n_chunks;               /* stored chunks */
    uint8_t  kind;                   /* pfx_layer_kind the compositor will use (an adjustment payload that fails to parse
                                        falls back to raster, io.rs:864-869) */
    uint8_t  _pad[3];
    float    adj[16];                /* parameters as in pfx_layer_info.adj */
} pfx_project_layer;

/* load_pfe_from_bytes (ref: io.rs:477-499; V3 :813, V2 :957, V1 :1110, V0 :1233): NULL + message in err on malformed input,
 * zero / oversized dimensions (validate_open_dimensions, :505), more than 256 layers, wrong chunk sizes, no layers */
pfx_project* pfx_project_load(const uint8_t* bytes, size_t n_bytes, char* err, size_t err_cap);
pfx_project* pfx_project_load_file(const char* path, char* err, size_t err_cap);
/* CanvasState::new-like empty document (no layers yet) */
pfx_project* pfx_project_new(uint32_t w, uint32_t h);
void     pfx_project_free(pfx_project* p);
uint32_t pfx_project_width(const pfx_project* p);
uint32_t pfx_project_height(const pfx_project* p);
uint32_t pfx_project_layer_count(const pfx_project* p);
uint32_t pfx_project_active_layer(const pfx_project* p);
int      pfx_project_version(const pfx_project* p);      /* 0..3: the version it was loaded from (new documents: 1) */
int      pfx_project_layer_get(const pfx_project* p, uint32_t index, pfx_project_layer* out);
/* Layer::pixels.to_rgba_image(): the layer flattened to w*h*4 on the host (missing chunks = zeros) */
int      pfx_project_layer_pixels(const pfx_project* p, uint32_t index, uint8_t* dst);
/* append a layer; rgba (w*h*4) is tiled with from_rgba_image's rule (all-transparent chunks dropped); rgba NULL = empty layer.
 * kind != PFX_LAYER_RASTER appends an adjustment layer with parameters adj[] (document becomes V3 on save). */
int      pfx_project_add_layer(pfx_project* p, const char* name, const uint8_t* rgba, float opacity, uint8_t blend_mode, uint8_t visible,
                               uint8_t kind, const float* adj);
int      pfx_project_set_active_layer(pfx_project* p, uint32_t index);
int      pfx_project_set_layer_folder(pfx_project* p, uint32_t index, int64_t folder_id /* -1 = none */);
int      pfx_project_add_folder(pfx_project* p, uint64_t id, const char* name, uint8_t visible);
/* replace a layer's pixels (and the document size when new_w/new_h differ is NOT changed: use pfx_project_resize) */
int      pfx_project_set_layer_pixels(pfx_project* p, uint32_t index, const uint8_t* rgba);
/* build_pfe + bincode::serialize (ref: io.rs:254-282): V3 if any folder / adjustment layer / experimental payload, V2 if any
 * text layer, else V1; chunks are written in row-major (cy, cx) order.  *bytes_out is malloc'ed: release with pfx_bytes_free. */
int      pfx_project_save(const pfx_project* p, uint8_t** bytes_out, size_t* n_out);
int      pfx_project_save_file(const pfx_project* p, const char* path);
void     pfx_bytes_free(uint8_t* bytes);
/* CanvasState::composite() of the document (ref: canvas_state.rs:482-698) on the device: only the stored chunks cross PCIe, a
 * kernel scatters them into flat layers (TiledImage import), then the compositor runs.  dst = w*h*4 */
int      pfx_project_composite(pfx_ctx* ctx, const pfx_project* p, uint8_t* dst);
int      pfx_project_composite_dev(pfx_ctx* ctx, const pfx_project* p, void* dst_dev);
/* run_one's script step (ref: cli.rs:238-270): the script runs on the active layer, its canvas ops are replayed on every other
 * layer (apply_canvas_ops, scripting.rs:1640-1723) and the document size follows */
int      pfx_project_run_script(pfx_ctx* ctx, pfx_project* p, const char* source, pfx_script_result* result);

/* TiledImage <-> flat image on the device.  packed = n stored chunks of 64*64*4 bytes (edge chunks zero-padded), slot[c] for
 * chunk c = cy * ceil(w/64) + cx is the chunk's index in `packed` or 0xffffffff when the TiledImage has no such chunk.
 * import = to_rgba_image (ref: tiled_image.rs:271-293), export = from_rgba_image given the populated set (:50-104). */
#define PFX_NO_CHUNK 0xffffffffu
int pfx_tiled_import_dev(pfx_ctx* ctx, const void* packed_dev, const uint32_t* slot_host, uint32_t w, uint32_t h, void* flat_dev);
int pfx_tiled_export_dev(pfx_ctx* ctx, const void* flat_dev, uint32_t w, uint32_t h, const uint32_t* slot_host, void* packed_dev);

/* ================= multi-GPU: one document across the GPUs of a node =================
 * The reference has no multi-device layer.  Its unit of independence is the 64x64 TiledImage chunk: the compositor is
 * `populated_chunks.par_iter()` (ref: src/canvas/canvas_state.rs:565), filters are row-parallel (ref: src/ops/filters.rs:258-313),
 * and one level up the CLI loops over independent files (ref: src/cli.rs:159-216).  A pfx_group maps that onto N HIP devices driven
 * by ONE process: a document is cut into bands of whole chunk rows, member k keeps its band of every layer resident, flatten needs
 * no communication, the Gaussian pulls ceil(3 sigma) rows of the flattened u8 neighbours' bands over xGMI (peer-to-peer copies
 * ordered by events), and an optional all-gather leaves the whole result on every member.  All work is asynchronous on the
 * members' streams.  Results are bit-identical to the single-GPU calls on the whole document. */
typedef struct pfx_group pfx_group;
/* rows [y0, y1) of `rank`'s band of an h-row image cut for `world` members: whole chunk rows, remainder to the first members */
void        pfx_band_rows(uint32_t h, uint32_t world, uint32_t rank, uint32_t* y0, uint32_t* y1);
/* one context per entry of `devices` (the same device may appear more than once: a 1-GPU box can exercise the whole path) */
int         pfx_group_create(const int* devices, uint32_t n, pfx_group** out);
void        pfx_group_destroy(pfx_group* g);
uint32_t    pfx_group_size(const pfx_group* g);
pfx_ctx*    pfx_group_ctx(pfx_group* g, uint32_t rank);
const char* pfx_group_last_error(const pfx_group* g);   /* never NULL */
/* allocate the members' bands for a w x h document of n_layers raster layers */
int         pfx_group_set_document(pfx_group* g, uint32_t w, uint32_t h, uint32_t n_layers);
int         pfx_group_band(const pfx_group* g, uint32_t rank, uint32_t* y0, uint32_t* y1);
/* scatter one full-size host layer (w*h*4) to the members' bands; blocking */
int         pfx_group_upload_layer(pfx_group* g, uint32_t index, const uint8_t* rgba_host);
/* device pointer of member `rank`'s band of layer `index` ((y1-y0)*w*4 bytes), for producers that already live on the device */
void*       pfx_group_layer_band_dev(pfx_group* g, uint32_t rank, uint32_t index);
/* CanvasState::composite() of the document followed by parallel_gaussian_blur(sigma) (sigma <= 0: flatten only).
 * layers[k].layer_idx indexes the document's layers.  all_gather != 0: afterwards every member holds the whole result. */
int         pfx_group_flatten_blur(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, float sigma, int all_gather);
/* the same with any of the band filters behind the flatten (the Gaussian, box and median vertical passes need halo rows):
 * PFX_BAND_GAUSSIAN param = sigma (halo ceil(3 sigma)); PFX_BAND_BOX param = radius (halo ceil(radius), ref: blur.rs:233-318);
 * PFX_BAND_MEDIAN param = radius (halo max(radius, 1), ref: noise.rs:357-410); PFX_BAND_NONE = flatten only */
enum { PFX_BAND_NONE = 0, PFX_BAND_GAUSSIAN = 1, PFX_BAND_BOX = 2, PFX_BAND_MEDIAN = 3 };
int         pfx_group_flatten_filter(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, int filter, float param, int all_gather);
/* how halo rows and gathered bands travel between the members.  PFX_GROUP_PEER (default): hipMemcpyPeerAsync over xGMI, pair by pair
 * replaced by staged copies where peer access cannot be enabled; PFX_GROUP_STAGED: always through pinned host memory;
 * PFX_GROUP_RCCL: one ncclSend / ncclRecv group for the halos and one ncclBroadcast group (one broadcast per ragged band) for the
 * all-gather — librccl is loaded with dlopen on first use (PFX_ERR_UNSUPPORTED if it is missing or two members share a device).
 * Results are identical under every transport. */
enum { PFX_GROUP_PEER = 0, PFX_GROUP_RCCL = 1, PFX_GROUP_STAGED = 2 };
int         pfx_group_set_transport(pfx_group* g, int transport);
int         pfx_group_transport(const pfx_group* g);
int         pfx_group_synchronize(pfx_group* g);
/* Watchdog: the next `calls` pipeline calls (0xFFFFFFFF: all) end with a host-side wait of at most `timeout_ms` for every member's streams; a member
 * that does not finish turns a hang (a mis-paired RCCL group, a peer that never sends) into PFX_ERR_HIP whose message names the busy members and the
 * halo transfers (consumer <- producer, rows) they take part in.  An RCCL transport that times out has its communicators aborted and is replaced by
 * PEER.  timeout_ms == 0 switches it off.  Selecting PFX_GROUP_RCCL arms it for the first two calls (20 s) unless it was configured before. */
int         pfx_group_set_watchdog(pfx_group* g, uint32_t timeout_ms, uint32_t calls);
int         pfx_group_synchronize_timeout(pfx_group* g, uint32_t timeout_ms); /* pfx_group_synchronize with the same bounded wait and report */
/* Per-phase clocks of the pipeline calls (off by default): with timing on, every member records HIP events around its phases, and pfx_group_phase_ms returns
 * for member `rank` of the LAST call (it waits for that member's work): out_ms[0] flatten, [1] from the end of its flatten to the arrival of its last halo row
 * (waiting for the neighbours' flattens + the transfer), [2] the band filter, [3] its all-gather pushes (0 without all_gather).  What a first run on a real
 * multi-GPU node should print beside the step time (bench.py --gpus N: "c_abi_group"). */
int         pfx_group_set_phase_timing(pfx_group* g, int on);
int         pfx_group_phase_ms(pfx_group* g, uint32_t rank, double out_ms[4]);
/* Warps of the sharded document ("replicate the source, warp bands of the output"; ref: src/ops/transform.rs:1288-1345, 1687-1761): the
 * flattened bands are all-gathered so that every member holds the whole source, then every member warps its band of the output with
 * pfx_warp_displacement_band_dev (`disp_host` = w*h xy pairs, scattered to the members by rows; blocking until the field has been copied) or
 * pfx_warp_mesh_catmull_rom_band_dev.  The result stays sharded (pfx_group_result_band_dev / pfx_group_download); bit-identical to the single-GPU calls. */
int         pfx_group_flatten_warp_displacement(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, const float* disp_host);
int         pfx_group_flatten_warp_mesh(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, const float* orig_pts_xy /* may be NULL */,
                                        const float* deformed_pts_xy, uint32_t cols, uint32_t rows);
void*       pfx_group_result_band_dev(pfx_group* g, uint32_t rank);  /* rows [y0, y1) of the last result on member `rank` */
void*       pfx_group_gathered_dev(pfx_group* g, uint32_t rank);     /* w*h*4 on member `rank` after an all_gather call */
int         pfx_group_download(pfx_group* g, uint8_t* dst_host);     /* concatenated bands -> host w*h*4; blocking */

/* ================= batch of independent images across the GPUs of a node (BASELINE config 5; ref: src/cli.rs:159-216) =========
 * The reference's CLI runs `run_one` per input file, serially.  pfx_batch_pipeline streams a batch through the devices instead:
 * image i goes to device i mod n_devices (no data-path collective), and on every device a ring of `slots` buffer sets keeps the
 * upload of one image, the kernels of another and the download of a third in flight (pinned host memory; one in-order stream each
 * for uploads, kernels and downloads, chained by events).  Per image:
 * parallel_gaussian_blur(sigma) -> hue_saturation_lightness(hue, saturation, lightness) -> CanvasState::composite() of the result
 * under `n_overlays` overlay layers (resident on every device).  Sources are taken round-robin from a pool of host images
 * (image i = pool[i mod n_pool]; the pool is pinned for the duration of the call); results of the images listed in keep_indices
 * are copied to keep_out[k] (w*h*4 each) — the rest only cross PCIe. */
typedef struct pfx_batch_params {
    uint32_t w, h;
    float    sigma;
    float    hue, saturation, lightness;
    uint32_t n_overlays;                  /* <= 8 */
    const uint8_t* const* overlays_host;  /* n_overlays x w*h*4 */
    const uint8_t* overlay_modes;         /* BlendMode::to_u8 per overlay */
    const float*   overlay_opacity;       /* NULL = 1.0 */
    uint32_t slots;                       /* buffer sets per device, 2..8 (0 = 3: measured best; more sets let the upload queue run ahead and the rate drops) */
    uint32_t n_keep;
    const uint32_t* keep_indices;
    uint8_t* const* keep_out;
    uint32_t out_of_contract_fast_gaussian; /* 0 (default) = the bit-exact f32 Gaussian: every result equals the CPU path exactly.  1 leaves the +-1 LSB contract: the
                                             default-mode Gaussian (f16 taps, +-1 LSB) feeds HSL, which amplifies it — up to 4 LSB in the result (measured); a
                                             development / comparison switch, never a production setting (the stream is PCIe-bound either way) */
} pfx_batch_params;
typedef struct pfx_batch_stats {
    double   seconds;              /* first enqueue .. last result back on the host, slowest device */
    double   images_per_s;         /* whole job */
    double   h2d_gbs, d2h_gbs;     /* whole job, each direction */
    double   kernel_ms_per_image;  /* blur + HSL + flatten on resident data (HIP events on every 8th image) */
    uint32_t images, devices;
} pfx_batch_stats;
int pfx_batch_pipeline(const int* devices, uint32_t n_devices, uint32_t n_images, const pfx_batch_params* params,
                       const uint8_t* const* image_pool_host, uint32_t n_pool, pfx_batch_stats* stats, char* err, size_t err_cap);

#ifdef __cplusplus
}
#endif
#endif /* PFX_H */
