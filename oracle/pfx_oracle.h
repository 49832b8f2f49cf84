/*
 * pfx_oracle.h — CPU restatement ("oracle") of PaintFE's raster pixel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / the timed CPU baseline.
 * The product path (paintfe_amd/csrc, libpfx.so) never links or calls it.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_*.py,
 * `-m "not gpu"`) against the reference's own committed golden images
 * (reference tests/golden/<category>/<name>.png, converted to tests/golden/golden.npz by
 * tests/golden/make_fixtures.py) with tolerance 0, exactly as the reference's
 * assert_golden does (reference tests/common/mod.rs:181-186,211-263).
 * The reference itself (Rust, edition 2024) cannot be compiled in this image
 * (no cargo/rustc), so there is no oracle/_ref build.
 *
 * Arithmetic rules followed throughout (reference is Rust, which never
 * contracts a*b+c into an FMA and evaluates f32 expressions in f32):
 *   - build with -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile);
 *   - `x as u8` on f32  = saturating truncation toward zero, NaN -> 0;
 *   - f32::round()      = roundf (half away from zero);
 *   - f32::exp/powf/sqrt = glibc expf/powf/sqrtf (what Rust calls on Linux).
 *
 * All images are tight row-major straight-alpha RGBA8 (image::RgbaImage layout).
 * File:line citations are relative to the reference checkout.
 */
#ifndef PFX_ORACLE_H
#define PFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFXO_CHUNK 64 /* src/canvas/defs.rs:7 CHUNK_SIZE */

/* ---- A0: TiledImage sparsity (src/canvas/tiled_image.rs:50-104, 271-293) ---- */
/* One byte per 64x64 chunk: 1 iff any pixel in the chunk has alpha != 0. */
void pfxo_chunk_populated(const uint8_t* rgba, uint32_t w, uint32_t h, uint8_t* populated);
/* from_rgba_image followed by to_rgba_image: unpopulated chunks become all-zero. */
void pfxo_tiled_roundtrip(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst);

/* ---- A1/A2/A14: compositor (src/canvas/canvas_state.rs:505-698,1246-1505) ---- */
enum { PFXO_LAYER_RASTER = 0, PFXO_ADJ_EXPOSURE = 1, PFXO_ADJ_BRIGHTNESS_CONTRAST = 2,
       PFXO_ADJ_INVERT = 3, PFXO_ADJ_CHANNEL_MIXER = 4 };

typedef struct pfxo_layer {
    const uint8_t* pixels;   /* w*h*4 RGBA8 (raster layers); may be NULL for adjustment layers */
    const uint8_t* mask;     /* optional w*h bytes: "conceal" (= mask TiledImage alpha), NULL = none */
    float opacity;
    uint8_t blend_mode;      /* BlendMode::to_u8, src/canvas/layers.rs:125-153 */
    uint8_t visible;         /* layer_effectively_visible */
    uint8_t kind;            /* PFXO_LAYER_RASTER or PFXO_ADJ_* (src/canvas/layers.rs:249-262) */
    uint8_t _pad;
    float adj[16];           /* Exposure: [ev]; B/C: [brightness, contrast]; ChannelMixer: red[4] green[4] blue[4] alpha[4] */
} pfxo_layer;

void pfxo_blend_pixel(const uint8_t base[4], const uint8_t top[4], int mode, float opacity, uint8_t out[4]);
/* CanvasState::composite(): flatten bottom->top.  threads<=0 -> all cores (chunk-parallel like rayon). */
void pfxo_composite(const pfxo_layer* layers, int n_layers, uint32_t w, uint32_t h, uint8_t* dst, int threads);
/* the same with a tool preview layer folded into the active layer (src/canvas/canvas_state.rs:541-548,593-658) */
typedef struct pfxo_preview {
    const uint8_t* pixels;        /* w*h*4, the preview TiledImage flattened */
    const uint8_t* chunk_present; /* one byte per 64x64 chunk (row-major), or NULL = chunks with any alpha != 0 */
    int32_t active_layer;         /* index into layers[] */
    uint8_t blend_mode;           /* preview_blend_mode */
    uint8_t is_eraser;            /* preview_is_eraser */
    uint8_t replaces_layer;       /* preview_replaces_layer */
    uint8_t _pad;
} pfxo_preview;
void pfxo_composite_preview(const pfxo_layer* layers, int n_layers, uint32_t w, uint32_t h, const pfxo_preview* preview, uint8_t* dst, int threads);
/* dense flavour used by the benchmark baseline: n layers stored back to back (layer stride w*h*4) */
/* 1 = the reference's collect + single-threaded put_pixel write-back (canvas_state.rs:686-695); 0 = parallel write-back */
void pfxo_set_serial_writeback(int on);
void pfxo_flatten_stack(const uint8_t* stack, int n_layers, const uint8_t* modes, const float* opacities,
                        uint32_t w, uint32_t h, uint8_t* dst, int threads);

/* ---- A3: Gaussian blur (src/ops/filters.rs:141-316) ---- */
/* returns kernel length (2r+1) written to `out` (caller provides >= 2*ceil(3*sigma)+1 floats) */
int  pfxo_gaussian_kernel(float sigma, float* out, int cap);
void pfxo_gaussian_blur(const uint8_t* src, uint32_t w, uint32_t h, float sigma, uint8_t* dst, int threads);
void pfxo_blur_with_selection(const uint8_t* src, uint32_t w, uint32_t h, float sigma,
                              const uint8_t* mask /* w*h or NULL */, uint8_t* dst, int threads);

/* ---- A4/A5/A6: box blur, median, pixelate (src/ops/effects/{blur,noise,distort}.rs) ---- */
void pfxo_box_blur(const uint8_t* src, uint32_t w, uint32_t h, float radius, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_median(const uint8_t* src, uint32_t w, uint32_t h, uint32_t radius, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_pixelate(const uint8_t* src, uint32_t w, uint32_t h, uint32_t block, const uint8_t* mask, uint8_t* dst, int threads);

/* ---- N3: effects that reuse the same kernels (src/ops/effects/stylize.rs:26-143, blur.rs:22-210) ---- */
void pfxo_glow(const uint8_t* src, uint32_t w, uint32_t h, float radius, float intensity, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_sharpen(const uint8_t* src, uint32_t w, uint32_t h, float amount, float radius, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_bokeh_blur(const uint8_t* src, uint32_t w, uint32_t h, float radius, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_motion_blur(const uint8_t* src, uint32_t w, uint32_t h, float angle_deg, float distance, const uint8_t* mask, uint8_t* dst, int threads);

/* the rest of the effect bank (o_effects2.c; src/ops/effects/{blur,distort,noise,stylize,render,glitch,artistic,contours}.rs) */
float pfxo_hash_f32(uint32_t x, uint32_t y, uint32_t seed);
float pfxo_turbulence_2d(float x, float y, uint32_t seed, uint32_t octaves, float roughness);
void pfxo_zoom_blur(const uint8_t* src, uint32_t w, uint32_t h, float center_x, float center_y, float strength, uint32_t samples,
                    const float tint_color[4], float tint_strength, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_crystallize(const uint8_t* src, uint32_t w, uint32_t h, float cell_size, uint32_t seed, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_dents(const uint8_t* src, uint32_t w, uint32_t h, float scale, float amount, uint32_t seed, uint32_t octaves, float roughness,
                int pinch, int wrap, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_bulge(const uint8_t* src, uint32_t w, uint32_t h, float amount, float origin_x, float origin_y, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_twist(const uint8_t* src, uint32_t w, uint32_t h, float angle_deg, float origin_x, float origin_y, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_add_noise(const uint8_t* src, uint32_t w, uint32_t h, float amount, int noise_type /* 0 uniform 1 gaussian 2 perlin */, int monochrome,
                    uint32_t seed, float scale, uint32_t octaves, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_reduce_noise(const uint8_t* src, uint32_t w, uint32_t h, float strength, uint32_t radius, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_vignette(const uint8_t* src, uint32_t w, uint32_t h, float amount, float softness, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_halftone(const uint8_t* src, uint32_t w, uint32_t h, float dot_size, float angle_deg, int shape /* 0 circle 1 square 2 diamond 3 line */,
                   const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_grid(const uint8_t* src, uint32_t w, uint32_t h, uint32_t cell_w, uint32_t cell_h, uint32_t line_width, const uint8_t color[4],
               int style /* 0 lines 1 checkerboard */, float opacity, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_canvas_border(const uint8_t* src, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4], const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_drop_shadow(const uint8_t* src, uint32_t w, uint32_t h, int32_t offset_x, int32_t offset_y, float blur_radius, int widen_radius,
                      const uint8_t color[4], float opacity, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_outline(const uint8_t* src, uint32_t w, uint32_t h, uint32_t width, const uint8_t color[4], int mode /* 0 outside 1 inside 2 center */,
                  int anti_alias, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_pixel_drag(const uint8_t* src, uint32_t w, uint32_t h, uint32_t seed, float amount, uint32_t distance, float direction,
                     const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_rgb_displace(const uint8_t* src, uint32_t w, uint32_t h, const int32_t off_rx_ry_gx_gy_bx_by[6], const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_ink(const uint8_t* src, uint32_t w, uint32_t h, float edge_strength, float threshold, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_oil_painting(const uint8_t* src, uint32_t w, uint32_t h, uint32_t radius, uint32_t levels, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_color_filter(const uint8_t* src, uint32_t w, uint32_t h, const uint8_t filter_color[4], float intensity,
                       int mode /* 0 multiply 1 screen 2 overlay 3 soft light */, const uint8_t* mask, uint8_t* dst, int threads);
void pfxo_contours(const uint8_t* src, uint32_t w, uint32_t h, float scale, float frequency, float line_width, const uint8_t line_color[4],
                   uint32_t seed, uint32_t octaves, float blend, const uint8_t* mask, uint8_t* dst, int threads);

/* imageops::resize of the `image` crate 0.25.9 as called by resize_image (o_resize.c; src/ops/transform.rs:347-359) */
enum { PFXO_RESIZE_NEAREST = 0, PFXO_RESIZE_BILINEAR = 1, PFXO_RESIZE_BICUBIC = 2, PFXO_RESIZE_LANCZOS3 = 3 };
size_t pfxo_resize_weights(uint32_t n_in, uint32_t n_out, int filter, uint32_t* left, uint32_t* count, size_t* off, float* wts);
void pfxo_resize(const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, int filter, uint8_t* dst, int threads);
void pfxo_flip_rotate(const uint8_t* src, uint32_t w, uint32_t h, int op /* 0 flip h, 1 flip v, 2 90cw, 3 90ccw, 4 180 */, uint8_t* dst);
void pfxo_resize_canvas(const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t anchor_x, uint32_t anchor_y, const uint8_t fill[4],
                        uint8_t* dst);

/* layer affine / perspective resampler (o_affine.c; src/ops/transform.rs:750-976); interpolation: 0 nearest, else bilinear */
void pfxo_affine_matrix(uint32_t canvas_w, uint32_t canvas_h, float rotation_z, float rotation_x, float rotation_y, float hi_out[9]);
void pfxo_affine(const uint8_t* src, uint32_t sw, uint32_t sh, uint32_t canvas_w, uint32_t canvas_h, float rotation_z, float rotation_x, float rotation_y,
                 float scale, float offset_x, float offset_y, int interpolation, uint8_t* dst, int threads);

/* ---- A7/A8: ops::adjustments flavour (f32, .round()) (src/ops/adjustments.rs, src/ops/filters.rs:321) ---- */
enum {
    PFXO_OP_INVERT = 0, PFXO_OP_INVERT_ALPHA, PFXO_OP_SEPIA, PFXO_OP_BRIGHTNESS_CONTRAST, PFXO_OP_HSL,
    PFXO_OP_EXPOSURE, PFXO_OP_HIGHLIGHTS_SHADOWS, PFXO_OP_TEMPERATURE_TINT, PFXO_OP_THRESHOLD, PFXO_OP_POSTERIZE,
    PFXO_OP_COLOR_BALANCE, PFXO_OP_GRADIENT_MAP, PFXO_OP_BLACK_AND_WHITE, PFXO_OP_VIBRANCE, PFXO_OP_LUT_RGBA,
    PFXO_OP_DESATURATE, PFXO_OP_COUNT
};
enum { PFXO_DENSE = 0, PFXO_FROM_FLAT = 1, PFXO_IN_PLACE = 2 };
/* One driver for the whole bank; `params` layout per op is documented in o_adjust.c:px_fn.
 * `lut`: 4x256 (PFXO_OP_LUT_RGBA) or 256x4 RGBA (PFXO_OP_GRADIENT_MAP), else NULL. mask: w*h, 0 = keep. */
void pfxo_adjust(const uint8_t* src, uint32_t w, uint32_t h, int op, const float* params, const uint8_t* lut,
                 const uint8_t* mask, int sparse_mode, uint8_t* dst, int threads);
/* LUT builders (host-side in the reference too) */
void pfxo_levels_lut(float in_black, float in_white, float gamma, float out_black, float out_white, uint8_t lut[256]);
void pfxo_rhai_levels_lut(float in_black, float in_white, float gamma, uint8_t lut[256]);
void pfxo_stretch_lut(uint8_t min, uint8_t max, uint8_t lut[256]);
void pfxo_curves_lut(const float* pts_xy, int n_pts, uint8_t lut[256]);
void pfxo_curves_luts_multi(const float* const pts[5], const int n[5], const int enabled[5], uint8_t out[4 * 256]);
void pfxo_auto_levels_luts(const uint8_t* src, uint32_t w, uint32_t h, const uint8_t* mask, uint8_t out[4 * 256]);
void pfxo_rgb_to_hsl(float r, float g, float b, float* h, float* s, float* l);
void pfxo_hsl_to_rgb(float h, float s, float l, float* r, float* g, float* b);

/* ---- A9: Rhai-inline flavour (truncating `as u8`, alpha untouched, mask ignored) (src/ops/scripting.rs:869-1075) ---- */
enum {
    PFXO_RHAI_INVERT = 0, PFXO_RHAI_DESATURATE, PFXO_RHAI_SEPIA, PFXO_RHAI_SEPIA_STRENGTH,
    PFXO_RHAI_BRIGHTNESS_CONTRAST, PFXO_RHAI_HSL, PFXO_RHAI_EXPOSURE, PFXO_RHAI_LEVELS, PFXO_RHAI_COUNT
};
void pfxo_rhai_adjust(uint8_t* px, size_t n_px, int op, const float* params);

/* ---- A10-A12: warp (src/ops/transform.rs:1015-1345,1558-1761) ---- */
void pfxo_catmull_rom_weights(float t, float w[4]);
void pfxo_catmull_rom_surface(const float* pts_xy, uint32_t cols, uint32_t rows, float u_global, float v_global,
                              float out[2]);
void pfxo_mesh_displacement(const float* orig_pts_xy, const float* def_pts_xy, uint32_t cols, uint32_t rows,
                            uint32_t w, uint32_t h, float* disp_xy /* w*h*2 */, int threads);
void pfxo_mesh_displacement_fast(const float* def_pts_xy, uint32_t cols, uint32_t rows, uint32_t w, uint32_t h,
                                 float* disp_xy, int threads);
void pfxo_warp_displacement(const uint8_t* src, uint32_t w, uint32_t h, const float* disp_xy, uint8_t* dst, int threads);
void pfxo_warp_displacement_ex(const uint8_t* src, uint32_t sw, uint32_t sh, const float* disp_xy,
                               uint32_t w, uint32_t h, uint8_t* dst, int threads);
void pfxo_warp_mesh_catmull_rom(const uint8_t* src, const float* orig_pts_xy, const float* def_pts_xy,
                                uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint8_t* dst, int threads);
/* DisplacementField brushes; mode: 0 push, 1 expand, 2 contract, 3 twirl cw, 4 twirl ccw */
void pfxo_displacement_brush(float* disp_xy, uint32_t w, uint32_t h, int mode, float cx, float cy,
                             float delta_x, float delta_y, float radius, float strength);

/* ---- A13: brush stamps (src/ui/panels/tools/behavior/raster/brush_render.rs) ---- */
enum { PFXO_BRUSH_NORMAL = 0, PFXO_BRUSH_DODGE = 1, PFXO_BRUSH_BURN = 2, PFXO_BRUSH_SPONGE = 3 };
typedef struct pfxo_brush {
    float size;          /* diameter (ToolProperties::size) */
    float hardness;      /* 0..1 */
    float flow;          /* 0..1 */
    float color[4];      /* brush colour, straight RGBA in 0..1 (primary/secondary_color_f32) */
    int   anti_aliased;
    int   is_eraser;
    int   mode;          /* PFXO_BRUSH_* (BrushMode) */
} pfxo_brush;
/* brush dynamics: the ToolProperties fields draw_circle_no_dirty also reads (state.rs:112-128) + ToolsPanel::stamp_counter */
typedef struct pfxo_brush_dyn {
    float scatter, hue_jitter, brightness_jitter;
    uint32_t stamp_counter;
    const uint8_t* tip_mask;     /* brush_tip_mask rescaled to the brush size (pfxo_brush_tip_rescale); NULL = circle tip */
    uint32_t tip_mask_size;
    float tip_rotation;          /* degrees */
    int32_t tip_random_rotation;
    float tip_rotation_lo, tip_rotation_hi;
} pfxo_brush_dyn;
void pfxo_brush_stamp_ex(uint8_t* target, uint32_t w, uint32_t h, const pfxo_brush* b, const pfxo_brush_dyn* d, float px, float py,
                         const uint8_t* selection);
uint32_t pfxo_brush_tip_rescale(const uint8_t* src_mask, uint32_t src_size, float brush_size, float hardness, uint8_t* out);
float pfxo_brush_alpha(float dist, float radius, float hardness, int anti_aliased);
void pfxo_brush_lut(float size, float hardness, int anti_aliased, uint8_t lut[256]);
void pfxo_brush_stamp(uint8_t* target, uint32_t w, uint32_t h, const pfxo_brush* b, float cx, float cy,
                      const uint8_t* selection /* w*h or NULL */);
int  pfxo_brush_line_points(float x0, float y0, float x1, float y1, uint32_t w, uint32_t h, float* out_xy, int cap);
void pfxo_brush_line(uint8_t* target, uint32_t w, uint32_t h, const pfxo_brush* b,
                     float x0, float y0, float x1, float y1, const uint8_t* selection);
/* stroke commit = blend_pixel_static(layer, preview, mode, 1.0) where preview.a>0 (bezier_commit.rs:103-161) */
void pfxo_brush_commit(uint8_t* layer, const uint8_t* preview, uint32_t w, uint32_t h, int mode,
                       const uint8_t* selection);
void pfxo_eraser_commit(uint8_t* layer, const uint8_t* preview, uint32_t w, uint32_t h, const uint8_t* selection);

int pfxo_version(void);

#ifdef __cplusplus
}
#endif
#endif
