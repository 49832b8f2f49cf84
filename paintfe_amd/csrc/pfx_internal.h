// pfx_internal.h — context object and host-side helpers behind the C ABI (include/pfx.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/pfx.h"
#include "pfx_kernels.h"

struct pfx_devbuf { // grow-on-demand device allocation (never shrinks; freed with the context)
    void* p = nullptr;
    size_t cap = 0;
};

struct pfx_layer_state { // GpuLayerState (ref: src/gpu/renderer.rs:206-209): device-resident, versioned
    pfx_devbuf pixels;
    pfx_devbuf mask;
    pfx_devbuf chunk_flags; // per 64 x 64 chunk: bit 0 = every alpha 255, bit 1 = no alpha 0 (pfxk_chunk_alpha_flags; refreshed with the pixels)
    bool has_mask = false;
    uint32_t w = 0, h = 0;
    uint64_t generation = 0;
};

struct pfx_timing_rec {
    std::string name;
    hipEvent_t start, stop;
};

struct pfx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    bool exact = false;
    // sharpen / glow / drop shadow feed a Gaussian into a gain (stylize.rs:96-143, :26-70, render.rs:297): a +-1 LSB input would come out as +-amount, so these
    // effects run the bit-exact Gaussian whatever `exact` says, unless the caller opts out (pfx_tune "gauss_fast_effects" = 1)
    bool gauss_fast_effects = false;
    bool resize_two_pass = false;       // pfx_tune("resize_two_pass"): keep the resamplers' f32 intermediate in HBM (the pre-fusion path)
    std::string err;
    // staging for the host-buffer tier (the reference keeps cached staging/ping-pong textures the same way,
    // ref: src/gpu/renderer.rs:232-236)
    pfx_devbuf st_in, st_out, st_mask, st_tmp, st_aux, st_aux2;
    bool outline_bits = true;  // pfx_tune "outline_bits": the outline's window search on a bit plane of alpha != 0 (0: the per-element scan)
    bool brush_binning = true; // pfx_tune "brush_binning": strokes of more than 64 stamps are dealt to 64 x 64 chunks on the host (pfx_brush_stamps_ex_dev)
    int median_bits_min = 3;   // pfx_tune "median_bits_min": radii >= this (and <= 8) take k_median_bits.hip (r = 2: 0.35 ms against the networks' 0.22)
    pfx_devbuf fx_a, fx_b; // effect-bank scratch (crystallize cell table, drop-shadow planes)
    // small parameter buffers
    pfx_devbuf d_desc, d_adj, d_chunks, d_wts, d_lut, d_pts, d_misc;
    bool shadow_plane_blur = true;       // pfx_tune "shadow_plane": the drop shadow blurs its one-channel alpha plane (1) or the reference's (a, a, a, a) image (0); same bits
    bool chain_fuse_heavy = false;       // pfx_tune "chain_fuse_heavy": HSL / vibrance ride in a Gaussian's store as well (A/B; parity tests run both)
    bool chain_mfma_epilogue = true;     // pfx_tune "chain_mfma": 0 = a default-mode Gaussian in a chain runs as its own launch (A/B of the fused store)
    pfx_devbuf st_chain, d_chain_luts;   // pfx_chain_dev: ping-pong image between two stencil stages; the tables of a chain's LUT ops (PFXK_CHAIN_LUTS x 1024)
    pfx_devbuf d_chunk_meta, d_chunk_start;     // per-layer summary pointers + wanted bits, and the per-chunk start table built from them
    std::vector<uint8_t> chunk_meta_cache;
    // whether the table of the stack in chunk_meta_cache skips anything: written by the table kernel into pinned memory (tag), read on a later
    // composite of the same stack once `ev_chunk_useful` has fired — 0 = not built, 1 = pending, 2 = useful (table kept), 3 = useless (no table)
    uint32_t* h_chunk_useful = nullptr;
    hipEvent_t ev_chunk_useful = nullptr;
    // shallow stacks (below dle_min_layers) with a reset layer: whether the elimination kernel pays depends on how coherent the reset layer's alpha is, which a
    // probe kernel samples once per stack; its verdict arrives through pinned memory and is used from a later composite of the same stack on
    // 0 = no probe, 1 = pending, 2 = elimination pays, 3 = it does not
    int dle_probe_state = 0;
    uint32_t* h_dle_verdict = nullptr;
    uint32_t dle_probe_tag = 0;
    hipEvent_t ev_dle_probe = nullptr;
    std::vector<uint8_t> dle_probe_key;   // the probed stack's descriptors + size
    bool dle_adaptive = true;             // pfx_tune "dle_adaptive"
    uint32_t chunk_tag = 0;
    int chunk_state = 0;
    uint64_t store_epoch = 0, chunk_epoch = 0;  // store_epoch moves whenever a stored layer's pixels or mask change
    std::vector<uint8_t> desc_cache, adj_cache; // host copies of what d_desc / d_adj hold (build_stack skips identical uploads)
    std::map<uint32_t, pfx_layer_state> layers;
    bool timing = false;
    std::vector<pfx_timing_rec> timings;
    int n_cus = 0;                  // multiProcessorCount of `device` (persistent-kernel grids)
    // Gaussian tap weights currently resident in d_wts / d_wsplit (re-uploaded only when sigma changes)
    uint32_t wts_sigma_bits = 0, wsplit_sigma_bits = 0;
    bool use_chunk_start = true; // stored layers: per-chunk start table from their alpha summaries (pfx_tune "chunk_start")
    int unorm_store_ok = -1;  // -1 not checked yet, 1 = typed UNORM8 stores round-trip RN(k / 255) exactly on this device (pfxk_unorm_store_check), 0 = they do not
    int stack_mode_class = 0; // set by build_stack: 0 heavy / unknown, 1 medium, 2 light blend arithmetic (pfxk_flatten picks the streaming kernel's shape from it)
    int dle_min_layers = 16;  // stacks at least this deep may take the compositor's dead-layer elimination kernel (pfx_api.cpp:build_stack)
    bool wts_valid = false, wsplit_valid = false; // explicit flags: every 32-bit pattern is some sigma (0xffffffff is a NaN)
    float wsplit_inv_scale = 1.0f, wsplit_bias = 0.0f, wsplit_bias_single = 0.0f;
    pfx_devbuf d_wsplit;
    // GpuLiquifyPipeline's cached source texture (ref: src/gpu/compute/liquify.rs:166-176); 0 x 0 = none / invalidated
    pfx_devbuf warp_src;
    uint32_t warp_src_w = 0, warp_src_h = 0;
};

// ---- error plumbing ----
int pfx_fail(pfx_ctx* ctx, int status, const char* fmt, ...);
#define PFX_HIP(ctx, call)                                                                      \
    do {                                                                                        \
        hipError_t _e = (call);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return pfx_fail((ctx), _e == hipErrorOutOfMemory ? PFX_ERR_OOM : PFX_ERR_HIP,       \
                            "%s failed: %s", #call, hipGetErrorString(_e));                     \
    } while (0)
#define PFX_TRY(expr)            \
    do {                         \
        int _s = (expr);         \
        if (_s != PFX_OK) return _s; \
    } while (0)
#define PFX_REQUIRE(ctx, cond, msg) \
    do {                            \
        if (!(cond)) return pfx_fail((ctx), PFX_ERR_INVALID, "%s", (msg)); \
    } while (0)

// do [a, a + a_bytes) and [b, b + b_bytes) share a byte?  (aliasing rule of the `_dev` entry points, include/pfx.h)
// The document-size limit every entry point enforces: non-zero sides and at most 256 M pixels (TiledImage::new clamps beyond that, ref: src/canvas/tiled_image.rs:15-26;
// io.rs:500 refuses sides over 25 000).  The kernels index pixels with 32-bit arithmetic under this bound.
inline bool pfx_dims_ok(uint32_t w, uint32_t h) { return w != 0 && h != 0 && (uint64_t)w * h <= 256000000ull; }
// [x, x + rw) x [y, y + rh) inside w x h, without the 32-bit wrap of x + rw
inline bool pfx_rect_inside(uint32_t x, uint32_t y, uint32_t rw, uint32_t rh, uint32_t w, uint32_t h)
{
    return rw != 0 && rh != 0 && (uint64_t)x + rw <= w && (uint64_t)y + rh <= h;
}

inline bool pfx_ranges_overlap(const void* a, size_t a_bytes, const void* b, size_t b_bytes)
{
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + b_bytes && y < x + a_bytes;
}

// ---- helpers (pfx_ctx.cpp) ----
int pfx_use(pfx_ctx* ctx);                                     // hipSetDevice
int pfx_reserve(pfx_ctx* ctx, pfx_devbuf& b, size_t bytes);    // grow-on-demand
int pfx_h2d(pfx_ctx* ctx, void* dst, const void* src, size_t bytes);
int pfx_d2h(pfx_ctx* ctx, void* dst, const void* src, size_t bytes);
int pfx_sync(pfx_ctx* ctx);

// RAII-less timing scope: records two events around a launch sequence when ctx->timing is on
struct pfx_timer {
    pfx_ctx* ctx;
    bool on;
    pfx_timing_rec rec;
    pfx_timer(pfx_ctx* c, const char* name);
    ~pfx_timer();
};

// flatten of device-resident flat layers + the union of the visible layers' TiledImage chunk keys (pfx_api.cpp)
extern "C" int pfx_int_flatten_with_chunk_keys_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const pfx_layer_info* layers, uint32_t n_layers,
                                                   uint32_t w, uint32_t h, void* dst_dev, const uint8_t* chunk_keys_host);

// blur_with_selection on device-resident images (pfx_api.cpp); mask_host may be NULL (= no selection)
extern "C" int pfx_int_gauss_exact_weights(pfx_ctx* ctx, float sigma, const float** wts);
extern "C" int pfx_int_gauss_exact_combine_applies(pfx_ctx* ctx, const void* src_dev, const void* dst_dev, uint32_t w, uint32_t h, float sigma);
extern "C" int pfx_int_gauss_exact_combine(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma, int epilogue, float p0, const void* mask_dev); // 1 ran, 0 not applicable
int pfx_int_blur_with_selection_dev(pfx_ctx* ctx, const void* d_src, void* d_dst, uint32_t w, uint32_t h, float sigma,
                                    const uint8_t* mask_host, const void* d_mask);

// script front-end on the device image held in ctx->st_in (pfx_script_host.cpp).  On success the result is in ctx->st_in and
// *w / *h hold the final size; console / ops may be NULL.
int pfx_int_script_run_dev(pfx_ctx* ctx, const char* source, uint32_t* w, uint32_t* h, const uint8_t* mask, pfx_script_result* result,
                           std::vector<std::string>* console, std::vector<pfx_canvas_op>* ops);

extern "C" int pfx_int_script_check_limited(const char* source, uint32_t w, uint32_t h, pfx_script_result* result, uint64_t max_ops);

// run_one's script step on a document (pfx_project.cpp): pfx_project_run_script plus the console lines for --verbose
int pfx_int_project_run_script(pfx_ctx* ctx, pfx_project* p, const char* source, pfx_script_result* result, std::vector<std::string>* console);

// host-side restatements that the reference also runs on the host (pfx_host_math.cpp)
int  pfx_host_gaussian_radius(float sigma);                            // ceil(3 sigma) as the reference casts it
int  pfx_host_gaussian_kernel(float sigma, std::vector<float>& out);  // ref: src/ops/filters.rs:214-234
// w * 2^s = w1 + w2 as two f16 arrays (pfxk_gauss_mfma layout); returns 2^-2s, *bias = 1024 * sum(w1 + w2)
float pfx_host_gaussian_split_f16(const std::vector<float>& k, int wlen, int woff, std::vector<uint16_t>& out, float* bias, float* bias_single);
float pfx_host_bc_factor(float contrast);                             // ref: src/ops/adjustments.rs:273
float pfx_host_exposure_gain(float ev);                               // ref: src/ops/adjustments.rs:353
