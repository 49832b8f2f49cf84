// pfx_api.cpp — the C ABI (include/pfx.h): argument checking, host<->device staging, kernel sequencing.
// Host-buffer entry points = upload -> the `_dev` entry point -> download -> stream sync; `dst` is written only
// after every step succeeded (the reference's "never poison the document" rule).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "k_brush_math.h"
#include "pfx_internal.h"

void pfx_host_brush_lut(float size, float hardness, bool anti_aliased, uint8_t lut[256]);
void pfx_host_line_points(float x0, float y0, float x1, float y1, uint32_t w, uint32_t h, std::vector<float>& out);
void pfx_host_rhai_levels_lut(float in_black, float in_white, float gamma, uint8_t lut[256]);

static_assert((int)PFX_OP_COUNT == (int)PFXK_OP_COUNT && (int)PFX_OP_DESATURATE == (int)PFXK_OP_DESATURATE, "op ids out of sync");
static_assert((int)PFX_RHAI_COUNT == (int)PFXK_RHAI_COUNT && (int)PFX_RHAI_LEVELS == (int)PFXK_RHAI_LEVELS, "rhai ids out of sync");
static_assert((int)PFX_ADJ_CHANNEL_MIXER == (int)PFXK_ADJ_CHANNEL_MIXER, "layer kinds out of sync");

namespace {

inline size_t img_bytes(uint32_t w, uint32_t h) { return (size_t)w * h * 4; }

// FAST instantiation of the compositor (k_flatten.hip): requires opacity.clamp(0,1) in [2^-40, 1] for every layer, so
// that (a) every division operand of blend_pixel_static stays in the normal range where v_div_scale / v_div_fixup
// are identities and (b) top_a > 0 for every non-skipped pixel (out_a > 0: no zero check, no clamp).  Anything else
// (zero, negative, tiny or NaN opacity) runs the plain instantiation with IEEE '/', clamps and zero checks.
inline bool opacity_allows_fast_div(float opacity) { return opacity >= 9.094947017729282e-13f; /* 2^-40 */ }

int check_img(pfx_ctx* ctx, const void* src, const void* dst, uint32_t w, uint32_t h, const char* who)
{
    if (!ctx) return PFX_ERR_INVALID;
    if (!src || !dst) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: null image pointer", who);
    if (w == 0 || h == 0) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: zero-sized image", who);
    if ((uint64_t)w * h > 256000000ull) // TiledImage::new clamps beyond 256 M px (ref: tiled_image.rs:15-26)
        return pfx_fail(ctx, PFX_ERR_INVALID, "%s: %ux%u exceeds the 256 Mpx document limit", who, w, h);
    return pfx_use(ctx);
}

// Aliasing rule of the `_dev` entry points (include/pfx.h): src == dst is allowed where the header says so; buffers that overlap in
// any other way, or at all for a neighbourhood operation, are refused — a kernel that reads a halo while other workgroups write the
// same memory would race silently.
bool ranges_overlap(const void* a, const void* b, size_t bytes)
{
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + bytes && y < x + bytes;
}
bool ranges_overlap2(const void* a, size_t a_bytes, const void* b, size_t b_bytes)
{
    const uintptr_t x = (uintptr_t)a, y = (uintptr_t)b;
    return x < y + b_bytes && y < x + a_bytes;
}
int check_disjoint(pfx_ctx* ctx, const void* src, const void* dst, uint32_t w, uint32_t h, const char* who, bool same_ok = false)
{
    if (src == dst) return same_ok ? PFX_OK : pfx_fail(ctx, PFX_ERR_INVALID, "%s: src and dst must be different buffers", who);
    if (ranges_overlap(src, dst, (size_t)w * h * 4))
        return pfx_fail(ctx, PFX_ERR_INVALID, "%s: src and dst overlap", who);
    return PFX_OK;
}
// the same for a source whose extent differs from the destination's (the warps sample an sw x sh image into a w x h one)
int check_disjoint2(pfx_ctx* ctx, const void* src, uint32_t sw, uint32_t sh, const void* dst, uint32_t w, uint32_t h, const char* who)
{
    if (ranges_overlap2(src, (size_t)sw * sh * 4, dst, (size_t)w * h * 4)) return pfx_fail(ctx, PFX_ERR_INVALID, "%s: src and dst overlap", who);
    return PFX_OK;
}

// stage host src (+ optional mask) into the context's device buffers
int stage_in(pfx_ctx* ctx, const uint8_t* src, const uint8_t* mask, uint32_t w, uint32_t h, const void** d_mask)
{
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, img_bytes(w, h)));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, img_bytes(w, h)));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, img_bytes(w, h)));
    *d_mask = nullptr;
    if (mask) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_mask, (size_t)w * h));
        PFX_TRY(pfx_h2d(ctx, ctx->st_mask.p, mask, (size_t)w * h));
        *d_mask = ctx->st_mask.p;
    }
    return PFX_OK;
}

int finish_out(pfx_ctx* ctx, uint8_t* dst, uint32_t w, uint32_t h)
{
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, img_bytes(w, h)));
    return pfx_sync(ctx);
}

// host-side parameter preparation for the pointwise bank: everything that is uniform per call is folded here with
// the same f32 expressions the reference evaluates per pixel (so the kernels see identical constants)
int prepare_adjust(pfx_ctx* ctx, int op, const float* p, uint32_t n, pfxk_params& P, bool& needs_lut)
{
    std::memset(&P, 0, sizeof P);
    needs_lut = false;
    auto need = [&](uint32_t k) { return n >= k && p != nullptr; };
    switch (op) {
    case PFX_OP_INVERT: case PFX_OP_INVERT_ALPHA: case PFX_OP_SEPIA: case PFX_OP_DESATURATE: return PFX_OK;
    case PFX_OP_BRIGHTNESS_CONTRAST:
        if (!need(2)) break;
        P.p[0] = p[0]; P.p[1] = pfx_host_bc_factor(p[1]); return PFX_OK;
    case PFX_OP_HSL:
        if (!need(3)) break;
        P.p[0] = p[0] / 360.0f;            // hue_shift / 360.0   (adjustments.rs:310)
        P.p[1] = 1.0f + p[1] / 100.0f;     // sat_factor          (:307)
        P.p[2] = p[2] * 255.0f / 100.0f;   // light_offset        (:308)
        // p[11] != 0: every parameter is finite, so the kernel's results are finite and its cheaper round-and-pack applies (k_pointwise.hip)
        P.p[11] = (std::isfinite(P.p[0]) && std::isfinite(P.p[1]) && std::isfinite(P.p[2])) ? 1.0f : 0.0f;
        return PFX_OK;
    case PFX_OP_EXPOSURE:
        if (!need(1)) break;
        P.p[0] = pfx_host_exposure_gain(p[0]); return PFX_OK;
    case PFX_OP_HIGHLIGHTS_SHADOWS:
        if (!need(2)) break;
        P.p[0] = p[0] / 100.0f; P.p[1] = p[1] / 100.0f; return PFX_OK;
    case PFX_OP_TEMPERATURE_TINT:
        if (!need(2)) break;
        P.p[0] = p[0] * 1.5f; P.p[1] = p[1] * 1.0f; return PFX_OK;
    case PFX_OP_THRESHOLD:
        if (!need(1)) break;
        P.p[0] = p[0]; return PFX_OK;
    case PFX_OP_POSTERIZE:
        if (!need(1)) break;
        P.p[0] = std::max(p[0], 2.0f); return PFX_OK; // levels.max(2) as f32
    case PFX_OP_COLOR_BALANCE:
        if (!need(9)) break;
        for (int i = 0; i < 9; ++i) P.p[i] = p[i];
        return PFX_OK;
    case PFX_OP_BLACK_AND_WHITE:
        if (!need(3)) break;
        P.p[0] = p[0]; P.p[1] = p[1]; P.p[2] = p[2]; return PFX_OK;
    case PFX_OP_VIBRANCE:
        if (!need(1)) break;
        P.p[0] = p[0] / 100.0f;
        P.p[11] = std::isfinite(P.p[0]) ? 1.0f : 0.0f; // as for HSL
        return PFX_OK;
    case PFX_OP_GRADIENT_MAP: case PFX_OP_LUT_RGBA: needs_lut = true; return PFX_OK;
    default: return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_adjust: unknown op %d", op);
    }
    return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_adjust: op %d needs more parameters (got %u)", op, n);
}

int upload_lut(pfx_ctx* ctx, const uint8_t* lut_host, size_t bytes)
{
    uint8_t padded[1024] = {0};
    std::memcpy(padded, lut_host, std::min<size_t>(bytes, 1024));
    PFX_TRY(pfx_reserve(ctx, ctx->d_lut, 1024));
    return pfx_h2d(ctx, ctx->d_lut.p, padded, 1024);
}

// layer stack -> device descriptors
// preview layer of a composite call (device pointers; see pfx_preview in include/pfx.h)
struct preview_arg {
    const void* d_pixels = nullptr;
    const void* d_chunk_present = nullptr; // NULL: derive with the from_rgba_image rule
    pfx_preview info{};
};

int build_stack(pfx_ctx* ctx, const void* const* layer_ptrs, const void* const* mask_ptrs, const pfx_layer_info* layers,
                uint32_t n, uint32_t w, uint32_t h, bool from_store, uint32_t* n_desc, bool* general, bool* has_adj,
                uint32_t track_info = 0xFFFFFFFFu, uint32_t* track_pos = nullptr, const uint8_t** track_pixels = nullptr,
                pfxk_dle_cands* cands = nullptr, std::vector<uint8_t>* chunk_meta = nullptr, bool* shallow = nullptr)
{
    std::vector<const uint8_t*> flag_ptrs; // per descriptor: the stored layer's chunk alpha summary (NULL: none)
    std::vector<pfxk_layer_desc> desc;
    std::vector<float> adj;
    desc.reserve(n);
    *general = false;
    *has_adj = false;
    if (track_pos) *track_pos = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < n; ++i) {
        const pfx_layer_info& L = layers[i];
        if (!L.visible) continue; // layer_effectively_visible == false: skipped entirely (canvas_state.rs:576)
        pfxk_layer_desc d{};
        d.opacity = L.opacity;
        d.mode = L.blend_mode > 24 ? 0u : (uint32_t)L.blend_mode; // BlendMode::from_u8 fallback
        d.kind = L.kind;
        if (L.kind != PFX_LAYER_RASTER) {
            if (L.kind > PFX_ADJ_CHANNEL_MIXER) return pfx_fail(ctx, PFX_ERR_INVALID, "layer %u: unknown kind %u", i, L.kind);
            d.adj_off = (uint32_t)adj.size();
            float a[16];
            std::memcpy(a, L.adj, sizeof a);
            if (L.kind == PFX_ADJ_EXPOSURE) a[0] = pfx_host_exposure_gain(L.adj[0]);
            if (L.kind == PFX_ADJ_BRIGHTNESS_CONTRAST) a[1] = pfx_host_bc_factor(L.adj[1]);
            adj.insert(adj.end(), a, a + 16);
            *general = true;
            *has_adj = true;
        } else if (from_store) {
            auto it = ctx->layers.find(L.layer_idx);
            if (it == ctx->layers.end()) continue; // not uploaded: skipped like the reference (renderer.rs:556)
            if (it->second.w != w || it->second.h != h)
                return pfx_fail(ctx, PFX_ERR_INVALID, "layer %u is %ux%u, canvas is %ux%u", L.layer_idx, it->second.w, it->second.h, w, h);
            d.pixels = (const uint8_t*)it->second.pixels.p;
            if (it->second.has_mask) { d.mask = (const uint8_t*)it->second.mask.p; *general = true; }
            flag_ptrs.resize(desc.size() + 1, nullptr);
            flag_ptrs[desc.size()] = it->second.has_mask ? nullptr : (const uint8_t*)it->second.chunk_flags.p;
        } else {
            d.pixels = (const uint8_t*)layer_ptrs[i];
            if (!d.pixels) return pfx_fail(ctx, PFX_ERR_INVALID, "layer %u: null device pointer", i);
            if (mask_ptrs && mask_ptrs[i]) { d.mask = (const uint8_t*)mask_ptrs[i]; *general = true; }
        }
        if (d.kind == PFX_LAYER_RASTER) { // Rust f32::clamp(0.0, 1.0); NaN stays NaN (such a stack never takes the streaming kernels)
            const float c = d.opacity < 0.0f ? 0.0f : (d.opacity > 1.0f ? 1.0f : d.opacity);
            std::memcpy(&d.adj_off, &c, 4);
        }
        if (i == track_info && track_pos && d.kind == PFX_LAYER_RASTER) { *track_pos = (uint32_t)desc.size(); *track_pixels = d.pixels; }
        desc.push_back(d);
    }
    *n_desc = (uint32_t)desc.size();
    if (chunk_meta) {
        // [n_desc summary pointers | n_desc wanted bits]: Normal at opacity >= 1 resets a chunk whose alpha is all 255 (canvas_state.rs:1258),
        // Overwrite one without a zero alpha (:1275); layer 0 has nothing below it
        chunk_meta->clear();
        flag_ptrs.resize(desc.size(), nullptr);
        std::vector<uint8_t> want(desc.size(), 0);
        bool any = false;
        for (size_t p = 1; p < desc.size(); ++p) {
            if (desc[p].kind != PFX_LAYER_RASTER || desc[p].mask || !flag_ptrs[p]) continue;
            if (desc[p].mode == 14u) want[p] = 2;
            else if (desc[p].mode == 0u && desc[p].opacity >= 1.0f) want[p] = 1;
            any = any || want[p] != 0;
        }
        if (any) {
            chunk_meta->resize(desc.size() * sizeof(void*) + desc.size());
            std::memcpy(chunk_meta->data(), flag_ptrs.data(), desc.size() * sizeof(void*));
            std::memcpy(chunk_meta->data() + desc.size() * sizeof(void*), want.data(), desc.size());
        }
    }
    {   // arithmetic weight of the stack's blend modes (k_blend.h instruction counts, profiles/r03_blend_isa.md): the streaming compositor's register shape follows it
        static const uint8_t cls_of_mode[25] = {2, 2, 1, 2, 0, 0, 0, 0, 1, 2, 1, 2, 2, 0, 2, 1, 0, 1, 2, 0, 2, 0, 1, 0, 1};
        int cls = 2;
        for (const pfxk_layer_desc& d : desc) cls = std::min(cls, d.kind == PFX_LAYER_RASTER && d.mode < 25u ? (int)cls_of_mode[d.mode] : 0);
        ctx->stack_mode_class = desc.empty() ? 0 : cls;
    }
    if (cands) {
        // reset layers (k_flatten.hip: dead-layer elimination): the topmost PFXK_DLE_MAX raster layers above the bottom one that are
        // Overwrite (canvas_state.rs:1275) or Normal at opacity >= 1 (:1258); only used by the raster-only streaming path
        std::memset(cands, 0, sizeof *cands);
        std::vector<std::pair<uint32_t, uint32_t>> found;
        for (uint32_t p = 1; p < desc.size(); ++p) {
            const pfxk_layer_desc& d = desc[p];
            if (d.kind != PFX_LAYER_RASTER || d.mask) continue;
            if (d.mode == 14u) found.emplace_back(p, 0u);
            else if (d.mode == 0u && d.opacity >= 1.0f) found.emplace_back(p, 1u);
        }
        // Every candidate the kernel inspects costs a layer's worth of reads for the units it examines, and the elimination kernel's load
        // stream is less efficient than the plain streaming kernel's (4.7 against 5+ TB/s with the arithmetic taken out): it pays where the
        // blend arithmetic dominates — deep stacks — and loses up to 30 % on shallow, memory-bound ones (tools/ab_docs_dle.py: nine Normal
        // layers with S2's alpha 0.43 against 0.33 ms).  Stacks below 16 layers keep the plain kernel (pfx_tune "dle_min_layers").
        // ... unless the caller can decide per stack (flatten_common: a probe of the candidate's alpha, `shallow` tells it the threshold applied)
        if (desc.size() < (size_t)ctx->dle_min_layers) { if (shallow && ctx->dle_adaptive) *shallow = true; else found.clear(); }
        const size_t kmax = std::min<size_t>(PFXK_DLE_MAX, std::max<size_t>(1, desc.size() / 6));
        const size_t first = found.size() > kmax ? found.size() - kmax : 0;
        for (size_t k = first; k < found.size(); ++k) {
            cands->layer[cands->n] = found[k].first;
            cands->kind[cands->n] = found[k].second;
            cands->n++;
        }
    }
    // PFXK_DESC_PAD copies of the last descriptor behind the stack: the class-sorting compositor requests descriptors ahead of the layer it blends without
    // clamping the index (k_flatten.hip: srt_layers: every fetch() loads the next descriptor, and the loop runs its fetches in pairs, so a pass over an
    // odd number of layers that ends at the top of the stack has read descriptor n + 2; srt_early<NB> runs NB - 1 layers ahead in groups of NB; what is
    // fetched through the copies is never blended)
    if (!desc.empty()) { const pfxk_layer_desc last = desc.back(); desc.insert(desc.end(), PFXK_DESC_PAD, last); }
    // the tables only travel when they differ from what the device already holds (a render loop re-composites the same stack: the
    // small pageable-memory copy in front of every launch was a ~10 us bubble on the stream)
    const size_t desc_bytes = desc.size() * sizeof(pfxk_layer_desc), adj_bytes = adj.size() * sizeof(float);
    const bool same_desc = ctx->d_desc.p && ctx->desc_cache.size() == desc_bytes && (desc_bytes == 0 || std::memcmp(ctx->desc_cache.data(), desc.data(), desc_bytes) == 0);
    if (!same_desc) {
        ctx->desc_cache.clear();
        PFX_TRY(pfx_reserve(ctx, ctx->d_desc, std::max<size_t>(desc.size(), 1) * sizeof(pfxk_layer_desc)));
        PFX_TRY(pfx_h2d(ctx, ctx->d_desc.p, desc.data(), desc_bytes));
        ctx->desc_cache.assign((const uint8_t*)desc.data(), (const uint8_t*)desc.data() + desc_bytes);
    }
    if (!adj.empty()) {
        const bool same_adj = ctx->d_adj.p && ctx->adj_cache.size() == adj_bytes && std::memcmp(ctx->adj_cache.data(), adj.data(), adj_bytes) == 0;
        if (!same_adj) {
            ctx->adj_cache.clear();
            PFX_TRY(pfx_reserve(ctx, ctx->d_adj, adj_bytes));
            PFX_TRY(pfx_h2d(ctx, ctx->d_adj.p, adj.data(), adj_bytes));
            ctx->adj_cache.assign((const uint8_t*)adj.data(), (const uint8_t*)adj.data() + adj_bytes);
        }
    }
    return PFX_OK;
}

int flatten_common(pfx_ctx* ctx, const void* const* layer_ptrs, const void* const* mask_ptrs, const pfx_layer_info* layers,
                   uint32_t n_layers, uint32_t w, uint32_t h, bool from_store, void* dst_dev, const preview_arg* pv = nullptr,
                   const pfxk_region* region = nullptr, const uint8_t* chunk_keys_host = nullptr)
{
    PFX_REQUIRE(ctx, n_layers <= PFX_MAX_LAYERS, "too many layers");
    PFX_REQUIRE(ctx, n_layers == 0 || layers, "null layer list");
    // in place is allowed — dst may BE one of the layers (every kernel reads a pixel of every layer before it writes that pixel) — a partial overlap is not:
    // a workgroup would read pixels another one has already replaced
    if (!from_store && layer_ptrs && dst_dev)
        for (uint32_t l = 0; l < n_layers; ++l)
            if (layer_ptrs[l] && layer_ptrs[l] != dst_dev && ranges_overlap(layer_ptrs[l], dst_dev, img_bytes(w, h)))
                return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_flatten_dev: dst partially overlaps layer %u (it may be a layer, or disjoint from all)", l);
    uint32_t n_desc = 0, active_pos = 0xFFFFFFFFu;
    const uint8_t* active_pixels = nullptr;
    bool general = false, has_adj = false;
    pfxk_dle_cands cands{};
    std::vector<uint8_t> chunk_meta;
    bool shallow = false;
    PFX_TRY(build_stack(ctx, layer_ptrs, mask_ptrs, layers, n_layers, w, h, from_store, &n_desc, &general, &has_adj,
                        pv ? pv->info.active_layer : 0xFFFFFFFFu, &active_pos, &active_pixels, &cands, from_store ? &chunk_meta : nullptr,
                        from_store ? nullptr : &shallow));
    bool probe_now = false;
    uint32_t probe_layer = 0, probe_kind = 0;
    if (shallow && cands.n > 0) {
        // Below the depth threshold the elimination kernel runs only where a probe of THIS stack found its topmost reset layer coherent (an opaque photo layer,
        // a filled background above old layers: most units start there and read nothing below; per-pixel-random alpha splits every unit and loses 30 %).  The
        // probe runs once per stack, behind the composite that found none, and reports through pinned memory: later composites of the same stack use it.
        const bool usable = !general && !region && !(pv && pv->d_pixels) && (uint64_t)w * h < (1ull << 30);
        std::vector<uint8_t> key(ctx->desc_cache);
        key.insert(key.end(), (const uint8_t*)&w, (const uint8_t*)&w + 4); key.insert(key.end(), (const uint8_t*)&h, (const uint8_t*)&h + 4);
        const bool same = ctx->dle_probe_state != 0 && key == ctx->dle_probe_key;
        if (same && ctx->dle_probe_state == 1 && hipEventQuery(ctx->ev_dle_probe) == hipSuccess)
            ctx->dle_probe_state = *ctx->h_dle_verdict == (ctx->dle_probe_tag | 0x80000000u) ? 2 : 3;
        if (usable && !same) { probe_now = true; ctx->dle_probe_key.swap(key); ctx->dle_probe_state = 0; }
        probe_layer = cands.layer[cands.n - 1u]; probe_kind = cands.kind[cands.n - 1u];
        if (!(usable && same && ctx->dle_probe_state == 2)) cands.n = 0;   // the plain streaming kernel
    }
    const size_t nchunks = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    uint8_t* d_chunks = nullptr;
    bool chunks_ready = false;
    if (has_adj) { // adjustment layers only run on chunks populated in some visible layer (canvas_state.rs:529-550)
        PFX_TRY(pfx_reserve(ctx, ctx->d_chunks, nchunks));
        d_chunks = (uint8_t*)ctx->d_chunks.p;
        // a caller that still has the layers' TiledImage chunk keys (a loaded document) passes their union: a chunk that is present
        // but fully transparent counts, exactly as in the reference; flat layers can only offer "some alpha != 0" (the kernel's pre-pass)
        if (chunk_keys_host && !(pv && pv->d_pixels)) { PFX_TRY(pfx_h2d(ctx, d_chunks, chunk_keys_host, nchunks)); chunks_ready = true; }
    }
    pfxk_preview PV{};
    if (pv && pv->d_pixels) { // chunk flags: [preview present | active layer present] (canvas_state.rs:541-548,587,593-597)
        PFX_TRY(pfx_reserve(ctx, ctx->fx_a, 2 * nchunks));
        uint8_t* flags = (uint8_t*)ctx->fx_a.p;
        if (pv->d_chunk_present) PFX_HIP(ctx, hipMemcpyAsync(flags, pv->d_chunk_present, nchunks, hipMemcpyDeviceToDevice, ctx->stream));
        else PFX_HIP(ctx, pfxk_chunk_populated(ctx->stream, (const uint8_t*)pv->d_pixels, w, h, flags));
        if (active_pixels) PFX_HIP(ctx, pfxk_chunk_populated(ctx->stream, active_pixels, w, h, flags + nchunks));
        PV.pixels = (const uint8_t*)pv->d_pixels;
        PV.chunk_present = flags;
        PV.layer_chunk_present = flags + nchunks;
        PV.active_pos = active_pos; // 0xFFFFFFFF when the active layer is hidden: only its chunk keys count
        PV.mode = pv->info.blend_mode > 24 ? 0u : pv->info.blend_mode;
        PV.is_eraser = pv->info.is_eraser != 0;
        PV.replaces = pv->info.replaces_layer != 0;
    }
    bool fast_div = true; // k_flatten.hip:rdiv is bit-identical to '/' unless an opacity is a positive value < 2^-40
    for (uint32_t i = 0; i < n_layers; ++i) fast_div = fast_div && opacity_allows_fast_div(layers[i].opacity);
    // stored layers: the per-chunk start table (which layer is the first that can show in a 64 x 64 chunk) from their alpha summaries —
    // only the plain streaming kernel takes it (whole-frame, raster-only stacks)
    const uint8_t* d_chunk_start = nullptr;
    if (ctx->use_chunk_start && !chunk_meta.empty() && !general && fast_div && !region && !PV.pixels && cands.n == 0 && n_desc < 255) {
        if (!ctx->h_chunk_useful) {
            PFX_HIP(ctx, hipHostMalloc((void**)&ctx->h_chunk_useful, sizeof(uint32_t), hipHostMallocDefault));
            *ctx->h_chunk_useful = 0;
            PFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_chunk_useful, hipEventDisableTiming));
        }
        const bool same = ctx->chunk_state != 0 && ctx->chunk_epoch == ctx->store_epoch && ctx->chunk_meta_cache == chunk_meta &&
                          ctx->d_chunk_start.cap >= nchunks;
        if (same && ctx->chunk_state == 1 && hipEventQuery(ctx->ev_chunk_useful) == hipSuccess)
            ctx->chunk_state = (*ctx->h_chunk_useful == ctx->chunk_tag) ? 2 : 3; // the table kernel of an earlier call has finished: its verdict is in
        if (!same) {
            ctx->chunk_meta_cache.clear();
            PFX_TRY(pfx_reserve(ctx, ctx->d_chunk_meta, chunk_meta.size()));
            PFX_TRY(pfx_h2d(ctx, ctx->d_chunk_meta.p, chunk_meta.data(), chunk_meta.size()));
            ctx->chunk_meta_cache = chunk_meta;
            PFX_TRY(pfx_reserve(ctx, ctx->d_chunk_start, nchunks));
            ctx->chunk_tag += 1u;
            PFX_HIP(ctx, pfxk_chunk_start(ctx->stream, (const uint8_t* const*)ctx->d_chunk_meta.p, (const uint8_t*)ctx->d_chunk_meta.p + (size_t)n_desc * sizeof(void*),
                                          n_desc, (uint32_t)nchunks, (uint8_t*)ctx->d_chunk_start.p, ctx->h_chunk_useful, ctx->chunk_tag));
            PFX_HIP(ctx, hipEventRecord(ctx->ev_chunk_useful, ctx->stream));
            ctx->chunk_state = 1;
            ctx->chunk_epoch = ctx->store_epoch;
        }
        if (ctx->chunk_state != 3) d_chunk_start = (const uint8_t*)ctx->d_chunk_start.p; // pending or useful: the table of THIS stack is in the buffer
    }
    // The class-sorting compositor (k_flatten.hip: flatten_srt_kernel) reads every layer through typed UNORM8 buffer loads and keeps its accumulators as
    // RN(k / 255); its result leaves as v_cvt_pk_u8_f32 + dword stores (the typed format store measured 1 % slower).  The device's UNORM8 <-> float
    // conversions must be exact for all 256 byte values, which is checked once per context on the device itself.  Otherwise round 3's kernel runs.
    int parking_ok = 0;
    if (cands.n > 0 && !general && fast_div && !region && !PV.pixels) {
        if (ctx->unorm_store_ok < 0) {
            PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 2048));
            PFX_HIP(ctx, hipMemsetAsync(ctx->d_misc.p, 0, 2048, ctx->stream));
            PFX_HIP(ctx, pfxk_unorm_store_check(ctx->stream, (uint8_t*)ctx->d_misc.p, (unsigned long long*)((uint8_t*)ctx->d_misc.p + 1024)));
            unsigned long long bad = 1;
            PFX_TRY(pfx_d2h(ctx, &bad, (uint8_t*)ctx->d_misc.p + 1024, sizeof bad));
            PFX_TRY(pfx_sync(ctx)); // once per context
            ctx->unorm_store_ok = bad == 0 ? 1 : 0;
        }
        parking_ok = ctx->unorm_store_ok == 1;
    }
    {
        pfx_timer t(ctx, "flatten");
        PFX_HIP(ctx, pfxk_flatten(ctx->stream, (const pfxk_layer_desc*)ctx->d_desc.p, n_desc, (const float*)ctx->d_adj.p,
                                  general ? 1 : 0, fast_div ? 1 : 0, d_chunks, chunks_ready ? 1 : 0, w, h, (uint8_t*)dst_dev, PV.pixels ? &PV : nullptr, region, &cands, d_chunk_start,
                                  parking_ok, ctx->stack_mode_class));
    }
    if (probe_now) {   // behind the composite (it may have been in place: the probe reads layers, and a layer that was the destination now holds the result — a hint all the same)
        if (!ctx->h_dle_verdict) {
            PFX_HIP(ctx, hipHostMalloc((void**)&ctx->h_dle_verdict, sizeof(uint32_t), hipHostMallocDefault));
            *ctx->h_dle_verdict = 0;
        }
        if (!ctx->ev_dle_probe) PFX_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_dle_probe, hipEventDisableTiming));
        ctx->dle_probe_tag = (ctx->dle_probe_tag + 1u) & 0x7fffffffu;
        PFX_HIP(ctx, pfxk_dle_probe(ctx->stream, (const pfxk_layer_desc*)ctx->d_desc.p, probe_layer, probe_kind, (uint32_t)((size_t)w * h), ctx->h_dle_verdict, ctx->dle_probe_tag));
        PFX_HIP(ctx, hipEventRecord(ctx->ev_dle_probe, ctx->stream));
        ctx->dle_probe_state = 1;
    }
    return PFX_OK;
}

int brush_prepare(pfx_ctx* ctx, const pfx_brush* b, pfxk_brush& B, bool& skip)
{
    PFX_REQUIRE(ctx, b != nullptr, "null brush");
    std::memset(&B, 0, sizeof B);
    skip = false;
    B.radius = b->size / 2.0f;                 // pressure_size() / 2 (brush_render.rs:196)
    B.radius_sq = B.radius * B.radius;
    if (B.radius_sq < 0.001f) { skip = true; return PFX_OK; }
    B.draw_radius = b->anti_aliased ? B.radius + 0.5f : B.radius;
    B.draw_radius_sq = B.draw_radius * B.draw_radius;
    B.use_direct_alpha = B.draw_radius > B.radius;
    B.inv_radius_sq = 1.0f / B.radius_sq;
    B.hardness = b->hardness;
    B.flow = b->flow;
    B.src_r = b->color[0]; B.src_g = b->color[1]; B.src_b = b->color[2]; B.src_a = b->color[3];
    auto u8 = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 255.0f ? 255u : (uint32_t)(int)v); };
    B.rgb8 = u8(B.src_r * 255.0f) | (u8(B.src_g * 255.0f) << 8) | (u8(B.src_b * 255.0f) << 16); // :223-225 truncating
    B.anti_aliased = b->anti_aliased != 0;
    B.is_eraser = b->is_eraser != 0;
    B.mode = b->mode;
    PFX_REQUIRE(ctx, b->mode >= 0 && b->mode <= 3, "unknown brush mode");
    return PFX_OK;
}

} // namespace

extern "C" {

// ======================================================================= device tier
int pfx_flatten_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const void* const* mask_ptrs_dev,
                    const pfx_layer_info* layers, uint32_t n_layers, uint32_t w, uint32_t h, void* dst_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, dst_dev && pfx_dims_ok(w, h) && (n_layers == 0 || layer_ptrs_dev), "pfx_flatten_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    return flatten_common(ctx, layer_ptrs_dev, mask_ptrs_dev, layers, n_layers, w, h, false, dst_dev);
}

// pfx_flatten_dev for a document whose layers are TiledImages: chunk_keys_host[c] != 0 where some visible raster layer HOLDS chunk c
// (canvas_state.rs:528-550 `chunk_keys()`), whatever its pixels are
int pfx_int_flatten_with_chunk_keys_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const pfx_layer_info* layers, uint32_t n_layers,
                                        uint32_t w, uint32_t h, void* dst_dev, const uint8_t* chunk_keys_host)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, dst_dev && w && h && (n_layers == 0 || layer_ptrs_dev), "pfx_int_flatten_with_chunk_keys_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    return flatten_common(ctx, layer_ptrs_dev, nullptr, layers, n_layers, w, h, false, dst_dev, nullptr, nullptr, chunk_keys_host);
}

// sharpen / glow with the Gaussian and the combine in ONE kernel (k_gauss_exact.hip, epilogue 1 / 2) where the bit-exact fused Gaussian applies: radius 1 .. 16,
// distinct buffers, exact mode.  Returns 1 when it ran, 0 when the caller has to take the two-step path, < 0 on error.
// the f32 taps of sigma on the device (cached per context): *wts points at tap 0, pfxk_gauss_weight_pad() zero taps on both sides
int pfx_int_gauss_exact_weights(pfx_ctx* ctx, float sigma, const float** wts)
{
    uint32_t sigma_bits; std::memcpy(&sigma_bits, &sigma, 4);
    const int pad = pfxk_gauss_weight_pad();
    if (!ctx->wts_valid || ctx->wts_sigma_bits != sigma_bits) {
        std::vector<float> k;
        pfx_host_gaussian_kernel(sigma, k);
        std::vector<float> padded(k.size() + 2 * (size_t)pad, 0.0f);
        std::copy(k.begin(), k.end(), padded.begin() + pad);
        PFX_TRY(pfx_reserve(ctx, ctx->d_wts, padded.size() * sizeof(float)));
        PFX_TRY(pfx_h2d(ctx, ctx->d_wts.p, padded.data(), padded.size() * sizeof(float)));
        ctx->wts_sigma_bits = sigma_bits;
        ctx->wts_valid = true;
    }
    *wts = (const float*)ctx->d_wts.p + pad;
    return PFX_OK;
}
int pfx_int_gauss_exact_combine_applies(pfx_ctx* ctx, const void* src_dev, const void* dst_dev, uint32_t w, uint32_t h, float sigma)
{
    const int radius = pfx_host_gaussian_radius(sigma);
    return ctx->exact && radius >= 1 && radius <= pfxk_gauss_fused_exact_max_radius() && src_dev != dst_dev && !ranges_overlap(src_dev, dst_dev, img_bytes(w, h));
}
int pfx_int_gauss_exact_combine(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma, int epilogue, float p0, const void* mask_dev)
{
    const int radius = pfx_host_gaussian_radius(sigma);
    if (!pfx_int_gauss_exact_combine_applies(ctx, src_dev, dst_dev, w, h, sigma)) return 0;
    const float* wts = nullptr;
    PFX_TRY(pfx_int_gauss_exact_weights(ctx, sigma, &wts));
    PFX_HIP(ctx, pfxk_gauss_fused_exact(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, wts, radius, w, h, epilogue, p0, (const uint8_t*)mask_dev));
    return 1;
}

// the matrix-core Gaussian's f16 tap tables for sigma on the device (cached per context)
static int gauss_mfma_tables(pfx_ctx* ctx, float sigma)
{
    uint32_t sigma_bits; std::memcpy(&sigma_bits, &sigma, 4);
    if (!ctx->wsplit_valid || ctx->wsplit_sigma_bits != sigma_bits) {
        std::vector<float> k;
        pfx_host_gaussian_kernel(sigma, k);
        std::vector<uint16_t> ws;
        ctx->wsplit_inv_scale = pfx_host_gaussian_split_f16(k, pfxk_gauss_mfma_wlen(), pfxk_gauss_mfma_woff(), ws, &ctx->wsplit_bias, &ctx->wsplit_bias_single);
        PFX_TRY(pfx_reserve(ctx, ctx->d_wsplit, ws.size() * sizeof(uint16_t)));
        PFX_TRY(pfx_h2d(ctx, ctx->d_wsplit.p, ws.data(), ws.size() * sizeof(uint16_t)));
        ctx->wsplit_sigma_bits = sigma_bits;
        ctx->wsplit_valid = true;
    }
    return PFX_OK;
}

int pfx_gaussian_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma, void* tmp_dev)
{
    return pfx_gaussian_blur_band_dev(ctx, src_dev, dst_dev, w, h, sigma, tmp_dev, 0);
}

int pfx_gaussian_blur_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float sigma, void* tmp_dev,
                               uint32_t first_row)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_gaussian_blur_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_gaussian_blur_dev", true)); // in place: the two-pass kernels (through tmp)
    PFX_REQUIRE(ctx, (uint64_t)first_row + h <= 0x7fffffffull, "pfx_gaussian_blur_band_dev: band outside any image");   // the kernels carry image rows in 32-bit signed integers
    // radius first: a huge sigma must be refused before a tap array of that size is built (the C ABI must not throw)
    const int radius = pfx_host_gaussian_radius(sigma);
    if (radius > pfxk_gauss_max_radius())
        return pfx_fail(ctx, PFX_ERR_UNSUPPORTED, "gaussian radius %d beyond the device tile limit %d", radius, pfxk_gauss_max_radius());
    uint32_t sigma_bits; std::memcpy(&sigma_bits, &sigma, 4);
    if (!ctx->exact && radius >= 1 && radius <= pfxk_gauss_mfma_max_radius() && src_dev != dst_dev) {
        // default mode: fused H+V on the matrix cores, no intermediate in HBM, no scratch (k_gauss.hip:gauss_strip_kernel)
        PFX_TRY(gauss_mfma_tables(ctx, sigma));
        pfx_timer t(ctx, "gauss_mfma");
        PFX_HIP(ctx, pfxk_gauss_mfma(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint16_t*)ctx->d_wsplit.p,
                                     radius, ctx->wsplit_inv_scale, ctx->wsplit_bias, ctx->wsplit_bias_single, w, h, first_row, ctx->n_cus > 0 ? ctx->n_cus : 256));
        return PFX_OK;
    }
    const int pad = pfxk_gauss_weight_pad(); // zero taps on both sides: the kernels' register blocking reads past the ends
    if (!ctx->wts_valid || ctx->wts_sigma_bits != sigma_bits) {
        std::vector<float> k;
        pfx_host_gaussian_kernel(sigma, k);
        std::vector<float> padded(k.size() + 2 * (size_t)pad, 0.0f);
        std::copy(k.begin(), k.end(), padded.begin() + pad);
        PFX_TRY(pfx_reserve(ctx, ctx->d_wts, padded.size() * sizeof(float)));
        PFX_TRY(pfx_h2d(ctx, ctx->d_wts.p, padded.data(), padded.size() * sizeof(float)));
        ctx->wts_sigma_bits = sigma_bits;
        ctx->wts_valid = true;
    }
    const float* wts = (const float*)ctx->d_wts.p + pad;
    if (ctx->exact && radius >= 1 && radius <= pfxk_gauss_fused_exact_max_radius() && src_dev != dst_dev) {
        // bit-exact mode at small radii (what sharpen / glow / drop shadow and the batch pipeline run): both passes in one kernel, no f32 intermediate in HBM
        pfx_timer t(ctx, "gauss_fused");
        PFX_HIP(ctx, pfxk_gauss_fused_exact(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, wts, radius, w, h, 0, 0.0f, nullptr));
        return PFX_OK;
    }
    if (!tmp_dev) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)w * h * 16));
        tmp_dev = ctx->st_tmp.p;
    }
    {
        pfx_timer t(ctx, "gauss_h");
        PFX_HIP(ctx, pfxk_gauss_h(ctx->stream, (const uint8_t*)src_dev, (float*)tmp_dev, wts, radius, w, h, ctx->exact ? 1 : 0));
    }
    {
        pfx_timer t(ctx, "gauss_v");
        PFX_HIP(ctx, pfxk_gauss_v(ctx->stream, (const float*)tmp_dev, (uint8_t*)dst_dev, wts, radius, w, h, ctx->exact ? 1 : 0));
    }
    return PFX_OK;
}

int pfx_box_blur_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius,
                     const void* mask_dev, void* tmp_dev)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_box_blur_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_box_blur_dev", true));
    if (radius < 0.5f) { // blur.rs:234: returns flat.clone()
        PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, img_bytes(w, h), hipMemcpyDeviceToDevice, ctx->stream));
        return PFX_OK;
    }
    const float rc = ceilf(radius);
    PFX_REQUIRE(ctx, rc < 2040.0f, "box blur radius too large");
    if (!tmp_dev) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, img_bytes(w, h)));
        tmp_dev = ctx->st_tmp.p;
    }
    pfx_timer t(ctx, "box_blur");
    // in place: the two-pass path (H into tmp, V reads tmp and only its own pixel of src); the fused kernel stages a halo tile from src
    // while neighbouring workgroups write dst
    PFX_HIP(ctx, pfxk_box_blur(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)tmp_dev, (uint8_t*)dst_dev,
                               (const uint8_t*)mask_dev, (int)rc, w, h, src_dev == dst_dev ? 1 : 0));
    return PFX_OK;
}

int pfx_box_blur_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, float radius, const void* mask_dev,
                          void* tmp_dev, uint32_t first_row)
{
    (void)first_row; // integer window sums: a row's result does not depend on where the band was cut (pfx.h)
    return pfx_box_blur_dev(ctx, src_dev, dst_dev, w, h, radius, mask_dev, tmp_dev);
}

int pfx_median_band_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius, const void* mask_dev,
                        uint32_t first_row)
{
    (void)first_row;
    return pfx_median_dev(ctx, src_dev, dst_dev, w, h, radius, mask_dev);
}

int pfx_median_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t radius, const void* mask_dev)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_median_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_median_dev"));
    const uint32_t r = std::max(radius, 1u); // noise.rs:364
    if (r > PFX_MEDIAN_MAX_RADIUS) return pfx_fail(ctx, PFX_ERR_UNSUPPORTED, "median radius %u > %d", r, PFX_MEDIAN_MAX_RADIUS);
    const bool xlane3 = r == 3u && (pfxk_median_get_xlane() & 4);   // 7x7 on the cross-lane network (pfx_tune "median_xlane" bit 2)
    if ((int)r >= ctx->median_bits_min && r <= 8u && !xlane3) { // bit-sliced radix select (k_median_bits.hip); scratch ~ the image size
        PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, pfxk_median_bits_scratch((int)r, w, h)));
        pfx_timer t(ctx, "median");
        PFX_HIP(ctx, pfxk_median_bits(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, (uint32_t*)ctx->st_tmp.p, (int)r, w, h));
        return PFX_OK;
    }
    pfx_timer t(ctx, "median");
    PFX_HIP(ctx, pfxk_median(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev, (int)r, w, h));
    return PFX_OK;
}

int pfx_pixelate_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, uint32_t block_size, const void* mask_dev)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_pixelate_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_pixelate_dev"));
    pfx_timer t(ctx, "pixelate");
    PFX_HIP(ctx, pfxk_pixelate(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev,
                               std::max(block_size, 2u), w, h)); // distort.rs:334
    return PFX_OK;
}

int pfx_adjust_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, int op, const float* params,
                   uint32_t n_params, const uint8_t* lut_host, const void* mask_dev, int sparse_mode)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_adjust_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_adjust_dev", true)); // in place is fine (pointwise); a partial overlap races
    PFX_REQUIRE(ctx, sparse_mode >= PFX_DENSE && sparse_mode <= PFX_IN_PLACE, "unknown sparse mode");
    pfxk_params P;
    bool needs_lut = false;
    PFX_TRY(prepare_adjust(ctx, op, params, n_params, P, needs_lut));
    if (needs_lut) {
        PFX_REQUIRE(ctx, lut_host != nullptr, "this op needs a LUT");
        PFX_TRY(upload_lut(ctx, lut_host, 1024));
    }
    pfx_timer t(ctx, "adjust");
    PFX_HIP(ctx, pfxk_adjust(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, (const uint8_t*)mask_dev,
                             (const uint8_t*)ctx->d_lut.p, op, &P, sparse_mode, w, h));
    return PFX_OK;
}

// parameter block of a Rhai-flavour op (scripting.rs:869-1075); `lut` (256 bytes) is filled when the op reads a table
static int prepare_rhai(pfx_ctx* ctx, int op, const float* p, uint32_t n, pfxk_params& P, uint8_t (&lut)[256], bool& needs_lut)
{
    std::memset(&P, 0, sizeof P);
    needs_lut = false;
    auto need = [&](uint32_t k) { return n >= k && p != nullptr; };
    bool ok = true;
    switch (op) {
    case PFX_RHAI_INVERT: case PFX_RHAI_DESATURATE: case PFX_RHAI_SEPIA: break;
    case PFX_RHAI_SEPIA_STRENGTH: // strength.clamp(0,1) as f32; inv = 1 - strength (scripting.rs:923-924)
        if ((ok = need(1))) { P.p[0] = (float)std::min(std::max((double)p[0], 0.0), 1.0); P.p[1] = 1.0f - P.p[0]; }
        break;
    case PFX_RHAI_BRIGHTNESS_CONTRAST:
        if ((ok = need(2))) { P.p[0] = p[0]; P.p[1] = pfx_host_bc_factor(p[1]); }
        break;
    case PFX_RHAI_HSL:
        if ((ok = need(3))) { P.p[0] = p[0] / 360.0f; P.p[1] = 1.0f + p[1] / 100.0f; P.p[2] = p[2] * 255.0f / 100.0f; }
        break;
    case PFX_RHAI_EXPOSURE:
        if ((ok = need(1))) P.p[0] = pfx_host_exposure_gain(p[0]);
        break;
    case PFX_RHAI_LEVELS:
        if ((ok = need(3))) { pfx_host_rhai_levels_lut(p[0], p[1], p[2], lut); needs_lut = true; }
        break;
    default: return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_rhai_adjust: unknown op %d", op);
    }
    if (!ok) return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_rhai_adjust: op %d needs more parameters", op);
    return PFX_OK;
}

int pfx_rhai_adjust_dev(pfx_ctx* ctx, void* pixels_dev, uint32_t w, uint32_t h, int op, const float* p, uint32_t n)
{
    PFX_TRY(check_img(ctx, pixels_dev, pixels_dev, w, h, "pfx_rhai_adjust_dev"));
    pfxk_params P;
    uint8_t lut[256];
    bool needs_lut = false;
    PFX_TRY(prepare_rhai(ctx, op, p, n, P, lut, needs_lut));
    if (needs_lut) PFX_TRY(upload_lut(ctx, lut, 256));
    pfx_timer t(ctx, "rhai_adjust");
    PFX_HIP(ctx, pfxk_rhai_adjust(ctx->stream, (uint8_t*)pixels_dev, (const uint8_t*)ctx->d_lut.p, op, &P, w, h));
    return PFX_OK;
}

// ---- chains (round 6): several ops, as few passes over memory as the kernels allow; results equal the single-op entry points applied one after the other ----
// A run of pointwise ops is ONE launch (k_pointwise.hip: pointwise_chain_kernel, each op re-quantises to u8 in registers); a bit-exact Gaussian of radius <= 16
// followed by pointwise ops is ONE launch too (k_gauss_exact.hip, epilogue 3: the blurred image never exists in memory).  Every other stencil op runs as its own
// launch and hands its result to the next stage through at most one scratch image.  References: the ops' own (src/ops/filters.rs:214-316, src/ops/adjustments.rs,
// src/ops/scripting.rs:869-1075); the chain itself is how a script `apply_gaussian_blur(..); apply_hsl(..);` or the batch pipeline strings them together.
namespace {
struct chain_stage { int stencil = -1; uint32_t first = 0, count = 0; };   // stencil: index of the stage's Gaussian / box op or -1; [first, first + count): its pointwise ops
bool chain_pointwise(const pfx_chain_op& o) { return o.kind == PFX_CHAIN_ADJUST || o.kind == PFX_CHAIN_RHAI; }
}

static int chain_prepare(pfx_ctx* ctx, const pfx_chain_op* ops, uint32_t first, uint32_t count, pfxk_chain& C)
{
    std::memset(&C, 0, sizeof C);
    uint8_t luts[PFXK_CHAIN_LUTS][1024];
    for (uint32_t k = 0; k < count; ++k) {
        const pfx_chain_op& o = ops[first + k];
        bool needs_lut = false;
        if (o.kind == PFX_CHAIN_ADJUST) {
            PFX_TRY(prepare_adjust(ctx, o.op, o.params, o.n_params, C.P[k], needs_lut));
            C.op[k] = (uint32_t)o.op;
            if (needs_lut) {
                PFX_REQUIRE(ctx, o.lut != nullptr, "pfx_chain_dev: this op needs a LUT");
                std::memcpy(luts[C.n_luts], o.lut, 1024);
            }
        } else {
            uint8_t l256[256];
            PFX_TRY(prepare_rhai(ctx, o.op, o.params, o.n_params, C.P[k], l256, needs_lut));
            C.op[k] = PFXK_CHAIN_RHAI | (uint32_t)o.op;
            if (needs_lut) { std::memset(luts[C.n_luts], 0, 1024); std::memcpy(luts[C.n_luts], l256, 256); }
        }
        if (needs_lut) C.lut_slot[k] = C.n_luts++;
    }
    C.n = count;
    if (C.n_luts) {
        PFX_TRY(pfx_reserve(ctx, ctx->d_chain_luts, (size_t)PFXK_CHAIN_LUTS * 1024));
        PFX_TRY(pfx_h2d(ctx, ctx->d_chain_luts.p, luts, (size_t)C.n_luts * 1024));
    }
    return PFX_OK;
}

int pfx_chain_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h, const pfx_chain_op* ops, uint32_t n_ops)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_chain_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_chain_dev", true));
    PFX_REQUIRE(ctx, ops != nullptr || n_ops == 0, "pfx_chain_dev: null op list");
    const size_t bytes = img_bytes(w, h);
    // stages: [stencil op +] a run of pointwise ops that one launch can carry (PFXK_CHAIN_MAX ops, PFXK_CHAIN_LUTS tables)
    std::vector<chain_stage> stages;
    uint32_t n_stencil = 0;
    for (uint32_t i = 0; i < n_ops;) {
        chain_stage st;
        if (!chain_pointwise(ops[i])) {
            if (ops[i].kind != PFX_CHAIN_GAUSSIAN && ops[i].kind != PFX_CHAIN_BOX) return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_chain_dev: unknown op kind %d", ops[i].kind);
            PFX_REQUIRE(ctx, ops[i].n_params >= 1, "pfx_chain_dev: a blur needs its sigma / radius in params[0]");
            st.stencil = (int)i++;
            ++n_stencil;
        }
        st.first = i;
        uint32_t luts = 0;
        while (i < n_ops && chain_pointwise(ops[i]) && st.count < PFXK_CHAIN_MAX) {
            const bool lut_op = ops[i].kind == PFX_CHAIN_ADJUST ? (ops[i].op == PFX_OP_GRADIENT_MAP || ops[i].op == PFX_OP_LUT_RGBA) : ops[i].op == PFX_RHAI_LEVELS;
            if (lut_op && luts == PFXK_CHAIN_LUTS) break;
            luts += lut_op ? 1u : 0u;
            ++st.count; ++i;
        }
        stages.push_back(st);
    }
    if (src_dev == dst_dev) PFX_REQUIRE(ctx, n_stencil == 0, "pfx_chain_dev: a chain with a blur cannot run in place");
    if (stages.empty()) {
        if (src_dev != dst_dev) PFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return PFX_OK;
    }
    if (n_stencil > 1 || (n_stencil == 1 && stages.front().stencil < 0)) PFX_TRY(pfx_reserve(ctx, ctx->st_chain, bytes));
    // a stencil stage writes dst or the scratch image, alternating so that the LAST one writes dst (pointwise stages behind a stencil run in place)
    uint32_t seen = 0;
    auto stencil_out = [&](uint32_t j) -> void* { return ((n_stencil - 1u - j) % 2u == 0u) ? dst_dev : ctx->st_chain.p; };
    const void* cur = src_dev;
    for (const chain_stage& st : stages) {
        pfxk_chain C;
        if (st.count) PFX_TRY(chain_prepare(ctx, ops, st.first, st.count, C));
        if (st.stencil < 0) {
            // pointwise only: in place behind a stencil; in front of the first one into the buffer that stencil does not write
            void* out = cur == src_dev ? (n_stencil == 0 ? dst_dev : (stencil_out(0) == dst_dev ? ctx->st_chain.p : dst_dev)) : const_cast<void*>(cur);
            pfx_timer t(ctx, "chain");
            PFX_HIP(ctx, pfxk_pointwise_chain(ctx->stream, (const uint8_t*)cur, (uint8_t*)out, (const uint8_t*)ctx->d_chain_luts.p, &C, w, h));
            cur = out;
            continue;
        }
        const pfx_chain_op& so = ops[st.stencil];
        void* out = stencil_out(seen++);
        bool fused = false;
        // The pointwise run rides in the Gaussian's store only when it is light: a Gaussian kernel is bound by its own arithmetic and dependency chains, not by HBM,
        // so an op of ~140 instructions per pixel (HSL, vibrance) costs there what it costs as a streaming pass of its own and the fused launch saves nothing
        // (8K sigma 16 -> HSL: 0.210 ms fused against 0.206 as two launches; bit-exact sigma 4 -> HSL 0.381 against 0.371: profiles/r06_tuning.md); light ops
        // (invert, exposure, brightness / contrast, tables ...) cost a few instructions and save the 8 B/px round trip.  pfx_tune "chain_fuse_heavy" = 1 fuses anyway.
        bool heavy = false;
        for (uint32_t k = 0; k < st.count; ++k) {
            const pfx_chain_op& o = ops[st.first + k];
            heavy = heavy || (o.kind == PFX_CHAIN_ADJUST && (o.op == PFX_OP_HSL || o.op == PFX_OP_VIBRANCE)) || (o.kind == PFX_CHAIN_RHAI && o.op == PFX_RHAI_HSL);
        }
        if (so.kind == PFX_CHAIN_GAUSSIAN && st.count && (!heavy || ctx->chain_fuse_heavy)) {
            const float sigma = so.params[0];
            const int radius = pfx_host_gaussian_radius(sigma);
            if (ctx->exact && radius >= 1 && radius <= pfxk_gauss_fused_exact_max_radius()) {
                const float* wts = nullptr;
                PFX_TRY(pfx_int_gauss_exact_weights(ctx, sigma, &wts));
                pfx_timer t(ctx, "gauss_fused_chain");
                PFX_HIP(ctx, pfxk_gauss_fused_exact_chain(ctx->stream, (const uint8_t*)cur, (uint8_t*)out, wts, radius, w, h, &C, (const uint8_t*)ctx->d_chain_luts.p));
                fused = true;
            } else if (!ctx->exact && radius >= 1 && radius <= pfxk_gauss_mfma_max_radius() && C.n_luts == 0 && ctx->chain_mfma_epilogue) {
                // default mode: the chain rides in the matrix-core Gaussian's store (table-free ops, aligned buffers; otherwise the two launches below)
                PFX_TRY(gauss_mfma_tables(ctx, sigma));
                pfx_timer t(ctx, "gauss_mfma_chain");
                const hipError_t e = pfxk_gauss_mfma_chain(ctx->stream, (const uint8_t*)cur, (uint8_t*)out, (const uint16_t*)ctx->d_wsplit.p, radius, ctx->wsplit_inv_scale,
                                                           ctx->wsplit_bias, ctx->wsplit_bias_single, w, h, 0u, ctx->n_cus > 0 ? ctx->n_cus : 256, &C);
                if (e == hipSuccess) fused = true;
                else if (e != hipErrorNotSupported) PFX_HIP(ctx, e);
                else (void)hipGetLastError();
            }
        }
        if (!fused) {
            if (so.kind == PFX_CHAIN_GAUSSIAN) PFX_TRY(pfx_gaussian_blur_dev(ctx, cur, out, w, h, so.params[0], nullptr));
            else PFX_TRY(pfx_box_blur_dev(ctx, cur, out, w, h, so.params[0], nullptr, nullptr));
            if (st.count) {
                pfx_timer t(ctx, "chain");
                PFX_HIP(ctx, pfxk_pointwise_chain(ctx->stream, (const uint8_t*)out, (uint8_t*)out, (const uint8_t*)ctx->d_chain_luts.p, &C, w, h));
            }
        }
        cur = out;
    }
    if (cur != dst_dev) PFX_HIP(ctx, hipMemcpyAsync(dst_dev, cur, bytes, hipMemcpyDeviceToDevice, ctx->stream));   // unreachable by construction; kept as a guard
    return PFX_OK;
}

int pfx_warp_displacement_band_dev(pfx_ctx* ctx, const void* src_dev, uint32_t sw, uint32_t sh, const void* disp_band_dev, uint32_t w,
                                   uint32_t band_rows, void* dst_band_dev, uint32_t first_row)
{
    PFX_TRY(check_img(ctx, src_dev, dst_band_dev, w, band_rows, "pfx_warp_displacement_dev"));
    PFX_REQUIRE(ctx, disp_band_dev && pfx_dims_ok(sw, sh), "pfx_warp_displacement_dev: bad arguments");
    PFX_REQUIRE(ctx, (uint64_t)first_row + band_rows <= 0x7fffffffull, "pfx_warp_displacement_dev: band outside any image");
    PFX_TRY(check_disjoint2(ctx, src_dev, sw, sh, dst_band_dev, w, band_rows, "pfx_warp_displacement_dev")); // the SOURCE's extent: it may be larger than the output
    pfx_timer t(ctx, "warp_displacement");
    PFX_HIP(ctx, pfxk_warp_displacement(ctx->stream, (const uint8_t*)src_dev, sw, sh, (const float*)disp_band_dev, w, band_rows, (uint8_t*)dst_band_dev, first_row));
    return PFX_OK;
}

int pfx_warp_displacement_dev(pfx_ctx* ctx, const void* src_dev, uint32_t sw, uint32_t sh, const void* disp_dev, uint32_t w,
                              uint32_t h, void* dst_dev)
{
    return pfx_warp_displacement_band_dev(ctx, src_dev, sw, sh, disp_dev, w, h, dst_dev, 0);
}

// DisplacementField::apply_* on a device-resident field: per-dab prologue (radius clamp, sigma, loop bounds) on the host exactly as
// the reference computes it (transform.rs:1056-1069), accumulation in k_warp.hip
int pfx_displacement_brushes_dev(pfx_ctx* ctx, void* disp_dev, uint32_t w, uint32_t h, const pfx_disp_dab* dabs, uint32_t n_dabs)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, disp_dev && w && h && (uint64_t)w * h <= 256000000ull && (n_dabs == 0 || dabs), "pfx_displacement_brushes_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    if (n_dabs == 0) return PFX_OK;
    auto f2i = [](float v) -> int32_t { return v != v ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)v)); };
    std::vector<pfxk_disp_dab> k(n_dabs);
    int bx0 = (int)w, by0 = (int)h, bx1 = 0, by1 = 0;
    for (uint32_t i = 0; i < n_dabs; ++i) {
        const pfx_disp_dab& D = dabs[i];
        PFX_REQUIRE(ctx, D.mode >= 0 && D.mode <= 4, "pfx_displacement_brushes_dev: unknown brush mode");
        pfxk_disp_dab& K = k[i];
        K.mode = D.mode;
        K.cx = D.cx; K.cy = D.cy; K.delta_x = D.delta_x; K.delta_y = D.delta_y; K.strength = D.strength;
        K.r = fmaxf(D.radius, 1.0f);
        const float sigma = K.r / 3.0f;
        K.sigma_sq_2 = 2.0f * sigma * sigma;
        K.x0 = std::max(f2i(floorf(D.cx - K.r)), 0);
        K.y0 = std::max(f2i(floorf(D.cy - K.r)), 0);
        K.x1 = std::min(f2i(ceilf(D.cx + K.r)), (int32_t)w);
        K.y1 = std::min(f2i(ceilf(D.cy + K.r)), (int32_t)h);
        if (K.x1 > K.x0 && K.y1 > K.y0) { bx0 = std::min(bx0, K.x0); by0 = std::min(by0, K.y0); bx1 = std::max(bx1, K.x1); by1 = std::max(by1, K.y1); }
    }
    if (bx1 <= bx0 || by1 <= by0) return PFX_OK; // every dab is off-canvas
    // Dabs spread over a large field: launch over the 64 x 64 chunks their boxes touch instead of the common bounding box when that is less than half of it
    std::vector<uint32_t> chunks;
    {
        const uint32_t cx0 = (uint32_t)bx0 >> 6, cy0 = (uint32_t)by0 >> 6, ncx = (((uint32_t)bx1 + 63u) >> 6) - cx0, ncy = (((uint32_t)by1 + 63u) >> 6) - cy0;
        if ((uint64_t)ncx * ncy >= 256u && ncx < 65536u && ncy < 65536u) {
            std::vector<uint8_t> hit((size_t)ncx * ncy, 0);
            for (const pfxk_disp_dab& K : k) {
                if (!(K.x1 > K.x0 && K.y1 > K.y0)) continue;
                for (uint32_t cy = (uint32_t)K.y0 >> 6; cy <= ((uint32_t)K.y1 - 1u) >> 6; ++cy)
                    for (uint32_t cx = (uint32_t)K.x0 >> 6; cx <= ((uint32_t)K.x1 - 1u) >> 6; ++cx) hit[(size_t)(cy - cy0) * ncx + (cx - cx0)] = 1;
            }
            for (uint32_t cy = 0; cy < ncy; ++cy)
                for (uint32_t cx = 0; cx < ncx; ++cx)
                    if (hit[(size_t)cy * ncx + cx]) chunks.push_back((cx0 + cx) | ((cy0 + cy) << 16));
            if (chunks.size() * 2u > (size_t)ncx * ncy) chunks.clear();   // mostly covered: the plain launch
        }
    }
    const size_t dab_bytes = (k.size() * sizeof(pfxk_disp_dab) + 15u) & ~(size_t)15u;
    PFX_TRY(pfx_reserve(ctx, ctx->d_pts, dab_bytes + chunks.size() * 4u + 16u));
    PFX_TRY(pfx_h2d(ctx, ctx->d_pts.p, k.data(), k.size() * sizeof(pfxk_disp_dab)));
    if (!chunks.empty()) PFX_TRY(pfx_h2d(ctx, (uint8_t*)ctx->d_pts.p + dab_bytes, chunks.data(), chunks.size() * 4u));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `k` / `chunks` are pageable host memory about to go out of scope
    pfx_timer t(ctx, "displacement_brush");
    if (!chunks.empty())
        PFX_HIP(ctx, pfxk_disp_brushes_chunked(ctx->stream, (float*)disp_dev, w, h, (const pfxk_disp_dab*)ctx->d_pts.p, n_dabs, bx0, by0, bx1, by1,
                                               (const uint32_t*)((const uint8_t*)ctx->d_pts.p + dab_bytes), (uint32_t)chunks.size()));
    else
        PFX_HIP(ctx, pfxk_disp_brushes(ctx->stream, (float*)disp_dev, w, h, (const pfxk_disp_dab*)ctx->d_pts.p, n_dabs, bx0, by0, bx1, by1));
    return PFX_OK;
}

static int upload_points(pfx_ctx* ctx, const float* orig, const float* def, uint32_t cols, uint32_t rows, const float** d_orig,
                         const float** d_def)
{
    PFX_REQUIRE(ctx, def && cols >= 1 && rows >= 1 && cols <= 4096 && rows <= 4096, "mesh: bad grid");
    const size_t npts = (size_t)(cols + 1) * (rows + 1);
    PFX_TRY(pfx_reserve(ctx, ctx->d_pts, npts * 2 * sizeof(float) * 2));
    float* base = (float*)ctx->d_pts.p;
    PFX_TRY(pfx_h2d(ctx, base, def, npts * 2 * sizeof(float)));
    *d_def = base;
    *d_orig = nullptr;
    if (orig) {
        PFX_TRY(pfx_h2d(ctx, base + npts * 2, orig, npts * 2 * sizeof(float)));
        *d_orig = base + npts * 2;
    }
    return PFX_OK;
}

int pfx_mesh_displacement_dev(pfx_ctx* ctx, const float* orig_pts_xy, const float* deformed_pts_xy, uint32_t cols, uint32_t rows,
                              uint32_t w, uint32_t h, void* disp_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, disp_dev && pfx_dims_ok(w, h), "pfx_mesh_displacement_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const float *d_orig, *d_def;
    PFX_TRY(upload_points(ctx, orig_pts_xy, deformed_pts_xy, cols, rows, &d_orig, &d_def));
    pfx_timer t(ctx, "mesh_displacement");
    PFX_HIP(ctx, pfxk_mesh_displacement(ctx->stream, d_orig, d_def, cols, rows, w, h, (float*)disp_dev));
    return PFX_OK;
}

int pfx_warp_mesh_catmull_rom_band_dev(pfx_ctx* ctx, const void* src_dev, const float* orig_pts_xy, const float* deformed_pts_xy,
                                       uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, void* dst_band_dev, uint32_t first_row, uint32_t band_rows)
{
    PFX_TRY(check_img(ctx, src_dev, dst_band_dev, w, h, "pfx_warp_mesh_catmull_rom_dev"));
    PFX_REQUIRE(ctx, band_rows >= 1 && (uint64_t)first_row + band_rows <= h, "pfx_warp_mesh_catmull_rom_dev: band outside the image");
    PFX_REQUIRE(ctx, !pfx_ranges_overlap(src_dev, (size_t)w * h * 4, dst_band_dev, (size_t)w * band_rows * 4), "pfx_warp_mesh_catmull_rom_dev: src and dst overlap");
    const float *d_orig, *d_def;
    PFX_TRY(upload_points(ctx, orig_pts_xy, deformed_pts_xy, cols, rows, &d_orig, &d_def));
    pfx_timer t(ctx, "warp_mesh");
    PFX_HIP(ctx, pfxk_warp_mesh(ctx->stream, (const uint8_t*)src_dev, d_orig, d_def, cols, rows, w, band_rows, (uint8_t*)dst_band_dev, first_row, h));
    return PFX_OK;
}

int pfx_warp_mesh_catmull_rom_dev(pfx_ctx* ctx, const void* src_dev, const float* orig_pts_xy, const float* deformed_pts_xy,
                                  uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, void* dst_dev)
{
    return pfx_warp_mesh_catmull_rom_band_dev(ctx, src_dev, orig_pts_xy, deformed_pts_xy, cols, rows, w, h, dst_dev, 0, h);
}

// Per-stamp host prologue of draw_circle_no_dirty / draw_image_tip_no_dirty (brush_render.rs:148-256, 552-632): the reference
// computes scatter, colour jitter and tip rotation on the CPU once per stamp; so does this, then the stamp list goes to the kernel.
static uint32_t stamp_hash(float x, float y, uint32_t counter) // :846-857
{
    auto u = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v); };
    uint32_t hsh = u(x * 100.0f) * 374761393u + u(y * 100.0f) * 668265263u + counter * 1013904223u;
    hsh ^= hsh >> 13;
    hsh *= 1274126177u;
    hsh ^= hsh >> 16;
    return hsh;
}
static void host_rgb_to_hsl(float r, float g, float b, float& hh, float& s, float& l) // adjustments.rs:944-974
{
    const float mx = fmaxf(fmaxf(r, g), b), mn = fminf(fminf(r, g), b);
    l = (mx + mn) / 2.0f;
    if (fabsf(mx - mn) < 1e-6f) { hh = 0.0f; s = 0.0f; return; }
    const float d = mx - mn;
    s = l > 0.5f ? d / (2.0f - mx - mn) : d / (mx + mn);
    if (fabsf(mx - r) < 1e-6f) { float t = (g - b) / d; if (t < 0.0f) t += 6.0f; hh = t / 6.0f; }
    else if (fabsf(mx - g) < 1e-6f) hh = ((b - r) / d + 2.0f) / 6.0f;
    else hh = ((r - g) / d + 4.0f) / 6.0f;
}
static float host_hue_to_rgb(float p, float q, float t) // adjustments.rs:995-1012
{
    if (t < 0.0f) t += 1.0f;
    if (t > 1.0f) t -= 1.0f;
    if (t < 1.0f / 6.0f) return p + (q - p) * 6.0f * t;
    if (t < 1.0f / 2.0f) return q;
    if (t < 2.0f / 3.0f) return p + (q - p) * (2.0f / 3.0f - t) * 6.0f;
    return p;
}
static void host_hsl_to_rgb(float hh, float s, float l, float& r, float& g, float& b) // adjustments.rs:976-993
{
    if (fabsf(s) < 1e-6f) { r = g = b = l; return; }
    const float q = l < 0.5f ? l * (1.0f + s) : l + s - l * s;
    const float p = 2.0f * l - q;
    r = host_hue_to_rgb(p, q, hh + 1.0f / 3.0f);
    g = host_hue_to_rgb(p, q, hh);
    b = host_hue_to_rgb(p, q, hh - 1.0f / 3.0f);
}

int pfx_brush_stamps_ex_dev(pfx_ctx* ctx, void* target_dev, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn,
                            const float* points_xy, uint32_t n_points, const void* selection_dev)
{
    PFX_TRY(check_img(ctx, target_dev, target_dev, w, h, "pfx_brush_stamps_dev"));
    if (n_points == 0) return PFX_OK;
    PFX_REQUIRE(ctx, points_xy != nullptr, "null stamp list");
    const bool image_tip = dyn && dyn->tip_mask;
    pfxk_brush B;
    bool skip;
    PFX_TRY(brush_prepare(ctx, brush, B, skip));
    if (image_tip) {
        if (dyn->tip_mask_size == 0) return PFX_OK; // :546-549
        PFX_REQUIRE(ctx, dyn->tip_mask_size <= 8192, "brush tip mask too large");
        B.tip_size = dyn->tip_mask_size;
    } else if (skip) return PFX_OK; // radius_sq < 0.001 (brush_render.rs:198)

    auto u8 = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 255.0f ? 255u : (uint32_t)(int)v); };
    std::vector<pfxk_stamp> st(n_points);
    const float half = image_tip ? (float)dyn->tip_mask_size / 2.0f : 0.0f;
    float reach = image_tip ? half : B.draw_radius;
    float minx = 0, maxx = 0, miny = 0, maxy = 0;
    for (uint32_t i = 0; i < n_points; ++i) {
        const float px = points_xy[2 * i], py = points_xy[2 * i + 1];
        pfxk_stamp& S = st[i];
        S.cx = px; S.cy = py; S.cos_a = 1.0f; S.sin_a = 0.0f; S.rotated = 0;
        if (dyn && dyn->scatter > 0.01f) { // :179-193 / :552-565
            const float diam = brush->size; // pressure_size() without pen pressure
            const float h1 = (float)stamp_hash(px, py, dyn->stamp_counter) / 4294967295.0f;
            const float h2 = (float)stamp_hash(py, px, dyn->stamp_counter + 99991u) / 4294967295.0f;
            S.cx = px + (h1 * 2.0f - 1.0f) * dyn->scatter * diam;
            S.cy = py + (h2 * 2.0f - 1.0f) * dyn->scatter * diam;
        }
        S.rgb8 = B.rgb8;
        if (dyn && (dyn->hue_jitter > 0.01f || dyn->brightness_jitter > 0.01f)) { // :226-256 / :602-632
            float hh, s, l, nr, ng, nb;
            host_rgb_to_hsl(brush->color[0], brush->color[1], brush->color[2], hh, s, l);
            if (dyn->hue_jitter > 0.01f) {
                const float hv = (float)stamp_hash(px + 0.1f, py + 0.2f, dyn->stamp_counter + 777u) / 4294967295.0f;
                const float v = hh + (hv * 2.0f - 1.0f) * dyn->hue_jitter * 0.5f;
                hh = v - truncf(v);
                if (hh < 0.0f) hh += 1.0f;
            }
            if (dyn->brightness_jitter > 0.01f) {
                const float bv = (float)stamp_hash(px + 0.3f, py + 0.4f, dyn->stamp_counter + 555u) / 4294967295.0f;
                l = pfx_clampf(l + (bv * 2.0f - 1.0f) * dyn->brightness_jitter * 0.5f, 0.0f, 1.0f);
            }
            host_hsl_to_rgb(hh, s, l, nr, ng, nb);
            S.rgb8 = u8(nr * 255.0f) | (u8(ng * 255.0f) << 8) | (u8(nb * 255.0f) << 16);
        }
        if (image_tip) { // :148-163, :570-581
            float rotation_deg = dyn->tip_rotation;
            if (dyn->tip_random_rotation) {
                const float lo = dyn->tip_rotation_lo, range = dyn->tip_rotation_hi - lo;
                rotation_deg = fabsf(range) < 0.01f ? lo : lo + (float)(stamp_hash(px, py, dyn->stamp_counter) % 10000u) / 10000.0f * range;
            }
            if (fabsf(rotation_deg) > 0.01f) {
                const float rad = -(rotation_deg * (3.14159265358979323846f / 180.0f)); // negative: inverse rotation for sampling
                S.cos_a = cosf(rad); S.sin_a = sinf(rad); S.rotated = 1;
                reach = std::max(reach, half * 1.41421356237309504880f);
            }
        }
        if (i == 0) { minx = maxx = S.cx; miny = maxy = S.cy; }
        else { minx = std::min(minx, S.cx); maxx = std::max(maxx, S.cx); miny = std::min(miny, S.cy); maxy = std::max(maxy, S.cy); }
        // the stamp's pixel box, with the reference's float operations (:209-215 round tip, :584-587 image tip; `as u32` saturates, NaN -> 0): the kernel tests
        // pixels against it and the binning below deals the stamp to the chunks it touches
        {
            auto f2u = [](float v) -> uint32_t { return (v > 0.0f) ? ((v >= 4294967296.0f) ? 0xffffffffu : (uint32_t)v) : 0u; };
            const uint32_t wm1 = w - 1u, hm1 = h - 1u;
            if (image_tip) {
                const float eh = S.rotated ? half * 1.41421356237309504880f : half;
                S.x0 = f2u(fmaxf(S.cx - eh, 0.0f)); S.y0 = f2u(fmaxf(S.cy - eh, 0.0f));
                S.x1 = std::min(f2u(S.cx + eh), wm1); S.y1 = std::min(f2u(S.cy + eh), hm1);
            } else {
                S.x0 = f2u(fmaxf(floorf(S.cx - B.draw_radius), 0.0f)); S.x1 = std::min(f2u(ceilf(S.cx + B.draw_radius)), wm1);
                S.y0 = f2u(fmaxf(floorf(S.cy - B.draw_radius), 0.0f)); S.y1 = std::min(f2u(ceilf(S.cy + B.draw_radius)), hm1);
            }
            if (S.x0 > S.x1 || S.y0 > S.y1) { S.x0 = 1u; S.x1 = 0u; S.y0 = 1u; S.y1 = 0u; }   // touches no pixel
            S.pad[0] = S.pad[1] = 0u;
        }
    }
    // bounding box of the whole stroke: union of the per-stamp boxes (brush_render.rs:209-215, 584-587) plus slack
    const float pad = reach + 2.0f;
    if (!(minx == minx && maxx == maxx && miny == miny && maxy == maxy)) return pfx_fail(ctx, PFX_ERR_INVALID, "stamp positions must be finite");
    const int bx0 = (int)std::max(0.0f, floorf(std::max(minx - pad, -1.0e9f))), by0 = (int)std::max(0.0f, floorf(std::max(miny - pad, -1.0e9f)));
    const int bx1 = (int)std::min((float)(w - 1), ceilf(std::min(maxx + pad, 1.0e9f))), by1 = (int)std::min((float)(h - 1), ceilf(std::min(maxy + pad, 1.0e9f)));
    if (bx1 < bx0 || by1 < by0) return PFX_OK;
    const size_t stamp_bytes = st.size() * sizeof(pfxk_stamp), tip_bytes = image_tip ? (size_t)dyn->tip_mask_size * dyn->tip_mask_size : 0;
    // Strokes of more than one cull group: deal the stamps to the 64 x 64 chunks (TiledImage's grid) their boxes touch, so that the kernel's work follows the painted
    // area instead of the stroke's bounding box times its length.  A chunk's list holds every stamp whose pixel box (above) touches it, in stroke order.
    std::vector<uint32_t> chunk_tab, bins;
    if (n_points > 64u && ctx->brush_binning) {
        const uint32_t ncx = (w + 63u) / 64u, ncy = (h + 63u) / 64u;
        std::vector<uint32_t> box(4 * (size_t)n_points), count((size_t)ncx * ncy, 0u);
        uint64_t entries = 0;
        for (uint32_t i = 0; i < n_points && entries <= (8ull << 20); ++i) {
            const pfxk_stamp& S = st[i];
            uint32_t* b = &box[4 * (size_t)i];
            if (S.x0 > S.x1) { b[0] = 1; b[1] = 0; b[2] = 1; b[3] = 0; continue; }
            b[0] = S.x0 >> 6; b[1] = S.x1 >> 6; b[2] = S.y0 >> 6; b[3] = S.y1 >> 6;
            entries += (uint64_t)(b[1] - b[0] + 1u) * (b[3] - b[2] + 1u);
        }
        if (entries > 0 && entries <= (8ull << 20)) {   // beyond 8 M entries (huge tips on long strokes) the unbinned kernel runs
            for (uint32_t i = 0; i < n_points; ++i) {
                const uint32_t* b = &box[4 * (size_t)i];
                for (uint32_t cy = b[2]; cy <= b[3] && b[0] <= b[1]; ++cy)
                    for (uint32_t cx = b[0]; cx <= b[1]; ++cx) ++count[(size_t)cy * ncx + cx];
            }
            std::vector<uint32_t> first(count.size());
            uint32_t off = 0;
            for (size_t c = 0; c < count.size(); ++c) {
                first[c] = off;
                if (count[c]) { chunk_tab.push_back((uint32_t)(c % ncx)); chunk_tab.push_back((uint32_t)(c / ncx)); chunk_tab.push_back(off); chunk_tab.push_back(count[c]); }
                off += count[c];
            }
            bins.resize(off);
            for (uint32_t i = 0; i < n_points; ++i) {   // stamps in order: every chunk's list comes out in stroke order
                const uint32_t* b = &box[4 * (size_t)i];
                for (uint32_t cy = b[2]; cy <= b[3] && b[0] <= b[1]; ++cy)
                    for (uint32_t cx = b[0]; cx <= b[1]; ++cx) bins[first[(size_t)cy * ncx + cx]++] = i;
            }
        }
    }
    const size_t tab_off = (stamp_bytes + tip_bytes + 15u) & ~(size_t)15u, tab_bytes = chunk_tab.size() * 4u, bins_bytes = bins.size() * 4u;
    PFX_TRY(pfx_reserve(ctx, ctx->d_pts, tab_off + tab_bytes + bins_bytes + 512));
    PFX_TRY(pfx_h2d(ctx, ctx->d_pts.p, st.data(), stamp_bytes));
    const uint8_t* d_tip = nullptr;
    if (image_tip) {
        PFX_TRY(pfx_h2d(ctx, (uint8_t*)ctx->d_pts.p + stamp_bytes, dyn->tip_mask, tip_bytes));
        d_tip = (const uint8_t*)ctx->d_pts.p + stamp_bytes;
    }
    if (!chunk_tab.empty()) {
        PFX_TRY(pfx_h2d(ctx, (uint8_t*)ctx->d_pts.p + tab_off, chunk_tab.data(), tab_bytes));
        PFX_TRY(pfx_h2d(ctx, (uint8_t*)ctx->d_pts.p + tab_off + tab_bytes, bins.data(), bins_bytes));
    }
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `st` is pageable host memory about to go out of scope
    const uint8_t* d_lut = nullptr;
    if (!image_tip && !B.use_direct_alpha) { // LUT path only when AA is off (brush_render.rs:329-337)
        uint8_t lut[256];
        pfx_host_brush_lut(brush->size, brush->hardness, brush->anti_aliased != 0, lut);
        PFX_TRY(upload_lut(ctx, lut, 256));
        d_lut = (const uint8_t*)ctx->d_lut.p;
    }
    pfx_timer t(ctx, "brush_stamps");
    if (!chunk_tab.empty())
        PFX_HIP(ctx, pfxk_brush_stamps_binned(ctx->stream, (uint8_t*)target_dev, w, h, &B, (const pfxk_stamp*)ctx->d_pts.p, n_points, d_lut, d_tip,
                                              (const uint8_t*)selection_dev, (const uint32_t*)((const uint8_t*)ctx->d_pts.p + tab_off), (uint32_t)(chunk_tab.size() / 4u),
                                              (const uint32_t*)((const uint8_t*)ctx->d_pts.p + tab_off + tab_bytes)));
    else
        PFX_HIP(ctx, pfxk_brush_stamps(ctx->stream, (uint8_t*)target_dev, w, h, &B, (const pfxk_stamp*)ctx->d_pts.p, n_points, d_lut, d_tip,
                                       (const uint8_t*)selection_dev, bx0, by0, bx1, by1));
    return PFX_OK;
}

int pfx_brush_stamps_dev(pfx_ctx* ctx, void* target_dev, uint32_t w, uint32_t h, const pfx_brush* brush, const float* points_xy,
                         uint32_t n_points, const void* selection_dev)
{
    return pfx_brush_stamps_ex_dev(ctx, target_dev, w, h, brush, nullptr, points_xy, n_points, selection_dev);
}

int pfx_tiled_roundtrip_dev(pfx_ctx* ctx, const void* src_dev, void* dst_dev, uint32_t w, uint32_t h)
{
    PFX_TRY(check_img(ctx, src_dev, dst_dev, w, h, "pfx_tiled_roundtrip_dev"));
    PFX_TRY(check_disjoint(ctx, src_dev, dst_dev, w, h, "pfx_tiled_roundtrip_dev", true));
    pfx_timer t(ctx, "tiled_roundtrip");
    PFX_HIP(ctx, pfxk_tiled_roundtrip(ctx->stream, (const uint8_t*)src_dev, (uint8_t*)dst_dev, w, h));
    return PFX_OK;
}

// ======================================================================= host-buffer tier
int pfx_gaussian_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float sigma, const uint8_t* mask)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_gaussian_blur_core"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_int_blur_with_selection_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, sigma, mask, d_mask));
    return finish_out(ctx, dst, w, h);
}

} // extern "C"

// blur_with_selection (ref: src/ops/filters.rs:141-207) on device-resident images.  The reference blurs the
// selection's bounding box padded by the kernel radius *as its own image* (clamp-to-edge happens at the crop border)
// and copies back where mask > 0; the crop is reproduced exactly.  `mask_host` is scanned for the bounding box on the
// host (like the reference), `d_mask` is the same mask on the device.
int pfx_int_blur_with_selection_dev(pfx_ctx* ctx, const void* d_src, void* d_dst, uint32_t w, uint32_t h, float sigma,
                                    const uint8_t* mask, const void* d_mask)
{
    if (!mask) return pfx_gaussian_blur_dev(ctx, d_src, d_dst, w, h, sigma, nullptr);
    struct { void* p; } in{const_cast<void*>(d_src)}, out{d_dst};
    {
        // bounding box of mask > 0 (filters.rs:150-163).  Per row: the first and the last non-zero byte, found 8 bytes at a time — the
        // byte-by-byte scan with four min / max per selected pixel took tens of milliseconds on an 8K mask, more than the blur
        uint32_t min_x = w, min_y = h, max_x = 0, max_y = 0;
        for (uint32_t y = 0; y < h; ++y) {
            const uint8_t* row = mask + (size_t)y * w;
            uint32_t a = 0;
            while (a + 8 <= w) { uint64_t v; std::memcpy(&v, row + a, 8); if (v) break; a += 8; }
            while (a < w && row[a] == 0) ++a;
            if (a == w) continue; // nothing selected in this row
            uint32_t b = w;
            while (b >= a + 8) { uint64_t v; std::memcpy(&v, row + b - 8, 8); if (v) break; b -= 8; }
            while (b > a && row[b - 1] == 0) --b;
            min_x = std::min(min_x, a); max_x = std::max(max_x, b - 1);
            min_y = std::min(min_y, y); max_y = std::max(max_y, y);
        }
        if (min_x > max_x || min_y > max_y) { // nothing selected: flat.clone()
            PFX_HIP(ctx, hipMemcpyAsync(out.p, in.p, img_bytes(w, h), hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            const float padf = ceilf(sigma * 3.0f);
            const uint32_t pad = !(padf > 0.0f) ? 0u : (padf >= 4294967296.0f ? 0xffffffffu : (uint32_t)padf);
            const uint32_t cx0 = min_x > pad ? min_x - pad : 0, cy0 = min_y > pad ? min_y - pad : 0;
            const uint32_t cx1 = (uint32_t)std::min<uint64_t>((uint64_t)max_x + 1 + pad, w);
            const uint32_t cy1 = (uint32_t)std::min<uint64_t>((uint64_t)max_y + 1 + pad, h);
            const uint32_t cw = cx1 - cx0, ch = cy1 - cy0;
            PFX_TRY(pfx_reserve(ctx, ctx->st_aux, img_bytes(cw, ch)));
            PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, img_bytes(w, h)));
            PFX_HIP(ctx, hipMemcpy2DAsync(ctx->st_aux.p, (size_t)cw * 4, (const uint8_t*)in.p + ((size_t)cy0 * w + cx0) * 4,
                                          (size_t)w * 4, (size_t)cw * 4, ch, hipMemcpyDeviceToDevice, ctx->stream));
            // blurred crop -> st_aux2 (the H pass reads its source while the V pass writes the destination)
            PFX_TRY(pfx_gaussian_blur_dev(ctx, ctx->st_aux.p, ctx->st_aux2.p, cw, ch, sigma, nullptr));
            // paste the blurred crop over a copy of the source, then select by mask
            PFX_HIP(ctx, hipMemcpyAsync(out.p, in.p, img_bytes(w, h), hipMemcpyDeviceToDevice, ctx->stream));
            PFX_HIP(ctx, hipMemcpy2DAsync((uint8_t*)out.p + ((size_t)cy0 * w + cx0) * 4, (size_t)w * 4, ctx->st_aux2.p,
                                          (size_t)cw * 4, (size_t)cw * 4, ch, hipMemcpyDeviceToDevice, ctx->stream));
            PFX_HIP(ctx, pfxk_select_by_mask(ctx->stream, (const uint8_t*)in.p, (const uint8_t*)out.p,
                                             (const uint8_t*)d_mask, (uint8_t*)out.p, w, h));
        }
    }
    return PFX_OK;
}

extern "C" {

int pfx_blur_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float sigma)
{
    return pfx_gaussian_blur_core(ctx, src, dst, w, h, sigma, nullptr);
}

int pfx_box_blur_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float radius, const uint8_t* mask)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_box_blur_core"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_box_blur_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, radius, d_mask, nullptr));
    return finish_out(ctx, dst, w, h);
}

int pfx_median_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t radius, const uint8_t* mask)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_median_core"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_median_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, radius, d_mask));
    return finish_out(ctx, dst, w, h);
}

int pfx_median_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t radius)
{
    return pfx_median_core(ctx, src, dst, w, h, radius, nullptr);
}

int pfx_pixelate_core(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, uint32_t block_size, const uint8_t* mask)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_pixelate_core"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_pixelate_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, block_size, d_mask));
    return finish_out(ctx, dst, w, h);
}

int pfx_adjust(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, int op, const float* params,
               uint32_t n_params, const uint8_t* lut, const uint8_t* mask, int sparse_mode)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_adjust"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_adjust_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, op, params, n_params, lut, d_mask, sparse_mode));
    return finish_out(ctx, dst, w, h);
}

int pfx_brightness_contrast_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float brightness, float contrast)
{
    const float p[2] = {brightness, contrast};
    return pfx_adjust(ctx, src, dst, w, h, PFX_OP_BRIGHTNESS_CONTRAST, p, 2, nullptr, nullptr, PFX_DENSE);
}

int pfx_hsl_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, float hue, float sat, float light)
{
    const float p[3] = {hue, sat, light};
    return pfx_adjust(ctx, src, dst, w, h, PFX_OP_HSL, p, 3, nullptr, nullptr, PFX_DENSE);
}

int pfx_invert_rgba(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h)
{
    return pfx_adjust(ctx, src, dst, w, h, PFX_OP_INVERT, nullptr, 0, nullptr, nullptr, PFX_DENSE);
}

int pfx_rhai_adjust(pfx_ctx* ctx, uint8_t* pixels_inout, uint32_t w, uint32_t h, int op, const float* params, uint32_t n_params)
{
    PFX_TRY(check_img(ctx, pixels_inout, pixels_inout, w, h, "pfx_rhai_adjust"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, pixels_inout, nullptr, w, h, &d_mask));
    PFX_TRY(pfx_rhai_adjust_dev(ctx, ctx->st_in.p, w, h, op, params, n_params));
    PFX_TRY(pfx_d2h(ctx, pixels_inout, ctx->st_in.p, img_bytes(w, h)));
    return pfx_sync(ctx);
}

int pfx_auto_levels(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h, const uint8_t* mask)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_auto_levels"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, mask, w, h, &d_mask));
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 64));
    {
        pfx_timer t(ctx, "minmax");
        PFX_HIP(ctx, pfxk_minmax_rgb(ctx->stream, (const uint8_t*)ctx->st_in.p, (const uint8_t*)d_mask, w, h, (uint32_t*)ctx->d_misc.p));
    }
    uint32_t mm[6];
    PFX_TRY(pfx_d2h(ctx, mm, ctx->d_misc.p, sizeof mm));
    PFX_TRY(pfx_sync(ctx));
    uint8_t luts[1024];
    for (int c = 0; c < 3; ++c) pfx_build_stretch_lut((uint8_t)mm[c * 2], (uint8_t)mm[c * 2 + 1], luts + c * 256);
    for (int i = 0; i < 256; ++i) luts[768 + i] = (uint8_t)i;
    // auto_levels writes through from_rgba_image (adjustments.rs:230-232) => FROM_FLAT
    PFX_TRY(pfx_adjust_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h, PFX_OP_LUT_RGBA, nullptr, 0, luts, d_mask, PFX_FROM_FLAT));
    return finish_out(ctx, dst, w, h);
}

int pfx_composite_region(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers, uint32_t x,
                         uint32_t y, uint32_t rw, uint32_t rh, uint8_t* dst_region)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, dst_region && pfx_dims_ok(w, h) && pfx_rect_inside(x, y, rw, rh, w, h), "pfx_composite: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, img_bytes(w, h)));
    if (rw == w && rh == h) {
        PFX_TRY(flatten_common(ctx, nullptr, nullptr, layers, n_layers, w, h, true, ctx->st_out.p));
        return finish_out(ctx, dst_region, w, h);
    }
    // dirty rectangle (composite_dirty_readback, renderer.rs:588): only its pixels are composited, into a compact rw x rh image
    const pfxk_region rg{x, y, rw, rh};
    PFX_TRY(flatten_common(ctx, nullptr, nullptr, layers, n_layers, w, h, true, ctx->st_out.p, nullptr, &rg));
    PFX_TRY(pfx_d2h(ctx, dst_region, ctx->st_out.p, (size_t)rw * rh * 4));
    return pfx_sync(ctx);
}

int pfx_composite(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers, uint8_t* dst)
{
    return pfx_composite_region(ctx, w, h, layers, n_layers, 0, 0, w, h, dst);
}

// composite with the tool preview layer folded into the active layer (canvas_state.rs:541-548,593-658)
int pfx_composite_preview(pfx_ctx* ctx, uint32_t w, uint32_t h, const pfx_layer_info* layers, uint32_t n_layers, const uint8_t* preview_pixels,
                          const uint8_t* preview_chunk_present, const pfx_preview* preview, uint8_t* dst)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, dst && pfx_dims_ok(w, h), "pfx_composite_preview: bad arguments");
    if (!preview_pixels) return pfx_composite(ctx, w, h, layers, n_layers, dst);
    PFX_REQUIRE(ctx, preview != nullptr, "pfx_composite_preview: null preview description");
    PFX_TRY(pfx_use(ctx));
    const size_t nchunks = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, img_bytes(w, h)));
    PFX_TRY(pfx_reserve(ctx, ctx->fx_b, img_bytes(w, h) + nchunks));
    PFX_TRY(pfx_h2d(ctx, ctx->fx_b.p, preview_pixels, img_bytes(w, h)));
    preview_arg pv;
    pv.d_pixels = ctx->fx_b.p;
    if (preview_chunk_present) {
        PFX_TRY(pfx_h2d(ctx, (uint8_t*)ctx->fx_b.p + img_bytes(w, h), preview_chunk_present, nchunks));
        pv.d_chunk_present = (const uint8_t*)ctx->fx_b.p + img_bytes(w, h);
    }
    pv.info = *preview;
    PFX_TRY(flatten_common(ctx, nullptr, nullptr, layers, n_layers, w, h, true, ctx->st_out.p, &pv));
    return finish_out(ctx, dst, w, h);
}

int pfx_flatten_preview_dev(pfx_ctx* ctx, const void* const* layer_ptrs_dev, const void* const* mask_ptrs_dev, const pfx_layer_info* layers,
                            uint32_t n_layers, uint32_t w, uint32_t h, const void* preview_dev, const void* preview_chunk_present_dev,
                            const pfx_preview* preview, void* dst_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, dst_dev && pfx_dims_ok(w, h) && (n_layers == 0 || layer_ptrs_dev) && (!preview_dev || preview), "pfx_flatten_preview_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    preview_arg pv;
    pv.d_pixels = preview_dev;
    pv.d_chunk_present = preview_chunk_present_dev;
    if (preview) pv.info = *preview;
    return flatten_common(ctx, layer_ptrs_dev, mask_ptrs_dev, layers, n_layers, w, h, false, dst_dev, preview_dev ? &pv : nullptr);
}

int pfx_blend_pixels(pfx_ctx* ctx, const uint8_t* base, const uint8_t* top, uint8_t* dst, size_t n_pixels, uint8_t blend_mode, float opacity)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, base && top && dst && n_pixels, "pfx_blend_pixels: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, n_pixels * 4));
    PFX_TRY(pfx_reserve(ctx, ctx->st_aux, n_pixels * 4));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, n_pixels * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, base, n_pixels * 4));
    PFX_TRY(pfx_h2d(ctx, ctx->st_aux.p, top, n_pixels * 4));
    PFX_HIP(ctx, pfxk_blend_arrays(ctx->stream, (const uint8_t*)ctx->st_in.p, (const uint8_t*)ctx->st_aux.p, (uint8_t*)ctx->st_out.p,
                                   n_pixels, blend_mode > 24 ? 0u : blend_mode, opacity, opacity_allows_fast_div(opacity) ? 1 : 0));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, n_pixels * 4));
    return pfx_sync(ctx);
}

int pfx_warp_displacement(pfx_ctx* ctx, const uint8_t* src, uint32_t sw, uint32_t sh, const float* disp_xy, uint32_t w, uint32_t h, uint8_t* dst)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_warp_displacement"));
    PFX_REQUIRE(ctx, disp_xy && pfx_dims_ok(sw, sh), "pfx_warp_displacement: bad arguments");
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, img_bytes(sw, sh)));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, img_bytes(w, h)));
    PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)w * h * 8));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, src, img_bytes(sw, sh)));
    PFX_TRY(pfx_h2d(ctx, ctx->st_tmp.p, disp_xy, (size_t)w * h * 8));
    PFX_TRY(pfx_warp_displacement_dev(ctx, ctx->st_in.p, sw, sh, ctx->st_tmp.p, w, h, ctx->st_out.p));
    return finish_out(ctx, dst, w, h);
}

// GpuLiquifyPipeline keeps the source texture across warp_into calls until invalidate_source (ref: src/gpu/compute/liquify.rs:166-176):
// an interactive Liquify session re-sends only the displacement field
int pfx_warp_set_source(pfx_ctx* ctx, const uint8_t* src, uint32_t sw, uint32_t sh)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, src && pfx_dims_ok(sw, sh), "pfx_warp_set_source: bad arguments");
    PFX_TRY(pfx_use(ctx));
    ctx->warp_src_w = ctx->warp_src_h = 0;
    PFX_TRY(pfx_reserve(ctx, ctx->warp_src, img_bytes(sw, sh)));
    PFX_TRY(pfx_h2d(ctx, ctx->warp_src.p, src, img_bytes(sw, sh)));
    PFX_TRY(pfx_sync(ctx)); // the host buffer may be released after the call
    ctx->warp_src_w = sw; ctx->warp_src_h = sh;
    return PFX_OK;
}

int pfx_warp_invalidate_source(pfx_ctx* ctx)
{
    if (!ctx) return PFX_ERR_INVALID;
    ctx->warp_src_w = ctx->warp_src_h = 0;
    return PFX_OK;
}

int pfx_warp_displacement_cached(pfx_ctx* ctx, const float* disp_xy, uint32_t w, uint32_t h, uint8_t* dst)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, disp_xy && dst && pfx_dims_ok(w, h), "pfx_warp_displacement_cached: bad arguments");
    PFX_REQUIRE(ctx, ctx->warp_src_w != 0, "pfx_warp_displacement_cached: no source (pfx_warp_set_source, or it was invalidated)");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, img_bytes(w, h)));
    PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)w * h * 8));
    PFX_TRY(pfx_h2d(ctx, ctx->st_tmp.p, disp_xy, (size_t)w * h * 8));
    PFX_TRY(pfx_warp_displacement_dev(ctx, ctx->warp_src.p, ctx->warp_src_w, ctx->warp_src_h, ctx->st_tmp.p, w, h, ctx->st_out.p));
    return finish_out(ctx, dst, w, h);
}

int pfx_mesh_displacement(pfx_ctx* ctx, const float* orig_pts_xy, const float* deformed_pts_xy, uint32_t cols, uint32_t rows,
                          uint32_t w, uint32_t h, float* disp_xy_out)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, disp_xy_out && pfx_dims_ok(w, h), "pfx_mesh_displacement: bad arguments");
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->st_tmp, (size_t)w * h * 8));
    PFX_TRY(pfx_mesh_displacement_dev(ctx, orig_pts_xy, deformed_pts_xy, cols, rows, w, h, ctx->st_tmp.p));
    PFX_TRY(pfx_d2h(ctx, disp_xy_out, ctx->st_tmp.p, (size_t)w * h * 8));
    return pfx_sync(ctx);
}

int pfx_warp_mesh_catmull_rom(pfx_ctx* ctx, const uint8_t* src, const float* orig_pts_xy, const float* deformed_pts_xy,
                              uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint8_t* dst)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_warp_mesh_catmull_rom"));
    PFX_REQUIRE(ctx, orig_pts_xy != nullptr, "warp_mesh_catmull_rom needs the original grid (transform.rs:1743)");
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, nullptr, w, h, &d_mask));
    PFX_TRY(pfx_warp_mesh_catmull_rom_dev(ctx, ctx->st_in.p, orig_pts_xy, deformed_pts_xy, cols, rows, w, h, ctx->st_out.p));
    return finish_out(ctx, dst, w, h);
}

int pfx_brush_stamps_ex(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn,
                        const float* points_xy, uint32_t n_points, const uint8_t* selection)
{
    PFX_TRY(check_img(ctx, target_inout, target_inout, w, h, "pfx_brush_stamps"));
    const void* d_sel;
    PFX_TRY(stage_in(ctx, target_inout, selection, w, h, &d_sel));
    PFX_TRY(pfx_brush_stamps_ex_dev(ctx, ctx->st_in.p, w, h, brush, dyn, points_xy, n_points, d_sel));
    PFX_TRY(pfx_d2h(ctx, target_inout, ctx->st_in.p, img_bytes(w, h)));
    return pfx_sync(ctx);
}

int pfx_brush_stamps(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, const float* points_xy,
                     uint32_t n_points, const uint8_t* selection)
{
    return pfx_brush_stamps_ex(ctx, target_inout, w, h, brush, nullptr, points_xy, n_points, selection);
}

int pfx_brush_line_ex(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, const pfx_brush_dynamics* dyn, float x0,
                      float y0, float x1, float y1, const uint8_t* selection)
{
    PFX_TRY(check_img(ctx, target_inout, target_inout, w, h, "pfx_brush_line"));   // before the "no stamps" early return: bad arguments are bad whatever the line
    PFX_REQUIRE(ctx, brush != nullptr, "pfx_brush_line: null brush");
    std::vector<float> pts;
    pfx_host_line_points(x0, y0, x1, y1, w, h, pts);
    if (pts.empty()) return PFX_OK;
    return pfx_brush_stamps_ex(ctx, target_inout, w, h, brush, dyn, pts.data(), (uint32_t)(pts.size() / 2), selection);
}

int pfx_brush_line(pfx_ctx* ctx, uint8_t* target_inout, uint32_t w, uint32_t h, const pfx_brush* brush, float x0, float y0,
                   float x1, float y1, const uint8_t* selection)
{
    return pfx_brush_line_ex(ctx, target_inout, w, h, brush, nullptr, x0, y0, x1, y1, selection);
}

// rebuild_tip_mask (brush_render.rs:404-528), host side like the reference
uint32_t pfx_brush_tip_rescale(const uint8_t* src, uint32_t src_size, float brush_size, float hardness, uint8_t* out)
{
    if (!src || !out || src_size == 0) return 0;
    auto u32 = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v); };
    auto u8 = [](float v) -> uint8_t { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)(int)v); };
    const uint32_t dst = std::max(u32(ceilf(brush_size)), 1u);
    const float scale = (float)src_size / (float)dst;
    for (uint32_t dy = 0; dy < dst; ++dy)
        for (uint32_t dx = 0; dx < dst; ++dx) {
            const float sx = (float)dx * scale, sy = (float)dy * scale;
            const uint32_t sx0 = u32(floorf(sx)), sy0 = u32(floorf(sy));
            const uint32_t sx1 = std::min(sx0 + 1, src_size - 1), sy1 = std::min(sy0 + 1, src_size - 1);
            const float fx = sx - (float)sx0, fy = sy - (float)sy0;
            const float v00 = src[sy0 * src_size + sx0], v10 = src[sy0 * src_size + sx1], v01 = src[sy1 * src_size + sx0], v11 = src[sy1 * src_size + sx1];
            const float top = v00 * (1.0f - fx) + v10 * fx, bot = v01 * (1.0f - fx) + v11 * fx;
            out[dy * dst + dx] = u8(fminf(roundf(top * (1.0f - fy) + bot * fy), 255.0f));
        }
    if (hardness < 0.99f) { // hardness as a contrast modifier
        const float threshold = (1.0f - hardness) * 0.6f, range = 1.0f - threshold;
        for (uint32_t i = 0; i < dst * dst; ++i)
            out[i] = u8(roundf(pfx_clampf(((float)out[i] / 255.0f - threshold) / range, 0.0f, 1.0f) * 255.0f));
    }
    if (dst < src_size && dst >= 3) { // anti-alias box passes when downscaling
        const float ratio = (float)src_size / (float)dst;
        const int passes = ratio > 4.0f ? 2 : (ratio > 1.5f ? 1 : 0);
        std::vector<uint8_t> tmp((size_t)dst * dst);
        for (int p = 0; p < passes; ++p) {
            for (uint32_t y = 0; y < dst; ++y)
                for (uint32_t x = 0; x < dst; ++x) {
                    uint32_t sum = out[y * dst + x], count = 1;
                    if (x > 0) { sum += out[y * dst + x - 1]; ++count; }
                    if (x + 1 < dst) { sum += out[y * dst + x + 1]; ++count; }
                    tmp[y * dst + x] = (uint8_t)(sum / count);
                }
            for (uint32_t y = 0; y < dst; ++y)
                for (uint32_t x = 0; x < dst; ++x) {
                    uint32_t sum = tmp[y * dst + x], count = 1;
                    if (y > 0) { sum += tmp[(y - 1) * dst + x]; ++count; }
                    if (y + 1 < dst) { sum += tmp[(y + 1) * dst + x]; ++count; }
                    out[y * dst + x] = (uint8_t)(sum / count);
                }
        }
    }
    return dst;
}

int pfx_brush_commit(pfx_ctx* ctx, uint8_t* layer_inout, const uint8_t* preview, uint32_t w, uint32_t h, uint8_t blend_mode,
                     int is_eraser, const uint8_t* selection)
{
    PFX_TRY(check_img(ctx, layer_inout, layer_inout, w, h, "pfx_brush_commit"));
    PFX_REQUIRE(ctx, preview != nullptr, "null preview");
    const void* d_sel;
    PFX_TRY(stage_in(ctx, layer_inout, selection, w, h, &d_sel));
    PFX_TRY(pfx_h2d(ctx, ctx->st_out.p, preview, img_bytes(w, h)));
    PFX_HIP(ctx, pfxk_brush_commit(ctx->stream, (uint8_t*)ctx->st_in.p, (const uint8_t*)ctx->st_out.p, (const uint8_t*)d_sel, w, h,
                                   blend_mode > 24 ? 0u : blend_mode, is_eraser));
    PFX_TRY(pfx_d2h(ctx, layer_inout, ctx->st_in.p, img_bytes(w, h)));
    return pfx_sync(ctx);
}

int pfx_tiled_roundtrip(pfx_ctx* ctx, const uint8_t* src, uint8_t* dst, uint32_t w, uint32_t h)
{
    PFX_TRY(check_img(ctx, src, dst, w, h, "pfx_tiled_roundtrip"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, nullptr, w, h, &d_mask));
    PFX_TRY(pfx_tiled_roundtrip_dev(ctx, ctx->st_in.p, ctx->st_out.p, w, h));
    return finish_out(ctx, dst, w, h);
}

int pfx_chunk_populated(pfx_ctx* ctx, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* populated)
{
    PFX_TRY(check_img(ctx, src, populated, w, h, "pfx_chunk_populated"));
    const void* d_mask;
    PFX_TRY(stage_in(ctx, src, nullptr, w, h, &d_mask));
    const size_t nchunks = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    PFX_TRY(pfx_reserve(ctx, ctx->d_chunks, nchunks));
    PFX_HIP(ctx, pfxk_chunk_populated(ctx->stream, (const uint8_t*)ctx->st_in.p, w, h, (uint8_t*)ctx->d_chunks.p));
    PFX_TRY(pfx_d2h(ctx, populated, ctx->d_chunks.p, nchunks));
    return pfx_sync(ctx);
}

int pfx_tune(pfx_ctx* ctx, const char* key, int value)
{
    if (!ctx || !key) return PFX_ERR_INVALID;
    if (std::strcmp(key, "gauss_v_cfg") == 0) { pfxk_gauss_set_v_config(value); return PFX_OK; }
    if (std::strcmp(key, "gauss_mfma_segments") == 0) { pfxk_gauss_set_mfma_segments(value); return PFX_OK; }
    if (std::strcmp(key, "gauss_parts") == 0) { pfxk_gauss_set_mfma_parts(value / 10, value % 10); return PFX_OK; } // f16 pieces per weight, per horizontal result: 22, 12, 11
    if (std::strcmp(key, "flatten_variant") == 0) { pfxk_flatten_set_variant(value); return PFX_OK; }
    if (std::strcmp(key, "dle_units") == 0) { pfxk_flatten_set_dle(value, -1); return PFX_OK; }
    if (std::strcmp(key, "dle_ring") == 0) { pfxk_flatten_set_dle(-1, value); return PFX_OK; }
    if (std::strcmp(key, "dle_stats") == 0) { pfxk_flatten_set_dle_dev(value, -1); return PFX_OK; }
    if (std::strcmp(key, "chunk_start") == 0) { ctx->use_chunk_start = value != 0; return PFX_OK; }
    if (std::strcmp(key, "dle_min_layers") == 0) { ctx->dle_min_layers = value; return PFX_OK; }
    if (std::strcmp(key, "dle_sched") == 0) { pfxk_flatten_set_dle_sched(value, -1, -1); return PFX_OK; }
    if (std::strcmp(key, "dle_frac_a") == 0) { pfxk_flatten_set_dle_sched(-1, value, -1); return PFX_OK; }
    if (std::strcmp(key, "dle_frac_b") == 0) { pfxk_flatten_set_dle_sched(-1, -1, value); return PFX_OK; }
    if (std::strcmp(key, "dle_cfg") == 0) { pfxk_flatten_set_dle_dev(-1, value); return PFX_OK; }
    if (std::strcmp(key, "dle_kernel") == 0) { pfxk_flatten_set_dle_plan(value, -2, -2); return PFX_OK; }   // 0 = class sorting, 1 = round-3 kernel
    if (std::strcmp(key, "dle_s1") == 0) { pfxk_flatten_set_dle_plan(-1, value, -2); return PFX_OK; }       // re-deal attempts: first one this many layers above the topmost candidate (-1: 1, 0: never)
    if (std::strcmp(key, "dle_s2") == 0) { pfxk_flatten_set_dle_plan(-1, -2, value); return PFX_OK; }       // ... then every this many layers (-1: 3, 0: only the first)
    if (std::strcmp(key, "median_search1") == 0) { pfxk_median_set_search1(value); return PFX_OK; }
    if (std::strcmp(key, "outline_bits") == 0) { ctx->outline_bits = value != 0; return PFX_OK; }
    if (std::strcmp(key, "brush_binning") == 0) { ctx->brush_binning = value != 0; return PFX_OK; }   // 0: long strokes on the bounding-box kernel too
    if (std::strcmp(key, "median_xlane") == 0) { pfxk_median_set_xlane(value); return PFX_OK; } // 0: radius 2 on the per-lane shared-column network (round 3's kernel)
    if (std::strcmp(key, "median_pair") == 0) { pfxk_median_bits_set_pair(value); return PFX_OK; } // 0: the single-column bit-plane kernel for every radius
    if (std::strcmp(key, "median_bits_min") == 0) { ctx->median_bits_min = value; return PFX_OK; } // smallest radius on the bit-plane kernel (8: never)
    if (std::strcmp(key, "median_single") == 0) { pfxk_median_set_single(value); return PFX_OK; }
    if (std::strcmp(key, "box_prefix_from") == 0) { pfxk_box_set_prefix_from(value); return PFX_OK; }
    if (std::strcmp(key, "box_px") == 0) { pfxk_box_set_force(value, -1); return PFX_OK; }
    if (std::strcmp(key, "box_py") == 0) { pfxk_box_set_force(-1, value); return PFX_OK; }
    if (std::strcmp(key, "box_px_switch") == 0) { pfxk_box_set_switch(value, -1); return PFX_OK; }
    if (std::strcmp(key, "box_py_switch") == 0) { pfxk_box_set_switch(-1, value); return PFX_OK; }
    if (std::strcmp(key, "dle_adaptive") == 0) { ctx->dle_adaptive = value != 0; ctx->dle_probe_state = 0; return PFX_OK; }   // 0 = shallow stacks never take the elimination kernel
    if (std::strcmp(key, "gauss_fused_exact") == 0) { pfxk_gauss_set_fused_exact(value); return PFX_OK; }  // 0 = the bit-exact Gaussian always through the two kernels
    if (std::strcmp(key, "mesh_xcd") == 0) { pfxk_warp_set_mesh_xcd(value); return PFX_OK; }            // 0 = the fused mesh warp's plain 2-D tile order
    if (std::strcmp(key, "box_strip") == 0) { pfxk_box_set_strip(value, 0, -1); return PFX_OK; }          // 0 = radii >= 5 through the two-pass kernels
    if (std::strcmp(key, "box_strip_fill") == 0) { pfxk_box_set_strip(-1, value, -1); return PFX_OK; }
    if (std::strcmp(key, "box_strip_nseg") == 0) { pfxk_box_set_strip(-1, 0, value); return PFX_OK; }
    if (std::strcmp(key, "box_two_pass") == 0) { pfxk_box_set_two_pass(value); return PFX_OK; }
    if (std::strcmp(key, "shadow_plane") == 0) { ctx->shadow_plane_blur = value != 0; return PFX_OK; } // drop shadow: blur the alpha plane (1) or the RGBA expansion (0); identical results
    if (std::strcmp(key, "gauss_cols64") == 0) { pfxk_gauss_set_mfma_cols64(value); return PFX_OK; } // matrix-core Gaussian at 8 K blocks: 64-column (1) / 32-column (0) strips, same bits
    if (std::strcmp(key, "chain_fuse_heavy") == 0) { ctx->chain_fuse_heavy = value != 0; return PFX_OK; } // pfx_chain_dev: HSL / vibrance in a Gaussian's store too (measured: no gain)
    if (std::strcmp(key, "chain_mfma") == 0) { ctx->chain_mfma_epilogue = value != 0; return PFX_OK; } // pfx_chain_dev: the chain in the matrix-core Gaussian's store (1) or as its own launch (0)
    if (std::strcmp(key, "gauss_fast_effects") == 0) { ctx->gauss_fast_effects = value != 0; return PFX_OK; } // sharpen / glow / shadow on the default-mode Gaussian (+-amount LSB)
    if (std::strcmp(key, "resize_two_pass") == 0) { ctx->resize_two_pass = value != 0; return PFX_OK; } // A/B and the parity test of the fused kernel
    return pfx_fail(ctx, PFX_ERR_INVALID, "pfx_tune: unknown key %s", key);
}

int pfx_flatten_stats(pfx_ctx* ctx, uint64_t out[8], int reset)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, out != nullptr, "pfx_flatten_stats: null output");
    PFX_TRY(pfx_sync(ctx));
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "counter width");
    unsigned long long all[16];
    PFX_HIP(ctx, pfxk_flatten_dle_stats(all, reset));
    for (int i = 0; i < 8; ++i) out[i] = all[i];
    return PFX_OK;
}

int pfx_flatten_trace(pfx_ctx* ctx, uint64_t out[16], int reset)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, out != nullptr, "pfx_flatten_trace: null output");
    PFX_TRY(pfx_sync(ctx));
    PFX_HIP(ctx, pfxk_flatten_dle_stats((unsigned long long*)out, reset));
    return PFX_OK;
}

int pfx_selftest_division(pfx_ctx* ctx, uint64_t seed, uint32_t n_millions, uint64_t* mismatches)
{
    if (!ctx || !mismatches) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 64));
    PFX_HIP(ctx, hipMemsetAsync(ctx->d_misc.p, 0, 8, ctx->stream));
    const uint32_t blocks = 1024, iters = (uint32_t)(((uint64_t)n_millions * 1000000ull + blocks * 256ull - 1) / (blocks * 256ull));
    PFX_HIP(ctx, pfxk_rdiv_check(ctx->stream, seed, blocks, iters, (unsigned long long*)ctx->d_misc.p));
    unsigned long long bad = 0;
    PFX_TRY(pfx_d2h(ctx, &bad, ctx->d_misc.p, 8));
    PFX_TRY(pfx_sync(ctx));
    *mismatches = bad;
    return PFX_OK;
}

int pfx_selftest_unorm_store(pfx_ctx* ctx, uint64_t* mismatches)
{
    if (!ctx || !mismatches) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 2048));
    PFX_HIP(ctx, hipMemsetAsync(ctx->d_misc.p, 0, 2048, ctx->stream));
    PFX_HIP(ctx, pfxk_unorm_store_check(ctx->stream, (uint8_t*)ctx->d_misc.p, (unsigned long long*)((uint8_t*)ctx->d_misc.p + 1024)));
    unsigned long long bad = 0;
    PFX_TRY(pfx_d2h(ctx, &bad, (uint8_t*)ctx->d_misc.p + 1024, sizeof bad));
    PFX_TRY(pfx_sync(ctx));
    *mismatches = bad;
    return PFX_OK;
}

int pfx_selftest_round_pack(pfx_ctx* ctx, uint64_t* mismatches, uint64_t* signalling_nan_mismatches)
{
    if (!ctx || !mismatches) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 64));
    PFX_HIP(ctx, hipMemsetAsync(ctx->d_misc.p, 0, 16, ctx->stream));
    PFX_HIP(ctx, pfxk_round_pack_check(ctx->stream, (unsigned long long*)ctx->d_misc.p));
    unsigned long long bad[2] = {0, 0};
    PFX_TRY(pfx_d2h(ctx, bad, ctx->d_misc.p, 16));
    PFX_TRY(pfx_sync(ctx));
    *mismatches = bad[0];
    if (signalling_nan_mismatches) *signalling_nan_mismatches = bad[1];
    return PFX_OK;
}

} // extern "C"
