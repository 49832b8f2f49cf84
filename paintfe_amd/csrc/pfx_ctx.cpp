// pfx_ctx.cpp — context lifetime, device memory plumbing, layer store, timing.
// Mirrors GpuRenderer's resource handling: one device + queue per renderer (ref: src/gpu/context.rs:9-15),
// cached staging buffers (ref: src/gpu/renderer.rs:232-236), per-layer device images keyed by index and versioned
// by `generation` (ref: src/gpu/renderer.rs:324-460).
#include <cstdarg>
#include <cstdio>

#include "pfx_internal.h"
#include "pfx_kernels.h"

static thread_local std::string g_no_ctx_error;

int pfx_fail(pfx_ctx* ctx, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_no_ctx_error = buf;
    return status;
}

int pfx_use(pfx_ctx* ctx)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_HIP(ctx, hipSetDevice(ctx->device));
    return PFX_OK;
}

int pfx_reserve(pfx_ctx* ctx, pfx_devbuf& b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return PFX_OK;
    if (b.p) {
        PFX_HIP(ctx, hipStreamSynchronize(ctx->stream)); // nothing in flight may still use the old block
        PFX_HIP(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    if (bytes == 0) bytes = 256;
    PFX_HIP(ctx, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return PFX_OK;
}

int pfx_h2d(pfx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return PFX_OK;
    PFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return PFX_OK;
}

int pfx_d2h(pfx_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return PFX_OK;
    PFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return PFX_OK;
}

int pfx_sync(pfx_ctx* ctx)
{
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PFX_OK;
}

pfx_timer::pfx_timer(pfx_ctx* c, const char* name) : ctx(c), on(c && c->timing)
{
    if (!on) return;
    rec.name = name;
    if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(rec.start, ctx->stream);
}
pfx_timer::~pfx_timer()
{
    if (!on) return;
    (void)hipEventRecord(rec.stop, ctx->stream);
    ctx->timings.push_back(rec);
}

static void free_buf(pfx_devbuf& b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

extern "C" {

int pfx_abi_version(void) { return PFX_ABI_VERSION; }

int pfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pfx_ctx_create(int device, pfx_ctx** out)
{
    if (!out) return PFX_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return pfx_fail(nullptr, PFX_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= n) return pfx_fail(nullptr, PFX_ERR_INVALID, "device %d out of range (%d)", device, n);
    pfx_ctx* ctx = new (std::nothrow) pfx_ctx();
    if (!ctx) return PFX_ERR_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return pfx_fail(nullptr, PFX_ERR_HIP, "could not create a stream on device %d", device);
    }
    ctx->stream = ctx->own_stream;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) ctx->n_cus = cus;
    *out = ctx;
    return PFX_OK;
}

void pfx_ctx_destroy(pfx_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& t : ctx->timings) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto& kv : ctx->layers) { free_buf(kv.second.pixels); free_buf(kv.second.mask); free_buf(kv.second.chunk_flags); }
    pfx_devbuf* bufs[] = {&ctx->st_in, &ctx->st_out, &ctx->st_mask, &ctx->st_tmp, &ctx->st_aux, &ctx->st_aux2, &ctx->fx_a, &ctx->fx_b, &ctx->d_desc,
                          &ctx->d_adj, &ctx->d_chunks, &ctx->d_chunk_meta, &ctx->d_chunk_start, &ctx->d_wts, &ctx->d_wsplit, &ctx->warp_src, &ctx->d_lut, &ctx->d_pts, &ctx->d_misc, &ctx->st_chain, &ctx->d_chain_luts};
    for (auto* b : bufs) free_buf(*b);
    if (ctx->h_chunk_useful) (void)hipHostFree(ctx->h_chunk_useful);
    if (ctx->ev_chunk_useful) (void)hipEventDestroy(ctx->ev_chunk_useful);
    if (ctx->h_dle_verdict) (void)hipHostFree(ctx->h_dle_verdict);
    if (ctx->ev_dle_probe) (void)hipEventDestroy(ctx->ev_dle_probe);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* pfx_last_error(const pfx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_no_ctx_error.c_str(); }

int pfx_ctx_set_exact(pfx_ctx* ctx, int exact)
{
    if (!ctx) return PFX_ERR_INVALID;
    ctx->exact = exact != 0;
    return PFX_OK;
}

void* pfx_ctx_stream(pfx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int pfx_ctx_set_stream(pfx_ctx* ctx, void* hip_stream, int adopt)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = adopt ? (hipStream_t)hip_stream : ctx->own_stream;
    return PFX_OK;
}

int pfx_ctx_synchronize(pfx_ctx* ctx)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    return pfx_sync(ctx);
}

int pfx_dev_alloc(pfx_ctx* ctx, size_t bytes, void** out_dev)
{
    if (!ctx || !out_dev) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    *out_dev = nullptr;
    PFX_HIP(ctx, hipMalloc(out_dev, bytes ? bytes : 256));
    return PFX_OK;
}

int pfx_host_alloc(pfx_ctx* ctx, size_t bytes, void** out_host)
{
    if (!ctx || !out_host) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    *out_host = nullptr;
    PFX_HIP(ctx, hipHostMalloc(out_host, bytes ? bytes : 256, hipHostMallocPortable));   // portable: every device of a pfx_group may copy from / to it
    return PFX_OK;
}

int pfx_host_free(pfx_ctx* ctx, void* host)
{
    if (!ctx) return PFX_ERR_INVALID;
    if (!host) return PFX_OK;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // nothing in flight may still read or write it
    PFX_HIP(ctx, hipHostFree(host));
    return PFX_OK;
}

int pfx_dev_free(pfx_ctx* ctx, void* dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    if (!dev) return PFX_OK;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    PFX_HIP(ctx, hipFree(dev));
    return PFX_OK;
}

int pfx_dev_upload(pfx_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes)
{
    if (!ctx || (bytes && (!dst_dev || !src_host))) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_h2d(ctx, dst_dev, src_host, bytes));
    return pfx_sync(ctx);
}

int pfx_dev_download(pfx_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes)
{
    if (!ctx || (bytes && (!dst_host || !src_dev))) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_d2h(ctx, dst_host, src_dev, bytes));
    return pfx_sync(ctx);
}

int pfx_dev_memset(pfx_ctx* ctx, void* dst_dev, int value, size_t bytes)
{
    if (!ctx || (bytes && !dst_dev)) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipMemsetAsync(dst_dev, value, bytes, ctx->stream));
    return PFX_OK;
}

// ---------------------------------------------------------------- layer store (B2)
int pfx_layer_upload(pfx_ctx* ctx, uint32_t idx, uint32_t w, uint32_t h, const uint8_t* rgba, uint64_t generation)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, rgba && pfx_dims_ok(w, h), "pfx_layer_upload: null data, zero size or more than 256 Mpx");
    PFX_REQUIRE(ctx, idx < 65536u, "pfx_layer_upload: layer index too large");
    PFX_TRY(pfx_use(ctx));
    auto it = ctx->layers.find(idx);
    if (it != ctx->layers.end() && it->second.generation == generation && it->second.w == w && it->second.h == h)
        return PFX_OK; // ref: src/gpu/renderer.rs:336-342 (same generation and size: skip the upload)
    pfx_layer_state& L = ctx->layers[idx];
    const size_t bytes = (size_t)w * h * 4;
    int st = pfx_reserve(ctx, L.pixels, bytes);
    if (st != PFX_OK) { if (!L.pixels.p) ctx->layers.erase(idx); return st; }
    if (L.w != w || L.h != h) L.has_mask = false;
    L.w = w; L.h = h; L.generation = generation;
    ctx->store_epoch++;
    PFX_TRY(pfx_h2d(ctx, L.pixels.p, rgba, bytes));
    // per-chunk alpha summary for the compositor's start table (k_flatten.hip: a tile starts at the topmost layer that covers its chunks)
    const uint32_t cxn = (w + 63u) / 64u, cyn = (h + 63u) / 64u;
    PFX_TRY(pfx_reserve(ctx, L.chunk_flags, (size_t)cxn * cyn));
    PFX_HIP(ctx, pfxk_chunk_alpha_flags(ctx->stream, (const uint8_t*)L.pixels.p, w, h, 0, 0, cxn, cyn, (uint8_t*)L.chunk_flags.p));
    return pfx_sync(ctx);
}

int pfx_layer_update_rect(pfx_ctx* ctx, uint32_t idx, uint32_t x, uint32_t y, uint32_t rw, uint32_t rh, const uint8_t* rgba)
{
    if (!ctx) return PFX_ERR_INVALID;
    auto it = ctx->layers.find(idx);
    PFX_REQUIRE(ctx, it != ctx->layers.end(), "pfx_layer_update_rect: layer not uploaded");
    pfx_layer_state& L = it->second;
    PFX_REQUIRE(ctx, rgba && pfx_rect_inside(x, y, rw, rh, L.w, L.h), "pfx_layer_update_rect: region out of bounds");
    PFX_TRY(pfx_use(ctx));
    ctx->store_epoch++;
    PFX_HIP(ctx, hipMemcpy2DAsync((uint8_t*)L.pixels.p + ((size_t)y * L.w + x) * 4, (size_t)L.w * 4, rgba, (size_t)rw * 4,
                                  (size_t)rw * 4, rh, hipMemcpyHostToDevice, ctx->stream));
    if (L.chunk_flags.p) { // the chunks the rectangle touches get their summary refreshed
        const uint32_t cx0 = x / 64u, cy0 = y / 64u, cx1 = (x + rw - 1u) / 64u, cy1 = (y + rh - 1u) / 64u;
        PFX_HIP(ctx, pfxk_chunk_alpha_flags(ctx->stream, (const uint8_t*)L.pixels.p, L.w, L.h, cx0, cy0, cx1 - cx0 + 1u, cy1 - cy0 + 1u, (uint8_t*)L.chunk_flags.p));
    }
    return pfx_sync(ctx);
}

int pfx_layer_set_mask(pfx_ctx* ctx, uint32_t idx, const uint8_t* conceal)
{
    if (!ctx) return PFX_ERR_INVALID;
    auto it = ctx->layers.find(idx);
    PFX_REQUIRE(ctx, it != ctx->layers.end(), "pfx_layer_set_mask: layer not uploaded");
    pfx_layer_state& L = it->second;
    ctx->store_epoch++;
    if (!conceal) { L.has_mask = false; return PFX_OK; }
    PFX_TRY(pfx_use(ctx));
    PFX_TRY(pfx_reserve(ctx, L.mask, (size_t)L.w * L.h));
    PFX_TRY(pfx_h2d(ctx, L.mask.p, conceal, (size_t)L.w * L.h));
    L.has_mask = true;
    return pfx_sync(ctx);
}

int pfx_layer_remove(pfx_ctx* ctx, uint32_t idx)
{
    if (!ctx) return PFX_ERR_INVALID;
    auto it = ctx->layers.find(idx);
    if (it == ctx->layers.end()) return PFX_OK;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_buf(it->second.pixels);
    free_buf(it->second.mask);
    free_buf(it->second.chunk_flags);
    ctx->layers.erase(it);
    ctx->store_epoch++;
    return PFX_OK;
}

int pfx_layer_clear(pfx_ctx* ctx)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& kv : ctx->layers) { free_buf(kv.second.pixels); free_buf(kv.second.mask); free_buf(kv.second.chunk_flags); }
    ctx->layers.clear();
    ctx->store_epoch++;
    return PFX_OK;
}

uint32_t pfx_layer_count(const pfx_ctx* ctx) { return ctx ? (uint32_t)ctx->layers.size() : 0u; }

size_t pfx_layer_memory(const pfx_ctx* ctx)
{
    size_t n = 0;
    if (ctx) for (auto& kv : ctx->layers) n += (size_t)kv.second.w * kv.second.h * 4;
    return n;
}

// ---------------------------------------------------------------- timing
int pfx_timing_enable(pfx_ctx* ctx, int on)
{
    if (!ctx) return PFX_ERR_INVALID;
    ctx->timing = on != 0;
    return PFX_OK;
}

int pfx_timing_reset(pfx_ctx* ctx)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& t : ctx->timings) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    ctx->timings.clear();
    return PFX_OK;
}

int pfx_timing_read(pfx_ctx* ctx, const char* kernel_name, double* total_ms, uint64_t* launches)
{
    if (!ctx || !kernel_name) return PFX_ERR_INVALID;
    PFX_TRY(pfx_use(ctx));
    PFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double ms = 0.0;
    uint64_t n = 0;
    for (auto& t : ctx->timings) {
        if (t.name != kernel_name) continue;
        float e = 0.f;
        PFX_HIP(ctx, hipEventElapsedTime(&e, t.start, t.stop));
        ms += e;
        ++n;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    return PFX_OK;
}

} // extern "C"
