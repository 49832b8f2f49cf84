// pfx_group.cpp — one document across the GPUs of a node, behind the C ABI (include/pfx.h: pfx_group_*).
//
// The reference has no multi-device layer; its unit of independence is the 64x64 TiledImage chunk (the compositor runs
// `populated_chunks.par_iter()`, ref: src/canvas/canvas_state.rs:565) and, one level up, the file (the CLI loop,
// ref: src/cli.rs:159-216).  SURVEY.md §5 / §8(e) fix the MI355X mapping this file implements, in ONE process driving N HIP
// devices (what a Rust host linking libpfx would do):
//   * a document is cut into bands of whole chunk rows (pfx_band_rows); member k keeps rows [y0_k, y1_k) of EVERY layer resident;
//   * flatten is per-pixel: every member composites its band with no communication;
//   * the Gaussian needs ceil(3 sigma) rows of its INPUT (the flattened u8 band) from the neighbouring bands: they are pulled
//     over xGMI with peer-to-peer copies (hipMemcpyPeerAsync on the consumer's stream, ordered by events behind the producers'
//     flatten), straight into the halo rows around the member's own band; the blur then runs on band + halo and the centre is
//     kept — the 1.47 MB per neighbour and direction at 8K / sigma = 16 that SURVEY §8(e) prices, not the 5.9 MB of f32
//     intermediate rows;
//   * all-gather (optional): every member pushes its result band into every member's full-size image, one peer copy per pair —
//     xGMI is point-to-point, so the N-1 copies of a member travel on N-1 different links at once (no ring).  The pushes run on a
//     copy stream per member, so the next call's flatten overlaps them (only its blur, which rewrites the source, waits).
// Everything is asynchronous on the members' streams; the host thread only enqueues.  Nothing here touches pixels on the CPU.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "pfx_internal.h"

struct pfx_group_member {
    pfx_ctx* ctx = nullptr;
    int device = 0;
    uint32_t y0 = 0, y1 = 0;            // band rows
    uint32_t top = 0, bottom = 0;       // halo rows held around the band for the current radius
    std::vector<void*> layers;          // band of every layer: (y1 - y0) * w * 4 bytes each
    void* padded = nullptr;             // [top halo | flattened band | bottom halo], (rows + 2 * halo_cap) * w * 4
    void* blurred = nullptr;            // same shape: blur of `padded`
    void* gathered = nullptr;           // full w * h * 4 image (all-gather target), allocated on first use
    size_t padded_cap = 0;
    hipEvent_t ev_flat = nullptr, ev_done = nullptr; // flatten finished / result band final (halo pulls and blur done), on the compute stream
    hipStream_t s_copy = nullptr;       // the all-gather's pushes run here, behind ev_done, so that the next call's flatten overlaps them
    hipEvent_t ev_gather = nullptr;     // this member's pushes finished (s_copy)
    bool gather_pending = false;        // ev_gather was recorded and not yet waited for by the compute stream
};

struct pfx_group {
    std::vector<pfx_group_member> m;
    uint32_t w = 0, h = 0, n_layers = 0;
    uint32_t halo_cap = 0;              // halo rows the padded buffers were sized for
    bool have_result = false, result_blurred = false;
    std::string err;
};

namespace {

int gfail(pfx_group* g, int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return status;
}

#define PFXG_HIP(g, call)                                                                                     \
    do {                                                                                                      \
        hipError_t _e = (call);                                                                               \
        if (_e != hipSuccess)                                                                                 \
            return gfail((g), _e == hipErrorOutOfMemory ? PFX_ERR_OOM : PFX_ERR_HIP, "%s failed: %s", #call, \
                         hipGetErrorString(_e));                                                              \
    } while (0)
#define PFXG_CTX(g, mem, call)                                                                     \
    do {                                                                                           \
        int _s = (call);                                                                           \
        if (_s != PFX_OK) return gfail((g), _s, "member on device %d: %s", (mem).device, pfx_last_error((mem).ctx)); \
    } while (0)

hipStream_t stream_of(const pfx_group_member& mem) { return (hipStream_t)pfx_ctx_stream(mem.ctx); }

// device-to-device copy between two members, enqueued on `dst`'s stream
hipError_t copy_between(const pfx_group_member& dst, void* d, const pfx_group_member& src, const void* s, size_t bytes)
{
    if (bytes == 0) return hipSuccess;
    if (dst.device == src.device) return hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, stream_of(dst));
    return hipMemcpyPeerAsync(d, dst.device, s, src.device, bytes, stream_of(dst));
}

void free_member_buffers(pfx_group_member& mem)
{
    (void)hipSetDevice(mem.device);
    for (void* p : mem.layers) if (p) (void)hipFree(p);
    mem.layers.clear();
    if (mem.padded) (void)hipFree(mem.padded);
    if (mem.blurred) (void)hipFree(mem.blurred);
    if (mem.gathered) (void)hipFree(mem.gathered);
    mem.padded = mem.blurred = mem.gathered = nullptr;
    mem.padded_cap = 0;
}

int ensure_padded(pfx_group* g, uint32_t halo)
{
    if (halo <= g->halo_cap && g->m[0].padded_cap != 0) return PFX_OK;
    for (auto& mem : g->m) {
        const size_t rows = (size_t)(mem.y1 - mem.y0) + 2 * (size_t)halo;
        const size_t bytes = std::max<size_t>(rows * g->w * 4, 256);
        PFXG_HIP(g, hipSetDevice(mem.device));
        if (mem.padded) (void)hipFree(mem.padded);
        if (mem.blurred) (void)hipFree(mem.blurred);
        mem.padded = mem.blurred = nullptr;
        PFXG_HIP(g, hipMalloc(&mem.padded, bytes));
        PFXG_HIP(g, hipMalloc(&mem.blurred, bytes));
        mem.padded_cap = bytes;
    }
    g->halo_cap = halo;
    return PFX_OK;
}

} // namespace

extern "C" {

// whole chunk rows per band, remainder spread over the first members; empty bands when there are fewer chunk rows than members
void pfx_band_rows(uint32_t h, uint32_t world, uint32_t rank, uint32_t* y0, uint32_t* y1)
{
    uint32_t a = 0, b = 0;
    if (world != 0 && rank < world) {
        const uint32_t chunk_rows = (h + PFX_CHUNK - 1) / PFX_CHUNK;
        const uint32_t base = chunk_rows / world, rem = chunk_rows % world;
        const uint32_t c0 = rank * base + std::min(rank, rem);
        const uint32_t c1 = c0 + base + (rank < rem ? 1u : 0u);
        a = std::min(c0 * PFX_CHUNK, h);
        b = std::min(c1 * PFX_CHUNK, h);
    }
    if (y0) *y0 = a;
    if (y1) *y1 = b;
}

int pfx_group_create(const int* devices, uint32_t n, pfx_group** out)
{
    if (!out) return PFX_ERR_INVALID;
    *out = nullptr;
    if (!devices || n == 0 || n > 64) return PFX_ERR_INVALID;
    pfx_group* g = new (std::nothrow) pfx_group();
    if (!g) return PFX_ERR_OOM;
    g->m.resize(n);
    for (uint32_t k = 0; k < n; ++k) {
        g->m[k].device = devices[k];
        const int s = pfx_ctx_create(devices[k], &g->m[k].ctx);
        if (s != PFX_OK) { pfx_group_destroy(g); return s; }
        if (hipSetDevice(devices[k]) != hipSuccess || hipEventCreateWithFlags(&g->m[k].ev_flat, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->m[k].ev_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->m[k].ev_gather, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&g->m[k].s_copy, hipStreamNonBlocking) != hipSuccess) {
            pfx_group_destroy(g);
            return PFX_ERR_HIP;
        }
    }
    // direct xGMI access between every pair of distinct devices (idempotent; a refusal only makes the copies staged)
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t b = 0; b < n; ++b) {
            if (devices[a] == devices[b]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                (void)hipSetDevice(devices[a]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                if (e != hipSuccess) (void)hipGetLastError(); // hipErrorPeerAccessAlreadyEnabled et al.
            }
        }
    *out = g;
    return PFX_OK;
}

void pfx_group_destroy(pfx_group* g)
{
    if (!g) return;
    for (auto& mem : g->m) {
        if (mem.ctx) (void)pfx_ctx_synchronize(mem.ctx);
        (void)hipSetDevice(mem.device);
        if (mem.s_copy) { (void)hipStreamSynchronize(mem.s_copy); (void)hipStreamDestroy(mem.s_copy); }
        free_member_buffers(mem);
        if (mem.ev_flat) (void)hipEventDestroy(mem.ev_flat);
        if (mem.ev_done) (void)hipEventDestroy(mem.ev_done);
        if (mem.ev_gather) (void)hipEventDestroy(mem.ev_gather);
        if (mem.ctx) pfx_ctx_destroy(mem.ctx);
    }
    delete g;
}

uint32_t pfx_group_size(const pfx_group* g) { return g ? (uint32_t)g->m.size() : 0u; }
pfx_ctx* pfx_group_ctx(pfx_group* g, uint32_t rank) { return (g && rank < g->m.size()) ? g->m[rank].ctx : nullptr; }
const char* pfx_group_last_error(const pfx_group* g) { return g ? g->err.c_str() : "null group"; }

int pfx_group_set_document(pfx_group* g, uint32_t w, uint32_t h, uint32_t n_layers)
{
    if (!g) return PFX_ERR_INVALID;
    if (w == 0 || h == 0 || n_layers == 0 || n_layers > PFX_MAX_LAYERS) return gfail(g, PFX_ERR_INVALID, "bad document geometry");
    for (auto& mem : g->m) {
        if (mem.ctx) (void)pfx_ctx_synchronize(mem.ctx);
        if (mem.s_copy) { (void)hipSetDevice(mem.device); (void)hipStreamSynchronize(mem.s_copy); }
        mem.gather_pending = false;
        free_member_buffers(mem);
    }
    g->w = w; g->h = h; g->n_layers = n_layers; g->halo_cap = 0; g->have_result = false;
    const uint32_t world = (uint32_t)g->m.size();
    for (uint32_t k = 0; k < world; ++k) {
        auto& mem = g->m[k];
        pfx_band_rows(h, world, k, &mem.y0, &mem.y1);
        mem.top = mem.bottom = 0;
        mem.layers.assign(n_layers, nullptr);
        const size_t bytes = std::max<size_t>((size_t)(mem.y1 - mem.y0) * w * 4, 256);
        PFXG_HIP(g, hipSetDevice(mem.device));
        for (uint32_t l = 0; l < n_layers; ++l) PFXG_HIP(g, hipMalloc(&mem.layers[l], bytes));
    }
    return ensure_padded(g, 0);
}

int pfx_group_upload_layer(pfx_group* g, uint32_t index, const uint8_t* rgba_host)
{
    if (!g || !rgba_host) return PFX_ERR_INVALID;
    if (index >= g->n_layers) return gfail(g, PFX_ERR_INVALID, "layer %u outside the document (%u layers)", index, g->n_layers);
    for (auto& mem : g->m) {
        const size_t bytes = (size_t)(mem.y1 - mem.y0) * g->w * 4;
        if (bytes == 0) continue;
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipMemcpyAsync(mem.layers[index], rgba_host + (size_t)mem.y0 * g->w * 4, bytes, hipMemcpyHostToDevice, stream_of(mem)));
    }
    for (auto& mem : g->m) PFXG_CTX(g, mem, pfx_ctx_synchronize(mem.ctx)); // the host buffer may go away after the call
    return PFX_OK;
}

void* pfx_group_layer_band_dev(pfx_group* g, uint32_t rank, uint32_t index)
{
    return (g && rank < g->m.size() && index < g->m[rank].layers.size()) ? g->m[rank].layers[index] : nullptr;
}

int pfx_group_band(const pfx_group* g, uint32_t rank, uint32_t* y0, uint32_t* y1)
{
    if (!g || rank >= g->m.size()) return PFX_ERR_INVALID;
    if (y0) *y0 = g->m[rank].y0;
    if (y1) *y1 = g->m[rank].y1;
    return PFX_OK;
}

int pfx_group_flatten_blur(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, float sigma, int all_gather)
{
    if (!g || !layers || n_layers == 0) return PFX_ERR_INVALID;
    if (g->n_layers == 0) return gfail(g, PFX_ERR_INVALID, "no document: call pfx_group_set_document first");
    for (uint32_t l = 0; l < n_layers; ++l)
        if (layers[l].kind == PFX_LAYER_RASTER && layers[l].layer_idx >= g->n_layers)
            return gfail(g, PFX_ERR_INVALID, "layer_idx %u outside the document", layers[l].layer_idx);
    const int radius = pfx_host_gaussian_radius(sigma);
    const bool blur = radius >= 1;
    const uint32_t halo = blur ? (uint32_t)radius : 0u;
    if (halo > g->h + 4096u) return gfail(g, PFX_ERR_UNSUPPORTED, "gaussian radius %d is not a sensible halo", radius);
    {
        const int s = ensure_padded(g, halo);
        if (s != PFX_OK) return s;
    }
    const uint32_t world = (uint32_t)g->m.size(), w = g->w, h = g->h;
    const size_t row_bytes = (size_t)w * 4;

    // 0. a member may not overwrite what the previous call is still reading: its neighbours' halo pulls out of its padded buffer
    // (finished when their ev_done fires) and — when that call gathered an un-blurred result — its own pushes out of it
    for (auto& mem : g->m) {
        PFXG_HIP(g, hipSetDevice(mem.device));
        for (auto& other : g->m)
            if (&other != &mem && g->have_result) PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), other.ev_done, 0));
        if (mem.gather_pending && !g->result_blurred) {
            PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), mem.ev_gather, 0));
            mem.gather_pending = false;
        }
    }
    // 1. every member flattens its band straight into the centre of its padded buffer
    for (auto& mem : g->m) {
        const uint32_t rows = mem.y1 - mem.y0;
        mem.top = std::min(halo, mem.y0);
        mem.bottom = std::min(halo, h - mem.y1);
        PFXG_HIP(g, hipSetDevice(mem.device));
        if (rows) {
            std::vector<const void*> ptrs(n_layers, nullptr);
            std::vector<pfx_layer_info> li(layers, layers + n_layers);
            for (uint32_t l = 0; l < n_layers; ++l)
                if (li[l].kind == PFX_LAYER_RASTER) ptrs[l] = mem.layers[li[l].layer_idx];
            PFXG_CTX(g, mem, pfx_flatten_dev(mem.ctx, ptrs.data(), nullptr, li.data(), n_layers, w, rows,
                                             (uint8_t*)mem.padded + (size_t)mem.top * row_bytes));
        }
        PFXG_HIP(g, hipEventRecord(mem.ev_flat, stream_of(mem)));
    }
    if (blur) {
        // 2. halo rows: member i pulls rows [y0 - top, y0) and [y1, y1 + bottom) from whichever members own them
        for (uint32_t i = 0; i < world; ++i) {
            auto& mem = g->m[i];
            if (mem.y1 == mem.y0) continue;
            PFXG_HIP(g, hipSetDevice(mem.device));
            const uint32_t lo0 = mem.y0 - mem.top, lo1 = mem.y0, hi0 = mem.y1, hi1 = mem.y1 + mem.bottom;
            for (uint32_t j = 0; j < world; ++j) {
                if (j == i) continue;
                const auto& src = g->m[j];
                if (src.y1 == src.y0) continue;
                const uint32_t ranges[2][2] = {{lo0, lo1}, {hi0, hi1}};
                bool waited = false;
                for (const auto& rg : ranges) {
                    const uint32_t s0 = std::max(rg[0], src.y0), s1 = std::min(rg[1], src.y1);
                    if (s1 <= s0) continue;
                    if (!waited) { PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), src.ev_flat, 0)); waited = true; }
                    uint8_t* d = (uint8_t*)mem.padded + (size_t)(s0 - lo0) * row_bytes;                       // padded row 0 = image row lo0
                    const uint8_t* s = (const uint8_t*)src.padded + (size_t)(src.top + (s0 - src.y0)) * row_bytes;
                    PFXG_HIP(g, copy_between(mem, d, src, s, (size_t)(s1 - s0) * row_bytes));
                }
            }
            // 3. blur band + halo; rows [top, top + rows) of the output are the member's result (the previous call's pushes read it)
            if (mem.gather_pending) { PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), mem.ev_gather, 0)); mem.gather_pending = false; }
            const uint32_t prow = mem.top + (mem.y1 - mem.y0) + mem.bottom;
            PFXG_CTX(g, mem, pfx_gaussian_blur_band_dev(mem.ctx, mem.padded, mem.blurred, w, prow, sigma, nullptr, lo0));
        }
    }
    g->result_blurred = blur;
    for (auto& mem : g->m) {
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipEventRecord(mem.ev_done, stream_of(mem)));
    }
    // 4. all-gather: every member pushes its result band into every member's full image (one xGMI link per pair).  The pushes run on
    // the member's copy stream behind ev_done: the next call's flatten (compute stream) overlaps them, its blur waits for them.
    if (all_gather) {
        for (auto& mem : g->m)
            if (!mem.gathered) {
                PFXG_HIP(g, hipSetDevice(mem.device));
                PFXG_HIP(g, hipMalloc(&mem.gathered, std::max<size_t>((size_t)h * row_bytes, 256)));
            }
        for (auto& mem : g->m) {
            const uint32_t rows = mem.y1 - mem.y0;
            if (!rows) continue;
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, hipStreamWaitEvent(mem.s_copy, mem.ev_done, 0));
            const uint8_t* band = (const uint8_t*)(blur ? mem.blurred : mem.padded) + (size_t)mem.top * row_bytes;
            for (auto& dst : g->m) {
                // enqueued on the PRODUCER's copy stream; hipMemcpyPeerAsync accepts any stream
                uint8_t* d = (uint8_t*)dst.gathered + (size_t)mem.y0 * row_bytes;
                if (dst.device == mem.device) PFXG_HIP(g, hipMemcpyAsync(d, band, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, mem.s_copy));
                else PFXG_HIP(g, hipMemcpyPeerAsync(d, dst.device, band, mem.device, (size_t)rows * row_bytes, mem.s_copy));
            }
            PFXG_HIP(g, hipEventRecord(mem.ev_gather, mem.s_copy));
            mem.gather_pending = true;
        }
    }
    g->have_result = true;
    return PFX_OK;
}

int pfx_group_synchronize(pfx_group* g)
{
    if (!g) return PFX_ERR_INVALID;
    for (auto& mem : g->m) {
        PFXG_CTX(g, mem, pfx_ctx_synchronize(mem.ctx));
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipStreamSynchronize(mem.s_copy));
    }
    return PFX_OK;
}

void* pfx_group_result_band_dev(pfx_group* g, uint32_t rank)
{
    if (!g || rank >= g->m.size() || !g->have_result) return nullptr;
    auto& mem = g->m[rank];
    return (uint8_t*)(g->result_blurred ? mem.blurred : mem.padded) + (size_t)mem.top * g->w * 4;
}

void* pfx_group_gathered_dev(pfx_group* g, uint32_t rank)
{
    return (g && rank < g->m.size() && g->have_result) ? g->m[rank].gathered : nullptr;
}

int pfx_group_download(pfx_group* g, uint8_t* dst_host)
{
    if (!g || !dst_host) return PFX_ERR_INVALID;
    if (!g->have_result) return gfail(g, PFX_ERR_INVALID, "no result yet");
    for (uint32_t k = 0; k < g->m.size(); ++k) {
        auto& mem = g->m[k];
        const size_t bytes = (size_t)(mem.y1 - mem.y0) * g->w * 4;
        if (!bytes) continue;
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipMemcpyAsync(dst_host + (size_t)mem.y0 * g->w * 4, pfx_group_result_band_dev(g, k), bytes, hipMemcpyDeviceToHost, stream_of(mem)));
    }
    return pfx_group_synchronize(g);
}

} // extern "C"
