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
//
// Three transports move the halo rows and the gathered bands (pfx_group_set_transport):
//   PEER   (default) hipMemcpyPeerAsync as described above; a pair of devices whose peer access cannot be enabled falls back, pair by
//          pair, to STAGED copies;
//   STAGED device -> pinned host -> device, ordered by events (what a node without xGMI / with peer access disabled can do);
//   RCCL   the halo exchange as ONE group of ncclSend / ncclRecv on the members' compute streams and the all-gather as one group of
//          ncclBroadcast calls (one per band: bands are ragged) on the copy streams — BASELINE's "RCCL halo exchange over xGMI ... and
//          an all-gather".  librccl is loaded at run time (dlopen) the first time this transport is selected: libpfx.so itself does
//          not link it, and a process that already carries an RCCL (PyTorch) keeps using that copy.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <vector>

#include "pfx_internal.h"

// The handful of RCCL declarations this file needs, restated locally (NCCL's stable C API: nccl.h 2.x — ncclResult_t 0 = success, ncclUint8 = 1 in
// ncclDataType_t): libpfx.so neither links librccl nor needs its development headers to build; the functions are resolved with dlsym at run time.
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;   // enum in nccl.h; passed and returned as int by the C ABI
typedef int ncclDataType_t;
}
enum : int { ncclSuccess = 0, ncclUint8 = 1 };
typedef ncclResult_t (*pfx_ncclCommInitAll_t)(ncclComm_t* comms, int ndev, const int* devlist);
typedef ncclResult_t (*pfx_ncclCommDestroy_t)(ncclComm_t comm);
typedef ncclResult_t (*pfx_ncclGroupStart_t)(void);
typedef ncclResult_t (*pfx_ncclGroupEnd_t)(void);
typedef ncclResult_t (*pfx_ncclSend_t)(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
typedef ncclResult_t (*pfx_ncclRecv_t)(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
typedef ncclResult_t (*pfx_ncclBroadcast_t)(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream);
typedef const char* (*pfx_ncclGetErrorString_t)(ncclResult_t result);
typedef ncclResult_t (*pfx_ncclCommGetAsyncError_t)(ncclComm_t comm, ncclResult_t* asyncError);
typedef ncclResult_t (*pfx_ncclCommAbort_t)(ncclComm_t comm);

struct pfx_group_member {
    pfx_ctx* ctx = nullptr;
    int device = 0;
    uint32_t y0 = 0, y1 = 0;            // band rows
    uint32_t top = 0, bottom = 0;       // halo rows held around the band for the current radius
    std::vector<void*> layers;          // band of every layer: (y1 - y0) * w * 4 bytes each
    void* padded = nullptr;             // [top halo | flattened band | bottom halo], (rows + 2 * halo_cap) * w * 4
    void* blurred = nullptr;            // same shape: blur of `padded`
    void* gathered = nullptr;           // full w * h * 4 image (all-gather target), allocated on first use
    void* disp = nullptr;               // band of a displacement field (pfx_group_flatten_warp_displacement), (y1 - y0) * w * 8 bytes
    size_t disp_cap = 0;
    size_t padded_cap = 0;
    hipEvent_t ev_flat = nullptr, ev_done = nullptr; // flatten finished / result band final (halo pulls and blur done), on the compute stream
    hipStream_t s_copy = nullptr;       // the all-gather's pushes run here, behind ev_done, so that the next call's flatten overlaps them
    hipEvent_t ev_gather = nullptr;     // this member's pushes finished (s_copy)
    bool gather_pending = false;        // ev_gather was recorded and not yet waited for by the compute stream
    // STAGED transport: pinned bounce buffers of this member as a SOURCE (its halo rows for the neighbours / its result band)
    uint8_t* h_halo = nullptr;          // [rows it sends upwards | rows it sends downwards], 2 * halo_cap rows
    uint8_t* h_band = nullptr;          // the member's result band
    size_t h_halo_cap = 0, h_band_cap = 0;
    hipEvent_t ev_staged_halo = nullptr, ev_staged_band = nullptr; // the D2H into the bounce buffer finished
    hipEvent_t ev_gin = nullptr;        // every H2D of a gather INTO this member finished (its copy stream)
    bool gin_pending = false;
    ncclComm_t comm = nullptr;          // RCCL transport
    // per-phase clocks of the last pipeline call (pfx_group_set_phase_timing): timing-enabled events on the compute stream in front of the flatten, behind it,
    // behind the last halo row's arrival, behind the filter, and on the copy stream behind the member's all-gather pushes
    hipEvent_t tm[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool tm_valid = false, tm_gather = false;
};

struct pfx_group {
    std::vector<pfx_group_member> m;
    uint32_t w = 0, h = 0, n_layers = 0;
    uint32_t halo_cap = 0;              // halo rows the padded buffers were sized for
    bool have_result = false, result_blurred = false;
    int transport = PFX_GROUP_PEER;
    std::vector<uint8_t> peer_ok;       // [a * n + b]: device of member a may access member b's memory directly (or they share a device)
    bool rccl_ready = false;
    // watchdog (pfx_group_set_watchdog): the next `watch_calls` pipeline calls end with a bounded host-side wait for every member's work; a member
    // that does not finish in time turns a hang (a deadlocked RCCL group, a peer that never sends) into PFX_ERR_HIP naming the pairs involved
    uint32_t watch_ms = 0, watch_calls = 0;
    bool watch_configured = false;      // pfx_group_set_watchdog was called (an explicit "off" must survive selecting RCCL)
    bool phase_timing = false;
    int test_stall_ms = 0;              // PFX_GROUP_TEST_STALL_MS, read once at creation (tests: a member that is late)
    struct halo_piece { uint32_t i, j, s0, s1; };
    std::vector<halo_piece> last_pieces; // halo transfers of the last call (consumer i <- producer j, image rows [s0, s1)): the watchdog's report
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

// ---- RCCL, resolved at run time ----
struct rccl_api {
    void* lib = nullptr;
    pfx_ncclCommInitAll_t CommInitAll = nullptr;
    pfx_ncclCommDestroy_t CommDestroy = nullptr;
    pfx_ncclGroupStart_t GroupStart = nullptr;
    pfx_ncclGroupEnd_t GroupEnd = nullptr;
    pfx_ncclSend_t Send = nullptr;
    pfx_ncclRecv_t Recv = nullptr;
    pfx_ncclBroadcast_t Broadcast = nullptr;
    pfx_ncclGetErrorString_t GetErrorString = nullptr;
    pfx_ncclCommAbort_t CommAbort = nullptr;     // optional: only the watchdog uses it
    bool ok = false;
    std::string why;                             // why loading failed (dlerror() text, taken where the failure happened)
};
rccl_api& rccl()
{
    static rccl_api R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.lib) break;
            const char* e = dlerror();               // read once: the call clears the message
            if (e) R.why = e;
        }
        if (!R.lib) { if (R.why.empty()) R.why = "librccl.so not found"; return; }
        R.why.clear();
        auto sym = [](const char* name) -> void* {
            (void)dlerror();
            void* p = dlsym(R.lib, name);
            if (!p) { const char* e = dlerror(); R.why += (R.why.empty() ? "missing symbol " : ", ") + std::string(e ? e : name); }
            return p;
        };
        R.CommInitAll = (pfx_ncclCommInitAll_t)sym("ncclCommInitAll"); R.CommDestroy = (pfx_ncclCommDestroy_t)sym("ncclCommDestroy");
        R.GroupStart = (pfx_ncclGroupStart_t)sym("ncclGroupStart"); R.GroupEnd = (pfx_ncclGroupEnd_t)sym("ncclGroupEnd");
        R.Send = (pfx_ncclSend_t)sym("ncclSend"); R.Recv = (pfx_ncclRecv_t)sym("ncclRecv"); R.Broadcast = (pfx_ncclBroadcast_t)sym("ncclBroadcast");
        R.GetErrorString = (pfx_ncclGetErrorString_t)sym("ncclGetErrorString");
        R.ok = R.CommInitAll && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv && R.Broadcast && R.GetErrorString;
        R.CommAbort = (pfx_ncclCommAbort_t)dlsym(R.lib, "ncclCommAbort");
    });
    return R;
}
#define PFXG_NCCL(g, call)                                                                                              \
    do {                                                                                                                \
        ncclResult_t _r = (call);                                                                                       \
        if (_r != ncclSuccess) return gfail((g), PFX_ERR_HIP, "%s failed: %s", #call, rccl().GetErrorString(_r));       \
    } while (0)

bool direct(const pfx_group* g, uint32_t dst, uint32_t src)
{
    return g->transport != PFX_GROUP_STAGED && g->peer_ok[(size_t)dst * g->m.size() + src] != 0;
}

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
    if (mem.disp) (void)hipFree(mem.disp);
    mem.disp = nullptr; mem.disp_cap = 0;
    mem.padded = mem.blurred = mem.gathered = nullptr;
    mem.padded_cap = 0;
    if (mem.h_halo) (void)hipHostFree(mem.h_halo);
    if (mem.h_band) (void)hipHostFree(mem.h_band);
    mem.h_halo = mem.h_band = nullptr;
    mem.h_halo_cap = mem.h_band_cap = 0;
}

// every stream of every member idle: nothing in flight may still read or write a buffer that is about to be freed or re-used by hand
void quiesce(pfx_group* g)
{
    for (auto& mem : g->m) {
        if (mem.ctx) (void)pfx_ctx_synchronize(mem.ctx);
        if (mem.s_copy) { (void)hipSetDevice(mem.device); (void)hipStreamSynchronize(mem.s_copy); }
        mem.gather_pending = mem.gin_pending = false;
    }
}

int ensure_padded(pfx_group* g, uint32_t halo)
{
    if (halo <= g->halo_cap && g->m[0].padded_cap != 0) return PFX_OK;
    // the previous call may still be pulling halo rows out of / pushing result bands out of the buffers that are about to be freed
    // (hipFree only waits for the owning device): drain every stream first, and forget the result those buffers held
    quiesce(g);
    g->have_result = false;
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

static int group_create_impl(const int* devices, uint32_t n, pfx_group** out);
int pfx_group_create(const int* devices, uint32_t n, pfx_group** out)
{
    try { return group_create_impl(devices, n, out); }
    catch (const std::bad_alloc&) { return PFX_ERR_OOM; }
    catch (...) { return PFX_ERR_HIP; }
}
static int group_create_impl(const int* devices, uint32_t n, pfx_group** out)
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
            hipEventCreateWithFlags(&g->m[k].ev_staged_halo, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->m[k].ev_staged_band, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->m[k].ev_gin, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithFlags(&g->m[k].s_copy, hipStreamNonBlocking) != hipSuccess) {
            pfx_group_destroy(g);
            return PFX_ERR_HIP;
        }
    }
    // direct xGMI access between every pair of distinct devices.  A pair whose access cannot be enabled (no link, IOMMU / container
    // restrictions, PFX_GROUP_DENY_PEER=1 for tests) is served by staged copies through pinned host memory instead — the calls keep
    // working, only slower.
    if (const char* stall = std::getenv("PFX_GROUP_TEST_STALL_MS")) g->test_stall_ms = std::max(0, std::atoi(stall));
    const char* deny = getenv("PFX_GROUP_DENY_PEER");
    const bool deny_peer = deny && deny[0] == '1';
    g->peer_ok.assign((size_t)n * n, 0);
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t b = 0; b < n; ++b) {
            if (devices[a] == devices[b]) { g->peer_ok[(size_t)a * n + b] = deny_peer ? 0 : 1; continue; }
            int can = 0;
            if (!deny_peer && hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                (void)hipSetDevice(devices[a]);
                const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->peer_ok[(size_t)a * n + b] = 1;
                (void)hipGetLastError(); // clear the sticky "already enabled"
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
        if (mem.ev_staged_halo) (void)hipEventDestroy(mem.ev_staged_halo);
        if (mem.ev_staged_band) (void)hipEventDestroy(mem.ev_staged_band);
        if (mem.ev_gin) (void)hipEventDestroy(mem.ev_gin);
        for (hipEvent_t& e : mem.tm) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (mem.comm && rccl().ok) (void)rccl().CommDestroy(mem.comm);
        if (mem.ctx) pfx_ctx_destroy(mem.ctx);
    }
    delete g;
}

uint32_t pfx_group_size(const pfx_group* g) { return g ? (uint32_t)g->m.size() : 0u; }
pfx_ctx* pfx_group_ctx(pfx_group* g, uint32_t rank) { return (g && rank < g->m.size()) ? g->m[rank].ctx : nullptr; }
const char* pfx_group_last_error(const pfx_group* g) { return g ? g->err.c_str() : "null group"; }

static int group_set_document_impl(pfx_group* g, uint32_t w, uint32_t h, uint32_t n_layers);
int pfx_group_set_document(pfx_group* g, uint32_t w, uint32_t h, uint32_t n_layers)
{
    try { return group_set_document_impl(g, w, h, n_layers); }
    catch (const std::bad_alloc&) { return g ? gfail(g, PFX_ERR_OOM, "out of host memory") : PFX_ERR_OOM; }
    catch (...) { return g ? gfail(g, PFX_ERR_HIP, "unexpected exception") : PFX_ERR_HIP; }
}
static int group_set_document_impl(pfx_group* g, uint32_t w, uint32_t h, uint32_t n_layers)
{
    if (!g) return PFX_ERR_INVALID;
    if (!pfx_dims_ok(w, h) || n_layers == 0 || n_layers > PFX_MAX_LAYERS) return gfail(g, PFX_ERR_INVALID, "bad document geometry");
    quiesce(g);
    for (auto& mem : g->m) free_member_buffers(mem);
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

int pfx_group_set_transport(pfx_group* g, int transport)
{
    if (!g) return PFX_ERR_INVALID;
    if (transport != PFX_GROUP_PEER && transport != PFX_GROUP_STAGED && transport != PFX_GROUP_RCCL)
        return gfail(g, PFX_ERR_INVALID, "unknown transport %d", transport);
    quiesce(g); // a transport change must not overtake copies of the old one
    if (transport == PFX_GROUP_RCCL && !g->rccl_ready) {
        if (!rccl().ok) return gfail(g, PFX_ERR_UNSUPPORTED, "librccl could not be loaded: %s", rccl().why.c_str());
        const size_t n = g->m.size();
        for (size_t a = 0; a < n; ++a)
            for (size_t b2 = a + 1; b2 < n; ++b2)
                if (g->m[a].device == g->m[b2].device)
                    return gfail(g, PFX_ERR_UNSUPPORTED, "RCCL needs one device per member (members %zu and %zu share device %d)", a, b2, g->m[a].device);
        std::vector<int> devs(n);
        std::vector<ncclComm_t> comms(n, nullptr);
        for (size_t k = 0; k < n; ++k) devs[k] = g->m[k].device;
        PFXG_NCCL(g, rccl().CommInitAll(comms.data(), (int)n, devs.data()));
        for (size_t k = 0; k < n; ++k) g->m[k].comm = comms[k];
        g->rccl_ready = true;
        // the first exchanges over fresh communicators are where a mis-paired group would hang: watch them unless the caller configured otherwise
        if (!g->watch_configured) { g->watch_ms = 20000; g->watch_calls = 2; }
    }
    g->transport = transport;
    return PFX_OK;
}

int pfx_group_transport(const pfx_group* g) { return g ? g->transport : -1; }

// ---- watchdog ----
// Bounded wait for everything the members have enqueued (compute and copy streams).  On expiry the report names the members that are still busy
// and, from the last call's halo plan, the (consumer <- producer) pairs they take part in; an RCCL group that never completes is aborted (its
// communicators would block every later call) and the transport falls back to PEER.
static void stall_host_fn(void* ms) { std::this_thread::sleep_for(std::chrono::milliseconds((intptr_t)ms)); }
static int wait_bounded(pfx_group* g, uint32_t timeout_ms, const char* what)
{
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
    std::vector<uint8_t> busy(g->m.size(), 1);
    for (;;) {
        bool any = false;
        for (size_t k = 0; k < g->m.size(); ++k) {
            if (!busy[k]) continue;
            auto& mem = g->m[k];
            (void)hipSetDevice(mem.device);
            const hipError_t a = hipStreamQuery(stream_of(mem)), b = mem.s_copy ? hipStreamQuery(mem.s_copy) : hipSuccess;
            if (a == hipSuccess && b == hipSuccess) busy[k] = 0;
            else if ((a != hipSuccess && a != hipErrorNotReady) || (b != hipSuccess && b != hipErrorNotReady))
                return gfail(g, PFX_ERR_HIP, "%s: member %zu (device %d): %s", what, k, mem.device, hipGetErrorString(a != hipSuccess && a != hipErrorNotReady ? a : b));
            else any = true;
        }
        if (!any) return PFX_OK;
        if (std::chrono::steady_clock::now() >= deadline) break;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    std::string who, pairs;
    for (size_t k = 0; k < g->m.size(); ++k)
        if (busy[k]) who += (who.empty() ? "" : ", ") + std::to_string(k) + " (device " + std::to_string(g->m[k].device) + ")";
    for (const auto& p : g->last_pieces)
        if (busy[p.i] || busy[p.j]) {
            if (pairs.size() > 200) { pairs += " ..."; break; }
            pairs += (pairs.empty() ? "" : ", ") + std::to_string(p.i) + " <- " + std::to_string(p.j) + " rows [" + std::to_string(p.s0) + ", " + std::to_string(p.s1) + ")";
        }
    static const char* const tname[] = {"PEER", "RCCL", "STAGED"};
    const int t = g->transport;
    if (t == PFX_GROUP_RCCL && g->rccl_ready) {
        // without ncclCommAbort the communicators cannot be torn down while a collective hangs on them: they are dropped (a later RCCL selection builds new
        // ones; the hung ones are the price of the hang), never left dangling in the members
        for (auto& mem : g->m) if (mem.comm) { if (rccl().CommAbort) (void)rccl().CommAbort(mem.comm); mem.comm = nullptr; }
        g->rccl_ready = false;
        g->transport = PFX_GROUP_PEER;
    }
    g->have_result = false;
    return gfail(g, PFX_ERR_HIP, "%s: member(s) %s did not finish within %u ms (transport %s%s); halo transfers involving them: %s", what, who.c_str(), timeout_ms,
                 t >= 0 && t <= 2 ? tname[t] : "?", t == PFX_GROUP_RCCL ? ": communicators aborted, transport reset to PEER" : "", pairs.empty() ? "none" : pairs.c_str());
}
static int watchdog_after_call(pfx_group* g, const char* what)
{
    if (g->watch_calls == 0 || g->watch_ms == 0) return PFX_OK;
    if (g->watch_calls != 0xFFFFFFFFu) g->watch_calls -= 1u;
    return wait_bounded(g, g->watch_ms, what);
}

// all-gather: every member's result band lands in every member's full image.  The transfers run on the members' copy streams behind ev_done:
// the next call's flatten (compute stream) overlaps them, its filter waits for them.
static int gather_result(pfx_group* g, bool blur)
{
    const uint32_t world = (uint32_t)g->m.size(), w = g->w, h = g->h;
    const size_t row_bytes = (size_t)w * 4;
    const bool use_rccl = g->transport == PFX_GROUP_RCCL;
    for (auto& mem : g->m)
        if (!mem.gathered) {
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, hipMalloc(&mem.gathered, std::max<size_t>((size_t)h * row_bytes, 256)));
        }
    auto band_of = [&](const pfx_group_member& mem) { return (const uint8_t*)(blur ? mem.blurred : mem.padded) + (size_t)mem.top * row_bytes; };
    if (use_rccl) {
        for (auto& mem : g->m) {
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, hipStreamWaitEvent(mem.s_copy, mem.ev_done, 0));
        }
        // one broadcast per band (bands are ragged by a chunk row, so ncclAllGather's equal counts do not fit), all in one group
        PFXG_NCCL(g, rccl().GroupStart());
        for (uint32_t k = 0; k < world; ++k) {
            const uint32_t rows = g->m[k].y1 - g->m[k].y0;
            if (!rows) continue;
            for (auto& mem : g->m) {
                uint8_t* d = (uint8_t*)mem.gathered + (size_t)g->m[k].y0 * row_bytes;
                const ncclResult_t r = rccl().Broadcast(&mem == &g->m[k] ? (const void*)band_of(mem) : (const void*)d, d, (size_t)rows * row_bytes, ncclUint8,
                                                        (int)k, mem.comm, mem.s_copy);
                if (r != ncclSuccess) { (void)rccl().GroupEnd(); return gfail(g, PFX_ERR_HIP, "ncclBroadcast failed: %s", rccl().GetErrorString(r)); }
            }
        }
        PFXG_NCCL(g, rccl().GroupEnd());
        for (auto& mem : g->m) {
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, hipEventRecord(mem.ev_gather, mem.s_copy));
            mem.gather_pending = true;
        }
    } else {
        for (uint32_t k = 0; k < world; ++k) {
            auto& mem = g->m[k];
            const uint32_t rows = mem.y1 - mem.y0;
            if (!rows) continue;
            const size_t bytes = (size_t)rows * row_bytes;
            bool any_staged = false;
            for (uint32_t d = 0; d < world; ++d) any_staged = any_staged || !direct(g, d, k);
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, hipStreamWaitEvent(mem.s_copy, mem.ev_done, 0));
            if (any_staged) {
                if (mem.h_band_cap < bytes) {
                    quiesce(g);
                    if (mem.h_band) (void)hipHostFree(mem.h_band);
                    mem.h_band = nullptr; mem.h_band_cap = 0;
                    PFXG_HIP(g, hipHostMalloc((void**)&mem.h_band, bytes, hipHostMallocPortable));
                    mem.h_band_cap = bytes;
                    PFXG_HIP(g, hipStreamWaitEvent(mem.s_copy, mem.ev_done, 0));
                }
                // the previous gather's H2Ds out of this bounce buffer run on the destinations' copy streams
                for (auto& dst : g->m)
                    if (dst.gin_pending) PFXG_HIP(g, hipStreamWaitEvent(mem.s_copy, dst.ev_gin, 0));
                PFXG_HIP(g, hipMemcpyAsync(mem.h_band, band_of(mem), bytes, hipMemcpyDeviceToHost, mem.s_copy));
                PFXG_HIP(g, hipEventRecord(mem.ev_staged_band, mem.s_copy));
            }
            for (uint32_t d = 0; d < world; ++d) {
                auto& dst = g->m[d];
                uint8_t* out = (uint8_t*)dst.gathered + (size_t)mem.y0 * row_bytes;
                if (direct(g, d, k)) {
                    // enqueued on the PRODUCER's copy stream; hipMemcpyPeerAsync accepts any stream
                    if (dst.device == mem.device) PFXG_HIP(g, hipMemcpyAsync(out, band_of(mem), bytes, hipMemcpyDeviceToDevice, mem.s_copy));
                    else PFXG_HIP(g, hipMemcpyPeerAsync(out, dst.device, band_of(mem), mem.device, bytes, mem.s_copy));
                } else {
                    PFXG_HIP(g, hipSetDevice(dst.device));
                    PFXG_HIP(g, hipStreamWaitEvent(dst.s_copy, mem.ev_staged_band, 0));
                    PFXG_HIP(g, hipMemcpyAsync(out, mem.h_band, bytes, hipMemcpyHostToDevice, dst.s_copy));
                    PFXG_HIP(g, hipSetDevice(mem.device));
                }
            }
            PFXG_HIP(g, hipEventRecord(mem.ev_gather, mem.s_copy));
            mem.gather_pending = true;
        }
        for (auto& dst : g->m) { // staged arrivals into `dst` are complete when its copy stream reaches this point
            PFXG_HIP(g, hipSetDevice(dst.device));
            PFXG_HIP(g, hipEventRecord(dst.ev_gin, dst.s_copy));
            dst.gin_pending = true;
        }
    }
    return PFX_OK;
}

// halo rows of a band filter: what pfx_*_band_dev needs around a band ("halo rows of the Gaussian / box / median vertical pass")
static int band_filter_halo(int filter, float param, uint32_t* halo)
{
    switch (filter) {
    case PFX_BAND_NONE: *halo = 0; return PFX_OK;
    case PFX_BAND_GAUSSIAN: { const int r = pfx_host_gaussian_radius(param); *halo = r >= 1 ? (uint32_t)r : 0u; return PFX_OK; }
    case PFX_BAND_BOX: *halo = param < 0.5f ? 0u : (uint32_t)std::min(ceilf(param), 4096.0f); return PFX_OK;       // blur.rs:234,241
    case PFX_BAND_MEDIAN: *halo = (uint32_t)std::max(std::min(param, 4096.0f), 1.0f); return PFX_OK;              // noise.rs:364
    default: return PFX_ERR_INVALID;
    }
}

static int flatten_filter_impl(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, int filter, float param, int all_gather)
{
    if (!g || !layers || n_layers == 0) return PFX_ERR_INVALID;
    if (g->n_layers == 0) return gfail(g, PFX_ERR_INVALID, "no document: call pfx_group_set_document first");
    for (uint32_t l = 0; l < n_layers; ++l)
        if (layers[l].kind == PFX_LAYER_RASTER && layers[l].layer_idx >= g->n_layers)
            return gfail(g, PFX_ERR_INVALID, "layer_idx %u outside the document", layers[l].layer_idx);
    // parameters the filter itself would refuse are refused HERE, before any buffer is (re)allocated or a member has flattened anything — and a NaN
    // never reaches the float -> integer conversions below
    if (filter != PFX_BAND_NONE && !std::isfinite(param)) return gfail(g, PFX_ERR_INVALID, "band filter parameter is not finite");
    if (filter == PFX_BAND_BOX && ceilf(param) >= 2040.0f) return gfail(g, PFX_ERR_UNSUPPORTED, "box blur radius %g beyond the device limit", (double)param);
    if (filter == PFX_BAND_MEDIAN && param > (float)PFX_MEDIAN_MAX_RADIUS) return gfail(g, PFX_ERR_UNSUPPORTED, "median radius %g beyond PFX_MEDIAN_MAX_RADIUS", (double)param);
    uint32_t halo = 0;
    if (band_filter_halo(filter, param, &halo) != PFX_OK) return gfail(g, PFX_ERR_INVALID, "unknown band filter %d", filter);
    const bool blur = halo >= 1;
    if (halo > g->h + 4096u) return gfail(g, PFX_ERR_UNSUPPORTED, "a halo of %u rows is not sensible", halo);
    {
        const int s = ensure_padded(g, halo);
        if (s != PFX_OK) return s;
    }
    const uint32_t world = (uint32_t)g->m.size(), w = g->w, h = g->h;
    const size_t row_bytes = (size_t)w * 4;
    const bool use_rccl = g->transport == PFX_GROUP_RCCL;

    // 0. a member may not overwrite what the previous call is still reading: its neighbours' halo pulls out of its padded buffer
    // (finished when their ev_done fires; staged pulls read the bounce buffer, whose D2H is ordered on the member's own stream) and —
    // when that call gathered an un-filtered result — its own pushes out of it
    for (auto& mem : g->m) {
        PFXG_HIP(g, hipSetDevice(mem.device));
        for (auto& other : g->m)
            if (&other != &mem && g->have_result) PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), other.ev_done, 0));
        if (mem.gather_pending && !g->result_blurred) {
            PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), mem.ev_gather, 0));
            mem.gather_pending = false;
        }
    }
    g->last_pieces.clear();   // a call without a halo has no transfers to report
    auto mark = [&](pfx_group_member& mem, int k, hipStream_t st) -> hipError_t {   // phase clocks (off by default: five more events per member and call)
        if (!g->phase_timing) return hipSuccess;
        if (!mem.tm[k]) { const hipError_t e = hipEventCreate(&mem.tm[k]); if (e != hipSuccess) return e; }
        return hipEventRecord(mem.tm[k], st);
    };
    // 1. every member flattens its band straight into the centre of its padded buffer
    for (auto& mem : g->m) {
        const uint32_t rows = mem.y1 - mem.y0;
        mem.top = std::min(halo, mem.y0);
        mem.bottom = std::min(halo, h - mem.y1);
        PFXG_HIP(g, hipSetDevice(mem.device));
        mem.tm_valid = false; mem.tm_gather = false;
        PFXG_HIP(g, mark(mem, 0, stream_of(mem)));
        if (rows) {
            std::vector<const void*> ptrs(n_layers, nullptr);
            std::vector<pfx_layer_info> li(layers, layers + n_layers);
            for (uint32_t l = 0; l < n_layers; ++l)
                if (li[l].kind == PFX_LAYER_RASTER) ptrs[l] = mem.layers[li[l].layer_idx];
            PFXG_CTX(g, mem, pfx_flatten_dev(mem.ctx, ptrs.data(), nullptr, li.data(), n_layers, w, rows,
                                             (uint8_t*)mem.padded + (size_t)mem.top * row_bytes));
        }
        if (&mem == &g->m.back()) {
            // test hook: a peer that is late (PFX_GROUP_TEST_STALL_MS, read when the group was created): its stream sleeps in a host function in front of the
            // event its neighbours wait for
            if (g->test_stall_ms > 0) PFXG_HIP(g, hipLaunchHostFunc(stream_of(mem), stall_host_fn, (void*)(intptr_t)g->test_stall_ms));
        }
        PFXG_HIP(g, hipEventRecord(mem.ev_flat, stream_of(mem)));
        PFXG_HIP(g, mark(mem, 1, stream_of(mem)));
    }
    if (blur) {
        // 2. halo rows: member i needs rows [y0 - top, y0) and [y1, y1 + bottom) from whichever members own them.
        // Every (consumer i, producer j, image rows [s0, s1)) piece is listed once and then moved by the transport in use.
        using piece = pfx_group::halo_piece;
        std::vector<piece>& pieces = g->last_pieces;
        for (uint32_t i = 0; i < world; ++i) {
            const auto& mem = g->m[i];
            if (mem.y1 == mem.y0) continue;
            const uint32_t ranges[2][2] = {{mem.y0 - mem.top, mem.y0}, {mem.y1, mem.y1 + mem.bottom}};
            for (uint32_t j = 0; j < world; ++j) {
                if (j == i || g->m[j].y1 == g->m[j].y0) continue;
                for (const auto& rg : ranges) {
                    const uint32_t s0 = std::max(rg[0], g->m[j].y0), s1 = std::min(rg[1], g->m[j].y1);
                    if (s1 > s0) pieces.push_back({i, j, s0, s1});
                }
            }
        }
        auto dst_of = [&](const piece& p) { const auto& mem = g->m[p.i]; return (uint8_t*)mem.padded + (size_t)(p.s0 - (mem.y0 - mem.top)) * row_bytes; };
        auto src_of = [&](const piece& p) { const auto& src = g->m[p.j]; return (const uint8_t*)src.padded + (size_t)(src.top + (p.s0 - src.y0)) * row_bytes; };
        if (use_rccl) {
            // one group: every send sits on the producer's compute stream (behind its flatten), every receive on the consumer's (in
            // front of its filter) — no events, RCCL pairs them up over xGMI
            PFXG_NCCL(g, rccl().GroupStart());
            for (const auto& p : pieces) {
                const size_t bytes = (size_t)(p.s1 - p.s0) * row_bytes;
                const ncclResult_t r1 = rccl().Send(src_of(p), bytes, ncclUint8, (int)p.i, g->m[p.j].comm, stream_of(g->m[p.j]));
                const ncclResult_t r2 = rccl().Recv(dst_of(p), bytes, ncclUint8, (int)p.j, g->m[p.i].comm, stream_of(g->m[p.i]));
                if (r1 != ncclSuccess || r2 != ncclSuccess) {
                    (void)rccl().GroupEnd();
                    return gfail(g, PFX_ERR_HIP, "ncclSend / ncclRecv failed: %s", rccl().GetErrorString(r1 != ncclSuccess ? r1 : r2));
                }
            }
            PFXG_NCCL(g, rccl().GroupEnd());
        } else {
            // producers whose rows travel through the host stage them once (D2H into their pinned bounce buffer, behind their flatten)
            std::vector<uint8_t> stages(world, 0);
            for (const auto& p : pieces) if (!direct(g, p.i, p.j)) stages[p.j] = 1;
            for (uint32_t j = 0; j < world; ++j) {
                if (!stages[j]) continue;
                auto& src = g->m[j];
                const uint32_t rows = src.y1 - src.y0, n_up = std::min(halo, rows), n_dn = std::min(halo, rows);
                const size_t need = (size_t)(n_up + n_dn) * row_bytes;
                PFXG_HIP(g, hipSetDevice(src.device));
                if (src.h_halo_cap < need) {
                    // consumers of the old buffer were drained by step 0's waits only on the device side: their H2Ds ran on THEIR streams
                    quiesce(g);
                    if (src.h_halo) (void)hipHostFree(src.h_halo);
                    src.h_halo = nullptr; src.h_halo_cap = 0;
                    PFXG_HIP(g, hipHostMalloc((void**)&src.h_halo, need, hipHostMallocPortable));
                    src.h_halo_cap = need;
                }
                const uint8_t* band0 = (const uint8_t*)src.padded + (size_t)src.top * row_bytes;
                // [first n_up rows of the band | last n_dn rows of the band]: what the members above / below can ask for
                PFXG_HIP(g, hipMemcpyAsync(src.h_halo, band0, (size_t)n_up * row_bytes, hipMemcpyDeviceToHost, stream_of(src)));
                PFXG_HIP(g, hipMemcpyAsync(src.h_halo + (size_t)n_up * row_bytes, band0 + (size_t)(rows - n_dn) * row_bytes, (size_t)n_dn * row_bytes,
                                           hipMemcpyDeviceToHost, stream_of(src)));
                PFXG_HIP(g, hipEventRecord(src.ev_staged_halo, stream_of(src)));
            }
            for (const auto& p : pieces) {
                auto& mem = g->m[p.i];
                const auto& src = g->m[p.j];
                const size_t bytes = (size_t)(p.s1 - p.s0) * row_bytes;
                PFXG_HIP(g, hipSetDevice(mem.device));
                if (direct(g, p.i, p.j)) {
                    PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), src.ev_flat, 0));
                    PFXG_HIP(g, copy_between(mem, dst_of(p), src, src_of(p), bytes));
                } else {
                    const uint32_t rows = src.y1 - src.y0, n_up = std::min(halo, rows), n_dn = std::min(halo, rows);
                    // rows below the consumer come from the TOP of the producer's band, rows above it from the BOTTOM
                    const bool from_top = p.s0 >= mem.y1;
                    const size_t off = from_top ? (size_t)(p.s0 - src.y0) * row_bytes
                                                : (size_t)n_up * row_bytes + (size_t)(p.s0 - (src.y1 - n_dn)) * row_bytes;
                    PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), src.ev_staged_halo, 0));
                    PFXG_HIP(g, hipMemcpyAsync(dst_of(p), src.h_halo + off, bytes, hipMemcpyHostToDevice, stream_of(mem)));
                }
            }
        }
        // 3. filter band + halo; rows [top, top + rows) of the output are the member's result (the previous call's pushes read it)
        for (uint32_t i = 0; i < world; ++i) {
            auto& mem = g->m[i];
            if (mem.y1 == mem.y0) continue;
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, mark(mem, 2, stream_of(mem)));   // every halo row has arrived (the transfers above sit on this stream or are waited for by it)
            if (mem.gather_pending) { PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), mem.ev_gather, 0)); mem.gather_pending = false; }
            const uint32_t prow = mem.top + (mem.y1 - mem.y0) + mem.bottom, first = mem.y0 - mem.top;
            if (filter == PFX_BAND_GAUSSIAN) PFXG_CTX(g, mem, pfx_gaussian_blur_band_dev(mem.ctx, mem.padded, mem.blurred, w, prow, param, nullptr, first));
            else if (filter == PFX_BAND_BOX) PFXG_CTX(g, mem, pfx_box_blur_band_dev(mem.ctx, mem.padded, mem.blurred, w, prow, param, nullptr, nullptr, first));
            else PFXG_CTX(g, mem, pfx_median_band_dev(mem.ctx, mem.padded, mem.blurred, w, prow, (uint32_t)halo, nullptr, first));
        }
    }
    g->result_blurred = blur;
    for (auto& mem : g->m) {
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipEventRecord(mem.ev_done, stream_of(mem)));
        if (!blur || mem.y1 == mem.y0) PFXG_HIP(g, mark(mem, 2, stream_of(mem)));   // a member with an empty band skipped the filter loop: its phase 2 is empty, not stale
        PFXG_HIP(g, mark(mem, 3, stream_of(mem)));
        mem.tm_valid = g->phase_timing;
    }
    // 4. all-gather (gather_result above)
    if (all_gather) {
        const int gs = gather_result(g, blur);
        if (gs != PFX_OK) return gs;
        for (auto& mem : g->m) {
            if (!mem.gather_pending) continue;
            PFXG_HIP(g, hipSetDevice(mem.device));
            PFXG_HIP(g, mark(mem, 4, mem.s_copy));
            mem.tm_gather = g->phase_timing;
        }
    }
    g->have_result = true;
    return PFX_OK;
}

// the C ABI must not let a C++ exception (std::vector growth) escape
int pfx_group_flatten_filter(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, int filter, float param, int all_gather)
{
    try {
        const int s = flatten_filter_impl(g, layers, n_layers, filter, param, all_gather);
        return s != PFX_OK ? s : watchdog_after_call(g, "pfx_group_flatten_filter");
    }
    catch (const std::bad_alloc&) { return g ? gfail(g, PFX_ERR_OOM, "out of host memory") : PFX_ERR_OOM; }
    catch (...) { return g ? gfail(g, PFX_ERR_HIP, "unexpected exception") : PFX_ERR_HIP; }
}

int pfx_group_flatten_blur(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, float sigma, int all_gather)
{
    return pfx_group_flatten_filter(g, layers, n_layers, pfx_host_gaussian_radius(sigma) >= 1 ? PFX_BAND_GAUSSIAN : PFX_BAND_NONE, sigma, all_gather);
}

int pfx_group_set_watchdog(pfx_group* g, uint32_t timeout_ms, uint32_t calls)
{
    if (!g) return PFX_ERR_INVALID;
    g->watch_ms = timeout_ms;
    g->watch_calls = timeout_ms ? calls : 0u;
    g->watch_configured = true;
    return PFX_OK;
}

int pfx_group_set_phase_timing(pfx_group* g, int on)
{
    if (!g) return PFX_ERR_INVALID;
    g->phase_timing = on != 0;
    return PFX_OK;
}

int pfx_group_phase_ms(pfx_group* g, uint32_t rank, double out_ms[4])
{
    if (!g || !out_ms || rank >= g->m.size()) return PFX_ERR_INVALID;
    auto& mem = g->m[rank];
    out_ms[0] = out_ms[1] = out_ms[2] = out_ms[3] = 0.0;
    if (!mem.tm_valid) return gfail(g, PFX_ERR_INVALID, "no phase clocks: pfx_group_set_phase_timing(g, 1) before the pipeline call");
    PFXG_HIP(g, hipSetDevice(mem.device));
    PFXG_HIP(g, hipEventSynchronize(mem.tm[3]));
    for (int k = 0; k < 3; ++k) {
        float ms = 0.0f;
        PFXG_HIP(g, hipEventElapsedTime(&ms, mem.tm[k], mem.tm[k + 1]));
        out_ms[k] = ms;
    }
    if (mem.tm_gather) {
        float ms = 0.0f;
        PFXG_HIP(g, hipEventSynchronize(mem.tm[4]));
        PFXG_HIP(g, hipEventElapsedTime(&ms, mem.tm[3], mem.tm[4]));
        out_ms[3] = ms;
    }
    return PFX_OK;
}

int pfx_group_synchronize_timeout(pfx_group* g, uint32_t timeout_ms)
{
    if (!g) return PFX_ERR_INVALID;
    return wait_bounded(g, timeout_ms, "pfx_group_synchronize_timeout");
}

// ---- warps of a sharded document (replicate the source, bands of the output; ref: src/ops/transform.rs:1288-1345, 1687-1761) ----
// flatten (bands) -> all-gather of the flattened bands, so that every member holds the whole source -> every member warps its band of the output.
// kind 0: displacement field (disp_host = w * h xy pairs), kind 1: fused Catmull-Rom mesh warp.
static int flatten_warp_impl(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, int kind, const float* disp_host, const float* orig_pts,
                             const float* def_pts, uint32_t cols, uint32_t rows_grid)
{
    if (!g || !layers || n_layers == 0) return PFX_ERR_INVALID;
    if (kind == 0 && !disp_host) return gfail(g, PFX_ERR_INVALID, "null displacement field");
    if (kind == 1 && (!def_pts || cols == 0 || rows_grid == 0)) return gfail(g, PFX_ERR_INVALID, "bad mesh");
    {
        const int s = flatten_filter_impl(g, layers, n_layers, PFX_BAND_NONE, 0.0f, 1);
        if (s != PFX_OK) return s;
    }
    const uint32_t w = g->w, h = g->h;
    const size_t row_bytes = (size_t)w * 4;
    for (auto& mem : g->m) {
        const uint32_t rows = mem.y1 - mem.y0;
        PFXG_HIP(g, hipSetDevice(mem.device));
        // the whole source has arrived when every producer's pushes (their copy streams) and the staged arrivals into this member are done
        for (auto& prod : g->m) PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), prod.ev_gather, 0));
        if (mem.gin_pending) PFXG_HIP(g, hipStreamWaitEvent(stream_of(mem), mem.ev_gin, 0));
        mem.gather_pending = false;
        if (!rows) continue;
        uint8_t* out = (uint8_t*)mem.blurred + (size_t)mem.top * row_bytes;
        if (kind == 0) {
            const size_t need = (size_t)rows * w * 8;
            if (mem.disp_cap < need) {
                PFXG_CTX(g, mem, pfx_ctx_synchronize(mem.ctx)); // the previous warp may still read the old buffer
                if (mem.disp) (void)hipFree(mem.disp);
                mem.disp = nullptr; mem.disp_cap = 0;
                PFXG_HIP(g, hipMalloc(&mem.disp, need));
                mem.disp_cap = need;
            }
            PFXG_HIP(g, hipMemcpyAsync(mem.disp, disp_host + (size_t)mem.y0 * w * 2, need, hipMemcpyHostToDevice, stream_of(mem)));
            PFXG_CTX(g, mem, pfx_warp_displacement_band_dev(mem.ctx, mem.gathered, w, h, mem.disp, w, rows, out, mem.y0));
        } else {
            PFXG_CTX(g, mem, pfx_warp_mesh_catmull_rom_band_dev(mem.ctx, mem.gathered, orig_pts, def_pts, cols, rows_grid, w, h, out, mem.y0, rows));
        }
    }
    for (auto& mem : g->m) {
        PFXG_HIP(g, hipSetDevice(mem.device));
        PFXG_HIP(g, hipEventRecord(mem.ev_done, stream_of(mem)));
    }
    g->result_blurred = true; // the result bands live in `blurred`
    g->have_result = true;
    if (kind == 0) // the field is pageable host memory of the caller: its copies must have left it when this returns
        for (auto& mem : g->m) PFXG_CTX(g, mem, pfx_ctx_synchronize(mem.ctx));
    return PFX_OK;
}

int pfx_group_flatten_warp_displacement(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, const float* disp_host)
{
    try {
        const int s = flatten_warp_impl(g, layers, n_layers, 0, disp_host, nullptr, nullptr, 0, 0);
        return s != PFX_OK ? s : watchdog_after_call(g, "pfx_group_flatten_warp_displacement");
    }
    catch (const std::bad_alloc&) { return g ? gfail(g, PFX_ERR_OOM, "out of host memory") : PFX_ERR_OOM; }
    catch (...) { return g ? gfail(g, PFX_ERR_HIP, "unexpected exception") : PFX_ERR_HIP; }
}

int pfx_group_flatten_warp_mesh(pfx_group* g, const pfx_layer_info* layers, uint32_t n_layers, const float* orig_pts_xy, const float* deformed_pts_xy,
                                uint32_t cols, uint32_t rows)
{
    try {
        const int s = flatten_warp_impl(g, layers, n_layers, 1, nullptr, orig_pts_xy, deformed_pts_xy, cols, rows);
        return s != PFX_OK ? s : watchdog_after_call(g, "pfx_group_flatten_warp_mesh");
    }
    catch (const std::bad_alloc&) { return g ? gfail(g, PFX_ERR_OOM, "out of host memory") : PFX_ERR_OOM; }
    catch (...) { return g ? gfail(g, PFX_ERR_HIP, "unexpected exception") : PFX_ERR_HIP; }
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
