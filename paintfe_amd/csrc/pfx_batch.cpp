// pfx_batch.cpp — the file loop of the reference's CLI (src/cli.rs:159-216: `for input in inputs { run_one(...) }`, one
// independent image per iteration) as a streamed multi-GPU pipeline behind the C ABI (include/pfx.h: pfx_batch_pipeline).
//
// BASELINE config 5 / SURVEY.md §8(d) S4: a batch of independent images, per image  Gaussian(sigma) -> HSL -> flatten under
// `n_overlays` overlay layers.  Images are sharded BY IMAGE across the devices (image i -> device i mod N: no data-path
// collective).  On every device three in-order HIP streams form the pipeline — one for the H2D copies, one for the
// kernels, one for the D2H copies, chained by events — over a ring of `slots` buffer sets, so that the upload of image k+1, the
// kernels of image k and the download of image k-1 are in flight at once and each copy direction keeps ONE DMA queue busy back to
// back.  (One stream per slot, the first version, let the runtime spread the copies of 3+ streams over its hardware queues:
// 830-1110 images/s depending on the slot count, against 1440 images/s that the link gives both ways at once —
// tools/lab/pcie_duplex.py.)  The PCIe link (Gen5 x16: 33 MB per 4K image each way ~ 0.7 ms), not the 0.15 ms of kernels, sets the
// pace.  One host thread per device enqueues; nothing touches pixels on the CPU.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "pfx_internal.h"

namespace {

struct slot {
    void *d_in = nullptr, *d_out = nullptr;
    uint8_t* h_out = nullptr;         // pinned
    hipEvent_t uploaded = nullptr, computed = nullptr, done = nullptr, k0 = nullptr, k1 = nullptr;
    int64_t image = -1;               // image whose result is in flight in this slot
    bool timed = false;
};

struct worker_result {
    int status = PFX_OK;
    std::string err;
    uint64_t images = 0;
    double kernel_ms = 0.0;
    uint32_t kernel_samples = 0;
    std::chrono::steady_clock::time_point t_begin, t_end; // first enqueue .. last result retired (setup excluded)
};

void run_device_body(int device, uint32_t rank, uint32_t world, uint32_t n_images, const pfx_batch_params& P, const uint8_t* const* pool,
                     uint32_t n_pool, worker_result& R);

// thread entry: no C++ exception may leave a worker thread (std::terminate) or the C ABI (include/pfx.h)
void run_device(int device, uint32_t rank, uint32_t world, uint32_t n_images, const pfx_batch_params& P, const uint8_t* const* pool,
                uint32_t n_pool, worker_result& R) noexcept
{
    try { run_device_body(device, rank, world, n_images, P, pool, n_pool, R); }
    catch (const std::bad_alloc&) { R.status = PFX_ERR_OOM; try { R.err = "out of host memory in a batch worker"; } catch (...) {} }
    catch (...) { R.status = PFX_ERR_HIP; try { R.err = "unexpected exception in a batch worker"; } catch (...) {} }
}

void run_device_body(int device, uint32_t rank, uint32_t world, uint32_t n_images, const pfx_batch_params& P, const uint8_t* const* pool,
                     uint32_t n_pool, worker_result& R)
{
    const size_t bytes = (size_t)P.w * P.h * 4;
    const uint32_t n_slots = P.slots >= 2 && P.slots <= 8 ? P.slots : 3;
    std::vector<slot> S(n_slots);
    std::vector<void*> d_ov(P.n_overlays, nullptr);
    pfx_ctx* ctx = nullptr;            // kernels of all slots run in order on this context's stream: one scratch set serves them
    void* d_b = nullptr;   // blur -> HSL result, layer 0 of the flatten (the blurred image itself needs no buffer: pfx_chain_dev)
    hipStream_t s_up = nullptr, s_down = nullptr;
    auto fail = [&](int st, const std::string& m) { if (R.status == PFX_OK) { R.status = st; R.err = m; } };
    auto hip = [&](hipError_t e, const char* what) { if (e != hipSuccess) fail(e == hipErrorOutOfMemory ? PFX_ERR_OOM : PFX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };

    if (!hip(hipSetDevice(device), "hipSetDevice")) return;
    if (pfx_ctx_create(device, &ctx) != PFX_OK) fail(PFX_ERR_HIP, "pfx_ctx_create failed");
    else if (!P.out_of_contract_fast_gaussian) (void)pfx_ctx_set_exact(ctx, 1);
    if (R.status == PFX_OK) {
        (void)(hip(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking), "hipStreamCreate") && hip(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking), "hipStreamCreate") &&
               hip(hipMalloc(&d_b, bytes), "hipMalloc"));
    }
    if (R.status == PFX_OK)
        for (auto& s : S) {
            if (!hip(hipMalloc(&s.d_in, bytes), "hipMalloc") || !hip(hipMalloc(&s.d_out, bytes), "hipMalloc") ||
                !hip(hipHostMalloc((void**)&s.h_out, bytes, hipHostMallocDefault), "hipHostMalloc") ||
                !hip(hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming), "hipEventCreate") ||
                !hip(hipEventCreateWithFlags(&s.computed, hipEventDisableTiming), "hipEventCreate") ||
                !hip(hipEventCreateWithFlags(&s.done, hipEventDisableTiming), "hipEventCreate") || !hip(hipEventCreate(&s.k0), "hipEventCreate") ||
                !hip(hipEventCreate(&s.k1), "hipEventCreate"))
                break;
        }
    // overlays: resident on the device for the whole batch (uploaded once)
    if (R.status == PFX_OK)
        for (uint32_t o = 0; o < P.n_overlays; ++o) {
            if (!hip(hipMalloc(&d_ov[o], bytes), "hipMalloc")) break;
            if (!hip(hipMemcpy(d_ov[o], P.overlays_host[o], bytes, hipMemcpyHostToDevice), "overlay upload")) break;
        }

    auto retire = [&](slot& s) { // the slot's previous image: wait for its D2H, hand the result over
        if (s.image < 0) return;
        if (!hip(hipEventSynchronize(s.done), "hipEventSynchronize")) return;
        if (s.timed) {
            float ms = 0.0f;
            if (hipEventElapsedTime(&ms, s.k0, s.k1) == hipSuccess) { R.kernel_ms += ms; ++R.kernel_samples; }
            s.timed = false;
        }
        for (uint32_t k = 0; k < P.n_keep; ++k)
            if (P.keep_indices[k] == (uint32_t)s.image) std::memcpy(P.keep_out[k], s.h_out, bytes);
        ++R.images;
        s.image = -1;
    };

    std::vector<pfx_layer_info> li(1 + P.n_overlays);
    std::memset(li.data(), 0, li.size() * sizeof(pfx_layer_info));
    li[0].opacity = 1.0f; li[0].visible = 1; li[0].blend_mode = 0; li[0].kind = PFX_LAYER_RASTER;
    for (uint32_t o = 0; o < P.n_overlays; ++o) {
        li[1 + o].opacity = P.overlay_opacity ? P.overlay_opacity[o] : 1.0f;
        li[1 + o].visible = 1; li[1 + o].blend_mode = P.overlay_modes[o]; li[1 + o].kind = PFX_LAYER_RASTER;
    }
    const float hsl[3] = {P.hue, P.saturation, P.lightness};

    R.t_begin = std::chrono::steady_clock::now();
    uint32_t turn = 0;
    const hipStream_t s_comp = R.status == PFX_OK ? (hipStream_t)pfx_ctx_stream(ctx) : nullptr;
    for (uint32_t img = rank; img < n_images && R.status == PFX_OK; img += world, ++turn) {
        slot& s = S[turn % n_slots];
        retire(s); // its d_in was consumed and its d_out downloaded: the buffers are free
        if (R.status != PFX_OK) break;
        const uint8_t* src = pool[img % n_pool]; // pinned (registered) by the caller of run_device
        if (!hip(hipMemcpyAsync(s.d_in, src, bytes, hipMemcpyHostToDevice, s_up), "H2D") || !hip(hipEventRecord(s.uploaded, s_up), "hipEventRecord") ||
            !hip(hipStreamWaitEvent(s_comp, s.uploaded, 0), "hipStreamWaitEvent"))
            break;
        s.timed = (turn % 8) == 0;
        if (s.timed) (void)hipEventRecord(s.k0, s_comp);
        const void* ptrs[1 + 8];
        ptrs[0] = d_b;
        for (uint32_t o = 0; o < P.n_overlays; ++o) ptrs[1 + o] = d_ov[o];
        // Gaussian -> HSL as a chain: ONE launch in the bit-exact mode at sigma <= 5.33 (the blurred image never exists in memory), blur + in-place HSL otherwise
        pfx_chain_op ch[2];
        std::memset(ch, 0, sizeof ch);
        ch[0].kind = PFX_CHAIN_GAUSSIAN; ch[0].n_params = 1; ch[0].params[0] = P.sigma;
        ch[1].kind = PFX_CHAIN_ADJUST; ch[1].op = PFX_OP_HSL; ch[1].n_params = 3;
        ch[1].params[0] = hsl[0]; ch[1].params[1] = hsl[1]; ch[1].params[2] = hsl[2];
        int rc = pfx_chain_dev(ctx, s.d_in, d_b, P.w, P.h, ch, 2);
        if (rc == PFX_OK) rc = pfx_flatten_dev(ctx, ptrs, nullptr, li.data(), 1 + P.n_overlays, P.w, P.h, s.d_out);
        if (rc != PFX_OK) { fail(rc, pfx_last_error(ctx)); break; }
        if (s.timed) (void)hipEventRecord(s.k1, s_comp);
        if (!hip(hipEventRecord(s.computed, s_comp), "hipEventRecord") || !hip(hipStreamWaitEvent(s_down, s.computed, 0), "hipStreamWaitEvent") ||
            !hip(hipMemcpyAsync(s.h_out, s.d_out, bytes, hipMemcpyDeviceToHost, s_down), "D2H") || !hip(hipEventRecord(s.done, s_down), "hipEventRecord"))
            break;
        s.image = img;
    }
    for (uint32_t k = 0; k < n_slots && R.status == PFX_OK; ++k) retire(S[(turn + k) % n_slots]); // drain in issue order
    R.t_end = std::chrono::steady_clock::now();

    (void)hipSetDevice(device);
    if (ctx) (void)pfx_ctx_synchronize(ctx);
    if (s_up) { (void)hipStreamSynchronize(s_up); (void)hipStreamDestroy(s_up); }
    if (s_down) { (void)hipStreamSynchronize(s_down); (void)hipStreamDestroy(s_down); }
    for (auto& s : S) {
        if (s.d_in) (void)hipFree(s.d_in);
        if (s.d_out) (void)hipFree(s.d_out);
        if (s.h_out) (void)hipHostFree(s.h_out);
        for (hipEvent_t e : {s.uploaded, s.computed, s.done, s.k0, s.k1})
            if (e) (void)hipEventDestroy(e);
    }
    if (d_b) (void)hipFree(d_b);
    for (void* p : d_ov) if (p) (void)hipFree(p);
    if (ctx) pfx_ctx_destroy(ctx);
}

} // namespace

static int batch_pipeline_impl(const int* devices, uint32_t n_devices, uint32_t n_images, const pfx_batch_params* params,
                               const uint8_t* const* image_pool_host, uint32_t n_pool, pfx_batch_stats* stats, char* err, size_t err_cap);

extern "C" int pfx_batch_pipeline(const int* devices, uint32_t n_devices, uint32_t n_images, const pfx_batch_params* params,
                                  const uint8_t* const* image_pool_host, uint32_t n_pool, pfx_batch_stats* stats, char* err, size_t err_cap)
{
    auto say = [&](const char* m) { if (err && err_cap) { std::strncpy(err, m, err_cap - 1); err[err_cap - 1] = 0; } };
    try { return batch_pipeline_impl(devices, n_devices, n_images, params, image_pool_host, n_pool, stats, err, err_cap); }
    catch (const std::bad_alloc&) { say("out of host memory"); return PFX_ERR_OOM; }
    catch (const std::system_error& e) { say(e.what()); return PFX_ERR_HIP; } // std::thread could not start
    catch (...) { say("unexpected exception"); return PFX_ERR_HIP; }
}

static int batch_pipeline_impl(const int* devices, uint32_t n_devices, uint32_t n_images, const pfx_batch_params* params,
                               const uint8_t* const* image_pool_host, uint32_t n_pool, pfx_batch_stats* stats, char* err, size_t err_cap)
{
    auto say = [&](const std::string& m) { if (err && err_cap) { std::strncpy(err, m.c_str(), err_cap - 1); err[err_cap - 1] = 0; } };
    if (!devices || n_devices == 0 || n_devices > 64 || !params || !image_pool_host || n_pool == 0 || !stats) { say("bad arguments"); return PFX_ERR_INVALID; }
    const pfx_batch_params& P = *params;
    if (P.w == 0 || P.h == 0 || P.n_overlays > 8 || (P.n_overlays && (!P.overlays_host || !P.overlay_modes)) || (P.n_keep && (!P.keep_indices || !P.keep_out))) {
        say("bad batch parameters");
        return PFX_ERR_INVALID;
    }
    std::memset(stats, 0, sizeof *stats);
    const size_t bytes = (size_t)P.w * P.h * 4;
    // the source pool is read by DMA: pin it for the duration of the call (a caller that already pinned it gets "already registered")
    std::vector<bool> registered(n_pool, false);
    for (uint32_t k = 0; k < n_pool; ++k) {
        const hipError_t e = hipHostRegister(const_cast<uint8_t*>(image_pool_host[k]), bytes, hipHostRegisterPortable);
        registered[k] = (e == hipSuccess);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    std::vector<worker_result> R(n_devices);
    std::vector<std::thread> th;
    th.reserve(n_devices);
    try {
        for (uint32_t d = 0; d < n_devices; ++d)
            th.emplace_back(run_device, devices[d], d, n_devices, n_images, std::cref(P), image_pool_host, n_pool, std::ref(R[d]));
    } catch (...) { // a thread could not start: the ones that did must be joined before the exception travels on
        for (auto& t : th) t.join();
        for (uint32_t k = 0; k < n_pool; ++k)
            if (registered[k]) (void)hipHostUnregister(const_cast<uint8_t*>(image_pool_host[k]));
        throw;
    }
    for (auto& t : th) t.join();
    for (uint32_t k = 0; k < n_pool; ++k)
        if (registered[k]) (void)hipHostUnregister(const_cast<uint8_t*>(image_pool_host[k]));

    int status = PFX_OK;
    double kms = 0.0; uint32_t ks = 0; uint64_t images = 0;
    auto t_lo = std::chrono::steady_clock::time_point::max(), t_hi = std::chrono::steady_clock::time_point::min();
    for (auto& r : R) {
        if (r.status != PFX_OK && status == PFX_OK) { status = r.status; say(r.err); }
        kms += r.kernel_ms; ks += r.kernel_samples; images += r.images;
        if (r.images) { t_lo = std::min(t_lo, r.t_begin); t_hi = std::max(t_hi, r.t_end); }
    }
    const double sec = images ? std::chrono::duration<double>(t_hi - t_lo).count() : 0.0; // devices start together: setup is before t_begin
    stats->seconds = sec;
    stats->images = (uint32_t)images;
    stats->devices = n_devices;
    stats->images_per_s = sec > 0 ? (double)images / sec : 0.0;
    stats->h2d_gbs = sec > 0 ? (double)images * bytes / sec / 1e9 : 0.0; // whole job, each direction
    stats->d2h_gbs = stats->h2d_gbs;
    stats->kernel_ms_per_image = ks ? kms / ks : 0.0;
    return status;
}
