// pfx_script_host.cpp — the script front-end's host API (B5): every function the reference registers with its Rhai engine
// (src/ops/scripting.rs:323-1482), bound to the device kernels.  The language runtime is pfx_rhai.cpp.
//
//   canvas info   :323-349    width height is_selected
//   pixel access  :355-435    get_pixel set_pixel get_r/g/b/a set_r/g/b/a           (host mirror of the image, synced lazily)
//   bulk iterators:437-609    for_each_pixel for_region map_channels                  (closure -> bytecode -> one GPU kernel)
//   transforms    :640-819    flip_* rotate_* resize_canvas resize_image
//   effects       :822-1165   apply_* — `_core` flavour through the effect kernels, inline flavour through pfx_rhai_adjust
//   utility       :1171-1350  print_line sleep progress rand_* clamp lerp distance abs min max floor ... rgb_to_hsl hsl_to_rgb
//   selection     :1356-1481  select_rect select_ellipse clear_selection has_selection invert_selection fill_selected delete_selected
//
// Residency: the image lives on the device in one of two ping-pong buffers (ctx->st_in / st_out) for the whole script; the
// selection mask lives in ctx->st_mask.  Host mirrors exist only while scalar accessors are in use.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "pfx_internal.h"
#include "pfx_rhai.h"

using rhai::Value;

struct pfx_script_output {
    std::vector<uint8_t> pixels;
    uint32_t w = 0, h = 0;
    std::vector<std::string> console;
    std::vector<pfx_canvas_op> ops;
};

namespace {

using VT = Value::T;

struct ScriptHost : rhai::Host {
    pfx_ctx* ctx = nullptr; // NULL: language-only check mode
    uint32_t w = 0, h = 0;
    pfx_devbuf* cur = nullptr;
    pfx_devbuf* other = nullptr;
    std::vector<uint8_t> host_px;
    bool host_valid = false, dev_stale = false;
    bool has_mask = false, host_mask_valid = false;
    std::vector<uint8_t> host_mask;
    uint64_t rng = 0;
    std::vector<pfx_canvas_op> ops;

    size_t bytes() const { return (size_t)w * h * 4; }
    const void* d_mask() const { return has_mask ? ctx->st_mask.p : nullptr; }

    // Consecutive Rhai-inline effects (apply_exposure(); apply_sepia(); apply_invert(); ...) are QUEUED and run as one pass over the image when something
    // needs it — another kind of effect, a pixel read, the end of the script (pfx_chain_dev: each op still re-quantises to u8 in registers, so the result is
    // what the one-launch-per-call form gives).  On the CPU every such call is a full pass over the image (scripting.rs:869-1075).
    std::vector<pfx_chain_op> pending;
    int dev_ready()
    {
        if (dev_stale) {
            PFX_TRY(pfx_reserve(ctx, *cur, bytes()));
            PFX_TRY(pfx_h2d(ctx, cur->p, host_px.data(), bytes()));
            PFX_TRY(pfx_sync(ctx)); // host_px may be modified again right away
            dev_stale = false;
        }
        if (!pending.empty()) {
            const std::vector<pfx_chain_op> run(std::move(pending));
            pending.clear();
            PFX_TRY(pfx_chain_dev(ctx, cur->p, cur->p, w, h, run.data(), (uint32_t)run.size()));
            image_changed_on_device();
        }
        return PFX_OK;
    }
    int host_ready()
    {
        if (!pending.empty()) PFX_TRY(dev_ready());   // queued effects first: the mirror must show their result
        if (!host_valid) {
            host_px.resize(bytes());
            PFX_TRY(pfx_d2h(ctx, host_px.data(), cur->p, bytes()));
            PFX_TRY(pfx_sync(ctx));
            host_valid = true;
        }
        return PFX_OK;
    }
    int mask_host_ready()
    {
        if (has_mask && !host_mask_valid) {
            host_mask.resize((size_t)w * h);
            PFX_TRY(pfx_d2h(ctx, host_mask.data(), ctx->st_mask.p, (size_t)w * h));
            PFX_TRY(pfx_sync(ctx));
            host_mask_valid = true;
        }
        return PFX_OK;
    }
    void image_changed_on_device() { host_valid = false; }

    // dst = f(src) through the ping-pong pair
    template <class F>
    int pingpong(F&& f, uint32_t new_w = 0, uint32_t new_h = 0)
    {
        PFX_TRY(dev_ready());
        const uint32_t ow = new_w ? new_w : w, oh = new_h ? new_h : h;
        PFX_TRY(pfx_reserve(ctx, *other, (size_t)ow * oh * 4));
        PFX_TRY(f(cur->p, other->p));
        std::swap(cur, other);
        w = ow;
        h = oh;
        image_changed_on_device();
        return PFX_OK;
    }
    template <class F>
    int inplace(F&& f)
    {
        PFX_TRY(dev_ready());
        PFX_TRY(f(cur->p));
        image_changed_on_device();
        return PFX_OK;
    }

    int run_closure(rhai::Interp& in, const rhai::Closure& c, int n_params, int x0, int y0, int x1, int y1, rhai::Error& err)
    {
        rhai::BcProgram prog;
        if (!in.compile_closure(c, n_params, (int64_t)w, (int64_t)h, prog, err)) return err.status;
        if (x0 >= x1 || y0 >= y1) return PFX_OK; // nothing to visit
        PFX_TRY(dev_ready());
        PFX_TRY(pfx_reserve(ctx, *other, bytes()));
        const size_t consts_bytes = std::max<size_t>(prog.consts.size(), 1) * 8, code_bytes = prog.code.size() * sizeof(rhai::BcIns);
        PFX_TRY(pfx_reserve(ctx, ctx->d_misc, 8 + consts_bytes + code_bytes));
        uint8_t* base = (uint8_t*)ctx->d_misc.p;
        const unsigned long long none = ~0ull;
        PFX_TRY(pfx_h2d(ctx, base, &none, 8));
        if (!prog.consts.empty()) PFX_TRY(pfx_h2d(ctx, base + 8, prog.consts.data(), prog.consts.size() * 8));
        PFX_TRY(pfx_h2d(ctx, base + 8 + consts_bytes, prog.code.data(), code_bytes));
        const bool full = x0 == 0 && y0 == 0 && x1 == (int)w && y1 == (int)h;
        if (!full) PFX_HIP(ctx, hipMemcpyAsync(other->p, cur->p, bytes(), hipMemcpyDeviceToDevice, ctx->stream));
        pfxk_vm_args A{};
        A.src = (const uint32_t*)cur->p;
        A.dst = (uint32_t*)other->p;
        A.mask = (const uint8_t*)d_mask();
        A.code = base + 8 + consts_bytes;
        A.consts = (const uint64_t*)(base + 8);
        A.err = (unsigned long long*)base;
        A.n_code = (int)prog.code.size();
        A.n_pre = prog.n_pre;
        A.n_regs = prog.n_regs;
        A.n_params = n_params;
        for (const rhai::BcIns& ins : prog.code) // f64 libm routines live in the heavier of the two kernel instantiations
            if (ins.op == rhai::BC_FMOD || ins.op == rhai::BC_FPOW || ins.op == rhai::BC_FSIN || ins.op == rhai::BC_FCOS || ins.op == rhai::BC_FTAN ||
                ins.op == rhai::BC_FATAN2 || ins.op == rhai::BC_FEXP || ins.op == rhai::BC_FLN) A.heavy = 1;
        A.w = (int)w; A.h = (int)h;
        A.x0 = x0; A.y0 = y0; A.x1 = x1; A.y1 = y1;
        // Operation budget.  The reference counts a closure's operations against the script's single 50 M budget (set_max_operations,
        // scripting.rs:288; closures run through call_within_context), serially, and can be cancelled between operations.  A kernel
        // launch cannot be cancelled, so the budget is made launch-wide up front: what is left of the 50 M is divided over the pixels
        // of the call, with a floor of 4096 steps per pixel so that ordinary closures on large images — which the reference's serial
        // interpreter could never finish — still run.  A runaway loop therefore costs at most 4096 steps per pixel before the launch
        // reports 'Too many operations'; scripts whose closures need more than the floor AND more than the reference's budget per
        // pixel fail here as they do there.  (Remaining divergence, see pfx_rhai.h: cheap closures on big images succeed here.)
        {
            const uint64_t used = in.ops(), left = used < 50000000ull ? 50000000ull - used : 0ull;
            const uint64_t n_px = (uint64_t)(x1 - x0) * (uint64_t)(y1 - y0);
            const uint64_t share = n_px ? left / n_px : left;
            A.step_budget = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(share, 4096ull), 4000000ull);
        }
        {
            pfx_timer t(ctx, "script_vm");
            PFX_HIP(ctx, pfxk_vm_run(ctx->stream, &A));
        }
        unsigned long long e = 0;
        PFX_TRY(pfx_d2h(ctx, &e, base, 8));
        PFX_TRY(pfx_sync(ctx)); // also keeps `prog` alive until the copies above are done
        if (e != none) { // the closure works on a copy that is written back only on success (scripting.rs:446-491)
            err.msg = rhai::Interp::bc_error_text((int)((e >> 16) & 0xff));
            err.line = (int)(e & 0xffff);
            err.col = 0;
            err.status = PFX_ERR_SCRIPT;
            return PFX_ERR_SCRIPT;
        }
        std::swap(cur, other);
        image_changed_on_device();
        return PFX_OK;
    }

    uint64_t next_rand() // xorshift64 (scripting.rs:1223-1227)
    {
        uint64_t s = rng;
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        rng = s;
        return s;
    }

    int call(rhai::Interp& in, const std::string& name, std::vector<Value>& a, Value& out, rhai::Error& err) override;
};

inline uint32_t i64_as_u32(int64_t v) { return (uint32_t)(uint64_t)v; } // Rust `as u32` on an i64 truncates
inline uint8_t clamp_u8(int64_t v) { return (uint8_t)std::min<int64_t>(std::max<int64_t>(v, 0), 255); }

std::string lower(std::string s) { for (char& c : s) c = (char)std::tolower((unsigned char)c); return s; }

int ScriptHost::call(rhai::Interp& in, const std::string& name, std::vector<Value>& a, Value& out, rhai::Error& err)
{
    const size_t n = a.size();
    auto sig = [&](std::initializer_list<VT> want) {
        if (want.size() != n) return false;
        size_t k = 0;
        for (VT t : want) if (a[k++].t != t) return false;
        return true;
    };
    bool known = false; // name exists with another signature
    auto dev = [&](int st) { // device status -> script error
        if (st != PFX_OK && err.msg.empty()) { err.msg = ctx ? pfx_last_error(ctx) : "device error"; err.status = st; }
        return 2;
    };
    auto need_image = [&]() {
        if (ctx) return false;
        err.msg = "'" + name + "' needs an image: not available in pfx_script_check";
        err.status = PFX_ERR_UNSUPPORTED;
        return true;
    };
    const int I = VT::Int, F = VT::Float;
    (void)I; (void)F;
#define FN(NAME) if (name == NAME && (known = true))
    // ---------------------------------------------------------------- canvas info
    FN("width") if (sig({})) { out = Value::from_int(w); return 2; }
    FN("height") if (sig({})) { out = Value::from_int(h); return 2; }
    FN("is_selected") if (sig({VT::Int, VT::Int})) {
        if (need_image()) return 2;
        const int64_t x = a[0].i, y = a[1].i;
        if (x < 0 || y < 0 || x >= (int64_t)w || y >= (int64_t)h) { out = Value::from_bool(false); return 2; }
        if (!has_mask) { out = Value::from_bool(true); return 2; }
        if (mask_host_ready() != PFX_OK) return dev(PFX_ERR_HIP);
        out = Value::from_bool(host_mask[(size_t)y * w + (size_t)x] > 0);
        return 2;
    }
    // ---------------------------------------------------------------- pixel access (host mirror)
    FN("get_pixel") if (sig({VT::Int, VT::Int})) {
        if (need_image()) return 2;
        std::vector<Value> px(4, Value::from_int(0));
        const int64_t x = a[0].i, y = a[1].i;
        if (!(x < 0 || y < 0 || x >= (int64_t)w || y >= (int64_t)h)) {
            const int st = host_ready();
            if (st != PFX_OK) return dev(st);
            const uint8_t* p = &host_px[((size_t)y * w + (size_t)x) * 4];
            for (int c = 0; c < 4; ++c) px[c] = Value::from_int(p[c]);
        }
        out = Value::from_array(px);
        return 2;
    }
    FN("set_pixel") if (sig({VT::Int, VT::Int, VT::Int, VT::Int, VT::Int, VT::Int})) {
        if (need_image()) return 2;
        const int64_t x = a[0].i, y = a[1].i;
        if (x < 0 || y < 0 || x >= (int64_t)w || y >= (int64_t)h) return 2;
        const int st = host_ready();
        if (st != PFX_OK) return dev(st);
        uint8_t* p = &host_px[((size_t)y * w + (size_t)x) * 4];
        for (int c = 0; c < 4; ++c) p[c] = clamp_u8(a[2 + c].i);
        dev_stale = true;
        return 2;
    }
    for (int c = 0; c < 4; ++c) {
        static const char* getn[] = {"get_r", "get_g", "get_b", "get_a"};
        static const char* setn[] = {"set_r", "set_g", "set_b", "set_a"};
        FN(getn[c]) if (sig({VT::Int, VT::Int})) {
            if (need_image()) return 2;
            const int64_t x = a[0].i, y = a[1].i;
            if (x < 0 || y < 0 || x >= (int64_t)w || y >= (int64_t)h) { out = Value::from_int(0); return 2; }
            const int st = host_ready();
            if (st != PFX_OK) return dev(st);
            out = Value::from_int(host_px[((size_t)y * w + (size_t)x) * 4 + c]);
            return 2;
        }
        FN(setn[c]) if (sig({VT::Int, VT::Int, VT::Int})) {
            if (need_image()) return 2;
            const int64_t x = a[0].i, y = a[1].i;
            if (x < 0 || y < 0 || x >= (int64_t)w || y >= (int64_t)h) return 2;
            const int st = host_ready();
            if (st != PFX_OK) return dev(st);
            host_px[((size_t)y * w + (size_t)x) * 4 + c] = clamp_u8(a[2].i);
            dev_stale = true;
            return 2;
        }
    }
    // ---------------------------------------------------------------- bulk iterators
    FN("for_each_pixel") if (sig({VT::Fn})) { if (need_image()) return 2; return dev(run_closure(in, *a[0].fn, 6, 0, 0, (int)w, (int)h, err)); }
    FN("map_channels") if (sig({VT::Fn})) { if (need_image()) return 2; return dev(run_closure(in, *a[0].fn, 4, 0, 0, (int)w, (int)h, err)); }
    FN("for_region") if (sig({VT::Int, VT::Int, VT::Int, VT::Int, VT::Fn})) { // scripting.rs:513-516
        if (need_image()) return 2;
        const int64_t rx = a[0].i, ry = a[1].i;
        const uint32_t x0 = i64_as_u32(std::max<int64_t>(rx, 0)), y0 = i64_as_u32(std::max<int64_t>(ry, 0));
        const uint32_t x1 = std::min(i64_as_u32((int64_t)((uint64_t)rx + (uint64_t)a[2].i)), w), y1 = std::min(i64_as_u32((int64_t)((uint64_t)ry + (uint64_t)a[3].i)), h);
        const bool empty = x0 >= x1 || y0 >= y1;
        return dev(run_closure(in, *a[4].fn, 6, empty ? 0 : (int)x0, empty ? 0 : (int)y0, empty ? 0 : (int)x1, empty ? 0 : (int)y1, err));
    }
    // ---------------------------------------------------------------- transforms
    struct { const char* n; int mode; int op; } perms[] = {
        {"flip_horizontal", 0, -1}, {"flip_vertical", 1, -1}, {"rotate_180", 2, -1}, {"flip_canvas_horizontal", 0, PFX_CANVAS_FLIP_HORIZONTAL},
        {"flip_canvas_vertical", 1, PFX_CANVAS_FLIP_VERTICAL}, {"rotate_canvas_90cw", 3, PFX_CANVAS_ROTATE_90CW},
        {"rotate_canvas_90ccw", 4, PFX_CANVAS_ROTATE_90CCW}, {"rotate_canvas_180", 2, PFX_CANVAS_ROTATE_180}};
    for (const auto& p : perms)
        FN(p.n) if (sig({})) {
            if (need_image()) return 2;
            const uint32_t sw = w, sh = h;
            const int st = pingpong([&](const void* s, void* d) {
                pfx_timer t(ctx, "permute");
                PFX_HIP(ctx, pfxk_permute(ctx->stream, (const uint8_t*)s, (uint8_t*)d, p.mode, sw, sh));
                return (int)PFX_OK;
            }, p.mode >= 3 ? sh : sw, p.mode >= 3 ? sw : sh);
            if (st == PFX_OK && p.op >= 0) ops.push_back({p.op, 0, 0, 0, 0});
            return dev(st);
        }
    FN("resize_canvas") if (sig({VT::Int, VT::Int, VT::Str})) { // scripting.rs:781-818
        if (need_image()) return 2;
        const uint32_t nw = std::min(i64_as_u32(std::max<int64_t>(a[0].i, 1)), 32768u), nh = std::min(i64_as_u32(std::max<int64_t>(a[1].i, 1)), 32768u);
        const std::string an = lower(*a[2].s);
        uint32_t ax = 0, ay = 0; // parse_anchor, scripting.rs:69-82
        if (an == "top-center" || an == "tc" || an == "n" || an == "top") { ax = 1; ay = 0; }
        else if (an == "top-right" || an == "tr" || an == "ne") { ax = 2; ay = 0; }
        else if (an == "center-left" || an == "cl" || an == "w" || an == "left") { ax = 0; ay = 1; }
        else if (an == "center" || an == "c" || an == "middle") { ax = 1; ay = 1; }
        else if (an == "center-right" || an == "cr" || an == "e" || an == "right") { ax = 2; ay = 1; }
        else if (an == "bottom-left" || an == "bl" || an == "sw") { ax = 0; ay = 2; }
        else if (an == "bottom-center" || an == "bc" || an == "s" || an == "bottom") { ax = 1; ay = 2; }
        else if (an == "bottom-right" || an == "br" || an == "se") { ax = 2; ay = 2; }
        const uint32_t ow = w, oh = h;
        const int32_t off_x = ax == 0 ? 0 : (ax == 1 ? ((int32_t)nw - (int32_t)ow) / 2 : (int32_t)nw - (int32_t)ow);
        const int32_t off_y = ay == 0 ? 0 : (ay == 1 ? ((int32_t)nh - (int32_t)oh) / 2 : (int32_t)nh - (int32_t)oh);
        const int st = pingpong([&](const void* s, void* d) {
            pfx_timer t(ctx, "resize_canvas");
            PFX_HIP(ctx, pfxk_recanvas(ctx->stream, (const uint8_t*)s, (uint8_t*)d, ow, oh, nw, nh, off_x, off_y, 0u)); // RgbaImage::new: transparent
            return (int)PFX_OK;
        }, nw, nh);
        if (st == PFX_OK) {
            ops.push_back({PFX_CANVAS_RESIZE_CANVAS, nw, nh, ax, ay});
            if (has_mask && (nw != ow || nh != oh)) { has_mask = false; host_mask_valid = false; } // a mask of the old size no longer applies
        }
        return dev(st);
    }
    FN("resize_image") if (sig({VT::Int, VT::Int, VT::Str})) { // scripting.rs:749-770
        if (need_image()) return 2;
        const uint32_t nw = std::min(i64_as_u32(std::max<int64_t>(a[0].i, 1)), 32768u), nh = std::min(i64_as_u32(std::max<int64_t>(a[1].i, 1)), 32768u);
        const std::string m = lower(*a[2].s); // parse_script_filter, scripting.rs:60-67
        const int filter = (m == "nearest" || m == "nn") ? PFX_RESIZE_NEAREST
                         : (m == "bicubic" || m == "catmull" || m == "catmullrom") ? PFX_RESIZE_BICUBIC
                         : (m == "lanczos" || m == "lanczos3") ? PFX_RESIZE_LANCZOS3 : PFX_RESIZE_BILINEAR;
        if (nw == w && nh == h) return 2;
        const uint32_t ow = w, oh = h;
        const int st = pingpong([&](const void* s, void* d) { return pfx_resize_image_dev(ctx, s, ow, oh, d, nw, nh, filter); }, nw, nh);
        if (st == PFX_OK) {
            ops.push_back({PFX_CANVAS_RESIZE_IMAGE, nw, nh, (uint32_t)filter, 0});
            has_mask = false; // a mask of the old size no longer applies
            host_mask_valid = false;
        }
        return dev(st);
    }
    // ---------------------------------------------------------------- effects: `_core` flavour, selection aware
#define FX(NAME, SIG, CALL)                                                                              \
    FN(NAME) if (sig SIG) {                                                                              \
        if (need_image()) return 2;                                                                      \
        return dev(pingpong([&](const void* s, void* d) { return CALL; }));                              \
    }
    // `apply_gaussian_blur` is BASELINE.json config 1's spelling of the reference's `apply_blur` (registered at scripting.rs:825-829):
    // an alias, so that the literal config-1 script runs
    if ((name == "apply_blur" || name == "apply_gaussian_blur") && (known = true)) if (sig({VT::Float})) { // :825 blur_with_selection_pub(img, sigma as f32, mask)
        if (need_image()) return 2;
        if (has_mask) { const int st = mask_host_ready(); if (st != PFX_OK) return dev(st); }
        return dev(pingpong([&](const void* s, void* d) { return pfx_int_blur_with_selection_dev(ctx, s, d, w, h, (float)a[0].f, has_mask ? host_mask.data() : nullptr, d_mask()); }));
    }
    FX("apply_box_blur", ({VT::Int}), pfx_box_blur_dev(ctx, s, d, w, h, (float)a[0].i, d_mask(), nullptr))                                   // :832
    FX("apply_motion_blur", ({VT::Float, VT::Float}), pfx_motion_blur_dev(ctx, s, d, w, h, (float)a[0].f, (float)a[1].f, d_mask()))          // :839
    FX("apply_sharpen", ({VT::Float}), pfx_sharpen_dev(ctx, s, d, w, h, (float)a[0].f, 1.0f, d_mask()))                                      // :847
    FX("apply_reduce_noise", ({VT::Float}), pfx_reduce_noise_dev(ctx, s, d, w, h, (float)a[0].f, 2, d_mask()))                               // :854
    FX("apply_median", ({VT::Int}), pfx_median_dev(ctx, s, d, w, h, i64_as_u32(std::max<int64_t>(a[0].i, 1)), d_mask()))                      // :861
    FX("apply_noise", ({VT::Float, VT::Bool}), pfx_add_noise_dev(ctx, s, d, w, h, (float)a[0].f, PFX_NOISE_GAUSSIAN, a[1].b, 42, 1.0f, 1, d_mask())) // :1079
    FX("apply_pixelate", ({VT::Int}), pfx_pixelate_dev(ctx, s, d, w, h, i64_as_u32(std::max<int64_t>(a[0].i, 1)), d_mask()))                  // :1096
    FX("apply_crystallize", ({VT::Int}), pfx_crystallize_dev(ctx, s, d, w, h, (float)std::max<int64_t>(a[0].i, 1), 42, d_mask()))             // :1103
    FX("apply_bulge", ({VT::Float}), pfx_bulge_dev(ctx, s, d, w, h, (float)a[0].f, 0.5f, 0.5f, d_mask()))                                    // :1110
    FX("apply_twist", ({VT::Float}), pfx_twist_dev(ctx, s, d, w, h, (float)a[0].f, 0.5f, 0.5f, d_mask()))                                    // :1117
    FX("apply_glow", ({VT::Float, VT::Float}), pfx_glow_dev(ctx, s, d, w, h, (float)a[0].f, (float)a[1].f, d_mask()))                        // :1125
    FX("apply_vignette", ({VT::Float, VT::Float}), pfx_vignette_dev(ctx, s, d, w, h, (float)a[0].f, (float)a[1].f, d_mask()))                // :1132
    FX("apply_halftone", ({VT::Float}), pfx_halftone_dev(ctx, s, d, w, h, (float)a[0].f, 45.0f, PFX_HALFTONE_CIRCLE, d_mask()))              // :1139
    FX("apply_ink", ({VT::Float, VT::Float}), pfx_ink_dev(ctx, s, d, w, h, (float)a[0].f, (float)a[1].f, d_mask()))                          // :1153
    FX("apply_oil_painting", ({VT::Int}), pfx_oil_painting_dev(ctx, s, d, w, h, i64_as_u32(std::max<int64_t>(a[0].i, 1)), 20, d_mask()))      // :1160
#undef FX
    // ---------------------------------------------------------------- effects: Rhai-inline flavour (truncating, mask ignored; :869-1075)
    auto inline_fx = [&](int op, const float* p, uint32_t np) {
        if (need_image()) return 2;
        pfx_chain_op o{};
        o.kind = PFX_CHAIN_RHAI; o.op = op; o.n_params = np;
        for (uint32_t k = 0; k < np; ++k) o.params[k] = p[k];
        pending.push_back(o);   // runs with its neighbours in one pass (dev_ready)
        return 2;
    };
    FN("apply_invert") if (sig({})) return inline_fx(PFX_RHAI_INVERT, nullptr, 0);
    FN("apply_desaturate") if (sig({})) return inline_fx(PFX_RHAI_DESATURATE, nullptr, 0);
    FN("apply_sepia") {
        if (sig({})) return inline_fx(PFX_RHAI_SEPIA, nullptr, 0);
        if (sig({VT::Float})) { const float p[1] = {(float)std::min(std::max(a[0].f, 0.0), 1.0)}; return inline_fx(PFX_RHAI_SEPIA_STRENGTH, p, 1); } // :923
    }
    FN("apply_brightness_contrast") if (sig({VT::Float, VT::Float})) { const float p[2] = {(float)a[0].f, (float)a[1].f}; return inline_fx(PFX_RHAI_BRIGHTNESS_CONTRAST, p, 2); }
    FN("apply_hsl") if (sig({VT::Float, VT::Float, VT::Float})) { const float p[3] = {(float)a[0].f, (float)a[1].f, (float)a[2].f}; return inline_fx(PFX_RHAI_HSL, p, 3); }
    FN("apply_exposure") if (sig({VT::Float})) { const float p[1] = {(float)a[0].f}; return inline_fx(PFX_RHAI_EXPOSURE, p, 1); }
    FN("apply_levels") if (sig({VT::Float, VT::Float, VT::Float})) { const float p[3] = {(float)a[0].f, (float)a[1].f, (float)a[2].f}; return inline_fx(PFX_RHAI_LEVELS, p, 3); }
    // ---------------------------------------------------------------- utility
    FN("print_line") if (sig({VT::Str})) { in.console.push_back(*a[0].s); return 2; }
    FN("sleep") if (sig({VT::Int})) return 2;      // :1191 preview pause: there is no preview consumer in the headless back-end
    FN("progress") if (sig({VT::Float})) return 2; // :1208 progress bar only
    FN("rand_int") if (sig({VT::Int, VT::Int})) {
        const int64_t lo = a[0].i, hi = a[1].i;
        if (lo >= hi) { out = Value::from_int(lo); return 2; }
        const uint64_t s = next_rand();
        const uint64_t range = (uint64_t)hi - (uint64_t)lo;
        out = Value::from_int((int64_t)((uint64_t)lo + (uint64_t)(int64_t)(s % std::max<uint64_t>(range, 1))));
        return 2;
    }
    FN("rand_float") {
        if (sig({VT::Float, VT::Float})) {
            if (a[0].f >= a[1].f) { out = Value::from_float(a[0].f); return 2; }
            out = Value::from_float(a[0].f + ((double)next_rand() / (double)UINT64_MAX) * (a[1].f - a[0].f));
            return 2;
        }
        if (sig({})) { out = Value::from_float((double)next_rand() / (double)UINT64_MAX); return 2; }
    }
    FN("clamp") if (sig({VT::Int, VT::Int, VT::Int})) {
        if (a[1].i > a[2].i) { err.msg = "clamp: min > max"; return 2; } // i64::clamp asserts min <= max
        out = Value::from_int(std::min(std::max(a[0].i, a[1].i), a[2].i));
        return 2;
    }
    FN("clamp_f") if (sig({VT::Float, VT::Float, VT::Float})) { double v = a[0].f; if (v < a[1].f) v = a[1].f; if (v > a[2].f) v = a[2].f; out = Value::from_float(v); return 2; }
    FN("lerp") if (sig({VT::Float, VT::Float, VT::Float})) { out = Value::from_float(a[0].f + (a[1].f - a[0].f) * a[2].f); return 2; }
    FN("distance") if (sig({VT::Float, VT::Float, VT::Float, VT::Float})) {
        const double dx = a[2].f - a[0].f, dy = a[3].f - a[1].f;
        out = Value::from_float(std::sqrt(dx * dx + dy * dy));
        return 2;
    }
    FN("abs") {
        if (sig({VT::Float})) { out = Value::from_float(std::fabs(a[0].f)); return 2; }
        if (sig({VT::Int})) { out = Value::from_int(a[0].i < 0 ? (int64_t)(0ull - (uint64_t)a[0].i) : a[0].i); return 2; }
    }
    for (const char* nm : {"min", "min_i"}) FN(nm) if (sig({VT::Int, VT::Int})) { out = Value::from_int(std::min(a[0].i, a[1].i)); return 2; }
    for (const char* nm : {"max", "max_i"}) FN(nm) if (sig({VT::Int, VT::Int})) { out = Value::from_int(std::max(a[0].i, a[1].i)); return 2; }
    for (const char* nm : {"min", "min_f"}) FN(nm) if (sig({VT::Float, VT::Float})) { out = Value::from_float(std::fmin(a[0].f, a[1].f)); return 2; }
    for (const char* nm : {"max", "max_f"}) FN(nm) if (sig({VT::Float, VT::Float})) { out = Value::from_float(std::fmax(a[0].f, a[1].f)); return 2; }
    FN("abs_i") if (sig({VT::Int})) { out = Value::from_int(a[0].i < 0 ? (int64_t)(0ull - (uint64_t)a[0].i) : a[0].i); return 2; }
    {
        struct { const char* n; double (*f)(double); } f1[] = {{"floor", std::floor}, {"ceil", std::ceil}, {"round", std::round}, {"sqrt", std::sqrt},
                                                               {"sin", std::sin}, {"cos", std::cos}, {"tan", std::tan}};
        for (const auto& e : f1) FN(e.n) if (sig({VT::Float})) { out = Value::from_float(e.f(a[0].f)); return 2; }
    }
    FN("pow") if (sig({VT::Float, VT::Float})) { out = Value::from_float(std::pow(a[0].f, a[1].f)); return 2; }
    FN("atan2") if (sig({VT::Float, VT::Float})) { out = Value::from_float(std::atan2(a[0].f, a[1].f)); return 2; }
    FN("PI") if (sig({})) { out = Value::from_float(3.14159265358979323846); return 2; }
    FN("rgb_to_hsl") if (sig({VT::Int, VT::Int, VT::Int})) { // :1295-1327
        const double rf = (double)clamp_u8(a[0].i) / 255.0, gf = (double)clamp_u8(a[1].i) / 255.0, bf = (double)clamp_u8(a[2].i) / 255.0;
        const double mx = std::fmax(std::fmax(rf, gf), bf), mn = std::fmin(std::fmin(rf, gf), bf);
        const double l = (mx + mn) / 2.0;
        if (std::fabs(mx - mn) < 1e-10) { out = Value::from_array({Value::from_float(0.0), Value::from_float(0.0), Value::from_float(l * 100.0)}); return 2; }
        const double d = mx - mn;
        const double s = l > 0.5 ? d / (2.0 - mx - mn) : d / (mx + mn);
        double hh;
        if (std::fabs(mx - rf) < 1e-10) hh = (gf - bf) / d + (gf < bf ? 6.0 : 0.0);
        else if (std::fabs(mx - gf) < 1e-10) hh = (bf - rf) / d + 2.0;
        else hh = (rf - gf) / d + 4.0;
        out = Value::from_array({Value::from_float(hh * 60.0), Value::from_float(s * 100.0), Value::from_float(l * 100.0)});
        return 2;
    }
    FN("hsl_to_rgb") if (sig({VT::Float, VT::Float, VT::Float})) { // :1329-1349
        const double s = a[1].f / 100.0, l = a[2].f / 100.0;
        const double c = (1.0 - std::fabs(2.0 * l - 1.0)) * s;
        const double h2 = a[0].f / 60.0;
        const double x = c * (1.0 - std::fabs(std::fmod(h2, 2.0) - 1.0));
        double r1, g1, b1;
        const int32_t sector = (h2 != h2) ? 0 : (h2 >= 2147483648.0 ? INT32_MAX : (h2 <= -2147483648.0 ? INT32_MIN : (int32_t)h2)); // `as i32`
        switch (sector) {
        case 0: r1 = c; g1 = x; b1 = 0; break;
        case 1: r1 = x; g1 = c; b1 = 0; break;
        case 2: r1 = 0; g1 = c; b1 = x; break;
        case 3: r1 = 0; g1 = x; b1 = c; break;
        case 4: r1 = x; g1 = 0; b1 = c; break;
        default: r1 = c; g1 = 0; b1 = x; break;
        }
        const double m = l - c / 2.0;
        auto to_i64 = [](double v) { v = std::round(v); return (v != v) ? (int64_t)0 : (v >= 9223372036854775808.0 ? INT64_MAX : (v <= -9223372036854775808.0 ? INT64_MIN : (int64_t)v)); };
        out = Value::from_array({Value::from_int(to_i64((r1 + m) * 255.0)), Value::from_int(to_i64((g1 + m) * 255.0)), Value::from_int(to_i64((b1 + m) * 255.0))});
        return 2;
    }
    // ---------------------------------------------------------------- selection (device-resident mask)
    auto mask_op = [&](int op, int x0, int y0, int x1, int y1, double cx, double cy, double rx2, double ry2) {
        int st = pfx_reserve(ctx, ctx->st_mask, (size_t)w * h);
        if (st == PFX_OK) {
            pfx_timer t(ctx, "mask_op");
            const hipError_t e = pfxk_mask_op(ctx->stream, (uint8_t*)ctx->st_mask.p, op, x0, y0, x1, y1, cx, cy, rx2, ry2, w, h);
            if (e != hipSuccess) st = pfx_fail(ctx, PFX_ERR_HIP, "pfxk_mask_op failed: %s", hipGetErrorString(e));
        }
        if (st == PFX_OK) { has_mask = true; host_mask_valid = false; }
        return dev(st);
    };
    FN("select_rect") if (sig({VT::Int, VT::Int, VT::Int, VT::Int})) { // :1359-1375
        if (need_image()) return 2;
        auto cl = [&](int64_t v, uint32_t lim) { return (int)std::min(i64_as_u32(std::max<int64_t>(v, 0)), lim); };
        return mask_op(0, cl(a[0].i, w), cl(a[1].i, h), cl(a[2].i, w), cl(a[3].i, h), 0, 0, 1, 1);
    }
    FN("select_ellipse") if (sig({VT::Float, VT::Float, VT::Float, VT::Float})) { // :1381-1399
        if (need_image()) return 2;
        return mask_op(1, 0, 0, 0, 0, a[0].f, a[1].f, std::fmax(a[2].f * a[2].f, 0.001), std::fmax(a[3].f * a[3].f, 0.001));
    }
    FN("clear_selection") if (sig({})) { has_mask = false; host_mask_valid = false; return 2; }
    FN("has_selection") if (sig({})) { out = Value::from_bool(has_mask); return 2; }
    FN("invert_selection") if (sig({})) { // :1418-1431
        if (need_image()) return 2;
        return has_mask ? mask_op(2, 0, 0, 0, 0, 0, 0, 1, 1) : mask_op(3, 0, 0, 0, 0, 0, 0, 1, 1);
    }
    auto fill = [&](uint32_t rgba) {
        if (need_image()) return 2;
        return dev(inplace([&](void* img) {
            pfx_timer t(ctx, "fill_selected");
            PFX_HIP(ctx, pfxk_fill_masked(ctx->stream, (uint8_t*)img, (const uint8_t*)d_mask(), rgba, w, h));
            return (int)PFX_OK;
        }));
    };
    FN("fill_selected") if (sig({VT::Int, VT::Int, VT::Int, VT::Int}))
        return fill((uint32_t)clamp_u8(a[0].i) | ((uint32_t)clamp_u8(a[1].i) << 8) | ((uint32_t)clamp_u8(a[2].i) << 16) | ((uint32_t)clamp_u8(a[3].i) << 24));
    FN("delete_selected") if (sig({})) return fill(0u);
#undef FN
    return known ? 1 : 0;
}

void fill_result(pfx_script_result* r, const rhai::Error* err, const std::vector<std::string>& console, uint64_t ops)
{
    if (!r) return;
    std::memset(r, 0, sizeof *r);
    r->ops_executed = (uint32_t)std::min<uint64_t>(ops, 0xffffffffu);
    std::string joined;
    for (const std::string& l : console) { joined += l; joined += '\n'; }
    std::snprintf(r->console, sizeof r->console, "%s", joined.c_str());
    if (err) {
        // ScriptError::friendly_message header (scripting.rs:97-115)
        std::string head = err->line > 0 ? "Error on line " + std::to_string(err->line) + (err->col > 0 ? ", column " + std::to_string(err->col) : "") + ":\n  "
                                         : "Script error:\n  ";
        std::snprintf(r->error, sizeof r->error, "%s%s", head.c_str(), err->msg.c_str());
        r->error_line = err->line;
        r->error_col = err->col;
    }
}

uint64_t time_seed() // scripting.rs:1745-1751
{
    const uint64_t ns = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    const uint64_t s = ns ^ 0x517cc1b727220a95ull;
    return s ? s : 0x517cc1b727220a95ull;
}

} // namespace

int pfx_int_script_run_dev(pfx_ctx* ctx, const char* source, uint32_t* w, uint32_t* h, const uint8_t* mask, pfx_script_result* result,
                           std::vector<std::string>* console, std::vector<pfx_canvas_op>* ops)
try {
    ScriptHost host;
    host.ctx = ctx;
    host.w = *w;
    host.h = *h;
    host.cur = &ctx->st_in;
    host.other = &ctx->st_out;
    host.rng = time_seed();
    if (mask) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_mask, (size_t)*w * *h));
        PFX_TRY(pfx_h2d(ctx, ctx->st_mask.p, mask, (size_t)*w * *h));
        host.has_mask = true;
        host.host_mask.assign(mask, mask + (size_t)*w * *h);
        host.host_mask_valid = true;
    }
    rhai::Interp in(&host);
    in.on_run_thread = [ctx] { (void)hipSetDevice(ctx->device); };   // the device binding is per thread; the host functions launch on ctx->stream
    rhai::Error err;
    const bool ok = in.run(source, err);
    int st = ok ? PFX_OK : (err.status ? err.status : PFX_ERR_SCRIPT);
    if (ok) { // the final image must be on the device, in ctx->st_in
        st = host.dev_ready();
        if (st == PFX_OK && host.cur != &ctx->st_in) std::swap(ctx->st_in, ctx->st_out);
        if (st != PFX_OK) err = {pfx_last_error(ctx), 0, 0, st};
    }
    fill_result(result, st == PFX_OK ? nullptr : &err, in.console, in.ops());
    if (console) *console = in.console;
    if (st != PFX_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        return pfx_fail(ctx, st, "%s", err.msg.c_str());
    }
    *w = host.w;
    *h = host.h;
    if (ops) *ops = host.ops;
    return PFX_OK;
} catch (const std::bad_alloc&) { // exception barrier: nothing may unwind through the C ABI
    return pfx_fail(ctx, PFX_ERR_OOM, "out of memory while running the script");
} catch (const std::exception& e) {
    return pfx_fail(ctx, PFX_ERR_SCRIPT, "internal error while running the script: %s", e.what());
}

extern "C" {

int pfx_script_execute(pfx_ctx* ctx, const char* source, const uint8_t* pixels, uint32_t w, uint32_t h, const uint8_t* mask, pfx_script_output** out,
                       pfx_script_result* result)
{
    if (result) std::memset(result, 0, sizeof *result);
    if (out) *out = nullptr;
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, source && pixels && out && w && h && (uint64_t)w * h <= 256000000ull, "pfx_script_execute: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const size_t bytes = (size_t)w * h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, bytes));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, pixels, bytes));
    std::unique_ptr<pfx_script_output> o(new pfx_script_output);
    o->w = w;
    o->h = h;
    PFX_TRY(pfx_int_script_run_dev(ctx, source, &o->w, &o->h, mask, result, &o->console, &o->ops));
    o->pixels.resize((size_t)o->w * o->h * 4);
    PFX_TRY(pfx_d2h(ctx, o->pixels.data(), ctx->st_in.p, o->pixels.size()));
    PFX_TRY(pfx_sync(ctx));
    *out = o.release();
    return PFX_OK;
}

const uint8_t* pfx_script_output_pixels(const pfx_script_output* out, uint32_t* w, uint32_t* h)
{
    if (!out) return nullptr;
    if (w) *w = out->w;
    if (h) *h = out->h;
    return out->pixels.data();
}
uint32_t pfx_script_output_console_lines(const pfx_script_output* out) { return out ? (uint32_t)out->console.size() : 0; }
const char* pfx_script_output_console_line(const pfx_script_output* out, uint32_t index)
{
    return (out && index < out->console.size()) ? out->console[index].c_str() : nullptr;
}
uint32_t pfx_script_output_canvas_ops(const pfx_script_output* out, pfx_canvas_op* ops, uint32_t capacity)
{
    if (!out) return 0;
    for (uint32_t k = 0; ops && k < capacity && k < out->ops.size(); ++k) ops[k] = out->ops[k];
    return (uint32_t)out->ops.size();
}
void pfx_script_output_free(pfx_script_output* out) { delete out; }

int pfx_script_run(pfx_ctx* ctx, const char* source, uint8_t* pixels_inout, uint32_t w, uint32_t h, const uint8_t* mask, pfx_script_result* result)
{
    pfx_script_output* o = nullptr;
    const int st = pfx_script_execute(ctx, source, pixels_inout, w, h, mask, &o, result);
    if (st != PFX_OK) return st;
    int rc = PFX_OK;
    if (o->w != w || o->h != h) rc = pfx_fail(ctx, PFX_ERR_UNSUPPORTED, "the script changed the image size to %ux%u: use pfx_script_execute", o->w, o->h);
    else std::memcpy(pixels_inout, o->pixels.data(), o->pixels.size()); // only reached on success: pixels untouched on error
    pfx_script_output_free(o);
    return rc;
}

int pfx_script_check(const char* source, uint32_t w, uint32_t h, pfx_script_result* result)
{
    return pfx_int_script_check_limited(source, w, h, result, 0);
}

// pfx_script_check with a lower operation budget (0 = the reference's 50 M): the hardening harness runs 10^5 mutated scripts, most of which loop
int pfx_int_script_check_limited(const char* source, uint32_t w, uint32_t h, pfx_script_result* result, uint64_t max_ops)
{
    if (result) std::memset(result, 0, sizeof *result);
    if (!source) return PFX_ERR_INVALID;
    ScriptHost host;
    host.w = w;
    host.h = h;
    host.rng = time_seed();
    rhai::Interp in(&host);
    if (max_ops) in.max_ops = max_ops;
    rhai::Error err;
    const bool ok = in.run(source, err);
    fill_result(result, ok ? nullptr : &err, in.console, in.ops());
    return ok ? PFX_OK : (err.status ? err.status : PFX_ERR_SCRIPT);
}

} // extern "C"
