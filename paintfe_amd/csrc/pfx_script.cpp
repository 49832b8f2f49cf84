// pfx_script.cpp — the Rhai Effect-API front-end (B5) and the batch CLI (B6) on top of the device kernels.
//
// Reference: execute_script_sync src/ops/scripting.rs:1733-1821; registered effect names, arities and numeric
// flavours src/ops/scripting.rs:822-1165; utility API :1170-1230; CLI flags, file loop, naming and exit codes
// src/cli.rs:43-427.
//
// Scope: the *call-statement subset* of Rhai — `name(literal, ...);` sequences with
// comments — which is what effect scripts such as `apply_blur(4.0);` consist of.  Rhai's typing rule is kept: an
// i64 literal does not match an f64 parameter ("Function not found: apply_blur (i64)").  The image stays on the
// device for the whole script: one upload, N kernels, one download.  Closures / variables / loops
// (`map_channels`, `for_each_pixel`, ...) need the full language runtime (rhai 1.25.1, a third-party crate) and are
// reported as PFX_ERR_UNSUPPORTED so the caller can fall back to its CPU interpreter.
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <glob.h>
#include <sys/stat.h>

#include "pfx_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------ tokenizer
enum class Tok { Ident, Int, Float, Str, Bool, LParen, RParen, Comma, Semi, End, Other };
struct Token {
    Tok t = Tok::End;
    std::string text;
    int64_t i = 0;
    double f = 0.0;
    bool b = false;
    int line = 1, col = 1;
};

struct ScriptErr {
    std::string msg;
    int line = 0, col = 0;
    int status = PFX_ERR_SCRIPT;
};

class Lexer {
public:
    explicit Lexer(const char* s) : p_(s) {}
    bool next(Token& out, ScriptErr& err)
    {
        skip_ws_comments();
        out = Token();
        out.line = line_;
        out.col = col_;
        const char c = *p_;
        if (c == '\0') { out.t = Tok::End; return true; }
        if (std::isalpha((unsigned char)c) || c == '_') {
            const char* s = p_;
            while (std::isalnum((unsigned char)*p_) || *p_ == '_') adv();
            out.text.assign(s, p_);
            if (out.text == "true" || out.text == "false") { out.t = Tok::Bool; out.b = out.text == "true"; }
            else out.t = Tok::Ident;
            return true;
        }
        if (std::isdigit((unsigned char)c) || ((c == '-' || c == '+') && std::isdigit((unsigned char)p_[1]))) {
            const char* s = p_;
            adv();
            bool is_float = false;
            while (std::isdigit((unsigned char)*p_) || *p_ == '_') adv();
            if (*p_ == '.' && std::isdigit((unsigned char)p_[1])) { is_float = true; adv(); while (std::isdigit((unsigned char)*p_) || *p_ == '_') adv(); }
            else if (*p_ == '.' && !std::isalpha((unsigned char)p_[1]) && p_[1] != '.') { is_float = true; adv(); } // `4.` is a float in Rhai
            if (*p_ == 'e' || *p_ == 'E') {
                const char* q = p_ + 1;
                if (*q == '+' || *q == '-') ++q;
                if (std::isdigit((unsigned char)*q)) { is_float = true; while (p_ < q) adv(); while (std::isdigit((unsigned char)*p_)) adv(); }
            }
            std::string num(s, p_);
            num.erase(std::remove(num.begin(), num.end(), '_'), num.end());
            if (is_float) { out.t = Tok::Float; out.f = std::strtod(num.c_str(), nullptr); }
            else { out.t = Tok::Int; out.i = std::strtoll(num.c_str(), nullptr, 10); }
            out.text = num;
            return true;
        }
        if (c == '"' || c == '`') {
            const char q = c;
            adv();
            std::string s;
            while (*p_ && *p_ != q) {
                if (*p_ == '\\' && p_[1]) {
                    adv();
                    switch (*p_) { case 'n': s += '\n'; break; case 't': s += '\t'; break; default: s += *p_; }
                    adv();
                } else { s += *p_; adv(); }
            }
            if (*p_ != q) { err = {"Open string is not terminated", out.line, out.col, PFX_ERR_SCRIPT}; return false; }
            adv();
            if (s.size() > 10000) { err = {"Length of string too large", out.line, out.col, PFX_ERR_SCRIPT}; return false; } // :291
            out.t = Tok::Str;
            out.text = s;
            return true;
        }
        adv();
        switch (c) {
        case '(': out.t = Tok::LParen; break;
        case ')': out.t = Tok::RParen; break;
        case ',': out.t = Tok::Comma; break;
        case ';': out.t = Tok::Semi; break;
        default: out.t = Tok::Other; out.text = std::string(1, c);
        }
        return true;
    }

private:
    void adv()
    {
        if (*p_ == '\n') { ++line_; col_ = 1; } else ++col_;
        ++p_;
    }
    void skip_ws_comments()
    {
        for (;;) {
            while (std::isspace((unsigned char)*p_)) adv();
            if (p_[0] == '/' && p_[1] == '/') { while (*p_ && *p_ != '\n') adv(); continue; }
            if (p_[0] == '/' && p_[1] == '*') {
                int depth = 0; // Rhai block comments nest
                do {
                    if (p_[0] == '/' && p_[1] == '*') { ++depth; adv(); adv(); }
                    else if (p_[0] == '*' && p_[1] == '/') { --depth; adv(); adv(); }
                    else adv();
                } while (*p_ && depth > 0);
                continue;
            }
            break;
        }
    }
    const char* p_;
    int line_ = 1, col_ = 1;
};

struct Arg {
    Tok t;
    int64_t i;
    double f;
    bool b;
    std::string s;
};
struct Call {
    std::string name;
    std::vector<Arg> args;
    int line, col;
};

const char* type_name(Tok t)
{
    switch (t) { case Tok::Int: return "i64"; case Tok::Float: return "f64"; case Tok::Bool: return "bool"; case Tok::Str: return "&str | ImmutableString | String"; default: return "?"; }
}

std::string signature(const Call& c)
{
    std::string s = c.name + " (";
    for (size_t k = 0; k < c.args.size(); ++k) { if (k) s += ", "; s += type_name(c.args[k].t); }
    return s + ")";
}

bool parse(const char* src, std::vector<Call>& calls, ScriptErr& err)
{
    Lexer lx(src);
    Token t;
    for (;;) {
        if (!lx.next(t, err)) return false;
        if (t.t == Tok::End) return true;
        if (t.t == Tok::Semi) continue; // empty statement
        if (t.t != Tok::Ident) {
            err = {"only Effect-API call statements are handled by the HIP back-end (found '" + t.text + "')", t.line, t.col, PFX_ERR_UNSUPPORTED};
            return false;
        }
        static const char* keywords[] = {"let", "const", "fn", "if", "else", "for", "while", "loop", "return", "switch", "import", "export", "do", "break", "continue"};
        for (const char* kw : keywords)
            if (t.text == kw) {
                err = {"Rhai statement '" + t.text + "' needs the full language runtime; only call statements run on the HIP back-end", t.line, t.col, PFX_ERR_UNSUPPORTED};
                return false;
            }
        Call c;
        c.name = t.text;
        c.line = t.line;
        c.col = t.col;
        if (!lx.next(t, err)) return false;
        if (t.t != Tok::LParen) {
            err = {"expected '(' after '" + c.name + "': only call statements run on the HIP back-end", t.line, t.col, PFX_ERR_UNSUPPORTED};
            return false;
        }
        if (!lx.next(t, err)) return false;
        while (t.t != Tok::RParen) {
            if (t.t == Tok::Int || t.t == Tok::Float || t.t == Tok::Bool || t.t == Tok::Str) c.args.push_back({t.t, t.i, t.f, t.b, t.text});
            else if (t.t == Tok::End) { err = {"Expecting ')' to close the parameters list of function call '" + c.name + "'", t.line, t.col, PFX_ERR_SCRIPT}; return false; }
            else {
                err = {"argument of '" + c.name + "' is not a literal; expressions need the full language runtime", t.line, t.col, PFX_ERR_UNSUPPORTED};
                return false;
            }
            if (!lx.next(t, err)) return false;
            if (t.t == Tok::Comma) { if (!lx.next(t, err)) return false; }
            else if (t.t == Tok::End) { err = {"Expecting ')' to close the parameters list of function call '" + c.name + "'", t.line, t.col, PFX_ERR_SCRIPT}; return false; }
            else if (t.t != Tok::RParen) { err = {"Expecting ',' to separate the parameters of function call '" + c.name + "'", t.line, t.col, PFX_ERR_SCRIPT}; return false; }
        }
        calls.push_back(c);
        if (!lx.next(t, err)) return false;
        if (t.t == Tok::End) return true;
        if (t.t != Tok::Semi) { err = {"Expecting ';' to terminate this statement", t.line, t.col, PFX_ERR_SCRIPT}; return false; }
    }
}

bool sig(const Call& c, std::initializer_list<Tok> want)
{
    if (c.args.size() != want.size()) return false;
    size_t k = 0;
    for (Tok w : want) if (c.args[k++].t != w) return false;
    return true;
}

// effect names that exist in the reference but are outside this back-end's scope
const char* kNotProvided[] = {"apply_reduce_noise", "apply_noise", "apply_crystallize", "apply_bulge",
                              "apply_twist", "apply_vignette", "apply_halftone", "apply_ink", "apply_oil_painting",
                              "for_each_pixel", "map_channels", "for_region", "get_pixel", "set_pixel", "flip_horizontal", "flip_vertical",
                              "rotate_180", "rotate_canvas_90cw", "rotate_canvas_90ccw", "rotate_canvas_180", "flip_canvas_horizontal",
                              "flip_canvas_vertical", "resize_image", "resize_canvas", "select_rect", "clear_selection", "invert_selection",
                              "fill_selected", "delete_selected"};

int run_calls(pfx_ctx* ctx, const std::vector<Call>& calls, void* d_img, void* d_tmp, uint32_t w, uint32_t h, const uint8_t* mask,
              const void* d_mask, std::string& console, uint32_t& ops, ScriptErr& err)
{
    void* cur = d_img;
    void* other = d_tmp;
    auto fail = [&](const Call& c, int status, const std::string& m) { err = {m, c.line, c.col, status}; return status; };
    for (const Call& c : calls) {
        ++ops;
        int st = PFX_OK;
        bool swapped = false;
        if (c.name == "apply_blur" && sig(c, {Tok::Float})) { // :825 blur_with_selection_pub(img, sigma as f32, mask)
            st = pfx_int_blur_with_selection_dev(ctx, cur, other, w, h, (float)c.args[0].f, mask, d_mask);
            swapped = true;
        } else if (c.name == "apply_box_blur" && sig(c, {Tok::Int})) { // :832 box_blur_core(img, radius as f32, mask)
            st = pfx_box_blur_dev(ctx, cur, other, w, h, (float)c.args[0].i, d_mask, nullptr);
            swapped = true;
        } else if (c.name == "apply_median" && sig(c, {Tok::Int})) { // :861 median_core(img, radius.max(1) as u32, mask)
            st = pfx_median_dev(ctx, cur, other, w, h, (uint32_t)std::max<int64_t>(c.args[0].i, 1), d_mask);
            swapped = true;
        } else if (c.name == "apply_pixelate" && sig(c, {Tok::Int})) { // :1096 pixelate_core(img, size.max(1) as u32, mask)
            st = pfx_pixelate_dev(ctx, cur, other, w, h, (uint32_t)std::max<int64_t>(c.args[0].i, 1), d_mask);
            swapped = true;
        } else if (c.name == "apply_motion_blur" && sig(c, {Tok::Float, Tok::Float})) { // :839 motion_blur_core(img, angle, distance, mask)
            st = pfx_motion_blur_dev(ctx, cur, other, w, h, (float)c.args[0].f, (float)c.args[1].f, d_mask);
            swapped = true;
        } else if (c.name == "apply_sharpen" && sig(c, {Tok::Float})) { // :847 sharpen_core(img, amount as f32, 1.0, mask)
            st = pfx_sharpen_dev(ctx, cur, other, w, h, (float)c.args[0].f, 1.0f, d_mask);
            swapped = true;
        } else if (c.name == "apply_glow" && sig(c, {Tok::Float, Tok::Float})) { // :1125 glow_core(img, radius, intensity, mask)
            st = pfx_glow_dev(ctx, cur, other, w, h, (float)c.args[0].f, (float)c.args[1].f, d_mask);
            swapped = true;
        } else if (c.name == "apply_invert" && sig(c, {})) {
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_INVERT, nullptr, 0);
        } else if (c.name == "apply_desaturate" && sig(c, {})) {
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_DESATURATE, nullptr, 0);
        } else if (c.name == "apply_sepia" && sig(c, {})) {
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_SEPIA, nullptr, 0);
        } else if (c.name == "apply_sepia" && sig(c, {Tok::Float})) { // strength.clamp(0,1) as f32 (:923)
            const float p[1] = {(float)std::min(std::max(c.args[0].f, 0.0), 1.0)};
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_SEPIA_STRENGTH, p, 1);
        } else if (c.name == "apply_brightness_contrast" && sig(c, {Tok::Float, Tok::Float})) {
            const float p[2] = {(float)c.args[0].f, (float)c.args[1].f};
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_BRIGHTNESS_CONTRAST, p, 2);
        } else if (c.name == "apply_hsl" && sig(c, {Tok::Float, Tok::Float, Tok::Float})) {
            const float p[3] = {(float)c.args[0].f, (float)c.args[1].f, (float)c.args[2].f};
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_HSL, p, 3);
        } else if (c.name == "apply_exposure" && sig(c, {Tok::Float})) {
            const float p[1] = {(float)c.args[0].f};
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_EXPOSURE, p, 1);
        } else if (c.name == "apply_levels" && sig(c, {Tok::Float, Tok::Float, Tok::Float})) {
            const float p[3] = {(float)c.args[0].f, (float)c.args[1].f, (float)c.args[2].f};
            st = pfx_rhai_adjust_dev(ctx, cur, w, h, PFX_RHAI_LEVELS, p, 3);
        } else if ((c.name == "print_line" || c.name == "print") && sig(c, {Tok::Str})) { // :1174
            console += c.args[0].s;
            console += '\n';
        } else if (c.name == "progress" && sig(c, {Tok::Float})) { // :1208: progress bar only
        } else if (c.name == "sleep" && sig(c, {Tok::Int})) {      // :1191: preview pause; nothing to show headless
        } else if ((c.name == "width" || c.name == "height" || c.name == "has_selection") && sig(c, {})) {
            // pure getters: a bare call statement has no effect
        } else {
            for (const char* n : kNotProvided)
                if (c.name == n)
                    return fail(c, PFX_ERR_UNSUPPORTED, "'" + c.name + "' is not provided by the HIP back-end (outside the accelerated path); use the CPU path");
            return fail(c, PFX_ERR_SCRIPT, "Function not found: " + signature(c)); // Rhai's ErrorFunctionNotFound text
        }
        if (st != PFX_OK) return fail(c, st, std::string(pfx_last_error(ctx)));
        if (swapped) std::swap(cur, other);
    }
    if (cur != d_img) { // result must end in d_img
        hipError_t e = hipMemcpyAsync(d_img, cur, (size_t)w * h * 4, hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) { err = {hipGetErrorString(e), 0, 0, PFX_ERR_HIP}; return PFX_ERR_HIP; }
    }
    return PFX_OK;
}

void fill_result(pfx_script_result* r, const ScriptErr* err, const std::string& console, uint32_t ops)
{
    if (!r) return;
    std::memset(r, 0, sizeof *r);
    r->ops_executed = ops;
    std::snprintf(r->console, sizeof r->console, "%s", console.c_str());
    if (err) {
        // ScriptError::friendly_message header (scripting.rs:97-115)
        std::string head = err->line > 0 ? "Error on line " + std::to_string(err->line) + (err->col > 0 ? ", column " + std::to_string(err->col) : "") + ":\n  "
                                         : "Script error:\n  ";
        std::snprintf(r->error, sizeof r->error, "%s%s", head.c_str(), err->msg.c_str());
        r->error_line = err->line;
        r->error_col = err->col;
    }
}

// script on a device-resident image; d_img is updated in place
int script_run_dev(pfx_ctx* ctx, const char* source, void* d_img, uint32_t w, uint32_t h, const uint8_t* mask, pfx_script_result* result)
{
    std::vector<Call> calls;
    ScriptErr err;
    std::string console;
    uint32_t ops = 0;
    if (!parse(source, calls, err)) {
        fill_result(result, &err, console, ops);
        return pfx_fail(ctx, err.status, "%s", err.msg.c_str());
    }
    const void* d_mask = nullptr;
    if (mask) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_mask, (size_t)w * h));
        PFX_TRY(pfx_h2d(ctx, ctx->st_mask.p, mask, (size_t)w * h));
        d_mask = ctx->st_mask.p;
    }
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, (size_t)w * h * 4));
    const int st = run_calls(ctx, calls, d_img, ctx->st_out.p, w, h, mask, d_mask, console, ops, err);
    fill_result(result, st == PFX_OK ? nullptr : &err, console, ops);
    if (st != PFX_OK) return pfx_fail(ctx, st, "%s", err.msg.c_str());
    return PFX_OK;
}

// ------------------------------------------------------------------------------------------------ PNG (RGBA8 out; 8-bit in)
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// decodes non-interlaced 8-bit gray / gray+alpha / RGB / RGBA / palette PNGs to RGBA8 (what `image::open(..).to_rgba8()` yields)
bool png_decode(const std::vector<uint8_t>& file, std::vector<uint8_t>& rgba, uint32_t& w, uint32_t& h, std::string& why)
{
    static const uint8_t sigbytes[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), sigbytes, 8) != 0) { why = "not a PNG file"; return false; }
    size_t pos = 8;
    int bit_depth = 0, color_type = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    w = h = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { why = "truncated chunk"; return false; }
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = be32(data); h = be32(data + 4); bit_depth = data[8]; color_type = data[9]; interlace = data[12]; }
        else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!w || !h || (uint64_t)w * h > 256000000ull) { why = "bad dimensions"; return false; }
    if (bit_depth != 8 || interlace != 0) { why = "only 8-bit non-interlaced PNGs are supported by this build"; return false; }
    int ch;
    switch (color_type) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: why = "bad colour type"; return false; }
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { why = "zlib inflate failed"; return false; }
    std::vector<uint8_t> img(stride * h);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        uint8_t* out = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int v = in[i];
            switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: break; }
            out[i] = (uint8_t)v;
        }
    }
    rgba.resize((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        uint8_t* o = &rgba[i * 4];
        const uint8_t* p = &img[i * ch];
        switch (color_type) {
        case 0: o[0] = o[1] = o[2] = p[0]; o[3] = (trns.size() >= 2 && trns[1] == p[0]) ? 0 : 255; break;
        case 2: o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = (trns.size() >= 6 && trns[1] == p[0] && trns[3] == p[1] && trns[5] == p[2]) ? 0 : 255; break;
        case 3: { const size_t k = p[0]; o[0] = k * 3 + 2 < plte.size() ? plte[k * 3] : 0; o[1] = k * 3 + 2 < plte.size() ? plte[k * 3 + 1] : 0;
                  o[2] = k * 3 + 2 < plte.size() ? plte[k * 3 + 2] : 0; o[3] = k < trns.size() ? trns[k] : 255; break; }
        case 4: o[0] = o[1] = o[2] = p[0]; o[3] = p[1]; break;
        default: std::memcpy(o, p, 4);
        }
    }
    return true;
}

void put_chunk(std::vector<uint8_t>& out, const char* type, const uint8_t* data, size_t len)
{
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    out.insert(out.end(), l, l + 4);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), data, data + len);
    const uint32_t crc = (uint32_t)crc32(0L, &out[start], (uInt)(len + 4));
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    out.insert(out.end(), c, c + 4);
}

bool png_encode(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw(((size_t)w * 4 + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[((size_t)w * 4 + 1) * y] = 0; // filter: None
        std::memcpy(&raw[((size_t)w * 4 + 1) * y + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0};
    put_chunk(out, "IHDR", ihdr, 13);
    put_chunk(out, "IDAT", comp.data(), clen);
    put_chunk(out, "IEND", nullptr, 0);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

// ------------------------------------------------------------------------------------------------ CLI helpers
std::string lower(std::string s) { for (char& c : s) c = (char)std::tolower((unsigned char)c); return s; }
std::string ext_of(const std::string& p) { const size_t d = p.find_last_of('.'), s = p.find_last_of('/'); return (d == std::string::npos || (s != std::string::npos && d < s)) ? "" : lower(p.substr(d + 1)); }
std::string stem_of(const std::string& p) { const size_t s = p.find_last_of('/'); std::string f = s == std::string::npos ? p : p.substr(s + 1); const size_t d = f.find_last_of('.'); return (d == std::string::npos || d == 0) ? f : f.substr(0, d); }
std::string dir_of(const std::string& p) { const size_t s = p.find_last_of('/'); return s == std::string::npos ? "." : (s == 0 ? "/" : p.substr(0, s)); }
bool exists(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

std::string canonical_format(const std::string& f) // parse_format, src/cli.rs:354-392
{
    const std::string l = lower(f);
    if (l == "jpeg" || l == "jpg") return "jpg";
    if (l == "tiff" || l == "tif") return "tiff";
    for (const char* k : {"webp", "bmp", "tga", "ico", "gif", "pfe"}) if (l == k) return k;
    return "png";
}

int mkdir_p(const std::string& dir)
{
    std::string acc;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!acc.empty() && !exists(acc) && ::mkdir(acc.c_str(), 0777) != 0 && !exists(acc)) return -1;
        }
        if (i < dir.size()) acc += dir[i];
    }
    return 0;
}

} // namespace

extern "C" {

int pfx_script_run(pfx_ctx* ctx, const char* source, uint8_t* pixels_inout, uint32_t w, uint32_t h, const uint8_t* mask,
                   pfx_script_result* result)
{
    if (result) std::memset(result, 0, sizeof *result);
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, source && pixels_inout && w && h, "pfx_script_run: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const size_t bytes = (size_t)w * h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, bytes));
    PFX_TRY(pfx_h2d(ctx, ctx->st_in.p, pixels_inout, bytes));
    PFX_TRY(script_run_dev(ctx, source, ctx->st_in.p, w, h, mask, result));
    PFX_TRY(pfx_d2h(ctx, pixels_inout, ctx->st_in.p, bytes)); // only reached on success: pixels untouched on error
    return pfx_sync(ctx);
}

// The `pfx` batch tool: same flags, loop and exit codes as src/cli.rs.  PNG in / PNG out in this build.
int pfx_cli_main(int argc, char** argv)
{
    std::vector<std::string> inputs_raw;
    std::string script_path, output, output_dir, format;
    bool have_output = false, have_dir = false, have_format = false, verbose = false;
    int device = 0;
    auto usage = [](int rc) {
        std::printf("pfx — headless batch image processor (HIP back-end of PaintFE's CLI)\n"
                    "  -i, --input <FILE>...   input file(s), glob patterns accepted\n  -s, --script <SCRIPT.rhai>\n  -o, --output <FILE>\n"
                    "      --output-dir <DIR>\n  -f, --format <FORMAT>   png (other formats are not built in)\n  -q, --quality <1-100>\n"
                    "      --webp-lossy  --tiff-compression <MODE>  --flatten  -v, --verbose  --device <N>\n");
        return rc;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](std::string& dst) -> bool { if (i + 1 >= argc) return false; dst = argv[++i]; return true; };
        std::string dummy;
        if (a == "-i" || a == "--input") {
            while (i + 1 < argc && argv[i + 1][0] != '-') inputs_raw.push_back(argv[++i]);
        } else if (a == "-s" || a == "--script") { if (!need(script_path)) return usage(2); }
        else if (a == "-o" || a == "--output") { if (!need(output)) return usage(2); have_output = true; }
        else if (a == "--output-dir") { if (!need(output_dir)) return usage(2); have_dir = true; }
        else if (a == "-f" || a == "--format") { if (!need(format)) return usage(2); have_format = true; }
        else if (a == "-q" || a == "--quality" || a == "--tiff-compression") { if (!need(dummy)) return usage(2); }
        else if (a == "--webp-lossy" || a == "--flatten") {}
        else if (a == "-v" || a == "--verbose") verbose = true;
        else if (a == "--device") { if (!need(dummy)) return usage(2); device = std::atoi(dummy.c_str()); }
        else if (a == "-h" || a == "--help") return usage(0);
        else { std::fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
    }
    if (inputs_raw.empty()) { std::fprintf(stderr, "error: the following required arguments were not provided:\n  --input <INPUT>...\n"); return 2; }

    // resolve_inputs (cli.rs:315-350): literal paths first, otherwise glob
    std::vector<std::string> inputs;
    for (const std::string& pat : inputs_raw) {
        if (exists(pat)) { if (std::find(inputs.begin(), inputs.end(), pat) == inputs.end()) inputs.push_back(pat); continue; }
        glob_t g;
        bool matched = false;
        if (::glob(pat.c_str(), 0, nullptr, &g) == 0) {
            for (size_t k = 0; k < g.gl_pathc; ++k) {
                const std::string e = g.gl_pathv[k];
                if (std::find(inputs.begin(), inputs.end(), e) == inputs.end()) inputs.push_back(e);
                matched = true;
            }
        }
        globfree(&g);
        if (!matched) std::fprintf(stderr, "warning: pattern '%s' matched no files.\n", pat.c_str());
    }
    if (inputs.empty()) { std::fprintf(stderr, "error: no input files matched the given pattern(s).\n"); return 1; }
    if (inputs.size() > 1 && have_output && !have_dir) {
        std::fprintf(stderr, "error: %zu input files given but --output only accepts a single file path.\n"
                             "Use --output-dir to specify a destination directory for batch processing.\n", inputs.size());
        return 1;
    }
    const std::string fmt = have_format ? canonical_format(format) : (have_output ? canonical_format(ext_of(output)) : "png");
    std::string script_src;
    bool have_script = false;
    if (!script_path.empty()) {
        std::vector<uint8_t> s;
        if (!read_file(script_path, s)) { std::fprintf(stderr, "error: could not read script '%s'\n", script_path.c_str()); return 1; }
        script_src.assign(s.begin(), s.end());
        have_script = true;
    }
    if (have_dir && mkdir_p(output_dir) != 0) { std::fprintf(stderr, "error: could not create output directory '%s'\n", output_dir.c_str()); return 1; }

    pfx_ctx* ctx = nullptr;
    if (pfx_ctx_create(device, &ctx) != PFX_OK) { std::fprintf(stderr, "error: no usable HIP device: %s\n", pfx_last_error(nullptr)); return 1; }

    const size_t total = inputs.size();
    const bool multi = total > 1;
    bool any_failure = false;
    for (size_t idx = 0; idx < total; ++idx) {
        const std::string& in = inputs[idx];
        if (multi || verbose) std::printf("[%zu/%zu] %s\n", idx + 1, total, in.c_str());
        const auto t0 = std::chrono::steady_clock::now();
        // build_output_path (cli.rs:399-427)
        std::string out;
        if (have_output) out = output;
        else if (have_dir) out = output_dir + "/" + stem_of(in) + "." + fmt;
        else { out = dir_of(in) + "/" + stem_of(in) + "." + fmt; if (out == in || out == "./" + in) out = dir_of(in) + "/" + stem_of(in) + "_out." + fmt; }

        std::string error;
        do { // run_one (cli.rs:222-308)
            std::vector<uint8_t> file, px;
            uint32_t w = 0, h = 0;
            std::string why;
            if (ext_of(in) != "png") { error = "load failed: only PNG input is built into this back-end"; break; }
            if (!read_file(in, file) || !png_decode(file, px, w, h, why)) { error = "load failed: " + (why.empty() ? std::string("cannot read file") : why); break; }
            const size_t bytes = (size_t)w * h * 4;
            if (pfx_use(ctx) != PFX_OK || pfx_reserve(ctx, ctx->st_in, bytes) != PFX_OK || pfx_reserve(ctx, ctx->st_aux, bytes) != PFX_OK ||
                pfx_h2d(ctx, ctx->st_aux.p, px.data(), bytes) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
            // load_image_sync stores the image as a TiledImage (all-transparent chunks dropped), the script sees
            // extract_region_rgba of it, and the result goes back through from_rgba_image (cli.rs:247-260)
            if (pfx_tiled_roundtrip_dev(ctx, ctx->st_aux.p, ctx->st_in.p, w, h) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
            if (have_script) {
                pfx_script_result res;
                if (script_run_dev(ctx, script_src.c_str(), ctx->st_in.p, w, h, nullptr, &res) != PFX_OK) { error = std::string("script error: ") + res.error; break; }
                if (verbose) {
                    std::string line;
                    for (const char* c = res.console; *c; ++c) { if (*c == '\n') { std::printf("  [script] %s\n", line.c_str()); line.clear(); } else line += *c; }
                }
                if (pfx_tiled_roundtrip_dev(ctx, ctx->st_in.p, ctx->st_aux.p, w, h) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
                if (pfx_d2h(ctx, px.data(), ctx->st_aux.p, bytes) != PFX_OK || pfx_sync(ctx) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
            } else {
                if (pfx_d2h(ctx, px.data(), ctx->st_in.p, bytes) != PFX_OK || pfx_sync(ctx) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
            }
            if (fmt != "png") { error = "save failed: format '" + fmt + "' is not built into this back-end (PNG only)"; break; }
            if (!png_encode(out, px.data(), w, h)) { error = "save failed: cannot write '" + out + "'"; break; }
        } while (false);

        if (!error.empty()) { std::fprintf(stderr, "  error: %s\n", error.c_str()); any_failure = true; continue; } // keep going (cli.rs:204-208)
        if (verbose || multi) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::printf("  \xe2\x86\x92 %s (%.0fms)\n", out.c_str(), ms);
        }
    }
    pfx_ctx_destroy(ctx);
    return any_failure ? 1 : 0;
}

} // extern "C"
