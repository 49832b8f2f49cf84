// pfx_script.cpp — the batch CLI (B6: flags, file loop, naming and exit codes of src/cli.rs:43-427) and its PNG codec.
// The script front-end it drives lives in pfx_script_host.cpp (host API) and pfx_rhai.cpp (language runtime).
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <glob.h>
#include <sys/stat.h>

#include "pfx_internal.h"

namespace {

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
    int ch;
    switch (color_type) { case 0: ch = 1; break; case 2: ch = 3; break; case 3: ch = 1; break; case 4: ch = 2; break; case 6: ch = 4; break; default: why = "bad colour type"; return false; }
    const bool depth_ok = (color_type == 0 && (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8 || bit_depth == 16)) ||
                          (color_type == 3 && (bit_depth == 1 || bit_depth == 2 || bit_depth == 4 || bit_depth == 8)) ||
                          ((color_type == 2 || color_type == 4 || color_type == 6) && (bit_depth == 8 || bit_depth == 16));
    if (!depth_ok || interlace > 1) { why = "bad bit depth / interlace method"; return false; }
    // passes: the whole image, or Adam7's seven sub-images (PNG spec 8.2); each is filtered on its own
    struct pass { uint32_t x0, y0, dx, dy; };
    static const pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const pass whole = {0, 0, 1, 1};
    const int n_pass = interlace ? 7 : 1;
    const size_t bpp_bits = (size_t)ch * bit_depth, fbpp = std::max<size_t>(1, bpp_bits / 8); // filter unit (PNG spec 9.2)
    size_t total = 0;
    for (int p = 0; p < n_pass; ++p) {
        const pass& P = interlace ? adam7[p] : whole;
        if (w <= P.x0 || h <= P.y0) continue;
        const size_t pw = (w - P.x0 + P.dx - 1) / P.dx, ph = (h - P.y0 + P.dy - 1) / P.dy;
        total += ph * (1 + (pw * bpp_bits + 7) / 8);
    }
    // deflate cannot expand by more than 1032 : 1 (RFC 1951: a 258-byte match per 2 bits): a header that promises more scanline bytes than the IDAT data can
    // inflate to is refused BEFORE the buffers for it are allocated (a 60-byte file must not cost a gigabyte)
    if (total / 1032u > idat.size()) { why = "image data too short for the declared dimensions"; return false; }
    std::vector<uint8_t> raw(total);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { why = "zlib inflate failed"; return false; }
    // 16 -> 8 bits like the `image` crate's to_rgba8 (rounded v / 257); 1 / 2 / 4-bit grey is stretched to 0..255
    auto to8 = [&](uint32_t v) -> uint8_t {
        if (bit_depth == 16) return (uint8_t)((v + 128u) / 257u);
        if (bit_depth == 8) return (uint8_t)v;
        return (uint8_t)(v * 255u / ((1u << bit_depth) - 1u));
    };
    auto trns16 = [&](size_t k) -> uint32_t { return ((uint32_t)trns[2 * k] << 8) | trns[2 * k + 1]; };
    rgba.assign((size_t)w * h * 4, 0);
    size_t off = 0;
    std::vector<uint8_t> cur, prev;
    for (int p = 0; p < n_pass; ++p) {
        const pass& P = interlace ? adam7[p] : whole;
        if (w <= P.x0 || h <= P.y0) continue;
        const size_t pw = (w - P.x0 + P.dx - 1) / P.dx, ph = (h - P.y0 + P.dy - 1) / P.dy, rb = (pw * bpp_bits + 7) / 8;
        cur.assign(rb, 0); prev.assign(rb, 0);
        for (size_t py = 0; py < ph; ++py) {
            const uint8_t ft = raw[off];
            const uint8_t* in = &raw[off + 1];
            off += 1 + rb;
            if (ft > 4) { why = "bad filter type"; return false; }
            for (size_t i = 0; i < rb; ++i) {
                const int a = i >= fbpp ? cur[i - fbpp] : 0, b = py ? prev[i] : 0, c = (py && i >= fbpp) ? prev[i - fbpp] : 0;
                int v = in[i];
                switch (ft) { case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) / 2; break; case 4: v += paeth(a, b, c); break; default: break; }
                cur[i] = (uint8_t)v;
            }
            const size_t y = P.y0 + py * P.dy;
            for (size_t px = 0; px < pw; ++px) {
                uint32_t sm[4] = {0, 0, 0, 0}; // samples at their native depth
                for (int k = 0; k < ch; ++k) {
                    const size_t bit = (px * ch + k) * bit_depth;
                    if (bit_depth == 16) sm[k] = ((uint32_t)cur[bit / 8] << 8) | cur[bit / 8 + 1];
                    else if (bit_depth == 8) sm[k] = cur[bit / 8];
                    else sm[k] = (cur[bit / 8] >> (8 - bit_depth - (bit & 7))) & ((1u << bit_depth) - 1u);
                }
                uint8_t* o = &rgba[(y * w + P.x0 + px * P.dx) * 4];
                switch (color_type) {
                case 0: o[0] = o[1] = o[2] = to8(sm[0]); o[3] = (trns.size() >= 2 && trns16(0) == sm[0]) ? 0 : 255; break;
                case 2: o[0] = to8(sm[0]); o[1] = to8(sm[1]); o[2] = to8(sm[2]);
                        o[3] = (trns.size() >= 6 && trns16(0) == sm[0] && trns16(1) == sm[1] && trns16(2) == sm[2]) ? 0 : 255; break;
                case 3: { const size_t k = sm[0]; const bool ok = k * 3 + 2 < plte.size();
                          o[0] = ok ? plte[k * 3] : 0; o[1] = ok ? plte[k * 3 + 1] : 0; o[2] = ok ? plte[k * 3 + 2] : 0; o[3] = k < trns.size() ? trns[k] : 255; break; }
                case 4: o[0] = o[1] = o[2] = to8(sm[0]); o[3] = to8(sm[1]); break;
                default: o[0] = to8(sm[0]); o[1] = to8(sm[1]); o[2] = to8(sm[2]); o[3] = to8(sm[3]);
                }
            }
            cur.swap(prev);
        }
    }
    return true;
}

} // namespace

extern "C" int pfx_png_decode_mem(const uint8_t* bytes, size_t n_bytes, uint8_t** rgba_out, uint32_t* w_out, uint32_t* h_out, char* err, size_t err_cap)
{
    if (err && err_cap) err[0] = 0;
    if (!bytes || !rgba_out || !w_out || !h_out) return PFX_ERR_INVALID;
    *rgba_out = nullptr; *w_out = *h_out = 0;
    std::string why;
    try {
        const std::vector<uint8_t> file(bytes, bytes + n_bytes);
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        if (!png_decode(file, px, w, h, why)) {
            if (err && err_cap) std::snprintf(err, err_cap, "%s", why.c_str());
            return PFX_ERR_INVALID;
        }
        uint8_t* out = (uint8_t*)std::malloc(px.size());
        if (!out) return PFX_ERR_OOM;
        std::memcpy(out, px.data(), px.size());
        *rgba_out = out; *w_out = w; *h_out = h;
        return PFX_OK;
    } catch (const std::bad_alloc&) {
        if (err && err_cap) std::snprintf(err, err_cap, "out of memory");
        return PFX_ERR_OOM;
    } catch (const std::exception& e) {   // nothing may unwind through the C ABI
        if (err && err_cap) std::snprintf(err, err_cap, "internal error: %s", e.what());
        return PFX_ERR_INVALID;
    }
}
extern "C" void pfx_png_free(uint8_t* rgba) { std::free(rgba); }

namespace {

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

struct doc_guard { // owns the document of one run_one iteration
    pfx_project* p = nullptr;
    ~doc_guard() { pfx_project_free(p); }
};

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

// The `pfx` batch tool: same flags, loop and exit codes as src/cli.rs.  PNG and PFE in / out in this build.
int pfx_cli_main(int argc, char** argv)
{
    std::vector<std::string> inputs_raw;
    std::string script_path, output, output_dir, format;
    bool have_output = false, have_dir = false, have_format = false, verbose = false;
    int device = 0, gpus = 1;
    auto usage = [](int rc) {
        std::printf("pfx — headless batch image processor (HIP back-end of PaintFE's CLI)\n"
                    "  -i, --input <FILE>...   input file(s), glob patterns accepted\n  -s, --script <SCRIPT.rhai>\n  -o, --output <FILE>\n"
                    "      --output-dir <DIR>\n  -f, --format <FORMAT>   png, pfe (other formats are not built in)\n  -q, --quality <1-100>\n"
                    "      --webp-lossy  --tiff-compression <MODE>  --flatten  -v, --verbose  --device <N>\n"
                    "      --gpus <N>              process the input files on N GPUs at once (file i -> GPU i mod N)\n");
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
        else if (a == "--gpus") { if (!need(dummy)) return usage(2); gpus = std::atoi(dummy.c_str()); if (gpus < 1) gpus = 1; }
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

    const size_t total = inputs.size();
    const bool multi = total > 1;
    // run_one (cli.rs:222-308) on one device context; returns what the serial loop would have printed for this file
    struct file_report { std::string out_text, err_text; bool failed = false; };
    auto process = [&](pfx_ctx* ctx, size_t idx) -> file_report {
        file_report rep;
        char line[1024];
        const std::string& in = inputs[idx];
        if (multi || verbose) { std::snprintf(line, sizeof line, "[%zu/%zu] %s\n", idx + 1, total, in.c_str()); rep.out_text += line; }
        const auto t0 = std::chrono::steady_clock::now();
        // build_output_path (cli.rs:399-427)
        std::string out;
        if (have_output) out = output;
        else if (have_dir) out = output_dir + "/" + stem_of(in) + "." + fmt;
        else { out = dir_of(in) + "/" + stem_of(in) + "." + fmt; if (out == in || out == "./" + in) out = dir_of(in) + "/" + stem_of(in) + "_out." + fmt; }

        std::string error;
        do {
            // Step 1, load_image_sync (io.rs:693-760): a .pfe project keeps its layers, every other format becomes a one-layer
            // document named after the file; layers are TiledImages (all-transparent chunks dropped)
            doc_guard doc;
            const std::string in_ext = ext_of(in);
            if (in_ext == "pfe") {
                char why[512] = {0};
                doc.p = pfx_project_load_file(in.c_str(), why, sizeof why);
                if (!doc.p) { error = std::string("load failed: ") + why; break; }
            } else if (in_ext == "png") {
                std::vector<uint8_t> file, px;
                uint32_t w = 0, h = 0;
                std::string why;
                if (!read_file(in, file) || !png_decode(file, px, w, h, why)) { error = "load failed: " + (why.empty() ? std::string("cannot read file") : why); break; }
                doc.p = pfx_project_new(w, h);
                if (!doc.p) { error = "load failed: Image size " + std::to_string(w) + "x" + std::to_string(h) + " is not accepted"; break; }
                const std::string stem = stem_of(in);
                if (pfx_project_add_layer(doc.p, stem.empty() ? "Background" : stem.c_str(), px.data(), 1.0f, 0, 1, PFX_LAYER_RASTER, nullptr) != PFX_OK) { error = "load failed: out of memory"; break; }
            } else { error = "load failed: only PNG and PFE input are built into this back-end"; break; }

            // Step 2, the script runs on the active layer; its canvas-wide ops are replayed on the other layers (cli.rs:238-270)
            if (have_script) {
                pfx_script_result res;
                std::vector<std::string> console;
                const int st = pfx_int_project_run_script(ctx, doc.p, script_src.c_str(), &res, &console);
                if (st != PFX_OK) { error = std::string("script error: ") + (res.error[0] ? res.error : pfx_last_error(ctx)); break; }
                if (verbose)
                    for (const std::string& l : console) rep.out_text += "  [script] " + l + "\n";
            }

            // Step 3, save (cli.rs:272-306): PFE keeps the layers; raster formats get the composite of a multi-layer document
            // (--flatten defaults to true) or the active layer of a single-layer one
            if (fmt == "pfe") {
                if (pfx_project_save_file(doc.p, out.c_str()) != PFX_OK) { error = "PFE save failed: cannot write '" + out + "'"; break; }
                break;
            }
            if (fmt != "png") { error = "save failed: format '" + fmt + "' is not built into this back-end (PNG and PFE only)"; break; }
            const uint32_t w = pfx_project_width(doc.p), h = pfx_project_height(doc.p);
            std::vector<uint8_t> px((size_t)w * h * 4);
            if (pfx_project_layer_count(doc.p) > 1) {
                if (pfx_project_composite(ctx, doc.p, px.data()) != PFX_OK) { error = std::string("device error: ") + pfx_last_error(ctx); break; }
            } else if (pfx_project_layer_pixels(doc.p, pfx_project_active_layer(doc.p), px.data()) != PFX_OK) { error = "internal error: active layer"; break; }
            if (!png_encode(out, px.data(), w, h)) { error = "save failed: cannot write '" + out + "'"; break; }
        } while (false);

        if (!error.empty()) { rep.err_text = "  error: " + error + "\n"; rep.failed = true; return rep; } // keep going (cli.rs:204-208)
        if (verbose || multi) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::snprintf(line, sizeof line, "  \xe2\x86\x92 %s (%.0fms)\n", out.c_str(), ms);
            rep.out_text += line;
        }
        return rep;
    };

    // The reference loops over the files on one thread (cli.rs:159-216).  Files are independent units, so with --gpus N worker k
    // owns device (device + k) mod device_count and takes files k, k + N, ...; reports are printed in input order.
    int n_dev = pfx_device_count();
    if (n_dev <= 0) { std::fprintf(stderr, "error: no usable HIP device: %s\n", pfx_last_error(nullptr)); return 1; }
    const size_t workers = std::min<size_t>((size_t)gpus, total);
    std::vector<pfx_ctx*> ctxs(workers, nullptr);
    for (size_t k = 0; k < workers; ++k)
        if (pfx_ctx_create((device + (int)k) % n_dev, &ctxs[k]) != PFX_OK) {
            std::fprintf(stderr, "error: no usable HIP device: %s\n", pfx_last_error(nullptr));
            for (pfx_ctx* c : ctxs) pfx_ctx_destroy(c);
            return 1;
        }
    bool any_failure = false;
    auto emit = [&](const file_report& rep) {
        std::fputs(rep.out_text.c_str(), stdout);
        if (!rep.err_text.empty()) { std::fflush(stdout); std::fputs(rep.err_text.c_str(), stderr); }
        any_failure = any_failure || rep.failed;
    };
    if (workers <= 1) {
        for (size_t idx = 0; idx < total; ++idx) emit(process(ctxs[0], idx));
    } else {
        std::vector<file_report> reports(total);
        std::vector<std::thread> th;
        th.reserve(workers);
        // no exception may leave a worker thread (std::terminate): a file whose processing throws is reported as failed
        auto work = [&](size_t k) noexcept {
            for (size_t idx = k; idx < total; idx += workers) {
                try { reports[idx] = process(ctxs[k], idx); }
                catch (...) { reports[idx].failed = true; try { reports[idx].err_text = "error: out of memory or internal error while processing the file\n"; } catch (...) {} }
            }
        };
        size_t started = 0;
        try { for (; started < workers; ++started) th.emplace_back(work, started); }
        catch (...) { std::fprintf(stderr, "warning: only %zu of %zu worker threads could start\n", started, workers); }
        for (auto& t : th) t.join();
        for (size_t k = started; k < workers; ++k) work(k); // the shares of threads that did not start run here
        for (const auto& rep : reports) emit(rep);
    }
    for (pfx_ctx* c : ctxs) pfx_ctx_destroy(c);
    return any_failure ? 1 : 0;
}

} // extern "C"
