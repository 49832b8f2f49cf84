// fuzz_host.cpp — mutation fuzzing of the library's three parsers of untrusted bytes, built with AddressSanitizer + UndefinedBehaviorSanitizer
// (tests/cpp/Makefile: the host-only sources pfx_script.cpp, pfx_project.cpp, pfx_rhai.cpp, pfx_script_host.cpp, pfx_host_math.cpp are compiled INTO this
// binary with -fsanitize=address,undefined; whatever else they reference resolves from libpfx.so).  No device is touched.
//   * PNG reader      pfx_png_decode_mem        (load_image_sync, /root/reference/src/io.rs:693-723; caps: io.rs:500-503)
//   * PFE reader      pfx_project_load          (load_pfe_from_bytes, src/io.rs:477-499) + layer_pixels + save of what loaded
//   * script front end pfx_script_check          (compile_script / execute_script_sync, src/ops/scripting.rs:1489-1508, sandbox limits :288-293)
//   * host math on hostile floats (NaN, infinities, denormals, 1e30, random bit patterns): the LUT builders, the brush-line walk (checked against the reference's
//     full walk wherever that is affordable), the displacement brush, the f16 tap tables — float-to-integer conversions are sanitized too (-fsanitize=float-cast-overflow)
// usage: fuzz_host <seed_dir> <iterations_per_kind> <rng_seed>     seed_dir holds *.png, *.pfe, *.rhai written by tests/test_host_hardening.py
// Prints one JSON line with the outcome counts; any sanitizer report aborts the process (non-zero exit).
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include <dirent.h>
#include <pthread.h>
#include <sanitizer/lsan_interface.h>

#include "../../include/pfx.h"

extern "C" int pfx_int_script_check_limited(const char* source, uint32_t w, uint32_t h, pfx_script_result* result, uint64_t max_ops);
// pfx_host_math.cpp (C++ linkage: internal to the library, compiled into this binary)
void pfx_host_line_points(float x0, float y0, float x1, float y1, uint32_t width, uint32_t height, std::vector<float>& out);
void pfx_host_brush_lut(float size, float hardness, bool anti_aliased, uint8_t lut[256]);
void pfx_host_rhai_levels_lut(float in_black, float in_white, float gamma, uint8_t lut[256]);
int pfx_host_gaussian_radius(float sigma);

namespace {

using bytes = std::vector<uint8_t>;

struct rng64 {
    uint64_t s;
    explicit rng64(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { next(); next(); }
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
    uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0u; }
    bool coin(uint32_t one_in = 2) { return below(one_in) == 0; }
};

bool has_suffix(const std::string& s, const char* suf) { const size_t n = std::strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }

std::vector<bytes> load_seeds(const std::string& dir, const char* suffix)
{
    std::vector<std::string> names;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) if (has_suffix(e->d_name, suffix)) names.push_back(e->d_name);
        closedir(d);
    }
    std::sort(names.begin(), names.end());   // directory order is not deterministic
    std::vector<bytes> out;
    for (const auto& n : names) {
        FILE* f = std::fopen((dir + "/" + n).c_str(), "rb");
        if (!f) continue;
        bytes b;
        uint8_t buf[65536];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + got);
        std::fclose(f);
        out.push_back(std::move(b));
    }
    return out;
}

const uint64_t INTERESTING[] = {0ull, 1ull, 2ull, 0x7Full, 0x80ull, 0xFFull, 0x100ull, 0x7FFFull, 0x8000ull, 0xFFFFull, 0x10000ull, 25000ull, 25001ull, 256ull, 257ull,
                                0x7FFFFFFFull, 0x80000000ull, 0xFFFFFFFFull, 0x100000000ull, 1ull << 40, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 16384ull, 4096ull};

void put_int(bytes& b, size_t pos, uint64_t v, int width, bool big)
{
    for (int i = 0; i < width && pos + i < b.size(); ++i) b[pos + i] = (uint8_t)(v >> (8 * (big ? width - 1 - i : i)));
}

// byte-level mutations that know nothing about the format; `hot` = length of the prefix where the structure lives (mutated more often)
void mutate_bytes(bytes& b, rng64& r, const std::vector<bytes>& seeds, size_t hot)
{
    const int n_ops = 1 + (int)r.below(4);
    for (int op = 0; op < n_ops; ++op) {
        if (b.empty()) { b.push_back((uint8_t)r.next()); continue; }
        const size_t span = (hot && hot < b.size() && r.coin(3) == false) ? hot : b.size();
        const size_t pos = r.below((uint32_t)span);
        switch (r.below(10)) {
        case 0: b[pos] ^= (uint8_t)(1u << r.below(8)); break;
        case 1: b[pos] = (uint8_t)INTERESTING[r.below(6)]; break;
        case 2: b[pos] = (uint8_t)r.next(); break;
        case 3: put_int(b, pos, INTERESTING[r.below(sizeof INTERESTING / sizeof *INTERESTING)], 4, r.coin()); break;
        case 4: put_int(b, pos, INTERESTING[r.below(sizeof INTERESTING / sizeof *INTERESTING)], 8, false); break;
        case 5: b.resize(r.below((uint32_t)b.size() + 1)); break;                                                          // truncate
        case 6: { const size_t len = std::min<size_t>(1 + r.below(64), b.size() - pos); b.erase(b.begin() + pos, b.begin() + pos + len); break; }
        case 7: { const size_t len = std::min<size_t>(1 + r.below(64), b.size() - pos); const bytes cp(b.begin() + pos, b.begin() + pos + len);
                  b.insert(b.begin() + r.below((uint32_t)b.size() + 1), cp.begin(), cp.end()); break; }                    // duplicate a range
        case 8: { bytes ins(1 + r.below(16)); for (auto& x : ins) x = (uint8_t)r.next(); b.insert(b.begin() + pos, ins.begin(), ins.end()); break; }
        default: if (!seeds.empty()) {                                                                                       // splice another seed's tail
                  const bytes& o = seeds[r.below((uint32_t)seeds.size())];
                  if (!o.empty()) { const size_t cut = r.below((uint32_t)o.size()); b.resize(pos); b.insert(b.end(), o.begin() + cut, o.end()); }
              }
        }
        if (b.size() > (1u << 22)) b.resize(1u << 22);
    }
}

// ---- PNG, structure-aware: keep the container valid, mutate the header fields and the INFLATED scanline bytes (so that defiltering, bit unpacking,
// palette / tRNS handling and Adam7 placement run on hostile data instead of stopping at "zlib inflate failed") ----
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void chunk_out(bytes& out, const char* type, const bytes& data)
{
    const uint32_t len = (uint32_t)data.size();
    const uint8_t l[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    out.insert(out.end(), l, l + 4);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    const uint32_t crc = (uint32_t)crc32(0L, &out[start], (uInt)(len + 4));
    const uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    out.insert(out.end(), c, c + 4);
}
bool png_structured(const bytes& seed, rng64& r, bytes& out)
{
    if (seed.size() < 8) return false;
    bytes ihdr, plte, trns, idat;
    size_t pos = 8;
    while (pos + 12 <= seed.size()) {
        const uint32_t len = be32(&seed[pos]);
        if (pos + 12 + (size_t)len > seed.size()) break;
        const char* t = (const char*)&seed[pos + 4];
        const uint8_t* d = &seed[pos + 8];
        if (!std::memcmp(t, "IHDR", 4)) ihdr.assign(d, d + len);
        else if (!std::memcmp(t, "PLTE", 4)) plte.assign(d, d + len);
        else if (!std::memcmp(t, "tRNS", 4)) trns.assign(d, d + len);
        else if (!std::memcmp(t, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        pos += 12 + (size_t)len;
    }
    if (ihdr.size() < 13) return false;
    bytes raw(1u << 20);
    uLongf rl = (uLongf)raw.size();
    if (uncompress(raw.data(), &rl, idat.data(), (uLong)idat.size()) != Z_OK) return false;
    raw.resize(rl);
    // header fields
    auto new_dim = [&](uint32_t v) -> uint64_t { const uint32_t k = r.below(8); return k < 3 ? v + r.below(5) - 2 : k < 6 ? 1 + r.below(k == 5 ? 700 : 40) : INTERESTING[r.below(16)]; };
    if (r.coin(3)) put_int(ihdr, 0, new_dim(be32(&ihdr[0])), 4, true);
    if (r.coin(3)) put_int(ihdr, 4, new_dim(be32(&ihdr[4])), 4, true);
    if (r.coin(4)) { static const uint8_t depths[] = {1, 2, 4, 8, 16, 0, 3, 32}; ihdr[8] = depths[r.below(8)]; }
    if (r.coin(4)) { static const uint8_t types[] = {0, 2, 3, 4, 6, 1, 5, 7}; ihdr[9] = types[r.below(8)]; }
    if (r.coin(6)) ihdr[12] = (uint8_t)r.below(3);
    // scanline bytes: filter types and samples
    const int n_mut = (int)r.below(12);
    for (int i = 0; i < n_mut && !raw.empty(); ++i) {
        const size_t p = r.below((uint32_t)raw.size());
        raw[p] = r.coin(3) ? (uint8_t)r.below(6) : (uint8_t)r.next();
    }
    // most of the time the scanline data is made to FIT the (mutated) header, with valid filter types at the row starts, so that the decoder goes on to
    // unfilter, unpack and place hostile samples
    if (!r.coin(4)) {
        const uint32_t w = be32(&ihdr[0]), h = be32(&ihdr[4]);
        const int depth = ihdr[8], ct = ihdr[9];
        const int ch = ct == 2 ? 3 : ct == 4 ? 2 : ct == 6 ? 4 : 1;
        if (w && h && w <= 4096 && h <= 4096 && depth >= 1 && depth <= 16) {
            static const uint32_t A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
            static const uint32_t WHOLE[1][4] = {{0, 0, 1, 1}};
            std::vector<size_t> rows;
            size_t total = 0;
            const auto* P = ihdr[12] ? A7 : WHOLE;
            for (int k = 0; k < (ihdr[12] ? 7 : 1); ++k) {
                if (w <= P[k][0] || h <= P[k][1]) continue;
                const size_t pw = (w - P[k][0] + P[k][2] - 1) / P[k][2], ph = (h - P[k][1] + P[k][3] - 1) / P[k][3];
                for (size_t y = 0; y < ph && total <= (1u << 21); ++y) { rows.push_back(total); total += 1 + (pw * ch * depth + 7) / 8; }
            }
            if (total <= (1u << 21)) {
                const size_t old = raw.size();
                raw.resize(total);
                for (size_t i = old; i < total; ++i) raw[i] = (uint8_t)r.next();
                for (size_t o : rows) if (raw[o] > 4 && !r.coin(64)) raw[o] = (uint8_t)r.below(5);
            }
        }
    }
    if (r.coin(8)) raw.resize(r.below((uint32_t)raw.size() + 1));
    if (r.coin(8)) raw.resize(raw.size() + r.below(4096), (uint8_t)r.next());
    if (r.coin(6) && !plte.empty()) plte.resize(r.below((uint32_t)plte.size() + 1));
    if (r.coin(6)) { trns.resize(r.below(8)); for (auto& x : trns) x = (uint8_t)r.next(); }
    bytes comp(compressBound((uLong)raw.size()));
    uLongf cl = (uLongf)comp.size();
    if (compress2(comp.data(), &cl, raw.data(), (uLong)raw.size(), 1) != Z_OK) return false;
    comp.resize(cl);
    out.assign(seed.begin(), seed.begin() + 8);
    chunk_out(out, "IHDR", ihdr);
    if (!plte.empty()) chunk_out(out, "PLTE", plte);
    if (!trns.empty()) chunk_out(out, "tRNS", trns);
    if (r.coin(4) && comp.size() > 2) {   // IDAT split in two
        const size_t cut = 1 + r.below((uint32_t)comp.size() - 1);
        chunk_out(out, "IDAT", bytes(comp.begin(), comp.begin() + cut));
        chunk_out(out, "IDAT", bytes(comp.begin() + cut, comp.end()));
    } else chunk_out(out, "IDAT", comp);
    chunk_out(out, "IEND", bytes());
    return true;
}

// ---- scripts: token-level mutations on top of the byte-level ones ----
const char* const DICT[] = {"let", "const", "fn", "if", "else", "while", "loop", "for", "in", "break", "continue", "return", "throw", "try", "catch", "switch", "true", "false",
                            "(", ")", "{", "}", "[", "]", ";", ",", ".", "..", "..=", "=>", "|", "||", "&&", "!", "==", "!=", "<", "<=", ">", ">=", "+", "-", "*", "/", "%", "**",
                            "<<", ">>", "+=", "-=", "*=", "/=", "%=", "=", "#{", "\"", "'", "`", "${", "//", "/*", "*/", "0", "1", "-1", "255", "256", "9223372036854775807",
                            "-9223372036854775808", "0x7fffffffffffffff", "1e308", "1e-320", "0.0", "-0.0", "1.0/0.0", "width()", "height()", "print", "map_channels", "for_each_pixel",
                            "for_each_region", "get_pixel", "set_pixel", "apply_gaussian_blur", "apply_hsl", "rand_int", "rand_float", "clamp", "abs", "sqrt", "pow", "floor", "len",
                            "push", "pop", "x", "y", "r", "g", "b", "a", "i", "this", "resize_image", "flip_horizontal", "to_string", "to_int", "to_float", "PI()", "range", "type_of"};
void mutate_script(bytes& b, rng64& r, const std::vector<bytes>& seeds)
{
    const int n_ops = 1 + (int)r.below(3);
    for (int op = 0; op < n_ops; ++op) {
        const size_t pos = b.empty() ? 0 : r.below((uint32_t)b.size());
        switch (r.below(8)) {
        case 0: case 1: { const char* t = DICT[r.below(sizeof DICT / sizeof *DICT)]; std::string s = std::string(" ") + t + " "; b.insert(b.begin() + pos, s.begin(), s.end()); break; }
        case 2: {   // replace an identifier / number run by a dictionary token
            size_t e = pos; while (e < b.size() && (std::isalnum(b[e]) || b[e] == '_')) ++e;
            const char* t = DICT[r.below(sizeof DICT / sizeof *DICT)];
            b.erase(b.begin() + pos, b.begin() + e); b.insert(b.begin() + pos, t, t + std::strlen(t)); break; }
        case 3: {   // deep nesting
            static const char* open[] = {"(", "[", "{", "#{a:", "!", "-", "if true {", "|x| ", "fn f(){", "[[", "((", "\"${"};
            const char* o = open[r.below(12)]; const int depth = 1 + (int)r.below(r.coin(8) ? 4000 : 80);
            std::string s; for (int i = 0; i < depth; ++i) s += o;
            b.insert(b.begin() + pos, s.begin(), s.end()); break; }
        case 4: {   // duplicate a line
            size_t s0 = pos; while (s0 > 0 && b[s0 - 1] != '\n') --s0;
            size_t e = pos; while (e < b.size() && b[e] != '\n') ++e;
            const bytes line(b.begin() + s0, b.begin() + std::min(e + 1, b.size()));
            const int reps = 1 + (int)r.below(4);
            for (int i = 0; i < reps; ++i) b.insert(b.begin() + s0, line.begin(), line.end());
            break; }
        case 5: {   // delete a line
            size_t s0 = pos; while (s0 > 0 && b[s0 - 1] != '\n') --s0;
            size_t e = pos; while (e < b.size() && b[e] != '\n') ++e;
            b.erase(b.begin() + s0, b.begin() + e); break; }
        default: mutate_bytes(b, r, seeds, 0);
        }
        if (b.size() > (1u << 16)) b.resize(1u << 16);
    }
}

struct tally { unsigned long ok = 0, err = 0; };

// ---- host math: hostile floats ----
float hostile_float(rng64& r, float scale)
{
    static const float special[] = {0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 255.0f, 256.0f, 1e-38f, 1e-45f, 1e30f, -1e30f, 3.4e38f, 2147483648.0f, -2147483649.0f, 4294967296.0f, 16777216.0f};
    switch (r.below(6)) {
    case 0: { uint32_t b = (uint32_t)r.next(); float f; std::memcpy(&f, &b, 4); return f; }      // any bit pattern: NaNs, infinities, denormals
    case 1: return special[r.below(sizeof special / sizeof special[0])];
    case 2: return r.coin() ? __builtin_inff() : (r.coin() ? -__builtin_inff() : __builtin_nanf(""));
    default: return ((float)(r.next() % 2000001u) / 1000000.0f - 1.0f) * scale;                    // ordinary values around the image
    }
}

// the reference's walk, every step (draw_line_no_dirty, brush_render.rs:762-835): what pfx_host_line_points must equal stamp for stamp
bool line_points_full_walk(float x0, float y0, float x1, float y1, uint32_t w, uint32_t h, size_t max_steps, std::vector<float>& out)
{
    out.clear();
    auto as_u32 = [](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v); };
    auto inside = [&](float x, float y) { return x >= 0.0f && as_u32(x) < w && y >= 0.0f && as_u32(y) < h; };
    const float dx = x1 - x0, dy = y1 - y0, distance = std::sqrt(dx * dx + dy * dy);
    if (distance < 0.1f) { if (inside(x0, y0)) { out.push_back(x0); out.push_back(y0); } return true; }
    const size_t steps = (size_t)as_u32(std::ceil(distance / 1.0f));
    if (steps > max_steps) return false;
    for (size_t i = 0; i <= steps; ++i) {
        const float t = (float)i / (float)steps, x = x0 + dx * t, y = y0 + dy * t;
        if (inside(x, y)) { out.push_back(x); out.push_back(y); }
    }
    return true;
}

int host_math_stage(rng64& r, unsigned long iters, unsigned long& compared, unsigned long& compared_clipped, unsigned long& long_lines)
{
    std::vector<float> got, want;
    uint8_t lut[256];
    std::vector<float> field(64 * 48 * 2);
    for (unsigned long i = 0; i < iters; ++i) {
        const uint32_t w = 1 + r.below(3000), h = 1 + r.below(3000);
        // lines: mostly around the image, sometimes from / to absurd coordinates (the walk must stay bounded and equal)
        const float x0 = hostile_float(r, 4000.0f), y0 = hostile_float(r, 4000.0f);
        float x1 = hostile_float(r, 4000.0f), y1 = hostile_float(r, 4000.0f);
        if (r.coin(24)) { x1 = x0 + hostile_float(r, 1.0f) * 4.0e5f; y1 = y0 + hostile_float(r, 1.0f) * 4.0e5f; }   // long but affordable for the full walk
        pfx_host_line_points(x0, y0, x1, y1, w, h, got);
        // a line crosses an image in at most its diagonal's worth of unit steps (+ the duplicates f32 `i / steps` gives very long lines)
        if (got.size() / 2 > 4 * (size_t)(w + h) + 5000) { std::fprintf(stderr, "line: %zu stamps inside a %ux%u image\n", got.size() / 2, w, h); return 1; }
        if (line_points_full_walk(x0, y0, x1, y1, w, h, (size_t)1 << 21, want)) {
            ++compared;
            const float ddx = x1 - x0, ddy = y1 - y0;
            if (std::sqrt(ddx * ddx + ddy * ddy) > 65536.0f) ++compared_clipped;   // the library walked only part of this line
            if (got != want) { std::fprintf(stderr, "line (%a,%a)-(%a,%a) on %ux%u: %zu stamps, the full walk gives %zu\n", x0, y0, x1, y1, w, h, got.size() / 2, want.size() / 2); return 1; }
        } else ++long_lines;
        // table builders and friends: any float must give a table, never a fault or an out-of-range conversion
        pfx_build_levels_lut(hostile_float(r, 300.0f), hostile_float(r, 300.0f), hostile_float(r, 10.0f), hostile_float(r, 300.0f), hostile_float(r, 300.0f), lut);
        pfx_host_rhai_levels_lut(hostile_float(r, 300.0f), hostile_float(r, 300.0f), hostile_float(r, 10.0f), lut);
        pfx_host_brush_lut(hostile_float(r, 500.0f), hostile_float(r, 2.0f), r.coin(), lut);
        (void)pfx_host_gaussian_radius(hostile_float(r, 100.0f));
        float pts[16];
        const uint32_t n = r.below(9);
        for (uint32_t k = 0; k < 2 * n; ++k) pts[k] = hostile_float(r, 300.0f);
        pfx_build_curves_lut(pts, n, lut);
        pfx_build_stretch_lut((uint8_t)r.below(256), (uint8_t)r.below(256), lut);
        pfx_displacement_brush(field.data(), 64, 48, (int)r.below(6) - 1, hostile_float(r, 80.0f), hostile_float(r, 80.0f), hostile_float(r, 20.0f), hostile_float(r, 20.0f),
                               hostile_float(r, 60.0f), hostile_float(r, 2.0f));
        if (r.coin(16)) {
            uint16_t tabs[768]; float b2, b1;
            (void)pfx_gaussian_f16_tables(hostile_float(r, 30.0f), tabs, &b2, &b1);
        }
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: fuzz_host <seed_dir> <iterations_per_kind> <rng_seed>\n"); return 2; }
    const std::string dir = argv[1];
    const unsigned long iters = std::strtoul(argv[2], nullptr, 10);
    rng64 r(std::strtoull(argv[3], nullptr, 10));
    const auto pngs = load_seeds(dir, ".png"), pfes = load_seeds(dir, ".pfe"), scripts = load_seeds(dir, ".rhai");
    if (pngs.empty() || pfes.empty() || scripts.empty()) { std::fprintf(stderr, "fuzz_host: seed directory lacks *.png / *.pfe / *.rhai\n"); return 2; }
    tally t_png, t_pfe, t_script;
    const bool leak_each = std::getenv("PFX_FUZZ_LEAK_EACH") != nullptr;   // debugging aid: LeakSanitizer after every script (names the input that leaked)
    unsigned long png_structured_n = 0, max_px = 0;
    char err[256];
    std::map<std::string, unsigned long> why_png, why_pfe;   // PFX_FUZZ_VERBOSE=1: which checks the mutants die at

    const char* only = std::getenv("PFX_FUZZ_ONLY");   // debugging aid: "png" | "pfe" | "script" | "math" runs one stage
    auto stage = [&](const char* name) { return !only || std::strcmp(only, name) == 0; };
    for (unsigned long i = 0; stage("png") && i < iters; ++i) {   // ---- PNG
        const bytes& seed = pngs[r.below((uint32_t)pngs.size())];
        bytes b;
        if (!r.coin(3) && png_structured(seed, r, b)) { ++png_structured_n; if (r.coin(4)) mutate_bytes(b, r, pngs, 48); }
        else { b = seed; mutate_bytes(b, r, pngs, 48); }
        uint8_t* px = nullptr; uint32_t w = 0, h = 0;
        const int st = pfx_png_decode_mem(b.data(), b.size(), &px, &w, &h, err, sizeof err);
        if (st == PFX_OK) {
            if (!px || !w || !h) { std::fprintf(stderr, "png: PFX_OK without an image\n"); return 1; }
            volatile uint8_t sink = px[0] ^ px[(size_t)w * h * 4 - 1]; (void)sink;   // the whole buffer is addressable
            max_px = std::max<unsigned long>(max_px, (unsigned long)w * h);
            pfx_png_free(px);
            ++t_png.ok;
        } else {
            if (px) { std::fprintf(stderr, "png: error status with an image\n"); return 1; }
            ++t_png.err;
            ++why_png[err];
        }
    }
    for (unsigned long i = 0; stage("pfe") && i < iters; ++i) {   // ---- PFE
        bytes b = pfes[r.below((uint32_t)pfes.size())];
        mutate_bytes(b, r, pfes, 400);
        pfx_project* p = pfx_project_load(b.data(), b.size(), err, sizeof err);
        if (!p) { ++t_pfe.err; ++why_pfe[std::string(err).substr(0, 40)]; continue; }
        ++t_pfe.ok;
        const uint32_t w = pfx_project_width(p), h = pfx_project_height(p), n = pfx_project_layer_count(p);
        if (!w || !h || w > 25000u || h > 25000u || n < 1 || n > 256) { std::fprintf(stderr, "pfe: loaded document violates the open limits (%ux%u, %u layers)\n", w, h, n); return 1; }
        for (uint32_t k = 0; k < n; ++k) { pfx_project_layer L; if (pfx_project_layer_get(p, k, &L) != PFX_OK) { std::fprintf(stderr, "pfe: layer_get failed\n"); return 1; } }
        if ((uint64_t)w * h <= (1u << 20)) {
            bytes img((size_t)w * h * 4);
            pfx_project_layer_pixels(p, pfx_project_active_layer(p), img.data());
            uint8_t* out = nullptr; size_t n_out = 0;
            if (pfx_project_save(p, &out, &n_out) == PFX_OK && out) {
                pfx_project* q = pfx_project_load(out, n_out, err, sizeof err);   // what the library writes, it reads
                if (!q) { std::fprintf(stderr, "pfe: a saved document does not load: %s\n", err); return 1; }
                pfx_project_free(q);
                pfx_bytes_free(out);
            }
        }
        pfx_project_free(p);
    }
    // ---- scripts: on one thread with a stack as large as the one the interpreter would give itself per script (it then evaluates in place: a sanitizer build
    // pays milliseconds for every thread it starts)
    struct script_job { const std::vector<bytes>* scripts; rng64* r; unsigned long iters; tally* t; bool leak_each; int rc; } job{&scripts, &r, stage("script") ? iters : 0ul, &t_script, leak_each, 0};
    auto script_stage = [](void* q) -> void* {
        script_job& J = *static_cast<script_job*>(q);
        rng64& r = *J.r;
        for (unsigned long i = 0; i < J.iters; ++i) {
            bytes b = (*J.scripts)[r.below((uint32_t)J.scripts->size())];
            mutate_script(b, r, *J.scripts);
            b.erase(std::remove(b.begin(), b.end(), (uint8_t)0), b.end());   // the C ABI takes a NUL-terminated string
            b.push_back(0);
            pfx_script_result res;
            const int st = pfx_int_script_check_limited((const char*)b.data(), 64, 48, &res, 6000);
            if (st == PFX_OK) ++J.t->ok; else ++J.t->err;
            if (J.leak_each && __lsan_do_recoverable_leak_check()) { std::fprintf(stderr, "script: leak after input %lu:\n%s\n", i, (const char*)b.data()); J.rc = 1; return nullptr; }
            if (std::memchr(res.error, 0, sizeof res.error) == nullptr || std::memchr(res.console, 0, sizeof res.console) == nullptr) { std::fprintf(stderr, "script: unterminated result text\n"); J.rc = 1; return nullptr; }
        }
        return nullptr;
    };
    {
        pthread_attr_t at;
        pthread_t th;
        pthread_attr_init(&at);
        pthread_attr_setstacksize(&at, (size_t)192 << 20);
        if (pthread_create(&th, &at, script_stage, &job) != 0) { std::fprintf(stderr, "fuzz_host: cannot start the script thread\n"); return 2; }
        pthread_attr_destroy(&at);
        pthread_join(th, nullptr);
        if (job.rc) return job.rc;
    }
    unsigned long lines_compared = 0, lines_clipped = 0, long_lines = 0;
    if (stage("math")) if (const int rc = host_math_stage(r, iters, lines_compared, lines_clipped, long_lines)) return rc;
    if (std::getenv("PFX_FUZZ_VERBOSE")) {
        for (const auto& kv : why_png) std::fprintf(stderr, "png %8lu  %s\n", kv.second, kv.first.c_str());
        for (const auto& kv : why_pfe) std::fprintf(stderr, "pfe %8lu  %s\n", kv.second, kv.first.c_str());
    }
    std::printf("{\"iterations_per_kind\": %lu, \"png\": {\"ok\": %lu, \"error\": %lu, \"structured\": %lu, \"max_decoded_px\": %lu}, \"pfe\": {\"ok\": %lu, \"error\": %lu}, "
                "\"script\": {\"ok\": %lu, \"error\": %lu}, \"host_math\": {\"lines_equal_to_the_full_walk\": %lu, \"of_them_clipped_walks\": %lu, \"lines_too_long_to_walk\": %lu}}\n", iters, t_png.ok, t_png.err,
                png_structured_n, max_px, t_pfe.ok, t_pfe.err, t_script.ok, t_script.err, lines_compared, lines_clipped, long_lines);
    return 0;
}
