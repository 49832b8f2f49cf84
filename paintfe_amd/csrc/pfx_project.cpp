// pfx_project.cpp — PFE project files (N2) and the document-level operations the CLI needs around the compositor.
//
// Reference: src/io.rs:85-208 (ProjectFileV0..V3 and their layer records), :242-465 (build_pfe / write_pfe_*), :469-1300
// (load_pfe_from_bytes and the four loaders), src/canvas/layers.rs:194-275,378-388 (PixelFormat, HdrMetadata, ImageMetadata,
// WebpFrameCompression, AdjustmentKind, LayerFolder), src/experimental.rs:4-10 (DeepRgbaBuffer), src/canvas/tiled_image.rs:50-104,
// 271-293,660-662,876-881 (tiling rule, flat_index, set_chunk), src/cli.rs:222-308 (run_one), src/ops/scripting.rs:1640-1723
// (apply_canvas_ops).  Encoding = bincode 1.x defaults: little endian, fixed-width integers, u64 lengths and usize, one-byte
// bool / Option tags, u32 enum variant indices, trailing bytes ignored.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "pfx_internal.h"

namespace {

constexpr uint32_t CHUNK = 64;
constexpr size_t CHUNK_BYTES = (size_t)CHUNK * CHUNK * 4;
constexpr uint32_t MAX_OPEN_IMAGE_DIM = 25000; // io.rs:500

struct OptF32 { bool some = false; float v = 0.0f; };
struct OptStr { bool some = false; std::string v; };

struct Folder { // LayerFolder, layers.rs:378-388
    uint64_t id = 0;
    std::string name;
    bool visible = true, collapsed = false;
    bool has_insert = false; uint64_t insert_above = 0;
    bool has_color = false; uint8_t color = 0;
};
struct HdrMeta { bool enabled = false; OptF32 max_lum, ref_white; OptStr transfer; };
struct ImgMeta {
    OptStr source_format, source_name, color_profile;
    std::vector<std::pair<std::string, std::string>> png_text;
    std::vector<std::vector<uint8_t>> raw_chunks;
};
struct Deep { bool some = false; uint32_t variant = 0; uint64_t count = 0; std::vector<uint8_t> raw; }; // element bytes kept verbatim

struct Layer {
    std::string name;
    bool visible = true;
    bool has_folder = false; uint64_t folder_id = 0;
    float opacity = 1.0f;
    uint8_t blend_mode = 0;
    uint8_t layer_type = 0;                 // 0 raster, 1 text, 2 adjustment
    std::vector<uint32_t> slot;             // per canvas chunk: index into `pixels` / PFX_NO_CHUNK
    std::vector<uint8_t> pixels;            // stored chunks, CHUNK_BYTES each
    bool has_content = false;
    std::vector<uint8_t> content;           // TextLayerData / AdjustmentLayerData bincode, verbatim
    uint32_t pixel_format = 0;              // PixelFormat::RgbaU8
    HdrMeta hdr;
    ImgMeta meta;
    uint32_t webp = 1;                      // WebpFrameCompression::Lossless (the default)
    Deep deep;
    uint8_t kind = PFX_LAYER_RASTER;
    float adj[16] = {0};
    uint32_t n_chunks() const { return (uint32_t)(pixels.size() / CHUNK_BYTES); }
};

} // namespace

struct pfx_project {
    int version = 1;
    uint32_t w = 0, h = 0;
    uint64_t active = 0;
    std::vector<Folder> folders;
    uint64_t next_folder_id = 1;
    std::vector<Layer> layers;
    uint32_t cxn() const { return (w + CHUNK - 1) / CHUNK; }
    uint32_t cyn() const { return (h + CHUNK - 1) / CHUNK; }
    size_t n_canvas_chunks() const { return (size_t)cxn() * cyn(); }
};

namespace {

// ------------------------------------------------------------------------------------------------ bincode reader / writer
struct Reader {
    const uint8_t* p;
    size_t n, pos = 0;
    std::string why;
    bool fail(const char* m) { if (why.empty()) why = m; return false; }
    bool need(size_t k) { return (k <= n - pos) ? true : fail("io error: unexpected end of file"); }
    bool u8(uint8_t& v) { if (!need(1)) return false; v = p[pos++]; return true; }
    bool u32(uint32_t& v) { if (!need(4)) return false; std::memcpy(&v, p + pos, 4); pos += 4; return true; }
    bool u64(uint64_t& v) { if (!need(8)) return false; std::memcpy(&v, p + pos, 8); pos += 8; return true; }
    bool f32(float& v) { if (!need(4)) return false; std::memcpy(&v, p + pos, 4); pos += 4; return true; }
    bool boolean(bool& v) { uint8_t b; if (!u8(b)) return false; if (b > 1) return fail("invalid u8 while decoding bool"); v = b != 0; return true; }
    bool tag(bool& some) { uint8_t b; if (!u8(b)) return false; if (b > 1) return fail("invalid tag encoding for Option"); some = b != 0; return true; }
    bool len(uint64_t& v, size_t elem) { if (!u64(v)) return false; return (v <= (n - pos) / (elem ? elem : 1)) ? true : fail("io error: unexpected end of file"); }
    bool bytes(std::vector<uint8_t>& v) { uint64_t l; if (!len(l, 1)) return false; v.assign(p + pos, p + pos + l); pos += l; return true; }
    bool str(std::string& s)
    {
        uint64_t l;
        if (!len(l, 1)) return false;
        s.assign((const char*)p + pos, l);
        pos += l;
        // String must be UTF-8 (bincode: InvalidUtf8Encoding)
        for (size_t i = 0; i < s.size();) {
            const uint8_t c = (uint8_t)s[i];
            const int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
            if (extra < 0 || i + extra >= s.size() + (extra == 0)) return fail("string is not valid utf8");
            for (int k = 1; k <= extra; ++k) if (((uint8_t)s[i + k] >> 6) != 2) return fail("string is not valid utf8");
            i += 1 + extra;
        }
        return true;
    }
    bool opt_f32(OptF32& o) { if (!tag(o.some)) return false; return o.some ? f32(o.v) : true; }
    bool opt_str(OptStr& o) { if (!tag(o.some)) return false; return o.some ? str(o.v) : true; }
};

struct Writer {
    std::vector<uint8_t> out;
    void raw(const void* d, size_t k) { const uint8_t* b = (const uint8_t*)d; out.insert(out.end(), b, b + k); }
    void u8(uint8_t v) { out.push_back(v); }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void f32(float v) { raw(&v, 4); }
    void boolean(bool v) { u8(v ? 1 : 0); }
    void str(const std::string& s) { u64(s.size()); raw(s.data(), s.size()); }
    void bytes(const std::vector<uint8_t>& v) { u64(v.size()); raw(v.data(), v.size()); }
    void opt_f32(const OptF32& o) { boolean(o.some); if (o.some) f32(o.v); }
    void opt_str(const OptStr& o) { boolean(o.some); if (o.some) str(o.v); }
};

void set_err(char* err, size_t cap, const std::string& m)
{
    if (err && cap) { std::snprintf(err, cap, "%s", m.c_str()); }
}

// TiledImage::set_chunk through flat_index (tiled_image.rs:660-662,876-881): the index is computed in u32 and a chunk whose
// index falls outside the table is silently dropped; a later chunk with the same index replaces the earlier one
void set_chunk(const pfx_project& P, Layer& L, uint32_t cx, uint32_t cy, const uint8_t* px)
{
    const uint32_t idx = cy * P.cxn() + cx; // wrapping u32, as in release builds of the reference
    if ((size_t)idx >= L.slot.size()) return;
    if (L.slot[idx] == PFX_NO_CHUNK) {
        L.slot[idx] = L.n_chunks();
        L.pixels.insert(L.pixels.end(), px, px + CHUNK_BYTES);
    } else {
        std::memcpy(L.pixels.data() + (size_t)L.slot[idx] * CHUNK_BYTES, px, CHUNK_BYTES);
    }
}

// TiledImage::from_rgba_image (tiled_image.rs:50-104): a chunk is kept iff some alpha inside the canvas is non-zero
void tile_from_flat(const pfx_project& P, Layer& L, const uint8_t* rgba)
{
    L.slot.assign(P.n_canvas_chunks(), PFX_NO_CHUNK);
    L.pixels.clear();
    if (!rgba) return;
    std::vector<uint8_t> chunk(CHUNK_BYTES);
    for (uint32_t cy = 0; cy < P.cyn(); ++cy)
        for (uint32_t cx = 0; cx < P.cxn(); ++cx) {
            const uint32_t bx = cx * CHUNK, by = cy * CHUNK;
            const uint32_t cw = std::min(CHUNK, P.w - bx), ch = std::min(CHUNK, P.h - by);
            std::memset(chunk.data(), 0, CHUNK_BYTES);
            bool has = false;
            for (uint32_t ly = 0; ly < ch; ++ly) {
                const uint8_t* s = rgba + ((size_t)(by + ly) * P.w + bx) * 4;
                std::memcpy(chunk.data() + (size_t)ly * CHUNK * 4, s, (size_t)cw * 4);
                for (uint32_t lx = 0; lx < cw && !has; ++lx) has = s[lx * 4 + 3] != 0;
            }
            if (has) set_chunk(P, L, cx, cy, chunk.data());
        }
}

// TiledImage::to_rgba_image (tiled_image.rs:271-293)
void flat_from_tiles(const pfx_project& P, const Layer& L, uint8_t* dst)
{
    std::memset(dst, 0, (size_t)P.w * P.h * 4);
    for (uint32_t cy = 0; cy < P.cyn(); ++cy)
        for (uint32_t cx = 0; cx < P.cxn(); ++cx) {
            const uint32_t sl = L.slot[(size_t)cy * P.cxn() + cx];
            if (sl == PFX_NO_CHUNK) continue;
            const uint32_t bx = cx * CHUNK, by = cy * CHUNK;
            const uint32_t cw = std::min(CHUNK, P.w - bx), ch = std::min(CHUNK, P.h - by);
            const uint8_t* c = L.pixels.data() + (size_t)sl * CHUNK_BYTES;
            for (uint32_t ly = 0; ly < ch; ++ly)
                std::memcpy(dst + ((size_t)(by + ly) * P.w + bx) * 4, c + (size_t)ly * CHUNK * 4, (size_t)cw * 4);
        }
}

bool read_chunks(Reader& R, const pfx_project& P, Layer& L)
{
    uint64_t n;
    if (!R.len(n, 16)) return false; // ChunkData: cx u32, cy u32, pixels Vec<u8> (8-byte length)
    L.slot.assign(P.n_canvas_chunks(), PFX_NO_CHUNK);
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t cx, cy;
        uint64_t l;
        if (!R.u32(cx) || !R.u32(cy) || !R.len(l, 1)) return false;
        if (l != CHUNK_BYTES) { // io.rs:829-838
            char m[256];
            std::snprintf(m, sizeof m, "Chunk (%u,%u) in layer '%s' has %llu bytes, expected %zu", cx, cy, L.name.c_str(), (unsigned long long)l, CHUNK_BYTES);
            R.pos += l;
            return R.fail(m);
        }
        set_chunk(P, L, cx, cy, R.p + R.pos);
        R.pos += l;
    }
    return true;
}

// AdjustmentLayerData { kind: AdjustmentKind } (layers.rs:247-273)
bool parse_adjustment(const std::vector<uint8_t>& b, uint8_t& kind, float adj[16])
{
    Reader R{b.data(), b.size()};
    uint32_t variant;
    if (!R.u32(variant)) return false;
    std::memset(adj, 0, 16 * sizeof(float));
    switch (variant) {
    case 0: kind = PFX_ADJ_EXPOSURE; return R.f32(adj[0]);
    case 1: kind = PFX_ADJ_BRIGHTNESS_CONTRAST; return R.f32(adj[0]) && R.f32(adj[1]);
    case 2: kind = PFX_ADJ_INVERT; return true;
    case 3: kind = PFX_ADJ_CHANNEL_MIXER; for (int i = 0; i < 16; ++i) if (!R.f32(adj[i])) return false; return true;
    default: return false;
    }
}

std::vector<uint8_t> encode_adjustment(uint8_t kind, const float* adj)
{
    Writer W;
    W.u32((uint32_t)kind - 1u);
    const int n = kind == PFX_ADJ_EXPOSURE ? 1 : kind == PFX_ADJ_BRIGHTNESS_CONTRAST ? 2 : kind == PFX_ADJ_CHANNEL_MIXER ? 16 : 0;
    for (int i = 0; i < n; ++i) W.f32(adj ? adj[i] : 0.0f);
    return W.out;
}

bool read_layer(Reader& R, pfx_project& P, int version, Layer& L)
{
    if (!R.str(L.name) || !R.boolean(L.visible)) return false;
    if (version >= 3) { if (!R.tag(L.has_folder)) return false; if (L.has_folder && !R.u64(L.folder_id)) return false; }
    if (!R.f32(L.opacity) || !R.u8(L.blend_mode)) return false;
    if (L.blend_mode > 24) L.blend_mode = 0; // BlendMode::from_u8: unknown ids are Normal, and a re-save writes to_u8() of that (layers.rs:125-185)
    if (version == 0) { // LayerDataV0: flat pixels, converted with from_rgba_image (io.rs:1247-1274)
        uint64_t l;
        if (!R.len(l, 1)) return false;
        const size_t expect = (size_t)P.w * P.h * 4;
        if (l != expect) {
            char m[256];
            std::snprintf(m, sizeof m, "Layer '%s' has %llu bytes, expected %zu (%ux%ux4)", L.name.c_str(), (unsigned long long)l, expect, P.w, P.h);
            return R.fail(m);
        }
        tile_from_flat(P, L, R.p + R.pos);
        R.pos += l;
        return true;
    }
    if (version >= 2 && !R.u8(L.layer_type)) return false;
    if (!read_chunks(R, P, L)) return false;
    if (version >= 2) {
        if (!R.tag(L.has_content)) return false;
        if (L.has_content && !R.bytes(L.content)) return false;
    }
    if (version >= 3) {
        if (!R.u32(L.pixel_format)) return false;
        if (L.pixel_format > 3) return R.fail("invalid value: PixelFormat variant index out of range");
        if (!R.boolean(L.hdr.enabled) || !R.opt_f32(L.hdr.max_lum) || !R.opt_f32(L.hdr.ref_white) || !R.opt_str(L.hdr.transfer)) return false;
        if (!R.opt_str(L.meta.source_format) || !R.opt_str(L.meta.source_name) || !R.opt_str(L.meta.color_profile)) return false;
        uint64_t n;
        if (!R.len(n, 16)) return false;
        L.meta.png_text.resize(n);
        for (auto& kv : L.meta.png_text) if (!R.str(kv.first) || !R.str(kv.second)) return false;
        if (!R.len(n, 8)) return false;
        L.meta.raw_chunks.resize(n);
        for (auto& c : L.meta.raw_chunks) if (!R.bytes(c)) return false;
        if (!R.u32(L.webp)) return false;
        if (L.webp > 1) return R.fail("invalid value: WebpFrameCompression variant index out of range");
        if (!R.tag(L.deep.some)) return false;
        if (L.deep.some) {
            if (!R.u32(L.deep.variant)) return false;
            if (L.deep.variant > 3) return R.fail("invalid value: DeepRgbaBuffer variant index out of range");
            const size_t es = L.deep.variant == 0 ? 1 : L.deep.variant == 3 ? 4 : 2;
            if (!R.len(L.deep.count, es)) return false;
            L.deep.raw.assign(R.p + R.pos, R.p + R.pos + L.deep.count * es);
            R.pos += L.deep.count * es;
        }
    }
    // content reconstruction (io.rs:852-870, 997-1020): an adjustment payload that does not parse degrades to a raster layer;
    // text payloads are kept verbatim (their rasterised chunks are what gets composited)
    if (version >= 3 && L.layer_type == 2) {
        if (!(L.has_content && parse_adjustment(L.content, L.kind, L.adj))) { L.layer_type = 0; L.has_content = false; L.content.clear(); L.kind = PFX_LAYER_RASTER; }
    } else if (L.layer_type == 1) {
        if (!L.has_content) L.layer_type = 0;
    } else {
        L.layer_type = 0; L.has_content = false; L.content.clear();
    }
    return true;
}

pfx_project* load_bytes(const uint8_t* raw, size_t n, std::string& why)
{
    if (!raw || n < 12) { why = "Invalid PFE format: File too small"; return nullptr; } // io.rs:478-480
    int version = -1;
    for (int v = 0; v <= 3; ++v) { const char m[5] = {'P', 'F', 'E', (char)('0' + v), 0}; if (!std::memcmp(raw + 8, m, 4)) version = v; }
    if (version < 0) {
        std::string magic((const char*)raw + 8, 4);
        for (char& c : magic) if ((unsigned char)c < 0x20 || (unsigned char)c > 0x7e) { magic.clear(); break; } // from_utf8(..).unwrap_or("")
        why = "Invalid PFE format: Unknown magic '" + magic + "'";
        return nullptr;
    }
    pfx_project* P = new pfx_project();
    P->version = version;
    Reader R{raw, n};
    std::string magic;
    uint64_t n_layers = 0;
    bool ok = R.str(magic) && R.u32(P->w) && R.u32(P->h) && R.u64(P->active);
    if (ok && version == 3) {
        uint64_t nf;
        ok = R.len(nf, 19);
        if (ok) P->folders.resize(nf);
        for (size_t i = 0; ok && i < P->folders.size(); ++i) {
            Folder& F = P->folders[i];
            ok = R.u64(F.id) && R.str(F.name) && R.boolean(F.visible) && R.boolean(F.collapsed) && R.tag(F.has_insert) &&
                 (!F.has_insert || R.u64(F.insert_above)) && R.tag(F.has_color) && (!F.has_color || R.u8(F.color));
        }
        ok = ok && R.u64(P->next_folder_id);
    }
    // the reference deserialises the whole file first and validates afterwards; a zero-sized canvas cannot hold chunk
    // tables, so dimensions are checked before the layers are materialised (the outcome — an error — is the same)
    if (ok && (P->w == 0 || P->h == 0)) { why = "Invalid PFE format: Image dimensions cannot be zero"; delete P; return nullptr; }
    if (ok && (P->w > MAX_OPEN_IMAGE_DIM || P->h > MAX_OPEN_IMAGE_DIM)) {
        char m[160];
        std::snprintf(m, sizeof m, "Invalid PFE format: Image size %ux%u exceeds maximum allowed %ux%u", P->w, P->h, MAX_OPEN_IMAGE_DIM, MAX_OPEN_IMAGE_DIM);
        why = m; delete P; return nullptr;
    }
    ok = ok && R.len(n_layers, 14);
    if (ok && n_layers > PFX_MAX_LAYERS) {
        char m[160];
        std::snprintf(m, sizeof m, "Invalid PFE format: Project contains %llu layers, which exceeds the maximum of %d", (unsigned long long)n_layers, PFX_MAX_LAYERS);
        why = m; delete P; return nullptr;
    }
    if (ok) P->layers.resize(n_layers);
    for (size_t i = 0; ok && i < P->layers.size(); ++i) ok = read_layer(R, *P, version, P->layers[i]);
    if (!ok) {
        const bool fmt = R.why.rfind("Chunk (", 0) == 0 || R.why.rfind("Layer '", 0) == 0;
        why = (fmt ? "Invalid PFE format: " : "Serialization error: ") + R.why;
        delete P;
        return nullptr;
    }
    if (P->layers.empty()) { why = "Invalid PFE format: Project contains no layers"; delete P; return nullptr; }
    if (P->active > P->layers.size() - 1) P->active = P->layers.size() - 1; // .min(layers.len() - 1)
    return P;
}

bool folder_visible(const pfx_project& P, const Layer& L)
{
    if (!L.has_folder) return true;
    for (const Folder& F : P.folders) if (F.id == L.folder_id) return F.visible; // first match (canvas_state.rs:204-208)
    return true; // is_none_or
}

void write_chunks(Writer& W, const pfx_project& P, const Layer& L)
{
    W.u64(L.n_chunks());
    for (uint32_t cy = 0; cy < P.cyn(); ++cy)
        for (uint32_t cx = 0; cx < P.cxn(); ++cx) { // chunk_keys(): flat-index order (tiled_image.rs:884-893)
            const uint32_t sl = L.slot[(size_t)cy * P.cxn() + cx];
            if (sl == PFX_NO_CHUNK) continue;
            W.u32(cx); W.u32(cy);
            W.u64(CHUNK_BYTES);
            W.raw(L.pixels.data() + (size_t)sl * CHUNK_BYTES, CHUNK_BYTES);
        }
}

int save_version(const pfx_project& P) // build_pfe, io.rs:254-282
{
    bool folders = !P.folders.empty(), experimental = false, text = false;
    for (const Layer& L : P.layers) {
        folders = folders || L.has_folder;
        experimental = experimental || L.layer_type == 2 || L.pixel_format != 0 || L.hdr.enabled || !L.meta.png_text.empty() ||
                       !L.meta.raw_chunks.empty() || L.meta.source_format.some || L.webp != 1 || L.deep.some;
        text = text || L.layer_type == 1;
    }
    return (experimental || folders) ? 3 : text ? 2 : 1;
}

void serialize(const pfx_project& P, Writer& W)
{
    const int v = save_version(P);
    const char magic[5] = {'P', 'F', 'E', (char)('0' + v), 0};
    W.str(magic);
    W.u32(P.w); W.u32(P.h); W.u64(P.active);
    if (v == 3) {
        W.u64(P.folders.size());
        for (const Folder& F : P.folders) {
            W.u64(F.id); W.str(F.name); W.boolean(F.visible); W.boolean(F.collapsed);
            W.boolean(F.has_insert); if (F.has_insert) W.u64(F.insert_above);
            W.boolean(F.has_color); if (F.has_color) W.u8(F.color);
        }
        W.u64(P.next_folder_id);
    }
    W.u64(P.layers.size());
    for (const Layer& L : P.layers) {
        W.str(L.name); W.boolean(L.visible);
        if (v == 3) { W.boolean(L.has_folder); if (L.has_folder) W.u64(L.folder_id); }
        W.f32(L.opacity); W.u8(L.blend_mode);
        if (v >= 2) W.u8(v == 2 && L.layer_type == 2 ? 0 : L.layer_type); // V2 stores adjustment layers as raster (io.rs:374)
        write_chunks(W, P, L);
        if (v >= 2) {
            const bool some = L.has_content && (v == 3 || L.layer_type == 1);
            W.boolean(some);
            if (some) W.bytes(L.content);
        }
        if (v == 3) {
            W.u32(L.pixel_format);
            W.boolean(L.hdr.enabled); W.opt_f32(L.hdr.max_lum); W.opt_f32(L.hdr.ref_white); W.opt_str(L.hdr.transfer);
            W.opt_str(L.meta.source_format); W.opt_str(L.meta.source_name); W.opt_str(L.meta.color_profile);
            W.u64(L.meta.png_text.size());
            for (const auto& kv : L.meta.png_text) { W.str(kv.first); W.str(kv.second); }
            W.u64(L.meta.raw_chunks.size());
            for (const auto& c : L.meta.raw_chunks) W.bytes(c);
            W.u32(L.webp);
            W.boolean(L.deep.some);
            if (L.deep.some) { W.u32(L.deep.variant); W.u64(L.deep.count); W.raw(L.deep.raw.data(), L.deep.raw.size()); }
        }
    }
}

bool read_whole_file(const char* path, std::vector<uint8_t>& out)
{
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && std::fread(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

// device-side temporaries of one document operation (freed on scope exit)
struct DevTemps {
    std::vector<void*> ptrs;
    ~DevTemps() { for (void* p : ptrs) if (p) (void)hipFree(p); }
    int alloc(pfx_ctx* ctx, size_t bytes, void** out)
    {
        *out = nullptr;
        PFX_HIP(ctx, hipMalloc(out, bytes ? bytes : 256));
        ptrs.push_back(*out);
        return PFX_OK;
    }
    void release(void* p)
    {
        for (void*& q : ptrs) if (q == p && p) { (void)hipFree(p); q = nullptr; }
    }
};

// to_rgba_image of one layer on the device: its stored chunks are uploaded packed and scattered by a kernel
int import_layer(pfx_ctx* ctx, const pfx_project& P, const Layer& L, void* d_slot, void* d_flat)
{
    PFX_TRY(pfx_h2d(ctx, d_slot, L.slot.data(), L.slot.size() * sizeof(uint32_t)));
    if (!L.pixels.empty()) {
        PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, L.pixels.size()));
        PFX_TRY(pfx_h2d(ctx, ctx->st_aux2.p, L.pixels.data(), L.pixels.size()));
    }
    PFX_HIP(ctx, pfxk_chunks_import(ctx->stream, (const uint8_t*)ctx->st_aux2.p, (const uint32_t*)d_slot, P.w, P.h, (uint8_t*)d_flat));
    return pfx_sync(ctx); // st_aux2 and the slot table are reused by the next layer
}

// from_rgba_image of a device image into a layer: populated set -> slot table -> packed export -> only those chunks come back
int export_layer(pfx_ctx* ctx, const pfx_project& P, Layer& L, const void* d_flat, void* d_slot)
{
    const size_t nc = P.n_canvas_chunks();
    PFX_TRY(pfx_reserve(ctx, ctx->d_chunks, nc));
    PFX_HIP(ctx, pfxk_chunk_populated(ctx->stream, (const uint8_t*)d_flat, P.w, P.h, (uint8_t*)ctx->d_chunks.p));
    std::vector<uint8_t> pop(nc);
    PFX_TRY(pfx_d2h(ctx, pop.data(), ctx->d_chunks.p, nc));
    PFX_TRY(pfx_sync(ctx));
    L.slot.assign(nc, PFX_NO_CHUNK);
    uint32_t n = 0;
    for (size_t i = 0; i < nc; ++i) if (pop[i]) L.slot[i] = n++;
    L.pixels.assign((size_t)n * CHUNK_BYTES, 0);
    if (n == 0) return PFX_OK;
    PFX_TRY(pfx_h2d(ctx, d_slot, L.slot.data(), nc * sizeof(uint32_t)));
    PFX_TRY(pfx_reserve(ctx, ctx->st_aux2, L.pixels.size()));
    PFX_HIP(ctx, pfxk_chunks_export(ctx->stream, (const uint8_t*)d_flat, (const uint32_t*)d_slot, P.w, P.h, (uint8_t*)ctx->st_aux2.p));
    PFX_TRY(pfx_d2h(ctx, L.pixels.data(), ctx->st_aux2.p, L.pixels.size()));
    return pfx_sync(ctx);
}

} // namespace

extern "C" {

pfx_project* pfx_project_load(const uint8_t* bytes, size_t n_bytes, char* err, size_t err_cap)
{
    // exception barrier: nothing may unwind through the C ABI (a corrupt length field must not become std::terminate)
    try {
        std::string why;
        pfx_project* P = load_bytes(bytes, n_bytes, why);
        if (!P) set_err(err, err_cap, why);
        return P;
    } catch (const std::bad_alloc&) {
        set_err(err, err_cap, "out of memory while reading the project");
    } catch (const std::exception& e) {
        set_err(err, err_cap, std::string("internal error: ") + e.what());
    }
    return nullptr;
}

pfx_project* pfx_project_load_file(const char* path, char* err, size_t err_cap)
{
    std::vector<uint8_t> raw;
    if (!path || !read_whole_file(path, raw)) { set_err(err, err_cap, std::string("IO error: cannot read '") + (path ? path : "") + "'"); return nullptr; }
    return pfx_project_load(raw.data(), raw.size(), err, err_cap);
}

pfx_project* pfx_project_new(uint32_t w, uint32_t h)
{
    if (w == 0 || h == 0 || w > MAX_OPEN_IMAGE_DIM || h > MAX_OPEN_IMAGE_DIM) return nullptr;
    pfx_project* P = new pfx_project();
    P->w = w; P->h = h;
    return P;
}

void pfx_project_free(pfx_project* p) { delete p; }
uint32_t pfx_project_width(const pfx_project* p) { return p ? p->w : 0; }
uint32_t pfx_project_height(const pfx_project* p) { return p ? p->h : 0; }
uint32_t pfx_project_layer_count(const pfx_project* p) { return p ? (uint32_t)p->layers.size() : 0; }
uint32_t pfx_project_active_layer(const pfx_project* p) { return p ? (uint32_t)p->active : 0; }
int pfx_project_version(const pfx_project* p) { return p ? p->version : -1; }

int pfx_project_layer_get(const pfx_project* p, uint32_t index, pfx_project_layer* out)
{
    if (!p || !out || index >= p->layers.size()) return PFX_ERR_INVALID;
    const Layer& L = p->layers[index];
    std::memset(out, 0, sizeof *out);
    out->name = L.name.c_str();
    out->visible = L.visible;
    out->effectively_visible = L.visible && folder_visible(*p, L);
    out->blend_mode = L.blend_mode;
    out->layer_type = L.layer_type;
    out->opacity = L.opacity;
    out->folder_id = L.has_folder ? (int64_t)L.folder_id : -1;
    out->n_chunks = L.n_chunks();
    out->kind = L.kind;
    std::memcpy(out->adj, L.adj, sizeof out->adj);
    return PFX_OK;
}

int pfx_project_layer_pixels(const pfx_project* p, uint32_t index, uint8_t* dst)
{
    if (!p || !dst || index >= p->layers.size()) return PFX_ERR_INVALID;
    flat_from_tiles(*p, p->layers[index], dst);
    return PFX_OK;
}

int pfx_project_add_layer(pfx_project* p, const char* name, const uint8_t* rgba, float opacity, uint8_t blend_mode, uint8_t visible, uint8_t kind,
                          const float* adj)
{
    if (!p || kind > PFX_ADJ_CHANNEL_MIXER || p->layers.size() >= PFX_MAX_LAYERS) return PFX_ERR_INVALID;
    Layer L;
    L.name = name ? name : "";
    L.visible = visible != 0;
    L.opacity = opacity;
    L.blend_mode = blend_mode;
    tile_from_flat(*p, L, kind == PFX_LAYER_RASTER ? rgba : nullptr);
    if (kind != PFX_LAYER_RASTER) {
        L.layer_type = 2;
        L.kind = kind;
        if (adj) std::memcpy(L.adj, adj, sizeof L.adj);
        L.has_content = true;
        L.content = encode_adjustment(kind, L.adj);
    }
    p->layers.push_back(std::move(L));
    return PFX_OK;
}

int pfx_project_set_active_layer(pfx_project* p, uint32_t index)
{
    if (!p || index >= p->layers.size()) return PFX_ERR_INVALID;
    p->active = index;
    return PFX_OK;
}

int pfx_project_set_layer_folder(pfx_project* p, uint32_t index, int64_t folder_id)
{
    if (!p || index >= p->layers.size()) return PFX_ERR_INVALID;
    p->layers[index].has_folder = folder_id >= 0;
    p->layers[index].folder_id = folder_id >= 0 ? (uint64_t)folder_id : 0;
    return PFX_OK;
}

int pfx_project_add_folder(pfx_project* p, uint64_t id, const char* name, uint8_t visible)
{
    if (!p) return PFX_ERR_INVALID;
    Folder F;
    F.id = id; F.name = name ? name : ""; F.visible = visible != 0;
    p->folders.push_back(F);
    if (id >= p->next_folder_id) p->next_folder_id = id + 1;
    return PFX_OK;
}

int pfx_project_set_layer_pixels(pfx_project* p, uint32_t index, const uint8_t* rgba)
{
    if (!p || !rgba || index >= p->layers.size()) return PFX_ERR_INVALID;
    tile_from_flat(*p, p->layers[index], rgba);
    return PFX_OK;
}

int pfx_project_save(const pfx_project* p, uint8_t** bytes_out, size_t* n_out)
{
    if (!p || !bytes_out || !n_out) return PFX_ERR_INVALID;
    Writer W;
    try { serialize(*p, W); } catch (const std::exception&) { return PFX_ERR_OOM; } // exception barrier of the C ABI
    *bytes_out = (uint8_t*)std::malloc(W.out.size() ? W.out.size() : 1);
    if (!*bytes_out) return PFX_ERR_OOM;
    std::memcpy(*bytes_out, W.out.data(), W.out.size());
    *n_out = W.out.size();
    return PFX_OK;
}

int pfx_project_save_file(const pfx_project* p, const char* path)
{
    if (!p || !path) return PFX_ERR_INVALID;
    Writer W;
    serialize(*p, W);
    FILE* f = std::fopen(path, "wb");
    if (!f) return PFX_ERR_INVALID;
    const bool ok = std::fwrite(W.out.data(), 1, W.out.size(), f) == W.out.size();
    return (std::fclose(f) == 0 && ok) ? PFX_OK : PFX_ERR_INVALID;
}

void pfx_bytes_free(uint8_t* bytes) { std::free(bytes); }

int pfx_tiled_import_dev(pfx_ctx* ctx, const void* packed_dev, const uint32_t* slot_host, uint32_t w, uint32_t h, void* flat_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, slot_host && flat_dev && pfx_dims_ok(w, h), "pfx_tiled_import_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const size_t nc = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, nc * sizeof(uint32_t)));
    PFX_TRY(pfx_h2d(ctx, ctx->d_misc.p, slot_host, nc * sizeof(uint32_t)));
    PFX_HIP(ctx, pfxk_chunks_import(ctx->stream, (const uint8_t*)packed_dev, (const uint32_t*)ctx->d_misc.p, w, h, (uint8_t*)flat_dev));
    return PFX_OK;
}

int pfx_tiled_export_dev(pfx_ctx* ctx, const void* flat_dev, uint32_t w, uint32_t h, const uint32_t* slot_host, void* packed_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, slot_host && flat_dev && packed_dev && pfx_dims_ok(w, h), "pfx_tiled_export_dev: bad arguments");
    PFX_TRY(pfx_use(ctx));
    const size_t nc = (size_t)((w + 63) / 64) * ((h + 63) / 64);
    PFX_TRY(pfx_reserve(ctx, ctx->d_misc, nc * sizeof(uint32_t)));
    PFX_TRY(pfx_h2d(ctx, ctx->d_misc.p, slot_host, nc * sizeof(uint32_t)));
    PFX_HIP(ctx, pfxk_chunks_export(ctx->stream, (const uint8_t*)flat_dev, (const uint32_t*)ctx->d_misc.p, w, h, (uint8_t*)packed_dev));
    return PFX_OK;
}

int pfx_project_composite_dev(pfx_ctx* ctx, const pfx_project* p, void* dst_dev)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, p && dst_dev, "pfx_project_composite_dev: null argument");
    PFX_TRY(pfx_use(ctx));
    const pfx_project& P = *p;
    const size_t bytes = (size_t)P.w * P.h * 4;
    DevTemps tmp;
    void* d_slot = nullptr;
    PFX_TRY(tmp.alloc(ctx, P.n_canvas_chunks() * sizeof(uint32_t), &d_slot));
    std::vector<pfx_layer_info> infos;
    std::vector<const void*> ptrs;
    std::vector<uint8_t> chunk_keys(P.n_canvas_chunks(), 0); // union of the visible layers' chunk_keys() (canvas_state.rs:528-540)
    for (const Layer& L : P.layers) {
        if (!(L.visible && folder_visible(P, L))) continue; // layer_effectively_visible (canvas_state.rs:216-227,576)
        if (L.kind == PFX_LAYER_RASTER)
            for (size_t c = 0; c < chunk_keys.size() && c < L.slot.size(); ++c)
                if (L.slot[c] != PFX_NO_CHUNK) chunk_keys[c] = 1;
        pfx_layer_info I{};
        I.layer_idx = (uint32_t)infos.size();
        I.opacity = L.opacity;
        I.visible = 1;
        I.blend_mode = L.blend_mode;
        I.kind = L.kind;
        std::memcpy(I.adj, L.adj, sizeof I.adj);
        void* d_flat = nullptr;
        if (L.kind == PFX_LAYER_RASTER) {
            PFX_TRY(tmp.alloc(ctx, bytes, &d_flat));
            PFX_TRY(import_layer(ctx, P, L, d_slot, d_flat));
        }
        infos.push_back(I);
        ptrs.push_back(d_flat);
    }
    if (infos.empty()) { // nothing visible: composite() of no layers is the zeroed image (canvas_state.rs:506)
        PFX_HIP(ctx, hipMemsetAsync(dst_dev, 0, bytes, ctx->stream));
        return pfx_sync(ctx);
    }
    PFX_TRY(pfx_int_flatten_with_chunk_keys_dev(ctx, ptrs.data(), infos.data(), (uint32_t)infos.size(), P.w, P.h, dst_dev, chunk_keys.data()));
    return pfx_sync(ctx); // the temporaries are freed on return
}

int pfx_project_composite(pfx_ctx* ctx, const pfx_project* p, uint8_t* dst)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, p && dst, "pfx_project_composite: null argument");
    PFX_TRY(pfx_use(ctx));
    const size_t bytes = (size_t)p->w * p->h * 4;
    PFX_TRY(pfx_reserve(ctx, ctx->st_out, bytes));
    PFX_TRY(pfx_project_composite_dev(ctx, p, ctx->st_out.p));
    PFX_TRY(pfx_d2h(ctx, dst, ctx->st_out.p, bytes));
    return pfx_sync(ctx);
}

} // extern "C"

int pfx_int_project_run_script(pfx_ctx* ctx, pfx_project* p, const char* source, pfx_script_result* result, std::vector<std::string>* console)
{
    if (!ctx) return PFX_ERR_INVALID;
    PFX_REQUIRE(ctx, p && source && !p->layers.empty(), "pfx_project_run_script: null argument or empty document");
    PFX_TRY(pfx_use(ctx));
    pfx_project& P = *p;
    const size_t active = (size_t)P.active;
    DevTemps tmp;
    void* d_slot = nullptr;
    PFX_TRY(tmp.alloc(ctx, P.n_canvas_chunks() * sizeof(uint32_t), &d_slot));
    // the script sees extract_region_rgba of the active layer (cli.rs:240-246)
    PFX_TRY(pfx_reserve(ctx, ctx->st_in, (size_t)P.w * P.h * 4));
    PFX_TRY(import_layer(ctx, P, P.layers[active], d_slot, ctx->st_in.p));
    uint32_t nw = P.w, nh = P.h;
    std::vector<pfx_canvas_op> ops;
    pfx_script_result local;
    PFX_TRY(pfx_int_script_run_dev(ctx, source, &nw, &nh, nullptr, result ? result : &local, console, &ops));

    // the active layer takes the script result, re-tiled with from_rgba_image (cli.rs:257-259)
    pfx_project Q; // geometry holder: the canvas size in effect for a tiling step
    Q.w = nw; Q.h = nh;
    void* d_slot_new = nullptr;
    PFX_TRY(tmp.alloc(ctx, Q.n_canvas_chunks() * sizeof(uint32_t), &d_slot_new));
    Layer act = P.layers[active]; // keeps name, flags and payloads
    PFX_TRY(export_layer(ctx, Q, act, ctx->st_in.p, d_slot_new));

    // canvas-wide ops are replayed on every other layer, each re-tiled after every op (apply_canvas_ops, scripting.rs:1640-1723)
    std::vector<Layer> done(P.layers.size());
    if (!ops.empty()) {
        for (size_t li = 0; li < P.layers.size(); ++li) {
            if (li == active) continue;
            uint32_t cw = P.w, ch = P.h;
            void* d_a = nullptr;
            PFX_TRY(tmp.alloc(ctx, (size_t)cw * ch * 4, &d_a));
            PFX_TRY(import_layer(ctx, P, P.layers[li], d_slot, d_a));
            for (const pfx_canvas_op& op : ops) {
                uint32_t ow = cw, oh = ch;
                if (op.kind == PFX_CANVAS_ROTATE_90CW || op.kind == PFX_CANVAS_ROTATE_90CCW) std::swap(ow, oh);
                else if (op.kind == PFX_CANVAS_RESIZE_IMAGE || op.kind == PFX_CANVAS_RESIZE_CANVAS) { ow = op.w; oh = op.h; }
                void *d_b = nullptr, *d_c = nullptr;
                PFX_TRY(tmp.alloc(ctx, (size_t)ow * oh * 4, &d_b));
                if (op.kind <= PFX_CANVAS_ROTATE_180) PFX_TRY(pfx_flip_rotate_dev(ctx, d_a, cw, ch, d_b, op.kind));
                else if (op.kind == PFX_CANVAS_RESIZE_IMAGE) PFX_TRY(pfx_resize_image_dev(ctx, d_a, cw, ch, d_b, ow, oh, (int)op.anchor_x));
                else PFX_TRY(pfx_resize_canvas_dev(ctx, d_a, cw, ch, d_b, ow, oh, op.anchor_x, op.anchor_y, nullptr));
                // TiledImage::from_rgba_image after every op: an all-transparent chunk loses its colour as well
                PFX_TRY(tmp.alloc(ctx, (size_t)ow * oh * 4, &d_c));
                PFX_TRY(pfx_tiled_roundtrip_dev(ctx, d_b, d_c, ow, oh));
                PFX_TRY(pfx_sync(ctx));
                tmp.release(d_a);
                tmp.release(d_b);
                d_a = d_c; cw = ow; ch = oh;
            }
            if (cw != nw || ch != nh)
                return pfx_fail(ctx, PFX_ERR_SCRIPT, "script result is %ux%u but its canvas ops end at %ux%u", nw, nh, cw, ch);
            done[li] = P.layers[li];
            PFX_TRY(export_layer(ctx, Q, done[li], d_a, d_slot_new));
            tmp.release(d_a);
        }
        for (size_t li = 0; li < P.layers.size(); ++li) if (li != active) P.layers[li] = std::move(done[li]);
    } else if (nw != P.w || nh != P.h) {
        // cannot happen with the registered host API (every size change is recorded as a canvas op); kept as a guard
        return pfx_fail(ctx, PFX_ERR_SCRIPT, "script changed the size without a canvas op");
    }
    P.layers[active] = std::move(act);
    P.w = nw; P.h = nh; // the document takes the new size (cli.rs:262-270)
    return PFX_OK;
}

extern "C" {

int pfx_project_run_script(pfx_ctx* ctx, pfx_project* p, const char* source, pfx_script_result* result)
{
    return pfx_int_project_run_script(ctx, p, source, result, nullptr);
}

} // extern "C"
