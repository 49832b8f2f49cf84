"""Independent restatement of the PFE project file layout (TEST INFRASTRUCTURE ONLY): bincode 1.x encoding of
ProjectFileV0..V3 as declared in the reference (src/io.rs:85-208; nested types src/canvas/layers.rs:194-275,378-388 and
src/experimental.rs:4-10), written from the struct declarations alone so that it can check paintfe_amd/csrc/pfx_project.cpp.

Parity status: UNPINNED against real reference output — the reference tree holds no .pfe fixture (its tests are save->load
round trips, tests/io_roundtrip.rs:124-330), and the Rust reference cannot be built here.  bincode 1.x defaults: little endian,
fixed-width integers, u64 for usize and sequence lengths, 1-byte bool and Option tags, u32 enum variant index."""
import struct

import numpy as np

CHUNK = 64
CHUNK_BYTES = CHUNK * CHUNK * 4


# ------------------------------------------------------------------ primitives
class W:
    def __init__(self):
        self.b = bytearray()

    def u8(self, v): self.b += struct.pack("<B", v)
    def u32(self, v): self.b += struct.pack("<I", v)
    def u64(self, v): self.b += struct.pack("<Q", v)
    def f32(self, v): self.b += struct.pack("<f", v)
    def boolean(self, v): self.u8(1 if v else 0)
    def raw(self, v): self.b += bytes(v)
    def string(self, s): e = s.encode("utf-8"); self.u64(len(e)); self.raw(e)
    def vec_u8(self, v): self.u64(len(v)); self.raw(v)

    def opt(self, v, put):
        self.boolean(v is not None)
        if v is not None:
            put(v)


class R:
    def __init__(self, raw):
        self.b, self.p = bytes(raw), 0

    def take(self, n):
        if self.p + n > len(self.b):
            raise ValueError("unexpected end of file")
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def u8(self): return struct.unpack("<B", self.take(1))[0]
    def u32(self): return struct.unpack("<I", self.take(4))[0]
    def u64(self): return struct.unpack("<Q", self.take(8))[0]
    def f32(self): return struct.unpack("<f", self.take(4))[0]

    def boolean(self):
        v = self.u8()
        if v > 1:
            raise ValueError("invalid bool")
        return v == 1

    def string(self): return self.take(self.u64()).decode("utf-8")
    def vec_u8(self): return self.take(self.u64())
    def opt(self, get): return get() if self.boolean() else None


# ------------------------------------------------------------------ TiledImage rule (tiled_image.rs:50-104, 271-293)
def tiles_from_image(img):
    """from_rgba_image: [(cx, cy, 16384 bytes)] in flat-index order; a chunk is kept iff some alpha inside the canvas != 0"""
    h, w = img.shape[:2]
    out = []
    for cy in range((h + CHUNK - 1) // CHUNK):
        for cx in range((w + CHUNK - 1) // CHUNK):
            part = img[cy * CHUNK:(cy + 1) * CHUNK, cx * CHUNK:(cx + 1) * CHUNK]
            if not (part[..., 3] != 0).any():
                continue
            c = np.zeros((CHUNK, CHUNK, 4), np.uint8)
            c[:part.shape[0], :part.shape[1]] = part
            out.append((cx, cy, c.tobytes()))
    return out


def image_from_tiles(chunks, w, h):
    """to_rgba_image through set_chunk's flat index (tiled_image.rs:660-662,876-881)"""
    cpr, rows = (w + CHUNK - 1) // CHUNK, (h + CHUNK - 1) // CHUNK
    slots = {}
    for cx, cy, px in chunks:
        idx = (cy * cpr + cx) & 0xFFFFFFFF
        if idx < cpr * rows:
            slots[idx] = px
    img = np.zeros((h, w, 4), np.uint8)
    for idx, px in slots.items():
        cx, cy = idx % cpr, idx // cpr
        c = np.frombuffer(px, np.uint8).reshape(CHUNK, CHUNK, 4)
        part = img[cy * CHUNK:(cy + 1) * CHUNK, cx * CHUNK:(cx + 1) * CHUNK]
        part[...] = c[:part.shape[0], :part.shape[1]]
    return img


# ------------------------------------------------------------------ records
def adjustment_bytes(kind, params=()):
    """AdjustmentLayerData { kind: AdjustmentKind } (layers.rs:247-273); kind: 0 Exposure{ev} 1 BrightnessContrast{b,c} 2 Invert
    3 ChannelMixer{red[4],green[4],blue[4],alpha[4]}"""
    w = W()
    w.u32(kind)
    for v in params:
        w.f32(v)
    return bytes(w.b)


def _put_chunks(w, chunks):
    w.u64(len(chunks))
    for cx, cy, px in chunks:
        w.u32(cx); w.u32(cy); w.vec_u8(px)


def _get_chunks(r):
    return [(r.u32(), r.u32(), r.vec_u8()) for _ in range(r.u64())]


DEFAULT_HDR = {"enabled": False, "max_luminance_nits": None, "reference_white_nits": None, "transfer_function": None}
DEFAULT_META = {"source_format": None, "source_name": None, "color_profile_name": None, "png_text_chunks": [], "raw_png_chunks": []}
_DEEP_FMT = {0: ("<B", 1), 1: ("<H", 2), 2: ("<H", 2), 3: ("<f", 4)}


def encode(project):
    """project: {version, width, height, active_layer_index, layers[, folders, next_layer_folder_id]}; every layer is a dict with
    the fields of LayerDataV<version> (io.rs:108-208); V0 layers carry `pixels` (flat bytes) instead of `chunks`"""
    v = project["version"]
    w = W()
    w.string("PFE%d" % v)
    w.u32(project["width"]); w.u32(project["height"]); w.u64(project["active_layer_index"])
    if v == 3:
        folders = project.get("folders", [])
        w.u64(len(folders))
        for f in folders:
            w.u64(f["id"]); w.string(f["name"]); w.boolean(f["visible"]); w.boolean(f["collapsed"])
            w.opt(f.get("insert_above_layer"), w.u64)
            w.opt(f.get("color_index"), w.u8)
        w.u64(project.get("next_layer_folder_id", 1))
    w.u64(len(project["layers"]))
    for L in project["layers"]:
        w.string(L["name"]); w.boolean(L["visible"])
        if v == 3:
            w.opt(L.get("folder_id"), w.u64)
        w.f32(L["opacity"]); w.u8(L["blend_mode"])
        if v == 0:
            w.vec_u8(L["pixels"])
            continue
        if v >= 2:
            w.u8(L.get("layer_type", 0))
        _put_chunks(w, L["chunks"])
        if v == 2:
            w.opt(L.get("text_data"), w.vec_u8)
        if v == 3:
            w.opt(L.get("content_data"), w.vec_u8)
            w.u32(L.get("pixel_format", 0))
            hdr = L.get("hdr_metadata", DEFAULT_HDR)
            w.boolean(hdr["enabled"]); w.opt(hdr["max_luminance_nits"], w.f32); w.opt(hdr["reference_white_nits"], w.f32)
            w.opt(hdr["transfer_function"], w.string)
            m = L.get("source_metadata", DEFAULT_META)
            w.opt(m["source_format"], w.string); w.opt(m["source_name"], w.string); w.opt(m["color_profile_name"], w.string)
            w.u64(len(m["png_text_chunks"]))
            for k, t in m["png_text_chunks"]:
                w.string(k); w.string(t)
            w.u64(len(m["raw_png_chunks"]))
            for c in m["raw_png_chunks"]:
                w.vec_u8(c)
            w.u32(L.get("webp_frame_compression", 1))
            deep = L.get("deep_pixels")
            w.boolean(deep is not None)
            if deep is not None:
                variant, values = deep
                w.u32(variant); w.u64(len(values))
                for x in values:
                    w.raw(struct.pack(_DEEP_FMT[variant][0], x))
    return bytes(w.b)


def decode(raw):
    r = R(raw)
    magic = r.string()
    assert magic in ("PFE0", "PFE1", "PFE2", "PFE3"), magic
    v = int(magic[3])
    p = {"version": v, "width": r.u32(), "height": r.u32(), "active_layer_index": r.u64()}
    if v == 3:
        p["folders"] = []
        for _ in range(r.u64()):
            f = {"id": r.u64(), "name": r.string(), "visible": r.boolean(), "collapsed": r.boolean()}
            f["insert_above_layer"] = r.opt(r.u64)
            f["color_index"] = r.opt(r.u8)
            p["folders"].append(f)
        p["next_layer_folder_id"] = r.u64()
    p["layers"] = []
    for _ in range(r.u64()):
        L = {"name": r.string(), "visible": r.boolean()}
        if v == 3:
            L["folder_id"] = r.opt(r.u64)
        L["opacity"] = r.f32(); L["blend_mode"] = r.u8()
        if v == 0:
            L["pixels"] = r.vec_u8()
            p["layers"].append(L)
            continue
        if v >= 2:
            L["layer_type"] = r.u8()
        L["chunks"] = _get_chunks(r)
        if v == 2:
            L["text_data"] = r.opt(r.vec_u8)
        if v == 3:
            L["content_data"] = r.opt(r.vec_u8)
            L["pixel_format"] = r.u32()
            L["hdr_metadata"] = {"enabled": r.boolean(), "max_luminance_nits": r.opt(r.f32), "reference_white_nits": r.opt(r.f32),
                                 "transfer_function": r.opt(r.string)}
            m = {"source_format": r.opt(r.string), "source_name": r.opt(r.string), "color_profile_name": r.opt(r.string)}
            m["png_text_chunks"] = [(r.string(), r.string()) for _ in range(r.u64())]
            m["raw_png_chunks"] = [r.vec_u8() for _ in range(r.u64())]
            L["source_metadata"] = m
            L["webp_frame_compression"] = r.u32()
            if r.boolean():
                variant = r.u32()
                fmt, size = _DEEP_FMT[variant]
                L["deep_pixels"] = (variant, [struct.unpack(fmt, r.take(size))[0] for _ in range(r.u64())])
            else:
                L["deep_pixels"] = None
        p["layers"].append(L)
    p["_consumed"] = r.p
    return p


def raster_layer(name, img, opacity=1.0, blend_mode=0, visible=True, **extra):
    L = {"name": name, "visible": visible, "opacity": opacity, "blend_mode": blend_mode, "layer_type": 0, "chunks": tiles_from_image(img)}
    L.update(extra)
    return L
