"""Host-side mirror of the reference's operator surface, over the C ABI.

``GpuRenderer`` keeps the method names and argument meaning of ``paintfe::gpu::GpuRenderer``
(ref: src/gpu/renderer.rs:249-947) and of the pure ``_core`` functions the dialogs / Rhai host call
(ref: SURVEY.md §8b B1-B4).  Images are ``numpy.uint8`` arrays of shape (h, w, 4) — the Python spelling of the
reference's ``&[u8]`` + ``(w, h)``.  All pixel work happens in libpfx.so's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import Brush, ChainOp, LayerInfo, PfxError, ScriptResult

DENSE, FROM_FLAT, IN_PLACE = 0, 1, 2

ADJUST_OPS = ["invert", "invert_alpha", "sepia", "brightness_contrast", "hsl", "exposure", "highlights_shadows",
              "temperature_tint", "threshold", "posterize", "color_balance", "gradient_map", "black_and_white",
              "vibrance", "lut_rgba", "desaturate"]
RHAI_OPS = ["invert", "desaturate", "sepia", "sepia_strength", "brightness_contrast", "hsl", "exposure", "levels"]
BLEND_MODES = ["normal", "multiply", "screen", "additive", "reflect", "glow", "color_burn", "color_dodge", "overlay",
               "difference", "negation", "lighten", "darken", "xor", "overwrite", "hard_light", "soft_light",
               "exclusion", "subtract", "divide", "linear_burn", "vivid_light", "linear_light", "pin_light",
               "hard_mix"]  # BlendMode::to_u8 order, ref: src/canvas/layers.rs:125-153
LAYER_RASTER, ADJ_EXPOSURE, ADJ_BRIGHTNESS_CONTRAST, ADJ_INVERT, ADJ_CHANNEL_MIXER = range(5)


def _u8(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint8)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _fparams(params: Sequence[float]):
    arr = (C.c_float * max(len(params), 1))(*[float(x) for x in params])
    return arr, C.c_uint32(len(params))


NOISE_TYPES = ["uniform", "gaussian", "perlin"]                       # NoiseType
HALFTONE_SHAPES = ["circle", "square", "diamond", "line"]             # HalftoneShape
GRID_STYLES = ["lines", "checkerboard"]                               # GridStyle
OUTLINE_MODES = ["outside", "inside", "center"]                       # OutlineMode
COLOR_FILTER_MODES = ["multiply", "screen", "overlay", "soft_light"]  # ColorFilterMode
RESIZE_FILTERS = ["nearest", "bilinear", "bicubic", "lanczos3"]       # ScriptFilterType / Interpolation
CANVAS_OPS = ["flip_horizontal", "flip_vertical", "rotate_90cw", "rotate_90ccw", "rotate_180", "resize_image", "resize_canvas"]  # CanvasOpRequest


def _enum(names, v):
    return C.c_int(names.index(v) if isinstance(v, str) else int(v))


def _c4(color):
    return (C.c_uint8 * 4)(*[int(v) for v in color])


def _f4(color):
    return None if color is None else (C.c_float * 4)(*[float(v) for v in color])


# The rest of the effect bank (ref: src/ops/effects/*.rs `*_core`): name -> marshaller of the reference's parameters, in the
# reference's order.  GpuRenderer grows `<name>_core(img, ..., mask=None)` and `<name>_dev(src_ptr, dst_ptr, w, h, ..., mask_ptr=0)`.
_EFFECTS = {
    "zoom_blur": lambda center_x, center_y, strength, samples, tint_color=None, tint_strength=0.0: [
        C.c_float(center_x), C.c_float(center_y), C.c_float(strength), C.c_uint32(samples), _f4(tint_color), C.c_float(tint_strength)],
    "crystallize": lambda cell_size, seed: [C.c_float(cell_size), C.c_uint32(seed)],
    "dents": lambda scale, amount, seed, octaves, roughness, pinch=False, wrap=False: [
        C.c_float(scale), C.c_float(amount), C.c_uint32(seed), C.c_uint32(octaves), C.c_float(roughness), C.c_int(int(pinch)), C.c_int(int(wrap))],
    "bulge": lambda amount, origin=(0.5, 0.5): [C.c_float(amount), C.c_float(origin[0]), C.c_float(origin[1])],
    "twist": lambda angle_deg, origin=(0.5, 0.5): [C.c_float(angle_deg), C.c_float(origin[0]), C.c_float(origin[1])],
    "add_noise": lambda amount, noise_type, monochrome, seed, scale, octaves: [
        C.c_float(amount), _enum(NOISE_TYPES, noise_type), C.c_int(int(monochrome)), C.c_uint32(seed), C.c_float(scale), C.c_uint32(octaves)],
    "reduce_noise": lambda strength, radius: [C.c_float(strength), C.c_uint32(radius)],
    "vignette": lambda amount, softness: [C.c_float(amount), C.c_float(softness)],
    "halftone": lambda dot_size, angle_deg, shape="circle": [C.c_float(dot_size), C.c_float(angle_deg), _enum(HALFTONE_SHAPES, shape)],
    "grid": lambda cell_w, cell_h, line_width, color, style="lines", opacity=1.0: [
        C.c_uint32(cell_w), C.c_uint32(cell_h), C.c_uint32(line_width), _c4(color), _enum(GRID_STYLES, style), C.c_float(opacity)],
    "canvas_border": lambda width, color: [C.c_uint32(width), _c4(color)],
    "shadow": lambda offset_x, offset_y, blur_radius, widen_radius, color, opacity: [
        C.c_int32(offset_x), C.c_int32(offset_y), C.c_float(blur_radius), C.c_int(int(widen_radius)), _c4(color), C.c_float(opacity)],
    "outline": lambda width, color, mode="outside", anti_alias=True: [
        C.c_uint32(width), _c4(color), _enum(OUTLINE_MODES, mode), C.c_int(int(anti_alias))],
    "pixel_drag": lambda seed, amount, distance, direction: [C.c_uint32(seed), C.c_float(amount), C.c_uint32(distance), C.c_float(direction)],
    "rgb_displace": lambda r_off, g_off, b_off: [C.c_int32(r_off[0]), C.c_int32(r_off[1]), C.c_int32(g_off[0]), C.c_int32(g_off[1]),
                                                 C.c_int32(b_off[0]), C.c_int32(b_off[1])],
    "ink": lambda edge_strength, threshold: [C.c_float(edge_strength), C.c_float(threshold)],
    "oil_painting": lambda radius, levels: [C.c_uint32(radius), C.c_uint32(levels)],
    "color_filter": lambda filter_color, intensity, mode="multiply": [_c4(filter_color), C.c_float(intensity), _enum(COLOR_FILTER_MODES, mode)],
    "contours": lambda scale, frequency, line_width, line_color, seed, octaves, blend: [
        C.c_float(scale), C.c_float(frequency), C.c_float(line_width), _c4(line_color), C.c_uint32(seed), C.c_uint32(octaves), C.c_float(blend)],
}


def script_check(source: str, w: int = 64, h: int = 64):
    """Language-only evaluation of a script (no device, no image functions; ref: compile_script, scripting.rs:1489): returns
    the console lines or raises PfxError with .line / .col."""
    lib = _lib.load()
    res = ScriptResult()
    st = lib.pfx_script_check(source.encode(), C.c_uint32(w), C.c_uint32(h), C.byref(res))
    if st != _lib.OK:
        err = PfxError(st, res.error.decode(errors="replace"))
        err.line, err.col = res.error_line, res.error_col
        raise err
    return [s for s in res.console.decode(errors="replace").split("\n") if s]


def png_decode(raw: bytes) -> np.ndarray:
    """The CLI's PNG reader on a buffer (pfx_png_decode_mem; ref: load_image_sync, src/io.rs:693-723): (h, w, 4) uint8.  No device needed."""
    lib = _lib.load()
    lib.pfx_png_decode_mem.restype = C.c_int
    lib.pfx_png_free.restype = None
    out, w, h = C.POINTER(C.c_uint8)(), C.c_uint32(), C.c_uint32()
    err = C.create_string_buffer(256)
    buf = (C.c_uint8 * max(len(raw), 1)).from_buffer_copy(raw if raw else b"\0")
    st = lib.pfx_png_decode_mem(buf, C.c_size_t(len(raw)), C.byref(out), C.byref(w), C.byref(h), err, C.c_size_t(256))
    if st != _lib.OK:
        raise PfxError(st, err.value.decode(errors="replace"))
    try:
        return np.ctypeslib.as_array(out, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.pfx_png_free(out)


def _install_effects(cls):
    def make(name, marshal):
        def core(self, img, *a, mask=None, **kw):
            return self._img_call(getattr(self._lib, f"pfx_{name}_core"), img, *marshal(*a, **kw), mask=mask)

        def dev(self, src_ptr, dst_ptr, w, h, *a, mask_ptr=0, **kw):
            self._check(getattr(self._lib, f"pfx_{name}_dev")(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h),
                                                              *marshal(*a, **kw), C.c_void_p(mask_ptr or None)))
        core.__name__, dev.__name__ = f"{name}_core", f"{name}_dev"
        core.__doc__ = dev.__doc__ = f"{name}_core of src/ops/effects/ (see include/pfx.h: pfx_{name}_core / pfx_{name}_dev)"
        setattr(cls, f"{name}_core", core)
        setattr(cls, f"{name}_dev", dev)
    for name, marshal in _EFFECTS.items():
        make(name, marshal)
    return cls


@_install_effects
class GpuRenderer:
    """One HIP device + stream (ref: GpuRenderer::try_new, src/gpu/renderer.rs:261)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        st = self._lib.pfx_ctx_create(C.c_int(device), C.byref(h))
        if st != _lib.OK:
            raise PfxError(st, self._lib.pfx_last_error(None).decode())
        self._h = h
        self.available = True  # ref: renderer.rs:241

    @classmethod
    def try_new(cls, device: int = 0) -> Optional["GpuRenderer"]:
        try:
            return cls(device)
        except (PfxError, ImportError, OSError):
            return None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pfx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ plumbing
    def _check(self, st: int):
        if st != _lib.OK:
            raise PfxError(st, self._lib.pfx_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    @property
    def stream(self) -> int:
        return int(self._lib.pfx_ctx_stream(self._h) or 0)

    def set_stream(self, hip_stream: int):
        """hip_stream: an integer hipStream_t (0 = the device's default stream, e.g. torch's default), or None to go
        back to the context's own stream."""
        if hip_stream is None:
            self._check(self._lib.pfx_ctx_set_stream(self._h, None, C.c_int(0)))
        else:
            self._check(self._lib.pfx_ctx_set_stream(self._h, C.c_void_p(hip_stream or None), C.c_int(1)))

    def set_exact(self, exact: bool):
        self._check(self._lib.pfx_ctx_set_exact(self._h, C.c_int(int(exact))))

    def synchronize(self):
        self._check(self._lib.pfx_ctx_synchronize(self._h))

    def _img_call(self, fn, img, *args, mask=None, has_mask=True, out=None):
        src = _u8(img)
        h, w = src.shape[:2]
        dst = np.empty_like(src) if out is None else out   # out: a caller's buffer (e.g. page-locked: host_alloc)
        m = None if mask is None else _u8(mask)
        a = [self._h, _p(src), _p(dst), C.c_uint32(w), C.c_uint32(h), *args]
        if has_mask:
            a.append(_p(m))
        self._check(fn(*a))
        return dst

    # ------------------------------------------------------------------ B1: GpuRenderer filter methods
    def blur_rgba(self, data, sigma: float):
        return self._img_call(self._lib.pfx_blur_rgba, data, C.c_float(sigma), has_mask=False)

    def brightness_contrast_rgba(self, data, brightness: float, contrast: float):
        return self._img_call(self._lib.pfx_brightness_contrast_rgba, data, C.c_float(brightness), C.c_float(contrast),
                              has_mask=False)

    def hsl_rgba(self, data, hue: float, sat: float, light: float):
        return self._img_call(self._lib.pfx_hsl_rgba, data, C.c_float(hue), C.c_float(sat), C.c_float(light),
                              has_mask=False)

    def invert_rgba(self, data, out=None):
        return self._img_call(self._lib.pfx_invert_rgba, data, has_mask=False, out=out)

    def median_rgba(self, data, radius: int):
        """Returns None where the reference does (device path does not cover the radius)."""
        try:
            return self._img_call(self._lib.pfx_median_rgba, data, C.c_uint32(radius), has_mask=False)
        except PfxError as e:
            if e.status == _lib.ERR_UNSUPPORTED:
                return None
            raise

    # ------------------------------------------------------------------ B4: `_core` functions
    def gaussian_blur_core(self, img, sigma: float, mask=None):
        return self._img_call(self._lib.pfx_gaussian_blur_core, img, C.c_float(sigma), mask=mask)

    def box_blur_core(self, img, radius: float, mask=None):
        return self._img_call(self._lib.pfx_box_blur_core, img, C.c_float(radius), mask=mask)

    # GpuLiquifyPipeline source cache (liquify.rs:166-176)
    def warp_set_source(self, src):
        a = _u8(src)
        self._check(self._lib.pfx_warp_set_source(self._h, _p(a), C.c_uint32(a.shape[1]), C.c_uint32(a.shape[0])))

    def warp_invalidate_source(self):
        self._check(self._lib.pfx_warp_invalidate_source(self._h))

    def warp_displacement_cached(self, disp, w: int, h: int):
        d = np.ascontiguousarray(disp, dtype=np.float32)
        out = np.zeros((h, w, 4), np.uint8)
        self._check(self._lib.pfx_warp_displacement_cached(self._h, d.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), _p(out)))
        return out

    def median_core(self, img, radius: int, mask=None):
        return self._img_call(self._lib.pfx_median_core, img, C.c_uint32(radius), mask=mask)

    def pixelate_core(self, img, block_size: int, mask=None):
        return self._img_call(self._lib.pfx_pixelate_core, img, C.c_uint32(block_size), mask=mask)

    def sharpen_core(self, img, amount: float, radius: float, mask=None):      # stylize.rs:96
        return self._img_call(self._lib.pfx_sharpen_core, img, C.c_float(amount), C.c_float(radius), mask=mask)

    def glow_core(self, img, radius: float, intensity: float, mask=None):      # stylize.rs:26
        return self._img_call(self._lib.pfx_glow_core, img, C.c_float(radius), C.c_float(intensity), mask=mask)

    def bokeh_blur_core(self, img, radius: float, mask=None):                   # blur.rs:22
        return self._img_call(self._lib.pfx_bokeh_blur_core, img, C.c_float(radius), mask=mask)

    def motion_blur_core(self, img, angle_deg: float, distance: float, mask=None):  # blur.rs:144
        return self._img_call(self._lib.pfx_motion_blur_core, img, C.c_float(angle_deg), C.c_float(distance), mask=mask)

    def adjust(self, img, op, params: Sequence[float] = (), lut=None, mask=None, sparse: int = DENSE):
        opi = ADJUST_OPS.index(op) if isinstance(op, str) else int(op)
        src = _u8(img)
        h, w = src.shape[:2]
        dst = np.empty_like(src)
        arr, n = _fparams(params)
        l = None if lut is None else _u8(lut)
        m = None if mask is None else _u8(mask)
        self._check(self._lib.pfx_adjust(self._h, _p(src), _p(dst), C.c_uint32(w), C.c_uint32(h), C.c_int(opi), arr, n,
                                         _p(l), _p(m), C.c_int(sparse)))
        return dst

    def rhai_adjust(self, img, op, params: Sequence[float] = ()):
        opi = RHAI_OPS.index(op) if isinstance(op, str) else int(op)
        px = _u8(img).copy()
        h, w = px.shape[:2]
        arr, n = _fparams(params)
        self._check(self._lib.pfx_rhai_adjust(self._h, _p(px), C.c_uint32(w), C.c_uint32(h), C.c_int(opi), arr, n))
        return px

    def auto_levels(self, img, mask=None):
        return self._img_call(self._lib.pfx_auto_levels, img, mask=mask)

    def levels(self, img, in_black, in_white, gamma, out_black, out_white, mask=None, sparse=FROM_FLAT):
        lv = self.build_levels_lut(in_black, in_white, gamma, out_black, out_white)
        luts = np.stack([lv, lv, lv, np.arange(256, dtype=np.uint8)])
        return self.adjust(img, "lut_rgba", lut=luts, mask=mask, sparse=sparse)

    def build_levels_lut(self, in_black, in_white, gamma, out_black, out_white) -> np.ndarray:
        lut = np.zeros(256, np.uint8)
        self._lib.pfx_build_levels_lut(C.c_float(in_black), C.c_float(in_white), C.c_float(gamma), C.c_float(out_black),
                                       C.c_float(out_white), _p(lut))
        return lut

    def build_curves_lut(self, points) -> np.ndarray:
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        lut = np.zeros(256, np.uint8)
        self._lib.pfx_build_curves_lut(_p(pts), C.c_uint32(len(pts)), _p(lut))
        return lut

    def tiled_roundtrip(self, img):
        return self._img_call(self._lib.pfx_tiled_roundtrip, img, has_mask=False)

    def chunk_populated(self, img):
        src = _u8(img)
        h, w = src.shape[:2]
        out = np.zeros(((h + 63) // 64, (w + 63) // 64), np.uint8)
        self._check(self._lib.pfx_chunk_populated(self._h, _p(src), C.c_uint32(w), C.c_uint32(h), _p(out)))
        return out

    # ------------------------------------------------------------------ B2: compositor
    def ensure_layer_texture(self, layer_idx: int, data, generation: int):
        src = _u8(data)
        h, w = src.shape[:2]
        self._check(self._lib.pfx_layer_upload(self._h, C.c_uint32(layer_idx), C.c_uint32(w), C.c_uint32(h), _p(src),
                                               C.c_uint64(generation)))

    def update_layer_rect(self, layer_idx: int, x: int, y: int, data):
        reg = _u8(data)
        rh, rw = reg.shape[:2]
        self._check(self._lib.pfx_layer_update_rect(self._h, C.c_uint32(layer_idx), C.c_uint32(x), C.c_uint32(y),
                                                    C.c_uint32(rw), C.c_uint32(rh), _p(reg)))

    def set_layer_mask(self, layer_idx: int, conceal):
        m = None if conceal is None else _u8(conceal)
        self._check(self._lib.pfx_layer_set_mask(self._h, C.c_uint32(layer_idx), _p(m)))

    def remove_layer(self, layer_idx: int):
        self._check(self._lib.pfx_layer_remove(self._h, C.c_uint32(layer_idx)))

    def clear_layers(self):
        self._check(self._lib.pfx_layer_clear(self._h))

    def active_texture_count(self) -> int:
        return int(self._lib.pfx_layer_count(self._h))

    def active_texture_memory(self) -> int:
        return int(self._lib.pfx_layer_memory(self._h))

    @staticmethod
    def _infos(layer_info: Iterable):
        """layer_info: tuples (layer_idx, opacity, visible, blend_mode_u8[, kind, adj]) bottom -> top."""
        items = list(layer_info)
        arr = (LayerInfo * max(len(items), 1))()
        for i, t in enumerate(items):
            arr[i].layer_idx, arr[i].opacity, arr[i].visible, arr[i].blend_mode = int(t[0]), float(t[1]), int(bool(t[2])), int(t[3])
            if len(t) > 4:
                arr[i].kind = int(t[4])
                for j, v in enumerate(t[5] if len(t) > 5 else ()):
                    arr[i].adj[j] = float(v)
        return arr, len(items)

    def composite(self, canvas_w: int, canvas_h: int, layer_info: Iterable):
        arr, n = self._infos(layer_info)
        dst = np.empty((canvas_h, canvas_w, 4), np.uint8)
        self._check(self._lib.pfx_composite(self._h, C.c_uint32(canvas_w), C.c_uint32(canvas_h), arr, C.c_uint32(n), _p(dst)))
        return dst

    def composite_preview(self, canvas_w: int, canvas_h: int, layer_info: Iterable, preview_pixels, active_layer: int, blend_mode: int = 0,
                          is_eraser: bool = False, replaces_layer: bool = False, chunk_present=None):
        """composite with the tool preview layer folded into layer_info[active_layer] (ref: canvas_state.rs:593-658)"""
        arr, n = self._infos(layer_info)
        dst = np.empty((canvas_h, canvas_w, 4), np.uint8)
        pv = _lib.Preview(active_layer, blend_mode, int(is_eraser), int(replaces_layer), 0)
        px = _u8(preview_pixels)
        cp = None if chunk_present is None else _u8(chunk_present)
        self._check(self._lib.pfx_composite_preview(self._h, C.c_uint32(canvas_w), C.c_uint32(canvas_h), arr, C.c_uint32(n), _p(px), _p(cp),
                                                    C.byref(pv), _p(dst)))
        return dst

    def composite_dirty_readback(self, canvas_w, canvas_h, layer_info, rect):
        x, y, rw, rh = rect
        arr, n = self._infos(layer_info)
        dst = np.empty((rh, rw, 4), np.uint8)
        self._check(self._lib.pfx_composite_region(self._h, C.c_uint32(canvas_w), C.c_uint32(canvas_h), arr, C.c_uint32(n),
                                                   C.c_uint32(x), C.c_uint32(y), C.c_uint32(rw), C.c_uint32(rh), _p(dst)))
        return dst

    def blend_pixels(self, base, top, mode: int, opacity: float):
        b, t = _u8(base).reshape(-1, 4), _u8(top).reshape(-1, 4)
        dst = np.empty_like(b)
        self._check(self._lib.pfx_blend_pixels(self._h, _p(b), _p(t), _p(dst), C.c_size_t(len(b)), C.c_uint8(mode),
                                               C.c_float(opacity)))
        return dst

    # ------------------------------------------------------------------ B3: warp
    def warp_displacement(self, src, disp):
        s = _u8(src)
        d = np.ascontiguousarray(disp, np.float32)
        sh, sw = s.shape[:2]
        h, w = d.shape[:2]
        dst = np.empty((h, w, 4), np.uint8)
        self._check(self._lib.pfx_warp_displacement(self._h, _p(s), C.c_uint32(sw), C.c_uint32(sh), _p(d), C.c_uint32(w),
                                                    C.c_uint32(h), _p(dst)))
        return dst

    def generate_displacement(self, deformed_points, cols, rows, w, h, original_points=None):
        d = np.ascontiguousarray(deformed_points, np.float32)
        o = None if original_points is None else np.ascontiguousarray(original_points, np.float32)
        out = np.empty((h, w, 2), np.float32)
        self._check(self._lib.pfx_mesh_displacement(self._h, _p(o), _p(d), C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w),
                                                    C.c_uint32(h), _p(out)))
        return out

    def warp_mesh_catmull_rom(self, src, original_points, deformed_points, cols, rows):
        s = _u8(src)
        h, w = s.shape[:2]
        o = np.ascontiguousarray(original_points, np.float32)
        d = np.ascontiguousarray(deformed_points, np.float32)
        dst = np.empty_like(s)
        self._check(self._lib.pfx_warp_mesh_catmull_rom(self._h, _p(s), _p(o), _p(d), C.c_uint32(cols), C.c_uint32(rows),
                                                        C.c_uint32(w), C.c_uint32(h), _p(dst)))
        return dst

    def displacement_brush(self, disp: np.ndarray, mode, cx, cy, dx, dy, radius, strength):
        assert disp.dtype == np.float32 and disp.flags.c_contiguous
        h, w = disp.shape[:2]
        self._lib.pfx_displacement_brush(_p(disp), C.c_uint32(w), C.c_uint32(h), C.c_int(mode), C.c_float(cx), C.c_float(cy),
                                         C.c_float(dx), C.c_float(dy), C.c_float(radius), C.c_float(strength))
        return disp

    # ------------------------------------------------------------------ brush
    @staticmethod
    def make_brush(size, hardness, anti_aliased, color=(0, 0, 0, 1), flow=1.0, is_eraser=False, mode=0) -> Brush:
        b = Brush()
        b.size, b.hardness, b.flow = size, hardness, flow
        for i in range(4):
            b.color[i] = color[i]
        b.anti_aliased, b.is_eraser, b.mode = int(anti_aliased), int(is_eraser), int(mode)
        return b

    @staticmethod
    def make_dynamics(scatter=0.0, hue_jitter=0.0, brightness_jitter=0.0, stamp_counter=0, tip_mask=None, tip_rotation=0.0,
                      tip_random_rotation=False, tip_rotation_range=(0.0, 360.0)):
        """pfx_brush_dynamics (ToolProperties scatter / jitter / image tip; ref: state.rs:112-128).  Returns (struct, keep-alive)."""
        d = _lib.BrushDynamics(scatter, hue_jitter, brightness_jitter, stamp_counter, None, 0, tip_rotation, int(tip_random_rotation),
                               tip_rotation_range[0], tip_rotation_range[1])
        keep = None
        if tip_mask is not None:
            keep = _u8(tip_mask)
            d.tip_mask = keep.ctypes.data
            d.tip_mask_size = keep.shape[0]
        return d, keep

    def brush_tip_rescale(self, src_mask, brush_size: float, hardness: float):        # rebuild_tip_mask, brush_render.rs:404
        src = _u8(src_mask)
        n = max(int(np.ceil(np.float32(brush_size))), 1)
        out = np.zeros((n, n), np.uint8)
        self._lib.pfx_brush_tip_rescale.restype = C.c_uint32
        got = self._lib.pfx_brush_tip_rescale(_p(src), C.c_uint32(src.shape[0]), C.c_float(brush_size), C.c_float(hardness), _p(out))
        assert got == n
        return out

    def brush_stamps(self, target, brush: Brush, points, selection=None, dyn=None):
        t = _u8(target).copy()
        h, w = t.shape[:2]
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        sel = None if selection is None else _u8(selection)
        d, keep = self.make_dynamics(**dyn) if dyn is not None else (None, None)
        self._check(self._lib.pfx_brush_stamps_ex(self._h, _p(t), C.c_uint32(w), C.c_uint32(h), C.byref(brush), C.byref(d) if d is not None else None,
                                                  _p(pts), C.c_uint32(len(pts)), _p(sel)))
        return t

    def brush_line(self, target, brush: Brush, p0, p1, selection=None, dyn=None):
        t = _u8(target).copy()
        h, w = t.shape[:2]
        sel = None if selection is None else _u8(selection)
        d, keep = self.make_dynamics(**dyn) if dyn is not None else (None, None)
        self._check(self._lib.pfx_brush_line_ex(self._h, _p(t), C.c_uint32(w), C.c_uint32(h), C.byref(brush), C.byref(d) if d is not None else None,
                                                C.c_float(p0[0]), C.c_float(p0[1]), C.c_float(p1[0]), C.c_float(p1[1]), _p(sel)))
        return t

    def brush_commit(self, layer, preview, blend_mode: int, is_eraser=False, selection=None):
        l = _u8(layer).copy()
        h, w = l.shape[:2]
        p = _u8(preview)
        sel = None if selection is None else _u8(selection)
        self._check(self._lib.pfx_brush_commit(self._h, _p(l), _p(p), C.c_uint32(w), C.c_uint32(h), C.c_uint8(blend_mode),
                                               C.c_int(int(is_eraser)), _p(sel)))
        return l

    # ------------------------------------------------------------------ script front-end
    def resize_image(self, img, new_w: int, new_h: int, filter="bilinear"):   # transform.rs:347 (imageops::resize)
        src = _u8(img)
        h, w = src.shape[:2]
        dst = np.empty((new_h, new_w, 4), np.uint8)
        self._check(self._lib.pfx_resize_image(self._h, _p(src), C.c_uint32(w), C.c_uint32(h), _p(dst), C.c_uint32(new_w), C.c_uint32(new_h),
                                               _enum(RESIZE_FILTERS, filter)))
        return dst

    def affine_transform(self, img, canvas_w: int, canvas_h: int, rotation_z=0.0, rotation_x=0.0, rotation_y=0.0, scale=1.0, offset=(0.0, 0.0),
                         interpolation="bilinear"):                            # transform.rs:826 apply_affine
        src = _u8(img)
        h, w = src.shape[:2]
        dst = np.empty((canvas_h, canvas_w, 4), np.uint8)
        self._check(self._lib.pfx_affine_transform(self._h, _p(src), C.c_uint32(w), C.c_uint32(h), _p(dst), C.c_uint32(canvas_w), C.c_uint32(canvas_h),
                                                   C.c_float(rotation_z), C.c_float(rotation_x), C.c_float(rotation_y), C.c_float(scale),
                                                   C.c_float(offset[0]), C.c_float(offset[1]), _enum(RESIZE_FILTERS, interpolation)))
        return dst

    def flip_rotate(self, img, op):                                            # transform.rs flip_canvas_* / rotate_canvas_*
        src = _u8(img)
        h, w = src.shape[:2]
        opi = _enum(CANVAS_OPS, op)
        dst = np.empty((w, h, 4) if opi.value in (2, 3) else (h, w, 4), np.uint8)
        self._check(self._lib.pfx_flip_rotate(self._h, _p(src), C.c_uint32(w), C.c_uint32(h), _p(dst), opi))
        return dst

    def resize_canvas(self, img, new_w: int, new_h: int, anchor=(0, 0), fill=(0, 0, 0, 0)):   # transform.rs:382
        src = _u8(img)
        h, w = src.shape[:2]
        dst = np.empty((new_h, new_w, 4), np.uint8)
        self._check(self._lib.pfx_resize_canvas(self._h, _p(src), C.c_uint32(w), C.c_uint32(h), _p(dst), C.c_uint32(new_w), C.c_uint32(new_h),
                                                C.c_uint32(anchor[0]), C.c_uint32(anchor[1]), _c4(fill)))
        return dst

    def displacement_brushes_dev(self, disp_ptr: int, w: int, h: int, dabs):
        """DisplacementField::apply_* on a device-resident w*h*2 f32 field; dabs = [(mode, cx, cy, delta_x, delta_y, radius, strength), ...]"""
        arr = (_lib.DispDab * max(len(dabs), 1))(*[_lib.DispDab(int(d[0]), *[float(v) for v in d[1:]]) for d in dabs])
        self._check(self._lib.pfx_displacement_brushes_dev(self._h, C.c_void_p(disp_ptr), C.c_uint32(w), C.c_uint32(h), arr, C.c_uint32(len(dabs))))

    def brush_stamps_dev(self, target_ptr: int, w: int, h: int, brush: Brush, points, selection_ptr: int = 0, dyn=None):
        """pfx_brush_stamps_ex_dev: the stamp loop into a device-resident w*h RGBA8 target (the interactive path: the preview layer stays on the GPU)"""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        d, keep = self.make_dynamics(**dyn) if dyn is not None else (None, None)
        self._check(self._lib.pfx_brush_stamps_ex_dev(self._h, C.c_void_p(target_ptr), C.c_uint32(w), C.c_uint32(h), C.byref(brush),
                                                      C.byref(d) if d is not None else None, _p(pts), C.c_uint32(len(pts)), C.c_void_p(selection_ptr or None)))

    def resize_image_dev(self, src_ptr, w, h, dst_ptr, new_w, new_h, filter="bilinear"):
        self._check(self._lib.pfx_resize_image_dev(self._h, C.c_void_p(src_ptr), C.c_uint32(w), C.c_uint32(h), C.c_void_p(dst_ptr), C.c_uint32(new_w),
                                                   C.c_uint32(new_h), _enum(RESIZE_FILTERS, filter)))

    def execute_script_sync(self, source: str, pixels, mask=None, with_ops: bool = False):
        """execute_script_sync (ref: src/ops/scripting.rs:1733): returns (result_pixels, console_output) — the result may have
        another size than the input (rotate_canvas_90*, resize_canvas) — plus the CanvasOpRequest list with with_ops=True, as
        (kind, w, h, anchor_x, anchor_y) tuples."""
        px = _u8(pixels)
        h, w = px.shape[:2]
        m = None if mask is None else _u8(mask)
        res = ScriptResult()
        out = C.c_void_p()
        st = self._lib.pfx_script_execute(self._h, source.encode(), _p(px), C.c_uint32(w), C.c_uint32(h), _p(m), C.byref(out), C.byref(res))
        if st != _lib.OK:
            msg = res.error.decode(errors="replace") or self._lib.pfx_last_error(self._h).decode()
            err = PfxError(st, msg)
            err.line, err.col = res.error_line, res.error_col
            raise err
        try:
            ow, oh = C.c_uint32(), C.c_uint32()
            self._lib.pfx_script_output_pixels.restype = C.POINTER(C.c_uint8)
            self._lib.pfx_script_output_console_line.restype = C.c_char_p
            p = self._lib.pfx_script_output_pixels(out, C.byref(ow), C.byref(oh))
            result = np.ctypeslib.as_array(p, shape=(oh.value, ow.value, 4)).copy()
            console = [self._lib.pfx_script_output_console_line(out, C.c_uint32(k)).decode(errors="replace")
                       for k in range(self._lib.pfx_script_output_console_lines(out))]
            n_ops = self._lib.pfx_script_output_canvas_ops(out, None, C.c_uint32(0))
            ops_arr = (_lib.CanvasOp * max(n_ops, 1))()
            self._lib.pfx_script_output_canvas_ops(out, ops_arr, C.c_uint32(n_ops))
            ops = [(o.kind, o.w, o.h, o.anchor_x, o.anchor_y) for o in ops_arr[:n_ops]]
        finally:
            self._lib.pfx_script_output_free(out)
        return (result, console, ops) if with_ops else (result, console)

    def script_check(self, source: str, w: int = 64, h: int = 64):
        return script_check(source, w, h)

    # ------------------------------------------------------------------ device tier (raw device pointers as ints)
    def host_alloc(self, shape, dtype=np.uint8) -> np.ndarray:
        """a numpy array on page-locked host memory (pfx_host_alloc): host-buffer calls on it move at the link's rate; release with host_free(array)"""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._check(self._lib.pfx_host_alloc(self._h, C.c_size_t(n), C.byref(p)))
        buf = (C.c_uint8 * n).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._check(self._lib.pfx_host_free(self._h, C.c_void_p(p)))

    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._check(self._lib.pfx_dev_alloc(self._h, C.c_size_t(nbytes), C.byref(p)))
        return int(p.value)

    def dev_free(self, ptr: int):
        self._check(self._lib.pfx_dev_free(self._h, C.c_void_p(ptr)))

    def dev_upload(self, ptr: int, host: np.ndarray):
        a = np.ascontiguousarray(host)
        self._check(self._lib.pfx_dev_upload(self._h, C.c_void_p(ptr), _p(a), C.c_size_t(a.nbytes)))

    def dev_download(self, ptr: int, shape, dtype=np.uint8) -> np.ndarray:
        out = np.empty(shape, dtype)
        self._check(self._lib.pfx_dev_download(self._h, _p(out), C.c_void_p(ptr), C.c_size_t(out.nbytes)))
        return out

    def flatten_dev(self, layer_ptrs: Sequence[int], layer_info: Iterable, w: int, h: int, dst_ptr: int, mask_ptrs=None):
        arr, n = self._infos(layer_info)
        lp = (C.c_void_p * max(n, 1))(*[C.c_void_p(p) for p in layer_ptrs])
        mp = None
        if mask_ptrs is not None:
            mp = (C.c_void_p * max(n, 1))(*[C.c_void_p(p or 0) for p in mask_ptrs])
        self._check(self._lib.pfx_flatten_dev(self._h, lp, mp, arr, C.c_uint32(n), C.c_uint32(w), C.c_uint32(h),
                                              C.c_void_p(dst_ptr)))

    def gaussian_blur_dev(self, src_ptr: int, dst_ptr: int, w: int, h: int, sigma: float, tmp_ptr: int = 0, first_row: int = 0):
        """first_row: index of the buffer's row 0 in the whole image when the buffer is a band of it (pfx_gaussian_blur_band_dev)"""
        self._check(self._lib.pfx_gaussian_blur_band_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w),
                                                         C.c_uint32(h), C.c_float(sigma), C.c_void_p(tmp_ptr or None), C.c_uint32(first_row)))

    def pixelate_dev(self, src_ptr, dst_ptr, w, h, block_size, mask_ptr=0):
        self._check(self._lib.pfx_pixelate_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), C.c_uint32(block_size),
                                               C.c_void_p(mask_ptr or None)))

    def tiled_roundtrip_dev(self, src_ptr, dst_ptr, w, h):
        """TiledImage::from_rgba_image -> to_rgba_image on device buffers (tiled_image.rs:50-104, 271-293): chunks whose alpha is all zero are dropped"""
        self._check(self._lib.pfx_tiled_roundtrip_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h)))

    def adjust_dev(self, src_ptr, dst_ptr, w, h, op, params=(), lut=None, mask_ptr=0, sparse=DENSE):
        opi = ADJUST_OPS.index(op) if isinstance(op, str) else int(op)
        arr, n = _fparams(params)
        l = None if lut is None else _u8(lut)
        self._check(self._lib.pfx_adjust_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h),
                                             C.c_int(opi), arr, n, _p(l), C.c_void_p(mask_ptr or None), C.c_int(sparse)))

    def chain_dev(self, src_ptr, dst_ptr, w, h, ops):
        """pfx_chain_dev: ops = [("gaussian", sigma) | ("box", radius) | ("adjust", name, params[, lut]) | ("rhai", name, params)], applied in order with as few
        passes over memory as the kernels allow; equals the single-op calls one after the other, bit for bit"""
        arr = (ChainOp * max(len(ops), 1))()
        keep = []
        for k, o in enumerate(ops):
            kind = o[0]
            if kind in ("gaussian", "box"):
                arr[k].kind = 2 if kind == "gaussian" else 3
                arr[k].n_params = 1
                arr[k].params[0] = float(o[1])
            else:
                names = ADJUST_OPS if kind == "adjust" else RHAI_OPS
                arr[k].kind = 0 if kind == "adjust" else 1
                arr[k].op = names.index(o[1]) if isinstance(o[1], str) else int(o[1])
                ps = list(o[2]) if len(o) > 2 and o[2] is not None else []
                arr[k].n_params = len(ps)
                for i, v in enumerate(ps):
                    arr[k].params[i] = float(v)
                if len(o) > 3 and o[3] is not None:
                    l = _u8(o[3]); keep.append(l)
                    arr[k].lut = l.ctypes.data
        self._check(self._lib.pfx_chain_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), arr, C.c_uint32(len(ops))))

    def sharpen_dev(self, src_ptr, dst_ptr, w, h, amount, radius, mask_ptr=0):
        self._check(self._lib.pfx_sharpen_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), C.c_float(amount), C.c_float(radius),
                                              C.c_void_p(mask_ptr or None)))

    def glow_dev(self, src_ptr, dst_ptr, w, h, radius, intensity, mask_ptr=0):
        self._check(self._lib.pfx_glow_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), C.c_float(radius), C.c_float(intensity),
                                           C.c_void_p(mask_ptr or None)))

    def box_blur_dev(self, src_ptr, dst_ptr, w, h, radius, mask_ptr=0, tmp_ptr=0):
        self._check(self._lib.pfx_box_blur_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h),
                                               C.c_float(radius), C.c_void_p(mask_ptr or None), C.c_void_p(tmp_ptr or None)))

    def box_blur_band_dev(self, src_ptr, dst_ptr, w, h, radius, mask_ptr=0, tmp_ptr=0, first_row=0):
        """pfx_box_blur_band_dev: a band with its halo rows (rows at least ceil(radius) from the buffer's ends equal the whole image's)"""
        self._check(self._lib.pfx_box_blur_band_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), C.c_float(radius),
                                                    C.c_void_p(mask_ptr or None), C.c_void_p(tmp_ptr or None), C.c_uint32(first_row)))

    def median_band_dev(self, src_ptr, dst_ptr, w, h, radius, mask_ptr=0, first_row=0):
        self._check(self._lib.pfx_median_band_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h), C.c_uint32(radius),
                                                  C.c_void_p(mask_ptr or None), C.c_uint32(first_row)))

    def median_dev(self, src_ptr, dst_ptr, w, h, radius, mask_ptr=0):
        self._check(self._lib.pfx_median_dev(self._h, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_uint32(w), C.c_uint32(h),
                                             C.c_uint32(radius), C.c_void_p(mask_ptr or None)))

    def warp_mesh_catmull_rom_dev(self, src_ptr, orig, deformed, cols, rows, w, h, dst_ptr):
        o = None if orig is None else np.ascontiguousarray(orig, np.float32)
        d = np.ascontiguousarray(deformed, np.float32)
        self._check(self._lib.pfx_warp_mesh_catmull_rom_dev(self._h, C.c_void_p(src_ptr), _p(o), _p(d), C.c_uint32(cols),
                                                            C.c_uint32(rows), C.c_uint32(w), C.c_uint32(h), C.c_void_p(dst_ptr)))

    def mesh_displacement_dev(self, orig, deformed, cols, rows, w, h, disp_ptr):
        o = None if orig is None else np.ascontiguousarray(orig, np.float32)
        d = np.ascontiguousarray(deformed, np.float32)
        self._check(self._lib.pfx_mesh_displacement_dev(self._h, _p(o), _p(d), C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w), C.c_uint32(h),
                                                        C.c_void_p(disp_ptr)))

    def warp_displacement_dev(self, src_ptr, sw, sh, disp_ptr, w, h, dst_ptr):
        self._check(self._lib.pfx_warp_displacement_dev(self._h, C.c_void_p(src_ptr), C.c_uint32(sw), C.c_uint32(sh),
                                                        C.c_void_p(disp_ptr), C.c_uint32(w), C.c_uint32(h), C.c_void_p(dst_ptr)))

    def warp_displacement_band_dev(self, src_ptr, sw, sh, disp_band_ptr, w, band_rows, dst_band_ptr, first_row):
        """rows [first_row, first_row + band_rows) of the warp: the field and the output are bands, the source is whole (pfx_warp_displacement_band_dev)"""
        self._check(self._lib.pfx_warp_displacement_band_dev(self._h, C.c_void_p(src_ptr), C.c_uint32(sw), C.c_uint32(sh), C.c_void_p(disp_band_ptr),
                                                             C.c_uint32(w), C.c_uint32(band_rows), C.c_void_p(dst_band_ptr), C.c_uint32(first_row)))

    def warp_mesh_catmull_rom_band_dev(self, src_ptr, orig, deformed, cols, rows, w, h, dst_band_ptr, first_row, band_rows):
        o = None if orig is None else np.ascontiguousarray(orig, np.float32)
        d = np.ascontiguousarray(deformed, np.float32)
        self._check(self._lib.pfx_warp_mesh_catmull_rom_band_dev(self._h, C.c_void_p(src_ptr), _p(o), _p(d), C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w),
                                                                 C.c_uint32(h), C.c_void_p(dst_band_ptr), C.c_uint32(first_row), C.c_uint32(band_rows)))

    def selftest_round_pack(self):
        """(mismatches, signalling-NaN mismatches) of the device's round-and-pack against the step-by-step formula over all 2^32 floats"""
        bad, snan = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.pfx_selftest_round_pack(self._h, C.byref(bad), C.byref(snan)))
        return bad.value, snan.value

    def selftest_unorm_store(self) -> int:
        bad = C.c_uint64(0)
        self._check(self._lib.pfx_selftest_unorm_store(self._h, C.byref(bad)))
        return int(bad.value)

    def selftest_division(self, seed: int, n_millions: int) -> int:
        bad = C.c_uint64(0)
        self._check(self._lib.pfx_selftest_division(self._h, C.c_uint64(seed), C.c_uint32(n_millions), C.byref(bad)))
        return int(bad.value)

    def tune(self, key: str, value: int):
        self._check(self._lib.pfx_tune(self._h, key.encode(), C.c_int(value)))

    def flatten_stats(self, reset: bool = False):
        """work counters of the compositor's dead-layer elimination (pfx_flatten_stats)"""
        out = (C.c_uint64 * 8)()
        self._check(self._lib.pfx_flatten_stats(self._h, out, C.c_int(int(reset))))
        keys = ("rounds", "round_px", "round_layers", "nat_units", "nat_layers", "alpha_reads", "queue_units", "reserved")
        return dict(zip(keys, [int(v) for v in out]))

    def flatten_trace(self, reset: bool = False):
        """per-phase wave clocks of the class-sorting compositor's diagnostic build (pfx_flatten_trace; tune("dle_stats", 4) first)"""
        out = (C.c_uint64 * 16)()
        self._check(self._lib.pfx_flatten_trace(self._h, out, C.c_int(int(reset))))
        v = [int(x) for x in out]
        keys = ("classify", "deal_early", "natural", "store", "early_wait", "early_blend", "natural_wait", "natural_blend")
        return {"wave_clocks": v[0], "waves": v[1], **dict(zip(keys, v[8:16]))}

    def timing_enable(self, on: bool):
        self._check(self._lib.pfx_timing_enable(self._h, C.c_int(int(on))))

    def timing_reset(self):
        self._check(self._lib.pfx_timing_reset(self._h))

    def timing_read(self, name: str):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self._lib.pfx_timing_read(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
