"""ctypes front-end for the CPU oracle (oracle/libpfx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, ``__graft_entry__.smoke()`` and
``bench.py``'s cpu_baseline leg — never by the product package ``paintfe_amd``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ORACLE_DIR, "libpfx_oracle.so")

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"] + (["-B"] if force else []))
    return SO


class Layer(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("mask", C.c_void_p), ("opacity", C.c_float),
                ("blend_mode", C.c_uint8), ("visible", C.c_uint8), ("kind", C.c_uint8), ("_pad", C.c_uint8),
                ("adj", C.c_float * 16)]


class Brush(C.Structure):
    _fields_ = [("size", C.c_float), ("hardness", C.c_float), ("flow", C.c_float), ("color", C.c_float * 4),
                ("anti_aliased", C.c_int), ("is_eraser", C.c_int), ("mode", C.c_int)]


# op ids (keep in sync with oracle/pfx_oracle.h)
OPS = ["invert", "invert_alpha", "sepia", "brightness_contrast", "hsl", "exposure", "highlights_shadows",
       "temperature_tint", "threshold", "posterize", "color_balance", "gradient_map", "black_and_white",
       "vibrance", "lut_rgba", "desaturate"]
OP = {n: i for i, n in enumerate(OPS)}
RHAI_OPS = ["invert", "desaturate", "sepia", "sepia_strength", "brightness_contrast", "hsl", "exposure", "levels"]
RHAI = {n: i for i, n in enumerate(RHAI_OPS)}
DENSE, FROM_FLAT, IN_PLACE = 0, 1, 2
ADJ_EXPOSURE, ADJ_BC, ADJ_INVERT, ADJ_MIXER = 1, 2, 3, 4

BLEND_MODES = ["normal", "multiply", "screen", "additive", "reflect", "glow", "color_burn", "color_dodge",
               "overlay", "difference", "negation", "lighten", "darken", "xor", "overwrite", "hard_light",
               "soft_light", "exclusion", "subtract", "divide", "linear_burn", "vivid_light", "linear_light",
               "pin_light", "hard_mix"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.pfxo_brush_alpha.restype = C.c_float
        _lib.pfxo_gaussian_kernel.restype = C.c_int
        _lib.pfxo_brush_line_points.restype = C.c_int
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(C.c_void_p)


def _opt_u8(a):
    if a is None:
        return None, None
    return _u8(a)


def chunk_populated(img):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros(((h + 63) // 64, (w + 63) // 64), np.uint8)
    lib().pfxo_chunk_populated(ps, C.c_uint32(w), C.c_uint32(h), out.ctypes.data_as(C.c_void_p))
    return out


def tiled_roundtrip(img):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.empty_like(src)
    lib().pfxo_tiled_roundtrip(ps, C.c_uint32(w), C.c_uint32(h), out.ctypes.data_as(C.c_void_p))
    return out


def blend_pixel(base, top, mode, opacity):
    b, pb = _u8(np.asarray(base))
    t, pt = _u8(np.asarray(top))
    o = np.zeros(4, np.uint8)
    lib().pfxo_blend_pixel(pb, pt, C.c_int(mode), C.c_float(opacity), o.ctypes.data_as(C.c_void_p))
    return o


class Preview(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("chunk_present", C.c_void_p), ("active_layer", C.c_int32), ("blend_mode", C.c_uint8),
                ("is_eraser", C.c_uint8), ("replaces_layer", C.c_uint8), ("_pad", C.c_uint8)]


def composite(layers, w, h, threads=0, preview=None):
    """layers: list of dicts {pixels, mask, opacity, mode, visible, kind, adj}.
    preview: dict {pixels, active_layer, blend_mode, is_eraser, replaces_layer, chunk_present} or None."""
    arr = (Layer * len(layers))()
    keep = []
    for i, L in enumerate(layers):
        px = L.get("pixels")
        if px is not None:
            a, p = _u8(px)
            keep.append(a)
            arr[i].pixels = p.value
        m = L.get("mask")
        if m is not None:
            a, p = _u8(m)
            keep.append(a)
            arr[i].mask = p.value
        arr[i].opacity = L.get("opacity", 1.0)
        arr[i].blend_mode = L.get("mode", 0)
        arr[i].visible = 1 if L.get("visible", True) else 0
        arr[i].kind = L.get("kind", 0)
        for j, v in enumerate(L.get("adj", [])):
            arr[i].adj[j] = v
    out = np.zeros((h, w, 4), np.uint8)
    if preview is not None:
        ppx, pp = _u8(preview["pixels"])
        pcp, pc = _opt_u8(preview.get("chunk_present"))
        pv = Preview(pp.value, pc.value if pc is not None else None, preview["active_layer"], preview.get("blend_mode", 0),
                     int(preview.get("is_eraser", False)), int(preview.get("replaces_layer", False)), 0)
        lib().pfxo_composite_preview(arr, C.c_int(len(layers)), C.c_uint32(w), C.c_uint32(h), C.byref(pv),
                                     out.ctypes.data_as(C.c_void_p), C.c_int(threads))
        return out
    lib().pfxo_composite(arr, C.c_int(len(layers)), C.c_uint32(w), C.c_uint32(h),
                         out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def set_serial_writeback(on: bool):
    """the reference's single-threaded compositor write-back (canvas_state.rs:686-695) for the faithful CPU baseline"""
    lib().pfxo_set_serial_writeback(C.c_int(1 if on else 0))


def flatten_stack(stack, modes, opacities, threads=0):
    n, h, w, _ = stack.shape
    s, ps = _u8(stack)
    m, pm = _u8(np.asarray(modes))
    o, po = _f32(np.asarray(opacities))
    out = np.zeros((h, w, 4), np.uint8)
    lib().pfxo_flatten_stack(ps, C.c_int(n), pm, po, C.c_uint32(w), C.c_uint32(h),
                             out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def gaussian_kernel(sigma):
    cap = 2 * int(np.ceil(np.float32(sigma) * np.float32(3.0))) + 3
    out = np.zeros(cap, np.float32)
    n = lib().pfxo_gaussian_kernel(C.c_float(sigma), out.ctypes.data_as(C.c_void_p), C.c_int(cap))
    return out[:n].copy()


def _img_op(fn, img, *args, mask=None, threads=0, with_mask=True):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros_like(src)
    m, pm = _opt_u8(mask)
    a = [ps, C.c_uint32(w), C.c_uint32(h), *args]
    if with_mask:
        a.append(pm)
    a += [out.ctypes.data_as(C.c_void_p), C.c_int(threads)]
    fn(*a)
    return out


def gaussian_blur(img, sigma, mask=None, threads=0):
    if mask is None:
        return _img_op(lib().pfxo_gaussian_blur, img, C.c_float(sigma), with_mask=False, threads=threads)
    return _img_op(lib().pfxo_blur_with_selection, img, C.c_float(sigma), mask=mask, threads=threads)


def box_blur(img, radius, mask=None, threads=0):
    return _img_op(lib().pfxo_box_blur, img, C.c_float(radius), mask=mask, threads=threads)


def median(img, radius, mask=None, threads=0):
    return _img_op(lib().pfxo_median, img, C.c_uint32(radius), mask=mask, threads=threads)


def pixelate(img, block, mask=None, threads=0):
    return _img_op(lib().pfxo_pixelate, img, C.c_uint32(block), mask=mask, threads=threads)


def glow(img, radius, intensity, mask=None, threads=0):
    return _img_op(lib().pfxo_glow, img, C.c_float(radius), C.c_float(intensity), mask=mask, threads=threads)


def sharpen(img, amount, radius, mask=None, threads=0):
    return _img_op(lib().pfxo_sharpen, img, C.c_float(amount), C.c_float(radius), mask=mask, threads=threads)


def bokeh_blur(img, radius, mask=None, threads=0):
    return _img_op(lib().pfxo_bokeh_blur, img, C.c_float(radius), mask=mask, threads=threads)


def motion_blur(img, angle_deg, distance, mask=None, threads=0):
    return _img_op(lib().pfxo_motion_blur, img, C.c_float(angle_deg), C.c_float(distance), mask=mask, threads=threads)


def _c4(color):
    return (C.c_uint8 * 4)(*[int(v) for v in color])


NOISE_TYPES = {"uniform": 0, "gaussian": 1, "perlin": 2}
HALFTONE_SHAPES = {"circle": 0, "square": 1, "diamond": 2, "line": 3}
GRID_STYLES = {"lines": 0, "checkerboard": 1}
OUTLINE_MODES = {"outside": 0, "inside": 1, "center": 2}
COLOR_FILTER_MODES = {"multiply": 0, "screen": 1, "overlay": 2, "soft_light": 3}


def zoom_blur(img, center_x, center_y, strength, samples, tint_color=(0.0, 0.0, 0.0, 0.0), tint_strength=0.0, mask=None, threads=0):
    tint = (C.c_float * 4)(*tint_color)
    return _img_op(lib().pfxo_zoom_blur, img, C.c_float(center_x), C.c_float(center_y), C.c_float(strength), C.c_uint32(samples),
                   tint, C.c_float(tint_strength), mask=mask, threads=threads)


def crystallize(img, cell_size, seed, mask=None, threads=0):
    return _img_op(lib().pfxo_crystallize, img, C.c_float(cell_size), C.c_uint32(seed), mask=mask, threads=threads)


def dents(img, scale, amount, seed, octaves, roughness, pinch=False, wrap=False, mask=None, threads=0):
    return _img_op(lib().pfxo_dents, img, C.c_float(scale), C.c_float(amount), C.c_uint32(seed), C.c_uint32(octaves),
                   C.c_float(roughness), C.c_int(int(pinch)), C.c_int(int(wrap)), mask=mask, threads=threads)


def bulge(img, amount, origin=(0.5, 0.5), mask=None, threads=0):
    return _img_op(lib().pfxo_bulge, img, C.c_float(amount), C.c_float(origin[0]), C.c_float(origin[1]), mask=mask, threads=threads)


def twist(img, angle_deg, origin=(0.5, 0.5), mask=None, threads=0):
    return _img_op(lib().pfxo_twist, img, C.c_float(angle_deg), C.c_float(origin[0]), C.c_float(origin[1]), mask=mask, threads=threads)


def add_noise(img, amount, noise_type, monochrome, seed, scale, octaves, mask=None, threads=0):
    return _img_op(lib().pfxo_add_noise, img, C.c_float(amount), C.c_int(NOISE_TYPES[noise_type]), C.c_int(int(monochrome)),
                   C.c_uint32(seed), C.c_float(scale), C.c_uint32(octaves), mask=mask, threads=threads)


def reduce_noise(img, strength, radius, mask=None, threads=0):
    return _img_op(lib().pfxo_reduce_noise, img, C.c_float(strength), C.c_uint32(radius), mask=mask, threads=threads)


def vignette(img, amount, softness, mask=None, threads=0):
    return _img_op(lib().pfxo_vignette, img, C.c_float(amount), C.c_float(softness), mask=mask, threads=threads)


def halftone(img, dot_size, angle_deg, shape="circle", mask=None, threads=0):
    return _img_op(lib().pfxo_halftone, img, C.c_float(dot_size), C.c_float(angle_deg), C.c_int(HALFTONE_SHAPES[shape]), mask=mask, threads=threads)


def grid(img, cell_w, cell_h, line_width, color, style="lines", opacity=1.0, mask=None, threads=0):
    return _img_op(lib().pfxo_grid, img, C.c_uint32(cell_w), C.c_uint32(cell_h), C.c_uint32(line_width), _c4(color),
                   C.c_int(GRID_STYLES[style]), C.c_float(opacity), mask=mask, threads=threads)


def canvas_border(img, width, color, mask=None, threads=0):
    return _img_op(lib().pfxo_canvas_border, img, C.c_uint32(width), _c4(color), mask=mask, threads=threads)


def drop_shadow(img, offset_x, offset_y, blur_radius, widen_radius, color, opacity, mask=None, threads=0):
    return _img_op(lib().pfxo_drop_shadow, img, C.c_int32(offset_x), C.c_int32(offset_y), C.c_float(blur_radius), C.c_int(int(widen_radius)),
                   _c4(color), C.c_float(opacity), mask=mask, threads=threads)


shadow = drop_shadow  # the reference calls it shadow_core (render.rs:220)


def outline(img, width, color, mode="outside", anti_alias=True, mask=None, threads=0):
    return _img_op(lib().pfxo_outline, img, C.c_uint32(width), _c4(color), C.c_int(OUTLINE_MODES[mode]), C.c_int(int(anti_alias)),
                   mask=mask, threads=threads)


def pixel_drag(img, seed, amount, distance, direction, mask=None, threads=0):
    return _img_op(lib().pfxo_pixel_drag, img, C.c_uint32(seed), C.c_float(amount), C.c_uint32(distance), C.c_float(direction),
                   mask=mask, threads=threads)


def rgb_displace(img, r_off, g_off, b_off, mask=None, threads=0):
    off = (C.c_int32 * 6)(r_off[0], r_off[1], g_off[0], g_off[1], b_off[0], b_off[1])
    return _img_op(lib().pfxo_rgb_displace, img, off, mask=mask, threads=threads)


def ink(img, edge_strength, threshold, mask=None, threads=0):
    return _img_op(lib().pfxo_ink, img, C.c_float(edge_strength), C.c_float(threshold), mask=mask, threads=threads)


def oil_painting(img, radius, levels, mask=None, threads=0):
    return _img_op(lib().pfxo_oil_painting, img, C.c_uint32(radius), C.c_uint32(levels), mask=mask, threads=threads)


def color_filter(img, filter_color, intensity, mode="multiply", mask=None, threads=0):
    return _img_op(lib().pfxo_color_filter, img, _c4(filter_color), C.c_float(intensity), C.c_int(COLOR_FILTER_MODES[mode]),
                   mask=mask, threads=threads)


def contours(img, scale, frequency, line_width, line_color, seed, octaves, blend, mask=None, threads=0):
    return _img_op(lib().pfxo_contours, img, C.c_float(scale), C.c_float(frequency), C.c_float(line_width), _c4(line_color),
                   C.c_uint32(seed), C.c_uint32(octaves), C.c_float(blend), mask=mask, threads=threads)


def affine(img, canvas_w, canvas_h, rotation_z=0.0, rotation_x=0.0, rotation_y=0.0, scale=1.0, offset=(0.0, 0.0), interpolation="bilinear", threads=0):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros((canvas_h, canvas_w, 4), np.uint8)
    lib().pfxo_affine(ps, C.c_uint32(w), C.c_uint32(h), C.c_uint32(canvas_w), C.c_uint32(canvas_h), C.c_float(rotation_z), C.c_float(rotation_x),
                      C.c_float(rotation_y), C.c_float(scale), C.c_float(offset[0]), C.c_float(offset[1]), C.c_int(0 if interpolation == "nearest" else 1),
                      out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


CANVAS_OPS = {"flip_horizontal": 0, "flip_vertical": 1, "rotate_90cw": 2, "rotate_90ccw": 3, "rotate_180": 4}


def flip_rotate(img, op):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    k = CANVAS_OPS[op]
    out = np.zeros((w, h, 4) if k in (2, 3) else (h, w, 4), np.uint8)
    lib().pfxo_flip_rotate(ps, C.c_uint32(w), C.c_uint32(h), C.c_int(k), out.ctypes.data_as(C.c_void_p))
    return out


def resize_canvas(img, new_w, new_h, anchor=(0, 0), fill=(0, 0, 0, 0)):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros((new_h, new_w, 4), np.uint8)
    lib().pfxo_resize_canvas(ps, C.c_uint32(w), C.c_uint32(h), C.c_uint32(new_w), C.c_uint32(new_h), C.c_uint32(anchor[0]), C.c_uint32(anchor[1]),
                             _c4(fill), out.ctypes.data_as(C.c_void_p))
    return out


RESIZE_FILTERS = {"nearest": 0, "bilinear": 1, "bicubic": 2, "lanczos3": 3}


def resize(img, nw, nh, filter="bilinear", threads=0):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros((nh, nw, 4), np.uint8)
    lib().pfxo_resize(ps, C.c_uint32(w), C.c_uint32(h), C.c_uint32(nw), C.c_uint32(nh), C.c_int(RESIZE_FILTERS[filter]),
                      out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def adjust(img, op, params=(), lut=None, mask=None, sparse=DENSE, threads=0):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    out = np.zeros_like(src)
    p, pp = _f32(np.asarray(list(params) + [0.0] * (16 - len(params)), np.float32))
    l, pl = _opt_u8(lut)
    m, pm = _opt_u8(mask)
    lib().pfxo_adjust(ps, C.c_uint32(w), C.c_uint32(h), C.c_int(OP[op] if isinstance(op, str) else op), pp, pl, pm,
                      C.c_int(sparse), out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def rhai_adjust(img, op, params=()):
    out = np.ascontiguousarray(img, dtype=np.uint8).copy()
    p, pp = _f32(np.asarray(list(params) + [0.0] * (8 - len(params)), np.float32))
    lib().pfxo_rhai_adjust(out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size // 4),
                           C.c_int(RHAI[op] if isinstance(op, str) else op), pp)
    return out


def levels_lut(in_black, in_white, gamma, out_black, out_white):
    lut = np.zeros(256, np.uint8)
    lib().pfxo_levels_lut(C.c_float(in_black), C.c_float(in_white), C.c_float(gamma), C.c_float(out_black),
                          C.c_float(out_white), lut.ctypes.data_as(C.c_void_p))
    return lut


def rhai_levels_lut(in_black, in_white, gamma):
    lut = np.zeros(256, np.uint8)
    lib().pfxo_rhai_levels_lut(C.c_float(in_black), C.c_float(in_white), C.c_float(gamma),
                               lut.ctypes.data_as(C.c_void_p))
    return lut


def curves_lut(points):
    pts, pp = _f32(np.asarray(points, np.float32).reshape(-1, 2))
    lut = np.zeros(256, np.uint8)
    lib().pfxo_curves_lut(pp, C.c_int(len(pts)), lut.ctypes.data_as(C.c_void_p))
    return lut


def auto_levels_luts(img, mask=None):
    h, w = img.shape[:2]
    src, ps = _u8(img)
    m, pm = _opt_u8(mask)
    out = np.zeros((4, 256), np.uint8)
    lib().pfxo_auto_levels_luts(ps, C.c_uint32(w), C.c_uint32(h), pm, out.ctypes.data_as(C.c_void_p))
    return out


def identity_luts():
    return np.tile(np.arange(256, dtype=np.uint8), (4, 1))


def catmull_rom_weights(t):
    w = np.zeros(4, np.float32)
    lib().pfxo_catmull_rom_weights(C.c_float(t), w.ctypes.data_as(C.c_void_p))
    return w


def mesh_displacement(orig, deformed, cols, rows, w, h, threads=0):
    o, po = _f32(orig)
    d, pd = _f32(deformed)
    out = np.zeros((h, w, 2), np.float32)
    lib().pfxo_mesh_displacement(po, pd, C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w), C.c_uint32(h),
                                 out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def mesh_displacement_fast(deformed, cols, rows, w, h, threads=0):
    d, pd = _f32(deformed)
    out = np.zeros((h, w, 2), np.float32)
    lib().pfxo_mesh_displacement_fast(pd, C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w), C.c_uint32(h),
                                      out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def warp_displacement(img, disp, threads=0):
    h, w = disp.shape[:2]
    sh, sw = img.shape[:2]
    s, ps = _u8(img)
    d, pd = _f32(disp)
    out = np.zeros((h, w, 4), np.uint8)
    lib().pfxo_warp_displacement_ex(ps, C.c_uint32(sw), C.c_uint32(sh), pd, C.c_uint32(w), C.c_uint32(h),
                                    out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def warp_mesh_catmull_rom(img, orig, deformed, cols, rows, threads=0):
    h, w = img.shape[:2]
    s, ps = _u8(img)
    o, po = _f32(orig)
    d, pd = _f32(deformed)
    out = np.zeros((h, w, 4), np.uint8)
    lib().pfxo_warp_mesh_catmull_rom(ps, po, pd, C.c_uint32(cols), C.c_uint32(rows), C.c_uint32(w), C.c_uint32(h),
                                     out.ctypes.data_as(C.c_void_p), C.c_int(threads))
    return out


def displacement_brush(disp, mode, cx, cy, dx, dy, radius, strength):
    h, w = disp.shape[:2]
    assert disp.dtype == np.float32 and disp.flags.c_contiguous
    lib().pfxo_displacement_brush(disp.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.c_int(mode),
                                  C.c_float(cx), C.c_float(cy), C.c_float(dx), C.c_float(dy), C.c_float(radius),
                                  C.c_float(strength))
    return disp


def make_brush(size, hardness, anti_aliased, color=(0, 0, 0, 1), flow=1.0, is_eraser=False, mode=0):
    b = Brush()
    b.size, b.hardness, b.flow = size, hardness, flow
    for i in range(4):
        b.color[i] = color[i]
    b.anti_aliased, b.is_eraser, b.mode = int(anti_aliased), int(is_eraser), mode
    return b


class BrushDyn(C.Structure):
    _fields_ = [("scatter", C.c_float), ("hue_jitter", C.c_float), ("brightness_jitter", C.c_float), ("stamp_counter", C.c_uint32),
                ("tip_mask", C.c_void_p), ("tip_mask_size", C.c_uint32), ("tip_rotation", C.c_float), ("tip_random_rotation", C.c_int32),
                ("tip_rotation_lo", C.c_float), ("tip_rotation_hi", C.c_float)]


def make_dyn(scatter=0.0, hue_jitter=0.0, brightness_jitter=0.0, stamp_counter=0, tip_mask=None, tip_rotation=0.0, tip_random_rotation=False,
             tip_rotation_range=(0.0, 360.0)):
    """returns (BrushDyn, keep-alive) — tip_mask: square uint8 array already rescaled to the brush size"""
    d = BrushDyn(scatter, hue_jitter, brightness_jitter, stamp_counter, None, 0, tip_rotation, int(tip_random_rotation),
                 tip_rotation_range[0], tip_rotation_range[1])
    keep = None
    if tip_mask is not None:
        keep, p = _u8(tip_mask)
        d.tip_mask = p.value
        d.tip_mask_size = keep.shape[0]
    return d, keep


def brush_tip_rescale(src_mask, brush_size, hardness):
    src, ps = _u8(src_mask)
    n = max(int(np.ceil(np.float32(brush_size))), 1)
    out = np.zeros((n, n), np.uint8)
    lib().pfxo_brush_tip_rescale.restype = C.c_uint32
    got = lib().pfxo_brush_tip_rescale(ps, C.c_uint32(src.shape[0]), C.c_float(brush_size), C.c_float(hardness), out.ctypes.data_as(C.c_void_p))
    assert got == n
    return out


def brush_stamp(target, brush, cx, cy, selection=None, dyn=None):
    h, w = target.shape[:2]
    assert target.dtype == np.uint8 and target.flags.c_contiguous
    m, pm = _opt_u8(selection)
    if dyn is not None:
        d, keep = make_dyn(**dyn)
        lib().pfxo_brush_stamp_ex(target.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.byref(brush), C.byref(d),
                                  C.c_float(cx), C.c_float(cy), pm)
        return target
    lib().pfxo_brush_stamp(target.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.byref(brush),
                           C.c_float(cx), C.c_float(cy), pm)
    return target


def brush_line(target, brush, p0, p1, selection=None):
    h, w = target.shape[:2]
    assert target.dtype == np.uint8 and target.flags.c_contiguous
    m, pm = _opt_u8(selection)
    lib().pfxo_brush_line(target.ctypes.data_as(C.c_void_p), C.c_uint32(w), C.c_uint32(h), C.byref(brush),
                          C.c_float(p0[0]), C.c_float(p0[1]), C.c_float(p1[0]), C.c_float(p1[1]), pm)
    return target


def brush_line_points(p0, p1, w, h):
    n = lib().pfxo_brush_line_points(C.c_float(p0[0]), C.c_float(p0[1]), C.c_float(p1[0]), C.c_float(p1[1]),
                                     C.c_uint32(w), C.c_uint32(h), None, C.c_int(0))
    out = np.zeros((max(n, 0), 2), np.float32)
    if n > 0:
        lib().pfxo_brush_line_points(C.c_float(p0[0]), C.c_float(p0[1]), C.c_float(p1[0]), C.c_float(p1[1]),
                                     C.c_uint32(w), C.c_uint32(h), out.ctypes.data_as(C.c_void_p), C.c_int(n))
    return out


def brush_commit(layer, preview, mode, selection=None):
    h, w = layer.shape[:2]
    out = np.ascontiguousarray(layer, np.uint8).copy()
    p, pp = _u8(preview)
    m, pm = _opt_u8(selection)
    lib().pfxo_brush_commit(out.ctypes.data_as(C.c_void_p), pp, C.c_uint32(w), C.c_uint32(h), C.c_int(mode), pm)
    return out


def eraser_commit(layer, preview, selection=None):
    h, w = layer.shape[:2]
    out = np.ascontiguousarray(layer, np.uint8).copy()
    p, pp = _u8(preview)
    m, pm = _opt_u8(selection)
    lib().pfxo_eraser_commit(out.ctypes.data_as(C.c_void_p), pp, C.c_uint32(w), C.c_uint32(h), pm)
    return out
