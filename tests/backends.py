"""Two interchangeable back-ends for the known-answer table in golden_cases.py.

OracleBackend  -> oracle/libpfx_oracle.so (CPU restatement; the checker)
GpuBackend     -> libpfx.so through its C-ABI (the product; HIP kernels on cuda:0)
"""
from __future__ import annotations

import numpy as np

from . import golden_cases as GC
from . import oracle_lib as O


class OracleBackend:
    name = "oracle"

    def composite(self, layers, w, h):
        return O.composite(layers, w, h)

    def gaussian_blur(self, img, sigma, mask=None):
        return O.gaussian_blur(img, sigma, mask)

    def box_blur(self, img, radius, mask=None):
        return O.box_blur(img, radius, mask)

    def median(self, img, radius, mask=None):
        return O.median(img, radius, mask)

    def pixelate(self, img, block, mask=None):
        return O.pixelate(img, block, mask)

    def sharpen(self, img, amount, radius, mask=None):
        return O.sharpen(img, amount, radius, mask)

    def glow(self, img, radius, intensity, mask=None):
        return O.glow(img, radius, intensity, mask)

    def bokeh_blur(self, img, radius, mask=None):
        return O.bokeh_blur(img, radius, mask)

    def motion_blur(self, img, angle_deg, distance, mask=None):
        return O.motion_blur(img, angle_deg, distance, mask)

    def effect(self, name, img, **kw):
        """the rest of the effect bank (golden_cases.effect_cases): same keyword names on both back-ends"""
        return getattr(O, name)(img, **kw)

    def resize(self, img, new_w, new_h, filter):
        return O.resize(img, new_w, new_h, filter)

    def flip_rotate(self, img, op):
        return O.flip_rotate(img, op)

    def resize_canvas(self, img, new_w, new_h, anchor, fill):
        return O.resize_canvas(img, new_w, new_h, anchor, fill)

    def affine_layer(self, img, composite=False, **kw):
        """affine_transform_layer on a one-layer document, then extract_layer (TiledImage round trip) or state.composite()"""
        h, w = img.shape[:2]
        layer = O.tiled_roundtrip(O.affine(img, w, h, **kw))
        return O.composite([dict(pixels=layer)], w, h) if composite else layer

    def rhai_adjust(self, img, op, params=()):
        return O.rhai_adjust(img, op, params)

    def adjust(self, img, op, params=(), lut=None, mask=None, sparse=0):
        return O.adjust(img, op, params, lut, mask, sparse)

    def auto_levels(self, img, mask=None):
        return O.adjust(img, "lut_rgba", lut=O.auto_levels_luts(img, mask), mask=mask, sparse=O.FROM_FLAT)

    def levels(self, img, in_black, in_white, gamma, out_black, out_white, mask=None):
        lv = O.levels_lut(in_black, in_white, gamma, out_black, out_white)
        luts = np.stack([lv, lv, lv, np.arange(256, dtype=np.uint8)])
        return O.adjust(img, "lut_rgba", lut=luts, mask=mask, sparse=O.FROM_FLAT)

    def warp_push(self, img, brushes):
        h, w = img.shape[:2]
        d = np.zeros((h, w, 2), np.float32)
        for (mode, cx, cy, dx, dy, radius, strength) in brushes:
            O.displacement_brush(d, mode, cx, cy, dx, dy, radius, strength)
        return O.warp_displacement(img, d)

    def warp_displacement(self, img, disp):
        return O.warp_displacement(img, disp)

    def warp_mesh(self, img, orig, deformed, cols, rows):
        return O.warp_mesh_catmull_rom(img, orig, deformed, cols, rows)

    def brush_stamps(self, target, brush, points, selection=None, dyn=None):
        t = GC.brush_target(target) if isinstance(target, str) else np.ascontiguousarray(target).copy()
        b = O.make_brush(**brush)
        for (x, y) in points:
            O.brush_stamp(t, b, x, y, selection, dyn)
        return t

    def brush_line(self, target, brush, p0, p1, selection=None):
        t = GC.brush_target(target) if isinstance(target, str) else np.ascontiguousarray(target).copy()
        return O.brush_line(t, O.make_brush(**brush), p0, p1, selection)


class GpuBackend:
    """Same method names, routed through libpfx.so's C ABI (HIP kernels).  No oracle involved."""
    name = "gpu"

    def __init__(self, device: int = 0):
        from paintfe_amd import GpuRenderer
        self.r = GpuRenderer(device)

    def composite(self, layers, w, h):
        self.r.clear_layers()
        info = []
        for i, L in enumerate(layers):
            kind = L.get("kind", 0)
            if kind == 0:
                self.r.ensure_layer_texture(i, L["pixels"], generation=1)
                if L.get("mask") is not None:
                    self.r.set_layer_mask(i, L["mask"])
            info.append((i, L.get("opacity", 1.0), L.get("visible", True), L.get("mode", 0), kind, L.get("adj", ())))
        return self.r.composite(w, h, info)

    def gaussian_blur(self, img, sigma, mask=None):
        return self.r.gaussian_blur_core(img, sigma, mask)

    def box_blur(self, img, radius, mask=None):
        return self.r.box_blur_core(img, radius, mask)

    def median(self, img, radius, mask=None):
        return self.r.median_core(img, radius, mask)

    def pixelate(self, img, block, mask=None):
        return self.r.pixelate_core(img, block, mask)

    # sharpen / glow / shadow run in the context's DEFAULT mode here: the library itself takes the bit-exact Gaussian inside composite effects
    # (pfx_effects.cpp: effect_gaussian; the reference holds them at tolerance 0, tests/visual_filters.rs:43-55,154-165)
    def sharpen(self, img, amount, radius, mask=None):
        return self.r.sharpen_core(img, amount, radius, mask)

    def glow(self, img, radius, intensity, mask=None):
        return self.r.glow_core(img, radius, intensity, mask)

    def bokeh_blur(self, img, radius, mask=None):
        return self.r.bokeh_blur_core(img, radius, mask)

    def motion_blur(self, img, angle_deg, distance, mask=None):
        return self.r.motion_blur_core(img, angle_deg, distance, mask)

    def resize(self, img, new_w, new_h, filter):
        return self.r.resize_image(img, new_w, new_h, filter)

    def flip_rotate(self, img, op):
        return self.r.flip_rotate(img, op)

    def resize_canvas(self, img, new_w, new_h, anchor, fill):
        return self.r.resize_canvas(img, new_w, new_h, anchor, fill)

    def affine_layer(self, img, composite=False, **kw):
        h, w = img.shape[:2]
        layer = self.r.tiled_roundtrip(self.r.affine_transform(img, w, h, **kw))
        return self.composite([dict(pixels=layer)], w, h) if composite else layer

    def effect(self, name, img, **kw):
        return getattr(self.r, name + "_core")(img, **kw)

    def rhai_adjust(self, img, op, params=()):
        return self.r.rhai_adjust(img, op, params)

    def adjust(self, img, op, params=(), lut=None, mask=None, sparse=0):
        return self.r.adjust(img, op, params, lut, mask, sparse)

    def auto_levels(self, img, mask=None):
        return self.r.auto_levels(img, mask)

    def levels(self, img, in_black, in_white, gamma, out_black, out_white, mask=None):
        return self.r.levels(img, in_black, in_white, gamma, out_black, out_white, mask)

    def warp_push(self, img, brushes):
        h, w = img.shape[:2]
        d = np.zeros((h, w, 2), np.float32)
        for (mode, cx, cy, dx, dy, radius, strength) in brushes:
            self.r.displacement_brush(d, mode, cx, cy, dx, dy, radius, strength)
        return self.r.warp_displacement(img, d)

    def warp_displacement(self, img, disp):
        return self.r.warp_displacement(img, disp)

    def warp_mesh(self, img, orig, deformed, cols, rows):
        return self.r.warp_mesh_catmull_rom(img, orig, deformed, cols, rows)

    def brush_stamps(self, target, brush, points, selection=None, dyn=None):
        t = GC.brush_target(target) if isinstance(target, str) else target
        return self.r.brush_stamps(t, self.r.make_brush(**brush), points, selection, dyn)

    def brush_line(self, target, brush, p0, p1, selection=None):
        t = GC.brush_target(target) if isinstance(target, str) else target
        return self.r.brush_line(t, self.r.make_brush(**brush), p0, p1, selection)
