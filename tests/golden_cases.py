"""Known-answer cases: (golden key) -> how to reproduce it.

Each case restates the *call* the reference's own test makes (file:line cited) on the
reference's deterministic input generator, so the same table drives
  * tests/test_oracle_golden.py  (oracle vs golden, CPU, pins the oracle), and
  * tests/test_gpu_golden.py     (HIP path through the C-ABI vs golden, GPU).

A case is ``(key, backend_method, kwargs)`` where ``backend_method`` names a method that both
back-ends implement (tests/backends.py).
"""
from __future__ import annotations

import numpy as np

from . import inputs as I

f32 = np.float32
W = H = 64

BLEND_MODES = ["normal", "multiply", "screen", "additive", "reflect", "glow", "color_burn", "color_dodge",
               "overlay", "difference", "negation", "lighten", "darken", "xor", "overwrite", "hard_light",
               "soft_light", "exclusion", "subtract", "divide", "linear_burn", "vivid_light", "linear_light",
               "pin_light", "hard_mix"]


def _grad():
    return I.create_test_gradient(64, 64)


def _gradient_map_lut():
    i = np.arange(256, dtype=f32)
    t = i / f32(255.0)
    lut = np.zeros((256, 4), np.uint8)
    lut[:, 0] = (t * f32(255.0)).astype(np.uint8)
    lut[:, 1] = (t * t * f32(200.0)).astype(np.uint8)
    lut[:, 2] = (t * t * t * f32(150.0)).astype(np.uint8)
    lut[:, 3] = 255
    return lut


def _swirl_field():
    """tests/transform_ops.rs:345-358"""
    x = np.arange(32, dtype=f32)[None, :]
    y = np.arange(32, dtype=f32)[:, None]
    dx = x - f32(16.0)
    dy = y - f32(16.0)
    r = np.maximum(np.sqrt(dx * dx + dy * dy), f32(0.001))
    s = np.maximum(f32(1.0) - r / f32(16.0), f32(0.0))
    d = np.zeros((32, 32, 2), f32)
    d[..., 0] = -dy * s * f32(0.5)
    d[..., 1] = dx * s * f32(0.5)
    return d


BLACK = (0.0, 0.0, 0.0, 1.0)
WHITE = (1.0, 1.0, 1.0, 1.0)
RED = (1.0, 0.0, 0.0, 1.0)
BLUE_SEMI = (0.0, 0.0, 1.0, 0.5)


def blend_cases():
    """tests/visual_blend.rs:19-106"""
    bg = I.create_test_checkerboard(64, 64)
    fg = I.blend_foreground()
    out = []
    for m, name in enumerate(BLEND_MODES):
        out.append((f"blend/{name}", "composite",
                    dict(layers=[dict(pixels=bg), dict(pixels=fg, mode=m)], w=64, h=64)))
    out.append(("blend/normal_half_opacity", "composite",
                dict(layers=[dict(pixels=bg), dict(pixels=_grad(), opacity=0.5)], w=64, h=64)))
    return out


def filter_cases():
    """tests/visual_filters.rs:30-62,90-94,143-147 and tests/scripting.rs:119-152"""
    t = _grad()
    return [
        ("filters/gaussian_blur_s2", "gaussian_blur", dict(img=t, sigma=2.0)),
        ("filters/gaussian_blur_s5", "gaussian_blur", dict(img=t, sigma=5.0)),
        ("scripting/apply_blur", "gaussian_blur", dict(img=t, sigma=2.0)),
        ("filters/box_blur_r3", "box_blur", dict(img=t, radius=3.0)),
        ("filters/median_r2", "median", dict(img=t, radius=2)),
        ("filters/pixelate_8", "pixelate", dict(img=t, block=8)),
        ("scripting/apply_pixelate", "pixelate", dict(img=t, block=4)),
        # effects built from the same kernels (tests/visual_filters.rs:43-55,154-165)
        ("filters/sharpen_a1_r1", "sharpen", dict(img=t, amount=1.0, radius=1.0)),
        ("filters/glow_r3_i05", "glow", dict(img=t, radius=3.0, intensity=0.5)),
        ("filters/bokeh_blur_r5", "bokeh_blur", dict(img=t, radius=5.0)),
        ("filters/motion_blur_45_10", "motion_blur", dict(img=t, angle_deg=45.0, distance=10.0)),
    ]


def rhai_cases():
    """tests/scripting.rs:125-146 (Rhai-inline flavour, src/ops/scripting.rs:869-965)"""
    t = _grad()
    return [
        ("scripting/apply_invert", "rhai_adjust", dict(img=t, op="invert")),
        ("scripting/for_each_pixel_invert", "rhai_adjust", dict(img=t, op="invert")),
        ("scripting/map_channels_invert", "rhai_adjust", dict(img=t, op="invert")),
        ("scripting/apply_sepia", "rhai_adjust", dict(img=t, op="sepia")),
        ("scripting/apply_desaturate", "rhai_adjust", dict(img=t, op="desaturate")),
        ("scripting/apply_brightness_contrast", "rhai_adjust",
         dict(img=t, op="brightness_contrast", params=[20.0, 10.0])),
    ]


def adjustment_cases():
    """tests/visual_adjustments.rs:50-333.  sparse: 2 = in-place (apply_pixel_transform), 1 = *_from_flat."""
    t = _grad()
    A = "adjustments/"
    return [
        (A + "invert_colors", "adjust", dict(img=t, op="invert", sparse=2)),
        (A + "invert_alpha", "adjust", dict(img=t, op="invert_alpha", sparse=1)),
        (A + "invert_alpha_double", "adjust", dict(img=t, op="invert_alpha", sparse=1)),
        (A + "sepia", "adjust", dict(img=t, op="sepia", sparse=2)),
        (A + "auto_levels", "auto_levels", dict(img=t)),
        (A + "desaturate", "adjust", dict(img=t, op="desaturate", sparse=1)),
        (A + "brightness_30_contrast_20", "adjust", dict(img=t, op="brightness_contrast", params=[30.0, 20.0], sparse=1)),
        (A + "hsl_h30_s-20_l10", "adjust", dict(img=t, op="hsl", params=[30.0, -20.0, 10.0], sparse=1)),
        (A + "exposure_1ev", "adjust", dict(img=t, op="exposure", params=[1.0], sparse=1)),
        (A + "highlights_shadows", "adjust", dict(img=t, op="highlights_shadows", params=[30.0, -20.0], sparse=1)),
        (A + "levels", "levels", dict(img=t, in_black=20.0, in_white=235.0, gamma=1.2, out_black=0.0, out_white=255.0)),
        (A + "temperature_tint", "adjust", dict(img=t, op="temperature_tint", params=[30.0, 10.0], sparse=1)),
        (A + "threshold_128", "adjust", dict(img=t, op="threshold", params=[128.0], sparse=1)),
        (A + "posterize_4", "adjust", dict(img=t, op="posterize", params=[4.0], sparse=1)),
        (A + "color_balance", "adjust",
         dict(img=t, op="color_balance", params=[10.0, 0.0, -10.0, 0.0, 0.0, 0.0, -10.0, 0.0, 10.0], sparse=1)),
        (A + "gradient_map", "adjust", dict(img=t, op="gradient_map", lut=_gradient_map_lut(), sparse=1)),
        (A + "black_and_white", "adjust",
         dict(img=I.create_color_bands(64, 64), op="black_and_white", params=[0.3, 0.59, 0.11], sparse=1)),
        (A + "vibrance_50", "adjust", dict(img=t, op="vibrance", params=[50.0], sparse=1)),
    ]


def warp_cases():
    """tests/transform_ops.rs:162-169,201-209,345-360"""
    g = I.gradient_32()
    orig = I.uniform_grid(2, 2, 32.0, 32.0)
    deformed = orig.copy()
    deformed[4] = [20.0, 20.0]
    return [
        ("transform/displacement_radial_push", "warp_push",
         dict(img=g, brushes=[(0, 16.0, 16.0, 3.0, 0.0, 10.0, 0.8)])),
        ("transform/displacement_swirl", "warp_displacement", dict(img=g, disp=_swirl_field())),
        ("transform/mesh_warp_deformed", "warp_mesh", dict(img=g, orig=orig, deformed=deformed, cols=2, rows=2)),
    ]


def _stamp(key, size, hard, aa, pos=(32.0, 32.0), color=BLACK, target="blank", eraser=False, mode=0, selection=None):
    return (f"tools/{key}", "brush_stamps",
            dict(target=target, brush=dict(size=size, hardness=hard, anti_aliased=aa, color=color,
                                           is_eraser=eraser, mode=mode),
                 points=[pos], selection=selection))


def _line(key, size, hard, aa, p0, p1, color=BLACK, target="blank", eraser=False):
    return (f"tools/{key}", "brush_line",
            dict(target=target, brush=dict(size=size, hardness=hard, anti_aliased=aa, color=color,
                                           is_eraser=eraser, mode=0), p0=p0, p1=p1))


def brush_cases():
    """tests/tool_strokes.rs:65-570"""
    left_half = np.zeros((64, 64), np.uint8)
    left_half[:, :32] = 255
    multi = ("tools/stroke_multiple_stamps", "brush_stamps",
             dict(target="blank", brush=dict(size=10.0, hardness=0.8, anti_aliased=True, color=BLACK,
                                             is_eraser=False, mode=0),
                  points=[(float(f32(8.0) + f32(i) * f32(7.0)), 32.0) for i in range(8)], selection=None))
    return [
        _stamp("brush_circle_center", 20.0, 1.0, True),
        _stamp("brush_circle_soft", 30.0, 0.0, True),
        _stamp("brush_circle_hard", 20.0, 1.0, False),
        _stamp("brush_circle_tiny", 3.0, 1.0, True, color=RED),
        _stamp("brush_circle_large", 60.0, 0.5, True),
        _stamp("brush_semi_transparent", 20.0, 1.0, True, color=BLUE_SEMI),
        _stamp("brush_secondary_color", 20.0, 1.0, True, color=RED),
        _stamp("eraser_circle", 20.0, 1.0, True, target="white", eraser=True),
        _stamp("eraser_soft", 30.0, 0.0, True, target="white", eraser=True),
        _line("line_horizontal", 8.0, 1.0, True, (4.0, 32.0), (60.0, 32.0)),
        _line("line_vertical", 8.0, 1.0, True, (32.0, 4.0), (32.0, 60.0)),
        _line("line_diagonal", 6.0, 0.8, True, (4.0, 4.0), (60.0, 60.0)),
        _line("line_soft_thick", 16.0, 0.3, True, (10.0, 50.0), (54.0, 10.0), color=RED),
        _line("line_eraser", 10.0, 1.0, True, (4.0, 32.0), (60.0, 32.0), target="white", eraser=True),
        _stamp("brush_with_selection_mask", 40.0, 1.0, True, selection=left_half),
        multi,
        _stamp("brush_at_origin", 10.0, 1.0, True, pos=(0.0, 0.0)),
        _stamp("brush_at_corner", 20.0, 1.0, True, pos=(63.0, 63.0)),
        _line("line_zero_length", 12.0, 1.0, True, (32.0, 32.0), (32.0, 32.0)),
        _stamp("brush_dodge_mode", 24.0, 1.0, True, target="gradient", mode=1),
        _stamp("brush_burn_mode", 24.0, 1.0, True, target="gradient", mode=2),
        _stamp("pencil_circle", 12.0, 1.0, False),
        _line("pencil_line", 4.0, 1.0, False, (4.0, 4.0), (60.0, 60.0), color=RED),
    ]


def brush_target(kind: str) -> np.ndarray:
    if kind == "blank":
        return I.create_transparent(W, H)
    if kind == "white":
        return I.create_solid(W, H, (255, 255, 255, 255))
    if kind == "gradient":
        return I.create_test_gradient(W, H)
    raise ValueError(kind)


def _square(color):
    img = np.zeros((64, 64, 4), np.uint8)
    img[16:48, 16:48] = color
    return img


def effect_cases():
    """tests/visual_filters.rs:64-284: the rest of the effect bank, one golden each"""
    t = _grad()
    E = lambda key, name, img=t, **kw: (f"filters/{key}", "effect", dict(name=name, img=img, **kw))
    return [
        E("zoom_blur", "zoom_blur", center_x=0.5, center_y=0.5, strength=0.3, samples=8, tint_color=(0.0, 0.0, 0.0, 0.0), tint_strength=0.0),
        E("crystallize_s16", "crystallize", cell_size=16.0, seed=42),
        E("dents", "dents", scale=20.0, amount=10.0, seed=42, octaves=2, roughness=0.5, pinch=False, wrap=False),
        E("bulge_05", "bulge", amount=0.5),
        E("twist_45", "twist", angle_deg=45.0),
        E("add_noise_uniform", "add_noise", amount=30.0, noise_type="uniform", monochrome=False, seed=42, scale=1.0, octaves=1),
        E("add_noise_gaussian_mono", "add_noise", amount=30.0, noise_type="gaussian", monochrome=True, seed=42, scale=1.0, octaves=1),
        E("add_noise_perlin", "add_noise", amount=50.0, noise_type="perlin", monochrome=False, seed=42, scale=5.0, octaves=3),
        E("reduce_noise", "reduce_noise", strength=0.5, radius=2),
        E("vignette_08_05", "vignette", amount=0.8, softness=0.5),
        E("halftone_circle", "halftone", dot_size=4.0, angle_deg=45.0, shape="circle"),
        E("grid_lines_16", "grid", cell_w=16, cell_h=16, line_width=1, color=(0, 0, 0, 255), style="lines", opacity=1.0),
        E("drop_shadow", "shadow", img=_square((255, 255, 255, 255)), offset_x=5, offset_y=5, blur_radius=3.0, widen_radius=False,
          color=(0, 0, 0, 255), opacity=0.8),
        E("outline_outside", "outline", img=_square((255, 0, 0, 255)), width=2, color=(0, 0, 255, 255), mode="outside", anti_alias=True),
        E("contours", "contours", scale=10.0, frequency=5.0, line_width=1.0, line_color=(0, 0, 0, 255), seed=42, octaves=2, blend=0.5),
        E("pixel_drag", "pixel_drag", seed=42, amount=50.0, distance=20, direction=0.0),
        E("rgb_displace", "rgb_displace", r_off=(5, 0), g_off=(0, 0), b_off=(-5, 0)),
        E("ink", "ink", edge_strength=1.0, threshold=0.5),
        E("oil_painting", "oil_painting", radius=3, levels=20),
        E("color_filter_multiply", "color_filter", filter_color=(255, 128, 0, 255), intensity=0.5, mode="multiply"),
    ]


def resize_cases():
    """tests/visual_transforms.rs:134-164: imageops::resize on the asymmetric 64x48 gradient"""
    t = I.create_test_gradient(64, 48)
    return [
        ("transforms/resize_2x_nearest", "resize", dict(img=t, new_w=128, new_h=96, filter="nearest")),
        ("transforms/resize_half_bilinear", "resize", dict(img=t, new_w=32, new_h=24, filter="bilinear")),
        ("transforms/resize_half_lanczos", "resize", dict(img=t, new_w=32, new_h=24, filter="lanczos3")),
    ]


def canvas_transform_cases():
    """tests/visual_transforms.rs:25-96,170-228 (64x48 gradient) and tests/scripting.rs:158-168 (64x64 gradient)"""
    t, s = I.create_test_gradient(64, 48), _grad()
    T = "transforms/"
    return [
        (T + "flip_canvas_h", "flip_rotate", dict(img=t, op="flip_horizontal")),
        (T + "flip_canvas_v", "flip_rotate", dict(img=t, op="flip_vertical")),
        (T + "flip_layer_h", "flip_rotate", dict(img=t, op="flip_horizontal")),
        (T + "flip_layer_v", "flip_rotate", dict(img=t, op="flip_vertical")),
        (T + "rotate_90cw", "flip_rotate", dict(img=t, op="rotate_90cw")),
        (T + "rotate_90ccw", "flip_rotate", dict(img=t, op="rotate_90ccw")),
        (T + "rotate_180", "flip_rotate", dict(img=t, op="rotate_180")),
        ("scripting/flip_horizontal", "flip_rotate", dict(img=s, op="flip_horizontal")),
        ("scripting/flip_vertical", "flip_rotate", dict(img=s, op="flip_vertical")),
        (T + "resize_canvas_center", "resize_canvas", dict(img=t, new_w=96, new_h=80, anchor=(1, 1), fill=(0, 0, 0, 0))),
        (T + "resize_canvas_topleft", "resize_canvas", dict(img=t, new_w=80, new_h=64, anchor=(0, 0), fill=(255, 0, 0, 255))),
        (T + "flatten_single", "composite", dict(layers=[dict(pixels=t)], w=64, h=48)),
    ]


def affine_cases():
    """tests/visual_transforms.rs:234-249 (note: passes radians where the function expects degrees) and tests/transform_ops.rs:279-303"""
    deg = lambda v: float(f32(v))
    return [
        ("transforms/affine_rotate_45", "affine_layer", dict(img=I.create_test_gradient(64, 48), rotation_z=deg(f32(45.0) * (f32(np.pi) / f32(180.0))))),
        ("transform/affine_rotate_90", "affine_layer", dict(img=I.create_test_gradient(32, 32), composite=True, rotation_z=deg(f32(np.pi) / f32(2.0)))),
        ("transform/affine_scale_half", "affine_layer", dict(img=I.create_test_gradient(32, 32), composite=True, scale=0.5)),
    ]


def all_cases():
    return (blend_cases() + filter_cases() + rhai_cases() + adjustment_cases() + warp_cases() + brush_cases() + effect_cases() + resize_cases() +
            canvas_transform_cases() + affine_cases())
