"""Parity (GPU): the rest of the effect bank (src/ops/effects/*.rs `*_core`, SURVEY §8f N3 / the Rhai Effect API) through
the C ABI vs the CPU oracle on seeded random inputs — ragged sizes, selection masks, parameter edge cases (identity
settings, clamped parameters, off-canvas origins).

Bars (written per case below):
  * EXACT (tolerance 0): everything built from integer / hash / sqrt / divide arithmetic;
  * LIBM  (+-1 LSB, < 0.1 % of channels off): the four effects that evaluate one libm function per pixel on the device
    (twist: sin/cos, gaussian noise: ln/cos, reduce_noise: exp, vignette: powf) — see k_effects2.hip's header.
"""
import numpy as np
import pytest

from . import inputs as I

pytestmark = pytest.mark.gpu
EXACT, LIBM = "exact", "libm"


@pytest.fixture(scope="module")
def gpu():
    from .backends import GpuBackend
    return GpuBackend(0)


@pytest.fixture(scope="module")
def oracle():
    from .backends import OracleBackend
    return OracleBackend()


def check(a, b, cls, what):
    assert a.shape == b.shape, what
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    if cls == EXACT:
        assert d.max() == 0, f"{what}: max diff {int(d.max())}, {int((d.max(-1) > 0).sum())} px differ"
    else:
        assert d.max() <= 1, f"{what}: max diff {int(d.max())}"
        assert (d > 0).mean() < 1e-3, f"{what}: {(d > 0).mean():.2e} of channels off by one"


def shapes_image(w, h, seed):
    """transparent canvas with a few opaque / semi-transparent blobs (outline, drop shadow)"""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 4), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(5):
        cx, cy, r = rng.integers(0, w), rng.integers(0, h), rng.integers(4, max(5, min(w, h) // 4))
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= r * r
        img[m] = rng.integers(0, 256, 4, dtype=np.uint8)
    img[..., 3] = np.where(img[..., 3] < 40, 0, img[..., 3])
    return img


W, H = 203, 117  # ragged: not a multiple of 4 / 64

CASES = [
    # (effect, class, kwargs)
    ("zoom_blur", EXACT, dict(center_x=0.5, center_y=0.5, strength=0.3, samples=8)),
    ("zoom_blur", EXACT, dict(center_x=0.1, center_y=0.9, strength=0.8, samples=32, tint_color=(1.0, 0.5, 0.25, 1.0), tint_strength=0.7)),
    ("zoom_blur", EXACT, dict(center_x=1.5, center_y=-0.5, strength=5.0, samples=1)),          # clamped strength / samples, off-canvas origin
    ("zoom_blur", EXACT, dict(center_x=0.5, center_y=0.5, strength=0.0005, samples=16)),       # identity (strength < 0.001)
    ("crystallize", EXACT, dict(cell_size=16.0, seed=42)),
    ("crystallize", EXACT, dict(cell_size=5.5, seed=7)),
    ("crystallize", EXACT, dict(cell_size=0.5, seed=1)),                                        # clamped to 2.0
    ("crystallize", EXACT, dict(cell_size=500.0, seed=3)),                                      # one cell
    ("dents", EXACT, dict(scale=20.0, amount=10.0, seed=42, octaves=2, roughness=0.5)),
    ("dents", EXACT, dict(scale=7.0, amount=25.0, seed=5, octaves=8, roughness=0.8, pinch=True, wrap=True)),
    ("dents", EXACT, dict(scale=0.1, amount=3.0, seed=9, octaves=0, roughness=0.3, pinch=True)),
    ("bulge", EXACT, dict(amount=0.5)),
    ("bulge", EXACT, dict(amount=-1.3, origin=(0.2, 0.9))),
    ("bulge", EXACT, dict(amount=0.0)),                                                         # identity
    ("twist", LIBM, dict(angle_deg=45.0)),
    ("twist", LIBM, dict(angle_deg=-300.0, origin=(0.8, 0.1))),
    ("twist", LIBM, dict(angle_deg=0.0)),                                                       # identity
    ("add_noise", EXACT, dict(amount=30.0, noise_type="uniform", monochrome=False, seed=42, scale=1.0, octaves=1)),
    ("add_noise", EXACT, dict(amount=80.0, noise_type="uniform", monochrome=True, seed=3, scale=4.0, octaves=1)),
    ("add_noise", LIBM, dict(amount=30.0, noise_type="gaussian", monochrome=True, seed=42, scale=1.0, octaves=1)),
    ("add_noise", EXACT, dict(amount=30.0, noise_type="gaussian", monochrome=False, seed=42, scale=0.01, octaves=1)),  # colour branch is uniform
    ("add_noise", EXACT, dict(amount=50.0, noise_type="perlin", monochrome=False, seed=42, scale=5.0, octaves=3)),
    ("add_noise", EXACT, dict(amount=50.0, noise_type="perlin", monochrome=True, seed=1, scale=12.0, octaves=20)),
    ("reduce_noise", LIBM, dict(strength=0.5, radius=2)),
    ("reduce_noise", LIBM, dict(strength=40.0, radius=4)),
    ("reduce_noise", LIBM, dict(strength=0.0, radius=0)),
    ("vignette", LIBM, dict(amount=0.8, softness=0.5)),
    ("vignette", LIBM, dict(amount=2.5, softness=0.0)),
    ("vignette", LIBM, dict(amount=0.0, softness=0.5)),                                         # identity
    ("halftone", EXACT, dict(dot_size=4.0, angle_deg=45.0, shape="circle")),
    ("halftone", EXACT, dict(dot_size=7.3, angle_deg=15.0, shape="square")),
    ("halftone", EXACT, dict(dot_size=1.0, angle_deg=-60.0, shape="diamond")),
    ("halftone", EXACT, dict(dot_size=9.0, angle_deg=0.0, shape="line")),
    ("grid", EXACT, dict(cell_w=16, cell_h=16, line_width=1, color=(0, 0, 0, 255), style="lines", opacity=1.0)),
    ("grid", EXACT, dict(cell_w=7, cell_h=13, line_width=3, color=(200, 30, 90, 128), style="lines", opacity=0.35)),
    ("grid", EXACT, dict(cell_w=0, cell_h=9, line_width=0, color=(10, 250, 60, 255), style="checkerboard", opacity=0.6)),
    ("canvas_border", EXACT, dict(width=2, color=(255, 0, 0, 255))),
    ("canvas_border", EXACT, dict(width=0, color=(1, 2, 3, 4))),
    ("canvas_border", EXACT, dict(width=1000, color=(9, 8, 7, 6))),
    ("pixel_drag", EXACT, dict(seed=42, amount=50.0, distance=20, direction=0.0)),
    ("pixel_drag", EXACT, dict(seed=5, amount=100.0, distance=60, direction=37.0)),
    ("pixel_drag", EXACT, dict(seed=5, amount=0.0, distance=0, direction=180.0)),
    ("rgb_displace", EXACT, dict(r_off=(5, 0), g_off=(0, 0), b_off=(-5, 0))),
    ("rgb_displace", EXACT, dict(r_off=(-300, 7), g_off=(3, -4), b_off=(0, 500))),
    ("ink", EXACT, dict(edge_strength=1.0, threshold=0.5)),
    ("ink", EXACT, dict(edge_strength=35.0, threshold=40.0)),
    ("oil_painting", EXACT, dict(radius=3, levels=20)),
    ("oil_painting", EXACT, dict(radius=1, levels=2)),
    ("oil_painting", EXACT, dict(radius=10, levels=64)),
    ("oil_painting", EXACT, dict(radius=0, levels=30)),
    ("oil_painting", EXACT, dict(radius=50, levels=500)),                                       # clamped to 10 / 64
    ("color_filter", EXACT, dict(filter_color=(255, 128, 0, 255), intensity=0.5, mode="multiply")),
    ("color_filter", EXACT, dict(filter_color=(20, 200, 90, 255), intensity=0.8, mode="screen")),
    ("color_filter", EXACT, dict(filter_color=(120, 130, 250, 255), intensity=1.0, mode="overlay")),
    ("color_filter", EXACT, dict(filter_color=(60, 127, 128, 255), intensity=0.65, mode="soft_light")),
    ("color_filter", EXACT, dict(filter_color=(255, 255, 255, 255), intensity=0.0, mode="multiply")),  # identity
    ("contours", EXACT, dict(scale=10.0, frequency=5.0, line_width=1.0, line_color=(0, 0, 0, 255), seed=42, octaves=2, blend=0.5)),
    ("contours", EXACT, dict(scale=33.0, frequency=0.1, line_width=4.0, line_color=(250, 20, 20, 200), seed=8, octaves=5, blend=1.0)),
]


@pytest.mark.parametrize("name,cls,kw", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_effect_vs_oracle(gpu, oracle, name, cls, kw):
    img = I.random_rgba(W, H, 31 + len(name))
    mask = (np.random.default_rng(77).random((H, W)) < 0.6).astype(np.uint8) * 255
    check(gpu.effect(name, img, **kw), oracle.effect(name, img, **kw), cls, name)
    check(gpu.effect(name, img, mask=mask, **kw), oracle.effect(name, img, mask=mask, **kw), cls, name + " masked")


SHAPE_CASES = [
    ("shadow", dict(offset_x=5, offset_y=5, blur_radius=3.0, widen_radius=False, color=(0, 0, 0, 255), opacity=0.8)),
    ("shadow", dict(offset_x=-7, offset_y=3, blur_radius=4.4, widen_radius=True, color=(30, 60, 200, 180), opacity=1.0)),
    ("shadow", dict(offset_x=0, offset_y=0, blur_radius=0.3, widen_radius=True, color=(255, 255, 255, 255), opacity=0.5)),   # no blur, spread 1
    ("shadow", dict(offset_x=500, offset_y=-500, blur_radius=2.0, widen_radius=False, color=(0, 0, 0, 255), opacity=0.8)),   # shadow off-canvas
    ("outline", dict(width=2, color=(0, 0, 255, 255), mode="outside", anti_alias=True)),
    ("outline", dict(width=5, color=(10, 200, 30, 160), mode="inside", anti_alias=True)),
    ("outline", dict(width=3, color=(250, 250, 0, 255), mode="center", anti_alias=False)),
    ("outline", dict(width=0, color=(0, 0, 0, 255), mode="outside", anti_alias=False)),
]


@pytest.mark.parametrize("name,kw", SHAPE_CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(SHAPE_CASES)])
def test_shape_effects_bitexact(gpu, oracle, name, kw):
    img = shapes_image(180, 140, 5)
    mask = np.zeros((140, 180), np.uint8)
    mask[10:120, 20:170] = 255
    check(gpu.effect(name, img, **kw), oracle.effect(name, img, **kw), EXACT, name)
    check(gpu.effect(name, img, mask=mask, **kw), oracle.effect(name, img, mask=mask, **kw), EXACT, name + " masked")


def test_outline_on_empty_and_full_canvas(gpu, oracle):
    kw = dict(width=3, color=(255, 0, 0, 255), mode="center", anti_alias=True)
    empty = np.zeros((70, 90, 4), np.uint8)
    empty[..., :3] = 77  # colour under zero alpha must survive (`flat.clone()`)
    check(gpu.effect("outline", empty, **kw), empty, EXACT, "outline of an empty layer is the layer")
    full = I.random_rgba(90, 70, 3)
    full[..., 3] = np.maximum(full[..., 3], 1)
    check(gpu.effect("outline", full, **kw), oracle.effect("outline", full, **kw), EXACT, "outline of a full layer")


@pytest.mark.parametrize("width", [1, 2, 6, 8, 14, 15, 20])
@pytest.mark.parametrize("size", [(97, 61), (64, 64), (33, 5), (1, 1), (200, 2)])
def test_outline_bit_plane_search_equals_the_scan_and_the_oracle(gpu, oracle, width, size):
    """the outline's nearest filled / empty texel from a bit plane of alpha != 0 (search radius = width + 1 <= 15; wider windows keep the per-element
    scan): every mode, with and without anti-aliasing, shapes touching the borders (window columns and rows outside the image are skipped, not
    clamped), plane rows of 2 .. 9 dwords; the scan (pfx_tune outline_bits = 0) must give the same image"""
    w, h = size
    img = shapes_image(w, h, 31 * w + width)
    img[0, :, 3] = 255                                                 # a filled top row: windows hanging over the border
    img[:, w - 1, 3] = 0
    for mode in ("outside", "inside", "center"):
        for aa in (False, True):
            kw = dict(width=width, color=(20, 200, 250, 200), mode=mode, anti_alias=aa)
            ref = oracle.effect("outline", img, **kw)
            check(gpu.effect("outline", img, **kw), ref, EXACT, f"outline bits {kw} {size}")
            gpu.r.tune("outline_bits", 0)
            try:
                check(gpu.effect("outline", img, **kw), ref, EXACT, f"outline scan {kw} {size}")
            finally:
                gpu.r.tune("outline_bits", 1)


def test_effects_small_and_wide_images(gpu, oracle):
    """1-pixel-high / 1-pixel-wide / single-pixel images hit every clamp"""
    for (w, h) in ((1, 1), (300, 1), (1, 77), (65, 3)):
        img = I.random_rgba(w, h, w * 7 + h)
        for name, cls, kw in (("bulge", EXACT, dict(amount=0.7)), ("ink", EXACT, dict(edge_strength=2.0, threshold=0.5)),
                              ("oil_painting", EXACT, dict(radius=2, levels=8)), ("crystallize", EXACT, dict(cell_size=4.0, seed=2)),
                              ("zoom_blur", EXACT, dict(center_x=0.5, center_y=0.5, strength=0.5, samples=4)),
                              ("reduce_noise", LIBM, dict(strength=10.0, radius=3)), ("vignette", LIBM, dict(amount=0.9, softness=0.4)),
                              ("dents", EXACT, dict(scale=3.0, amount=4.0, seed=1, octaves=2, roughness=0.5, wrap=True))):
            check(gpu.effect(name, img, **kw), oracle.effect(name, img, **kw), cls, f"{name} {w}x{h}")


@pytest.mark.parametrize("filter", ["nearest", "bilinear", "bicubic", "lanczos3"])
@pytest.mark.parametrize("size,new", [((64, 48), (128, 96)), ((203, 117), (50, 30)), ((203, 117), (640, 11)), ((37, 90), (1, 1)), ((5, 3), (333, 222)),
                                      ((120, 80), (120, 40)), ((120, 80), (120, 80))])
def test_resize_image_bitexact(gpu, oracle, filter, size, new):
    img = I.random_rgba(size[0], size[1], 17)
    check(gpu.resize(img, new[0], new[1], filter), oracle.resize(img, new[0], new[1], filter), EXACT, f"resize {size}->{new} {filter}")


AFFINE = [
    dict(rotation_z=33.0), dict(rotation_z=-170.0, scale=1.7, offset=(12.5, -8.25)), dict(rotation_x=35.0, rotation_y=-20.0, rotation_z=10.0),
    dict(scale=0.0), dict(scale=-0.4, rotation_y=80.0), dict(rotation_x=90.0), dict(rotation_z=90.0, interpolation="nearest"),
    dict(scale=3.0, offset=(-40.0, 15.0), interpolation="nearest"), dict(),
]


@pytest.mark.parametrize("kw", AFFINE, ids=[str(i) for i in range(len(AFFINE))])
def test_affine_transform_bitexact(gpu, kw):
    from . import oracle_lib as O
    img = I.random_rgba(131, 77, 21)
    img[:20, :30, 3] = 0
    check(gpu.r.affine_transform(img, 131, 77, **kw), O.affine(img, 131, 77, **kw), EXACT, f"affine {kw}")
    # canvas larger / smaller than the layer image
    check(gpu.r.affine_transform(img, 200, 40, **kw), O.affine(img, 200, 40, **kw), EXACT, f"affine {kw} on a 200x40 canvas")


def test_resize_in_scripts(gpu):
    img = I.random_rgba(90, 60, 4)
    out, _, ops = gpu.r.execute_script_sync('resize_image(45, 30, "lanczos"); resize_image(45, 30, "nearest"); resize_image(200, 10, "whatever");', img, with_ops=True)
    from . import oracle_lib as O
    ref = O.resize(O.resize(img, 45, 30, "lanczos3"), 200, 10, "bilinear")   # unknown method names mean bilinear; same-size calls are no-ops
    assert np.array_equal(out, ref)
    assert ops == [(5, 45, 30, 3, 0), (5, 200, 10, 1, 0)]


def test_effect_argument_errors(gpu):
    from paintfe_amd import PfxError
    img = I.random_rgba(32, 32, 1)
    with pytest.raises(PfxError):
        gpu.r.reduce_noise_core(img, 1.0, 1000)
    with pytest.raises(PfxError):
        gpu.r.outline_core(img, 100000, (0, 0, 0, 255))
    with pytest.raises(PfxError):
        gpu.r.add_noise_core(img, 1.0, 7, False, 1, 1.0, 1)
    # parameters that set a per-pixel loop count are bounded: a hostile value is a status, not a launch that never ends
    with pytest.raises(PfxError):
        gpu.r.motion_blur_core(img, 30.0, 1.0e9)
    with pytest.raises(PfxError):
        gpu.r.bokeh_blur_core(img, 1.0e9)
    with pytest.raises(PfxError):
        gpu.r.box_blur_core(img, 1.0e9)
    with pytest.raises(PfxError):
        gpu.r.gaussian_blur_core(img, 1.0e9)
    from . import oracle_lib as O   # `distance < 1.0` is false for NaN and ceil(NaN) as i32 = 0 steps (blur.rs:150-160): whatever that gives, both sides give it
    assert np.array_equal(gpu.r.motion_blur_core(img, 30.0, float("nan")), O.motion_blur(img, 30.0, float("nan")))


def test_drop_shadow_blurs_its_alpha_plane_with_the_same_bits(gpu, oracle):
    """round 6: where the bit-exact fused Gaussian applies (radius <= 16, rows dword-aligned) the drop shadow blurs its ONE-CHANNEL alpha plane instead of the reference's
    (a, a, a, a) expansion (render.rs:291-301) — per element the same products and sums.  Equal to the oracle and to the RGBA form (pfx_tune "shadow_plane" = 0) for blur
    radii on both sides of every switch (none / fused / beyond 16), with and without widening and a selection"""
    rng = np.random.default_rng(91)
    for (w, h) in [(256, 96), (64, 64), (1024, 40), (4, 9), (260, 33)]:
        img = shapes_image(w, h, seed=w + h)
        mask = None if (w // 4) % 2 else ((rng.random((h, w)) < 0.6).astype(np.uint8) * 255)
        for blur in (0.4, 1.0, 3.0, 5.33, 6.0):
            for widen in (False, True):
                kw = dict(offset_x=5, offset_y=-3, blur_radius=blur, widen_radius=widen, color=(20, 40, 200, 230), opacity=0.8, mask=mask)
                want = oracle.effect("shadow", img, **kw)
                try:
                    gpu.r.tune("shadow_plane", 1)
                    got = gpu.effect("shadow", img, **kw)
                    gpu.r.tune("shadow_plane", 0)
                    old = gpu.effect("shadow", img, **kw)
                finally:
                    gpu.r.tune("shadow_plane", 1)
                check(got, want, EXACT, f"drop shadow (plane) {w}x{h} blur {blur} widen {widen}")
                check(old, want, EXACT, f"drop shadow (rgba) {w}x{h} blur {blur} widen {widen}")
