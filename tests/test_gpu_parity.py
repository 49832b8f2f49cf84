"""Parity gate 2 (GPU): HIP path (through the C ABI) vs the CPU oracle on seeded random inputs at sizes the oracle
finishes in seconds, including ragged sizes (not multiples of 4 / 64), masks, empty chunks and the edge cases the
reference tests.  Bars: bit-exact for the compositor, the integer filters, LUT ops and (as it turns out) every
float op evaluated in reference order; Gaussian with fused multiply-add is held to the stated ±1 LSB and is
bit-exact in `exact` mode."""
import numpy as np
import pytest

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from .backends import GpuBackend
    return GpuBackend(0)


@pytest.fixture(scope="module")
def oracle():
    from .backends import OracleBackend
    return OracleBackend()


def assert_same(a, b, tol=0, what=""):
    assert a.shape == b.shape, what
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    assert d.max() <= tol, f"{what}: max diff {int(d.max())}, {int((d.max(-1) > tol).sum())} px over tolerance {tol}"


def sparse_alpha_image(w, h, seed):
    """random RGBA with fully transparent 64x64 chunks and transparent-but-coloured pixels (chunk rule coverage)"""
    img = I.random_rgba(w, h, seed)
    rng = np.random.default_rng(seed + 99)
    img[..., 3] = np.where(rng.random((h, w)) < 0.3, 0, img[..., 3])
    for cy in range(0, h, 64):
        for cx in range(0, w, 64):
            if rng.random() < 0.35:
                img[cy:cy + 64, cx:cx + 64, 3] = 0
    return img


# ------------------------------------------------------------------ compositor
@pytest.mark.parametrize("mode", range(25))
def test_blend_mode_random_bitexact(gpu, oracle, mode):
    w, h = 333, 131  # ragged: not a multiple of 4 or 64
    base = I.random_rgba(w, h, 1000 + mode)
    top = sparse_alpha_image(w, h, 2000 + mode)
    for opacity in (1.0, 0.37, 0.0, 1.5):
        layers = [dict(pixels=base), dict(pixels=top, mode=mode, opacity=opacity)]
        assert_same(gpu.composite(layers, w, h), oracle.composite(layers, w, h), 0, f"mode {mode} opacity {opacity}")


def test_blend_pixels_exhaustive_alpha_lattice(gpu):
    """every (base alpha, top alpha) pair x a colour sweep, all 25 modes, through pfx_blend_pixels"""
    ba, ta = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (256 * 256, 4), dtype=np.uint8)
    top = rng.integers(0, 256, (256 * 256, 4), dtype=np.uint8)
    base[:, 3] = ba.ravel()
    top[:, 3] = ta.ravel()
    for mode in range(25):
        for opacity in (1.0, 0.5):
            got = gpu.r.blend_pixels(base, top, mode, opacity)
            exp = np.stack([O.blend_pixel(base[i], top[i], mode, opacity) for i in range(0, len(base), 97)])
            assert_same(got[::97], exp, 0, f"mode {mode} opacity {opacity}")


def test_typed_unorm8_store_and_load_round_trip_every_byte_value(gpu):
    """the streaming compositors read every layer through typed UNORM8 buffer loads and hold their accumulators as RN(k / 255) (k_flatten.hip:
    flatten_srt_kernel / flatten_stream_kernel; round 3's kernel also parks accumulators with typed stores): the texture path's conversions must be exact
    in both directions for all 256 byte values on every channel"""
    assert gpu.r.selftest_unorm_store() == 0


def test_round_and_pack_matches_rust_rounding_for_every_float(gpu):
    """k_common.h:pack_round_rgba (v_med3_f32, OR 1, v_cvt_pk_u8_f32) vs `v.round().clamp(0.0, 255.0) as u8` evaluated step by step, for all
    2^32 f32 bit patterns: identical except for signalling NaNs, which no arithmetic produces"""
    bad, snan = gpu.r.selftest_round_pack()
    assert bad == 0
    assert snan <= 2 * (2 ** 22 - 1)


def test_fast_division_matches_ieee(gpu):
    """k_flatten.hip:rdiv (shared refined reciprocal) vs the compiler's IEEE f32 divide on 2^28 random operand pairs
    from the compositor's operand range — must be bit-identical."""
    assert gpu.r.selftest_division(seed=0xD1D1, n_millions=268) == 0


def test_tiny_opacity_takes_ieee_division_path(gpu, oracle):
    """opacity below 2^-40 selects the plain '/' instantiation (operands may leave the normal range)"""
    w, h = 130, 70
    base = I.random_rgba(w, h, 91)
    top = I.random_rgba(w, h, 92)
    for mode in (0, 1, 6, 13, 19):
        for opacity in (1e-20, 3e-39, 1e-44):
            layers = [dict(pixels=base), dict(pixels=top, mode=mode, opacity=opacity)]
            assert_same(gpu.composite(layers, w, h), oracle.composite(layers, w, h), 0, f"mode {mode} opacity {opacity}")


def test_flatten_stack_8_layers(gpu):
    w, h, n = 515, 259, 8
    stack, modes, opac = I.layer_stack(w, h, n, seed=42)
    layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(n)]
    assert_same(gpu.composite(layers, w, h), O.flatten_stack(stack, modes, opac), 0, "8-layer stack")


def test_flatten_all_modes_32_layers(gpu):
    w, h, n = 260, 130, 32
    stack, modes, opac = I.layer_stack(w, h, n, seed=77)
    layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(n)]
    assert_same(gpu.composite(layers, w, h), O.flatten_stack(stack, modes, opac), 0, "32-layer stack")


@pytest.mark.parametrize("mode", range(25))
def test_flatten_opaque_base_path_bitexact(gpu, mode):
    """the wave-uniform "accumulator is opaque" specialisation of the compositor (k_flatten.hip, blend_px<.., OB>): an opaque
    background under a layer of every mode with the full alpha lattice and several opacities, then two more layers so that
    whatever alpha the mode leaves behind (Xor / Overwrite lower it) is carried through the general path again"""
    w, h = 512, 64                                   # 128 waves of 256 px, all with an opaque accumulator after layer 0
    rng = np.random.default_rng(900 + mode)
    bg = I.random_rgba(w, h, 901 + mode)
    bg[..., 3] = 255
    top = I.random_rgba(w, h, 902 + mode)
    top[..., 3] = np.tile(np.arange(256, dtype=np.uint8), (h, w // 256))   # every alpha value in every row
    top[:8, :, :3] = rng.integers(0, 2, (8, w, 3), dtype=np.uint8) * 255    # extreme colours: 0/255 hit every branch edge
    over = I.random_rgba(w, h, 903 + mode)
    for opacity in (1.0, 0.37, 1e-3, 0.999999):
        stack = np.stack([bg, top, over, top[::-1].copy()])
        modes = np.array([0, mode, (mode + 7) % 25, mode], np.uint8)
        opac = np.array([1.0, opacity, 0.8, 1.0], np.float32)
        layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(4)]
        assert_same(gpu.composite(layers, w, h), O.flatten_stack(stack, modes, opac), 0, f"mode {mode} opacity {opacity}")
    # second level of the specialisation: the top layer is opaque everywhere and at 100 % (or more) opacity
    solid = top.copy()
    solid[..., 3] = 255
    for opacity in (1.0, 1.5):
        stack = np.stack([bg, solid, over, solid[:, ::-1].copy()])
        modes = np.array([0, mode, (mode + 7) % 25, mode], np.uint8)
        opac = np.array([1.0, opacity, 0.8, 1.0], np.float32)
        layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(4)]
        assert_same(gpu.composite(layers, w, h), O.flatten_stack(stack, modes, opac), 0, f"mode {mode} opaque top, opacity {opacity}")


def test_composite_masks_hidden_and_adjustment_layers(gpu, oracle):
    w, h = 200, 150
    rng = np.random.default_rng(3)
    bg = sparse_alpha_image(w, h, 11)
    fg = sparse_alpha_image(w, h, 12)
    mask = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mask[rng.random((h, w)) < 0.3] = 0
    layers = [
        dict(pixels=bg),
        dict(kind=O.ADJ_INVERT, opacity=0.6),
        dict(pixels=fg, mode=8, opacity=0.8, mask=mask),
        dict(pixels=I.random_rgba(w, h, 13), visible=False, mode=3),
        dict(kind=O.ADJ_EXPOSURE, adj=[0.7], opacity=1.0),
        dict(kind=O.ADJ_BC, adj=[12.0, 30.0], opacity=0.5),
        dict(kind=O.ADJ_MIXER, adj=[0.5, 0.3, 0.2, 0.0, 0.1, 0.8, 0.1, 0.0, 0.0, 0.2, 0.8, 0.0, 0.0, 0.0, 0.0, 1.0], opacity=0.9),
    ]
    assert_same(gpu.composite(layers, w, h), oracle.composite(layers, w, h), 0, "masks + adjustment layers")


def test_composite_empty_and_single_hidden(gpu, oracle):
    w, h = 70, 66
    img = I.random_rgba(w, h, 1)
    assert_same(gpu.composite([dict(pixels=img, visible=False)], w, h), np.zeros((h, w, 4), np.uint8), 0, "all hidden")
    assert_same(gpu.composite([dict(pixels=img)], w, h), oracle.composite([dict(pixels=img)], w, h), 0, "single")


def test_layer_store_generation_and_dirty_rect(gpu):
    r = gpu.r
    r.clear_layers()
    w, h = 128, 96
    a = I.random_rgba(w, h, 21)
    b = I.random_rgba(w, h, 22)
    r.ensure_layer_texture(0, a, generation=5)
    r.ensure_layer_texture(0, b, generation=5)  # same generation: upload skipped (ref: renderer.rs:336-342)
    assert_same(r.composite(w, h, [(0, 1.0, True, 0)]), O.composite([dict(pixels=a)], w, h), 0, "generation skip")
    r.ensure_layer_texture(0, b, generation=6)
    patch = I.random_rgba(40, 20, 23)
    r.update_layer_rect(0, 64, 10, patch)
    b2 = b.copy()
    b2[10:30, 64:104] = patch
    assert_same(r.composite(w, h, [(0, 1.0, True, 0)]), O.composite([dict(pixels=b2)], w, h), 0, "update_rect")
    reg = r.composite_dirty_readback(w, h, [(0, 0.5, True, 0)], (8, 4, 64, 32))
    assert_same(reg, O.composite([dict(pixels=b2, opacity=0.5)], w, h)[4:36, 8:72], 0, "dirty readback")
    assert r.active_texture_count() == 1 and r.active_texture_memory() == w * h * 4
    r.remove_layer(0)
    assert r.active_texture_count() == 0


@pytest.mark.parametrize("canvas", [(256, 192), (203, 117)])
def test_composite_dirty_rectangles(gpu, canvas):
    """composite_dirty_readback composites only the rectangle (renderer.rs:588): aligned and ragged rectangles, edge-touching ones,
    1-pixel ones, on a stack with a mask and an adjustment layer — each must equal the crop of the full composite"""
    w, h = canvas
    layers = [dict(pixels=I.random_rgba(w, h, 31)), dict(pixels=sparse_alpha_image(w, h, 32), mode=6, opacity=0.7,
                                                          mask=(np.random.default_rng(3).random((h, w)) < 0.4).astype(np.uint8) * 180),
              dict(kind=2, adj=[15.0, 30.0], opacity=0.6), dict(pixels=sparse_alpha_image(w, h, 33), mode=21, opacity=1.0)]
    full = O.composite(layers, w, h)
    gpu.composite(layers, w, h)  # uploads the stack
    info = [(i, L.get("opacity", 1.0), True, L.get("mode", 0), L.get("kind", 0), L.get("adj", ())) for i, L in enumerate(layers)]
    for (x, y, rw, rh) in ((64, 32, 64, 64), (0, 0, w, 1), (0, 0, 1, h), (w - 1, h - 1, 1, 1), (5, 7, 50, 33), (128, 0, w - 128, h), (3, 100, 13, 2),
                           (0, 0, w, h)):
        reg = gpu.r.composite_dirty_readback(w, h, info, (x, y, rw, rh))
        assert_same(reg, full[y:y + rh, x:x + rw], 0, f"dirty rect {(x, y, rw, rh)}")


# ------------------------------------------------------------------ stencils
@pytest.mark.parametrize("sigma", [0.0, 0.3, 1.0, 2.5, 5.0, 16.0])
@pytest.mark.parametrize("size", [(64, 64), (301, 97), (1100, 150)])
def test_gaussian_vs_oracle(gpu, sigma, size):
    w, h = size
    img = I.random_rgba(w, h, int(sigma * 10) + w)
    ref = O.gaussian_blur(img, sigma)
    gpu.r.set_exact(False)
    assert_same(gpu.gaussian_blur(img, sigma), ref, 1, f"fma sigma={sigma} {w}x{h}")  # stated tolerance: ±1 LSB
    gpu.r.set_exact(True)
    assert_same(gpu.gaussian_blur(img, sigma), ref, 0, f"exact sigma={sigma} {w}x{h}")
    gpu.r.set_exact(False)


@pytest.mark.parametrize("sigma", [17.0, 20.0, 24.0, 26.6, 26.7, 40.0, 60.0, 75.0, 100.0, 120.0])
def test_gaussian_advanced_dialog_sigmas(gpu, sigma):
    """sigma 16 .. 100 (the reference's advanced blur dialog, ui/dialogs/core/image.rs) is beyond the matrix-core kernel: the VALU passes with
    the vertical tile chosen by radius — 8 x 128 (r <= 110), 16 x 256 (<= 160), 8 x 512 (<= 240), 8 x 256 (<= 340), 8 x 128 again beyond;
    exact mode bit-exact, default mode within the stated 1 LSB.  The image is shorter than the window at the large radii (all rows clamp)."""
    w, h = 150, 330
    img = I.random_rgba(w, h, int(sigma))
    ref = O.gaussian_blur(img, sigma)
    gpu.r.set_exact(False)
    assert_same(gpu.gaussian_blur(img, sigma), ref, 1, f"fma sigma={sigma}")
    gpu.r.set_exact(True)
    try:
        assert_same(gpu.gaussian_blur(img, sigma), ref, 0, f"exact sigma={sigma}")
    finally:
        gpu.r.set_exact(False)


@pytest.mark.parametrize("size", [(1, 1), (5, 3), (33, 2), (2, 70), (31, 31), (64, 1), (4, 200), (129, 7)])
def test_gaussian_wide_kernels_on_small_images(gpu, size):
    """10 / 12 K-block instantiations of the matrix-core kernel (sigma 17 .. 26.6) on images far smaller than their 160 / 192-sample windows and
    than one 32 x 32 output block: every sample clamps, strips and row segments are partial"""
    w, h = size
    img = I.random_rgba(w, h, 7 * w + h)
    for sigma in (17.0, 20.0, 26.6):
        assert_same(gpu.gaussian_blur(img, sigma), O.gaussian_blur(img, sigma), 1, f"sigma={sigma} {w}x{h}")


def test_gaussian_fma_mismatch_rate_is_float_noise(gpu):
    img = I.random_rgba(1024, 512, 5)
    ref = O.gaussian_blur(img, 16.0)
    out = gpu.gaussian_blur(img, 16.0)
    d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    assert d.max() <= 1
    assert (d > 0).mean() < 1e-3, f"{(d > 0).mean():.2e} of channels differ"


@pytest.mark.parametrize("parts", [12, 22, 11])
@pytest.mark.parametrize("sigma", [0.8, 4.0, 16.0, 24.0])
def test_gaussian_matrix_core_piece_counts(gpu, parts, sigma):
    """k_gauss.hip WP / HP: one f16 per weight (shipped, "gauss_parts" = 12), the two-piece split of rounds 2-3 (22) and one piece for the horizontal
    result as well (11, measured and not shipped) all stay within 1 LSB of the CPU path — on noise, on flat fields at every level that matters
    (0, 1, 127, 128, 254, 255: a flat image must blur to ITSELF, the table's sum is the exact taps' sum), on a hard edge and on a 1-pixel checkerboard."""
    w, h = 320, 160
    rng = np.random.default_rng(int(sigma * 10) + parts)
    noise = I.random_rgba(w, h, 77 + parts)
    edge = np.zeros((h, w, 4), np.uint8); edge[:, w // 2:, :] = 255; edge[h // 2:, :, 1] = 255 - edge[h // 2:, :, 1]
    yy, xx = np.mgrid[0:h, 0:w]
    checker = (((xx + yy) & 1) * 255).astype(np.uint8)[..., None].repeat(4, axis=2)
    gpu.r.tune("gauss_parts", parts)
    try:
        for name, img in (("noise", noise), ("edge", edge), ("checker", checker)):
            assert_same(gpu.gaussian_blur(img, sigma), O.gaussian_blur(img, sigma), 1, f"parts={parts} sigma={sigma} {name}")
        for level in (0, 1, 127, 128, 254, 255):
            flat = np.full((h, w, 4), level, np.uint8)
            assert_same(gpu.gaussian_blur(flat, sigma), flat, 0, f"parts={parts} sigma={sigma} flat {level}")
    finally:
        gpu.r.tune("gauss_parts", 12)
    _ = rng


def test_gaussian_identity_and_selection(gpu, oracle):
    img = I.random_rgba(200, 120, 9)
    assert_same(gpu.gaussian_blur(img, 0.0), img, 0, "sigma=0 identity (visual_filters.rs:291)")
    mask = np.zeros((120, 200), np.uint8)
    mask[30:70, 50:140] = 255
    mask[35, 60] = 0
    mask[100:110, 180:200] = 7  # grey mask values count as selected (mask > 0)
    gpu.r.set_exact(True)
    assert_same(gpu.gaussian_blur(img, 3.0, mask), oracle.gaussian_blur(img, 3.0, mask), 0, "blur_with_selection")
    assert_same(gpu.gaussian_blur(img, 3.0, np.zeros_like(mask)), img, 0, "empty selection")
    gpu.r.set_exact(False)


@pytest.mark.parametrize("radius", [0.2, 0.5, 1.0, 3.0, 7.5, 48.0])
def test_box_blur_bitexact(gpu, oracle, radius):
    img = I.random_rgba(2100, 75, int(radius * 7))
    mask = (np.random.default_rng(1).random((75, 2100)) < 0.7).astype(np.uint8) * 200
    assert_same(gpu.box_blur(img, radius), oracle.box_blur(img, radius), 0, f"box r={radius}")
    assert_same(gpu.box_blur(img, radius, mask), oracle.box_blur(img, radius, mask), 0, f"box r={radius} masked")


MEDIAN_BITS_MIN = 3  # pfx_ctx default (pfx_internal.h): radii 3..7 take the bit-plane select


@pytest.mark.parametrize("radius", [0, 1, 2, 3, 7, 12])
def test_median_bitexact(gpu, oracle, radius):
    img = I.random_rgba(131, 77, 40 + radius)
    img[10:30, 10:50] = (img[10:30, 10:50] // 64) * 64  # heavy ties
    mask = (np.random.default_rng(2).random((77, 131)) < 0.6).astype(np.uint8)
    assert_same(gpu.median(img, radius), oracle.median(img, radius), 0, f"median r={radius}")
    assert_same(gpu.median(img, radius, mask), oracle.median(img, radius, mask), 0, f"median r={radius} masked")


@pytest.mark.parametrize("radius", [2, 3, 4])
@pytest.mark.parametrize("size", [(1, 1), (3, 2), (4, 9), (7, 7), (255, 5), (256, 4), (257, 6), (260, 64), (512, 3), (1031, 37)])
def test_median_shared_column_networks(gpu, oracle, radius, size):
    """radii 2, 3 and 4: four windows per lane on shared sorted columns (k_median_shared_net.h) — block edges (256 x 4 pixel blocks), image
    edges (clamped windows wider than the image), widths that are and are not multiples of 4, masks, heavy ties; the single-window
    networks / the value search (pfx_tune "median_single") must give the same image"""
    w, h = size
    img = I.random_rgba(w, h, 600 + w + radius)
    img[: h // 2] = (img[: h // 2] // 86) * 86  # three levels per channel: ties everywhere
    mask = (np.random.default_rng(w).random((h, w)) < 0.5).astype(np.uint8)
    ref, ref_m = oracle.median(img, radius), oracle.median(img, radius, mask)
    gpu.r.tune("median_bits_min", 9)  # the network kernels, not the bit-plane select
    try:
        for single in (0, 1):
            gpu.r.tune("median_single", single)
            assert_same(gpu.median(img, radius), ref, 0, f"median r={radius} {w}x{h} single={single}")
            assert_same(gpu.median(img, radius, mask), ref_m, 0, f"median r={radius} {w}x{h} masked single={single}")
    finally:
        gpu.r.tune("median_single", 0)
        gpu.r.tune("median_bits_min", MEDIAN_BITS_MIN)


@pytest.mark.parametrize("radius", [4, 5, 7, 12, 24])
@pytest.mark.parametrize("size", [(1, 1), (6, 3), (127, 9), (128, 8), (131, 23), (390, 41)])
def test_median_value_search_four_pixels_per_lane(gpu, oracle, radius, size):
    """radii 5..24 (and 4 when the networks are switched off): the value search with four pixels per lane sharing the columns they load —
    block edges every 128 x 8 pixels, widths that are not multiples of 4, windows wider than the image, masks, ties; the one-pixel-per-lane
    search (pfx_tune "median_search1") must give the same image"""
    w, h = size
    img = I.random_rgba(w, h, 1200 + w + radius)
    img[: h // 2] = (img[: h // 2] // 100) * 100
    mask = (np.random.default_rng(w + 3).random((h, w)) < 0.5).astype(np.uint8)
    ref, ref_m = oracle.median(img, radius), oracle.median(img, radius, mask)
    gpu.r.tune("median_single", 1)   # radius 4 takes the search too
    gpu.r.tune("median_bits_min", 9)  # ... and radii 4..8 do not take the bit-plane select
    try:
        for one in (0, 1):
            gpu.r.tune("median_search1", one)
            assert_same(gpu.median(img, radius), ref, 0, f"median r={radius} {w}x{h} search1={one}")
            assert_same(gpu.median(img, radius, mask), ref_m, 0, f"median r={radius} {w}x{h} masked search1={one}")
    finally:
        gpu.r.tune("median_search1", 0)
        gpu.r.tune("median_single", 0)
        gpu.r.tune("median_bits_min", MEDIAN_BITS_MIN)


@pytest.mark.parametrize("radius", [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("size", [(1, 1), (5, 3), (16, 40), (31, 33), (32, 32), (33, 31), (63, 70), (64, 16), (65, 97), (127, 9), (128, 35), (129, 66), (131, 23), (390, 41), (1031, 37)])
@pytest.mark.parametrize("pair", [1, 0])
def test_median_bit_plane_radix_select(gpu, oracle, radius, size, pair):
    """k_median_bits.hip: the window as bit planes, rank select from the top plane down.  Sizes cross the 32-pixel dwords of a plane row,
    the 16-column waves (32 for the column-pair kernel) and the blocks of 64 / 128 columns, the 32-row bands and the 2r+1-row ring (several turns),
    windows wider / taller than the image, odd widths (a pair whose second column is outside), masks, heavy ties (the select must count equal elements
    exactly like the sort).  pair = 1: two adjacent columns per lane on shared plane registers (radii 2..7, shipped); 0: one column per lane (radius 8, and
    every radius under pfx_tune "median_pair" = 0)"""
    w, h = size
    if pair and radius == 8:
        pytest.skip("radius 8 has no column-pair build (18-bit fields)")
    img = I.random_rgba(w, h, 2100 + 7 * w + radius)
    img[: h // 2] = (img[: h // 2] // 100) * 100
    img[:, : w // 3, 1] = 255
    img[:, w // 2:, 2] = 0
    mask = (np.random.default_rng(w + 5).random((h, w)) < 0.5).astype(np.uint8)
    gpu.r.tune("median_bits_min", 2)
    gpu.r.tune("median_pair", pair)
    try:
        assert_same(gpu.median(img, radius), oracle.median(img, radius), 0, f"median bits r={radius} {w}x{h}")
        assert_same(gpu.median(img, radius, mask), oracle.median(img, radius, mask), 0, f"median bits r={radius} {w}x{h} masked")
    finally:
        gpu.r.tune("median_bits_min", MEDIAN_BITS_MIN)
        gpu.r.tune("median_pair", 1)


@pytest.mark.parametrize("seed", range(12))
def test_median_and_box_blur_random_shapes(gpu, oracle, seed):
    """seeded random sizes and radii through every median / box blur kernel family (3x3 network, shared-column networks, bit-plane select for
    r = 3..8, value search, sliding histogram; fused and two-pass box blur at both lane-run sizes), with and without a selection mask"""
    rng = np.random.default_rng(9000 + seed)
    w, h = int(rng.integers(1, 700)), int(rng.integers(1, 260))
    img = I.random_rgba(w, h, 9100 + seed)
    if seed % 3 == 0:
        img[..., :3] = (img[..., :3] // 64) * 64                      # few levels: ties in every window
    mask = (rng.random((h, w)) < 0.5).astype(np.uint8) * 255
    for radius in sorted({int(rng.integers(1, 9)), int(rng.integers(1, 9)), int(rng.integers(9, 26))}):
        assert_same(gpu.median(img, radius), oracle.median(img, radius), 0, f"median r={radius} {w}x{h}")
        assert_same(gpu.median(img, radius, mask), oracle.median(img, radius, mask), 0, f"median r={radius} {w}x{h} masked")
    for radius in (float(rng.integers(1, 5)), float(rng.integers(5, 24)) + 0.5, float(rng.integers(24, 140))):
        assert_same(gpu.box_blur(img, radius), oracle.box_blur(img, radius), 0, f"box r={radius} {w}x{h}")
        assert_same(gpu.box_blur(img, radius, mask), oracle.box_blur(img, radius, mask), 0, f"box r={radius} {w}x{h} masked")


@pytest.mark.parametrize("size", [(4, 1), (4, 5), (8, 3), (256, 9), (260, 64), (1024, 33), (1, 1), (3, 7), (255, 6)])
def test_median_3x3_network_paths(gpu, oracle, size):
    """r <= 1 takes the min3/med3/max3 network: widths that are multiples of 4 use the 16-byte load + lane-exchange path
    (wave and image edges included), the others the scalar path"""
    w, h = size
    img = I.random_rgba(w, h, 900 + w + h)
    img[:, : w // 2] = (img[:, : w // 2] // 32) * 32  # ties
    mask = (np.random.default_rng(3).random((h, w)) < 0.5).astype(np.uint8)
    assert_same(gpu.median(img, 1), oracle.median(img, 1), 0, f"median3 {size}")
    assert_same(gpu.median(img, 1, mask), oracle.median(img, 1, mask), 0, f"median3 {size} masked")


def test_median_beyond_device_radius_returns_none(gpu):
    assert gpu.r.median_rgba(I.random_rgba(32, 32, 1), 128) is None  # ref: median_rgba -> None, renderer.rs:945


@pytest.mark.parametrize("radius,size", [(25, (90, 70)), (31, (64, 40)), (40, (33, 130)), (100, (48, 37))])
def test_median_large_radius_sliding_histogram_bitexact(gpu, oracle, radius, size):
    """median_core has no radius cap (noise.rs:357-410): beyond the tile kernels a sliding-histogram kernel takes over"""
    img = I.random_rgba(size[0], size[1], radius)
    img[::3, ::5] = 255
    img[1::4, 2::7] = 0
    mask = (I.random_rgba(size[0], size[1], radius + 1)[..., 0] > 60).astype(np.uint8) * 255
    assert np.array_equal(gpu.r.median_core(img, radius, None), oracle.median(img, radius))
    assert np.array_equal(gpu.r.median_core(img, radius, mask), oracle.median(img, radius, mask))


@pytest.mark.parametrize("block", [0, 1, 2, 3, 5, 8, 64, 1000])
@pytest.mark.parametrize("size", [(203, 99), (204, 99), (256, 64), (4, 1), (1028, 7)])
def test_pixelate_bitexact(gpu, oracle, block, size):
    """widths that are multiples of four take the four-pixels-per-lane kernel (no selection), the others and every masked call the one-pixel kernel"""
    w, h = size
    img = I.random_rgba(w, h, block)
    mask = (np.random.default_rng(3).random((h, w)) < 0.5).astype(np.uint8)
    assert_same(gpu.pixelate(img, block), oracle.pixelate(img, block), 0, f"pixelate {block} {w}x{h}")
    assert_same(gpu.pixelate(img, block, mask), oracle.pixelate(img, block, mask), 0, f"pixelate {block} {w}x{h} masked")


# ------------------------------------------------------------------ pointwise bank
def _lut_rgba(seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (4, 256), dtype=np.uint8)


ADJ_CASES = [
    ("invert", []), ("invert_alpha", []), ("sepia", []), ("brightness_contrast", [30.0, 20.0]),
    ("brightness_contrast", [-55.5, -80.0]), ("hsl", [30.0, -20.0, 10.0]), ("hsl", [-170.0, 60.0, -35.0]),
    ("hsl", [0.0, 0.0, 0.0]), ("exposure", [1.0]), ("exposure", [-2.3]), ("highlights_shadows", [30.0, -20.0]),
    ("temperature_tint", [30.0, 10.0]), ("threshold", [128.0]), ("posterize", [4.0]), ("posterize", [1.0]),
    ("color_balance", [10.0, 0.0, -10.0, 5.0, -5.0, 2.0, -10.0, 0.0, 10.0]), ("black_and_white", [30.0, 59.0, 11.0]),
    ("vibrance", [50.0]), ("vibrance", [-70.0]), ("desaturate", []),
]


@pytest.mark.parametrize("op,params", ADJ_CASES, ids=[f"{o}-{i}" for i, (o, _) in enumerate(ADJ_CASES)])
@pytest.mark.parametrize("sparse", [0, 1, 2])
def test_adjust_bank_bitexact(gpu, oracle, op, params, sparse):
    w, h = 197, 141
    img = sparse_alpha_image(w, h, 300 + len(params))
    mask = (np.random.default_rng(4).random((h, w)) < 0.8).astype(np.uint8) * 255
    assert_same(gpu.adjust(img, op, params, sparse=sparse), oracle.adjust(img, op, params, sparse=sparse), 0, f"{op} sparse={sparse}")
    assert_same(gpu.adjust(img, op, params, mask=mask, sparse=sparse), oracle.adjust(img, op, params, mask=mask, sparse=sparse), 0,
                f"{op} masked sparse={sparse}")


def test_adjust_all_rgb_triples_hsl(gpu, oracle):
    """HSL / vibrance over a dense colour lattice (every branch of rgb_to_hsl / hue_to_rgb)"""
    v = np.arange(0, 256, 3, dtype=np.uint8)
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    n = r.size
    w = 512
    h = (n + w - 1) // w
    img = np.zeros((h * w, 4), np.uint8)
    img[:n, 0], img[:n, 1], img[:n, 2] = r.ravel(), g.ravel(), b.ravel()
    img[:, 3] = 255
    img = img.reshape(h, w, 4)
    for op, params in (("hsl", [47.0, 35.0, -12.0]), ("hsl", [-90.0, -100.0, 0.0]), ("vibrance", [80.0])):
        assert_same(gpu.adjust(img, op, params), oracle.adjust(img, op, params), 0, f"{op} lattice")
    assert_same(gpu.rhai_adjust(img, "hsl", [47.0, 35.0, -12.0]), oracle.rhai_adjust(img, "hsl", [47.0, 35.0, -12.0]), 0, "rhai hsl lattice")


def test_lut_ops(gpu, oracle):
    img = sparse_alpha_image(130, 70, 8)
    luts = _lut_rgba(1)
    for sparse in (0, 1, 2):
        assert_same(gpu.adjust(img, "lut_rgba", lut=luts, sparse=sparse), oracle.adjust(img, "lut_rgba", lut=luts, sparse=sparse), 0, "lut_rgba")
    gm = np.random.default_rng(2).integers(0, 256, (256, 4), dtype=np.uint8)
    assert_same(gpu.adjust(img, "gradient_map", lut=gm), oracle.adjust(img, "gradient_map", lut=gm), 0, "gradient_map")
    assert_same(gpu.auto_levels(img), oracle.auto_levels(img), 0, "auto_levels")
    narrow = (img // 3 + 40).astype(np.uint8)
    narrow[..., 3] = img[..., 3]
    mask = np.zeros((70, 130), np.uint8)
    mask[10:60, 20:100] = 1
    assert_same(gpu.auto_levels(narrow, mask), oracle.auto_levels(narrow, mask), 0, "auto_levels masked")
    assert_same(gpu.levels(img, 20.0, 235.0, 1.2, 10.0, 240.0), oracle.levels(img, 20.0, 235.0, 1.2, 10.0, 240.0), 0, "levels")


def test_lut_builders_match_oracle(gpu):
    r = gpu.r
    for args in ((20.0, 235.0, 1.2, 0.0, 255.0), (0.0, 255.0, 1.0, 0.0, 255.0), (100.0, 90.0, 0.001, 255.0, 0.0), (5.5, 200.25, 3.7, 12.0, 199.0)):
        assert np.array_equal(r.build_levels_lut(*args), O.levels_lut(*args)), args
    for pts in ([(0, 0), (255, 255)], [(0, 0), (64, 40), (128, 160), (255, 255)], [(0, 255), (100, 100), (100.0000001, 50), (255, 0)],
                [(10, 20)], [(0, 0), (50, 200), (60, 10), (255, 255)]):
        assert np.array_equal(r.build_curves_lut(pts), O.curves_lut(pts)), pts


RHAI_CASES = [("invert", []), ("desaturate", []), ("sepia", []), ("sepia_strength", [0.4]), ("sepia_strength", [7.0]),
              ("brightness_contrast", [20.0, 10.0]), ("hsl", [30.0, -20.0, 10.0]), ("hsl", [-400.0, 150.0, 90.0]),
              ("exposure", [0.8]), ("levels", [20.0, 230.0, 0.8])]


@pytest.mark.parametrize("op,params", RHAI_CASES, ids=[f"{o}-{i}" for i, (o, _) in enumerate(RHAI_CASES)])
def test_rhai_flavour_bitexact(gpu, oracle, op, params):
    img = sparse_alpha_image(150, 70, 17)
    p = list(params)
    if op == "sepia_strength":  # the host clamps in f64 before the cast (scripting.rs:923)
        exp = oracle.rhai_adjust(img, op, [min(max(p[0], 0.0), 1.0)])
    else:
        exp = oracle.rhai_adjust(img, op, p)
    assert_same(gpu.rhai_adjust(img, op, p), exp, 0, f"rhai {op}")


# ------------------------------------------------------------------ TiledImage rule
@pytest.mark.parametrize("size", [(64, 64), (65, 63), (200, 130), (513, 258)])
def test_tiled_roundtrip_and_chunk_keys(gpu, size):
    w, h = size
    img = sparse_alpha_image(w, h, w * 3 + h)
    assert_same(gpu.r.tiled_roundtrip(img), O.tiled_roundtrip(img), 0, "tiled roundtrip")
    assert np.array_equal(gpu.r.chunk_populated(img), O.chunk_populated(img))


# ------------------------------------------------------------------ warp
def test_warp_displacement_random_field(gpu, oracle):
    w, h = 257, 190
    img = sparse_alpha_image(w, h, 31)
    rng = np.random.default_rng(6)
    disp = (rng.random((h, w, 2)).astype(np.float32) - np.float32(0.5)) * np.float32(40.0)
    disp[5, 5] = [1e9, 0]       # far outside -> transparent
    disp[6, 6] = [np.nan, 0.0]  # NaN -> `as i32` == 0 path
    disp[7, 7] = [7.0 + 1.0, 0.0]  # x0 == -1: blends with the transparent left texel (transform.rs:1310)
    assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, "random field")
    zero = np.zeros((h, w, 2), np.float32)
    assert_same(gpu.warp_displacement(img, zero), img, 0, "identity field (transform_ops.rs:125)")


@pytest.mark.parametrize("size", [(1, 1), (1, 9), (2, 2), (2, 70), (3, 5), (5, 4), (64, 3), (67, 66), (130, 2)])
def test_warp_samples_on_and_around_every_border(gpu, oracle, size):
    """k_warp.hip:bilinear_fetch — a row's two texels are one 8-byte load from clamp(x0, 0, w - 2) (w >= 2; four dword loads for a one-pixel-wide source),
    texels outside the source are masked to 0 afterwards and the border picks the half of the pair that exists.  Every output pixel is sent to a source
    coordinate on, just inside, just outside and far outside each border (x0 / y0 = -2, -1, 0, w - 2, w - 1, w and fractional neighbours), in waves
    that mix interior and border lanes and in waves that are all border (transform.rs:1288-1345)."""
    w, h = size
    img = I.random_rgba(w, h, 900 + 7 * w + h)
    img[..., 3] = np.maximum(img[..., 3], 1)
    rng = np.random.default_rng(w * 131 + h)
    xs = np.array([-2.5, -2.0, -1.5, -1.0, -0.75, -0.5, 0.0, 0.25, w - 2.0, w - 1.5, w - 1.0, w - 0.5, w - 0.001, w, w + 0.5, w + 7.0], np.float32)
    ys = np.array([-2.5, -2.0, -1.5, -1.0, -0.75, -0.5, 0.0, 0.25, h - 2.0, h - 1.5, h - 1.0, h - 0.5, h - 0.001, h, h + 0.5, h + 7.0], np.float32)
    gx, gy = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    for trial in range(3):
        tx = xs[rng.integers(0, len(xs), size=(h, w))]
        ty = ys[rng.integers(0, len(ys), size=(h, w))]
        if trial == 1: ty = gy + np.float32(0.25)          # only x crosses borders
        if trial == 2: tx = gx + np.float32(0.5)           # only y crosses borders
        disp = np.stack([gx - tx, gy - ty], axis=-1).astype(np.float32)   # sample coordinate = x - dx = tx
        assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, f"{size} trial {trial}")
    # a wider field than the source (transform.rs takes the source's size for the bounds, the field's for the output)
    fw, fh = w + 3, h + 2
    disp = (rng.random((fh, fw, 2)).astype(np.float32) - np.float32(0.5)) * np.float32(2 * max(w, h) + 4)
    assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, f"{size} wide random field")
    if w >= 2 and h >= 2:
        orig, deformed = I.jittered_mesh(2, 2, w, h, seed=w + h)
        deformed = deformed + np.float32(1.25) * np.array([w, h], np.float32) * (rng.random(deformed.shape).astype(np.float32) - np.float32(0.5))  # interior AND border points move: samples leave the image
        assert_same(gpu.warp_mesh(img, orig, deformed.astype(np.float32), 2, 2), oracle.warp_mesh(img, orig, deformed.astype(np.float32), 2, 2), 0, f"{size} mesh")


def test_warp_nan_and_infinite_coordinates_inside_interior_waves(gpu, oracle):
    """A NaN / infinite field entry among lanes that all sample strictly inside the source (the wave-uniform fast path of bilinear_fetch): `NaN as i32` is
    texel 0 with NaN weights, every channel `NaN as u8` = 0 (transform.rs:1300-1345); infinities saturate and land outside."""
    w, h = 256, 24
    img = I.random_rgba(w, h, 77)
    img[..., 3] = 255
    disp = np.full((h, w, 2), 0.25, np.float32)           # every sample 0.25 px up-left: interior everywhere but row 0 / column 0
    disp[:, 0, 0] = -0.5; disp[0, :, 1] = -0.5             # ... which sample inside as well
    for (y, x, v) in ((10, 100, (np.nan, 0.25)), (11, 101, (0.25, np.nan)), (12, 130, (np.nan, np.nan)), (13, 64, (np.inf, 0.25)), (14, 191, (0.25, -np.inf)),
                      (5, 5, (3.0e9, 0.25)), (6, 70, (-3.0e9, -3.0e9))):
        disp[y, x] = v
    assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, "NaN / inf entries in interior waves")


def test_warp_source_size_differs_from_field(gpu, oracle):
    img = I.random_rgba(90, 60, 4)
    disp = (np.random.default_rng(7).random((80, 120, 2)).astype(np.float32) - np.float32(0.5)) * np.float32(30.0)
    assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, "src != field size")


def test_warp_source_cache_contract(gpu, oracle):
    """liquify.rs:166-176: the source is uploaded once and reused until invalidate_source"""
    from paintfe_amd import _lib as L
    src = I.random_rgba(120, 90, 5)
    rng = np.random.default_rng(9)
    gpu.r.warp_set_source(src)
    for k in range(3):
        disp = rng.uniform(-9, 9, size=(90, 120, 2)).astype(np.float32)
        assert np.array_equal(gpu.r.warp_displacement_cached(disp, 120, 90), oracle.warp_displacement(src, disp))
    gpu.r.warp_invalidate_source()
    with pytest.raises(L.PfxError):
        gpu.r.warp_displacement_cached(disp, 120, 90)


@pytest.mark.parametrize("grid", [(2, 2), (6, 6), (1, 1), (9, 4), (63, 40)])  # (63,40): > 2048 points, non-LDS path
def test_mesh_warp_and_field(gpu, grid):
    cols, rows = grid
    w, h = 300, 170
    img = I.random_rgba(w, h, cols * 10 + rows)
    orig, deformed = I.jittered_mesh(cols, rows, w, h, seed=cols + rows)
    assert_same(gpu.warp_mesh(img, orig, deformed, cols, rows), O.warp_mesh_catmull_rom(img, orig, deformed, cols, rows), 0, "fused mesh warp")
    f_gpu = gpu.r.generate_displacement(deformed, cols, rows, w, h, original_points=orig)
    assert np.array_equal(f_gpu.view(np.uint32), O.mesh_displacement(orig, deformed, cols, rows, w, h).view(np.uint32)), "field bits"
    f_fast = gpu.r.generate_displacement(deformed, cols, rows, w, h)
    assert np.array_equal(f_fast.view(np.uint32), O.mesh_displacement_fast(deformed, cols, rows, w, h).view(np.uint32)), "fast field bits"


def test_displacement_brushes_host(gpu):
    for mode in range(5):
        a = np.zeros((64, 80, 2), np.float32)
        b = np.zeros((64, 80, 2), np.float32)
        gpu.r.displacement_brush(a, mode, 30.5, 20.25, 3.0, -2.0, 14.0, 0.7)
        O.displacement_brush(b, mode, 30.5, 20.25, 3.0, -2.0, 14.0, 0.7)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), mode


def test_displacement_brushes_spread_over_a_large_field(gpu):
    """dabs far apart on a large field take the launch over the chunks their boxes touch (pfx_displacement_brushes_dev: less than half of the common bounding
    box) — the result must be the plain launch's (a compact batch of the same dabs, one call each) and the oracle's; untouched field keeps its bits (-0.0 included)"""
    w, h = 1700, 1300
    rng = np.random.default_rng(21)
    centres = [(90.0, 80.0), (1600.0, 1210.0), (850.0, 640.0), (1650.0, 70.0), (63.5, 1240.0)]
    dabs = []
    for k in range(30):
        cx, cy = centres[k % len(centres)]
        dabs.append((int(rng.integers(0, 5)), cx + float(rng.uniform(-30, 30)), cy + float(rng.uniform(-30, 30)), float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)),
                     float(rng.uniform(5, 70)), float(rng.uniform(0.1, 1.0))))
    start = np.zeros((h, w, 2), np.float32)
    start[::7, ::5] = -0.0
    start[300:310, 400:420] = 3.25
    ref = start.copy()
    for d in dabs:
        O.displacement_brush(ref, *d)
    dev = gpu.r.dev_alloc(w * h * 8)
    dev2 = gpu.r.dev_alloc(w * h * 8)
    try:
        gpu.r.dev_upload(dev, start)
        gpu.r.displacement_brushes_dev(dev, w, h, dabs)            # one batch: spread -> chunk launch
        got = gpu.r.dev_download(dev, (h, w, 2), np.float32)
        gpu.r.dev_upload(dev2, start)
        for d in dabs:                                             # one dab per call: compact boxes -> plain launch
            gpu.r.displacement_brushes_dev(dev2, w, h, [d])
        one_by_one = gpu.r.dev_download(dev2, (h, w, 2), np.float32)
        assert np.array_equal(got.view(np.uint32), one_by_one.view(np.uint32))
        assert np.allclose(got, ref, rtol=2e-6, atol=2e-6), float(np.abs(got - ref).max())
        untouched = ref.view(np.uint32) == start.view(np.uint32)
        assert np.array_equal(got.view(np.uint32)[untouched], start.view(np.uint32)[untouched])
    finally:
        gpu.r.dev_free(dev)
        gpu.r.dev_free(dev2)


def test_displacement_brushes_on_a_device_field(gpu):
    """a Liquify stroke (mixed dabs, overlapping, partly off-canvas, integer-radius edge) accumulated on the device, then warped"""
    w, h = 300, 200
    rng = np.random.default_rng(12)
    dabs = [(int(rng.integers(0, 5)), float(rng.uniform(-20, w + 20)), float(rng.uniform(-20, h + 20)), float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8)),
             float(rng.uniform(0.2, 60)), float(rng.uniform(0.1, 1.0))) for _ in range(40)]
    dabs += [(0, 16.0, 16.0, 3.0, 0.0, 10.0, 0.8), (1, 100.0, 100.0, 0.0, 0.0, 25.0, 1.0), (0, 5000.0, 5000.0, 1.0, 1.0, 10.0, 1.0)]
    ref = np.zeros((h, w, 2), np.float32)
    for d in dabs:
        O.displacement_brush(ref, *d)
    dev = gpu.r.dev_alloc(w * h * 8)
    try:
        gpu.r.dev_upload(dev, np.zeros((h, w, 2), np.float32))
        gpu.r.displacement_brushes_dev(dev, w, h, dabs[:17])      # two calls: the field persists between dab batches
        gpu.r.displacement_brushes_dev(dev, w, h, dabs[17:])
        got = gpu.r.dev_download(dev, (h, w, 2), np.float32)
        # weights use a device exp(): last-ulp differences per dab, accumulated over up to 43 overlapping dabs
        assert np.allclose(got, ref, rtol=2e-6, atol=2e-6), float(np.abs(got - ref).max())
        assert (got.view(np.uint32) == ref.view(np.uint32)).mean() > 0.98
        img = I.random_rgba(w, h, 5)
        src = gpu.r.dev_alloc(w * h * 4)
        dst = gpu.r.dev_alloc(w * h * 4)
        try:
            gpu.r.dev_upload(src, img)
            gpu.r.warp_displacement_dev(src, w, h, dev, w, h, dst)
            out = gpu.r.dev_download(dst, (h, w, 4), np.uint8)
        finally:
            gpu.r.dev_free(src)
            gpu.r.dev_free(dst)
        d = np.abs(out.astype(int) - O.warp_displacement(img, ref).astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3
    finally:
        gpu.r.dev_free(dev)


# ------------------------------------------------------------------ brush
@pytest.mark.parametrize("mode,eraser,aa", [(0, False, True), (0, False, False), (0, True, True), (1, False, True), (2, False, False), (3, False, True)])
def test_brush_random_strokes(gpu, oracle, mode, eraser, aa):
    w, h = 220, 140
    rng = np.random.default_rng(50 + mode)
    target = I.random_rgba(w, h, 60) if mode else np.zeros((h, w, 4), np.uint8)
    sel = (rng.random((h, w)) < 0.9).astype(np.uint8) * 255
    pts = [(float(x), float(y)) for x, y in zip(rng.random(60) * (w + 40) - 20, rng.random(60) * (h + 40) - 20)]
    brush = dict(size=17.5, hardness=0.6, anti_aliased=aa, color=(0.9, 0.3, 0.1, 0.8), flow=0.85, is_eraser=eraser, mode=mode)
    assert_same(gpu.brush_stamps(target, brush, pts), oracle.brush_stamps(target, brush, pts), 0, "stamps")
    assert_same(gpu.brush_stamps(target, brush, pts, sel), oracle.brush_stamps(target, brush, pts, sel), 0, "stamps + selection")
    assert_same(gpu.brush_line(target, brush, (-10.0, 5.0), (250.0, 130.0)), oracle.brush_line(target, brush, (-10.0, 5.0), (250.0, 130.0)), 0, "line")


def _tip_source(n=96):
    """an asymmetric soft blob with a hole: rotation and hardness are visible"""
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32)
    a = np.clip(1.0 - np.hypot(xx - n * 0.5, (yy - n * 0.35) * 1.6) / (n * 0.42), 0, 1)
    a *= (np.hypot(xx - n * 0.62, yy - n * 0.3) > n * 0.08)
    return (a * 255).astype(np.uint8)


@pytest.mark.parametrize("case", ["scatter", "jitter", "scatter_jitter_eraser", "dodge_scatter", "tip", "tip_rotated", "tip_random_rotation",
                                  "tip_eraser_scatter", "tip_jitter_selection", "tip_upscaled"])
def test_brush_dynamics(gpu, oracle, case):
    """scatter, hue / brightness jitter and image tips (brush_render.rs:148-256, 404-760): host prologue per stamp + the stamp kernel,
    bit-exact with the oracle for strokes of overlapping stamps, including off-canvas and edge-clipped ones"""
    w, h = 230, 150
    rng = np.random.default_rng(sum(map(ord, case)))
    target = I.random_rgba(w, h, 9) if "dodge" in case else np.zeros((h, w, 4), np.uint8)
    if "eraser" in case:  # eraser strokes accumulate a strength mask in an initially empty preview (brush_render.rs:345-356)
        target = np.zeros((h, w, 4), np.uint8)
        target[:, : w // 2, 3] = 60  # part of it already holds a weaker mask
    pts = [(float(rng.uniform(-15, w + 15)), float(rng.uniform(-15, h + 15))) for _ in range(60)]
    pts += [(0.0, 0.0), (w - 1.0, h - 1.0), (w / 2, h / 2), (w / 2 + 0.5, h / 2 + 0.25)]
    brush = dict(size=26.0, hardness=0.6, anti_aliased=True, color=(0.85, 0.3, 0.1, 0.9), flow=0.8, is_eraser="eraser" in case, mode=1 if "dodge" in case else 0)
    dyn = dict(stamp_counter=int(rng.integers(0, 2 ** 31)))
    sel = None
    if "scatter" in case:
        dyn["scatter"] = 0.7
    if "jitter" in case:
        dyn["hue_jitter"], dyn["brightness_jitter"] = 0.8, 0.5
    if case.startswith("tip"):
        size = 41.0 if case == "tip_upscaled" else 26.0
        brush["size"] = size
        src = _tip_source(24 if case == "tip_upscaled" else 96)
        tip_o = O.brush_tip_rescale(src, size, brush["hardness"])
        tip_g = gpu.r.brush_tip_rescale(src, size, brush["hardness"])
        assert np.array_equal(tip_o, tip_g), "rebuild_tip_mask"
        dyn["tip_mask"] = tip_g
        if case == "tip_rotated":
            dyn["tip_rotation"] = 33.0
        if case == "tip_random_rotation":
            dyn["tip_random_rotation"], dyn["tip_rotation_range"] = True, (-90.0, 250.0)
        if "selection" in case:
            sel = (rng.random((h, w)) < 0.7).astype(np.uint8) * 255
    got = gpu.brush_stamps(target, brush, pts, sel, dyn)
    ref = oracle.brush_stamps(target, brush, pts, sel, dyn)
    assert_same(got, ref, 0, f"brush dynamics {case}")
    assert not np.array_equal(ref, target)
    if case == "scatter":  # scatter must actually move stamps: differs from the plain stroke
        assert not np.array_equal(ref, oracle.brush_stamps(target, brush, pts, sel))



@pytest.mark.parametrize("case", ["paint", "paint_noaa", "eraser", "dodge", "sponge_selection", "scatter_jitter", "tip_random_rotation", "tip_eraser"])
@pytest.mark.parametrize("size", [(333, 190), (64, 64), (130, 65)])
def test_brush_long_strokes_binned_by_chunk(gpu, oracle, case, size):
    """Strokes of more than 64 stamps are dealt to the 64 x 64 chunks their boxes touch on the host (pfx_brush_stamps_ex_dev) and the kernel walks per-chunk
    lists: the result must be the serial stamp loop's (oracle) and the bounding-box kernel's (pfx_tune "brush_binning" = 0) bit for bit — order-dependent modes
    (max-alpha ties, eraser, Dodge / Sponge read-modify-write), stamps off the canvas and across chunk borders, canvases that are not whole chunks, image tips
    whose rotated boxes are larger, scatter that moves stamps into other chunks."""
    w, h = size
    rng = np.random.default_rng(sum(map(ord, case)) + w)
    n = 420
    t = np.linspace(0, 9 * np.pi, n)
    pts = np.stack([w * (0.5 + 0.62 * np.cos(t * 0.7) * np.sin(t * 0.13 + 0.4)), h * (0.5 + 0.62 * np.sin(t * 0.9))], axis=1).astype(np.float32)
    pts[::37] += rng.uniform(-40, 40, size=pts[::37].shape).astype(np.float32)   # jumps: lists are not contiguous runs
    target = I.random_rgba(w, h, 9) if ("dodge" in case or "sponge" in case) else np.zeros((h, w, 4), np.uint8)
    if "eraser" in case:
        target[:, : w // 2, 3] = 60
    brush = dict(size=23.0, hardness=0.6, anti_aliased=case != "paint_noaa", color=(0.85, 0.3, 0.1, 0.9), flow=0.8, is_eraser="eraser" in case,
                 mode=1 if "dodge" in case else (3 if "sponge" in case else 0))
    dyn, sel = None, None
    if "selection" in case:
        sel = (rng.random((h, w)) < 0.8).astype(np.uint8) * 255
    if case == "scatter_jitter":
        dyn = dict(stamp_counter=77, scatter=0.9, hue_jitter=0.6, brightness_jitter=0.4)
    if case.startswith("tip"):
        tip = gpu.r.brush_tip_rescale(_tip_source(96), 31.0, 0.6)
        brush["size"] = 31.0
        dyn = dict(stamp_counter=5, tip_mask=tip)
        if "rotation" in case:
            dyn["tip_random_rotation"], dyn["tip_rotation_range"] = True, (-90.0, 250.0)
    ref = oracle.brush_stamps(target, brush, pts, sel, dyn)
    assert not np.array_equal(ref, target)
    for binning in (1, 0):
        gpu.r.tune("brush_binning", binning)
        try:
            assert_same(gpu.brush_stamps(target, brush, pts, sel, dyn), ref, 0, f"long stroke {case} {w}x{h} binning={binning}")
        finally:
            gpu.r.tune("brush_binning", 1)


def test_brush_commit(gpu):
    w, h = 100, 80
    layer = I.random_rgba(w, h, 70)
    preview = np.zeros((h, w, 4), np.uint8)
    b = dict(size=30.0, hardness=0.5, anti_aliased=True, color=(0.2, 0.4, 0.9, 1.0))
    preview = O.brush_stamp(preview, O.make_brush(**b), 50.0, 40.0)
    sel = np.ones((h, w), np.uint8)
    sel[:, :40] = 0
    for mode in (0, 1, 8, 13, 14):
        assert_same(gpu.r.brush_commit(layer, preview, mode, False, sel), O.brush_commit(layer, preview, mode, sel), 0, f"commit mode {mode}")
    assert_same(gpu.r.brush_commit(layer, preview, 0, True, None), O.eraser_commit(layer, preview, None), 0, "eraser commit")


# ------------------------------------------------------------------ error behaviour (dst untouched, status codes)
def test_errors_leave_dst_untouched(gpu):
    import ctypes as C
    from paintfe_amd import _lib
    lib = _lib.load()
    src = I.random_rgba(16, 16, 1)
    dst = np.full_like(src, 0xAB)
    st = lib.pfx_median_rgba(gpu.r.handle, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), C.c_uint32(16), C.c_uint32(16), C.c_uint32(1000))
    assert st == _lib.ERR_UNSUPPORTED and (dst == 0xAB).all()
    st = lib.pfx_blur_rgba(gpu.r.handle, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p), C.c_uint32(0), C.c_uint32(16), C.c_float(2.0))
    assert st == _lib.ERR_INVALID and (dst == 0xAB).all()
    st = lib.pfx_blur_rgba(gpu.r.handle, None, dst.ctypes.data_as(C.c_void_p), C.c_uint32(16), C.c_uint32(16), C.c_float(2.0))
    assert st == _lib.ERR_INVALID and b"null" in lib.pfx_last_error(gpu.r.handle)


# ------------------------------------------------------------------ degenerate image sizes through every kernel family
@pytest.mark.parametrize("size", [(1, 1), (1, 37), (41, 1), (3, 2), (65, 1), (1, 130)])
def test_degenerate_sizes(gpu, oracle, size):
    w, h = size
    img = I.random_rgba(w, h, 500 + w * 3 + h)
    top = I.random_rgba(w, h, 600 + w * 3 + h)
    mask = (np.random.default_rng(w * 7 + h).random((h, w)) < 0.5).astype(np.uint8) * 255
    layers = [dict(pixels=img), dict(pixels=top, mode=8, opacity=0.6, mask=mask), dict(kind=2, adj=[10.0, 20.0], opacity=0.7)]
    assert_same(gpu.composite(layers, w, h), oracle.composite(layers, w, h), 0, f"composite {size}")
    gpu.r.set_exact(True)
    try:
        for sigma in (0.4, 3.0, 20.0):
            assert_same(gpu.gaussian_blur(img, sigma), oracle.gaussian_blur(img, sigma), 0, f"gaussian {size} sigma {sigma}")
            assert_same(gpu.gaussian_blur(img, sigma, mask), oracle.gaussian_blur(img, sigma, mask), 0, f"gaussian {size} sigma {sigma} masked")
        assert_same(gpu.sharpen(img, 1.5, 2.0), oracle.sharpen(img, 1.5, 2.0), 0, f"sharpen {size}")
    finally:
        gpu.r.set_exact(False)
    for radius in (1.0, 6.0):
        assert_same(gpu.box_blur(img, radius, mask), oracle.box_blur(img, radius, mask), 0, f"box {size}")
    for radius in (1, 2, 3):
        assert_same(gpu.median(img, radius), oracle.median(img, radius), 0, f"median {size} r={radius}")
    assert_same(gpu.pixelate(img, 3), oracle.pixelate(img, 3), 0, f"pixelate {size}")
    assert_same(gpu.bokeh_blur(img, 2.5), oracle.bokeh_blur(img, 2.5), 0, f"bokeh {size}")
    assert_same(gpu.motion_blur(img, 30.0, 5.0), oracle.motion_blur(img, 30.0, 5.0), 0, f"motion {size}")
    for sparse in (0, 1, 2):
        assert_same(gpu.adjust(img, "hsl", [30.0, -20.0, 10.0], None, mask, sparse), oracle.adjust(img, "hsl", [30.0, -20.0, 10.0], None, mask, sparse), 0, f"hsl {size}")
    assert_same(gpu.rhai_adjust(img, "sepia"), oracle.rhai_adjust(img, "sepia"), 0, f"rhai sepia {size}")
    disp = (np.random.default_rng(5).random((h, w, 2)).astype(np.float32) - 0.5) * 6
    assert_same(gpu.warp_displacement(img, disp), oracle.warp_displacement(img, disp), 0, f"warp {size}")
    orig, deformed = I.jittered_mesh(2, 2, w, h)
    assert_same(gpu.warp_mesh(img, orig, deformed, 2, 2), oracle.warp_mesh(img, orig, deformed, 2, 2), 0, f"mesh {size}")
    assert_same(gpu.resize(img, 7, 5, "lanczos3"), oracle.resize(img, 7, 5, "lanczos3"), 0, f"resize {size}")
    assert_same(gpu.r.tiled_roundtrip(img), O.tiled_roundtrip(img), 0, f"tiled {size}")


# ------------------------------------------------------------------ tool preview layer in the compositor (canvas_state.rs:593-658)
def _preview_stack(w, h):
    bg = I.random_rgba(w, h, 71)
    bg[..., 3] = 255
    active = sparse_alpha_image(w, h, 72)           # transparent chunks and transparent-but-coloured pixels
    above = sparse_alpha_image(w, h, 73)
    mask = (np.random.default_rng(74).random((h, w)) < 0.3).astype(np.uint8) * 200
    layers = [dict(pixels=bg), dict(pixels=active, mode=8, opacity=0.8, mask=mask), dict(kind=1, adj=[0.4], opacity=0.5),
              dict(pixels=above, mode=2, opacity=0.6)]  # kind 1 = exposure adjustment layer (+0.4 EV at half strength)
    preview = sparse_alpha_image(w, h, 75)
    preview[h // 2:, :, 3] = 255                    # opaque region too
    return layers, preview


def _gpu_composite_preview(gpu, layers, w, h, pv):
    gpu.r.clear_layers()
    info = []
    for i, L in enumerate(layers):
        kind = L.get("kind", 0)
        if kind == 0:
            gpu.r.ensure_layer_texture(i, L["pixels"], generation=1)
            if L.get("mask") is not None:
                gpu.r.set_layer_mask(i, L["mask"])
        info.append((i, L.get("opacity", 1.0), L.get("visible", True), L.get("mode", 0), kind, L.get("adj", ())))
    return gpu.r.composite_preview(w, h, info, pv["pixels"], pv["active_layer"], pv.get("blend_mode", 0), pv.get("is_eraser", False),
                                   pv.get("replaces_layer", False), pv.get("chunk_present"))


@pytest.mark.parametrize("kind", ["normal", "multiply", "overwrite", "xor", "soft_light", "eraser", "replace", "replace_all_chunks", "hidden_active",
                                  "active_is_bottom"])
def test_composite_with_preview_layer(gpu, kind):
    w, h = 200, 150
    layers, preview = _preview_stack(w, h)
    modes = {"normal": 0, "multiply": 1, "overwrite": 14, "xor": 13, "soft_light": 16}
    pv = dict(pixels=preview, active_layer=1, blend_mode=modes.get(kind, 0))
    if kind == "eraser":
        pv["is_eraser"] = True
    if kind.startswith("replace"):
        pv["replaces_layer"] = True
    if kind == "replace_all_chunks":  # the caller's TiledImage may hold fully transparent chunks: there the layer is hidden by the preview
        pv["chunk_present"] = np.ones(((h + 63) // 64, (w + 63) // 64), np.uint8)
    if kind == "hidden_active":
        layers[1]["visible"] = False
    if kind == "active_is_bottom":
        pv["active_layer"] = 0
    ref = O.composite(layers, w, h, preview=pv)
    out = _gpu_composite_preview(gpu, layers, w, h, pv)
    assert_same(out, ref, 0, f"preview {kind}")
    if kind in ("normal", "eraser", "replace"):
        assert not np.array_equal(ref, O.composite(layers, w, h)), "the preview must change the picture"


# ------------------------------------------------------------------ effects built from the same kernels (N3)
@pytest.mark.parametrize("amount,radius", [(1.0, 1.0), (2.5, 3.0), (0.0, 2.0), (-0.5, 0.7)])
def test_sharpen_vs_oracle(gpu, oracle, amount, radius):
    img = I.random_rgba(211, 97, 5)
    mask = (np.random.default_rng(8).random((97, 211)) < 0.6).astype(np.uint8)
    assert_same(gpu.sharpen(img, amount, radius), oracle.sharpen(img, amount, radius), 0, "sharpen")
    assert_same(gpu.sharpen(img, amount, radius, mask), oracle.sharpen(img, amount, radius, mask), 0, "sharpen masked")


@pytest.mark.parametrize("radius,intensity", [(3.0, 0.5), (8.0, 1.7), (0.0, 1.0)])
def test_glow_vs_oracle(gpu, oracle, radius, intensity):
    img = I.random_rgba(140, 120, 6)
    assert_same(gpu.glow(img, radius, intensity), oracle.glow(img, radius, intensity), 0, "glow")


@pytest.mark.parametrize("radius", [0.3, 0.5, 1.0, 5.0, 12.7])
def test_bokeh_bitexact(gpu, oracle, radius):
    img = I.random_rgba(150, 90, int(radius * 10))
    mask = (np.random.default_rng(9).random((90, 150)) < 0.5).astype(np.uint8)
    assert_same(gpu.bokeh_blur(img, radius), oracle.bokeh_blur(img, radius), 0, f"bokeh r={radius}")
    assert_same(gpu.bokeh_blur(img, radius, mask), oracle.bokeh_blur(img, radius, mask), 0, f"bokeh r={radius} masked")


@pytest.mark.parametrize("angle,distance", [(45.0, 10.0), (0.0, 1.0), (90.0, 25.5), (-133.0, 7.2), (30.0, 0.5)])
def test_motion_blur_bitexact(gpu, oracle, angle, distance):
    img = I.random_rgba(173, 88, 12)
    assert_same(gpu.motion_blur(img, angle, distance), oracle.motion_blur(img, angle, distance), 0, f"motion {angle} {distance}")


# ------------------------------------------------------------------ resamplers: fused kernel vs the two-pass path and the oracle
@pytest.mark.parametrize("filt", ["nearest", "bilinear", "bicubic", "lanczos3"])
@pytest.mark.parametrize("src,dst", [((331, 257), (166, 129)), ((200, 150), (431, 322)), ((640, 480), (77, 601)), ((4099, 70), (33, 35)),
                                     ((64, 64), (64, 63)), ((129, 5), (1, 1))])
def test_resize_fused_equals_two_pass_and_oracle(gpu, filt, src, dst):
    """pfx_resize_image: both passes in one kernel (vertical results in LDS) against the two-pass path with its f32 intermediate in HBM
    (pfx_tune resize_two_pass) — the same operations in the same order, so bit-identical — and against the oracle.  (4099 -> 33 columns:
    a 64-column output tile would need more LDS than a block has, the call falls back to two passes by itself.)"""
    (w, h), (nw, nh) = src, dst
    img = I.random_rgba(w, h, 1234 + w + nw)
    fused = gpu.r.resize_image(img, nw, nh, filt)
    gpu.r.tune("resize_two_pass", 1)
    try:
        two = gpu.r.resize_image(img, nw, nh, filt)
    finally:
        gpu.r.tune("resize_two_pass", 0)
    assert np.array_equal(fused, two)
    assert_same(fused, O.resize(img, nw, nh, filt), 0, f"resize {filt} {src}->{dst}")


@pytest.mark.parametrize("radius", [1.0, 2.0, 3.0, 4.0, 5.5, 8.0, 9.0])
@pytest.mark.parametrize("size", [(257, 131), (64, 64), (70, 300), (5, 3)])
def test_box_blur_fused_equals_two_pass_and_oracle(gpu, radius, size):
    """box_blur_core for r <= 4 runs both passes in one kernel (u8 intermediate in LDS): bit-identical to the two-pass path (pfx_tune
    box_two_pass) and to the oracle, with and without a selection mask; larger radii take the two-pass path by themselves"""
    w, h = size
    img = I.random_rgba(w, h, 4242 + w)
    mask = (np.random.default_rng(w * h).random((h, w)) < 0.6).astype(np.uint8) * 255
    for m in (None, mask):
        fused = gpu.box_blur(img, radius, m)
        gpu.r.tune("box_two_pass", 1)
        try:
            two = gpu.box_blur(img, radius, m)
        finally:
            gpu.r.tune("box_two_pass", 0)
        assert np.array_equal(fused, two)
        gpu.r.tune("box_strip", 1)      # the 64 x 64 tile kernel for r <= 4 (the default since round 5 is the strip walk on every radius)
        try:
            tile = gpu.box_blur(img, radius, m)
        finally:
            gpu.r.tune("box_strip", 2)
        assert np.array_equal(fused, tile)
        assert_same(fused, O.box_blur(img, radius, m), 0, f"box blur r={radius} {size}")


@pytest.mark.parametrize("radius", [5.0, 23.0, 24.0, 60.0, 130.0, 700.0])
@pytest.mark.parametrize("size", [(2049, 40), (4100, 9), (1024, 3), (1025, 2), (300, 300), (33, 1), (1, 70)])
def test_box_blur_two_pass_lane_runs(gpu, radius, size):
    """the two-pass kernels: 4 / 8 / 16 columns per lane in the horizontal pass (1024-, 2048- and 4096-pixel row tiles: widths either side of one and
    two tiles), 16 / 32 / 64 / 128 rows per lane in the vertical pass with the window's rows requested eight outputs ahead (heights below one run,
    bands that end inside an 8-row group): the shapes the radius rule picks, and every shape forced on every radius through pfx_tune"""
    w, h = size
    img = I.random_rgba(w, h, 777 + w + int(radius))
    ref = O.box_blur(img, radius)
    assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} (kernel by radius)")
    gpu.r.tune("box_strip", 0)         # radii 5 .. 64 take the fused strip walk by default (test_box_blur_strip_walk); here: the two-pass kernels on every radius
    try:
        _two_pass_shapes(gpu, img, ref, radius, size)
    finally:
        gpu.r.tune("box_strip", 2)


def _two_pass_shapes(gpu, img, ref, radius, size):
    assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} (two-pass shapes by radius)")
    gpu.r.tune("box_prefix_from", 1)   # the prefix-sum horizontal pass (default: radii from 72) on every radius and tile shape
    try:
        assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} prefix-sum horizontal pass")
    finally:
        gpu.r.tune("box_prefix_from", 72)
    for px, py in ((4, 16), (8, 32), (16, 64), (16, 128), (8, 16), (4, 128)):
        gpu.r.tune("box_px", px)
        gpu.r.tune("box_py", py)
        try:
            assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} lane runs {px}/{py}")
        finally:
            gpu.r.tune("box_px", 0)
            gpu.r.tune("box_py", 0)


@pytest.mark.parametrize("radius", [1.0, 2.0, 4.0, 5.0, 6.5, 9.0, 16.0, 33.0, 48.0, 60.0, 64.0])
@pytest.mark.parametrize("size", [(300, 300), (129, 200), (128, 64), (1, 70), (33, 1), (700, 37), (260, 1000), (1100, 130)])
def test_box_blur_strip_walk(gpu, radius, size):
    """radii 5 .. 64: both passes in one kernel, a column-strip walk with the u8 intermediate in an LDS ring (k_stencil.hip: box_strip_kernel).  Bit-identical
    to the oracle and to the two-pass kernels for every way of cutting the image into strips (128 columns: widths either side of one, two, eight and nine
    strips — the XCD grouping) and segments (forced counts: one segment, segments shorter than the window, the launcher's own choice), with and without a
    selection mask; images smaller than the window in either direction"""
    w, h = size
    img = I.random_rgba(w, h, 999 + w + int(radius))
    mask = (np.random.default_rng(w + h).random((h, w)) < 0.5).astype(np.uint8) * 255
    ref, ref_m = O.box_blur(img, radius), O.box_blur(img, radius, mask)
    try:
        for nseg in (0, 1, 2, 7, 40):
            gpu.r.tune("box_strip_nseg", nseg)
            assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} strip walk, {nseg or 'auto'} segments")
        gpu.r.tune("box_strip_nseg", 3)
        assert_same(gpu.box_blur(img, radius, mask), ref_m, 0, f"box blur r={radius} {size} strip walk, masked")
    finally:
        gpu.r.tune("box_strip_nseg", 0)
    gpu.r.tune("box_strip", 0)
    try:
        assert_same(gpu.box_blur(img, radius), ref, 0, f"box blur r={radius} {size} two-pass")
    finally:
        gpu.r.tune("box_strip", 2)


@pytest.mark.parametrize("params", [(30.0, -20.0, float("inf")), (30.0, -20.0, float("-inf")), (float("nan"), 10.0, 5.0), (10.0, float("inf"), 0.0),
                                    (0.0, 0.0, 3.0e38), (720.0, 1.0e30, -1.0e30)])
def test_hsl_with_non_finite_and_huge_parameters(gpu, oracle, params):
    """the HSL / vibrance kernels use a cheaper round-and-pack when the host found every parameter finite; infinities and NaNs must still
    round like Rust's `.round().clamp(0, 255) as u8` (inf -> 255, NaN -> 0), huge finite values saturate"""
    img = I.random_rgba(130, 70, 321)
    assert_same(gpu.adjust(img, "hsl", params), oracle.adjust(img, "hsl", params), 0, f"hsl {params}")
    v = (params[2],) if np.isfinite(params[0]) else (params[0],)
    assert_same(gpu.adjust(img, "vibrance", v), oracle.adjust(img, "vibrance", v), 0, f"vibrance {v}")


@pytest.mark.parametrize("sigma", [0.2, 0.34, 0.5, 1.0, 1.7, 2.5, 3.3, 4.0, 4.2, 5.0, 5.33, 5.5])
@pytest.mark.parametrize("size", [(300, 300), (64, 32), (65, 33), (1, 50), (50, 1), (700, 45), (130, 1200), (1030, 70)])
def test_exact_gaussian_fused_small_radii(gpu, oracle, sigma, size):
    """bit-exact mode, radii 1 .. 16 (sigma <= 5.33): both passes in one kernel with the f32 intermediate in an LDS ring (k_gauss.hip: gauss_fused_exact_kernel) —
    bit-identical to the oracle and to the two-kernel path (pfx_tune "gauss_fused_exact" = 0) on widths either side of one, two, eight and sixteen 64-column
    strips, heights either side of a 32-row block and of a segment, images smaller than the window; radius 17 (sigma 5.5) takes the two kernels by itself.
    This is the Gaussian inside sharpen / glow / drop shadow and the batch pipeline since round 5."""
    w, h = size
    img = I.random_rgba(w, h, 31 + w + int(sigma * 10)) if (w + h) % 3 else I.create_test_gradient(w, h)
    ref = oracle.gaussian_blur(img, sigma)
    gpu.r.set_exact(True)
    try:
        fused = gpu.r.blur_rgba(img, sigma)
        gpu.r.tune("gauss_fused_exact", 0)
        try:
            two = gpu.r.blur_rgba(img, sigma)
        finally:
            gpu.r.tune("gauss_fused_exact", 1)
    finally:
        gpu.r.set_exact(False)
    assert np.array_equal(fused, two), f"sigma {sigma} {size}: fused differs from the two kernels on {int((fused != two).any(-1).sum())} px"
    assert_same(fused, ref, 0, f"exact gaussian sigma {sigma} {size}")


def test_mfma_gaussian_64_column_strips_are_bit_identical_to_32_column_strips(gpu):
    """round 6: up to 8 K blocks (sigma <= 16) the matrix-core Gaussian can run a workgroup of twelve waves on 64-column strips; every output column keeps the K-block
    grouping and accumulation order of the 32-column kernel, so the two must agree bit for bit — whole images, ragged widths (strips that leave the image), bands
    (first_row), all three K-block counts (pfx_tune "gauss_cols64": bit 0 / 1 / 2 = 4 / 6 / 8 K blocks on the 64-column kernel; 6 and 8 K blocks ship on it)"""
    r = gpu.r
    for (w, h) in [(256, 96), (1024, 300), (196, 70), (64, 33), (2052, 180), (388, 515)]:
        img = I.random_rgba(w, h, seed=w + 3 * h)
        a, b, c = (r.dev_alloc(img.nbytes) for _ in range(3))
        try:
            r.dev_upload(a, img)
            for sigma in (1.5, 4.0, 5.3, 6.0, 9.0, 10.5, 11.0, 13.7, 16.0):
                for first_row in (0, 17, 64):
                    r.tune("gauss_cols64", 7)
                    r.gaussian_blur_dev(a, b, w, h, sigma, first_row=first_row)
                    r.tune("gauss_cols64", 0)
                    r.gaussian_blur_dev(a, c, w, h, sigma, first_row=first_row)
                    r.synchronize()
                    x, y = r.dev_download(b, img.shape), r.dev_download(c, img.shape)
                    assert np.array_equal(x, y), (w, h, sigma, first_row, int((x != y).sum()))
        finally:
            r.tune("gauss_cols64", 6)
            for p in (a, b, c):
                r.dev_free(p)


def test_median_5x5_cross_lane_network_matches_the_oracle_and_the_per_lane_network(gpu, oracle):
    """round 6: radius 2 runs a network whose sorted columns are shared ACROSS lanes (wave shifts; a wave row = 62 output lanes + 2 halo lanes = 248 pixels).  Bit-exact
    against the oracle and identical to the per-lane network (pfx_tune "median_xlane" = 0) on widths around the 248-pixel wave tile, unaligned widths (the scalar
    load path), one-pixel-wide / one-row images, heavy ties and selections"""
    r = gpu.r
    rng = np.random.default_rng(77)
    for (w, h) in [(248, 9), (247, 7), (249, 5), (252, 12), (496, 6), (500, 40), (4, 4), (1, 17), (3, 3), (64, 1), (8, 2), (1000, 33), (744, 8), (745, 8)]:
        img = I.random_rgba(w, h, seed=w * 13 + h)
        if (w + h) % 3 == 0:
            img = (img // 64) * 64                                   # heavy ties
        mask = None if (w + h) % 2 else ((rng.random((h, w)) < 0.5).astype(np.uint8) * 255)
        want = oracle.median(img, 2, mask=mask)
        try:
            r.tune("median_xlane", 1)
            got = gpu.median(img, 2, mask=mask)
            r.tune("median_xlane", 2)          # two rows per lane (rows 1 .. 4 of a column sorted once for both)
            got2 = gpu.median(img, 2, mask=mask)
            r.tune("median_xlane", 0)
            old = gpu.median(img, 2, mask=mask)
        finally:
            r.tune("median_xlane", 1)
        assert np.array_equal(got, want), (w, h, mask is not None, int((got != want).any(-1).sum()))
        assert np.array_equal(got2, want), ("two rows", w, h, mask is not None, int((got2 != want).any(-1).sum()))
        assert np.array_equal(old, want), (w, h)


def test_median_7x7_cross_lane_network_matches_the_oracle(gpu, oracle):
    """the 7x7 form of the cross-lane network (pfx_tune "median_xlane" bit 2; the bit-plane select is the shipped r = 3 path): bit-exact on the same shapes"""
    r = gpu.r
    rng = np.random.default_rng(78)
    try:
        r.tune("median_xlane", 5)
        for (w, h) in [(248, 9), (247, 7), (252, 12), (500, 40), (4, 4), (1, 17), (3, 3), (64, 1), (745, 8)]:
            img = I.random_rgba(w, h, seed=w * 17 + h)
            if (w + h) % 3 == 0:
                img = (img // 64) * 64
            mask = None if (w + h) % 2 else ((rng.random((h, w)) < 0.5).astype(np.uint8) * 255)
            got, want = gpu.median(img, 3, mask=mask), oracle.median(img, 3, mask=mask)
            assert np.array_equal(got, want), (w, h, mask is not None, int((got != want).any(-1).sum()))
    finally:
        r.tune("median_xlane", 1)


def test_page_locked_host_buffers_are_ordinary_buffers_to_every_entry_point(gpu):
    """pfx_host_alloc / pfx_host_free: memory for the host-buffer seams that moves at the link's rate; results are those of pageable buffers"""
    r = gpu.r if hasattr(gpu, "r") else gpu
    img = I.random_rgba(203, 117, 5)
    pin_in, pin_out = r.host_alloc(img.shape), r.host_alloc(img.shape)
    try:
        pin_in[...] = img
        want = r.invert_rgba(img)
        got = r.invert_rgba(pin_in, out=pin_out)
        assert got is pin_out and np.array_equal(pin_out, want)
        assert np.array_equal(r.blur_rgba(pin_in, 2.5), r.blur_rgba(img, 2.5))
    finally:
        r.host_free(pin_in)
        r.host_free(pin_out)
    r.host_free(img)   # not page-locked: a no-op
