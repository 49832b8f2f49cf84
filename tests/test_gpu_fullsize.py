"""Parity gate 3 (GPU): the BASELINE.json configurations at their FULL sizes, through the device-resident C ABI.

Where the CPU oracle finishes in seconds the whole frame is compared (HSL at 8K, the 16K mesh warp and liquify warp, the
4K per-image pipeline); the 8K x 32-layer compositor and the sigma-16 Gaussian are compared with the oracle on windows
(per-pixel op: a window of the full-size result equals the op on the window; separable stencil: the same for a window
plus a halo of the kernel radius) and through size-independent properties (a band-by-band run concatenates to the
whole-frame run; a constant image is a fixed point of the blur; identity mesh and integer shifts are exact).

Bars as everywhere: bit-exact for the compositor, HSL and the warps, +-1 LSB for the Gaussian in its default FMA mode and
bit-exact in exact mode."""
import numpy as np
import pytest
import torch  # noqa: F401  -- at collection time, i.e. before any test loads libpfx.so: PyTorch must bring up the HIP runtime
#                              first (it ships its own copy; loaded second it reports no device), as bench.py does

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu

W8K, H8K = 7680, 4320
W16K, H16K = 15360, 8640
W4K, H4K = 3840, 2160


@pytest.fixture(scope="module")
def env():
    import torch
    from paintfe_amd import GpuRenderer
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    r = GpuRenderer(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    yield torch, r, torch.device("cuda", 0)
    torch.cuda.synchronize()
    r.close()


def host(t):
    return t.contiguous().cpu().numpy()


def windows(w, h, ww, wh, seed, n_random=3):
    """the four corners (clamped borders, ragged last chunk row: 4320 = 67.5 chunks) and a few random interior windows"""
    rng = np.random.default_rng(seed)
    out = [(0, 0), (w - ww, 0), (0, h - wh), (w - ww, h - wh)]
    out += [(int(rng.integers(1, w - ww)), int(rng.integers(1, h - wh))) for _ in range(n_random)]
    return out


@pytest.fixture(scope="module")
def stack8k(env):
    import bench
    torch, r, device = env
    stack, modes, opac = bench.synth_stack(torch, device, W8K, H8K, 32, seed=0x5EED0002)
    flat = torch.empty((H8K, W8K, 4), dtype=torch.uint8, device=device)
    info = [(k, float(opac[k]), True, int(modes[k])) for k in range(32)]
    r.flatten_dev([stack[k].data_ptr() for k in range(32)], info, W8K, H8K, flat.data_ptr())
    torch.cuda.synchronize()
    return stack, modes, opac, info, flat


# ------------------------------------------------------------------ config 2: 8K x 32 layers, all 25 blend modes

def test_flatten_8k_32_layers_windows_bitexact(env, stack8k):
    torch, r, device = env
    stack, modes, opac, info, flat = stack8k
    for (x, y) in windows(W8K, H8K, 384, 288, seed=1):
        ref = O.flatten_stack(host(stack[:, y:y + 288, x:x + 384, :]), modes, opac)
        got = host(flat[y:y + 288, x:x + 384, :])
        assert np.array_equal(ref, got), f"window at ({x},{y}): {int((ref != got).any(-1).sum())} px differ"


def test_flatten_8k_bands_concatenate_to_the_frame(env, stack8k):
    """the sharded path's property: flattening chunk-row bands one by one gives the whole-frame result bit for bit"""
    from paintfe_amd import sharding as S
    torch, r, device = env
    stack, modes, opac, info, flat = stack8k
    out = torch.empty_like(flat)
    for (y0, y1) in S.all_bands(H8K, 8):
        band = stack[:, y0:y1].contiguous()
        r.flatten_dev([band[k].data_ptr() for k in range(32)], info, W8K, y1 - y0, out[y0:y1].data_ptr())
        torch.cuda.synchronize()
        del band
    assert torch.equal(out, flat)


def test_flatten_8k_every_mode_is_exercised(stack8k):
    stack, modes, opac, info, flat = stack8k
    assert sorted(set(int(m) for m in modes)) == list(range(25))
    assert int(flat[..., 3].max()) == 255 and int(stack[0, ..., 3].min()) == 255


# ------------------------------------------------------------------ config 1: 8K Gaussian sigma=16 + HSL

@pytest.mark.parametrize("exact", [False, True])
def test_gaussian_8k_sigma16_windows(env, stack8k, exact):
    torch, r, device = env
    flat = stack8k[4]
    sigma = 16.0
    radius = int(np.ceil(np.float32(sigma) * np.float32(3.0)))
    blurred = torch.empty_like(flat)
    tmp = torch.empty((H8K, W8K, 4), dtype=torch.float32, device=device)
    r.set_exact(exact)
    try:
        r.gaussian_blur_dev(flat.data_ptr(), blurred.data_ptr(), W8K, H8K, sigma, tmp.data_ptr())
        torch.cuda.synchronize()
    finally:
        r.set_exact(False)
    tol = 0 if exact else 1
    ww, wh = 320, 200
    for (x, y) in windows(W8K, H8K, ww, wh, seed=2):
        # crop = window + halo; where the halo would leave the image the crop ends at the image border, so the oracle's
        # clamp-to-edge there is the full frame's
        x0, y0 = max(x - radius, 0), max(y - radius, 0)
        x1, y1 = min(x + ww + radius, W8K), min(y + wh + radius, H8K)
        ref = O.gaussian_blur(host(flat[y0:y1, x0:x1, :]), sigma)[y - y0:y - y0 + wh, x - x0:x - x0 + ww]
        got = host(blurred[y:y + wh, x:x + ww, :])
        d = np.abs(ref.astype(np.int16) - got.astype(np.int16))
        assert d.max() <= tol, f"window at ({x},{y}) exact={exact}: max diff {int(d.max())}"


@pytest.mark.parametrize("radius", [3, 5, 7, 8])
def test_median_8k_windows_bitexact(env, stack8k, radius):
    """the bit-plane radix select (k_median_bits.hip) at the full 8K frame: plane rows of 242 dwords, 135 row bands, the last band and the last
    64-column block ragged — corner windows (clamped borders) and interior windows against the oracle run on window + halo crops"""
    torch, r, device = env
    flat = stack8k[4]
    out = torch.empty_like(flat)
    r.median_dev(flat.data_ptr(), out.data_ptr(), W8K, H8K, radius)
    torch.cuda.synchronize()
    ww, wh = 200, 120
    for (x, y) in windows(W8K, H8K, ww, wh, seed=10 + radius):
        x0, y0 = max(x - radius, 0), max(y - radius, 0)
        x1, y1 = min(x + ww + radius, W8K), min(y + wh + radius, H8K)
        ref = O.median(host(flat[y0:y1, x0:x1, :]), radius)[y - y0:y - y0 + wh, x - x0:x - x0 + ww]
        assert np.array_equal(ref, host(out[y:y + wh, x:x + ww, :])), f"median r={radius} window at ({x},{y})"


@pytest.mark.parametrize("radius", [3.0, 9.0, 48.0])
def test_box_blur_8k_windows_bitexact(env, stack8k, radius):
    """box blur at the full 8K frame: the fused tile (r = 3), the two-pass kernels with 8 / 16 columns and 32 / 128 rows per lane (r = 9, 48;
    a row is 3.75 tiles of 2048 and 1.875 of 4096 pixels, the last row band ragged)"""
    torch, r, device = env
    flat = stack8k[4]
    out = torch.empty_like(flat)
    r.box_blur_dev(flat.data_ptr(), out.data_ptr(), W8K, H8K, radius)
    torch.cuda.synchronize()
    rad = int(np.ceil(radius))
    ww, wh = 300, 160
    for (x, y) in windows(W8K, H8K, ww, wh, seed=20 + rad):
        x0, y0 = max(x - rad, 0), max(y - rad, 0)
        x1, y1 = min(x + ww + rad, W8K), min(y + wh + rad, H8K)
        # the vertical pass reads horizontal results of rows up to `rad` away, whose own windows reach `rad` columns further: the crop's
        # left / right halo only has to cover the horizontal reach, its rows carry the vertical one
        ref = O.box_blur(host(flat[y0:y1, x0:x1, :]), radius)[y - y0:y - y0 + wh, x - x0:x - x0 + ww]
        assert np.array_equal(ref, host(out[y:y + wh, x:x + ww, :])), f"box blur r={radius} window at ({x},{y})"


def test_gaussian_8k_constant_is_a_fixed_point(env):
    torch, r, device = env
    img = torch.empty((H8K, W8K, 4), dtype=torch.uint8, device=device)
    img[...] = torch.tensor([37, 129, 250, 201], dtype=torch.uint8, device=device)
    out = torch.zeros_like(img)
    r.gaussian_blur_dev(img.data_ptr(), out.data_ptr(), W8K, H8K, 16.0, 0)
    torch.cuda.synchronize()
    assert torch.equal(out, img)


def test_hsl_8k_whole_frame_bitexact(env, stack8k):
    torch, r, device = env
    flat = stack8k[4]
    out = torch.empty_like(flat)
    r.adjust_dev(flat.data_ptr(), out.data_ptr(), W8K, H8K, "hsl", (30.0, -20.0, 10.0))
    torch.cuda.synchronize()
    src = host(flat)
    ref = O.adjust(src, "hsl", (30.0, -20.0, 10.0))
    got = host(out)
    assert np.array_equal(ref, got), f"{int((ref != got).any(-1).sum())} px differ"


# ------------------------------------------------------------------ config 3: 16K mesh warp + liquify displacement warp

@pytest.fixture(scope="module")
def img16k(env):
    torch, r, device = env
    g = torch.Generator(device=device)
    g.manual_seed(0x5EED0004)
    # smooth-ish content with hard edges: low-resolution noise upscaled by repetition, so that interpolation errors show
    small = torch.randint(0, 256, (H16K // 8, W16K // 8, 4), dtype=torch.uint8, device=device, generator=g)
    img = small.repeat_interleave(8, 0).repeat_interleave(8, 1).contiguous()
    fine = torch.randint(0, 32, (H16K, W16K, 4), dtype=torch.uint8, device=device, generator=g)
    img = (img // 2 + fine).contiguous()
    return img


def test_mesh_warp_16k_whole_frame_bitexact(env, img16k):
    torch, r, device = env
    orig, deformed = I.jittered_mesh(6, 6, W16K, H16K)
    out = torch.empty_like(img16k)
    r.warp_mesh_catmull_rom_dev(img16k.data_ptr(), orig, deformed, 6, 6, W16K, H16K, out.data_ptr())
    torch.cuda.synchronize()
    src = host(img16k)
    ref = O.warp_mesh_catmull_rom(src, orig, deformed, 6, 6)
    got = host(out)
    same = np.array_equal(ref, got)
    assert same, f"{int((ref != got).any(-1).sum())} px differ"


def test_mesh_warp_16k_identity_and_integer_shift(env, img16k):
    torch, r, device = env
    orig = I.uniform_grid(6, 6, float(W16K), float(H16K))
    out = torch.empty_like(img16k)
    r.warp_mesh_catmull_rom_dev(img16k.data_ptr(), orig, orig.copy(), 6, 6, W16K, H16K, out.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(out, img16k), "identity mesh must return the source"
    # every control point moved by (+64, -32): the surface moves rigidly, so away from the borders the result is the
    # source translated by that vector (the surface is evaluated in f32, hence the +-1 LSB allowance, not exactness)
    moved = orig.copy()
    moved[..., 0] += 64.0
    moved[..., 1] -= 32.0
    r.warp_mesh_catmull_rom_dev(img16k.data_ptr(), orig, moved, 6, 6, W16K, H16K, out.data_ptr())
    torch.cuda.synchronize()
    ys, xs = slice(4096, 4608), slice(8192, 8704)
    got = host(out[ys, xs])
    cands = [host(img16k[ys.start + sy * 32:ys.stop + sy * 32, xs.start + sx * 64:xs.stop + sx * 64]) for sx, sy in ((-1, 1), (1, -1))]
    d = min(int(np.abs(c.astype(np.int16) - got.astype(np.int16)).max()) for c in cands)
    assert d <= 1, f"rigid shift: max diff {d}"


def test_liquify_16k_whole_frame_bitexact(env, img16k):
    torch, r, device = env
    disp = torch.zeros((H16K, W16K, 2), dtype=torch.float32, device=device)
    rng = np.random.default_rng(7)
    dabs = []
    for _ in range(48):
        cx, cy = float(rng.uniform(0, W16K)), float(rng.uniform(0, H16K))
        dabs.append((int(rng.integers(0, 5)), cx, cy, float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)),
                     float(rng.uniform(200, 1500)), float(rng.uniform(0.2, 1.0))))
    r.displacement_brushes_dev(disp.data_ptr(), W16K, H16K, dabs)
    out = torch.empty_like(img16k)
    r.warp_displacement_dev(img16k.data_ptr(), W16K, H16K, disp.data_ptr(), W16K, H16K, out.data_ptr())
    torch.cuda.synchronize()
    field = host(disp)
    assert float(np.abs(field).max()) > 1.0, "the dabs must have moved something"
    ref = O.warp_displacement(host(img16k), field)
    got = host(out)
    assert np.array_equal(ref, got), f"{int((ref != got).any(-1).sum())} px differ"


# ------------------------------------------------------------------ config 4: one 4K image of the batch, whole pipeline

def test_pipeline_4k_image_whole_frame(env):
    """filter + flatten pipeline of the batch configuration on one 3840x2160 document: 8 layers flattened, Gaussian
    sigma=4 (exact mode: bit-exact), HSL, 3x3 median — each stage against the oracle fed with the oracle's own previous
    stage, so errors cannot hide behind one another"""
    import bench
    torch, r, device = env
    n = 8
    stack, modes, opac = bench.synth_stack(torch, device, W4K, H4K, n, seed=0x5EED0005)
    info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
    a = torch.empty((H4K, W4K, 4), dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    r.flatten_dev([stack[k].data_ptr() for k in range(n)], info, W4K, H4K, a.data_ptr())
    torch.cuda.synchronize()
    ref = O.flatten_stack(host(stack), modes, opac)
    assert np.array_equal(ref, host(a)), "flatten"
    r.set_exact(True)
    try:
        r.gaussian_blur_dev(a.data_ptr(), b.data_ptr(), W4K, H4K, 4.0, 0)
        torch.cuda.synchronize()
    finally:
        r.set_exact(False)
    ref = O.gaussian_blur(ref, 4.0)
    assert np.array_equal(ref, host(b)), "gaussian (exact mode)"
    r.adjust_dev(b.data_ptr(), a.data_ptr(), W4K, H4K, "hsl", (-45.0, 35.0, -5.0))
    torch.cuda.synchronize()
    ref = O.adjust(ref, "hsl", (-45.0, 35.0, -5.0))
    assert np.array_equal(ref, host(a)), "hsl"
    r.median_dev(a.data_ptr(), b.data_ptr(), W4K, H4K, 1)
    torch.cuda.synchronize()
    ref = O.median(ref, 1)
    assert np.array_equal(ref, host(b)), "median"
