"""Aliasing contract of the `_dev` tier (include/pfx.h): in-place box blur / Gaussian take their two-pass kernels and stay bit-exact;
neighbourhood operations refuse overlapping buffers instead of racing (the fused box kernel stages a halo tile from src while
neighbouring workgroups write dst)."""
import numpy as np
import pytest
import torch  # noqa: F401  -- before libpfx.so is loaded (see test_gpu_fullsize.py)

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("radius", [1.0, 3.0, 8.0, 20.0])
def test_box_blur_in_place_is_bitexact(env, radius):
    torch, r, dev = env
    w, h = 700, 333                                   # many 64 x 64 tiles: an in-place fused kernel would read neighbours' results
    img = I.random_rgba(w, h, 77)
    ref = O.box_blur(img, radius)
    for _ in range(3):                                # a race is nondeterministic: look more than once
        d = torch.from_numpy(img).to(dev)
        r.box_blur_dev(d.data_ptr(), d.data_ptr(), w, h, radius)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), ref), f"in-place box blur r={radius}"
    out = torch.empty_like(d)
    d = torch.from_numpy(img).to(dev)
    r.box_blur_dev(d.data_ptr(), out.data_ptr(), w, h, radius)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref), "out-of-place box blur"


def test_gaussian_in_place_exact_mode_is_bitexact(env):
    torch, r, dev = env
    w, h = 517, 211
    img = I.random_rgba(w, h, 78)
    r.set_exact(True)
    try:
        d = torch.from_numpy(img).to(dev)
        r.gaussian_blur_dev(d.data_ptr(), d.data_ptr(), w, h, 3.0)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy(), O.gaussian_blur(img, 3.0))
    finally:
        r.set_exact(False)
    d = torch.from_numpy(img).to(dev)                 # default mode in place: the f32 two-pass path, +-1 LSB class
    r.gaussian_blur_dev(d.data_ptr(), d.data_ptr(), w, h, 3.0)
    torch.cuda.synchronize()
    assert np.abs(d.cpu().numpy().astype(np.int16) - O.gaussian_blur(img, 3.0).astype(np.int16)).max() <= 1


def test_neighbourhood_ops_refuse_overlapping_buffers(env):
    from paintfe_amd._lib import PfxError, ERR_INVALID
    torch, r, dev = env
    w, h = 128, 64
    buf = torch.zeros((2 * h, w, 4), dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    before = buf.clone()
    same, shifted, shifted_back = (base, base), (base, base + w * 4 * 10), (base + 64, base)
    cases = [(lambda s, d: r.median_dev(s, d, w, h, 2), (same, shifted, shifted_back)),
             (lambda s, d: r.box_blur_dev(s, d, w, h, 2.0), (shifted, shifted_back)),       # src == dst is the documented in-place form
             (lambda s, d: r.gaussian_blur_dev(s, d, w, h, 2.0), (shifted, shifted_back))]
    for call, pairs in cases:
        for s_, d_ in pairs:
            with pytest.raises(PfxError) as e:
                call(s_, d_)
            assert e.value.status == ERR_INVALID
    torch.cuda.synchronize()
    assert torch.equal(buf, before), "a refused call must not touch the destination"
    # NaN sigma whose bit pattern used to be the weight cache's "empty" marker: radius 0, must not launch on a null table
    nan = np.frombuffer(np.uint32(0xFFFFFFFF).tobytes(), np.float32)[0]
    d2 = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    try:
        r.gaussian_blur_dev(base, d2.data_ptr(), w, h, float(nan))
    except PfxError:
        pass
    torch.cuda.synchronize()


def test_pointwise_and_resampling_ops_refuse_partial_overlaps(env):
    """pfx_adjust_dev accepts src == dst (pointwise) but a PARTIAL overlap races silently; the displacement warp must test the SOURCE's extent
    (an sw x sh image, possibly larger than the w x h output); the resize family only compared the two pointers"""
    from paintfe_amd._lib import PfxError, ERR_INVALID
    torch, r, dev = env
    w, h = 128, 64
    buf = torch.zeros((4 * h, w, 4), dtype=torch.uint8, device=dev)
    base = buf.data_ptr()
    before = buf.clone()
    r.adjust_dev(base, base, w, h, "invert")                          # in place: allowed
    torch.cuda.synchronize()
    buf.copy_(before)
    for s_, d_ in ((base, base + w * 4 * 7), (base + 16, base)):
        with pytest.raises(PfxError) as e:
            r.adjust_dev(s_, d_, w, h, "invert")
        assert e.value.status == ERR_INVALID
    # warp: a 128 x 192 source whose TAIL overlaps the 128 x 64 output buffer (the head does not)
    disp = torch.zeros((h, w, 2), dtype=torch.float32, device=dev)
    with pytest.raises(PfxError) as e:
        r.warp_displacement_dev(base, w, 3 * h, disp.data_ptr(), w, h, base + w * 4 * (2 * h))
    assert e.value.status == ERR_INVALID
    r.warp_displacement_dev(base, w, 3 * h, disp.data_ptr(), w, h, base + w * 4 * (3 * h))   # disjoint: fine
    with pytest.raises(PfxError) as e:
        r.resize_image_dev(base, w, h, base + w * 4 * (h // 2), w // 2, h // 2)
    assert e.value.status == ERR_INVALID
    torch.cuda.synchronize()
    assert torch.equal(buf[:3 * h], before[:3 * h]), "a refused call must not touch its buffers"
