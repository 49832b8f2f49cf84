"""Parity (GPU): pfx_chain_dev — several ops in as few passes as the kernels allow — against the CPU oracle applied op by op, bit for bit (in the bit-exact
Gaussian mode) and against the library's own single-op entry points in the default mode.  Covers: runs of pointwise ops of both numeric flavours with and
without lookup tables, runs longer than one launch carries, the bit-exact Gaussian with the chain in its store (radii 1 .. 16) and beyond (two launches),
box blurs between pointwise runs (the ping-pong through the scratch image), in-place chains, ragged sizes.
References: src/ops/filters.rs:214-316, src/ops/adjustments.rs:21-631, src/ops/scripting.rs:869-1075 (each op is a full pass over the image there)."""
import numpy as np
import pytest

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def r():
    from paintfe_amd import GpuRenderer
    return GpuRenderer(0)


def run_chain(r, img, ops, in_place=False):
    h, w = img.shape[:2]
    a = r.dev_alloc(img.nbytes)
    b = a if in_place else r.dev_alloc(img.nbytes)
    try:
        r.dev_upload(a, img)
        r.chain_dev(a, b, w, h, ops)
        r.synchronize()
        return r.dev_download(b, img.shape)
    finally:
        r.dev_free(a)
        if b != a:
            r.dev_free(b)


def run_oracle(img, ops):
    out = img
    for o in ops:
        if o[0] == "gaussian": out = O.gaussian_blur(out, o[1])
        elif o[0] == "box": out = O.box_blur(out, o[1])
        elif o[0] == "adjust": out = O.adjust(out, o[1], o[2] if len(o) > 2 else (), lut=o[3] if len(o) > 3 else None)
        else: out = O.rhai_adjust(out, o[1], o[2] if len(o) > 2 else ())
    return out


def lut_rgba(seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, 1024, dtype=np.uint8)


POINTWISE = [("adjust", "hsl", (30.0, -20.0, 10.0)), ("adjust", "invert"), ("adjust", "brightness_contrast", (12.0, 35.0)), ("adjust", "exposure", (0.7,)),
             ("adjust", "vibrance", (40.0,)), ("adjust", "sepia"), ("adjust", "posterize", (5.0,)), ("adjust", "temperature_tint", (20.0, -10.0)),
             ("adjust", "highlights_shadows", (30.0, -40.0)), ("adjust", "threshold", (128.0,)), ("adjust", "color_balance", (5, -5, 10, 0, 3, -3, 8, 2, -6)),
             ("adjust", "black_and_white", (30.0, 59.0, 11.0)), ("adjust", "desaturate"), ("adjust", "invert_alpha"), ("adjust", "lut_rgba", (), lut_rgba(1)),
             ("adjust", "gradient_map", (), lut_rgba(2)),
             ("rhai", "invert"), ("rhai", "desaturate"), ("rhai", "sepia"), ("rhai", "sepia_strength", (0.6,)), ("rhai", "brightness_contrast", (-15.0, 20.0)),
             ("rhai", "hsl", (-40.0, 25.0, -5.0)), ("rhai", "exposure", (-0.5,)), ("rhai", "levels", (10.0, 240.0, 1.3))]


@pytest.mark.parametrize("shape", [(64, 64), (203, 117), (256, 192), (61, 130)])
def test_every_pointwise_op_alone_and_in_runs(r, shape):
    w, h = shape
    img = I.random_rgba(w, h, seed=w * 7 + h)
    for o in POINTWISE:
        assert np.array_equal(run_chain(r, img, [o]), run_oracle(img, [o])), o[:2]
    rng = np.random.default_rng(w + h)
    for trial in range(12):
        n = int(rng.integers(2, 12))      # up to 11 ops: more than one launch carries (PFXK_CHAIN_MAX = 8), several tables
        ops = [POINTWISE[int(k)] for k in rng.integers(0, len(POINTWISE), n)]
        want = run_oracle(img, ops)
        assert np.array_equal(run_chain(r, img, ops), want), [o[:2] for o in ops]
        assert np.array_equal(run_chain(r, img, ops, in_place=True), want), ("in place", [o[:2] for o in ops])


def test_more_tables_than_one_launch_holds(r):
    img = I.random_rgba(130, 70, seed=5)
    ops = [("adjust", "lut_rgba", (), lut_rgba(k)) for k in range(6)] + [("rhai", "levels", (5.0, 250.0, 0.8))]
    assert np.array_equal(run_chain(r, img, ops), run_oracle(img, ops))


@pytest.mark.parametrize("sigma", [0.3, 0.5, 1.0, 2.0, 4.0, 5.33, 5.4, 8.0])
def test_exact_gaussian_with_the_chain_in_its_store(r, sigma):
    """bit-exact mode: radius <= 16 runs the blur and the pointwise ops as ONE launch, larger radii as blur + chain; both equal the oracle op by op"""
    r.set_exact(True)
    try:
        for (w, h) in [(203, 117), (256, 128), (70, 300)]:
            img = I.random_rgba(w, h, seed=int(sigma * 100) + w)
            for tail in ([("adjust", "hsl", (30.0, -20.0, 10.0))], [("rhai", "exposure", (0.4,)), ("rhai", "sepia"), ("rhai", "invert")],
                         [("adjust", "lut_rgba", (), lut_rgba(3)), ("adjust", "vibrance", (25.0,))], []):
                ops = [("gaussian", sigma)] + tail
                assert np.array_equal(run_chain(r, img, ops), run_oracle(img, ops)), (sigma, w, h, [o[:2] for o in tail])
    finally:
        r.set_exact(False)


def test_heavy_ops_fused_into_the_gaussians_on_request(r):
    """HSL / vibrance run as their own in-place pass behind a Gaussian by default (no gain from fusing, profiles/r06_tuning.md); pfx_tune chain_fuse_heavy = 1 puts
    them in the Gaussian's store — both kernels' chained forms must give the same bits"""
    img = I.random_rgba(256, 160, seed=9)
    ops = [("gaussian", 3.0), ("adjust", "hsl", (30.0, -20.0, 10.0)), ("adjust", "vibrance", (35.0,)), ("rhai", "hsl", (5.0, 5.0, 5.0))]
    try:
        r.set_exact(True)
        want = run_oracle(img, ops)
        assert np.array_equal(run_chain(r, img, ops), want)
        r.tune("chain_fuse_heavy", 1)
        assert np.array_equal(run_chain(r, img, ops), want)
        r.set_exact(False)                      # default mode: matrix-core Gaussian, fused against unfused
        fused = run_chain(r, img, ops)
        r.tune("chain_fuse_heavy", 0)
        assert np.array_equal(run_chain(r, img, ops), fused)
        for sigma in (3.0, 8.0, 14.0):          # 4 K blocks (32-column strips), 6 and 8 (64-column strips): every chained build of the matrix-core Gaussian
            light = [("gaussian", sigma), ("adjust", "exposure", (0.4,)), ("adjust", "invert")]
            r.tune("chain_mfma", 1)
            a = run_chain(r, img, light)        # light ops: in the matrix-core Gaussian's store by default
            r.tune("chain_mfma", 0)
            assert np.array_equal(run_chain(r, img, light), a), sigma
    finally:
        r.tune("chain_mfma", 1); r.tune("chain_fuse_heavy", 0); r.set_exact(False)


def test_stencils_between_pointwise_runs_ping_pong_through_the_scratch_image(r):
    r.set_exact(True)
    try:
        img = I.random_rgba(190, 140, seed=11)
        cases = [
            [("adjust", "invert"), ("gaussian", 2.0), ("adjust", "hsl", (10.0, 10.0, 0.0))],
            [("box", 3.0), ("rhai", "sepia"), ("gaussian", 1.5), ("rhai", "invert")],
            [("gaussian", 1.0), ("gaussian", 2.0), ("gaussian", 3.0)],
            [("adjust", "exposure", (0.3,)), ("box", 2.0), ("box", 5.0), ("adjust", "threshold", (100.0,)), ("gaussian", 6.0), ("adjust", "invert")],
            [("box", 0.2)],          # radius below 0.5: the box blur is a copy (blur.rs:234)
            [],
        ]
        for ops in cases:
            assert np.array_equal(run_chain(r, img, ops), run_oracle(img, ops)), [o[:2] for o in ops]
    finally:
        r.set_exact(False)


def test_default_mode_equals_the_single_op_calls(r):
    """default (matrix-core) Gaussian mode: the chain is defined as the single-op entry points one after the other"""
    w, h = 256, 200
    img = I.random_rgba(w, h, seed=3)
    a, b, c = (r.dev_alloc(img.nbytes) for _ in range(3))
    try:
        r.dev_upload(a, img)
        r.gaussian_blur_dev(a, b, w, h, 4.0)
        r.adjust_dev(b, c, w, h, "hsl", (30.0, -20.0, 10.0))
        r.synchronize()
        want = r.dev_download(c, img.shape)
    finally:
        for p in (a, b, c):
            r.dev_free(p)
    got = run_chain(r, img, [("gaussian", 4.0), ("adjust", "hsl", (30.0, -20.0, 10.0))])
    assert np.array_equal(got, want)
    assert np.abs(got.astype(np.int16) - run_oracle(img, [("gaussian", 4.0), ("adjust", "hsl", (30.0, -20.0, 10.0))]).astype(np.int16)).max() <= 8   # HSL amplifies the blur's +-1


def test_errors_leave_a_message(r):
    from paintfe_amd import PfxError
    img = I.random_rgba(64, 64, seed=1)
    with pytest.raises(PfxError):
        run_chain(r, img, [("gaussian", 2.0)], in_place=True)
    with pytest.raises(PfxError):
        run_chain(r, img, [("adjust", "hsl", (1.0,))])           # too few parameters
    with pytest.raises(PfxError):
        run_chain(r, img, [("adjust", "lut_rgba", ())])          # table missing
