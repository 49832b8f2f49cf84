"""Parity gate (GPU): dead-layer elimination in the streaming compositor (k_flatten.hip: flatten_dle_kernel).

blend_pixel_static has two results that ignore `base` (canvas_state.rs:1258 opaque Normal at opacity >= 1, :1275-1281 Overwrite
with non-zero alpha); the kernel skips the layers below a pixel's topmost such layer, compacting the pixels that still need them.
Everything here must equal the oracle — which walks every layer for every pixel — bit for bit: stacks with reset layers at random
depths, alpha = 0 holes in them, Normal layers below 100 % (must NOT reset), more candidates than the kernel tracks, spatially
coherent and per-pixel-random coverage at every fraction, ragged sizes, every queue / ring configuration, and masks or adjustment
layers that move the stack to the general kernel."""
import numpy as np
import pytest

from . import inputs as I
from . import oracle_lib as O

pytestmark = pytest.mark.gpu

OVERWRITE, NORMAL = 14, 0


@pytest.fixture(scope="module")
def gpu():
    from .backends import GpuBackend
    g = GpuBackend(0)
    g.r.tune("dle_min_layers", 0)  # the library only takes the elimination kernel for stacks of 16+ layers; here every stack must
    yield g
    g.r.tune("dle_min_layers", 16)
    g.r.tune("dle_units", 0)
    g.r.tune("dle_cfg", 0)
    g.r.tune("flatten_variant", 0)
    set_config(g, (0, 0, 1, -1, -1))


def check(gpu, stack, modes, opac, what):
    n, h, w, _ = stack.shape
    layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(n)]
    got = gpu.composite(layers, w, h)
    ref = O.flatten_stack(stack, np.asarray(modes, np.uint8), np.asarray(opac, np.float32))
    bad = (got != ref).any(-1)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {w * h} px differ, first at flat index {int(np.flatnonzero(bad)[0])}"


def noise_alpha(rng, h, w, p_zero, p_opaque):
    u = rng.random((h, w))
    a = rng.integers(1, 255, (h, w), dtype=np.uint8)
    return np.where(u < p_zero, 0, np.where(u < p_zero + p_opaque, 255, a)).astype(np.uint8)


def blocky_alpha(rng, h, w, cell, p_zero, p_opaque):
    gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
    coarse = noise_alpha(rng, gh, gw, p_zero, p_opaque)
    return np.kron(coarse, np.ones((cell, cell), np.uint8))[:h, :w]


# (dle_kernel, dle_cfg, dle_sched, dle_s1, dle_s2): kernel 0 = class sorting (flatten_srt_kernel: a unit's pixels re-dealt to the lanes — early pixels first
# for the layers below the split, then by "accumulator opaque?" along its natural pass), 1 = round 3's kernel (early pixels queued across units, accumulators
# parked in an LDS ring, lane order throughout).  dle_s1: the first re-deal attempt, in layers above the topmost candidate (-1 = 1, 0 = never); dle_s2: layers between
# attempts (-1 = 3, 0 = only the first).
CONFIGS = {
    "srt-px3-default": (0, 0, 1, -1, -1),
    "srt-px3-every-layer": (0, 0, 1, 1, 1),
    "srt-px3-from-3-every-3-equal": (0, 0, 0, 3, 3),
    "srt-px3-one-attempt": (0, 0, 1, 2, 0),
    "srt-px3-never": (0, 0, 1, 0, -1),
    "srt-px2-every-layer": (0, 1, 1, 1, 1),
    "srt-px2-equal-default": (0, 1, 0, -1, -1),
    "r3-px3x2sets-shrinking": (1, 0, 1, -1, -1),
    "r3-px2-shrinking": (1, 1, 1, -1, -1),
    "r3-px3x3sets-equal": (1, 2, 0, -1, -1),
}


def set_config(gpu, cfg):
    kernel, dcfg, sched, s1, s2 = cfg
    gpu.r.tune("dle_kernel", kernel)
    gpu.r.tune("dle_cfg", dcfg)
    gpu.r.tune("dle_sched", sched)
    gpu.r.tune("dle_s1", s1)
    gpu.r.tune("dle_s2", s2)


@pytest.fixture(params=list(CONFIGS.values()), ids=list(CONFIGS.keys()), autouse=True)
def every_kernel_configuration(request, gpu):
    """every test of this file runs on both elimination kernels, their instantiations (pfx_tune "dle_cfg": pixels per lane, register sets), both stream
    schedules ("dle_sched": equal streams, or streams that shrink towards the end of the launch) and several re-deal plans of the class-sorting kernel"""
    set_config(gpu, request.param)
    yield request.param
    set_config(gpu, (0, 0, 1, -1, -1))


@pytest.mark.parametrize("units", [0, 1, 2, 5, 340])
def test_s2_stack_every_queue_configuration(gpu, units):
    """BASELINE's S2 stack (Overwrite at depth 14, alpha non-zero on a random 75 %) with short and long wave streams"""
    gpu.r.tune("dle_units", units)
    w, h, n = 517, 263, 32
    stack, modes, opac = I.layer_stack(w, h, n, seed=1234 + units)
    check(gpu, stack, modes, opac, f"S2 units={units}")
    gpu.r.tune("dle_units", 0)


@pytest.mark.parametrize("p_live", [0.0, 0.004, 0.05, 0.28, 0.32, 0.5, 0.75, 0.97, 1.0])
@pytest.mark.parametrize("kind", ["overwrite", "normal"])
def test_single_reset_layer_at_every_coverage(gpu, kind, p_live):
    """one reset layer at depth 9 of 14 whose qualifying pixels are a random fraction p_live of the image: below the 30 % threshold
    (no compaction), around it, the compacted regime, and the forced partial flushes of a nearly empty queue"""
    w, h, n = 389, 211, 14
    rng = np.random.default_rng(int(p_live * 1000) + (7 if kind == "normal" else 0))
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    modes = [0] + [1 + (3 * k) % 24 for k in range(1, n)]
    modes = [m if m not in (OVERWRITE, NORMAL) else 2 for m in modes]
    opac = [1.0] + [float(np.float32(0.3 + 0.7 * rng.random())) for _ in range(1, n)]
    for k in range(1, n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.25, 0.25)
    if kind == "overwrite":
        modes[9], opac[9] = OVERWRITE, 0.6                       # any opacity resets
        stack[9, ..., 3] = np.where(rng.random((h, w)) < p_live, rng.integers(1, 256, (h, w)), 0).astype(np.uint8)
    else:
        modes[9], opac[9] = NORMAL, 1.0                          # only alpha 255 resets
        stack[9, ..., 3] = np.where(rng.random((h, w)) < p_live, 255, rng.integers(0, 255, (h, w))).astype(np.uint8)
    check(gpu, stack, modes, opac, f"{kind} p_live={p_live}")


@pytest.mark.parametrize("seed", range(12))
def test_random_stacks_with_reset_layers_at_random_depths(gpu, seed):
    rng = np.random.default_rng(7000 + seed)
    w, h = int(rng.integers(40, 700)), int(rng.integers(20, 300))
    n = int(rng.integers(2, 34))
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    modes, opac = [], []
    for k in range(n):
        t = rng.random()
        if t < 0.22:
            modes.append(OVERWRITE); opac.append(float(rng.choice([1.0, 0.5, 1e-3, 0.999])))
        elif t < 0.40:
            modes.append(NORMAL); opac.append(float(rng.choice([1.0, 1.5, 1.0, 2.0])))         # resets where alpha is 255
        elif t < 0.50:
            modes.append(NORMAL); opac.append(float(rng.choice([0.999999, 0.5])))              # never resets
        else:
            modes.append(int(rng.integers(0, 25))); opac.append(float(np.float32(0.05 + 0.95 * rng.random())))
        style = rng.integers(0, 6)
        if style == 0:
            a = noise_alpha(rng, h, w, 0.25, 0.25)
        elif style == 1:
            a = noise_alpha(rng, h, w, float(rng.random()), float(rng.random()) * 0.5)
        elif style == 2:
            a = blocky_alpha(rng, h, w, int(rng.choice([3, 16, 64, 200])), 0.3, 0.5)
        elif style == 3:
            a = np.full((h, w), 255, np.uint8)
        elif style == 4:
            a = np.zeros((h, w), np.uint8)
        else:
            a = np.broadcast_to((np.arange(w) * 256 // w).astype(np.uint8), (h, w)).copy()
        stack[k, ..., 3] = a
    gpu.r.tune("dle_units", int(rng.choice([0, 1, 3, 7, 24])))
    check(gpu, stack, modes, opac, f"seed {seed}: {w}x{h}x{n}, modes {modes}")
    gpu.r.tune("dle_units", 0)


def test_more_candidates_than_the_kernel_tracks_and_bottom_or_top_position(gpu):
    w, h, n = 300, 150, 12
    rng = np.random.default_rng(31)
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    for k in range(n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.4, 0.3)
    modes = [OVERWRITE, NORMAL, OVERWRITE, 5, NORMAL, OVERWRITE, 8, OVERWRITE, NORMAL, 16, NORMAL, OVERWRITE]
    opac = [1.0, 1.0, 0.7, 0.5, 1.0, 1.0, 0.9, 0.2, 1.0, 1.0, 1.0, 0.8]
    check(gpu, stack, modes, opac, "7+ candidates, reset layer at the bottom and on top")
    # a reset layer covering everything on top: every unit starts at the last layer
    stack[n - 1, ..., 3] = 255
    check(gpu, stack, modes, opac, "opaque Overwrite on top")
    modes[n - 1], opac[n - 1] = NORMAL, 1.0
    check(gpu, stack, modes, opac, "opaque Normal on top")


@pytest.mark.parametrize("size", [(1, 1), (191, 1), (192, 1), (193, 1), (64, 3), (5, 77), (4609, 1), (383, 13)])
def test_tiny_and_ragged_sizes(gpu, size):
    w, h = size
    n = 6
    rng = np.random.default_rng(w * 131 + h)
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    for k in range(n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.25, 0.25)
    modes, opac = [0, 3, 9, OVERWRITE, 20, 1], [1.0, 0.5, 1.0, 1.0, 0.7, 0.4]
    for units in (1, 0):
        gpu.r.tune("dle_units", units)
        check(gpu, stack, modes, opac, f"{w}x{h} units={units}")
    gpu.r.tune("dle_units", 0)


def test_coherent_documents_start_at_the_covering_layer(gpu):
    """photo-like documents: opaque Normal layers covering rectangles (units agree on their reset layer, units on a rectangle's
    edge are mixed), a soft-edged opaque blob, an Overwrite patch with a hole"""
    w, h, n = 640, 256, 9
    rng = np.random.default_rng(99)
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    stack[:, ..., 3] = 0
    stack[0, ..., 3] = 255
    stack[1, ..., 3] = noise_alpha(rng, h, w, 0.2, 0.2)
    stack[2, 30:200, 50:400, 3] = 255                                   # opaque photo (Normal 100 %)
    stack[3, ..., 3] = noise_alpha(rng, h, w, 0.5, 0.1)
    yy, xx = np.mgrid[0:h, 0:w]
    d = np.hypot(yy - 128, xx - 420)
    stack[4, ..., 3] = np.clip(255 * (90 - d) / 20, 0, 255).astype(np.uint8)   # soft-edged opaque disc
    stack[5, 100:180, 300:600, 3] = 200                                 # Overwrite patch ...
    stack[5, 120:140, 350:380, 3] = 0                                   # ... with a hole
    stack[6, ..., 3] = noise_alpha(rng, h, w, 0.6, 0.0)
    stack[7, :, 600:, 3] = 255                                          # opaque strip, Normal but at 99 %: no reset
    stack[8, ..., 3] = noise_alpha(rng, h, w, 0.7, 0.0)
    modes = [0, 1, NORMAL, 8, NORMAL, OVERWRITE, 2, NORMAL, 16]
    opac = [1.0, 0.8, 1.0, 0.6, 1.0, 0.75, 1.0, 0.99, 0.5]
    check(gpu, stack, modes, opac, "coherent document")


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 16, 17, 23])
def test_streaming_kernel_shapes_agree_with_the_oracle(gpu, n):
    """the plain streaming kernel's launch / register shapes (pfx_tune flatten_variant): the shipped one (2 pixels per lane, 2 register sets, grid stride
    up to 16 layers), the other pixels-per-lane x set combinations, the round-2 shape (6), each with one tile per wave and with a grid stride (+10) —
    stacks shorter than the register sets, at the stride switch (16 / 17 layers) and a width that leaves a ragged last tile"""
    w, h = 389, 67
    stack, modes, opac = I.layer_stack(w, h, n, seed=40 + n)
    modes = [m if m != OVERWRITE else 1 for m in modes]                 # no reset layer: the elimination kernel stays out of it
    for v in (0, 1, 2, 3, 4, 5, 6, 10, 11, 12, 13, 14, 15, 16):
        gpu.r.tune("flatten_variant", v)
        try:
            check(gpu, stack, modes, list(opac), f"{n} layers, variant {v}")
        finally:
            gpu.r.tune("flatten_variant", 0)


@pytest.mark.parametrize("n", [4, 5, 6, 7, 9, 17])
@pytest.mark.parametrize("cls", ["light", "medium", "heavy", "mixed"])
def test_streaming_kernel_shape_by_mode_class(gpu, n, cls):
    """the automatic shape (flatten_variant 0) follows the heaviest blend mode of the stack (pfx_api.cpp: build_stack -> pfxk_flatten mode_class): light and medium
    stacks of 5+ layers run 4 pixels per lane (one tile per wave from 7 layers, grid-stride below / for medium), heavy ones and stacks of up to 4 layers 2 — every
    class at the depths where the rule switches, on a width that leaves a ragged last tile"""
    w, h = 397, 71
    stack, _, opac = I.layer_stack(w, h, n, seed=90 + n)
    pool = {"light": [0, 1, 3, 9, 11, 12, 18, 20], "medium": [2, 8, 10, 15, 17, 22, 24, 1], "heavy": [4, 5, 6, 7, 13, 16, 19, 21, 23],
            "mixed": [0, 2, 21, 1, 16, 8, 3, 23]}[cls]
    modes = [pool[k % len(pool)] for k in range(n)]
    opac = [o if o < 1.0 else 0.97 for o in opac]                        # no Normal-at-100 % reset candidates: the streaming kernel runs
    gpu.r.tune("flatten_variant", 0)
    check(gpu, stack, modes, list(opac), f"{n} layers, {cls} modes, automatic shape")


def test_switched_off_and_general_path_agree(gpu):
    """the same stack through the plain streaming kernel (variant 8), the general kernel (variant 9) and with a live mask on the
    reset layer (the mask can lower alpha to 0: the host must not list that layer)"""
    w, h, n = 333, 120, 8
    stack, modes, opac = I.layer_stack(w, h, n, seed=5)
    modes = list(modes); opac = list(opac)
    modes[5], opac[5] = OVERWRITE, 1.0
    for v in (8, 9, 0):
        gpu.r.tune("flatten_variant", v)
        check(gpu, stack, modes, opac, f"variant {v}")
    gpu.r.tune("flatten_variant", 0)
    rng = np.random.default_rng(17)
    mask = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mask[:, : w // 2] = 255                                             # fully concealed half: the Overwrite layer vanishes there
    layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(n)]
    layers[5]["mask"] = mask
    got = gpu.composite(layers, w, h)
    ref = O.composite(layers, w, h)
    assert np.array_equal(got, ref), "masked reset layer"


def test_stored_layers_start_at_the_covering_layer_per_chunk(gpu):
    """layers of the layer store carry per-chunk alpha summaries: a tile of the plain streaming kernel starts at the topmost layer that
    resets all the chunks it touches (opaque Normal at 100 %, Overwrite without a hole).  Chunk-aligned and unaligned opaque regions,
    a one-pixel hole in an otherwise opaque chunk, a Normal layer at 99 %, a rectangle update that opens and closes a hole"""
    gpu.r.tune("dle_min_layers", 16)   # shallow stack: the plain kernel + the chunk table, not the elimination kernel
    try:
        w, h, n = 448, 200, 7
        rng = np.random.default_rng(4)
        stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
        for k in range(1, n):
            stack[k, ..., 3] = noise_alpha(rng, h, w, 0.3, 0.2)
        stack[0, ..., 3] = 255
        stack[2, :, :, 3] = 255                       # opaque photo over everything (Normal 100 %): every tile may start at layer 2
        stack[4, 64:128, 128:320, 3] = 255            # chunk-aligned opaque patch
        stack[4, 100, 200, 3] = 254                   # ... with a one-pixel dent in chunk (1, 3)
        stack[5, 30:170, 50:400, 3] = 200             # Overwrite patch covering whole chunks and partial ones
        stack[5, 70, 300, 3] = 0                      # ... with a hole
        stack[6, :64, :, 3] = 255                     # opaque strip at 99 %: never a reset
        modes = [0, 7, NORMAL, 3, NORMAL, OVERWRITE, NORMAL]
        opac = [1.0, 0.8, 1.0, 0.5, 1.0, 0.6, 0.99]
        check(gpu, stack, modes, opac, "stored layers with covering chunks")
        # the same through update_rect: close the dent and the hole, open a new hole — the summaries must follow the pixels
        layers = [dict(pixels=stack[k], mode=int(modes[k]), opacity=float(opac[k])) for k in range(n)]
        gpu.composite(layers, w, h)                   # uploads (generation 1)
        patch = stack[4, 96:104, 196:204].copy(); patch[..., 3] = 255
        gpu.r.update_layer_rect(4, 196, 96, patch)
        stack[4, 96:104, 196:204] = patch
        hole = stack[2, 10:12, 300:303].copy(); hole[..., 3] = 0
        gpu.r.update_layer_rect(2, 300, 10, hole)
        stack[2, 10:12, 300:303] = hole
        info = [(k, float(opac[k]), True, int(modes[k]), 0, ()) for k in range(n)]
        got = gpu.r.composite(w, h, info)
        ref = O.flatten_stack(stack, np.asarray(modes, np.uint8), np.asarray(opac, np.float32))
        assert np.array_equal(got, ref), f"after update_rect: {int((got != ref).any(-1).sum())} px differ"
    finally:
        gpu.r.tune("dle_min_layers", 0)


def class_stack(rng, w, h, n, reset_at, p_opaque_top, modes_above):
    """a stack with an Overwrite reset layer at `reset_at` (alpha non-zero on 75 %) and, above it, layers whose alpha is 255 with probability
    p_opaque_top at 100 % opacity on even positions (what turns an accumulator opaque) and translucent elsewhere"""
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    modes = [0] + [1 + (5 * k) % 24 for k in range(1, n)]
    modes = [m if m not in (OVERWRITE, NORMAL, 13) else 2 for m in modes]
    opac = [1.0 if k % 2 == 0 else float(np.float32(0.25 + 0.75 * rng.random())) for k in range(n)]
    for k in range(1, n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.25, p_opaque_top)
    stack[0, ..., 3] = 255
    modes[reset_at], opac[reset_at] = OVERWRITE, 1.0
    stack[reset_at, ..., 3] = noise_alpha(rng, h, w, 0.25, 0.25)
    for k, m in modes_above.items():
        modes[k] = m
    return stack, modes, opac


@pytest.mark.parametrize("p_opaque_top", [0.0, 0.02, 0.25, 0.6, 1.0])
def test_accumulator_classes_at_every_mixture(gpu, p_opaque_top):
    """above the reset layer the accumulators turn opaque at a rate set by p_opaque_top: never (no re-deal ever pays), rarely (one group completes late),
    S2's rate, mostly, and at once (every group is opaque after the first layer)"""
    rng = np.random.default_rng(int(p_opaque_top * 100) + 900)
    w, h, n = 451, 233, 20
    stack, modes, opac = class_stack(rng, w, h, n, 6, p_opaque_top, {})
    check(gpu, stack, modes, opac, f"class mixture p_opaque_top={p_opaque_top}")


@pytest.mark.parametrize("breaker", [13, OVERWRITE])
@pytest.mark.parametrize("pos", [8, 10, 14, 19])
def test_xor_or_overwrite_above_the_split_resets_the_classes(gpu, breaker, pos):
    """an Xor layer (canvas_state.rs:1283: lowers alpha, to zero where both are opaque) or a translucent Overwrite layer above the split points makes opaque
    accumulators non-opaque again in groups that were dealt as opaque: the grouping is only a hint, the per-layer test must notice"""
    rng = np.random.default_rng(breaker * 100 + pos)
    w, h, n = 333, 197, 20
    stack, modes, opac = class_stack(rng, w, h, n, 5, 0.4, {pos: breaker})
    opac[pos] = 0.7 if breaker == OVERWRITE else 1.0
    if breaker == OVERWRITE and pos == 19:
        stack[pos, ..., 3] = noise_alpha(rng, h, w, 0.5, 0.1)       # a second candidate on top, with holes
    check(gpu, stack, modes, opac, f"mode {breaker} at layer {pos}")


def test_opaque_bottom_without_early_pixels_and_all_early_pixels(gpu):
    """the reset layer covers everything (no early pixel anywhere: the natural pass starts from scratch) / nothing (every pixel is early)"""
    rng = np.random.default_rng(77)
    w, h, n = 300, 211, 18
    stack, modes, opac = class_stack(rng, w, h, n, 7, 0.25, {})
    stack[7, ..., 3] = rng.integers(1, 256, (h, w), dtype=np.uint8)
    check(gpu, stack, modes, opac, "reset layer without holes")
    stack[7, ..., 3] = 0
    check(gpu, stack, modes, opac, "reset layer all holes")
    stack[7, ..., 3] = np.where(np.arange(w)[None, :] < w // 2, 255, 0).astype(np.uint8)
    check(gpu, stack, modes, opac, "reset layer covering the left half")


def test_destination_that_aliases_a_layer(gpu):
    """pfx_flatten_dev allows the destination to BE one of the layers (include/pfx.h): every pixel is written after its last read, whichever kernel
    runs; compositing in place over layer 9 equals the oracle"""
    rng = np.random.default_rng(5)
    w, h, n = 384, 96, 18
    stack, modes, opac = class_stack(rng, w, h, n, 6, 0.3, {})
    ref = O.flatten_stack(stack, np.asarray(modes, np.uint8), np.asarray(opac, np.float32))
    r = gpu.r
    bufs = [r.dev_alloc(w * h * 4) for _ in range(n)]
    try:
        for k in range(n):
            r.dev_upload(bufs[k], stack[k])
        info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
        r.flatten_dev([b for b in bufs], info, w, h, bufs[9])
        got = r.dev_download(bufs[9], (h, w, 4))
        assert np.array_equal(got, ref)
    finally:
        for b in bufs:
            r.dev_free(b)


@pytest.mark.parametrize("n", [125, 126, 253, 254])
@pytest.mark.parametrize("odd_pass", [True, False])
def test_descriptor_table_that_ends_on_a_page_boundary(gpu, n, odd_pass):
    """srt_layers reads layer descriptors ahead of the layer it blends without clamping (up to descriptor n + 2 when a pass over an odd number of layers
    ends at the top of the stack); the host pads the table (PFXK_DESC_PAD).  (n + pad) * 32 bytes = 4096 / 8192 for these n with the round-4 / round-5
    padding: a read past the padding would cross the allocation's page.  Both parities of the natural pass's length."""
    rng = np.random.default_rng(1000 + n)
    w, h = 200, 3
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    for k in range(n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.3, 0.3)
    stack[0, ..., 3] = 255
    modes = [k % 25 if k % 25 not in (OVERWRITE,) else 1 for k in range(n)]
    modes = [m if not (m == NORMAL) else 2 for m in modes]
    opac = [1.0 if k % 2 == 0 else 0.6 for k in range(n)]
    r = n - 9 if ((n - (n - 9)) % 2 == 1) == odd_pass else n - 10   # the one reset layer: the natural pass covers layers [r, n)
    modes[r] = OVERWRITE
    stack[r, ..., 3] = 255                                         # no holes: every unit starts its natural pass exactly at r
    modes[0] = NORMAL
    check(gpu, stack, modes, opac, f"{n} layers, natural pass of {n - r}")


@pytest.mark.parametrize("which", ["srt", "stream", "general"])
def test_in_place_flatten_on_every_kernel_and_partial_overlap_refused(gpu, which):
    """include/pfx.h: dst may BE one of the layers (in place) but must not partially overlap any.  In place on the class-sorting kernel (a reset layer), the plain
    streaming kernel (none) and the general kernel (a masked layer); a destination shifted into a layer by one row is PFX_ERR_INVALID"""
    from paintfe_amd._lib import PfxError
    rng = np.random.default_rng(31)
    w, h, n = 256, 80, 18
    stack, modes, opac = class_stack(rng, w, h, n, 6, 0.3, {})
    if which != "srt":
        modes = [m if m not in (OVERWRITE, NORMAL) else 1 for m in modes]
        modes[0] = NORMAL
        opac = [0.7 if (m == NORMAL and k) else o for k, (m, o) in enumerate(zip(modes, opac))]
    ref = O.flatten_stack(stack, np.asarray(modes, np.uint8), np.asarray(opac, np.float32))
    r = gpu.r
    big = r.dev_alloc(w * h * 4 * (n + 1))
    try:
        ptrs = [big + k * w * h * 4 for k in range(n)]
        for k in range(n):
            r.dev_upload(ptrs[k], stack[k])
        info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
        masks = None
        if which == "general":   # a live mask that conceals nothing (canvas_state.rs:660-665: alpha * (255 - 0) / 255) changes no pixel and moves the stack to the general kernel
            mbuf = r.dev_alloc(w * h)
            r.dev_upload(mbuf, np.zeros((h, w), np.uint8))
            masks = [0] * n
            masks[3] = mbuf
        with pytest.raises(PfxError):
            r.flatten_dev(ptrs, info, w, h, ptrs[4] + w * 4, mask_ptrs=masks)      # one row into layer 4 (and into layer 5 at its end)
        r.flatten_dev(ptrs, info, w, h, ptrs[9], mask_ptrs=masks)
        assert np.array_equal(r.dev_download(ptrs[9], (h, w, 4)), ref)
    finally:
        r.dev_free(big)
        if which == "general":
            r.dev_free(mbuf)


def test_shallow_stacks_take_the_elimination_kernel_only_where_a_probe_found_the_reset_layer_coherent():
    """Below the depth threshold (16 layers) pfx_flatten_dev decides per stack: the first composite runs the streaming kernel and a probe of the topmost reset
    layer's alpha behind it; from a later composite of the SAME stack on, the class-sorting kernel runs where at least half of the sampled units start at that layer
    outright (an opaque photo layer) and never where its alpha is per-pixel random.  Whatever runs, the result is the oracle's."""
    from paintfe_amd import GpuRenderer
    r = GpuRenderer(0)                      # its own context: default threshold, no verdicts yet
    rng = np.random.default_rng(99)
    w, h, n = 640, 360, 9
    stack = rng.integers(0, 256, (n, h, w, 4), dtype=np.uint8)
    for k in range(1, n):
        stack[k, ..., 3] = noise_alpha(rng, h, w, 0.3, 0.3)
    stack[0, ..., 3] = 255
    modes = [NORMAL, 1, 2, 8, 5, NORMAL, 1, 2, 8]
    opac = [1.0] * n
    bufs = [r.dev_alloc(w * h * 4) for _ in range(n + 1)]

    def composite_and_count(calls):
        ref = O.flatten_stack(stack, np.asarray(modes, np.uint8), np.asarray(opac, np.float32))
        for k in range(n):
            r.dev_upload(bufs[k], stack[k])
        info = [(k, float(opac[k]), True, int(modes[k])) for k in range(n)]
        used = []
        for _ in range(calls):
            r.flatten_stats(reset=True)
            r.flatten_dev(bufs[:n], info, w, h, bufs[n])
            assert np.array_equal(r.dev_download(bufs[n], (h, w, 4)), ref)
            used.append(r.flatten_stats(reset=True)["nat_units"] > 0)   # units the elimination kernel walked
        return used

    r.tune("dle_stats", 1)
    try:
        stack[5, ..., 3] = 255                                        # layer 5: an opaque photo (Normal at 100 %)
        used = composite_and_count(4)
        assert used[0] is False and used[-1] is True, used              # streaming first, elimination once the verdict is in
        stack[5, ..., 3] = noise_alpha(rng, h, w, 0.3, 0.3)           # the same stack, new pixels: the verdict is a hint, the result still exact
        assert composite_and_count(1) == [True]
        opac[6] = 0.5                                                  # another stack (a descriptor differs): probed afresh, per-pixel-random alpha: never
        assert composite_and_count(4) == [False, False, False, False]
        r.tune("dle_adaptive", 0)
        opac[6] = 1.0; stack[5, ..., 3] = 255
        assert composite_and_count(3) == [False, False, False]
    finally:
        r.tune("dle_stats", 0)
        for b in bufs:
            r.dev_free(b)
