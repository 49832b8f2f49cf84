"""Deterministic test-image generators.

The small ones restate the reference's test helpers (no RNG, no files) so that the
committed golden images can be regenerated bit for bit:
  tests/common/mod.rs:272-340    create_test_gradient / checkerboard / solid / transparent / color_bands
  tests/visual_blend.rs:27-36    blend foreground
  tests/transform_ops.rs:25-44   gradient_32 / uniform_grid
The seeded synthetic generators (S1/S2/S3 of SURVEY.md §8d) feed the large-size property tests and bench.py.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def create_test_gradient(w: int, h: int) -> np.ndarray:
    img = np.zeros((h, w, 4), np.uint8)
    x = np.arange(w, dtype=np.uint32)
    y = np.arange(h, dtype=np.uint32)
    r = (x * 255 // (w - 1)).astype(np.uint8) if w > 1 else np.full(w, 128, np.uint8)
    b = (y * 255 // (h - 1)).astype(np.uint8) if h > 1 else np.full(h, 128, np.uint8)
    img[..., 0] = r[None, :]
    img[..., 1] = (255 - r)[None, :]
    img[..., 2] = b[:, None]
    img[..., 3] = 255
    return img


def create_test_checkerboard(w: int, h: int) -> np.ndarray:
    x = np.arange(w) // 8
    y = np.arange(h) // 8
    white = ((x[None, :] + y[:, None]) % 2) == 0
    img = np.zeros((h, w, 4), np.uint8)
    img[..., :3] = np.where(white, 255, 0)[..., None]
    img[..., 3] = 255
    return img


def create_solid(w: int, h: int, color) -> np.ndarray:
    img = np.zeros((h, w, 4), np.uint8)
    img[...] = np.asarray(color, np.uint8)
    return img


def create_transparent(w: int, h: int) -> np.ndarray:
    return np.zeros((h, w, 4), np.uint8)


def create_color_bands(w: int, h: int) -> np.ndarray:
    colors = np.array([[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [0, 255, 255, 255],
                       [255, 0, 255, 255], [255, 255, 0, 255], [255, 255, 255, 255], [0, 0, 0, 255]], np.uint8)
    band = np.minimum(np.arange(w) * 8 // w, 7)
    return np.broadcast_to(colors[band][None, :, :], (h, w, 4)).copy()


def blend_foreground(w: int = 64, h: int = 64) -> np.ndarray:
    """tests/visual_blend.rs:27-36 (f32 arithmetic, truncating casts)."""
    x = np.arange(w, dtype=f32)[None, :]
    y = np.arange(h, dtype=f32)[:, None]
    img = np.zeros((h, w, 4), np.uint8)
    img[..., 0] = np.broadcast_to(((x / f32(w)) * f32(255.0)).astype(np.uint8), (h, w))
    img[..., 1] = np.broadcast_to(((y / f32(h)) * f32(255.0)).astype(np.uint8), (h, w))
    img[..., 2] = 128
    xi = np.arange(w, dtype=np.uint32)[None, :]
    yi = np.arange(h, dtype=np.uint32)[:, None]
    s = (xi + yi).astype(f32)
    img[..., 3] = ((s / f32(w + h - 2)) * f32(200.0) + f32(55.0)).astype(np.uint8)
    return img


def gradient_32() -> np.ndarray:
    img = np.zeros((32, 32, 4), np.uint8)
    v = (np.arange(32) * 8).astype(np.uint8)
    img[..., 0] = v[None, :]
    img[..., 1] = v[:, None]
    img[..., 2] = 128
    img[..., 3] = 255
    return img


def uniform_grid(cols: int, rows: int, w: float, h: float) -> np.ndarray:
    pts = np.zeros((rows + 1, cols + 1, 2), f32)
    for r in range(rows + 1):
        for c in range(cols + 1):
            pts[r, c, 0] = f32(c) / f32(cols) * f32(w)
            pts[r, c, 1] = f32(r) / f32(rows) * f32(h)
    return pts.reshape(-1, 2)


# ---------------------------------------------------------------------------
# Seeded synthetic inputs.  Pure numpy, identical on every box.
# ---------------------------------------------------------------------------
def random_rgba(w: int, h: int, seed: int) -> np.ndarray:
    """S1: uniform random bytes in all four channels."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


def layer_stack(w: int, h: int, n: int, seed: int = 0x5EED0002):
    """S2: n layers; alpha 25% = 0, 25% = 255, 50% uniform 1..254; layer 0 Normal opaque background;
    mode = k mod 25; opacity 1.0 for even k, 0.25+0.75u for odd k."""
    stack = np.empty((n, h, w, 4), np.uint8)
    modes = np.zeros(n, np.uint8)
    opac = np.ones(n, f32)
    for k in range(n):
        rng = np.random.default_rng(seed + k)
        px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        sel = rng.integers(0, 4, size=(h, w), dtype=np.uint8)
        a = rng.integers(1, 255, size=(h, w), dtype=np.uint8)
        a = np.where(sel == 0, 0, np.where(sel == 1, 255, a)).astype(np.uint8)
        px[..., 3] = 255 if k == 0 else a
        stack[k] = px
        modes[k] = k % 25
        if k % 2 == 1:
            opac[k] = f32(0.25) + f32(0.75) * f32(rng.random())
    return stack, modes, opac


def jittered_mesh(cols: int, rows: int, w: int, h: int, seed: int = 0x5EED0004):
    """S3: uniform lattice + per-point jitter ±w/24, ±h/24, borders pinned."""
    orig = uniform_grid(cols, rows, w, h).reshape(rows + 1, cols + 1, 2)
    rng = np.random.default_rng(seed)
    jit = (rng.random((rows + 1, cols + 1, 2)).astype(f32) * f32(2) - f32(1)) * np.array([w / 24, h / 24], f32)
    jit[0, :, :] = 0
    jit[-1, :, :] = 0
    jit[:, 0, :] = 0
    jit[:, -1, :] = 0
    return orig.reshape(-1, 2).copy(), (orig + jit).reshape(-1, 2).astype(f32)
