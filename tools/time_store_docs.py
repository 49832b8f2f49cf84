#!/usr/bin/env python3
"""tools/time_store_docs.py — the layer-store compositor (pfx_composite) on document-like 8K stacks, with and without the per-chunk start
table built from the stored layers' alpha summaries (pfx_tune "chunk_start")."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
w, h, n = 7680, 4320, 9
rng = np.random.default_rng(1)
base = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
def layer(k, kind):
    img = np.roll(base, 977 * k, axis=1).copy()
    a = rng.integers(0, 4, (h // 8, w // 8))
    a = np.kron(a, np.ones((8, 8), np.uint8))
    img[..., 3] = np.where(a == 0, 0, np.where(a == 1, 255, img[..., 3]))
    if kind == "opaque": img[..., 3] = 255
    if kind == "half": img[:, : w // 2, 3] = 255
    return img
docs = {"photo (opaque) at layer 4 of 9": ["opaque" if k in (0, 4) else "mixed" for k in range(n)],
        "photo covering the left half at layer 6": ["opaque" if k == 0 else ("half" if k == 6 else "mixed") for k in range(n)],
        "no covering layer": ["opaque" if k == 0 else "mixed" for k in range(n)]}
modes = [0, 1, 2, 8, 0, 3, 0, 15, 0]
for name, kinds in docs.items():
    r.clear_layers()
    for k in range(n):
        r.ensure_layer_texture(k, layer(k, kinds[k]), generation=1)
    info = [(k, 1.0 if modes[k] == 0 else 0.7, True, modes[k], 0, ()) for k in range(n)]
    out = []
    for on in (1, 0):
        r.tune("chunk_start", on)
        for _ in range(3): res = r.composite(w, h, info)
        r.timing_reset(); r.timing_enable(True)
        for _ in range(5): res = r.composite(w, h, info)
        r.timing_enable(False)
        out.append((r.timing_read("flatten")[0] / 5, res))
    r.tune("chunk_start", 1)
    same = np.array_equal(out[0][1], out[1][1])
    print(f"{name:45s} table {out[0][0]:.4f} ms   without {out[1][0]:.4f} ms   ratio {out[0][0] / out[1][0]:.3f}   identical {same}")
