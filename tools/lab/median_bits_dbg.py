#!/usr/bin/env python3
"""mismatch pattern of the bit-plane median against the oracle (development aid)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import oracle_lib as O
from tests import inputs as I
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
r.tune("median_bits_min", 2)
for (w, h, rad) in [(40, 20, 3), (131, 77, 3), (40, 20, 7), (40, 20, 2)]:
    img = I.random_rgba(w, h, 5)
    got = r.median_core(img, rad, None)
    ref = O.median(img, rad)
    bad = (got != ref)
    print(w, h, rad, "mismatch", bad.sum(), "of", bad.size)
    if bad.any():
        print(" per channel", bad.sum(axis=(0, 1)))
        print(" per row", bad.sum(axis=(1, 2))[:40])
        print(" per col", bad.sum(axis=(0, 2))[:70])
        ys, xs, cs = np.nonzero(bad)
        for i in range(min(6, len(ys))):
            print("  ", ys[i], xs[i], cs[i], "got", got[ys[i], xs[i], cs[i]], "ref", ref[ys[i], xs[i], cs[i]])
