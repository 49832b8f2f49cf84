#!/usr/bin/env python3
"""share of channels where the matrix-core Gaussian differs from the CPU path (by 1 LSB) per sigma"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import oracle_lib as O, inputs as I
from paintfe_amd import GpuRenderer
r = GpuRenderer(0)
img = I.random_rgba(1024, 512, 5)
for sigma in (4.0, 16.0, 17.0, 20.0, 24.0, 26.6):
    ref = O.gaussian_blur(img, sigma); out = r.blur_rgba(img, sigma)
    d = np.abs(out.astype(np.int16) - ref.astype(np.int16))
    print(f"sigma={sigma}: max diff {d.max()}, differing channels {(d > 0).mean():.2e}")
