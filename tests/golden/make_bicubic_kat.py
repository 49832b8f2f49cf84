#!/usr/bin/env python3
"""Known-answer vector for the bicubic (Catmull-Rom) resize, from the published kernel in exact rational arithmetic — independent
of the f32 implementations.

Kernel (Mitchell-Netravali family, B = 0, C = 1/2 = Catmull-Rom; the `image` crate's `catmullrom_kernel`):
    k(x) = 1.5|x|^3 - 2.5|x|^2 + 1            |x| < 1
           -0.5|x|^3 + 2.5|x|^2 - 4|x| + 2    1 <= |x| < 2,   0 otherwise
Sampling scheme of `image::imageops::resize` (0.25, `sample.rs`: vertical pass into f32, then horizontal): for output index o of an
axis resized n_in -> n_out, ratio = n_in / n_out, s = max(ratio, 1), centre c = (o + 0.5) * ratio, window
left = clamp(floor(c - 2 s), 0, n_in - 1), right = clamp(ceil(c + 2 s), left + 1, n_in), weights k((i - (c - 0.5)) / s) normalised
to sum 1; the result is rounded half away from zero and clamped to 0..255.

The test image is a 6 x 5 single-colour-ramp RGBA image upscaled to 11 x 9; entries within 0.02 of a rounding boundary are left
out (an f32 evaluation may legitimately land on either side).  Output: tests/golden/bicubic_kat.json."""
import json
import math
import os
from fractions import Fraction as F


def k(x):
    a = abs(x)
    if a < 1:
        return F(3, 2) * a ** 3 - F(5, 2) * a ** 2 + 1
    if a < 2:
        return -F(1, 2) * a ** 3 + F(5, 2) * a ** 2 - 4 * a + 2
    return F(0)


def weights(n_in, n_out):
    ratio = F(n_in, n_out)
    s = max(ratio, F(1))
    out = []
    for o in range(n_out):
        c = (F(o) + F(1, 2)) * ratio
        left = min(max(math.floor(c - 2 * s), 0), n_in - 1)
        right = min(max(math.ceil(c + 2 * s), left + 1), n_in)
        cc = c - F(1, 2)
        ws = [k((F(i) - cc) / s) for i in range(left, right)]
        tot = sum(ws)
        out.append((left, [w / tot for w in ws]))
    return out


def main():
    w, h, nw, nh = 6, 5, 11, 9
    img = [[[(37 * x + 11 * y * y + 5) % 256, (200 - 23 * x + 7 * y) % 256, (x * y * 13 + 90) % 256, 255 - (17 * x + 29 * y) % 120] for x in range(w)]
           for y in range(h)]
    wv, wh = weights(h, nh), weights(w, nw)
    # vertical pass (exact), then horizontal pass (exact); the crate rounds only at the very end for u8 output of the second pass,
    # the first pass keeps f32 — exact arithmetic stands in for both
    tmp = [[[sum(wt * img[l + i][x][c] for i, wt in enumerate(ws)) for c in range(4)] for x in range(w)] for (l, ws) in wv]
    res = {}
    kept = 0
    for y in range(nh):
        for x in range(nw):
            l, ws = wh[x]
            for c in range(4):
                v = sum(wt * tmp[y][l + i][c] for i, wt in enumerate(ws))
                v = min(max(v, F(0)), F(255))
                frac = v - math.floor(v)
                if abs(frac - F(1, 2)) < F(2, 100):
                    continue
                res[f"{y},{x},{c}"] = int(math.floor(v + F(1, 2)))
                kept += 1
    print(kept, "robust entries of", nw * nh * 4)
    json.dump({"w": w, "h": h, "nw": nw, "nh": nh, "image": img, "expected": res},
              open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bicubic_kat.json"), "w"))


if __name__ == "__main__":
    main()
