#!/usr/bin/env python3
"""Known-answer vectors for non-identity Curves LUTs, derived from the published Fritsch-Carlson monotone cubic Hermite
scheme in 60-digit decimal arithmetic — independent of every f32 implementation (the reference, the oracle, the product).

The reference's variant (src/ops/adjustments.rs:640-729): secant slopes d_i; interior tangents m_i = (d_{i-1} + d_i) / 2, or 0
where the secants change sign; end tangents = the end secants; for every interval with alpha = m_i / d_i, beta = m_{i+1} / d_i
and alpha^2 + beta^2 > 9 both tangents are scaled by tau = 3 / sqrt(alpha^2 + beta^2); cubic Hermite evaluation at integer x;
`.round()` (half away from zero), clamped to 0..255.

An entry is kept as a known answer only if its exact value is at least 0.01 away from a rounding boundary, so an f32 evaluation
that follows the same formulas cannot legitimately round differently.  Output: tests/golden/curves_kat.json.
"""
import json
import os
from decimal import Decimal as D, getcontext

getcontext().prec = 60

CASES = {
    "gentle_s_curve": [(0, 0), (64, 32), (128, 160), (255, 255)],
    "limiter_active": [(0, 0), (100, 10), (120, 200), (255, 255)],          # alpha^2 + beta^2 > 9 on the first two intervals
    "non_monotone_flat_tangents": [(0, 40), (80, 200), (160, 60), (255, 230)],  # secants change sign: interior tangents are 0
    "clipped_ends": [(30, 0), (128, 128), (220, 255)],                       # x outside the points takes the end values
    "two_points_inverse": [(0, 255), (255, 0)],
}


def lut_exact(points):
    pts = [(D(x), D(y)) for x, y in points]
    n = len(pts)
    delta = [(pts[i + 1][1] - pts[i][1]) / (pts[i + 1][0] - pts[i][0]) for i in range(n - 1)]
    m = [D(0)] * n
    m[0], m[n - 1] = delta[0], delta[n - 2]
    for i in range(1, n - 1):
        m[i] = D(0) if delta[i - 1] * delta[i] <= 0 else (delta[i - 1] + delta[i]) / 2
    for i in range(n - 1):
        if delta[i] == 0:
            m[i] = m[i + 1] = D(0)
        else:
            a, b = m[i] / delta[i], m[i + 1] / delta[i]
            s = a * a + b * b
            if s > 9:
                tau = D(3) / s.sqrt()
                m[i], m[i + 1] = tau * a * delta[i], tau * b * delta[i]
    out = []
    for i in range(256):
        x = D(i)
        if x <= pts[0][0]:
            v = pts[0][1]
        elif x >= pts[-1][0]:
            v = pts[-1][1]
        else:
            seg = max(j for j in range(n - 1) if x >= pts[j][0])
            (x0, y0), (x1, y1) = pts[seg], pts[seg + 1]
            h = x1 - x0
            t = (x - x0) / h
            t2, t3 = t * t, t * t * t
            v = (2 * t3 - 3 * t2 + 1) * y0 + (t3 - 2 * t2 + t) * h * m[seg] + (-2 * t3 + 3 * t2) * y1 + (t3 - t2) * h * m[seg + 1]
        out.append(v)
    return out


def main():
    res = {}
    for name, pts in CASES.items():
        vals = lut_exact(pts)
        entries = {}
        for i, v in enumerate(vals):
            frac = v - v.to_integral_value(rounding="ROUND_FLOOR")
            if abs(frac - D("0.5")) < D("0.01"):
                continue  # too close to a rounding boundary to be a fair known answer for f32 arithmetic
            r = int((v + D("0.5")).to_integral_value(rounding="ROUND_FLOOR")) if v >= 0 else -int((-v + D("0.5")).to_integral_value(rounding="ROUND_FLOOR"))
            entries[str(i)] = max(0, min(255, r))
        res[name] = {"points": pts, "lut": entries}
        print(name, len(entries), "robust entries of 256")
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "curves_kat.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
