// pfx_host_math.cpp — the pieces of the path that the reference itself computes on the host, once per call:
// tap weights, LUTs, scalar gains, displacement-brush scatter, stroke point lists.  They use glibc's
// expf/powf/sqrtf, which is what Rust's f32::exp/powf/sqrt call on Linux, so the device kernels receive
// bit-identical constants.  Compiled with -ffp-contract=off: every f32 operation rounds once, as in Rust.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdint>
#include <vector>

#include "k_brush_math.h"
#include "pfx_internal.h"

namespace {

inline uint8_t f32_as_u8(float v) // Rust `as u8`
{
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}
inline uint32_t f32_as_u32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}
inline int32_t f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)v;
}
inline uint8_t round_u8(float v) { return f32_as_u8(pfx_clampf(roundf(v), 0.0f, 255.0f)); }

} // namespace

int pfx_host_gaussian_radius(float sigma)
{
    const uint32_t radius = f32_as_u32(ceilf(sigma * 3.0f));
    return radius > 0x7fffffffu ? 0x7fffffff : (int)radius;
}

namespace {
// f32 -> IEEE binary16, round to nearest even (values here are finite, non-negative or tiny residuals of either sign)
uint16_t f32_to_f16_rn(float f)
{
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | 0x7bffu); // >= 65536: clamp to the largest finite value (never reached here)
    if (x < 0x38800000u) { // below 2^-14: subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign; // < 2^-25 rounds to zero
        const int e = (int)(x >> 23);
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e; // 14..24
        uint32_t q = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) ++q;
        return (uint16_t)(sign | q);
    }
    uint32_t q = (x - 0x38000000u) >> 13; // rebias exponent 127 -> 15, keep 10 mantissa bits
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) ++q;
    return (uint16_t)(sign | q);
}
float f16_to_f32(uint16_t hv)
{
    const uint32_t sign = (uint32_t)(hv & 0x8000u) << 16, e = (hv >> 10) & 31u, m = hv & 0x3ffu;
    float r;
    if (e == 0) r = std::ldexp((float)m, -24);
    else r = std::ldexp((float)(m | 0x400u), (int)e - 25);
    uint32_t b; std::memcpy(&b, &r, 4); b |= sign; std::memcpy(&r, &b, 4);
    return r;
}
} // namespace

// The matrix-core Gaussian (k_gauss.hip:gauss_strip_kernel) multiplies f16 operands with exact f32 products: each weight is
// scaled by S = 256 (the largest power of two for which S * 255, the scaled horizontal result, stays below the largest f16;
// weights are < 1, so w * S is in range too) and split w * S = w1 + w2, two f16: 22 significant bits, or an absolute 2^-25 where
// w2 is subnormal.  Returns 1 / S^2 (the two passes' scales, applied once at the end); *bias = 1024 * sum(w1 + w2), the constant
// the kernel's 0x6400 | byte sample encoding adds to every horizontal sum.
// Third table (out[2 * wlen ...]): every weight as ONE f16, w1' = RN(w * S) nudged by single ulps (symmetric pairs, the taps whose rounding
// error asks for it most first) until the table's sum is within one smallest ulp of sum(w * S) — a flat image then blurs to itself as with the
// split weights; what remains is a relative 2^-12 per tap with zero sum: at most 0.1 LSB per pass on an adversarial image (sum |delta| * 255 / S),
// ~1e-3 LSB on noise.  *bias_single = 1024 * sum(w1').
float pfx_host_gaussian_split_f16(const std::vector<float>& k, int wlen, int woff, std::vector<uint16_t>& out, float* bias, float* bias_single)
{
    const int s = 8;
    out.assign((size_t)3 * wlen, 0);
    double sum = 0.0;
    for (size_t t = 0; t < k.size() && (int)t + woff < wlen; ++t) {
        const float ws = std::ldexp(k[t], s);
        const uint16_t h1 = f32_to_f16_rn(ws);
        const uint16_t h2 = f32_to_f16_rn(ws - f16_to_f32(h1));
        out[(size_t)woff + t] = h1;
        out[(size_t)wlen + woff + t] = h2;
        sum += (double)f16_to_f32(h1) + (double)f16_to_f32(h2);
    }
    if (bias) *bias = (float)(1024.0 * sum);
    {
        const size_t n = std::min(k.size(), (size_t)std::max(wlen - woff, 0));
        std::vector<uint16_t> q(n);
        std::vector<double> want(n);
        double target = 0.0, have = 0.0;
        for (size_t t = 0; t < n; ++t) { want[t] = (double)std::ldexp(k[t], s); q[t] = f32_to_f16_rn((float)want[t]); target += want[t]; have += (double)f16_to_f32(q[t]); }
        auto ulp_up = [&](uint16_t hv) { return (double)f16_to_f32((uint16_t)(hv + 1)) - (double)f16_to_f32(hv); };
        auto ulp_dn = [&](uint16_t hv) { return hv ? (double)f16_to_f32(hv) - (double)f16_to_f32((uint16_t)(hv - 1)) : 0.0; };
        std::vector<char> moved(n, 0);
        for (int guard = 0; guard < 4096; ++guard) {
            const double e = target - have; // > 0: the table is short
            // candidate = a symmetric pair (t, n-1-t) (or the centre tap) not moved yet whose step brings the sum closer; the largest such step first,
            // among equal steps the tap whose own rounding error points the same way most strongly.  No tap ends more than one f16 step from w * S.
            double best_gain = 0.0, best_step = 0.0; size_t best = n;
            for (size_t t = 0; t * 2 < n + 1 && t < n; ++t) {
                const size_t u = n - 1 - t;
                const int mult = (u == t) ? 1 : 2;
                const double step = e > 0 ? ulp_up(q[t]) : -ulp_dn(q[t]);
                if (moved[t] || step == 0.0 || q[t] != q[u] || std::fabs(mult * step) >= 2.0 * std::fabs(e)) continue; // would not reduce |e|
                const double own = want[t] - (double)f16_to_f32(q[t]);
                const double gain = own * (e > 0 ? 1.0 : -1.0) / std::fabs(step);
                if (best == n || std::fabs(step) > std::fabs(best_step) || (std::fabs(step) == std::fabs(best_step) && gain > best_gain)) { best_gain = gain; best = t; best_step = step; }
            }
            if (best == n) break;
            const size_t u = n - 1 - best;
            q[best] = (uint16_t)(q[best] + (e > 0 ? 1 : -1)); have += best_step; moved[best] = 1;
            if (u != best) { q[u] = q[best]; have += best_step; moved[u] = 1; }
        }
        double sum1 = 0.0;
        for (size_t t = 0; t < n; ++t) { out[(size_t)2 * wlen + woff + t] = q[t]; sum1 += (double)f16_to_f32(q[t]); }
        if (bias_single) *bias_single = (float)(1024.0 * sum1);
    }
    return std::ldexp(1.0f, -2 * s);
}

// ref: build_gaussian_kernel, src/ops/filters.rs:214-234
int pfx_host_gaussian_kernel(float sigma, std::vector<float>& out)
{
    const uint32_t radius = f32_as_u32(ceilf(sigma * 3.0f));
    out.clear();
    if (radius == 0) { out.push_back(1.0f); return 0; }
    const size_t len = (size_t)radius * 2 + 1;
    out.resize(len);
    const float s2 = 2.0f * sigma * sigma;
    float sum = 0.0f;
    for (size_t i = 0; i < len; ++i) {
        const float x = (float)i - (float)radius;
        const float v = expf(-x * x / s2);
        out[i] = v;
        sum += v;
    }
    const float inv = 1.0f / sum;
    for (float& v : out) v *= inv;
    return (int)radius;
}

// ref: src/ops/adjustments.rs:273 / src/canvas/layers.rs:293
float pfx_host_bc_factor(float contrast) { return (259.0f * (contrast + 255.0f)) / (255.0f * (259.0f - contrast)); }
// ref: src/ops/adjustments.rs:353 (`2.0f32.powf(exposure)`)
float pfx_host_exposure_gain(float ev) { return powf(2.0f, ev); }

extern "C" {

int pfx_gaussian_f16_tables(float sigma, uint16_t out_768[768], float* bias_split, float* bias_single)
{
    if (!out_768 || !(sigma > 0.0f)) return PFX_ERR_INVALID;
    const int radius = pfx_host_gaussian_radius(sigma);
    if (radius < 1 || radius > 80) return PFX_ERR_UNSUPPORTED;
    std::vector<float> k;
    pfx_host_gaussian_kernel(sigma, k);
    std::vector<uint16_t> ws;
    float b2 = 0.0f, b1 = 0.0f;
    pfx_host_gaussian_split_f16(k, 256, 48, ws, &b2, &b1);
    std::copy(ws.begin(), ws.end(), out_768);
    if (bias_split) *bias_split = b2;
    if (bias_single) *bias_single = b1;
    return (int)k.size();
}

// ref: build_levels_lut, src/ops/adjustments.rs:465-488
void pfx_build_levels_lut(float in_black, float in_white, float gamma, float out_black, float out_white, uint8_t lut[256])
{
    if (!lut) return;
    const float in_range = fmaxf(in_white - in_black, 1.0f);
    const float out_range = out_white - out_black;
    const float inv_gamma = 1.0f / fmaxf(gamma, 0.01f);
    for (int i = 0; i < 256; ++i) {
        const float normalized = pfx_clampf(((float)i - in_black) / in_range, 0.0f, 1.0f);
        const float gamma_corrected = powf(normalized, inv_gamma);
        lut[i] = round_u8(out_black + gamma_corrected * out_range);
    }
}

// ref: build_stretch_lut, src/ops/adjustments.rs:235-256
void pfx_build_stretch_lut(uint8_t mn, uint8_t mx, uint8_t lut[256])
{
    if (!lut) return;
    if (mx <= mn) { for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i; return; }
    const float range = (float)(mx - mn);
    for (int i = 0; i < 256; ++i) {
        float v;
        if ((uint8_t)i <= mn) v = 0.0f;
        else if ((uint8_t)i >= mx) v = 255.0f;
        else v = ((float)i - (float)mn) / range * 255.0f;
        lut[i] = round_u8(v);
    }
}

// Curves LUT: a monotone piecewise-cubic Hermite interpolant through the control points (Fritsch–Carlson limiter on the tangents), sampled at the 256 byte
// values.  Reference: build_curves_lut, src/ops/adjustments.rs:640-729 — the operation ORDER below is that function's (it is the rounding contract, pinned by
// tests/golden/curves_kat.json's 60-digit known answers); the organisation is ours: secant slopes, limited tangents, then one Hermite evaluation per byte.
namespace {
struct curve_knots {
    const float* p; uint32_t n;
    float x(uint32_t i) const { return p[2 * i]; }
    float y(uint32_t i) const { return p[2 * i + 1]; }
};
constexpr float CURVE_EPS = 1e-6f;
// slope of the chord between knots i and i + 1 (0 for a vertical chord)
float chord_slope(const curve_knots& K, uint32_t i)
{
    const float run = K.x(i + 1) - K.x(i), rise = K.y(i + 1) - K.y(i);
    return (fabsf(run) < CURVE_EPS) ? 0.0f : rise / run;
}
// cubic Hermite on [xa, xb] with end values ya, yb and end tangents ta, tb
float hermite_at(float x, float xa, float xb, float ya, float yb, float ta, float tb)
{
    const float span = xb - xa;
    if (fabsf(span) < CURVE_EPS) return ya;
    const float u = (x - xa) / span, u2 = u * u, u3 = u2 * u;
    const float w_ya = 2.0f * u3 - 3.0f * u2 + 1.0f, w_ta = u3 - 2.0f * u2 + u;
    const float w_yb = -2.0f * u3 + 3.0f * u2, w_tb = u3 - u2;
    return w_ya * ya + w_ta * span * ta + w_yb * yb + w_tb * span * tb;
}
}  // namespace

void pfx_build_curves_lut(const float* pts, uint32_t n, uint8_t lut[256])
{
    if (!lut) return;
    if (!pts || n < 2) { for (int i = 0; i < 256; ++i) lut[i] = (uint8_t)i; return; }
    const curve_knots K{pts, n};
    const uint32_t last = n - 1;
    std::vector<float> chord(last), tangent(n, 0.0f);
    for (uint32_t i = 0; i < last; ++i) chord[i] = chord_slope(K, i);
    // interior tangents: mean of the neighbouring chords, zero at a local extremum; ends take their one chord
    tangent[0] = chord[0];
    tangent[last] = chord[last - 1];
    for (uint32_t i = 1; i < last; ++i) {
        const float left = chord[i - 1], right = chord[i];
        tangent[i] = (left * right <= 0.0f) ? 0.0f : (left + right) / 2.0f;
    }
    // limiter: keep (tangent / chord) of both ends of a segment inside the circle of radius 3
    for (uint32_t i = 0; i < last; ++i) {
        const float c = chord[i];
        if (fabsf(c) < CURVE_EPS) { tangent[i] = 0.0f; tangent[i + 1] = 0.0f; continue; }
        const float ra = tangent[i] / c, rb = tangent[i + 1] / c;
        const float r2 = ra * ra + rb * rb;
        if (r2 > 9.0f) {
            const float shrink = 3.0f / sqrtf(r2);
            tangent[i] = shrink * ra * c;
            tangent[i + 1] = shrink * rb * c;
        }
    }
    for (int i = 0; i < 256; ++i) {
        const float x = (float)i;
        float v;
        if (x <= K.x(0)) v = K.y(0);
        else if (x >= K.x(last)) v = K.y(last);
        else {
            uint32_t seg = 0;   // the last segment whose left knot is not right of x
            for (uint32_t j = last; j-- > 0;) if (x >= K.x(j)) { seg = j; break; }
            v = hermite_at(x, K.x(seg), K.x(seg + 1), K.y(seg), K.y(seg + 1), tangent[seg], tangent[seg + 1]);
        }
        lut[i] = round_u8(v);
    }
}

// ref: DisplacementField::apply_push / apply_expand / apply_contract / apply_twirl, src/ops/transform.rs:1051-1200
void pfx_displacement_brush(float* disp, uint32_t w, uint32_t h, int mode, float cx, float cy, float delta_x, float delta_y,
                            float radius, float strength)
{
    if (!disp) return;
    const float r = fmaxf(radius, 1.0f);
    const float sigma = r / 3.0f;
    const float sigma_sq_2 = 2.0f * sigma * sigma;
    const float dir = (mode == 3) ? 1.0f : -1.0f;
    const int32_t x0 = std::max(f32_as_i32(floorf(cx - r)), 0), y0 = std::max(f32_as_i32(floorf(cy - r)), 0);
    const int32_t x1 = std::min(f32_as_i32(ceilf(cx + r)), (int32_t)w), y1 = std::min(f32_as_i32(ceilf(cy + r)), (int32_t)h);
    for (int32_t py = y0; py < y1; ++py)
        for (int32_t px = x0; px < x1; ++px) {
            const float dx = (float)px - cx, dy = (float)py - cy;
            const float dist_sq = dx * dx + dy * dy;
            if (dist_sq > r * r) continue;
            float* d = disp + ((size_t)py * w + (size_t)px) * 2;
            if (mode == 0) {
                const float weight = expf(-dist_sq / sigma_sq_2) * strength;
                d[0] += delta_x * weight;
                d[1] += delta_y * weight;
            } else if (mode == 1) {
                const float dist = fmaxf(sqrtf(dist_sq), 0.001f);
                const float t = dist / r;
                const float weight = (1.0f - t) * (1.0f - t) * strength * 3.0f;
                d[0] += dx / dist * weight;
                d[1] += dy / dist * weight;
            } else if (mode == 2) {
                const float dist = fmaxf(sqrtf(dist_sq), 0.001f);
                const float weight = expf(-dist_sq / sigma_sq_2) * strength;
                d[0] += -dx / dist * weight * 2.0f;
                d[1] += -dy / dist * weight * 2.0f;
            } else {
                const float weight = expf(-dist_sq / sigma_sq_2) * strength * dir;
                d[0] += -dy * weight * 0.1f;
                d[1] += dx * weight * 0.1f;
            }
        }
}

} // extern "C"

// ref: rebuild_brush_lut, src/ui/panels/tools/behavior/raster/brush_render.rs:27-50
void pfx_host_brush_lut(float size, float hardness, bool anti_aliased, uint8_t lut[256])
{
    const float radius = size / 2.0f;
    if (radius < 0.001f) { for (int i = 0; i < 256; ++i) lut[i] = 0; return; }
    for (int i = 0; i < 256; ++i) {
        const float t_sq = (float)i / 255.0f;
        const float dist = sqrtf(t_sq) * radius;
        const float alpha = pfx_brush_alpha(dist, radius, hardness, anti_aliased);
        lut[i] = f32_as_u8(fminf(roundf(alpha * 255.0f), 255.0f));
    }
}

// ref: draw_line_no_dirty, src/ui/panels/tools/behavior/raster/brush_render.rs:762-835 (circle tip: step 1.0)
void pfx_host_line_points(float x0, float y0, float x1, float y1, uint32_t width, uint32_t height, std::vector<float>& out)
{
    out.clear();
    const float dx = x1 - x0, dy = y1 - y0;
    const float distance = sqrtf(dx * dx + dy * dy);
    auto inside = [&](float x, float y) { return x >= 0.0f && f32_as_u32(x) < width && y >= 0.0f && f32_as_u32(y) < height; };
    if (distance < 0.1f) {
        if (inside(x0, y0)) { out.push_back(x0); out.push_back(y0); }
        return;
    }
    const size_t steps = (size_t)f32_as_u32(ceilf(distance / 1.0f));
    // The reference walks all `steps` + 1 stamps and keeps those inside the image.  A line that leaves the image far behind (a coordinate of 1e30: 2^32 steps)
    // would spend its time outside, so long lines walk only the part of [0, steps] whose stamps can be inside: the parameter interval in which both coordinates
    // are within one pixel of the image, computed in double, widened by the f32 granularity of `i / steps` (2^-20 of the range, and two steps).  Same stamps,
    // same order; non-finite coordinates give an empty interval, as every `inside` test of the reference's walk fails on them.
    size_t i_lo = 0, i_hi = steps;
    if (steps > 65536) {
        double t_lo = 0.0, t_hi = 1.0;
        bool empty = false;
        auto keep = [&](double p0, double d, double limit) {
            if (!(d == d) || !(p0 == p0)) { empty = true; return; }
            if (d == 0.0) { if (p0 < -1.0 || p0 > limit + 1.0) empty = true; return; }
            double a = (-1.0 - p0) / d, b = (limit + 1.0 - p0) / d;
            if (a > b) std::swap(a, b);
            if (!(a == a) || !(b == b)) { empty = true; return; }   // inf / inf
            t_lo = std::max(t_lo, a);
            t_hi = std::min(t_hi, b);
        };
        keep((double)x0, (double)dx, (double)width);
        keep((double)y0, (double)dy, (double)height);
        if (empty || !(t_lo <= t_hi)) return;
        const double slack = (double)steps / 1048576.0 + 2.0;
        const double lo = std::floor(t_lo * (double)steps) - slack, hi = std::ceil(t_hi * (double)steps) + slack;
        i_lo = lo <= 0.0 ? 0 : (size_t)lo;
        i_hi = hi >= (double)steps ? steps : (size_t)hi;
    }
    for (size_t i = i_lo; i <= i_hi; ++i) {
        const float t = (float)i / (float)steps;
        const float x = x0 + dx * t, y = y0 + dy * t;
        if (inside(x, y)) { out.push_back(x); out.push_back(y); }
    }
}

// ref: apply_levels LUT, src/ops/scripting.rs:1050-1061 (truncating, output range 0..255)
void pfx_host_rhai_levels_lut(float in_black, float in_white, float gamma, uint8_t lut[256])
{
    const float in_range = fmaxf(in_white - in_black, 1.0f);
    const float inv_gamma = 1.0f / fmaxf(gamma, 0.01f);
    for (int i = 0; i < 256; ++i) {
        const float normalized = pfx_clampf(((float)i - in_black) / in_range, 0.0f, 1.0f);
        lut[i] = f32_as_u8(pfx_clampf(powf(normalized, inv_gamma) * 255.0f, 0.0f, 255.0f));
    }
}
