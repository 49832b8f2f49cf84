/*
 * o_resize.c — oracle restatement of `image::imageops::resize` as the reference calls it for resize_image
 * (src/ops/transform.rs:347-359, src/ops/scripting.rs:749-770).  TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).
 *
 * The algorithm lives in the third-party crate `image` 0.25.9 (Cargo.lock), which is not under the reference tree; this
 * restates its published algorithm (imageops/sample.rs: `resize` = vertical_sample into an f32 image, then
 * horizontal_sample with clamp + round-to-nearest; per output sample: centre (o + 0.5) * ratio, window
 * [floor(c - support*sratio), ceil(c + support*sratio)) clamped to the image, weights kernel((i - (c - 0.5)) / sratio)
 * normalised by their f32 running sum, accumulation `t += v * w` in source order).  Pinned by the reference's goldens
 * transforms/resize_2x_nearest, resize_half_bilinear and resize_half_lanczos (tests/visual_transforms.rs:134-164) with
 * tolerance 0; the Catmull-Rom (bicubic) kernel has no golden in the reference; it is pinned by tests/golden/bicubic_kat.json
 * (exact-rational evaluation of the published kernel, tests/golden/make_bicubic_kat.py).
 */
#include "o_common.h"

#define PI_F 3.14159265358979323846f

static float k_box(float x) { (void)x; return 1.0f; }
static float k_triangle(float x) { float a = fabsf(x); return a < 1.0f ? 1.0f - a : 0.0f; }
static float sinc(float t)
{
    float a = t * PI_F;
    return t == 0.0f ? 1.0f : sinf(a) / a;
}
static float k_lanczos3(float x) { return fabsf(x) < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f; }
static float k_catmullrom(float x) /* bc_cubic_spline(x, b = 0, c = 0.5) */
{
    const float b = 0.0f, c = 0.5f;
    float a = fabsf(x), k;
    if (a < 1.0f) k = (12.0f - 9.0f * b - 6.0f * c) * (a * a * a) + (-18.0f + 12.0f * b + 6.0f * c) * (a * a) + (6.0f - 2.0f * b);
    else if (a < 2.0f) k = (-b - 6.0f * c) * (a * a * a) + (6.0f * b + 30.0f * c) * (a * a) + (-12.0f * b - 48.0f * c) * a + (8.0f * b + 24.0f * c);
    else k = 0.0f;
    return k / 6.0f;
}

typedef float (*kernel_fn)(float);
static kernel_fn pick(int filter, float* support)
{
    switch (filter) {
    case PFXO_RESIZE_NEAREST: *support = 0.0f; return k_box;
    case PFXO_RESIZE_BICUBIC: *support = 2.0f; return k_catmullrom;
    case PFXO_RESIZE_LANCZOS3: *support = 3.0f; return k_lanczos3;
    default: *support = 1.0f; return k_triangle;
    }
}

/* weights of one axis: for each output index `left[o]`, `count[o]` and the normalised weights at wts + off[o].  Returns the
 * total number of weights (call with wts == NULL to size the buffer). */
size_t pfxo_resize_weights(uint32_t n_in, uint32_t n_out, int filter, uint32_t* left, uint32_t* count, size_t* off, float* wts)
{
    float support;
    kernel_fn kernel = pick(filter, &support);
    float ratio = (float)n_in / (float)n_out;
    float sratio = ratio < 1.0f ? 1.0f : ratio;
    float src_support = support * sratio;
    size_t total = 0;
    for (uint32_t o = 0; o < n_out; ++o) {
        float input = ((float)o + 0.5f) * ratio;
        int64_t l = (int64_t)floorf(input - src_support);
        if (l < 0) l = 0;
        if (l > (int64_t)n_in - 1) l = (int64_t)n_in - 1;
        int64_t r = (int64_t)ceilf(input + src_support);
        if (r < l + 1) r = l + 1;
        if (r > (int64_t)n_in) r = (int64_t)n_in;
        input = input - 0.5f;
        if (left) { left[o] = (uint32_t)l; count[o] = (uint32_t)(r - l); off[o] = total; }
        if (wts) {
            float sum = 0.0f;
            for (int64_t i = l; i < r; ++i) {
                float w = kernel(((float)i - input) / sratio);
                wts[total + (size_t)(i - l)] = w;
                sum += w;
            }
            for (int64_t i = l; i < r; ++i) wts[total + (size_t)(i - l)] /= sum;
        }
        total += (size_t)(r - l);
    }
    return total;
}

/* imageops::flip_horizontal / flip_vertical / rotate90 / rotate270 / rotate180 (pure permutations); op = CanvasOpRequest order:
 * 0 flip h, 1 flip v, 2 rotate 90 cw, 3 rotate 90 ccw, 4 rotate 180.  dst is h x w for the 90-degree rotations. */
void pfxo_flip_rotate(const uint8_t* src, uint32_t w, uint32_t h, int op, uint8_t* dst)
{
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            uint32_t ox, oy, ow = w;
            switch (op) {
            case 0: ox = w - 1 - x; oy = y; break;
            case 1: ox = x; oy = h - 1 - y; break;
            case 2: ox = h - 1 - y; oy = x; ow = h; break;  /* rotate90: out.put_pixel(height - 1 - y, x, p) */
            case 3: ox = y; oy = w - 1 - x; ow = h; break;  /* rotate270: out.put_pixel(y, width - 1 - x, p) */
            default: ox = w - 1 - x; oy = h - 1 - y; break;
            }
            memcpy(dst + ((size_t)oy * ow + ox) * 4, src + ((size_t)y * w + x) * 4, 4);
        }
}

/* resize_canvas (src/ops/transform.rs:382-424; script flavour src/ops/scripting.rs:781-818 = transparent fill) */
void pfxo_resize_canvas(const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint32_t ax, uint32_t ay, const uint8_t fill[4], uint8_t* dst)
{
    int32_t off_x = ax == 0 ? 0 : (ax == 1 ? ((int32_t)nw - (int32_t)w) / 2 : (int32_t)nw - (int32_t)w);
    int32_t off_y = ay == 0 ? 0 : (ay == 1 ? ((int32_t)nh - (int32_t)h) / 2 : (int32_t)nh - (int32_t)h);
    for (size_t i = 0; i < (size_t)nw * nh; ++i) memcpy(dst + i * 4, fill, 4);
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            int32_t nx = (int32_t)x + off_x, ny = (int32_t)y + off_y;
            if (nx >= 0 && ny >= 0 && (uint32_t)nx < nw && (uint32_t)ny < nh) memcpy(dst + ((size_t)ny * nw + nx) * 4, src + ((size_t)y * w + x) * 4, 4);
        }
}

void pfxo_resize(const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, int filter, uint8_t* dst, int threads)
{
    if (nw == 0 || nh == 0) return;
    if (w == 0 || h == 0) { memset(dst, 0, (size_t)nw * nh * 4); return; }
    if (nw == w && nh == h) { memcpy(dst, src, (size_t)w * h * 4); return; }
    uint32_t *vl = (uint32_t*)malloc(sizeof(uint32_t) * nh), *vc = (uint32_t*)malloc(sizeof(uint32_t) * nh);
    size_t* vo = (size_t*)malloc(sizeof(size_t) * nh);
    size_t nv = pfxo_resize_weights(h, nh, filter, vl, vc, vo, NULL);
    float* vw = (float*)malloc(sizeof(float) * nv);
    pfxo_resize_weights(h, nh, filter, vl, vc, vo, vw);
    uint32_t *hl = (uint32_t*)malloc(sizeof(uint32_t) * nw), *hc = (uint32_t*)malloc(sizeof(uint32_t) * nw);
    size_t* ho = (size_t*)malloc(sizeof(size_t) * nw);
    size_t nhw = pfxo_resize_weights(w, nw, filter, hl, hc, ho, NULL);
    float* hw = (float*)malloc(sizeof(float) * nhw);
    pfxo_resize_weights(w, nw, filter, hl, hc, ho, hw);
    float* tmp = (float*)malloc(sizeof(float) * 4 * (size_t)w * nh);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long oy = 0; oy < (long)nh; ++oy)
        for (uint32_t x = 0; x < w; ++x) {
            float t[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < vc[oy]; ++i) {
                const uint8_t* p = src + ((size_t)(vl[oy] + i) * w + x) * 4;
                float wt = vw[vo[oy] + i];
                for (int c = 0; c < 4; ++c) t[c] += (float)p[c] * wt;
            }
            memcpy(tmp + ((size_t)oy * w + x) * 4, t, sizeof t);
        }
#pragma omp parallel for schedule(static)
    for (long oy = 0; oy < (long)nh; ++oy)
        for (uint32_t ox = 0; ox < nw; ++ox) {
            float t[4] = {0, 0, 0, 0};
            for (uint32_t i = 0; i < hc[ox]; ++i) {
                const float* p = tmp + ((size_t)oy * w + hl[ox] + i) * 4;
                float wt = hw[ho[ox] + i];
                for (int c = 0; c < 4; ++c) t[c] += p[c] * wt;
            }
            for (int c = 0; c < 4; ++c) /* FloatNearest(clamp(t, 0, 255)): round half away from zero */
                dst[((size_t)oy * nw + ox) * 4 + c] = rs_f32_as_u8(roundf(rs_clampf(t[c], 0.0f, 255.0f)));
        }
    free(vl); free(vc); free(vo); free(vw); free(hl); free(hc); free(ho); free(hw); free(tmp);
}
