/*
 * o_affine.c — oracle restatement of the layer affine / perspective resampler.  TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).
 * Follows:
 *   src/ops/transform.rs:826-946   apply_affine (R = Rz*Ry*Rx homography, inverse-mapped, bilinear against transparent / nearest)
 *   src/ops/transform.rs:949-976   invert_3x3
 *   src/ops/transform.rs:750-780   affine_transform_layer (canvas-sized output, Interpolation::Bilinear)
 * Pinned by transform/affine_rotate_90, transform/affine_scale_half (tests/transform_ops.rs:279-303) and
 * transforms/affine_rotate_45 (tests/visual_transforms.rs:234-249).
 */
#include "o_common.h"

#define PI_F 3.14159265358979323846f

void pfxo_affine_matrix(uint32_t canvas_w, uint32_t canvas_h, float rotation_z, float rotation_x, float rotation_y, float hi_out[9])
{
    float focal = (float)(canvas_w > canvas_h ? canvas_w : canvas_h) * 1.5f;
    float az = rotation_z * (PI_F / 180.0f), ax = rotation_x * (PI_F / 180.0f), ay = rotation_y * (PI_F / 180.0f);
    float sz = sinf(az), cz = cosf(az), sxr = sinf(ax), cxr = cosf(ax), syr = sinf(ay), cyr = cosf(ay);
    float r00 = cz * cyr, r01 = cz * syr * sxr - sz * cxr, r10 = sz * cyr, r11 = sz * syr * sxr + cz * cxr, r20 = -syr, r21 = cyr * sxr;
    float a = focal * r00, b = focal * r01, c = 0.0f, d = focal * r10, e = focal * r11, f = 0.0f, g = r20, h = r21, i = focal;
    float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (fabsf(det) < 1e-12f) {
        const float id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(hi_out, id, sizeof id);
        return;
    }
    float inv = 1.0f / det;
    hi_out[0] = (e * i - f * h) * inv; hi_out[1] = (c * h - b * i) * inv; hi_out[2] = (b * f - c * e) * inv;
    hi_out[3] = (f * g - d * i) * inv; hi_out[4] = (a * i - c * g) * inv; hi_out[5] = (c * d - a * f) * inv;
    hi_out[6] = (d * h - e * g) * inv; hi_out[7] = (b * g - a * h) * inv; hi_out[8] = (a * e - b * d) * inv;
}

/* interpolation: 0 nearest, otherwise bilinear */
void pfxo_affine(const uint8_t* src, uint32_t sw32, uint32_t sh32, uint32_t canvas_w, uint32_t canvas_h, float rotation_z, float rotation_x,
                 float rotation_y, float scale, float offset_x, float offset_y, int interpolation, uint8_t* dst, int threads)
{
    memset(dst, 0, (size_t)canvas_w * canvas_h * 4);
    float cx = (float)canvas_w * 0.5f, cy = (float)canvas_h * 0.5f;
    float inv_scale = fabsf(scale) > 1e-6f ? 1.0f / scale : 1.0f;
    float hi[9];
    pfxo_affine_matrix(canvas_w, canvas_h, rotation_z, rotation_x, rotation_y, hi);
    int32_t src_w = (int32_t)sw32, src_h = (int32_t)sh32;
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long dy = 0; dy < (long)canvas_h; ++dy) {
        float v = ((float)dy - cy - offset_y) * inv_scale;
        float base_sx = hi[1] * v + hi[2], base_sy = hi[4] * v + hi[5], base_sw = hi[7] * v + hi[8];
        for (uint32_t dx = 0; dx < canvas_w; ++dx) {
            float u = ((float)dx - cx - offset_x) * inv_scale;
            float w = hi[6] * u + base_sw;
            if (fabsf(w) < 1e-8f) continue;
            float inv_w = 1.0f / w;
            float src_x = (hi[0] * u + base_sx) * inv_w + cx;
            float src_y = (hi[3] * u + base_sy) * inv_w + cy;
            uint8_t* o = dst + ((size_t)dy * canvas_w + dx) * 4;
            if (interpolation == 0) {
                int32_t nx = rs_f32_as_i32(roundf(src_x)), ny = rs_f32_as_i32(roundf(src_y));
                if (nx >= 0 && ny >= 0 && nx < src_w && ny < src_h) memcpy(o, src + ((size_t)ny * src_w + nx) * 4, 4);
                continue;
            }
            int32_t x0 = rs_f32_as_i32(floorf(src_x)), y0 = rs_f32_as_i32(floorf(src_y));
            if (x0 < -1 || y0 < -1 || x0 >= src_w || y0 >= src_h) continue;
            float fx = src_x - (float)x0, fy = src_y - (float)y0;
            float s[4][4];
            for (int k = 0; k < 4; ++k) {
                int32_t sx = x0 + (k & 1), sy = y0 + (k >> 1);
                for (int c = 0; c < 4; ++c)
                    s[k][c] = (sx < 0 || sy < 0 || sx >= src_w || sy >= src_h) ? 0.0f : (float)src[((size_t)sy * src_w + sx) * 4 + c];
            }
            for (int c = 0; c < 4; ++c) {
                float top = s[0][c] + (s[1][c] - s[0][c]) * fx;
                float bot = s[2][c] + (s[3][c] - s[2][c]) * fx;
                o[c] = rs_f32_as_u8(rs_clampf(roundf(top + (bot - top) * fy), 0.0f, 255.0f));
            }
        }
    }
}
