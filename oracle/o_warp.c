/*
 * o_warp.c — oracle restatement of the displacement / Catmull-Rom mesh warp.
 * TEST INFRASTRUCTURE ONLY (see pfx_oracle.h).  Follows:
 *   src/ops/transform.rs:1015-1201  DisplacementField (get/add, apply_push/expand/contract/twirl)
 *   src/ops/transform.rs:1288-1345  warp_displacement_full
 *   src/ops/transform.rs:1558-1567  catmull_rom_weights
 *   src/ops/transform.rs:1589-1646  catmull_rom_surface
 *   src/ops/transform.rs:1670-1705  generate_displacement_from_mesh
 *   src/ops/transform.rs:1712-1740  generate_displacement_from_mesh_fast
 *   src/ops/transform.rs:1743-1761  warp_mesh_catmull_rom
 */
#include "o_common.h"

/* :1558-1567 */
void pfxo_catmull_rom_weights(float t, float w[4])
{
    float t2 = t * t;
    float t3 = t2 * t;
    w[0] = -0.5f * t3 + t2 - 0.5f * t;
    w[1] = 1.5f * t3 - 2.5f * t2 + 1.0f;
    w[2] = -1.5f * t3 + 2.0f * t2 + 0.5f * t;
    w[3] = 0.5f * t3 - 0.5f * t2;
}

/* :1589-1646; points row-major (rows+1) x (cols+1), interleaved xy */
void pfxo_catmull_rom_surface(const float* pts, uint32_t cols, uint32_t rows, float u_global, float v_global,
                              float out[2])
{
    uint32_t ppr = cols + 1, num_rows = rows + 1;
    float col_f = rs_clampf(u_global, 0.0f, (float)cols - 0.0001f);
    float row_f = rs_clampf(v_global, 0.0f, (float)rows - 0.0001f);
    uint32_t ci = rs_f32_as_u32(col_f); if (ci > cols - 1) ci = cols - 1;
    uint32_t ri = rs_f32_as_u32(row_f); if (ri > rows - 1) ri = rows - 1;
    float u_local = col_f - (float)ci;
    float v_local = row_f - (float)ri;
    float wv[4], wu[4];
    pfxo_catmull_rom_weights(v_local, wv);
    uint32_t rv[4] = { ri == 0 ? 0 : ri - 1, ri, (ri + 1 < num_rows - 1) ? ri + 1 : num_rows - 1,
                       (ri + 2 < num_rows - 1) ? ri + 2 : num_rows - 1 };
    pfxo_catmull_rom_weights(u_local, wu);
    uint32_t cu0 = ci == 0 ? 0 : ci - 1, cu1 = ci;
    uint32_t cu2 = (ci + 1 < ppr - 1) ? ci + 1 : ppr - 1, cu3 = (ci + 2 < ppr - 1) ? ci + 2 : ppr - 1;
    float rvx[4], rvy[4];
    for (int j = 0; j < 4; ++j) {
        const float* base = pts + (size_t)rv[j] * ppr * 2;
        const float *p0 = base + cu0 * 2, *p1 = base + cu1 * 2, *p2 = base + cu2 * 2, *p3 = base + cu3 * 2;
        rvx[j] = wu[0] * p0[0] + wu[1] * p1[0] + wu[2] * p2[0] + wu[3] * p3[0];
        rvy[j] = wu[0] * p0[1] + wu[1] * p1[1] + wu[2] * p2[1] + wu[3] * p3[1];
    }
    out[0] = wv[0] * rvx[0] + wv[1] * rvx[1] + wv[2] * rvx[2] + wv[3] * rvx[3];
    out[1] = wv[0] * rvy[0] + wv[1] * rvy[1] + wv[2] * rvy[2] + wv[3] * rvy[3];
}

/* :1670-1705 */
void pfxo_mesh_displacement(const float* orig, const float* def, uint32_t cols, uint32_t rows,
                            uint32_t w, uint32_t h, float* disp, int threads)
{
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            float u = ((float)x + 0.5f) / (float)w * (float)cols;
            float v = ((float)y + 0.5f) / (float)h * (float)rows;
            float o[2], d[2];
            pfxo_catmull_rom_surface(orig, cols, rows, u, v, o);
            pfxo_catmull_rom_surface(def, cols, rows, u, v, d);
            disp[((size_t)y * w + x) * 2 + 0] = d[0] - o[0];
            disp[((size_t)y * w + x) * 2 + 1] = d[1] - o[1];
        }
}

/* :1712-1740 (preview path: original grid assumed uniform) */
void pfxo_mesh_displacement_fast(const float* def, uint32_t cols, uint32_t rows, uint32_t w, uint32_t h,
                                 float* disp, int threads)
{
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            float u = ((float)x + 0.5f) / (float)w * (float)cols;
            float v = ((float)y + 0.5f) / (float)h * (float)rows;
            float d[2];
            pfxo_catmull_rom_surface(def, cols, rows, u, v, d);
            disp[((size_t)y * w + x) * 2 + 0] = d[0] - ((float)x + 0.5f);
            disp[((size_t)y * w + x) * 2 + 1] = d[1] - ((float)y + 0.5f);
        }
}

/* :1288-1345; field dims == output dims == (w,h); source dims (sw,sh) */
void pfxo_warp_displacement_ex(const uint8_t* src, uint32_t sw, uint32_t sh, const float* disp,
                               uint32_t w, uint32_t h, uint8_t* dst, int threads)
{
    int32_t src_w = (int32_t)sw, src_h = (int32_t)sh;
    memset(dst, 0, (size_t)w * h * 4);
    o_set_threads(threads);
#pragma omp parallel for schedule(static)
    for (long y = 0; y < (long)h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            float ddx = disp[((size_t)y * w + x) * 2], ddy = disp[((size_t)y * w + x) * 2 + 1];
            float sx = (float)x - ddx, sy = (float)y - ddy;
            int32_t x0 = rs_f32_as_i32(floorf(sx)), y0 = rs_f32_as_i32(floorf(sy));
            if (x0 < -1 || y0 < -1 || x0 >= src_w || y0 >= src_h) continue;
            float fx = sx - (float)x0, fy = sy - (float)y0;
            float tl[4] = {0, 0, 0, 0}, tr[4] = {0, 0, 0, 0}, bl[4] = {0, 0, 0, 0}, br[4] = {0, 0, 0, 0};
#define SAMPLE(dstv, xx, yy)                                                          \
    do {                                                                              \
        int32_t _x = (xx), _y = (yy);                                                 \
        if (!(_x < 0 || _y < 0 || _x >= src_w || _y >= src_h)) {                      \
            const uint8_t* _p = src + ((size_t)_y * sw + (size_t)_x) * 4;             \
            dstv[0] = _p[0]; dstv[1] = _p[1]; dstv[2] = _p[2]; dstv[3] = _p[3];       \
        }                                                                             \
    } while (0)
            SAMPLE(tl, x0, y0);
            SAMPLE(tr, x0 + 1, y0);
            SAMPLE(bl, x0, y0 + 1);
            SAMPLE(br, x0 + 1, y0 + 1);
#undef SAMPLE
            uint8_t* o = dst + ((size_t)y * w + x) * 4;
            for (int c = 0; c < 4; ++c) {
                float top = tl[c] + (tr[c] - tl[c]) * fx;
                float bot = bl[c] + (br[c] - bl[c]) * fx;
                o[c] = rs_f32_as_u8(rs_clampf(roundf(top + (bot - top) * fy), 0.0f, 255.0f));
            }
        }
}

void pfxo_warp_displacement(const uint8_t* src, uint32_t w, uint32_t h, const float* disp, uint8_t* dst, int threads)
{
    pfxo_warp_displacement_ex(src, w, h, disp, w, h, dst, threads);
}

/* :1743-1761 */
void pfxo_warp_mesh_catmull_rom(const uint8_t* src, const float* orig, const float* def,
                                uint32_t cols, uint32_t rows, uint32_t w, uint32_t h, uint8_t* dst, int threads)
{
    float* disp = (float*)malloc(sizeof(float) * (size_t)w * h * 2);
    pfxo_mesh_displacement(orig, def, cols, rows, w, h, disp, threads);
    pfxo_warp_displacement(src, w, h, disp, dst, threads);
    free(disp);
}

/* ---- DisplacementField brushes :1051-1200 (serial scatter-add, as in the reference) ---- */
static void brush_bounds(float cx, float cy, float r, uint32_t w, uint32_t h, int32_t b[4])
{
    int32_t x0 = rs_f32_as_i32(floorf(cx - r)), y0 = rs_f32_as_i32(floorf(cy - r));
    int32_t x1 = rs_f32_as_i32(ceilf(cx + r)), y1 = rs_f32_as_i32(ceilf(cy + r));
    b[0] = x0 > 0 ? x0 : 0; b[1] = y0 > 0 ? y0 : 0;
    b[2] = x1 < (int32_t)w ? x1 : (int32_t)w; b[3] = y1 < (int32_t)h ? y1 : (int32_t)h;
}
/* mode: 0 push (dx,dy used), 1 expand, 2 contract, 3 twirl cw, 4 twirl ccw */
void pfxo_displacement_brush(float* disp, uint32_t w, uint32_t h, int mode, float cx, float cy,
                             float delta_x, float delta_y, float radius, float strength)
{
    float r = fmaxf(radius, 1.0f);
    float sigma = r / 3.0f;
    float sigma_sq_2 = 2.0f * sigma * sigma;
    float dir = (mode == 3) ? 1.0f : -1.0f;
    int32_t b[4];
    brush_bounds(cx, cy, r, w, h, b);
    for (int32_t py = b[1]; py < b[3]; ++py)
        for (int32_t px = b[0]; px < b[2]; ++px) {
            float dx = (float)px - cx, dy = (float)py - cy;
            float dist_sq = dx * dx + dy * dy;
            if (dist_sq > r * r) continue;
            float* d = disp + ((size_t)py * w + (size_t)px) * 2;
            switch (mode) {
            case 0: {
                float weight = expf(-dist_sq / sigma_sq_2) * strength;
                d[0] += delta_x * weight; d[1] += delta_y * weight;
                break;
            }
            case 1: {
                float dist = fmaxf(sqrtf(dist_sq), 0.001f);
                float t = dist / r;
                float weight = (1.0f - t) * (1.0f - t) * strength * 3.0f;
                d[0] += dx / dist * weight; d[1] += dy / dist * weight;
                break;
            }
            case 2: {
                float dist = fmaxf(sqrtf(dist_sq), 0.001f);
                float weight = expf(-dist_sq / sigma_sq_2) * strength;
                d[0] += -dx / dist * weight * 2.0f; d[1] += -dy / dist * weight * 2.0f;
                break;
            }
            default: {
                float weight = expf(-dist_sq / sigma_sq_2) * strength * dir;
                d[0] += -dy * weight * 0.1f; d[1] += dx * weight * 0.1f;
            }
            }
        }
}
