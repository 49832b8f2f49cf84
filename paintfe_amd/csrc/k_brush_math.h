// k_brush_math.h — alpha of one brush stamp at a distance from its centre, shared by the device stamp kernel and the host LUT builder.
// Reference: compute_brush_alpha, src/ui/panels/tools/behavior/raster/brush_render.rs:54-82 (the operation order is that function's; it is the parity contract).
// Two factors: the body — a cubic ease from 1 at the centre to `hardness` at the rim — and the rim's coverage, a one-pixel-wide cubic ease when anti-aliased.
#pragma once
#include <hip/hip_runtime.h>

__host__ __device__ inline float pfx_clampf(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
__host__ __device__ inline float pfx_ease3(float u) { return u * u * (3.0f - 2.0f * u); }   // 3u^2 - 2u^3 on [0, 1]

__host__ __device__ inline float pfx_brush_alpha(float dist, float radius, float hardness, bool anti_aliased)
{
    if (radius <= 0.0f) return 0.0f;
    const float rim_level = pfx_clampf(hardness, 0.0f, 1.0f);
    const float body = 1.0f + (rim_level - 1.0f) * pfx_ease3(pfx_clampf(dist / radius, 0.0f, 1.0f));
    float cover;
    if (!anti_aliased) cover = (dist <= radius) ? 1.0f : 0.0f;
    else {
        const float outer = radius + 0.5f, inner = radius - 0.5f;
        if (dist <= inner) cover = 1.0f;
        else if (dist >= outer) cover = 0.0f;
        else cover = pfx_ease3(pfx_clampf((dist - outer) / (inner - outer), 0.0f, 1.0f));
    }
    return body * cover;
}
