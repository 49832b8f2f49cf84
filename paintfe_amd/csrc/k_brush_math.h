// k_brush_math.h — brush falloff shared by the device stamp kernel and the host LUT builder.
// Reference: compute_brush_alpha, src/ui/panels/tools/behavior/raster/brush_render.rs:54-82.
#pragma once
#include <hip/hip_runtime.h>

__host__ __device__ inline float pfx_clampf(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

__host__ __device__ inline float pfx_brush_alpha(float dist, float radius, float hardness, bool anti_aliased)
{
    if (radius <= 0.0f) return 0.0f;
    const float safe_hardness = pfx_clampf(hardness, 0.0f, 1.0f);
    const float t = pfx_clampf(dist / radius, 0.0f, 1.0f);
    const float falloff = t * t * (3.0f - 2.0f * t);
    const float material_alpha = 1.0f + (safe_hardness - 1.0f) * falloff;
    float coverage;
    if (anti_aliased) {
        const float edge0 = radius + 0.5f, edge1 = radius - 0.5f;
        if (dist <= edge1) coverage = 1.0f;
        else if (dist >= edge0) coverage = 0.0f;
        else {
            const float x = pfx_clampf((dist - edge0) / (edge1 - edge0), 0.0f, 1.0f);
            coverage = x * x * (3.0f - 2.0f * x);
        }
    } else coverage = (dist <= radius) ? 1.0f : 0.0f;
    return material_alpha * coverage;
}
