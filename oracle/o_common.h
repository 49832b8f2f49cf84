/* o_common.h — shared helpers for the oracle (TEST INFRASTRUCTURE ONLY, see pfx_oracle.h). */
#ifndef PFX_O_COMMON_H
#define PFX_O_COMMON_H
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "pfx_oracle.h"

/* Rust `f32 as u8`: truncate toward zero, saturate, NaN -> 0. */
static inline uint8_t rs_f32_as_u8(float v)
{
    if (!(v > 0.0f)) return 0; /* also NaN, -0.0, negatives */
    if (v >= 255.0f) return 255;
    return (uint8_t)(int)v;
}
/* Rust `f32 as u32` / `as usize` / `as i32` (saturating truncation) */
static inline uint32_t rs_f32_as_u32(float v)
{
    if (!(v > 0.0f)) return 0;
    if (v >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)v;
}
static inline int32_t rs_f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)v;
}
/* Rust f32::clamp (core::f32): NaN propagates */
static inline float rs_clampf(float x, float lo, float hi)
{
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
static inline void o_set_threads(int threads)
{
#ifdef _OPENMP
    if (threads <= 0) { /* all usable cores: the affinity mask, capped by the cgroup CPU quota (a quota-limited box
                           with a wide affinity mask would otherwise be oversubscribed and throttled) */
        threads = omp_get_num_procs();
        FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (f) {
            long long quota = 0, period = 0;
            if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
                int q = (int)((quota + period - 1) / period);
                if (q >= 1 && q < threads) threads = q;
            }
            fclose(f);
        }
    }
    omp_set_num_threads(threads);
#else
    (void)threads;
#endif
}
static inline int o_chunk_index(uint32_t x, uint32_t y, uint32_t w)
{
    uint32_t cxn = (w + PFXO_CHUNK - 1) / PFXO_CHUNK;
    return (int)((y / PFXO_CHUNK) * cxn + (x / PFXO_CHUNK));
}
#endif
