// tools/lab/div_exhaust.hip — exhaustive check of shortened correctly-rounded f32 division sequences on gfx950.
//
// For every pair of f32 significands (mn, md) in [2^23, 2^24)^2 the candidate sequence is compared with the IEEE
// quotient `n / d` (hipcc -fhip-fp32-correctly-rounded-divide-sqrt).  Scaling n or d by a power of two scales every
// intermediate of the sequences exactly (no operand, product, residual or quotient leaves the normal range for the
// compositor's operands: numerators 0 or in [2^-100, 2^20], denominators in [2^-48, 2^20]), so the significand pairs
// cover every operand pair the kernels can see.
//   variant 0: y = rcp(d) refined once (2 FMA);  q0 = n*y; r0 = fma(-d,q0,n); q1 = fma(r0,y,q0)          [3 ops / quotient]
//   variant 1: the full sequence hipcc emits (one more residual + correction)                                [5 ops / quotient]
//   variant 2: y = rcp(d) unrefined;             q0, r0, q1                                               [3 ops, 1-op prepare]
// usage: div_exhaust <variant> [d_begin d_end]   (d range in significand units, default the whole 2^23 range)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

template <int V>
__global__ __launch_bounds__(256) void check(uint32_t d_lo, uint32_t n_d, unsigned long long* bad, uint32_t* first_bad)
{
    // one block per denominator; its 256 threads sweep the 2^23 numerator significands
    const uint32_t di = blockIdx.x;
    if (di >= n_d) return;
    const uint32_t md = d_lo + di; // significand in [2^23, 2^24)
    const float d = __builtin_bit_cast(float, (127u << 23) | (md & 0x7fffffu)); // [1, 2)
    float y = __builtin_amdgcn_rcpf(d);
    if (V != 2) {
        const float e = __builtin_fmaf(-d, y, 1.0f);
        y = __builtin_fmaf(e, y, y);
    }
    unsigned long long mine = 0;
    for (uint32_t mn = threadIdx.x; mn < (1u << 23); mn += 256) {
        const float n = __builtin_bit_cast(float, (127u << 23) | mn);
        const float q_ref = n / d;
        const float q0 = n * y;
        const float r0 = __builtin_fmaf(-d, q0, n);
        float q = __builtin_fmaf(r0, y, q0);
        if (V == 1) {
            const float r1 = __builtin_fmaf(-d, q, n);
            q = __builtin_fmaf(r1, y, q);
        }
        if (__builtin_bit_cast(uint32_t, q) != __builtin_bit_cast(uint32_t, q_ref)) {
            ++mine;
            if (atomicAdd(&first_bad[0], 1u) < 16u) { /* keep a few examples */
                const uint32_t slot = atomicAdd(&first_bad[1], 1u);
                if (slot < 16u) { first_bad[2 + 2 * slot] = md; first_bad[3 + 2 * slot] = mn; }
            }
        }
    }
    if (mine) atomicAdd(bad, mine);
}

int main(int argc, char** argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    uint32_t lo = argc > 3 ? (uint32_t)strtoul(argv[2], 0, 0) : 0u, hi = argc > 3 ? (uint32_t)strtoul(argv[3], 0, 0) : (1u << 23);
    unsigned long long* d_bad; uint32_t* d_first;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4 * 40);
    hipMemset(d_bad, 0, 8); hipMemset(d_first, 0, 4 * 40);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const uint32_t step = 1u << 16; // denominators per launch (keeps a launch around a second)
    for (uint32_t b = lo; b < hi; b += step) {
        const uint32_t n_d = (hi - b < step) ? hi - b : step;
        const uint32_t md0 = (1u << 23) + b;
        if (variant == 0) check<0><<<n_d, 256>>>(md0, n_d, d_bad, d_first);
        else if (variant == 1) check<1><<<n_d, 256>>>(md0, n_d, d_bad, d_first);
        else check<2><<<n_d, 256>>>(md0, n_d, d_bad, d_first);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long bad = 0; uint32_t first[40];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, d_first, 160, hipMemcpyDeviceToHost);
    printf("{\"variant\": %d, \"d_range\": [%u, %u], \"pairs\": %.6g, \"mismatches\": %llu, \"seconds\": %.1f", variant, lo, hi,
           (double)(hi - lo) * (double)(1u << 23), bad, ms * 1e-3);
    if (bad) {
        printf(", \"examples_md_mn\": [");
        for (uint32_t i = 0; i < first[1] && i < 16; ++i) printf("%s[%u, %u]", i ? ", " : "", first[2 + 2 * i], first[3 + 2 * i]);
        printf("]");
    }
    printf("}\n");
    return 0;
}
