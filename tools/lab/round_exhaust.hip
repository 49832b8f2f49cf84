// tools/lab/round_exhaust.hip — Rust's `v.round().clamp(0.0, 255.0) as u8` against  v_cvt_pk_u8_f32(bits(med3(v, -1, 300)) | 1)  for ALL 2^32
// f32 bit patterns.  v_cvt_pk_u8_f32 rounds to nearest-even and saturates to [0, 255]; setting the lowest significand bit turns every exact
// tie k + 0.5 (whose low bit is 0) into the next float above it and moves no other value across a tie, so nearest-even then equals
// round-half-away-from-zero.  Negative values and NaN must give 0, +inf 255: v_med3_f32 first (with a NaN operand it returns the minimum
// of the others, -1; +inf becomes 300 instead of turning into a NaN under the OR).  Without the med3 the only mismatch is +inf.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/round_exhaust tools/lab/round_exhaust.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ uint32_t reference_u8(float v) // round half away from zero in exact steps, clamp, NaN -> 0
{
    if (v != v) return 0u;
    float t = __builtin_truncf(v);
    const float d = v - t; // exact
    if (__builtin_fabsf(d) >= 0.5f) t += __builtin_copysignf(1.0f, v);
    t = __builtin_fminf(__builtin_fmaxf(t, 0.0f), 255.0f);
    return (uint32_t)t;
}
__global__ void k(unsigned long long* mism, uint32_t* first, unsigned long long* mism_snan)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, bad_snan = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += nth) {
        const uint32_t bits = (uint32_t)b;
        const float v = __builtin_bit_cast(float, bits);
        const float m = __builtin_amdgcn_fmed3f(v, -1.0f, 300.0f);
        const uint32_t fast = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_bit_cast(float, __builtin_bit_cast(uint32_t, m) | 1u), 0, 0u);
        if (fast != reference_u8(v)) {
            // signalling NaNs (exponent all ones, quiet bit clear, payload != 0): no arithmetic instruction produces one
            const bool snan = (bits & 0x7f800000u) == 0x7f800000u && (bits & 0x007fffffu) != 0u && !(bits & 0x00400000u);
            if (snan) ++bad_snan; else { ++bad; atomicMin(first, bits); }
        }
    }
    if (bad) atomicAdd(mism, bad);
    if (bad_snan) atomicAdd(mism_snan, bad_snan);
}
int main()
{
    unsigned long long *d, *ds; uint32_t* f;
    hipMalloc(&d, 8); hipMalloc(&ds, 8); hipMalloc(&f, 4);
    hipMemset(d, 0, 8); hipMemset(ds, 0, 8); hipMemset(f, 0xff, 4);
    k<<<4096, 256>>>(d, f, ds);
    unsigned long long h = 0, hs = 0; uint32_t hf = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, ds, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
    printf("{\"patterns\": 4294967296, \"mismatches\": %llu, \"first_mismatch_bits\": \"0x%08x\", \"signalling_nan_patterns_differing\": %llu}\n", h, hf, hs);
    return h != 0;
}
