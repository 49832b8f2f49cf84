// tools/lab/valu_mix.hip — VERDICT r03 "next round" #3: what a VALU instruction costs INSIDE a multiply / fused-multiply-add stream at the compositor's occupancy
// (6 waves per SIMD: six 4-wave workgroups per CU, one round), instead of as an isolated chain (tools/lab/valu_tput.hip, profiles/r03_valu_rates.txt).  Two experiments
// contradicted the isolated prices in round 3 (v_bitop3 selects, the larger if-conversion budget: profiles/r03_tuning.md), so the question is the MARGINAL
// cost: a base block of 16 v_fma_f32 / v_mul_f32 on 8 chains, against the same block with 8 test instructions interleaved (one behind every second
// base instruction).  Cycles come from s_memtime inside the waves (shader clock: power-capped clock changes do not move them), averaged over the waves:
//     marginal cycles per test instruction = (cycles(base + test) - cycles(base)) * SIMD share / (8 * iterations)
// with SIMD share = 1 / 6 (six waves interleave on the SIMD: a wave's elapsed cycles cover all six).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define ITERS 4096
// base: 16 full-rate instructions on chains %0..%7 (operands %16, %17 are loop-invariant)
#define B2(a, b) "v_fma_f32 %" #a ", %" #a ", %16, %17\nv_mul_f32 %" #b ", %" #b ", %16\n"
#define T_NONE(n)
#define BLOCK(T) B2(0, 1) T(8) B2(2, 3) T(9) B2(4, 5) T(10) B2(6, 7) T(11) B2(0, 1) T(12) B2(2, 3) T(13) B2(4, 5) T(14) B2(6, 7) T(15)
#define T_FMA(n) "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define T_TRUNC(n) "v_trunc_f32 %" #n ", %" #n "\n"
#define T_MAX(n) "v_max_f32 %" #n ", %" #n ", %17\n"
#define T_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %17, vcc\n"
#define T_CND_SGPR(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %17, %20\n"
#define T_CMP_CND(n) "v_cmp_eq_f32 vcc, 0, %" #n "\nv_cndmask_b32 %" #n ", %" #n ", %17, vcc\n"
#define T_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %17\n"
#define T_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %16, %17 bitop3:0xca\n"
#define T_SDWA(n) "v_lshlrev_b32_sdwa %" #n ", 2, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define T_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 7, %17\n"
#define T_AND(n) "v_and_b32 %" #n ", 0x3fc, %" #n "\n"
#define T_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define T_CVTUB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define T_FMACLAMP(n) "v_fma_f32 %" #n ", %" #n ", %16, %17 clamp\n"
#define T_DSREAD(n) "ds_read_b32 %" #n ", %18\n"
#define T_DSREAD_RND(n) "ds_read_b32 %" #n ", %19\n"
#define T_MOV(n) "v_mov_b32 %" #n ", %17\n"

#define KERNEL(NAME, T, TAIL)                                                                                                              \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed)                                       \
    {                                                                                                                                       \
        __shared__ float tab[256];                                                                                                          \
        for (uint32_t i = threadIdx.x; i < 256u; i += 256u) tab[i] = (float)i;                                                               \
        __syncthreads();                                                                                                                    \
        float x[16];                                                                                                                        \
        for (int j = 0; j < 16; ++j) x[j] = (float)((threadIdx.x * 2654435761u + j + seed) & 1023u) * 0.001f;                               \
        const float c0 = 0.999f + seed * 1e-9f, c1 = 0.0007f;                                                                               \
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) float*)tab;                                            \
        const uint32_t a_lin = lds0 + (threadIdx.x & 63u) * 4u, a_rnd = lds0 + (((threadIdx.x * 2654435761u) >> 22) & 0x3fcu);                \
        const unsigned long long lane_mask = 0x5555555555555555ull + seed;                                                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                                         \
        for (int it = 0; it < ITERS; ++it) {                                                                                                \
            asm volatile(BLOCK(T) TAIL                                                                                                      \
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),  \
                           "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])                                      \
                         : "v"(c0), "v"(c1), "v"(a_lin), "v"(a_rnd), "s"(lane_mask)                                                         \
                         : "vcc", "memory");                                                                                  \
        }                                                                                                                                   \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                                         \
        float s = 0;                                                                                                                        \
        for (int j = 0; j < 16; ++j) s += x[j];                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(uint32_t, s);                                                               \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                                                                    \
    }
KERNEL(k_base, T_NONE, "")
KERNEL(k_fma, T_FMA, "")
KERNEL(k_trunc, T_TRUNC, "")
KERNEL(k_max, T_MAX, "")
KERNEL(k_cnd_vcc, T_CND_VCC, "")
KERNEL(k_cnd_sgpr, T_CND_SGPR, "")
KERNEL(k_cmp_cnd, T_CMP_CND, "")
KERNEL(k_cmp, T_CMP, "")
KERNEL(k_bitop3, T_BITOP3, "")
KERNEL(k_sdwa, T_SDWA, "")
KERNEL(k_lshladd, T_LSHLADD, "")
KERNEL(k_and, T_AND, "")
KERNEL(k_rcp, T_RCP, "")
KERNEL(k_cvtub, T_CVTUB, "")
KERNEL(k_fmaclamp, T_FMACLAMP, "")
KERNEL(k_mov, T_MOV, "")
KERNEL(k_dsread, T_DSREAD, "s_waitcnt lgkmcnt(0)\n")
KERNEL(k_dsread_rnd, T_DSREAD_RND, "s_waitcnt lgkmcnt(0)\n")
// round-mode switches around a group of adds (requant through an LDS table would add 2^23 under round-toward-zero): 8 adds + 2 s_setreg per block
#define T_ADD(n) "v_add_f32 %" #n ", %" #n ", %17\n"
__global__ __launch_bounds__(256) void k_setreg(uint32_t* out, unsigned long long* cyc, uint32_t seed)
{
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = (float)((threadIdx.x * 2654435761u + j + seed) & 1023u) * 0.001f;
    const float c0 = 0.999f + seed * 1e-9f, c1 = 0.0007f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(B2(0, 1) B2(2, 3) B2(4, 5) B2(6, 7) "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n" T_ADD(8) T_ADD(9) T_ADD(10) T_ADD(11) T_ADD(12) T_ADD(13) T_ADD(14) T_ADD(15)
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n" B2(0, 1) B2(2, 3) B2(4, 5) B2(6, 7)
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                       "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                     : "v"(c0), "v"(c1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(uint32_t, s);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ __launch_bounds__(256) void k_add8(uint32_t* out, unsigned long long* cyc, uint32_t seed)
{
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = (float)((threadIdx.x * 2654435761u + j + seed) & 1023u) * 0.001f;
    const float c0 = 0.999f + seed * 1e-9f, c1 = 0.0007f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        asm volatile(B2(0, 1) B2(2, 3) B2(4, 5) B2(6, 7) T_ADD(8) T_ADD(9) T_ADD(10) T_ADD(11) T_ADD(12) T_ADD(13) T_ADD(14) T_ADD(15) B2(0, 1) B2(2, 3) B2(4, 5) B2(6, 7)
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]),
                       "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                     : "v"(c0), "v"(c1));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(uint32_t, s);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <class K> static double run(K k, uint32_t* out, unsigned long long* cyc, int wgs, float* ms_out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<wgs, 256>>>(out, cyc, 3u); hipDeviceSynchronize();
    hipEventRecord(a);
    k<<<wgs, 256>>>(out, cyc, 3u);
    hipEventRecord(b); hipEventSynchronize(b);
    hipEventElapsedTime(ms_out, a, b);
    std::vector<unsigned long long> h(wgs * 4);
    hipMemcpy(h.data(), cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / (wgs * 4);
}

int main(int argc, char** argv)
{
    const int wps = argc > 1 ? atoi(argv[1]) : 6;             // waves per SIMD
    const int wgs = 256 * wps;                               // 4-wave workgroups, one wave per SIMD: wps workgroups per CU, one round
    uint32_t* out; hipMalloc(&out, (size_t)wgs * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, (size_t)wgs * 4 * 8);
    float ms;
    const double base = run(k_base, out, cyc, wgs, &ms);
    // s_memtime counts at a fixed 100 MHz on this part or at the shader clock?  report both views: cycles per base instruction should be 2 if it is the shader clock
    // s_memtime ticks are shader cycles (guides/MI355X_MICROARCH.md); one round of resident waves <=> kernel time ~ a wave's cycles / clock
    printf("{\"waves_per_simd\": %d, \"base_cycles_per_wave\": %.0f, \"base_ms\": %.4f, \"implied_clock_ghz_if_one_round\": %.2f, \"cycles_per_base_inst\": %.3f}\n", wps, base, ms, base / (ms * 1e6), base / wps / (16.0 * ITERS));
    const double unit = 1.0;
#define REPORT(NAME, K, NTEST) { const double c = run(K, out, cyc, wgs, &ms); \
    printf("{\"test\": \"%s\", \"marginal_cycles_per_inst\": %.2f, \"ms\": %.4f}\n", NAME, (c - base) / wps / ((NTEST) * (double)ITERS) / unit, ms); fflush(stdout); }
    REPORT("v_fma_f32 (control)", k_fma, 8) REPORT("v_mov_b32", k_mov, 8) REPORT("v_trunc_f32", k_trunc, 8) REPORT("v_max_f32", k_max, 8)
    REPORT("v_cndmask_b32 vcc (stale vcc)", k_cnd_vcc, 8) REPORT("v_cndmask_b32 sgpr pair", k_cnd_sgpr, 8) REPORT("v_cmp_eq_f32 + v_cndmask_b32 (pair)", k_cmp_cnd, 8)
    REPORT("v_cmp_lt_f32", k_cmp, 8) REPORT("v_bitop3_b32", k_bitop3, 8) REPORT("v_lshlrev_b32_sdwa BYTE_1", k_sdwa, 8) REPORT("v_lshl_add_u32", k_lshladd, 8)
    REPORT("v_and_b32 literal", k_and, 8) REPORT("v_rcp_f32", k_rcp, 8) REPORT("v_cvt_f32_ubyte1", k_cvtub, 8) REPORT("v_fma_f32 clamp", k_fmaclamp, 8)
    REPORT("ds_read_b32 lane order (+ one s_waitcnt per block)", k_dsread, 8) REPORT("ds_read_b32 random table offsets", k_dsread_rnd, 8)
    const double add8 = run(k_add8, out, cyc, wgs, &ms);
    const double setr = run(k_setreg, out, cyc, wgs, &ms);
    printf("{\"test\": \"2 x s_setreg_imm32_b32 MODE.round around 8 v_add_f32\", \"extra_cycles_per_pair_of_switches\": %.2f}\n", (setr - add8) / wps / (double)ITERS / unit);
    return 0;
}
