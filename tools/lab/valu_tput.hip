// tools/lab/valu_tput.hip — THROUGHPUT (SIMD issue cycles per wave64 instruction) of the integer min / max family the median kernels lean on,
// gfx950: 8 waves per SIMD, 8 independent chains per wave, whole-kernel time from HIP events against v_fma_f32 (2 cycles).
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/valu_tput tools/lab/valu_tput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BODY(OPSTR)                                                                                                  \
    uint32_t x[8];                                                                                                   \
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;                                         \
    uint32_t y = seed | 0x01020304u, z = 0x64646464u;                                                                \
    for (int it = 0; it < 512; ++it) {                                                                               \
        asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                          \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])  \
                     : "v"(y), "v"(z));                                                                              \
    }                                                                                                                \
    uint32_t s = 0;                                                                                                  \
    for (int j = 0; j < 8; ++j) s ^= x[j];                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_MINU32(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define OP_PKMIN(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define OP_PKMAX(n) "v_pk_max_u16 %" #n ", %" #n ", %8\n"
#define OP_MINU16(n) "v_min_u16 %" #n ", %" #n ", %8\n"
#define OP_MIN3(n) "v_min3_u32 %" #n ", %" #n ", %8, %9\n"
#define OP_MED3(n) "v_med3_u32 %" #n ", %" #n ", %8, %9\n"
#define OP_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define OP_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define OP_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_MINI16(n) "v_pk_min_i16 %" #n ", %" #n ", %8\n"
#define OP_PKMINF16(n) "v_pk_min_f16 %" #n ", %" #n ", %8\n"
#define OP_PKMAXF16(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define OP_MINF32(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define OP_MAXF32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define OP_MED3F32(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_MIN3F32(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_PKMULF16(n) "v_pk_mul_f16 %" #n ", %" #n ", %8\n"
#define OP_MULF32(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define OP_ADDF32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define OP_SUBF32(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define OP_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define OP_TRUNC(n) "v_trunc_f32 %" #n ", %" #n "\n"
#define OP_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8\nv_mov_b32 %" #n ", %9\n"
#define OP_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define OP_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define OP_CVTUB(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define OP_CVTPK(n) "v_cvt_pk_u8_f32 %" #n ", %8, 1, %" #n "\n"
#define OP_ADDU32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define OP_LSHR(n) "v_lshrrev_b32 %" #n ", 8, %" #n "\n"
#define OP_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_BFI(n) "v_bfi_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define OP_MED3CLAMP(n) "v_add_f32 %" #n ", %" #n ", %8 clamp\n"
#define OP_SAD(n) "v_sad_u8 %" #n ", %" #n ", %8, %9\n"
#define OP_BCNT(n) "v_bcnt_u32_b32 %" #n ", %8, %" #n "\n"
#define OP_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x28\n"
#define OP_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 3, %9\n"
#define OP_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define OP_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define OP_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define OP_MULHI24(n) "v_mul_hi_u32_u24 %" #n ", %" #n ", %8\n"
#define OP_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define OP_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define OP_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define OP_CVTU32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define OP_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %9\n"
#define KERNEL(NAME, OP) __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) { BODY(OP) }
KERNEL(k_fma, OP_FMA) KERNEL(k_minu32, OP_MINU32) KERNEL(k_pkmin, OP_PKMIN) KERNEL(k_pkmax, OP_PKMAX) KERNEL(k_minu16, OP_MINU16)
KERNEL(k_min3, OP_MIN3) KERNEL(k_med3, OP_MED3) KERNEL(k_pkadd, OP_PKADD) KERNEL(k_and, OP_AND) KERNEL(k_perm, OP_PERM) KERNEL(k_mini16, OP_MINI16)
KERNEL(k_sad, OP_SAD) KERNEL(k_pkminf16, OP_PKMINF16) KERNEL(k_pkmaxf16, OP_PKMAXF16) KERNEL(k_minf32, OP_MINF32) KERNEL(k_maxf32, OP_MAXF32)
KERNEL(k_mulf32, OP_MULF32) KERNEL(k_addf32, OP_ADDF32) KERNEL(k_subf32, OP_SUBF32) KERNEL(k_fmac, OP_FMAC) KERNEL(k_trunc, OP_TRUNC) KERNEL(k_cnd, OP_CNDMASK)
KERNEL(k_cmp, OP_CMP) KERNEL(k_mov, OP_MOV) KERNEL(k_rcp, OP_RCP) KERNEL(k_cvtub, OP_CVTUB) KERNEL(k_cvtpk, OP_CVTPK) KERNEL(k_addu32, OP_ADDU32) KERNEL(k_lshr, OP_LSHR)
KERNEL(k_andor, OP_ANDOR) KERNEL(k_bfi, OP_BFI) KERNEL(k_xor, OP_XOR) KERNEL(k_addclamp, OP_MED3CLAMP)
KERNEL(k_med3f32, OP_MED3F32) KERNEL(k_min3f32, OP_MIN3F32) KERNEL(k_pkmulf16, OP_PKMULF16)
KERNEL(k_bcnt, OP_BCNT) KERNEL(k_bitop3, OP_BITOP3) KERNEL(k_alignbit, OP_ALIGNBIT) KERNEL(k_lshlor, OP_LSHLOR) KERNEL(k_add3, OP_ADD3)
KERNEL(k_mulhi, OP_MULHI) KERNEL(k_mullo, OP_MULLO) KERNEL(k_mulhi24, OP_MULHI24) KERNEL(k_mul24, OP_MUL24) KERNEL(k_mad24, OP_MAD24) KERNEL(k_bfe, OP_BFE) KERNEL(k_cvtu32, OP_CVTU32) KERNEL(k_lshladd, OP_LSHLADD)
__global__ __launch_bounds__(256) void k_bitop3_mixed(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;
    for (int it = 0; it < 512; ++it) {
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca\nv_bitop3_b32 %1, %2, %3, %4 bitop3:0xca\nv_bitop3_b32 %2, %3, %4, %5 bitop3:0xca\nv_bitop3_b32 %3, %4, %5, %6 bitop3:0xca\n"
                     "v_bitop3_b32 %4, %5, %6, %7 bitop3:0xca\nv_bitop3_b32 %5, %6, %7, %0 bitop3:0xca\nv_bitop3_b32 %6, %7, %0, %1 bitop3:0xca\nv_bitop3_b32 %7, %0, %1, %2 bitop3:0xca\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma_mixed(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;
    for (int it = 0; it < 512; ++it) {
        asm volatile("v_fma_f32 %0, %1, %2, %3\nv_fma_f32 %1, %2, %3, %4\nv_fma_f32 %2, %3, %4, %5\nv_fma_f32 %3, %4, %5, %6\n"
                     "v_fma_f32 %4, %5, %6, %7\nv_fma_f32 %5, %6, %7, %0\nv_fma_f32 %6, %7, %0, %1\nv_fma_f32 %7, %0, %1, %2\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_cnd_mixed(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;
    for (int it = 0; it < 512; ++it) {
        asm volatile("v_cmp_lt_u32 vcc, %0, %4\nv_cndmask_b32 %0, %1, %2, vcc\nv_cndmask_b32 %1, %2, %3, vcc\nv_cndmask_b32 %2, %3, %4, vcc\nv_cndmask_b32 %3, %4, %5, vcc\n"
                     "v_cndmask_b32 %4, %5, %6, vcc\nv_cndmask_b32 %5, %6, %7, vcc\nv_cndmask_b32 %6, %7, %0, vcc\nv_cndmask_b32 %7, %0, %1, vcc\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : : "vcc");
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// packed FP32 (two f32 per lane and instruction, register pairs): does v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue at the scalar-f32 rate (= twice the flops)?
#define PK_KERNEL(NAME, OPSTR) __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) { \
    typedef float f2 __attribute__((ext_vector_type(2))); f2 x[8]; \
    for (int j = 0; j < 8; ++j) { x[j].x = (float)(threadIdx.x + j + seed) * 1e-3f; x[j].y = x[j].x + 0.5f; } \
    f2 y = {1.0001f, 0.9999f}, z = {1e-6f, -1e-6f}; \
    for (int it = 0; it < 512; ++it) { \
        asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7) \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z)); } \
    float s = 0; for (int j = 0; j < 8; ++j) s += x[j].x + x[j].y; out[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(uint32_t, s); }
#define OP_PKFMA32(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_PKMUL32(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define OP_PKADD32(n) "v_pk_add_f32 %" #n ", %" #n ", %9\n"
PK_KERNEL(k_pkfma32, OP_PKFMA32) PK_KERNEL(k_pkmul32, OP_PKMUL32) PK_KERNEL(k_pkadd32, OP_PKADD32)
template <class K> double run(K k, uint32_t* out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * 8 * 4; // 8 workgroups of 4 waves per CU: 8 waves per SIMD, 4 rounds
    k<<<grid, 256>>>(out, 3u); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<<<grid, 256>>>(out, 3u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    uint32_t* out; hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    const double base = run(k_fma, out);
    const double insts = 256.0 * 8 * 4 * 4 * 512 * 8; // wave-instructions per launch
    printf("v_fma_f32      %.3f ms  (reference: 2 cycles -> %.2f GHz effective)\n", base, insts * 2 / 1024 / (base * 1e-3) / 1e9);
#define REPORT(NAME, K) { const double t = run(K, out); printf("%-14s %.3f ms  %.2f cycles per wave-instruction (v_fma_f32 = 2)\n", NAME, t, 2.0 * t / base); }
    REPORT("v_min_u32", k_minu32) REPORT("v_pk_min_u16", k_pkmin) REPORT("v_pk_max_u16", k_pkmax) REPORT("v_pk_min_i16", k_mini16) REPORT("v_min_u16", k_minu16)
    REPORT("v_min3_u32", k_min3) REPORT("v_med3_u32", k_med3) REPORT("v_pk_add_u16", k_pkadd) REPORT("v_and_b32", k_and) REPORT("v_perm_b32", k_perm)
    REPORT("v_sad_u8", k_sad) REPORT("v_pk_min_f16", k_pkminf16) REPORT("v_pk_max_f16", k_pkmaxf16) REPORT("v_min_f32", k_minf32) REPORT("v_max_f32", k_maxf32)
    REPORT("v_mul_f32", k_mulf32) REPORT("v_add_f32", k_addf32) REPORT("v_sub_f32", k_subf32) REPORT("v_fmac_f32", k_fmac) REPORT("v_trunc_f32", k_trunc)
    REPORT("v_cndmask_b32", k_cnd) REPORT("v_cmp+v_mov", k_cmp) REPORT("v_mov_b32", k_mov) REPORT("v_rcp_f32", k_rcp) REPORT("v_cvt_f32_ubyte0", k_cvtub)
    REPORT("v_cvt_pk_u8_f32", k_cvtpk) REPORT("v_add_u32", k_addu32) REPORT("v_lshrrev_b32", k_lshr) REPORT("v_and_or_b32", k_andor) REPORT("v_bfi_b32", k_bfi)
    REPORT("v_xor_b32", k_xor) REPORT("v_add_f32 clamp", k_addclamp)
    REPORT("v_med3_f32", k_med3f32) REPORT("v_min3_f32", k_min3f32) REPORT("v_pk_mul_f16", k_pkmulf16)
    REPORT("v_bcnt_u32_b32", k_bcnt) REPORT("v_bitop3_b32", k_bitop3) REPORT("v_alignbit_b32", k_alignbit) REPORT("v_lshl_or_b32", k_lshlor) REPORT("v_add3_u32", k_add3)
    REPORT("v_mul_hi_u32", k_mulhi) REPORT("v_mul_lo_u32", k_mullo) REPORT("v_mul_hi_u32_u24", k_mulhi24) REPORT("v_mul_u32_u24", k_mul24) REPORT("v_mad_u32_u24", k_mad24) REPORT("v_bfe_u32", k_bfe) REPORT("v_cvt_f32_u32", k_cvtu32) REPORT("v_lshl_add_u32", k_lshladd)
    REPORT("v_pk_fma_f32 (2 f32 per lane)", k_pkfma32) REPORT("v_pk_mul_f32", k_pkmul32) REPORT("v_pk_add_f32", k_pkadd32)
    REPORT("v_bitop3 mixed regs (8 per iteration)", k_bitop3_mixed) REPORT("v_fma mixed regs", k_fma_mixed) REPORT("v_cmp + 8 v_cndmask mixed regs (9 insts counted as 8)", k_cnd_mixed)
    return 0;
}
