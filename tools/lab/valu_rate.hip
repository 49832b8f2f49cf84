// tools/lab/valu_rate.hip — issue cost (cycles per wave64 instruction) of the non-FMA VALU ops the Gaussian strip walk leans on,
// gfx950: one wave per SIMD, 8 independent chains, s_memtime around 64 x 8 instructions.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/valu_rate tools/lab/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define BODY(OPSTR)                                                                                                     \
    uint32_t x[8];                                                                                                      \
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;                                            \
    uint32_t y = seed | 0x01020304u, z = 0x64646464u;                                                                   \
    unsigned long long t0 = __builtin_readcyclecounter();                                                               \
    for (int it = 0; it < 64; ++it) {                                                                                   \
        asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                             \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])     \
                     : "v"(y), "v"(z));                                                                                 \
    }                                                                                                                   \
    unsigned long long t1 = __builtin_readcyclecounter();                                                               \
    uint32_t s = 0;                                                                                                     \
    for (int j = 0; j < 8; ++j) s ^= x[j];                                                                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                     \
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;

#define OP_ADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define OP_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define OP_PKRTZ(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define OP_CVTF16(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define OP_PKU8(n) "v_cvt_pk_u8_f32 %" #n ", %8, 1, %" #n "\n"
#define OP_CVTU32(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define OP_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %8\n"
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define OP_SDWA(n) "v_cvt_f32_f16_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"

#define KERNEL(NAME, OP) __global__ void NAME(uint32_t* out, unsigned long long* cyc, uint32_t seed) { BODY(OP) }
KERNEL(k_add, OP_ADD) KERNEL(k_perm, OP_PERM) KERNEL(k_pkrtz, OP_PKRTZ) KERNEL(k_cvtf16, OP_CVTF16) KERNEL(k_pku8, OP_PKU8)
KERNEL(k_cvtu32, OP_CVTU32) KERNEL(k_cndmask, OP_CNDMASK) KERNEL(k_lshlor, OP_LSHLOR) KERNEL(k_fma, OP_FMA) KERNEL(k_minu, OP_MINU)
KERNEL(k_sdwa, OP_SDWA)

template <class K>
void run(const char* name, K k, uint32_t* out, unsigned long long* cyc, int threads)
{
    k<<<256, threads>>>(out, cyc, 3u); hipDeviceSynchronize();
    k<<<256, threads>>>(out, cyc, 3u); hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %4d threads/WG: %6.2f cycles per wave-instruction\n", name, threads, (double)c / (64.0 * 8.0));
}
int main()
{
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
    for (int threads : {256, 512, 768, 1024}) {
        run("v_add_f32", k_add, out, cyc, threads); run("v_fma_f32", k_fma, out, cyc, threads);
        run("v_perm_b32", k_perm, out, cyc, threads); run("v_cvt_pkrtz_f16_f32", k_pkrtz, out, cyc, threads);
        run("v_cvt_f32_f16", k_cvtf16, out, cyc, threads); run("v_cvt_f32_f16_sdwa", k_sdwa, out, cyc, threads);
        run("v_cvt_pk_u8_f32", k_pku8, out, cyc, threads);
        run("v_cvt_u32_f32", k_cvtu32, out, cyc, threads); run("v_cndmask_b32", k_cndmask, out, cyc, threads);
        run("v_lshl_or_b32", k_lshlor, out, cyc, threads); run("v_min_u32", k_minu, out, cyc, threads);
    }
    return 0;
}
