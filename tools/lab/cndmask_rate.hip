// tools/lab/cndmask_rate.hip — what a per-lane select costs on gfx950 (SIMD issue cycles per wave64 instruction, 8 waves per SIMD,
// 8 independent chains), in the forms the compiler emits and in arithmetic / bitwise replacements.
// build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/cndmask_rate tools/lab/cndmask_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BODY(PRE, OPSTR, ...)                                                                                       \
    uint32_t x[8];                                                                                                   \
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;                                         \
    uint32_t y = seed | 0x01020304u, z = (threadIdx.x & 1) ? 0xffffffffu : 0u;                                       \
    asm volatile(PRE ::"v"(y), "v"(z) : __VA_ARGS__);                                                                      \
    for (int it = 0; it < 512; ++it) {                                                                               \
        asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                          \
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])  \
                     : "v"(y), "v"(z) : __VA_ARGS__);                                                                       \
    }                                                                                                                \
    uint32_t s = 0;                                                                                                  \
    for (int j = 0; j < 8; ++j) s ^= x[j];                                                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
#define OP_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define OP_CND_VCC(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_CND_SGPR(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define OP_CND_OTHER(n) "v_cndmask_b32 %" #n ", %9, %8, vcc\n"
#define OP_CMP_CND(n) "v_cmp_lt_f32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_CMP_ONLY(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n"
#define OP_CMPX(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8\n"
#define OP_BFI(n) "v_bfi_b32 %" #n ", %9, %8, %" #n "\n"
#define OP_ARITH(n) "v_mul_f32 %" #n ", %" #n ", %9\nv_fmac_f32 %" #n ", %8, %9\n"
#define OP_CND_CONST(n) "v_cndmask_b32_e64 %" #n ", 0, 1.0, vcc\n"
#define OP_CND_DPP(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0xf\n"
#define K(NAME, PRE, OP, ...) __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) { BODY(PRE, OP, __VA_ARGS__) }
K(k_fma, "", OP_FMA, "memory")
K(k_cnd_vcc, "v_cmp_lt_u32 vcc, %0, %1\n", OP_CND_VCC, "vcc")
K(k_cnd_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n", OP_CND_SGPR, "s20", "s21")
K(k_cnd_other, "v_cmp_lt_u32 vcc, %0, %1\n", OP_CND_OTHER, "vcc")
K(k_cmp_cnd, "", OP_CMP_CND, "vcc")
K(k_cmp_only, "", OP_CMP_ONLY, "vcc")
K(k_cmpx, "", OP_CMPX, "s20", "s21")
K(k_bfi, "", OP_BFI, "memory")
K(k_arith, "", OP_ARITH, "memory")
K(k_cnd_const, "v_cmp_lt_u32 vcc, %0, %1\n", OP_CND_CONST, "vcc")
K(k_dpp, "", OP_CND_DPP, "memory")
// one compare feeding four selects (what `skip ? acc : o` on four channels compiles to), VOP2 and VOP3 encodings
#define OP_G4(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define OP_G4E64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define GROUP4(C0, A, B, C, D) "v_cmp_lt_f32 vcc, %" #C0 ", %8\n" A B C D
#define OP_GRP(n) ""
__global__ __launch_bounds__(256) void k_grp(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;
    uint32_t y = seed | 0x01020304u, z = 0;
    for (int it = 0; it < 512; ++it)
        asm volatile(GROUP4(0, OP_G4(0), OP_G4(1), OP_G4(2), OP_G4(3)) GROUP4(4, OP_G4(4), OP_G4(5), OP_G4(6), OP_G4(7))
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z) : "vcc");
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_grp64(uint32_t* out, uint32_t seed)
{
    uint32_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 2654435761u + j + seed;
    uint32_t y = seed | 0x01020304u, z = 0;
    for (int it = 0; it < 512; ++it)
        asm volatile(GROUP4(0, OP_G4E64(0), OP_G4E64(1), OP_G4E64(2), OP_G4E64(3)) GROUP4(4, OP_G4E64(4), OP_G4E64(5), OP_G4E64(6), OP_G4E64(7))
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(y), "v"(z) : "vcc");
    uint32_t s = 0;
    for (int j = 0; j < 8; ++j) s ^= x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class Kf> double run(Kf k, uint32_t* out)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = 256 * 8 * 4;
    k<<<grid, 256>>>(out, 3u); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<<<grid, 256>>>(out, 3u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main()
{
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 4 * 256 * 4);
    const double base = run(k_fma, out);
    printf("v_fma_f32 %.3f ms = 2 cycles\n", base);
#define REPORT(NAME, Kf, N) { const double t = run(Kf, out); printf("%-44s %.3f ms  %.2f cycles per instruction\n", NAME, t, 2.0 * t / base / N); }
    REPORT("v_cndmask_b32 x, x, y, vcc", k_cnd_vcc, 1) REPORT("v_cndmask_b32_e64 x, x, y, s[20:21]", k_cnd_sgpr, 1)
    REPORT("v_cndmask_b32 x, z, y, vcc (no self-dependence)", k_cnd_other, 1) REPORT("v_cndmask_b32_e64 x, 0, 1.0, vcc", k_cnd_const, 1)
    REPORT("v_cmp_lt_f32 vcc + v_cndmask (per pair)", k_cmp_cnd, 1) REPORT("v_cmp_lt_f32 vcc", k_cmp_only, 1) REPORT("v_cmp_lt_f32_e64 s[20:21]", k_cmpx, 1)
    REPORT("v_bfi_b32 (mask in a VGPR)", k_bfi, 1) REPORT("v_mul_f32 + v_fmac_f32 (arithmetic select, per pair)", k_arith, 1) REPORT("v_mov_b32_dpp", k_dpp, 1)
    REPORT("v_cmp vcc + 4 x v_cndmask_b32 (VOP2), per instruction", k_grp, 1.25) REPORT("v_cmp vcc + 4 x v_cndmask_b32_e64, per instruction", k_grp64, 1.25)
    return 0;
}
