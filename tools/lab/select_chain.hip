// tools/lab/select_chain.hip — what does one plane step of the bit-sliced median select (k_median_bits.hip: and, bcnt, and, bcnt, add, sub, ashr, min,
// bitop3, bitop3 — one serial chain) cost a SIMD, and what changes it?  Explicit registers (the allocation decides VGPR-bank conflicts), 8 waves per
// SIMD, time from HIP events against a v_fma_f32 loop of the same instruction count (2 cycles per wave-instruction).
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/_bin/select_chain tools/lab/select_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// registers: planes v20..v27 (reg 0) and v28..v35 (reg 1) hold "np"; cand v10 / v11; k v12; scratch v13..v19
#define STEP_REAL(P0, P1)                                   \
    "v_and_b32 v13, v10, " P0 "\n"                          \
    "v_bcnt_u32_b32 v14, v13, 0\n"                          \
    "v_and_b32 v15, v11, " P1 "\n"                          \
    "v_bcnt_u32_b32 v16, v15, 0\n"                          \
    "v_add_u32 v14, v14, v16\n"                             \
    "v_sub_u32 v16, v12, v14\n"                             \
    "v_ashrrev_i32 v17, 31, v16\n"                          \
    "v_min_u32 v12, v12, v16\n"                             \
    "v_bitop3_b32 v10, v10, " P0 ", v17 bitop3:0x90\n"      \
    "v_bitop3_b32 v11, v11, " P1 ", v17 bitop3:0x90\n"
#define STEP_NOP(P0, P1)                                    \
    "v_and_b32 v13, v10, " P0 "\n"                          \
    "v_bcnt_u32_b32 v14, v13, 0\n"                          \
    "s_nop 0\n"                                             \
    "v_and_b32 v15, v11, " P1 "\n"                          \
    "v_bcnt_u32_b32 v16, v15, 0\n"                          \
    "s_nop 0\n"                                             \
    "v_add_u32 v14, v14, v16\n"                             \
    "v_sub_u32 v16, v12, v14\n"                             \
    "v_ashrrev_i32 v17, 31, v16\n"                          \
    "v_min_u32 v12, v12, v16\n"                             \
    "v_bitop3_b32 v10, v10, " P0 ", v17 bitop3:0x90\n"      \
    "v_bitop3_b32 v11, v11, " P1 ", v17 bitop3:0x90\n"      \
    "s_nop 0\n"
// the ands first, the second count accumulating onto the first (no add), k chosen by a bitop3 select instead of v_min_u32
#define STEP_LEAN(P0, P1)                                   \
    "v_and_b32 v13, v10, " P0 "\n"                          \
    "v_and_b32 v15, v11, " P1 "\n"                          \
    "v_bcnt_u32_b32 v14, v13, 0\n"                          \
    "v_bcnt_u32_b32 v14, v15, v14\n"                        \
    "v_sub_u32 v16, v12, v14\n"                             \
    "v_ashrrev_i32 v17, 31, v16\n"                          \
    "v_bitop3_b32 v12, v12, v16, v17 bitop3:0xca\n"         \
    "v_bitop3_b32 v10, v10, " P0 ", v17 bitop3:0x90\n"      \
    "v_bitop3_b32 v11, v11, " P1 ", v17 bitop3:0x90\n"
// two independent selects interleaved (second: cand v40 / v41, k v42, scratch v43..v47, the same planes)
#define STEP_TWO(P0, P1)                                    \
    "v_and_b32 v13, v10, " P0 "\n"                          \
    "v_and_b32 v43, v40, " P0 "\n"                          \
    "v_and_b32 v15, v11, " P1 "\n"                          \
    "v_and_b32 v45, v41, " P1 "\n"                          \
    "v_bcnt_u32_b32 v14, v13, 0\n"                          \
    "v_bcnt_u32_b32 v44, v43, 0\n"                          \
    "v_bcnt_u32_b32 v14, v15, v14\n"                        \
    "v_bcnt_u32_b32 v44, v45, v44\n"                        \
    "v_sub_u32 v16, v12, v14\n"                             \
    "v_sub_u32 v46, v42, v44\n"                             \
    "v_ashrrev_i32 v17, 31, v16\n"                          \
    "v_ashrrev_i32 v47, 31, v46\n"                          \
    "v_bitop3_b32 v12, v12, v16, v17 bitop3:0xca\n"         \
    "v_bitop3_b32 v42, v42, v46, v47 bitop3:0xca\n"         \
    "v_bitop3_b32 v10, v10, " P0 ", v17 bitop3:0x90\n"      \
    "v_bitop3_b32 v40, v40, " P0 ", v47 bitop3:0x90\n"      \
    "v_bitop3_b32 v11, v11, " P1 ", v17 bitop3:0x90\n"      \
    "v_bitop3_b32 v41, v41, " P1 ", v47 bitop3:0x90\n"
#define STEP_FMA(P0, P1) "v_fma_f32 v10, v10, " P0 ", " P1 "\nv_fma_f32 v11, v11, " P0 ", " P1 "\nv_fma_f32 v12, v12, " P0 ", " P1 "\nv_fma_f32 v13, v13, " P0 ", " P1 "\nv_fma_f32 v14, v14, " P0 ", " P1 "\n" \
                         "v_fma_f32 v15, v15, " P0 ", " P1 "\nv_fma_f32 v16, v16, " P0 ", " P1 "\nv_fma_f32 v17, v17, " P0 ", " P1 "\nv_fma_f32 v18, v18, " P0 ", " P1 "\nv_fma_f32 v19, v19, " P0 ", " P1 "\n"
// popcount alone / and alone / bitop3 alone in the same dependency shape (10 per step, serial pairs)
#define STEP_BCNT(P0, P1) "v_bcnt_u32_b32 v10, " P0 ", v10\nv_bcnt_u32_b32 v11, " P1 ", v11\nv_bcnt_u32_b32 v12, " P0 ", v12\nv_bcnt_u32_b32 v13, " P1 ", v13\nv_bcnt_u32_b32 v14, " P0 ", v14\n" \
                          "v_bcnt_u32_b32 v15, " P1 ", v15\nv_bcnt_u32_b32 v16, " P0 ", v16\nv_bcnt_u32_b32 v17, " P1 ", v17\nv_bcnt_u32_b32 v18, " P0 ", v18\nv_bcnt_u32_b32 v19, " P1 ", v19\n"
// the real step's instruction classes without its dependencies (every instruction on registers of its own)
#define STEP_INDEP(P0, P1)                                  \
    "v_and_b32 v13, v10, " P0 "\n"                          \
    "v_bcnt_u32_b32 v14, v36, 0\n"                          \
    "v_and_b32 v15, v11, " P1 "\n"                          \
    "v_bcnt_u32_b32 v16, v37, 0\n"                          \
    "v_add_u32 v18, v38, v39\n"                             \
    "v_sub_u32 v19, v12, v38\n"                             \
    "v_ashrrev_i32 v17, 31, v39\n"                          \
    "v_min_u32 v48, v12, v37\n"                             \
    "v_bitop3_b32 v49, v10, " P0 ", v36 bitop3:0x90\n"      \
    "v_bitop3_b32 v50, v11, " P1 ", v36 bitop3:0x90\n"

#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", \
             "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50"
#define EIGHT(S) S("v20", "v28") S("v21", "v29") S("v22", "v30") S("v23", "v31") S("v24", "v32") S("v25", "v33") S("v26", "v34") S("v27", "v35")
// the same with a plane layout that keeps the three sources of every bitop3 / and in different VGPR banks (bank = register number mod 4):
// cand v10 (bank 2) / v11 (3); zm v17 (1); planes of reg 0 in bank 0 or 3 …
#define EIGHT_NC(S) S("v20", "v24") S("v28", "v32") S("v20", "v24") S("v28", "v32") S("v20", "v24") S("v28", "v32") S("v20", "v24") S("v28", "v32")
#define KERNEL(NAME, BODY)                                                                                           \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed)                                        \
    {                                                                                                                \
        uint32_t r;                                                                                                  \
        asm volatile("v_mov_b32 v10, %1\nv_mov_b32 v11, %1\nv_mov_b32 v12, 24\nv_mov_b32 v40, %1\nv_mov_b32 v41, %1\nv_mov_b32 v42, 24\n"            \
                     "v_mov_b32 v20, %1\nv_mov_b32 v21, %1\nv_mov_b32 v22, %1\nv_mov_b32 v23, %1\nv_mov_b32 v24, %1\nv_mov_b32 v25, %1\nv_mov_b32 v26, %1\nv_mov_b32 v27, %1\n" \
                     "v_mov_b32 v28, %1\nv_mov_b32 v29, %1\nv_mov_b32 v30, %1\nv_mov_b32 v31, %1\nv_mov_b32 v32, %1\nv_mov_b32 v33, %1\nv_mov_b32 v34, %1\nv_mov_b32 v35, %1\n" \
                     "v_mov_b32 v36, %1\nv_mov_b32 v37, %1\nv_mov_b32 v38, %1\nv_mov_b32 v39, %1\n"                  \
                     "s_movk_i32 s20, 256\n"                                                                         \
                     "1:\n" BODY                                                                                     \
                     "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"                               \
                     "v_xor_b32 %0, v10, v11\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v40\nv_xor_b32 %0, %0, v14\n" \
                     : "=v"(r) : "v"(threadIdx.x * 2654435761u + seed) : CLOB, "s20", "scc");                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                              \
    }
KERNEL(k_fma, EIGHT(STEP_FMA)) KERNEL(k_real, EIGHT(STEP_REAL)) KERNEL(k_nop, EIGHT(STEP_NOP)) KERNEL(k_lean, EIGHT(STEP_LEAN)) KERNEL(k_two, EIGHT(STEP_TWO))
KERNEL(k_bcnt, EIGHT(STEP_BCNT)) KERNEL(k_indep, EIGHT(STEP_INDEP)) KERNEL(k_real_nc, EIGHT_NC(STEP_REAL)) KERNEL(k_lean_nc, EIGHT_NC(STEP_LEAN)) KERNEL(k_two_nc, EIGHT_NC(STEP_TWO))

template <class K> double run(K k, uint32_t* out, int wg_per_cu)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = 256 * wg_per_cu * 4;
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
    for (int wg : {8, 4, 2, 1}) {
        const double base = run(k_fma, out, wg);   // 80 instructions per loop trip at 2 cycles
        printf("--- %d waves per SIMD; v_fma_f32 loop %.3f ms (80 per trip = 160 cycles)\n", wg, base);
#define REPORT(NAME, K, N) { const double t = run(K, out, wg); printf("%-58s %.3f ms  %6.1f cycles per plane step  (%d instructions: %.2f each)\n", NAME, t, 160.0 * t / base / 8, N, 160.0 * t / base / 8 / N); }
        REPORT("plane step as shipped (10 VALU)", k_real, 10)
        REPORT("  + the compiler's 3 s_nop", k_nop, 10)
        REPORT("  sources in different VGPR banks", k_real_nc, 10)
        REPORT("lean step: chained bcnt, k by bitop3 (9 VALU)", k_lean, 9)
        REPORT("  sources in different VGPR banks", k_lean_nc, 9)
        REPORT("two selects interleaved, lean (18 VALU = two steps)", k_two, 18)
        REPORT("  sources in different VGPR banks", k_two_nc, 18)
        REPORT("the shipped step's instructions without dependencies", k_indep, 10)
        REPORT("10 v_bcnt_u32_b32 (accumulating, independent)", k_bcnt, 10)
    }
    return 0;
}
