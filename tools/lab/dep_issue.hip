// tools/lab/dep_issue.hip — does a DEPENDENT VALU instruction cost the SIMD more than an independent one when other waves are ready to issue?
// (tools/lab/select_chain.hip: the s_nop hipcc puts behind inline asm made a dependent chain 25 % faster.)  Loops of 48 v_fma_f32 per trip arranged as
// D independent chains dealt round-robin (dependency distance D = 1, 2, 3, 4, 6, 8), with and without an s_nop 0 behind every instruction, at 8 / 4 / 2
// waves per SIMD; and the same with v_bcnt_u32_b32 (a 3.3-cycle instruction).  Time per instruction relative to D = 8 of v_fma_f32 (= 2 cycles).
//   build: hipcc -O2 --offload-arch=gfx950 -o tools/lab/_bin/dep_issue tools/lab/dep_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <string>
#define F(r) "v_fma_f32 v" #r ", v" #r ", v20, v21\n"
#define B(r) "v_bcnt_u32_b32 v" #r ", v20, v" #r "\n"
#define N "s_nop 0\n"
// 48 instructions per trip
#define D1(I, P) I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P \
                 I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P I(10) P
#define G2(I, P) I(10) P I(11) P
#define G3(I, P) I(10) P I(11) P I(12) P
#define G4(I, P) I(10) P I(11) P I(12) P I(13) P
#define G6(I, P) I(10) P I(11) P I(12) P I(13) P I(14) P I(15) P
#define G8(I, P) I(10) P I(11) P I(12) P I(13) P I(14) P I(15) P I(16) P I(17) P
#define R2(X) X X
#define R3(X) X X X
#define R4(X) X X X X
#define R6(X) X X X X X X
#define R8(X) X X X X X X X X
#define D2(I, P) R8(R3(G2(I, P)))
#define D3(I, P) R8(R2(G3(I, P)))
#define D4(I, P) R4(R3(G4(I, P)))
#define D6(I, P) R8(G6(I, P))
#define D8(I, P) R6(G8(I, P))
#define CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v20", "v21", "s20", "scc"
#define KERNEL(NAME, BODY) __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) { uint32_t r; \
    asm volatile("v_mov_b32 v10, %1\nv_mov_b32 v11, %1\nv_mov_b32 v12, %1\nv_mov_b32 v13, %1\nv_mov_b32 v14, %1\nv_mov_b32 v15, %1\nv_mov_b32 v16, %1\nv_mov_b32 v17, %1\n" \
                 "v_mov_b32 v20, 1.0\nv_mov_b32 v21, 0\ns_movk_i32 s20, 1024\n1:\n" BODY "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n" \
                 "v_xor_b32 %0, v10, v11\nv_xor_b32 %0, %0, v12\nv_xor_b32 %0, %0, v13\nv_xor_b32 %0, %0, v14\nv_xor_b32 %0, %0, v15\nv_xor_b32 %0, %0, v16\nv_xor_b32 %0, %0, v17\n" \
                 : "=v"(r) : "v"(threadIdx.x * 2654435761u + seed) : CLOB); out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
KERNEL(f1, D1(F, "")) KERNEL(f2, D2(F, "")) KERNEL(f3, D3(F, "")) KERNEL(f4, D4(F, "")) KERNEL(f6, D6(F, "")) KERNEL(f8, D8(F, ""))
KERNEL(f1n, D1(F, N)) KERNEL(f2n, D2(F, N)) KERNEL(f3n, D3(F, N)) KERNEL(f4n, D4(F, N)) KERNEL(f8n, D8(F, N))
KERNEL(b1, D1(B, "")) KERNEL(b2, D2(B, "")) KERNEL(b4, D4(B, "")) KERNEL(b8, D8(B, "")) KERNEL(b1n, D1(B, N)) KERNEL(b2n, D2(B, N)) KERNEL(b8n, D8(B, N))
template <class K> double run(K k, uint32_t* out, int wg)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) k<<<256 * wg, 256>>>(out, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 8; ++i) k<<<256 * wg, 256>>>(out, 3u);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 8;
}
int main()
{
    uint32_t* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int i = 0; i < 200; ++i) f8<<<256 * 8, 256>>>(out, 3u);   // clocks up
    (void)hipDeviceSynchronize();
    for (int wg : {8, 4, 2, 1}) {
        const double base = run(f8, out, wg);
        printf("--- %d waves per SIMD; v_fma_f32, 8 chains: %.4f ms per launch = 2 cycles per instruction by definition\n", wg, base);
#define REP(NAME, K) { const double t = run(K, out, wg); printf("%-44s %.4f ms  %5.2f cycles per instruction\n", NAME, t, 2.0 * t / base); }
        REP("v_fma_f32  1 chain  (distance 1)", f1) REP("v_fma_f32  2 chains (distance 2)", f2) REP("v_fma_f32  3 chains", f3) REP("v_fma_f32  4 chains", f4) REP("v_fma_f32  6 chains", f6)
        REP("v_fma_f32  1 chain  + s_nop each", f1n) REP("v_fma_f32  2 chains + s_nop each", f2n) REP("v_fma_f32  3 chains + s_nop each", f3n) REP("v_fma_f32  4 chains + s_nop each", f4n) REP("v_fma_f32  8 chains + s_nop each", f8n)
        REP("v_bcnt     1 chain", b1) REP("v_bcnt     2 chains", b2) REP("v_bcnt     4 chains", b4) REP("v_bcnt     8 chains", b8)
        REP("v_bcnt     1 chain  + s_nop each", b1n) REP("v_bcnt     2 chains + s_nop each", b2n) REP("v_bcnt     8 chains + s_nop each", b8n)
    }
    return 0;
}
