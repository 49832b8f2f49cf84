// tools/lab/lds_banks.hip — LDS bank behaviour of the Gaussian strip kernel's access patterns on gfx950, one pattern per kernel instantiation so that
// rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS reports them separately (tools/lab/lds_banks.sh).
//   hipcc --offload-arch=gfx950 -O3 -o lds_banks lds_banks.hip && ./lds_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
constexpr int REPS = 2048;
// address (bytes) of a lane for pattern P; `k` walks 8 sub-positions the way the kernel's K blocks / groups do
template <int P> __device__ uint32_t addr_of(int lane, int k)
{
    const int i = lane & 31, hh = lane >> 5, frow = lane >> 3, fs = lane & 7, xl = i >> 2, c = i & 3;
    switch (P) {
    case 0:  return lane * 8;                                    // b64 contiguous
    case 1:  return i * 144 + 8 * hh + 16 * k;                   // producer A-fragment read (b64), patch pitch 144
    case 2:  return i * 136 + 8 * hh + 16 * (k & 3);             // ... pitch 136
    case 3:  return i * 144 + 8 * hh + 32 * (k & 3);             // ring write (b64), column pitch 144 (72 halves), rows 4 hh + 16 k'
    case 4:  return i * 136 + 8 * hh + 32 * (k & 3);             // ... pitch 136 (68 halves)
    case 5:  return lane * 16;                                   // b128 contiguous (B1 fragments)
    case 6:  return c * 4672 + xl * 144 + 16 * hh + 32 * (k & 1);// consumer fragment read (b128), pitch 144, plane 4672
    case 7:  return c * 4416 + xl * 136 + 16 * hh + 32 * (k & 1);// consumer fragment read as read2_b64, pitch 136, plane 4416
    case 8:  return frow * 144 + fs * 16 + 1152 * (k & 3);       // patch write (b128): [c][row][144]
    case 9:  return (36 * i + hh + 2 * (k & 3)) * 4;             // staged output write (b32), pitch 36 dwords
    case 10: return (lane >> 3) * 144 + (lane & 7) * 16;         // staged output read (b128): 8 lanes per 36-dword row
    case 13: return lane * 4;                                    // b32 contiguous
    case 14: return ((lane & 31) * 32 + ((8 * (k & 3) + hh + 2 * (k >> 2)) ^ (lane & 31))) * 4;   // staged output write, pitch 32, column ^ row
    case 15: return (frow * 136 + fs * 16 + 1088 * (k & 3)) ;   // patch write as b64 halves, pitch 136 (first half; the second is + 8)
    case 16: return lane * 16;                                   // write2_b64 contiguous pairs
    case 11: return i * 128 + ((((8 * hh + k) ^ ((((i >> 1) & 1) << 3) | ((i >> 2) & 7))) & 15) << 3);   // swizzled patch read (b64), pitch 128
    case 12: return i * 152 + 8 * hh + 16 * (k & 3);             // pitch 152
    default: return 0;
    }
}
template <int P, int W, bool WRITE> __global__ void pat(uint32_t* out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    for (int q = threadIdx.x; q < 8192; q += blockDim.x) reinterpret_cast<uint32_t*>(lds)[q] = q;
    __syncthreads();
    uint32_t acc = 0;
    uint32_t a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = addr_of<P>(lane, k);
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint8_t* p = lds + a[k];
            if constexpr (WRITE) {
                if constexpr (W == 4) *reinterpret_cast<volatile uint32_t*>(p) = acc + r;
                else if constexpr (W == 8) { u2 v = {acc + r, (uint32_t)k}; asm volatile("ds_write_b64 %0, %1" :: "v"((uint32_t)(uintptr_t)0 + a[k]), "v"(v) : "memory"); }
                else if constexpr (W == 16) { u4 v = {acc + r, (uint32_t)k, 1u, 2u}; asm volatile("ds_write_b128 %0, %1" :: "v"(a[k]), "v"(v) : "memory"); }
                else if constexpr (W == 88) { u2 v = {acc + r, (uint32_t)k}; asm volatile("ds_write2_b64 %0, %1, %1 offset1:1" :: "v"(a[k]), "v"(v) : "memory"); }
                else if constexpr (W == 44) { asm volatile("ds_write2_b32 %0, %1, %1 offset1:2" :: "v"(a[k]), "v"(acc + r) : "memory"); }
                else if constexpr (W == 40) { asm volatile("ds_write_b32 %0, %1" :: "v"(a[k]), "v"(acc + r) : "memory"); }
            } else {
                if constexpr (W == 8) { u2 v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[k]) : "memory"); acc += v.x; }
                else if constexpr (W == 16) { u4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[k]) : "memory"); acc += v.x; }
                else if constexpr (W == 88) { u4 v; asm volatile("ds_read2_b64 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[k]) : "memory"); acc += v.x; }
                else if constexpr (W == 89) { u4 v; asm volatile("ds_read2_b64 %0, %1 offset1:8\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a[k]) : "memory"); acc += v.x; }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (acc == 0x12345u) out[threadIdx.x] = acc;
}
template <int P, int W, bool WRITE> void run(const char* name, uint32_t* d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    pat<P, W, WRITE><<<1024, 256, 32768>>>(d);
    hipEventRecord(e0);
    pat<P, W, WRITE><<<1024, 256, 32768>>>(d);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups x 4 waves x REPS x 8 instructions over 256 CUs at ~2.1 GHz
    const double inst_per_cu = 1024.0 * 4 * REPS * 8 / 256;
    printf("{\"pattern\": \"%s\", \"P\": %d, \"ms\": %.4f, \"cycles_per_wave_instruction\": %.2f}\n", name, P, ms, ms * 1e-3 * 2.1e9 / inst_per_cu);
}
int main()
{
    uint32_t* d; hipMalloc(&d, 4096);
    run<0, 8, false>("read_b64 contiguous", d);
    run<1, 8, false>("read_b64 patch pitch 144", d);
    run<2, 8, false>("read_b64 patch pitch 136", d);
    run<12, 8, false>("read_b64 patch pitch 152", d);
    run<11, 8, false>("read_b64 patch pitch 128 xor-swizzled", d);
    run<0, 8, true>("write_b64 contiguous", d);
    run<3, 8, true>("write_b64 ring pitch 144", d);
    run<4, 8, true>("write_b64 ring pitch 136", d);
    run<5, 16, false>("read_b128 contiguous", d);
    run<6, 16, false>("read_b128 consumer pitch 144", d);
    run<7, 88, false>("read2_b64 consumer pitch 136", d);
    run<6, 88, false>("read2_b64 consumer pitch 144", d);
    run<5, 16, true>("write_b128 contiguous", d);
    run<8, 16, true>("write_b128 patch pitch 144", d);
    run<9, 4, true>("write_b32 staged output pitch 36", d);
    run<10, 16, false>("read_b128 staged output", d);
    run<13, 40, true>("write_b32 contiguous", d);
    run<9, 40, true>("write_b32 staged output pitch 36 (asm)", d);
    run<14, 40, true>("write_b32 staged output pitch 32 xor row", d);
    run<13, 44, true>("write2_b32 contiguous base, offset1:2", d);
    run<9, 44, true>("write2_b32 staged output pitch 36", d);
    run<15, 8, true>("write_b64 patch pitch 136", d);
    run<16, 88, true>("write2_b64 contiguous pairs", d);
    run<16, 89, false>("read2_b64 lane*16, offset1:8 (64 B apart)", d);
    run<16, 88, false>("read2_b64 contiguous pairs", d);
    return 0;
}
