// tools/lab/wide_load.hip — VERDICT r03 "next round" #2a: is the compositor's layer stream better served by raw dword loads (1 VGPR per pixel in
// flight, byte -> RN(k / 255) conversion done by the wave) than by the typed 8_8_8_8 UNORM load (conversion in the texture path, 4 VGPRs per pixel in
// flight, 1 KB of register write-back per 256 bytes read)?  Streams 32 layers of RANDOM bytes of an 8K image the way flatten_srt_kernel does (a wave
// walks units of 192 pixels, lane l of group j holds pixel 64 j + l) and runs F dependent fused multiply-adds per layer-pixel on the converted
// channels as a stand-in for the blend (F = 0: the load stream alone; the shipped kernel issues about 85 full-rate-equivalents per layer-pixel).
//   mode 0  typed buffer_load_format_xyzw, two register sets (the shipped shape)
//   mode 1  raw buffer_load_dword, DEPTH layers in flight, v_cvt_f32_ubyteN + the two-operation div255        (12 VALU per pixel)
//   mode 2  raw, DEPTH in flight, v_lshlrev_b32_sdwa (byte -> table offset) + ds_read_b32 from a 1 KB table   (4 VALU + 4 LDS reads; bank conflicts)
//   mode 3  raw, DEPTH in flight, the same through a 32 KB table with one copy per bank (conflict-free)        (8 VALU + 4 LDS reads)
//   mode 4  raw buffer_load_dword ... lds: DEPTH layers in flight in an LDS ring (no VGPR per pixel in flight), ds_read_b32 + v_cvt_f32_ubyteN + div255
// Every mode's sum of converted channels is checked against the typed mode's (the conversions are the same function of the byte).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ v4f ld4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ int ld1(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ void st1(int data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
#define DEV __device__ __forceinline__
DEV v4i rsrc(const void* base, uint32_t bytes, uint32_t w3)
{
    const unsigned long long a = (unsigned long long)base;
    v4i r; r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)w3; return r;
}
enum : uint32_t { UNORM8X4 = 0xFACu | (0u << 12) | (10u << 15), RAW32 = 0xFACu | (7u << 12) | (4u << 15) };
struct Layers { const uint8_t* p[34]; };
DEV float div255(float x) { return __builtin_fmaf(x, __builtin_bit_cast(float, 998277249u), x * __builtin_bit_cast(float, 2944335615u)); }
template <int B> DEV uint32_t byte_off(uint32_t raw, int) { return 0; }
#define SDWA_SHIFT(NAME, SH, SEL) DEV uint32_t NAME(uint32_t raw) { uint32_t r; \
    asm("v_lshlrev_b32_sdwa %0, " #SH ", %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:" #SEL : "=v"(r) : "v"(raw)); return r; }
SDWA_SHIFT(b0s2, 2, BYTE_0) SDWA_SHIFT(b1s2, 2, BYTE_1) SDWA_SHIFT(b2s2, 2, BYTE_2) SDWA_SHIFT(b3s2, 2, BYTE_3)
SDWA_SHIFT(b0s7, 7, BYTE_0) SDWA_SHIFT(b1s7, 7, BYTE_1) SDWA_SHIFT(b2s7, 7, BYTE_2) SDWA_SHIFT(b3s7, 7, BYTE_3)

template <int F> DEV void work(float (&acc)[4], const float (&t)[4])
{
#pragma unroll
    for (int i = 0; i < F / 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_fmaf(acc[c], t[c], t[c]);
    if (F == 0)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += t[c];
}

template <int MODE> DEV void convert(float (&t)[4], uint32_t raw, const float* tab, uint32_t lane_base)
{
    if constexpr (MODE == 1) {
        t[0] = div255((float)(raw & 0xffu)); t[1] = div255((float)((raw >> 8) & 0xffu)); t[2] = div255((float)((raw >> 16) & 0xffu)); t[3] = div255((float)(raw >> 24));
    } else if constexpr (MODE == 2) {
        const char* tb = reinterpret_cast<const char*>(tab);
        t[0] = *reinterpret_cast<const float*>(tb + b0s2(raw)); t[1] = *reinterpret_cast<const float*>(tb + b1s2(raw));
        t[2] = *reinterpret_cast<const float*>(tb + b2s2(raw)); t[3] = *reinterpret_cast<const float*>(tb + b3s2(raw));
    } else {
        const char* tb = reinterpret_cast<const char*>(tab);
        t[0] = *reinterpret_cast<const float*>(tb + (b0s7(raw) + lane_base)); t[1] = *reinterpret_cast<const float*>(tb + (b1s7(raw) + lane_base));
        t[2] = *reinterpret_cast<const float*>(tb + (b2s7(raw) + lane_base)); t[3] = *reinterpret_cast<const float*>(tb + (b3s7(raw) + lane_base));
    }
}

// WPB waves per workgroup share the table; every wave walks its own units
template <int MODE, int F, int DEPTH, int WPB, bool IL = false>
__global__ __launch_bounds__(64 * WPB) void k(Layers L, int n_layers, uint32_t n_px, uint32_t units_per_wave, uint8_t* dst)
{
    __shared__ float tab[MODE == 3 ? 256 * 32 : 256];
    __shared__ uint32_t ring[MODE == 4 ? WPB : 1][MODE == 4 ? DEPTH : 1][3][64];
    uint32_t rawv[3];
    if constexpr (MODE == 3) { for (uint32_t i = threadIdx.x; i < 256u * 32u; i += 64u * WPB) tab[i] = (float)(i >> 5) / 255.0f; }
    else { for (uint32_t i = threadIdx.x; i < 256u; i += 64u * WPB) tab[i] = (float)i / 255.0f; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));
    const uint32_t lane_base = (lane & 31u) * 4u;
    const uint32_t bytes = n_px * 4u, n_units = n_px / 192u;
    const v4i rd = rsrc(dst, bytes, RAW32);
    // IL: the WPB waves of a workgroup walk ADJACENT units side by side (a layer is read in pieces of WPB x 768 bytes instead of 768)
    const uint32_t wib = threadIdx.x >> 6;
    for (uint32_t it = 0; it < units_per_wave; ++it) {
        const uint32_t u = IL ? (blockIdx.x * units_per_wave + it) * WPB + wib : wave * units_per_wave + it;
        if (u >= n_units) break;
        int voff[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) voff[j] = (int)((u * 192u + 64u * j + lane) * 4u);
        float acc[3][4] = {};
        if constexpr (MODE == 0) {
            float tA[3][4], tB[3][4];
            auto fetch = [&](float (&t)[3][4], int l) {
                const v4i rs = rsrc(L.p[l], bytes, UNORM8X4);
#pragma unroll
                for (int j = 0; j < 3; ++j) { const v4f v = ld4(rs, voff[j], 0, 0); t[j][0] = v.x; t[j][1] = v.y; t[j][2] = v.z; t[j][3] = v.w; }
            };
            fetch(tA, 0);
            for (int l = 0; l < n_layers; l += 2) {        // L.p has two spare entries (copies of the last layer)
                fetch(tB, l + 1);
#pragma unroll
                for (int j = 0; j < 3; ++j) work<F>(acc[j], tA[j]);
                fetch(tA, l + 2);
#pragma unroll
                for (int j = 0; j < 3; ++j) work<F>(acc[j], tB[j]);
            }
        } else if constexpr (MODE == 4) {
            // the wave's ring: DEPTH layers x 3 groups x 64 dwords, filled by the loads themselves (M0 = slot base, lane l's dword lands at word l)
            uint32_t (*slot)[3][64] = ring[wib];
            auto fetch = [&](int sl, int l) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(L.p[l < n_layers ? l : n_layers]), 0, (int)bytes, (int)RAW32);
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)&slot[sl][j][0], 4, voff[j], 0, 0, 0);
            };
#pragma unroll
            for (int sl = 0; sl < DEPTH; ++sl) fetch(sl, sl);
            for (int l = 0; l < n_layers; l += DEPTH) {
#pragma unroll
                for (int sl = 0; sl < DEPTH; ++sl) {
                    // the oldest layer's three loads have landed when at most 3 (DEPTH - 1) are outstanding
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (DEPTH - 1)) : "memory");
                    float t[3][4];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        uint32_t raw;
                        asm volatile("ds_read_b32 %0, %1" : "=v"(raw) : "v"((uint32_t)(uintptr_t)&slot[sl][j][lane]) : "memory");
                        rawv[j] = raw;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rawv[0]), "+v"(rawv[1]), "+v"(rawv[2]) :: "memory");
#pragma unroll
                    for (int j = 0; j < 3; ++j) convert<1>(t[j], rawv[j], tab, lane_base);
                    fetch(sl, l + sl + DEPTH);
#pragma unroll
                    for (int j = 0; j < 3; ++j) work<F>(acc[j], t[j]);
                }
            }
        } else {
            uint32_t raw[DEPTH][3];
            auto fetch = [&](uint32_t (&r)[3], int l) {
                const v4i rs = rsrc(L.p[l < n_layers ? l : n_layers], bytes, RAW32);
#pragma unroll
                for (int j = 0; j < 3; ++j) r[j] = (uint32_t)ld1(rs, voff[j], 0, 0);
            };
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) fetch(raw[s], s);
            for (int l = 0; l < n_layers; l += DEPTH) {
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) {
                    float t[3][4];
#pragma unroll
                    for (int j = 0; j < 3; ++j) convert<MODE>(t[j], raw[s][j], tab, lane_base);
                    fetch(raw[s], l + s + DEPTH);
#pragma unroll
                    for (int j = 0; j < 3; ++j) work<F>(acc[j], t[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) st1(__builtin_bit_cast(int, acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3]), rd, voff[j], 0, 0);
    }
}

__global__ void fill(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 0x9E3779B9u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = x;
    }
}

template <int MODE, int F, int DEPTH, int WPB, bool IL = false>
static void run(const Layers& L, uint32_t n_px, uint8_t* dst, std::vector<uint32_t>* ref, uint32_t upw)
{
    const int n_layers = 32;
    const uint32_t n_units = n_px / 192u, waves = (n_units + upw - 1) / upw, blocks = (waves + WPB - 1) / WPB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 14; ++it) {
        hipEventRecord(e0);
        k<MODE, F, DEPTH, WPB, IL><<<blocks, 64 * WPB>>>(L, n_layers, n_px, upw, dst);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (it >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<uint32_t> out(1 << 16);
    hipMemcpy(out.data(), dst, out.size() * 4, hipMemcpyDeviceToHost);
    int same = -1;
    if (F == 0) { if (MODE == 0) *ref = out; else same = (out == *ref); }
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k<MODE, F, DEPTH, WPB, IL>);
    printf("{\"interleaved\": %d, \"mode\": %d, \"fma_per_px\": %d, \"depth\": %d, \"waves_per_wg\": %d, \"units_per_wave\": %u, \"vgprs\": %d, \"ms_min\": %.4f, \"ms_med\": %.4f, \"TBs\": %.3f, \"same_as_typed\": %d}\n",
           (int)IL, MODE, F, MODE == 0 ? 2 : DEPTH, WPB, upw, fa.numRegs, ms.front(), ms[ms.size() / 2], (double)n_px * 4 * 33 / ms.front() * 1e-9, same);
    fflush(stdout);
}

int main(int argc, char** argv)
{
    const uint32_t w = 7680, h = 4320, n_px = w * h;
    const uint32_t upw = argc > 1 ? (uint32_t)atoi(argv[1]) : 8u;
    Layers L;
    // argv[3] = skew in bytes: layer l starts l * skew bytes further than a 2 MiB-aligned slot (do 33 streams that advance in lock step at the same offset of
    // identically aligned buffers meet in the same HBM channels / banks?)
    const size_t skew = argc > 3 ? (size_t)atol(argv[3]) : 0;
    const size_t slot = (((size_t)n_px * 4 + (2u << 20) - 1) / (2u << 20)) * (2u << 20) + (skew ? (4u << 20) : 0);
    uint8_t* big; hipMalloc((void**)&big, slot * 32 + (8u << 20));
    for (int l = 0; l < 32; ++l) { L.p[l] = big + slot * l + skew * l; fill<<<4096, 256>>>((uint32_t*)L.p[l], n_px, 0x1234567u * (l + 1)); }
    L.p[32] = L.p[33] = L.p[31];
    uint8_t* dst; hipMalloc(&dst, (size_t)n_px * 4);
    hipDeviceSynchronize();
    std::vector<uint32_t> ref;
#define ROW(F) run<0, F, 2, 1>(L, n_px, dst, &ref, upw); run<1, F, 4, 1>(L, n_px, dst, &ref, upw); run<2, F, 4, 1>(L, n_px, dst, &ref, upw); \
               run<2, F, 8, 1>(L, n_px, dst, &ref, upw); run<3, F, 4, 8>(L, n_px, dst, &ref, upw); run<3, F, 8, 8>(L, n_px, dst, &ref, upw);
    if (argc > 3) {   // skew study: the load stream alone and F = 40, typed and raw
        printf("{\"skew_bytes\": %zu}\n", skew);
        run<0, 0, 2, 1>(L, n_px, dst, &ref, upw); run<2, 0, 4, 1>(L, n_px, dst, &ref, upw); run<0, 40, 2, 1>(L, n_px, dst, &ref, upw); run<2, 40, 4, 1>(L, n_px, dst, &ref, upw);
        return 0;
    }
    if (argc > 2) {   // chunk-size study: load stream alone and F = 40, waves of a workgroup side by side
#define IROW(F) run<0, F, 2, 4, true>(L, n_px, dst, &ref, upw); run<0, F, 2, 8, true>(L, n_px, dst, &ref, upw); run<0, F, 2, 4, false>(L, n_px, dst, &ref, upw); \
                run<2, F, 4, 4, true>(L, n_px, dst, &ref, upw); run<2, F, 4, 8, true>(L, n_px, dst, &ref, upw); run<2, F, 8, 4, true>(L, n_px, dst, &ref, upw); run<2, F, 4, 4, false>(L, n_px, dst, &ref, upw);
        run<0, 0, 2, 1>(L, n_px, dst, &ref, upw);
        IROW(0) IROW(40) IROW(64)
        return 0;
    }
    if (argc > 1 && atoi(argv[1]) < 0) {   // LDS-ring study
#define LROW(F) run<0, F, 2, 1>(L, n_px, dst, &ref, 8); run<4, F, 4, 4>(L, n_px, dst, &ref, 8); run<4, F, 6, 4>(L, n_px, dst, &ref, 8); run<4, F, 8, 2>(L, n_px, dst, &ref, 8); run<4, F, 3, 4>(L, n_px, dst, &ref, 8);
        LROW(0) LROW(40) LROW(64)
        return 0;
    }
    ROW(0) ROW(40) ROW(80) ROW(120)
    return 0;
}
