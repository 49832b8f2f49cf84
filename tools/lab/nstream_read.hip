// tools/lab/nstream_read.hip — what read rate does the chip sustain when one launch walks N separately allocated buffers in lock step (the compositor's
// access pattern: every output pixel needs the same offset of N layers) as against one contiguous stream of the same size?  No arithmetic but an XOR
// per loaded register, raw loads only; the result of a unit is stored (one stream of writes, 1 / N of the traffic).
//   VEC   dwords per lane and load (1 = buffer_load_dword, the compositor's 4 bytes per lane; 4 = buffer_load_dwordx4)
//   DEPTH loads in flight per wave
//   shape 0: a wave walks units; per unit it reads piece u of layer 0, 1, … N−1 (N streams interleaved at 256·VEC-byte grain)
//   shape 2: as shape 0 with the wave's units a whole launch apart (unit = wave + i * waves): long streams, compact window of addresses in flight
//   shape 1: the same bytes as ONE stream (layer-major: the wave reads N consecutive pieces of one buffer) — the contiguous ceiling
// Prints one JSON line per configuration.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/lab/_bin/nstream_read tools/lab/nstream_read.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ int ld1(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.i32");
__device__ v4i ld4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ void st1(int data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ void st4(v4i data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.v4i32");
#define DEV __device__ __forceinline__
DEV v4i rsrc(const void* base, uint32_t bytes)
{
    const unsigned long long a = (unsigned long long)base;
    v4i r; r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)(0xFACu | (7u << 12) | (4u << 15)); return r;
}
struct Layers { const uint8_t* p[34]; };

template <int VEC> struct Reg;
template <> struct Reg<1> { int v; DEV void ld(v4i r, int vo, int so) { v = ld1(r, vo, so, 0); } DEV void x(const Reg& o) { v ^= o.v; } DEV void st(v4i r, int vo, int so) { st1(v, r, vo, so, 0); } DEV void zero() { v = 0; } };
template <> struct Reg<4> { v4i v; DEV void ld(v4i r, int vo, int so) { v = ld4(r, vo, so, 0); } DEV void x(const Reg& o) { v ^= o.v; } DEV void st(v4i r, int vo, int so) { st4(v, r, vo, so, 0); } DEV void zero() { v = v4i{0, 0, 0, 0}; } };

// grid: waves of 64 lanes, WPB per block; every wave walks `upw` consecutive units; a unit = 256·VEC bytes per layer.
template <int VEC, int DEPTH, int SHAPE, int WPB>
__global__ __launch_bounds__(64 * WPB) void k(Layers L, int n_layers, uint32_t layer_bytes, uint32_t upw, uint8_t* dst)
{
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    const uint32_t unit_bytes = 256u * VEC, n_units = layer_bytes / unit_bytes;
    const uint32_t u0 = wave * upw, u1 = min(u0 + upw, n_units);
    const v4i rd = rsrc(dst, layer_bytes);
    const int vo = (int)(lane * 4u * VEC);
    if (SHAPE == 0 || SHAPE == 2) {
        // SHAPE 2: the wave's units are n_waves apart (unit = wave + i * n_waves): all resident waves advance through one compact window of the buffers
        const uint32_t n_waves = (n_units + upw - 1) / upw;
        for (uint32_t uu = u0; uu < u0 + upw; ++uu) {
            const uint32_t u = SHAPE == 2 ? wave + (uu - u0) * n_waves : uu;
            if (u >= n_units || (SHAPE == 2 && wave >= n_waves)) break;
            const int so = (int)(u * unit_bytes);
            Reg<VEC> acc; acc.zero();
            for (int l = 0; l < n_layers; l += DEPTH) {
                Reg<VEC> t[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const int ll = min(l + d, n_layers - 1);
                    const uint8_t* p = L.p[ll];   // uniform: s_load of the pointer
                    t[d].ld(rsrc(p, layer_bytes), vo, so);
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) if (l + d < n_layers) acc.x(t[d]);
            }
            acc.st(rd, vo, so);
        }
    } else {
        // one stream: treat the N buffers as N consecutive pieces PER UNIT of a single walk: wave reads unit (u·N + l) of the concatenation.  The buffers are
        // allocated back to back by the host for this shape (L.p[0] is the base, n_layers · layer_bytes long, < 4 GB per resource → split by layer).
        for (uint32_t u = u0; u < u1; ++u) {
            Reg<VEC> acc; acc.zero();
            const uint64_t first = (uint64_t)u * (uint32_t)n_layers;   // piece index in the concatenation
            for (int l = 0; l < n_layers; l += DEPTH) {
                Reg<VEC> t[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const uint64_t piece = first + (uint32_t)min(l + d, n_layers - 1);
                    const uint32_t buf = (uint32_t)(piece / n_units), pu = (uint32_t)(piece % n_units);
                    t[d].ld(rsrc(L.p[buf], layer_bytes), vo, (int)(pu * unit_bytes));
                }
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) if (l + d < n_layers) acc.x(t[d]);
            }
            acc.st(rd, vo, (int)(u * unit_bytes));
        }
    }
}

__global__ void fill(uint32_t* p, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 0x9E3779B9u + seed; x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; p[i] = x;
    }
}

template <int VEC, int DEPTH, int SHAPE, int WPB>
static void run(const Layers& L, int n_layers, uint32_t layer_bytes, uint8_t* dst, uint32_t upw)
{
    const uint32_t n_units = layer_bytes / (256u * VEC), waves = (n_units + upw - 1) / upw, blocks = (waves + WPB - 1) / WPB;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0);
        k<VEC, DEPTH, SHAPE, WPB><<<blocks, 64 * WPB>>>(L, n_layers, layer_bytes, upw, dst);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float t; hipEventElapsedTime(&t, e0, e1);
        if (it >= 4) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)k<VEC, DEPTH, SHAPE, WPB>);
    const double bytes = (double)layer_bytes * (n_layers + 1);
    printf("{\"shape\": \"%s\", \"layers\": %d, \"bytes_per_lane\": %d, \"depth\": %d, \"waves_per_wg\": %d, \"units_per_wave\": %u, \"vgprs\": %d, \"ms_min\": %.4f, \"ms_med\": %.4f, \"TBs_min\": %.3f, \"TBs_med\": %.3f}\n",
           SHAPE == 1 ? "one stream" : SHAPE == 2 ? "N strided" : "N streams", n_layers, 4 * VEC, DEPTH, WPB, upw, fa.numRegs, ms.front(), ms[ms.size() / 2], bytes / ms.front() * 1e-9, bytes / ms[ms.size() / 2] * 1e-9);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv)
{
    const uint32_t layer_bytes = 7680u * 4320u * 4u;
    const size_t skew = argc > 1 ? (size_t)atol(argv[1]) : 0;
    Layers L;
    // separately allocated layers (hipMalloc each, as the bench and a host application do) …
    for (int l = 0; l < 33; ++l) {
        uint8_t* p; if (hipMalloc(&p, layer_bytes + 33 * skew + 4096) != hipSuccess) { printf("alloc failed\n"); return 1; }
        L.p[l] = p + l * skew;
        fill<<<2048, 256>>>((uint32_t*)L.p[l], layer_bytes / 4, 0x1234u + l);
    }
    uint8_t* dst; hipMalloc(&dst, layer_bytes);
    hipDeviceSynchronize();
    printf("# skew %zu bytes per layer\n", skew);
    if (argc > 2) {   // units-per-wave sweep at the compositor's depth (2) and at 4, contiguous against strided streams
        for (uint32_t upw : {1u, 2u, 4u, 8u, 16u, 32u, 64u}) {
            run<1, 2, 0, 1>(L, 33, layer_bytes, dst, upw); run<1, 2, 2, 1>(L, 33, layer_bytes, dst, upw);
            run<1, 4, 0, 1>(L, 33, layer_bytes, dst, upw); run<1, 4, 2, 1>(L, 33, layer_bytes, dst, upw);
        }
        return 0;
    }
    for (int n : {1, 9, 33}) {
        for (uint32_t upw : {4u, 16u}) {
            run<1, 2, 0, 1>(L, n, layer_bytes, dst, upw);
            run<1, 4, 0, 1>(L, n, layer_bytes, dst, upw);
            run<1, 8, 0, 1>(L, n, layer_bytes, dst, upw);
            run<4, 2, 0, 1>(L, n, layer_bytes, dst, upw);
            run<4, 4, 0, 1>(L, n, layer_bytes, dst, upw);
            run<4, 8, 0, 1>(L, n, layer_bytes, dst, upw);
            run<4, 4, 0, 4>(L, n, layer_bytes, dst, upw);
        }
        run<1, 4, 1, 1>(L, n, layer_bytes, dst, 16u);
        run<4, 4, 1, 1>(L, n, layer_bytes, dst, 16u);
        run<4, 8, 1, 1>(L, n, layer_bytes, dst, 16u);
    }
    return 0;
}
