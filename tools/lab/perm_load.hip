// tools/lab/perm_load.hip — what does a typed buffer load cost when the lanes' addresses are a PERMUTATION of a contiguous span instead of
// lane order?  (The compositor wants a unit's pixels grouped by accumulator class: lane l, group j then holds pixel sigma(64 j + l) of the
// same 192-pixel unit — the same cache lines, a different lane order.)  Streams 32 layers of an 8K image through typed loads, 3 pixels per
// lane, sums the channels (so nothing but the loads and four adds per pixel happens) and stores one dword per pixel.
//   mode 0: lane order                      mode 1: bit-reversed within each 64-pixel group
//   mode 2: s -> 77 s mod 192 (whole unit)  mode 3: stable partition of the unit by a pseudo-random class (p = 0.5)
//   mode 4: partition of a 768-pixel span (4 units: what a FIFO across units does)   mode 5: mode 3 with p = 0.25
//   mode 6: lane order, but each lane's 3 pixels are CONSECUTIVE (12-byte stride per lane)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ v4f ld4(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");
__device__ void st1(int data, v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.i32");
__device__ __forceinline__ v4i rsrc(const void* base, uint32_t bytes, uint32_t w3)
{
    const unsigned long long a = (unsigned long long)base;
    v4i r; r.x = (int)(uint32_t)a; r.y = (int)((uint32_t)(a >> 32) & 0xffffu); r.z = (int)bytes; r.w = (int)w3; return r;
}
struct Layers { const uint8_t* p[32]; };
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int SPAN_UNITS>
__global__ __launch_bounds__(256) void k(Layers L, int n_layers, uint32_t n_px, uint8_t* dst, int mode, const uint16_t* perm /* per span: SPAN px */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t bytes = n_px * 4u;
    const uint32_t span = 192u * SPAN_UNITS;
    const uint32_t n_spans = n_px / span;
    if (wave >= n_spans) return;
    const v4i rd = rsrc(dst, bytes, 0xFACu | (7u << 12) | (4u << 15));
    for (int u = 0; u < SPAN_UNITS; ++u) {
        int voff[3];
        for (int j = 0; j < 3; ++j) {
            const uint32_t s = (uint32_t)u * 192u + 64u * j + lane;      // slot within the span
            uint32_t px;
            if (mode == 0) px = s;
            else if (mode == 1) px = (s & ~63u) | (__brev(s & 63u) >> 26);
            else if (mode == 6) px = (uint32_t)u * 192u + lane * 3u + j;
            else px = perm[(size_t)wave * span + s];
            voff[j] = (int)((wave * span + px) * 4u);
        }
        float a0 = 0, a1 = 0, a2 = 0;
        for (int l = 0; l < n_layers; ++l) {
            const v4i rs = rsrc(L.p[l], bytes, 0xFACu | (0u << 12) | (10u << 15));
            const v4f x = ld4(rs, voff[0], 0, 0), y = ld4(rs, voff[1], 0, 0), z = ld4(rs, voff[2], 0, 0);
            a0 += x.x + x.y + x.z + x.w; a1 += y.x + y.y + y.z + y.w; a2 += z.x + z.y + z.z + z.w;
        }
        st1(__builtin_bit_cast(int, a0), rd, voff[0], 0, 0); st1(__builtin_bit_cast(int, a1), rd, voff[1], 0, 0); st1(__builtin_bit_cast(int, a2), rd, voff[2], 0, 0);
    }
}

int main(int argc, char** argv)
{
    const uint32_t w = 7680, h = 4320, n_px = w * h;
    const int n_layers = 32;
    Layers L;
    for (int l = 0; l < n_layers; ++l) { hipMalloc((void**)&L.p[l], (size_t)n_px * 4); hipMemset((void*)L.p[l], l + 1, (size_t)n_px * 4); }
    uint8_t* dst; hipMalloc(&dst, (size_t)n_px * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode <= 6; ++mode) {
        const int span_units = mode == 4 ? 4 : 1;
        const uint32_t span = 192u * span_units, n_spans = n_px / span;
        std::vector<uint16_t> perm((size_t)n_spans * span);
        uint32_t rng = 12345u;
        for (uint32_t sp = 0; sp < n_spans; ++sp) {
            uint16_t* p = &perm[(size_t)sp * span];
            if (mode == 2) for (uint32_t s = 0; s < span; ++s) p[s] = (uint16_t)((s * 77u) % 192u);
            else if (mode >= 3) {
                const uint32_t thr = mode == 5 ? 0x40000000u : 0x80000000u;
                uint32_t k0 = 0;
                std::vector<uint8_t> cls(span);
                for (uint32_t s = 0; s < span; ++s) { rng = rng * 1664525u + 1013904223u; cls[s] = rng < thr; }
                for (uint32_t s = 0; s < span; ++s) if (cls[s]) p[k0++] = (uint16_t)s;
                for (uint32_t s = 0; s < span; ++s) if (!cls[s]) p[k0++] = (uint16_t)s;
            }
        }
        uint16_t* dperm; hipMalloc(&dperm, perm.size() * 2); hipMemcpy(dperm, perm.data(), perm.size() * 2, hipMemcpyHostToDevice);
        const uint32_t blocks = (n_spans + 3) / 4;
        float best = 1e9f;
        for (int it = 0; it < 12; ++it) {
            hipEventRecord(e0);
            if (span_units == 4) k<4><<<blocks, 256>>>(L, n_layers, n_px, dst, mode, dperm);
            else k<1><<<blocks, 256>>>(L, n_layers, n_px, dst, mode, dperm);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 2 && ms < best) best = ms;
        }
        printf("{\"mode\": %d, \"ms\": %.4f, \"TBs\": %.3f}\n", mode, best, (double)n_px * 4 * (n_layers + 1) / best * 1e-9);
        hipFree(dperm);
    }
    return 0;
}
