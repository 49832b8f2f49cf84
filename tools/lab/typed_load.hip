// tools/lab/typed_load.hip — does gfx950's typed buffer load (buffer_load_format_xyzw, 8_8_8_8 UNORM) return exactly
// RN(k / 255.0f) for every byte value k?  If so the texture path converts a layer pixel to the four f32 operands of
// blend_pixel_static for free (no v_cvt_f32_ubyteN + 2-op div255 per channel).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ v4f llvm_amdgcn_raw_buffer_load_format_v4f32(v4i rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f32");

__global__ void k(const uint32_t* p, float* out, uint32_t nbytes)
{
    const unsigned long long a = (unsigned long long)p;
    v4i r;
    r.x = (int)(uint32_t)a;
    r.y = (int)((uint32_t)(a >> 32) & 0xffffu); // stride 0
    r.z = (int)nbytes;
    r.w = (int)(0xFACu | (0u << 12) | (10u << 15)); // dst_sel x,y,z,w; num_format UNORM; data_format 8_8_8_8
    const v4f v = llvm_amdgcn_raw_buffer_load_format_v4f32(r, (int)(threadIdx.x * 4u), 0, 0);
    out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}

int main()
{
    uint32_t h[256]; float o[1024];
    for (uint32_t i = 0; i < 256; ++i) h[i] = i | ((255u - i) << 8) | ((i ^ 0x55u) << 16) | (((i * 7u) & 255u) << 24);
    uint32_t* d; float* dout;
    hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 256>>>(d, dout, sizeof h);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (uint32_t i = 0; i < 256; ++i) {
        const uint32_t b[4] = {i, 255u - i, i ^ 0x55u, (i * 7u) & 255u};
        for (int c = 0; c < 4; ++c) {
            const float ref = (float)b[c] / 255.0f;
            if (memcmp(&ref, &o[i * 4 + c], 4) != 0) {
                if (bad < 10) printf("byte %u channel %d: got %.9g (0x%08x) want %.9g\n", b[c], c, o[i * 4 + c], *(uint32_t*)&o[i * 4 + c], ref);
                ++bad;
            }
        }
    }
    printf("{\"typed_unorm8_load_mismatches\": %d, \"of\": 1024}\n", bad);
    return bad ? 1 : 0;
}
