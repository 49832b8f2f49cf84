// tools/lab/cvt_pk.hip — rounding / saturation behaviour of v_cvt_pk_u8_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}
int main() {
    const float v[] = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.0f, 255.4f, 255.5f, 256.0f, 300.0f, -0.4f, -0.6f, -3.0f, 1e9f, 127.49999f, 127.5f, 128.5f};
    const int n = sizeof v / sizeof v[0];
    float* d; unsigned* o; unsigned h[64];
    hipMalloc(&d, sizeof v); hipMalloc(&o, 256);
    hipMemcpy(d, v, sizeof v, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%g -> %u\n", v[i], h[i]);
    return 0;
}
