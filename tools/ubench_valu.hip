// ubench_valu.hip — VALU issue-rate probe for gfx950 (decides kernel structure: scalar vs packed f32, cost of IEEE divide).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITERS 4096
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

__global__ void k_fma(float* out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkfma(float* out, float a, float b)
{
    float2v av = {a, a}, bv = {b, b};
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    for (int i = 0; i < ITERS; ++i) {
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
        x4 = __builtin_elementwise_fma(x4, av, bv); x5 = __builtin_elementwise_fma(x5, av, bv);
        x6 = __builtin_elementwise_fma(x6, av, bv); x7 = __builtin_elementwise_fma(x7, av, bv);
    }
    float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_muladd(float* out, float a, float b) // separate mul + add (no contraction): 2 VALU per element
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i) {
        x0 = x0 * a + b; x1 = x1 * a + b; x2 = x2 * a + b; x3 = x3 * a + b;
        x4 = x4 * a + b; x5 = x5 * a + b; x6 = x6 * a + b; x7 = x7 * a + b;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pkmuladd(float* out, float a, float b)
{
    float2v av = {a, a}, bv = {b, b};
    float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < ITERS; ++i) {
        x0 = x0 * av + bv; x1 = x1 * av + bv; x2 = x2 * av + bv; x3 = x3 * av + bv;
    }
    float2v s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_div(float* out, float a, float b) // IEEE divide
{
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < ITERS; ++i) { x0 = a / (x0 + b); x1 = a / (x1 + b); x2 = a / (x2 + b); x3 = a / (x3 + b); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}
__global__ void k_cvt(float* out, uint32_t a) // v_cvt_f32_ubyteN
{
    uint32_t p = threadIdx.x * 0x01020304u + a;
    float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int i = 0; i < ITERS; ++i) {
        s0 += (float)(p & 0xff); s1 += (float)((p >> 8) & 0xff); s2 += (float)((p >> 16) & 0xff); s3 += (float)(p >> 24);
        p = p * 1664525u + 1013904223u;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3;
}

template <typename K, typename... A>
double run(const char* name, K kern, double ops_per_thread_iter, float* out, A... args)
{
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, threads>>>(out, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) kern<<<blocks, threads>>>(out, args...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double ops = (double)blocks * threads * ITERS * ops_per_thread_iter;
    printf("%-12s %8.3f ms  %8.2f T elem-ops/s\n", name, ms, ops / ms * 1e-9);
    return ms;
}

int main()
{
    float* out; CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
    run("fma", k_fma, 8, out, 1.0001f, 0.5f);
    run("pk_fma", k_pkfma, 16, out, 1.0001f, 0.5f);
    run("mul+add", k_muladd, 8, out, 1.0001f, 0.5f);
    run("pk_mul+add", k_pkmuladd, 8, out, 1.0001f, 0.5f);
    run("div", k_div, 4, out, 1.0001f, 0.5f);
    run("cvt_ubyte", k_cvt, 4, out, 7u);
    { // sustained: the same FMA loop for ~0.1 s (the clock settles under the power limit), what a VALU-bound kernel can hope for
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int blocks = 256 * 8, threads = 256, reps = 400;
        for (int r = 0; r < 50; ++r) k_fma<<<blocks, threads>>>(out, 1.0001f, 0.5f);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) k_fma<<<blocks, threads>>>(out, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = (double)blocks * threads * ITERS * 8 * reps;
        printf("fma sustained %8.3f ms  %8.2f T elem-ops/s  (= %.3f GHz x 256 CUs x 4 SIMDs x 32 lanes)\n", ms, ops / ms * 1e-9, ops / ms * 1e-9 / (256 * 4 * 32) * 1e3);
    }
    // HBM copy rate
    size_t n = (size_t)1 << 30; void *a, *b; CHK(hipMalloc(&a, n)); CHK(hipMalloc(&b, n));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 5; ++r) hipMemcpyAsync(b, a, n, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("d2d copy 1GiB: %.3f ms -> %.2f TB/s (read+write)\n", ms / 5, 2.0 * n / (ms / 5) * 1e-9);
    return 0;
}
