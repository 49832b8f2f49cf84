// tools/lab/hip_start.hip — what a process pays before its first kernel: hipFree(0) (runtime + context), then a first trivial launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(int* p) { *p = 1; }
int main()
{
    auto t0 = std::chrono::steady_clock::now();
    (void)hipFree(nullptr);
    auto t1 = std::chrono::steady_clock::now();
    int* d; (void)hipMalloc(&d, 4); k<<<1, 1>>>(d); (void)hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("hipFree(0) %.1f ms, first launch %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count());
}
