// Micro-benchmark: cost of one workgroup barrier per loop iteration for an 8-wave workgroup (1 per CU), with half the
// waves at raised priority as in the producer/consumer kernel.  (diagnostic tool, not shipped code)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int iters) {
    extern __shared__ float smem[];
    const int wave = threadIdx.x >> 6;
    if (MODE >= 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    float acc = 0;
    const unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 2) { smem[threadIdx.x] = acc; acc += smem[(threadIdx.x + 64) & 511]; }
        __syncthreads();
    }
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipHostMalloc(&clk, 16);
    const int iters = 2000;
    auto run = [&](const char* name, auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(256), dim3(512), 150 * 1024, 0, out, clk, iters); (void)hipDeviceSynchronize(); }
        printf("%-40s %7.1f cycles per iteration\n", name, (double)clk[0] / iters);
    };
    run("barrier only", k<0>);
    run("barrier, waves 4-7 at prio 3", k<1>);
    run("barrier + LDS write/read, prio split", k<2>);
    return 0;
}
