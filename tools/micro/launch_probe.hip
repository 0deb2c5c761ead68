// Micro-benchmark: fixed cost of a launch shaped like the fused step (256 WGs x 512 threads, 157 KB dynamic LDS)
// with an empty body vs a small-LDS one; back-to-back cadence from HIP events.  (diagnostic tool, not shipped code)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void empty_kernel(float* out, int spin) {
    extern __shared__ float smem[];
    if (spin) {
        const unsigned long long t0 = clock64();
        while (clock64() - t0 < (unsigned long long)spin) { }
    }
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = smem[0];
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lds_kb : {0, 64, 157}) {
        for (int spin : {0, 24000, 240000}) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), lds_kb * 1024, 0, nullptr, spin);
            (void)hipDeviceSynchronize();
            const int n = 200;
            (void)hipEventRecord(e0);
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), lds_kb * 1024, 0, nullptr, spin);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("lds %3d KB  spin %6d cycles (%.1f us)  cadence %.2f us/launch  -> overhead %.2f us\n", lds_kb, spin, spin / 2400.0, ms * 1e3 / n,
                   ms * 1e3 / n - spin / 2400.0);
        }
    }
    return 0;
}
