// Micro-benchmark: does a sustained exact-f32 MFMA stream hold the 2.4 GHz the 157.3 TFLOP/s peak assumes?
// Runs back-to-back launches of a pure v_mfma_f32_16x16x4_f32 chain for ~2 s per data pattern and reports, per
// 250 ms window, the achieved TFLOP/s and the shader clock implied by s_memtime (core clock) / s_memrealtime
// (constant 100 MHz).  (diagnostic tool, not shipped code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__global__ void chain(const float* __restrict__ in, float* out, unsigned long long* clk, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(t * 16 + i) & 0xFFFFF]; b[i] = in[(t * 16 + 8 + i) & 0xFFFFF]; }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = MFMA(a[(i + k) & 7], b[k], acc[i]);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[t] = s;
    if (t == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int N = 1 << 20;
    std::vector<float> h(N);
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, N * 4); hipMalloc(&out, 256 * 512 * 4); hipHostMalloc(&clk, 16);
    const int iters = 4000;                    // 64 MFMA * 4000 * 13.6 ns = 3.5 ms per launch at full clock
    for (int pattern = 0; pattern < 3; ++pattern) {
        srand(1);
        for (int i = 0; i < N; ++i)
            h[i] = pattern == 0 ? 0.0f : pattern == 1 ? 1.0f : (float)rand() / RAND_MAX * 2.0f - 1.0f;
        hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
        for (int wps = 1; wps <= 2; ++wps) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            double elapsed = 0; int win = 0;
            while (elapsed < seconds) {
                hipEventRecord(e0);
                int n = 0; double ratio = 0;
                for (; n < 64; ++n) chain<<<256, 256 * wps>>>(in, out, clk, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                ratio = (double)clk[0] / (double)clk[1];     // core ticks per 10 ns
                const double mfma = 64.0 * iters * wps * 1024 * n;
                if (win % 2 == 0 || elapsed + ms * 1e-3 >= seconds)
                    printf("pattern %s  %d wave/SIMD  t=%5.2fs  %6.1f TFLOP/s   s_memtime/s_memrealtime = %.2f (x100 MHz)  %5.1f ticks/MFMA\n",
                           pattern == 0 ? "zeros " : pattern == 1 ? "ones  " : "random", wps, elapsed, mfma * 2048 / (ms * 1e-3) / 1e12,
                           ratio, (double)clk[0] / (64.0 * iters * wps));
                elapsed += ms * 1e-3; ++win;
            }
        }
    }
    return 0;
}
