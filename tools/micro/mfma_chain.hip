// Micro-benchmark: what does a v_mfma_f32_16x16x4_f32 chain sustain per SIMD under the dependency patterns
// the fused NCA kernels use?  (diagnostic tool, not shipped code)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int NACC, int WPS>   // NACC independent accumulators; WPS waves per SIMD via block size
__global__ void chain(float* out, int iters) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 64 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MFMA(a, b, acc[i]);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the fused kernel's pattern: per hidden tile, 24 MFMAs into 2 accumulators, relu, 32 MFMAs into 8 accumulators
template <bool RELU_DEP>
__global__ void mlp_pattern(float* out, int iters) {
    f32x4 acc2[8];
    for (int i = 0; i < 8; ++i) acc2[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, p0 = 1.0f + threadIdx.x * 1e-4f, p1 = 0.5f;
    for (int it = 0; it < iters; ++it) {
        f32x4 acc1[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            acc1[0] = MFMA(a, p0, acc1[0]);
            acc1[1] = MFMA(a, p1, acc1[1]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float bb = RELU_DEP ? __int_as_float(max(__float_as_int(acc1[n][r]), 0)) : p0;
                    acc2[m2 * 2 + n] = MFMA(a, bb, acc2[m2 * 2 + n]);
                }
        if (!RELU_DEP) { p0 += acc1[0][0] * 1e-30f; p1 += acc1[1][0] * 1e-30f; }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc2[i][0] + acc2[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, int mfma_per_iter, int iters, int waves_per_simd) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double mf = (double)mfma_per_iter * iters * waves_per_simd;   // MFMAs per SIMD
    printf("%-44s %8.3f ms  %7.1f ns/MFMA/SIMD  = %5.1f cycles @2.4GHz   %6.1f TFLOP/s\n", name, ms, ms * 1e6 / mf,
           ms * 1e6 / mf * 2.4, 1024.0 * mf * 2048 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int it = 2000;
    run("2 acc, 1 wave/SIMD", [&] { chain<2, 1><<<256, 256>>>(out, it); }, 64, it, 1);
    run("4 acc, 1 wave/SIMD", [&] { chain<4, 1><<<256, 256>>>(out, it); }, 64, it, 1);
    run("8 acc, 1 wave/SIMD", [&] { chain<8, 1><<<256, 256>>>(out, it); }, 64, it, 1);
    run("1 acc, 1 wave/SIMD", [&] { chain<1, 1><<<256, 256>>>(out, it); }, 64, it, 1);
    run("2 acc, 2 waves/SIMD", [&] { chain<2, 2><<<256, 512>>>(out, it); }, 64, it, 2);
    run("8 acc, 2 waves/SIMD", [&] { chain<8, 2><<<256, 512>>>(out, it); }, 64, it, 2);
    run("mlp pattern, relu dep, 1 wave/SIMD", [&] { mlp_pattern<true><<<256, 256>>>(out, it); }, 56, it, 1);
    run("mlp pattern, no relu dep, 1 wave/SIMD", [&] { mlp_pattern<false><<<256, 256>>>(out, it); }, 56, it, 1);
    run("mlp pattern, relu dep, 2 waves/SIMD", [&] { mlp_pattern<true><<<256, 512>>>(out, it); }, 56, it, 2);
    return 0;
}
