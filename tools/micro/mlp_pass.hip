// Micro-benchmark: the consumer wave's MLP pass (mlp_tile_regs, 256 exact-f32 MFMAs for 2 rows of 16 cells) in
// isolation, one wave per SIMD, nothing else on the CU.  Reports s_memtime cycles per pass and per MFMA.
// Variants isolate what separates it from the bare-chain 32.0 cycles/MFMA.  (diagnostic tool, not shipped code)
#include "../../video-stylization-with-nca_amd/csrc/nca_cond_tile.h"
#include <cstdio>
#include <vector>

template <int VAR>
__global__ __launch_bounds__(256, 2) void pass_kernel(const float* __restrict__ in, float* out, unsigned long long* clk, int iters) {
    constexpr int CP = 16, NT = 2;
    using K = WCfg<CP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K::SHARED + 4 * 2048; i += 256) smem[i] = in[(i * 7 + blockIdx.x) & 0xFFFFF];
    __syncthreads();
    MlpRegs<CP> Wr;
    mlp_load_regs<CP>(smem, lane, Wr);
    float* XR = smem + K::SHARED + wave * 2048;
    float* MK = XR + 16 * XRS;
    float P[NT][K::K1S];
    for (int n = 0; n < NT; ++n)
        for (int s = 0; s < K::K1S; ++s) P[n][s] = in[(lane * 31 + n * 17 + s) & 0xFFFFF];
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0) mlp_tile_regs<CP, NT>(Wr, smem, XR, MK, lane, 0, P);
        if (VAR == 1) {   // same MFMA count and dependency structure, no relu
            f32x4 acc2[4][NT];
            for (int m2 = 0; m2 < 4; ++m2) for (int n = 0; n < NT; ++n) acc2[m2][n] = ld4(smem + K::OFF_B2 + 16 * m2 + 4 * (lane >> 4));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc1[NT];
                for (int n = 0; n < NT; ++n) acc1[n] = ld4(smem + K::OFF_B1 + 16 * m + 4 * (lane >> 4));
#pragma unroll
                for (int s = 0; s < K::K1S; ++s)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(Wr.w1[m][s >> 2][s & 3], P[n][s], acc1[n]);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc2[m2][n] = nca_mfma(Wr.w2[m2][m][r], acc1[n][r], acc2[m2][n]);
            }
            f32x4 acc3[NT] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc3[n] = nca_mfma(Wr.w3[0][m][r], acc2[m][n][r], acc3[n]);
            for (int n = 0; n < NT; ++n) XR[lane + 64 * n] += acc3[n][0] + acc3[n][1] + acc3[n][2] + acc3[n][3];
        }
        if (VAR == 2 || VAR == 3) {   // relus of one hidden tile issued as ONE group (VAR 3: fenced with sched_barrier)
            f32x4 acc2[4][NT];
            for (int m2 = 0; m2 < 4; ++m2) for (int n = 0; n < NT; ++n) acc2[m2][n] = ld4(smem + K::OFF_B2 + 16 * m2 + 4 * (lane >> 4));
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 acc1[NT];
                for (int n = 0; n < NT; ++n) acc1[n] = ld4(smem + K::OFF_B1 + 16 * m + 4 * (lane >> 4));
#pragma unroll
                for (int s = 0; s < K::K1S; ++s)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(Wr.w1[m][s >> 2][s & 3], P[n][s], acc1[n]);
                float h[NT][4];
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[n][r] = relu(acc1[n][r]);
                if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc2[m2][n] = nca_mfma(Wr.w2[m2][m][r], h[n][r], acc2[m2][n]);
            }
            float h2[4][NT][4];
            if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) h2[m][n][r] = relu(acc2[m][n][r]);
            if (VAR == 3) __builtin_amdgcn_sched_barrier(0);
            f32x4 acc3[NT] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc3[n] = nca_mfma(Wr.w3[0][m][r], h2[m][n][r], acc3[n]);
            for (int n = 0; n < NT; ++n) XR[lane + 64 * n] += acc3[n][0] + acc3[n][1] + acc3[n][2] + acc3[n][3];
        }
        asm volatile("" ::: "memory");
    }
    const unsigned long long c1 = clock64();
    out[blockIdx.x * 256 + tid] = XR[lane];
    if (tid == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    srand(1);
    for (int i = 0; i < N; ++i) h[i] = (float)rand() / RAND_MAX * 0.2f - 0.1f;
    float *in, *out; unsigned long long* clk;
    (void)hipMalloc(&in, N * 4); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipHostMalloc(&clk, 16);
    (void)hipMemcpy(in, h.data(), N * 4, hipMemcpyHostToDevice);
    const int iters = 400;
    const size_t lds = (WCfg<16>::SHARED + 4 * 2048) * 4;
    auto run = [&](const char* name, auto kern) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(kern, dim3(256), dim3(256), lds, 0, in, out, clk, iters); (void)hipDeviceSynchronize(); }
        printf("%-40s %8.0f cycles/pass   %5.1f cycles/MFMA\n", name, (double)clk[0] / iters, (double)clk[0] / iters / 256.0);
    };
    run("mlp_tile_regs (shipped)", pass_kernel<0>);
    run("same MFMA structure, no relu/residual", pass_kernel<1>);
    run("relu per tile, compiler-scheduled", pass_kernel<2>);
    run("relu per tile, fenced groups", pass_kernel<3>);
    return 0;
}
