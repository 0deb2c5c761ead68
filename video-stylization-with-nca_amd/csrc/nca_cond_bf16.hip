// nca_cond_bf16.hip -- bf16 state storage for the ConditionedNCA step (gfx950): finalize kernel and entry points.
//
// The fused step itself is the producer/consumer kernel of nca_cond_pc.hip instantiated with StBF16 (see the rounding
// points there and in include/ncahip.h): state and goal encoding travel as bf16 (90 B/cell at C=16 instead of 178), the
// UpdateNet runs on v_mfma_f32_16x16x16_bf16 with f32 accumulation.  A first, symmetric wave-private bf16 kernel (every
// wave staging and computing its own tile, 2 waves per SIMD) was latency-bound at 74.6 us/step on the bench grid -- the
// bf16 MFMA is cheap, the dependent LDS / global round trips of the staging are not -- which is why the staging runs in
// its own waves one tile ahead here as well.
// Requires W % 4 == 0 and 8-byte aligned tensors (no any-shape bf16 kernel: the entry points refuse other shapes).
#include "nca_cond_tile.h"

namespace {

__device__ __forceinline__ unsigned pk_bf16_(float lo, float hi) { return StBF16::pk2(lo, hi); }

// x_out = bf16(clamp(x_pend * (pre & alive(x_pend)), lo, hi)): every operation is exact on bf16 values
__global__ __launch_bounds__(256) void cond_finalize_bf16_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ pre,
                                                                 uint16_t* __restrict__ out, int B, int C, int H, int W,
                                                                 int alive_ch, float thr, float lo, float hi) {
    const size_t plane = (size_t)H * W;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * plane) return;
    const int xx = (int)(id % W), yy = (int)((id / W) % H), b = (int)(id / plane);
    const uint16_t* const xb = x + (size_t)b * C * plane;
    auto f = [](uint16_t v) { return __uint_as_float((unsigned)v << 16); };
    float life = 1.0f;
    if (alive_ch >= 0) {
        const uint16_t* const ap = xb + (size_t)alive_ch * plane;
        float m = NCA_NEG_INF;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int y2 = yy + dy, x2 = xx + dx;
                if (y2 >= 0 && y2 < H && x2 >= 0 && x2 < W) m = fmaxf(m, f(ap[(size_t)y2 * W + x2]));
            }
        life = (pre[id] != 0 && m > thr) ? 1.0f : 0.0f;
    }
    const size_t off = (size_t)yy * W + xx;
    uint16_t* const ob = out + (size_t)b * C * plane + off;
    for (int c = 0; c < C; ++c) {
        const float v = fminf(fmaxf(f(xb[(size_t)c * plane + off]) * life, lo), hi);
        ob[(size_t)c * plane] = (uint16_t)(pk_bf16_(v, 0.0f) & 0xffffu);
    }
}

}  // namespace

hipError_t nca_launch_cond_finalize_bf16(const uint16_t* x, const uint8_t* pre, uint16_t* out, int B, int C, int H, int W,
                                         int alive_ch, float thr, float lo, float hi, hipStream_t st) {
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(cond_finalize_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, pre, out, B, C, H, W,
                       alive_ch, thr, lo, hi);
    return hipGetLastError();
}
