// nca_common.h -- device helpers shared by the NCA kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NCA_PAD_ZERO 0
#define NCA_PAD_REPLICATE 1
#define NCA_PAD_CIRCULAR 2
#define NCA_PAD_REFLECT 3

#define NCA_WAVE 64
#define NCA_NEG_INF (-__builtin_huge_valf())

// Source index for padded position i (any int) on an axis of length n, F.pad semantics
// (ConditioneDyNCA/models/dynca.py:85).  Returns -1 when the position contributes zero
// ('constant' mode outside the image).  The result is always a safe index otherwise, also
// for positions more than one cell outside (those are never consumed by an in-image cell).
__device__ __forceinline__ int nca_pad_index(int i, int n, int mode) {
    if (i >= 0 && i < n) return i;
    if (mode == NCA_PAD_ZERO) return -1;
    if (mode == NCA_PAD_CIRCULAR) {
        int r = i % n;
        return r < 0 ? r + n : r;
    }
    if (mode == NCA_PAD_REFLECT) {
        int r = i < 0 ? -i : 2 * (n - 1) - i;
        return r < 0 ? 0 : (r > n - 1 ? n - 1 : r);
    }
    return i < 0 ? 0 : n - 1;  // replicate
}

// ---- fixed DyNCA filters (dynca.py:67-73), cross-correlation on the 3x3 neighbourhood
// a[dy][dx], dy/dx in {0,1,2} <-> offsets {-1,0,+1}.
__device__ __forceinline__ float nca_sobel_x(const float (&a)[3][3]) {
    return (a[0][2] - a[0][0]) + 2.0f * (a[1][2] - a[1][0]) + (a[2][2] - a[2][0]);
}
__device__ __forceinline__ float nca_sobel_y(const float (&a)[3][3]) {
    return (a[2][0] - a[0][0]) + 2.0f * (a[2][1] - a[0][1]) + (a[2][2] - a[0][2]);
}
__device__ __forceinline__ float nca_laplacian(const float (&a)[3][3]) {
    return ((a[0][0] + a[0][2]) + (a[2][0] + a[2][2])) +
           2.0f * ((a[0][1] + a[2][1]) + (a[1][0] + a[1][2])) - 12.0f * a[1][1];
}

// Two-scale perception (dynca.py:98, :105-110): bilinear x2 up-sampling (align_corners = False) of the coarse-level value from its
// four taps, then the mean over the two scales.  ONE definition with floating-point contraction off: the per-step kernel and the
// persistent kernel must produce the same bits, and left to the optimiser the same source expression was contracted into different
// fma / mul+add sequences in the two kernels (1-ulp differences on odd rows).
__device__ __forceinline__ float nca_up2_blend(float fine, float q00, float q01, float q10, float q11, float h0, float h1, float w0, float w1) {
#pragma clang fp contract(off)
    const float top = w0 * q00 + w1 * q01;
    const float bot = w0 * q10 + w1 * q11;
    const float up = h0 * top + h1 * bot;
    return (fine + up) / 2.0f;
}

// ---- Philox4x32-10 fire-mask stream (framework contract, restated in oracle/nca_oracle.py):
// key = (seed_lo, seed_hi), counter = (cell >> 2, step_lo, step_hi, 'NCA'), word = cell & 3,
// u = (word >> 8) * 2^-24.
__device__ __forceinline__ uint4 nca_philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        // one 32x32->64 multiply per round half (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: both are quarter rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ uint4 nca_philox_group(uint64_t seed, uint64_t step, uint32_t group) {
    return nca_philox4x32_10(make_uint4(group, (uint32_t)step, (uint32_t)(step >> 32), 0x4E4341u),
                             make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ float nca_u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }
__device__ __forceinline__ float nca_philox_cell(uint64_t seed, uint64_t step, uint64_t cell) {
    const uint4 r = nca_philox_group(seed, step, (uint32_t)(cell >> 2));
    const uint32_t l = (uint32_t)cell & 3u;
    return nca_u01(l == 0 ? r.x : (l == 1 ? r.y : (l == 2 ? r.z : r.w)));
}

// ---- exact-f32 MFMA: D(16x16) += A(16x4) * B(4x16).  Lane l = 16*g + i:
//   a = A[row i][k g], b = B[k g][col i], d[r] = D[row 4g + r][col i]   (guide 3, "gfx950 intrinsic list")
__device__ __forceinline__ f32x4 nca_mfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// XCD-aware persistent tile schedule: workgroups b and b+8 share an XCD (its L2), so each XCD
// group walks one contiguous chunk of the tile list (spatially adjacent tiles share halos).
struct NcaTileWalk {
    int t, end, stride;
    int base, local;   // t = base + local: first tile of this XCD's chunk, rank of the workgroup inside the XCD
};
__device__ __forceinline__ NcaTileWalk nca_tile_walk(int ntiles) {
    const int nwg = gridDim.x, wg = blockIdx.x;
    const int nx = nwg < 8 ? nwg : 8;
    const int xcd = wg % nx, local = wg / nx;
    const int nloc = (nwg - xcd + nx - 1) / nx;
    const int chunk = (ntiles + nx - 1) / nx;
    NcaTileWalk w;
    w.t = xcd * chunk + local;
    w.end = min(ntiles, (xcd + 1) * chunk);
    w.stride = nloc;
    w.base = xcd * chunk;
    w.local = local;
    return w;
}
