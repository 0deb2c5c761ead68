// nca_dynca_persist.hip -- T DyNCA steps in ONE launch for small grids (B = 1 video inference: ConditioneDyNCA/utils/misc/
// video_utils.py:50-82 runs forward_nsteps(h, step_n, cond_img=frame) per frame; WebGL twin docs/dynca.js:1057-1132), gfx950, fp32.
//
// At 1 x 256 x 256 one step is ONE 8 x 32 tile per CU and one wave per SIMD: 408 exact-f32 MFMAs per wave = 5.4 us, and the
// per-step launch adds ~13 us of launch gap, weight-image fill, cold first touch and tail to it (18.7 us per step, C = 12 /
// fc = 96).  Here a workgroup OWNS its tile for all T steps:
//   * weight images and the conditioning tile are built once; the tile's state stays in LDS (two buffers, ping-pong) and only the
//     one-cell halo ring (84 cells x C) is re-read per step;
//   * neighbours synchronise through per-tile monotonic step counters in global memory (tools/micro/neighbour_sync.hip measured
//     the protocol: 6 us per step for publish + poll + ring read): step t starts when the <= 8 neighbours have published step t,
//     i.e. their state t is in memory -- and, because a workgroup reads its ring BEFORE it computes and publishes, that also
//     means they are done READING my state t-1, so the ping-pong buffer holding it may be overwritten with state t+1;
//   * tiles live in different XCDs' L2s: state stores are write-through and ring loads coherent at agent scope (sc1), the
//     counter is stored relaxed after s_waitcnt vmcnt(0) + workgroup barrier (every lane's stores acknowledged);
//   * every poll is BOUNDED: an expired poll (a neighbour that never became resident) sets bit 1 of the sticky device error word
//     and every workgroup leaves its step loop at the next step -- the launch always drains, the host sees NCAHIP_EDEVICE.
// The host side launches this only when every workgroup can be co-resident (occupancy x CUs >= tiles), H % 8 == 0, W % 32 == 0,
// C <= 16, fc <= 128; anything else runs the per-step kernels (nca_step_fwd.hip).  Same arithmetic in the same order as the
// per-step kernel (perception from the LDS tile, exact-f32 MFMA chains with the same k order): bit-identical results.
#include "nca_common.h"
#include "nca_kernels.h"

namespace {

constexpr int kPT = 256;                     // threads per workgroup
constexpr int PTH = 8, PTW = 32;             // tile
constexpr int PROWS = PTH + 2, PRS = PTW + 8, PCS = 400;   // Z[ch][row 0..9][col: image col tx0-1+q at index q+3]; 400 % 32 == 16
static_assert(PROWS * PRS <= PCS && PCS % 32 == 16, "tile carve");
constexpr int kRing = 2 * (PTW + 2) + 2 * PTH;             // 84 halo cells

template <int CP, int FC, bool HAS_COND>
struct PersistCfg {
    static constexpr int K1S = CP + (HAS_COND ? 1 : 0);
    static constexpr int M1T = FC / 16, K2S = FC / 4;
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + M1T * K1S * 64;
    static constexpr int OFF_B1 = OFF_W2 + K2S * 64;
    static constexpr int OFF_B2 = OFF_B1 + FC;
    static constexpr int OFF_Z = OFF_B2 + 16;
    static constexpr int OFF_MK = OFF_Z + 2 * CP * PCS;
    static constexpr int OFF_CN = OFF_MK + PTH * PTW;
    static constexpr int LDS_FLOATS = OFF_CN + (HAS_COND ? 4 * PTH * PTW : 0);
    static constexpr int NLD = (kRing * CP + kPT - 1) / kPT;   // ring loads per thread and step
    static_assert(CP % 4 == 0 && CP <= 16 && FC % 16 == 0 && OFF_Z % 4 == 0, "shape");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ float ld_coherent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_through(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CP, int FC, bool HAS_COND>
__global__ __launch_bounds__(kPT, 1) void dynca_persist_kernel(const NcaDyncaPersistArgs a) {
    using K = PersistCfg<CP, FC, HAS_COND>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const W1L = smem + K::OFF_W1;
    float* const W2L = smem + K::OFF_W2;
    float* const B1L = smem + K::OFF_B1;
    float* const B2L = smem + K::OFF_B2;
    float* const MK = smem + K::OFF_MK;
    float* const CN = smem + K::OFF_CN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ci = lane & 15;
    const int C = a.C, H = a.H, W = a.W, fc = a.fc, CC = a.c_cond, K1 = 4 * C + CC;
    const size_t plane = (size_t)H * W, slot = (size_t)a.B * C * plane;
    const int tiles_x = W / PTW, tiles_y = H / PTH;
    const int tile = blockIdx.x, txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int ty0 = tyi * PTH, tx0 = txi * PTW;

    // ---- once per launch: A-operand weight images (same layouts and k order as dynca_step_fwd_kernel) ---------------------
    for (int idx = tid; idx < K::M1T * K::K1S * 64; idx += kPT) {
        const int l = idx & 63, s = (idx >> 6) % K::K1S, m = (idx >> 6) / K::K1S;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        long src = -1;
        if (o < fc) {
            if (s < CP) {   // k-step s = 4c'+f: channel 4c'+g, filter f (0 id, 1 sobel_x, 2 sobel_y, 3 lap); blocked [x|Sx|Sy|L], dynca.py:92-95
                const int ch = (s & ~3) + gg;
                if (ch < C) src = (long)o * K1 + (s & 3) * C + ch;
            } else if (gg < CC) src = (long)o * K1 + 4 * C + gg;
        }
        W1L[idx] = src >= 0 ? a.w1[src] : 0.0f;
    }
    for (int idx = tid; idx < K::K2S * 64; idx += kPT) {
        const int l = idx & 63, s = idx >> 6;
        const int gg = l >> 4, o = l & 15, k = 16 * (s >> 2) + 4 * gg + (s & 3);
        W2L[idx] = (o < C && k < fc) ? a.w2[(long)o * fc + k] : 0.0f;
    }
    for (int idx = tid; idx < FC; idx += kPT) B1L[idx] = idx < fc ? a.b1[idx] : 0.0f;
    if (tid < 16) B2L[tid] = tid < C ? a.b2[tid] : 0.0f;

    // this thread's cell (mask, conditioning): one cell per thread
    const int cr = tid / PTW, cq = tid % PTW;
    const size_t cell = (size_t)b * plane + (size_t)(ty0 + cr) * W + tx0 + cq;
    if (HAS_COND) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
            CN[cc * PTH * PTW + tid] = cc < CC ? a.cond[((size_t)b * CC + cc) * plane + (cell - (size_t)b * plane)] : 0.0f;
    }
    // neighbours whose step counters gate this tile (circular padding wraps; the other modes have no tile beyond the border)
    int nb = -1;
    if (tid < 8) {
        const int k = tid < 4 ? tid : tid + 1, dy = k / 3 - 1, dx = k % 3 - 1;
        int ny = tyi + dy, nx = txi + dx;
        if (a.pad_mode == NCA_PAD_CIRCULAR) {
            ny = (ny + tiles_y) % tiles_y;
            nx = (nx + tiles_x) % tiles_x;
        }
        if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) nb = (b * tiles_y + ny) * tiles_x + nx;
    }
    // ring items of this thread: item j = tid + 256 k  <->  halo cell j % 84 of channel j / 84; source offset once (pad resolved)
    int rz[K::NLD];        // LDS offset inside a Z buffer, or -1
    unsigned rsrc[K::NLD]; // element offset inside a batch item's state, or ~0u: contributes zero
#pragma unroll
    for (int k = 0; k < K::NLD; ++k) {
        const int j = tid + kPT * k, hc = j % kRing, ch = j / kRing;
        int r, q;
        if (hc < PTW + 2) { r = 0; q = hc; }
        else if (hc < 2 * (PTW + 2)) { r = PROWS - 1; q = hc - (PTW + 2); }
        else if (hc < 2 * (PTW + 2) + PTH) { r = hc - 2 * (PTW + 2) + 1; q = 0; }
        else { r = hc - 2 * (PTW + 2) - PTH + 1; q = PTW + 1; }
        const bool live = j < kRing * CP;
        rz[k] = live ? ch * PCS + r * PRS + q + 3 : -1;
        const int sy = nca_pad_index(ty0 - 1 + r, H, a.pad_mode), sx = nca_pad_index(tx0 - 1 + q, W, a.pad_mode);
        rsrc[k] = (live && ch < C && sy >= 0 && sx >= 0) ? (unsigned)((size_t)ch * plane + (size_t)sy * W + sx) : ~0u;
    }
    // the tile's interior at step 0 (plain loads: written before the launch)
    {
        float* const Z0 = smem + K::OFF_Z;
        const float* const xb = a.states + (size_t)b * C * plane;
        for (int i = tid; i < CP * PTH * (PTW / 4); i += kPT) {
            const int f4 = i % (PTW / 4), r = (i / (PTW / 4)) % PTH, ch = i / (PTH * (PTW / 4));
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < C) v = *reinterpret_cast<const f32x4*>(xb + (size_t)ch * plane + (size_t)(ty0 + r) * W + tx0 + 4 * f4);
            *reinterpret_cast<f32x4*>(Z0 + ch * PCS + (r + 1) * PRS + 4 + 4 * f4) = v;
        }
    }

    // abort word: flags[ntiles] (device memory; the sticky error word itself is host-mapped -- polling THAT from every workgroup and
    // step is a PCIe read storm: 180 us per step).  Lane 8 reads it while lanes 0..7 poll the neighbours; one LDS word tells the rest.
    __shared__ int s_abort;
    int* const abort_w = a.flags + gridDim.x;
    if (tid == 0) s_abort = 0;
    for (int t = 0; t < a.T; ++t) {
        const float* const src = a.states + (size_t)(t & 1) * slot + (size_t)b * C * plane;
        float* const dst = a.states + (size_t)((t + 1) & 1) * slot + (size_t)b * C * plane;
        float* const Zc = smem + K::OFF_Z + (t & 1) * (CP * PCS);
        float* const Zn = smem + K::OFF_Z + ((t + 1) & 1) * (CP * PCS);
        // fire mask of this step (dynca.py:131): explicit uniforms, bit-packed masks, or in-kernel Philox
        {
            float uu;
            if (a.u) {
                const size_t cells = (size_t)a.B * plane;
                if (a.u_bits) uu = ((reinterpret_cast<const uint32_t*>(a.u)[(size_t)t * ((cells + 31) / 32) + (cell >> 5)] >> (unsigned)(cell & 31)) & 1u) ? 1.0f : 0.0f;
                else uu = a.u[(size_t)t * cells + cell];
            } else uu = nca_philox_cell(a.seed, a.step0 + (uint64_t)t, cell);
            MK[tid] = floorf(uu + a.rate);
        }
        // ---- neighbours have published state t (bounded poll) ----------------------------------------------------------------
        if (t > 0 && nb >= 0) {
            int spins = 0;
            while (__hip_atomic_load(a.flags + nb, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < t && ++spins < (1 << 16)) __builtin_amdgcn_s_sleep(2);
            if (spins >= (1 << 16)) {   // the neighbour never published (not resident?): record it, tell every workgroup to drain
                if (a.err) __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(abort_w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
            }
        }
        if (t > 0 && tid == 8 && __hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) s_abort = 1;
        __syncthreads();
        if (s_abort) break;      // (uniform: written before the barrier, never cleared)
        // ---- halo ring of state t: coherent loads (the neighbours' tiles sit in other XCDs' L2s), all requested together ------
        {
            float hv[K::NLD];
#pragma unroll
            for (int k = 0; k < K::NLD; ++k) hv[k] = rsrc[k] != ~0u ? ld_coherent(src + rsrc[k]) : 0.0f;
#pragma unroll
            for (int k = 0; k < K::NLD; ++k)
                if (rz[k] >= 0) Zc[rz[k]] = hv[k];
        }
        __syncthreads();
        // ---- the step for this wave's four 16-cell groups: perception -> MLP on MFMA -> residual ------------------------------
        constexpr int NT = 4;
        int r0[NT], q0[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int j = wave * NT + n;
            r0[n] = j / (PTW / 16);
            q0[n] = (j % (PTW / 16)) * 16 + ci;
        }
        float P[NT][K::K1S];
#pragma unroll
        for (int cq4 = 0; cq4 < CP / 4; ++cq4) {
            const float* const zc = Zc + (4 * cq4 + g) * PCS + 3;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float nbv[3][3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) nbv[dy][dx] = zc[(r0[n] + dy) * PRS + q0[n] + dx];
                P[n][4 * cq4 + 0] = nbv[1][1];
                P[n][4 * cq4 + 1] = nca_sobel_x(nbv);
                P[n][4 * cq4 + 2] = nca_sobel_y(nbv);
                P[n][4 * cq4 + 3] = nca_laplacian(nbv);
            }
        }
        if (HAS_COND) {
#pragma unroll
            for (int n = 0; n < NT; ++n) P[n][CP] = CN[g * PTH * PTW + r0[n] * PTW + q0[n]];
        }
        f32x4 acc2[NT];
        {
            const f32x4 bias = *reinterpret_cast<const f32x4*>(B2L + 4 * g);
#pragma unroll
            for (int n = 0; n < NT; ++n) acc2[n] = bias;
        }
        float wa1[K::K1S], wa2[4];
        f32x4 bias1;
        auto fetch = [&](int m) {
            const float* const w1m = W1L + m * K::K1S * 64 + lane;
#pragma unroll
            for (int s = 0; s < K::K1S; ++s) wa1[s] = w1m[s * 64];
            const float* const w2m = W2L + (4 * m) * 64 + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) wa2[r] = w2m[r * 64];
            bias1 = *reinterpret_cast<const f32x4*>(B1L + 16 * m + 4 * g);
        };
        fetch(0);
#pragma unroll 1
        for (int m = 0; m < K::M1T; ++m) {
            f32x4 acc1[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc1[n] = bias1;
#pragma unroll
            for (int s = 0; s < K::K1S; ++s)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(wa1[s], P[n][s], acc1[n]);
            float w2c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w2c[r] = wa2[r];
            __builtin_amdgcn_sched_barrier(0);
            if (m + 1 < K::M1T) fetch(m + 1);        // in flight across this tile's layer-2 MFMAs and the next chain
            float h[NT][4];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) h[n][r] = __int_as_float(max(__float_as_int(acc1[n][r]), 0));   // relu
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc2[n] = nca_mfma(w2c[r], h[n][r], acc2[n]);
        }
        // residual + stochastic mask (dynca.py:131-133): state t+1 -> memory (write-through) and -> the next step's LDS tile
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const float mk = MK[r0[n] * PTW + q0[n]];
            const size_t o0 = (size_t)(ty0 + r0[n]) * W + tx0 + q0[n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 4 * g + r;
                if (ch < CP) {
                    const int zo = ch * PCS + (r0[n] + 1) * PRS + q0[n] + 4;
                    const float xn = Zc[zo] + acc2[n][r] * mk;
                    Zn[zo] = ch < C ? xn : 0.0f;
                    if (ch < C) st_through(dst + (size_t)ch * plane + o0, xn);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);      // this lane's stores have been acknowledged
        __syncthreads();                    // ... and every lane's; Zn complete, Zc free
        if (tid == 0) __hip_atomic_store(a.flags + tile, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int CP, int FC, bool HAS_COND>
hipError_t launch_persist(const NcaDyncaPersistArgs& a, hipStream_t st, bool query_only, bool* fits) {
    using K = PersistCfg<CP, FC, HAS_COND>;
    auto kern = dynca_persist_kernel<CP, FC, HAS_COND>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int ntiles = a.B * (a.H / PTH) * (a.W / PTW);
    int per_cu = 0;
    if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kPT, lds); e != hipSuccess) return e;
    *fits = (long)per_cu * nca_cu_count() >= ntiles;      // every workgroup must be resident at once (neighbours wait for each other)
    if (!*fits || query_only) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(kPT), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// Shapes the persistent kernel covers (the co-residency test needs the device: done at launch).
bool nca_dynca_persist_shape_ok(int B, int C, int H, int W, int fc, int c_cond) {
    return B >= 1 && C >= 1 && C <= 16 && fc >= 1 && fc <= 128 && c_cond >= 0 && c_cond <= 4 && H % PTH == 0 && W % PTW == 0 &&
           (long)B * (H / PTH) * (W / PTW) <= 4096 && (size_t)C * H * W < ((size_t)1 << 31);
}
int nca_dynca_persist_tiles(int B, int H, int W) { return B * (H / PTH) * (W / PTW); }

hipError_t nca_launch_dynca_persist(const NcaDyncaPersistArgs& a_in, hipStream_t st, bool query_only, bool* fits) {
    NcaDyncaPersistArgs a = a_in;
    a.err = nca_error_word_device();
    const bool small = a.C <= 12 && a.fc <= 96;
    if (a.c_cond > 0) return small ? launch_persist<12, 96, true>(a, st, query_only, fits) : launch_persist<16, 128, true>(a, st, query_only, fits);
    return small ? launch_persist<12, 96, false>(a, st, query_only, fits) : launch_persist<16, 128, false>(a, st, query_only, fits);
}
