// nca_dynca_persist.hip -- T DyNCA steps in ONE launch for small grids (B = 1 video inference: ConditioneDyNCA/utils/misc/
// video_utils.py:50-82 runs forward_nsteps(h, step_n, cond_img=frame) per frame; WebGL twin docs/dynca.js:1057-1132), gfx950, fp32.
//
// At 1 x 256 x 256 one step is 256 cells per CU and one wave per SIMD: 408 exact-f32 MFMAs per wave = 5.4 us, and the per-step
// launch adds ~13 us of launch gap, weight-image fill, cold first touch and tail to it (18.5 us per step, C = 12 / fc = 96).
// Here a workgroup OWNS one 16 x 16 tile for all T steps:
//   * weight images and the conditioning tile are built once; the tile's state stays in LDS (two buffers, ping-pong) and only the
//     one-cell halo (68 cells x C) is re-read per step;
//   * neighbours synchronise through per-tile monotonic step counters in global memory (tools/micro/neighbour_sync.hip measured
//     the protocol): step t may read its halo once the <= 8 neighbours have published step t, i.e. their state t is in memory --
//     and, because a workgroup reads its halo BEFORE it publishes, that also means they are done READING my state t-1, so the
//     ping-pong buffer holding it may be overwritten with state t+1;
//   * the exchange runs UNDER the compute: a fifth wave (the sync wave) polls the counters, loads the halo (coherent loads: the
//     tiles sit in other XCDs' L2s) and writes it into the LDS tile while the four compute waves work on the 12 groups of 16
//     INTERIOR cells (rows / columns 1..14: no halo needed); the tile's own border ring (60 cells + 4 left-over interior cells =
//     4 groups) comes last, and only the ring is stored to memory every step (write-through) -- the interior stays in LDS until the
//     final step.  16 groups exactly: no extra MFMA work for the split;
//   * the counter is published by the sync wave after every compute wave's stores are acknowledged (s_waitcnt vmcnt(0) + workgroup
//     barrier); the next step's interior groups start right behind that barrier, not behind the neighbours;
//   * every poll is BOUNDED: an expired poll (a neighbour that never became resident) sets bit 1 of the sticky device error word and
//     an abort word in device memory; every workgroup leaves its step loop at its next step -- the launch always drains, the host
//     sees NCAHIP_EDEVICE.
// The host side launches this only when every workgroup can be co-resident (occupancy x CUs >= tiles), H % 16 == 0, W % 16 == 0,
// C <= 16, fc <= 128; anything else runs the per-step kernels (nca_step_fwd.hip).  Same arithmetic per cell in the same order as the
// per-step kernel (perception from the LDS tile, exact-f32 MFMA chains with the same k order): bit-identical results.
#include "nca_common.h"
#include <cstdlib>
#include <type_traits>

#include "nca_kernels.h"

static int g_drop_tiles = 0;
void nca_set_persist_drop_tiles(int n) { g_drop_tiles = n; }

namespace {

constexpr int kPT = 320;                     // 4 compute waves + 1 sync wave
constexpr int PTH = 16, PTW = 16;            // tile
constexpr int PROWS = PTH + 2, PRS = 24, PCS = PROWS * PRS;   // Z[ch][row 0..17][col: image col tx0-1+q at index q+3]
static_assert(PCS % 32 == 16 && PTW + 2 + 3 <= PRS, "tile carve");
constexpr int kHalo = 2 * (PTW + 2) + 2 * PTH;             // 68 halo cells
constexpr int kIW = PTW - 2;                               // interior width (14)

template <int CP, int FC, bool HAS_COND>
struct PersistCfg {
    static constexpr int K1S = CP + (HAS_COND ? 1 : 0);
    static constexpr int M1T = FC / 16, K2S = FC / 4;
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + M1T * K1S * 64;
    static constexpr int OFF_B1 = OFF_W2 + K2S * 64;
    static constexpr int OFF_B2 = OFF_B1 + FC;
    static constexpr int OFF_Z = OFF_B2 + 16;
    static constexpr int OFF_MK = OFF_Z + 2 * CP * PCS;               // fire masks of two steps, [2][256]
    static constexpr int OFF_CN = OFF_MK + 2 * PTH * PTW;
    static constexpr int OFF_FLAG = OFF_CN + (HAS_COND ? 4 * PTH * PTW : 0);   // [0] halo of step t staged (t + 1), [1] abort
    static constexpr int LDS_FLOATS = OFF_FLAG + 4;
    static constexpr int NLD = (kHalo * CP + 63) / 64;       // halo loads per sync-wave lane and step
    static_assert(CP % 4 == 0 && CP <= 16 && FC % 16 == 0 && OFF_Z % 4 == 0, "shape");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ float ld_coherent(const float* p) {
    return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_through(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ring exchange: a cell value travels WITH the step it belongs to, as one 64-bit word (single-copy atomic): the reader needs no
// separate counter -- and the writer no acknowledgement round trip -- to know that what it loaded is state `step`
__device__ __forceinline__ void st_pair(unsigned long long* p, float v, int step) {
    __hip_atomic_store(p, ((unsigned long long)(unsigned)step << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_pair(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kRingCells = 2 * PTW + 2 * (PTH - 2);        // 60 cells of a tile that neighbours read
// position of tile cell (r, q) in the ring enumeration (group_cell): top row, bottom row, left column, right column
__device__ __forceinline__ int ring_index(int r, int q) {
    if (r == 0) return q;
    if (r == PTH - 1) return PTW + q;
    return q == 0 ? r + 31 : r + 45;
}
// cell (r, q) of the tile that lane ci of group j works on: groups 0..11 = interior cells 16 j + ci of the 14 x 14 interior
// (row-major), groups 12..15 = 15 ring cells each + one of the four left-over interior cells
__device__ __forceinline__ void group_cell(int j, int ci, int& r, int& q) {
    int idx = 16 * j + ci;                       // interior enumeration
    if (j >= 12) {
        const int jb = j - 12;
        if (ci < 15) {
            const int k = 15 * jb + ci;          // ring enumeration: top row, bottom row, left column, right column
            if (k < 16) { r = 0; q = k; }
            else if (k < 32) { r = PTH - 1; q = k - 16; }
            else if (k < 46) { r = k - 31; q = 0; }
            else { r = k - 45; q = PTW - 1; }
            return;
        }
        idx = 192 + jb;
    }
    r = 1 + idx / kIW;
    q = 1 + idx % kIW;
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
    int* const lflag = reinterpret_cast<int*>(smem + K::OFF_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, ci = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W, fc = a.fc, CC = a.c_cond, K1 = 4 * C + CC;
    const size_t plane = (size_t)H * W;
    const unsigned tag0 = a.epoch << 12;     // (value, tag) pairs: tag = epoch * 4096 + step: stale pairs of earlier launches never match
    const int tiles_x = W / PTW, tiles_y = H / PTH;
    const int tile = blockIdx.x, txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int ty0 = tyi * PTH, tx0 = txi * PTW;
    const size_t cell0 = (size_t)b * plane + (size_t)ty0 * W + tx0;     // linear index of the tile's first cell

    // ---- once per launch: A-operand weight images (same layouts and k order as dynca_step_fwd_kernel).  Two-phase gather: every
    //      load of both images is requested before the first LDS write (one cold round trip for the prologue, not one per element)
    {
        constexpr int N1 = K::M1T * K::K1S * 64, N2 = K::K2S * 64, U1 = (N1 + kPT - 1) / kPT, U2 = (N2 + kPT - 1) / kPT;
        float v1[U1], v2[U2];
#pragma unroll
        for (int u = 0; u < U1; ++u) {
            const int idx = tid + kPT * u;
            const int l = idx & 63, s_ = (idx >> 6) % K::K1S, m = (idx >> 6) / K::K1S;
            const int gg = l >> 4, o = 16 * m + (l & 15);
            long src = -1;
            if (idx < N1 && o < fc) {
                if (s_ < CP) {   // k-step s = 4c'+f: channel 4c'+g, filter f (0 id, 1 sobel_x, 2 sobel_y, 3 lap); blocked [x|Sx|Sy|L], dynca.py:92-95
                    const int ch = (s_ & ~3) + gg;
                    if (ch < C) src = (long)o * K1 + (s_ & 3) * C + ch;
                } else if (gg < CC) src = (long)o * K1 + 4 * C + gg;
            }
            const float w = a.w1[src >= 0 ? src : 0];
            v1[u] = src >= 0 ? w : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int idx = tid + kPT * u;
            const int l = idx & 63, s_ = idx >> 6;
            const int gg = l >> 4, o = l & 15, k = 16 * (s_ >> 2) + 4 * gg + (s_ & 3);
            const bool ok = idx < N2 && o < C && k < fc;
            const float w = a.w2[ok ? (long)o * fc + k : 0];
            v2[u] = ok ? w : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U1; ++u)
            if (tid + kPT * u < N1) W1L[tid + kPT * u] = v1[u];
#pragma unroll
        for (int u = 0; u < U2; ++u)
            if (tid + kPT * u < N2) W2L[tid + kPT * u] = v2[u];
    }
    for (int idx = tid; idx < FC; idx += kPT) B1L[idx] = idx < fc ? a.b1[idx] : 0.0f;
    if (tid < 16) B2L[tid] = tid < C ? a.b2[tid] : 0.0f;
    if (tid < 4) lflag[tid] = 0;
    if (HAS_COND) {
        for (int i = tid; i < 4 * PTH * PTW; i += kPT) {
            const int cc = i / (PTH * PTW), c = i % (PTH * PTW);
            CN[i] = cc < CC ? a.cond[((size_t)b * CC + cc) * plane + (size_t)(ty0 + c / PTW) * W + tx0 + c % PTW] : 0.0f;
        }
    }
    // fire mask of step t for the tile's 256 cells into MK[t & 1] (dynca.py:131): explicit uniforms, bit-packed masks, or Philox
    auto fill_mask = [&](int t, int first, int stride) {
        float* const mk = MK + (t & 1) * (PTH * PTW);
        const size_t cells = (size_t)a.B * plane;
        for (int c = first; c < PTH * PTW; c += stride) {
            const size_t cell = cell0 + (size_t)(c / PTW) * W + c % PTW;
            float uu;
            if (a.u) {
                if (a.u_bits) uu = ((reinterpret_cast<const uint32_t*>(a.u)[(size_t)t * ((cells + 31) / 32) + (cell >> 5)] >> (unsigned)(cell & 31)) & 1u) ? 1.0f : 0.0f;
                else uu = a.u[(size_t)t * cells + cell];
            } else uu = nca_philox_cell(a.seed, a.step0 + (uint64_t)t, cell);
            mk[c] = floorf(uu + a.rate);
        }
    };
    fill_mask(0, tid, kPT);
    // the tile's cells at step 0 (plain loads: written before the launch)
    {
        float* const Z0 = smem + K::OFF_Z;
        const float* const xb = a.x_in + (size_t)b * C * plane;
        for (int i = tid; i < CP * PTH * (PTW / 4); i += kPT) {
            const int f4 = i % (PTW / 4), r = (i / (PTW / 4)) % PTH, ch = i / (PTH * (PTW / 4));
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < C) v = *reinterpret_cast<const f32x4*>(xb + (size_t)ch * plane + (size_t)(ty0 + r) * W + tx0 + 4 * f4);
            *reinterpret_cast<f32x4*>(Z0 + ch * PCS + (r + 1) * PRS + 4 + 4 * f4) = v;
        }
    }
    __syncthreads();
    int* const abort_w = a.flags;   // device-memory abort word (the sticky error word itself is host-mapped: polling THAT is a PCIe read storm)

    if (wave == 4) {
        // =================================== sync wave: counters, halo, masks ================================================
        // halo items of this lane: item j = lane + 64 k  <->  halo cell j % 68 of channel j / 68; source offset once (pad resolved)
        int rz[K::NLD];
        unsigned rsrc[K::NLD], rxch[K::NLD];
#pragma unroll
        for (int k = 0; k < K::NLD; ++k) {
            const int j = lane + 64 * k, hc = j % kHalo, ch = j / kHalo;
            int r, q;
            if (hc < PTW + 2) { r = 0; q = hc; }
            else if (hc < 2 * (PTW + 2)) { r = PROWS - 1; q = hc - (PTW + 2); }
            else if (hc < 2 * (PTW + 2) + PTH) { r = hc - 2 * (PTW + 2) + 1; q = 0; }
            else { r = hc - 2 * (PTW + 2) - PTH + 1; q = PTW + 1; }
            const bool live = j < kHalo * CP;
            rz[k] = live ? ch * PCS + r * PRS + q + 3 : -1;
            const int sy = nca_pad_index(ty0 - 1 + r, H, a.pad_mode), sx = nca_pad_index(tx0 - 1 + q, W, a.pad_mode);
            rsrc[k] = (live && ch < C && sy >= 0 && sx >= 0) ? (unsigned)((size_t)ch * plane + (size_t)sy * W + sx) : ~0u;
            // a pad-resolved source inside this tile (reflect / replicate at the image border: row 1, row 0 ...) is read from the LDS
            // tile itself -- interior cells are not in memory between the first and the last step (high bit = LDS offset)
            if (rsrc[k] != ~0u && sy >= ty0 && sy < ty0 + PTH && sx >= tx0 && sx < tx0 + PTW)
                rsrc[k] = 0x80000000u | (unsigned)(ch * PCS + (sy - ty0 + 1) * PRS + (sx - tx0) + 4);
            // the same source in the ring-exchange buffer (steps >= 1): [tile][channel][ring cell] pairs
            rxch[k] = ~0u;
            if (rsrc[k] != ~0u && !(rsrc[k] & 0x80000000u)) {
                const int st_ = (b * tiles_y + sy / PTH) * tiles_x + sx / PTW;
                rxch[k] = (unsigned)((st_ * C + ch) * kRingCells + ring_index(sy % PTH, sx % PTW));
            }
        }
        // the halo of state tt into LDS buffer tt & 1: `global` = the items that come from memory (neighbours' ring cells, coherent
        // loads), else the items whose pad-resolved source is a cell of this tile (read from the LDS tile: needs it complete)
        // the halo of state tt into LDS buffer tt & 1.  global = the items that come from the neighbours: at tt = 0 plain loads of the
        // input state; later the ring-exchange pairs, re-read until every one carries step tt (bounded; false = gave up).
        // !global = the items whose pad-resolved source is a cell of this tile (read from the LDS tile: needs it complete).
        auto stage_halo = [&](int tt, bool global) -> bool {
            float* const Zt = smem + K::OFF_Z + (tt & 1) * (CP * PCS);
            float hv[K::NLD];
            if (!global) {
#pragma unroll
                for (int k = 0; k < K::NLD; ++k) {
                    const bool own = rsrc[k] != ~0u && (rsrc[k] & 0x80000000u);
                    if (own) hv[k] = Zt[rsrc[k] & 0x7fffffffu];
                }
#pragma unroll
                for (int k = 0; k < K::NLD; ++k) {
                    const bool own = rsrc[k] != ~0u && (rsrc[k] & 0x80000000u);
                    if (rz[k] >= 0 && own) Zt[rz[k]] = hv[k];
                }
                return true;
            }
            if (tt == 0) {
                const float* const src = a.x_in + (size_t)b * C * plane;
#pragma unroll
                for (int k = 0; k < K::NLD; ++k) hv[k] = rxch[k] != ~0u ? src[rsrc[k]] : 0.0f;
            } else {
                const unsigned long long* const xs = a.xch + (size_t)(tt & 1) * a.xch_words;
                for (int spins = 0;;) {
                    bool stale = false;
#pragma unroll
                    for (int k = 0; k < K::NLD; ++k) {
                        unsigned long long w = (unsigned long long)(tag0 + (unsigned)tt) << 32;
                        if (rxch[k] != ~0u) w = ld_pair(xs + rxch[k]);
                        hv[k] = __uint_as_float((unsigned)w);
                        stale = stale || (unsigned)(w >> 32) != tag0 + (unsigned)tt;
                    }
                    if (!__any(stale)) break;
                    bool give_up = ++spins >= (1 << 15);      // (uniform: every lane counts every round)
                    if ((spins & 31) == 0) give_up = give_up || __any(lane == 8 && (unsigned)__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch);
                    if (give_up) return false;
                    __builtin_amdgcn_s_sleep(1);
                }
            }
#pragma unroll
            for (int k = 0; k < K::NLD; ++k) {
                const bool own = rsrc[k] != ~0u && (rsrc[k] & 0x80000000u);
                if (rz[k] >= 0 && !own) Zt[rz[k]] = hv[k];
            }
            return true;
        };
        auto post_halo = [&](int value, bool stop_) {   // LDS operations of a wave execute in order: data, then the counter
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) {
                if (stop_) __hip_atomic_store(lflag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(lflag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        // state 0: everything is in memory / the LDS tile already
        stage_halo(0, true);
        stage_halo(0, false);
        post_halo(1, false);
        bool stop = false;
        for (int t = 0; t < a.T; ++t) {
            // the compute waves are in step t: ring first (its pairs go out early), then the interior.  Meanwhile: the neighbours'
            // rings of state t + 1 as soon as they arrive, into the OTHER LDS buffer's halo cells (nobody touches those in step t)
            if (t + 1 < a.T) {
                if (!(a.dbg & 1)) {
                    stop = !stage_halo(t + 1, true);
                    if (stop) {   // a neighbour never delivered (not resident?): record it, tell every workgroup to drain
                        if (lane == 0) {
                            if (a.err) __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(abort_w, (int)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            __syncthreads();      // step t's LDS tile (state t + 1) is complete
            if (t + 1 < a.T) {
                if (!stop && !(a.dbg & 1)) stage_halo(t + 1, false);
                post_halo(t + 2, stop);
            }
            if (stop) break;
        }
    } else {
        // =================================== compute waves ==================================================================
        for (int t = 0; t < a.T; ++t) {
            float* const dst = a.x_out + (size_t)b * C * plane;      // (written at the final step only)
            const float* const Zc = smem + K::OFF_Z + (t & 1) * (CP * PCS);
            float* const Zn = smem + K::OFF_Z + ((t + 1) & 1) * (CP * PCS);
            const float* const mkc = MK + (t & 1) * (PTH * PTW);
            const bool last = t + 1 == a.T;
            // one phase = NTP groups of 16 cells: perception -> MLP on MFMA -> residual
            auto phase = [&](auto ntp_tag, int j0, bool ack) {
                constexpr int NTP = decltype(ntp_tag)::value;
                int rr[NTP], qq[NTP];
#pragma unroll
                for (int n = 0; n < NTP; ++n) group_cell(j0 + n, ci, rr[n], qq[n]);
                float P[NTP][K::K1S];
#pragma unroll
                for (int cq4 = 0; cq4 < CP / 4; ++cq4) {
                    const float* const zc = Zc + (4 * cq4 + g) * PCS + 3;
#pragma unroll
                    for (int n = 0; n < NTP; ++n) {
                        float nbv[3][3];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) nbv[dy][dx] = zc[(rr[n] + dy) * PRS + qq[n] + dx];
                        P[n][4 * cq4 + 0] = nbv[1][1];
                        P[n][4 * cq4 + 1] = nca_sobel_x(nbv);
                        P[n][4 * cq4 + 2] = nca_sobel_y(nbv);
                        P[n][4 * cq4 + 3] = nca_laplacian(nbv);
                    }
                }
                if (HAS_COND) {
#pragma unroll
                    for (int n = 0; n < NTP; ++n) P[n][CP] = CN[g * PTH * PTW + rr[n] * PTW + qq[n]];
                }
                f32x4 acc2[NTP];
                {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(B2L + 4 * g);
#pragma unroll
                    for (int n = 0; n < NTP; ++n) acc2[n] = bias;
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
                for (int m = 0; m < ((a.dbg & 16) ? 0 : K::M1T); ++m) {
                    f32x4 acc1[NTP];
#pragma unroll
                    for (int n = 0; n < NTP; ++n) acc1[n] = bias1;
#pragma unroll
                    for (int s = 0; s < K::K1S; ++s)
#pragma unroll
                        for (int n = 0; n < NTP; ++n) acc1[n] = nca_mfma(wa1[s], P[n][s], acc1[n]);
                    float w2c[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) w2c[r] = wa2[r];
                    __builtin_amdgcn_sched_barrier(0);
                    if (m + 1 < K::M1T) fetch(m + 1);        // in flight across this tile's layer-2 MFMAs and the next chain
                    float h[NTP][4];
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[n][r] = __int_as_float(max(__float_as_int(acc1[n][r]), 0));   // relu
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int n = 0; n < NTP; ++n) acc2[n] = nca_mfma(w2c[r], h[n][r], acc2[n]);
                }
                // residual + stochastic mask (dynca.py:131-133): state t+1 -> the next step's LDS tile; ring cells (what the
                // neighbours read) -> memory every step, write-through; everything -> memory at the final step
#pragma unroll
                for (int n = 0; n < NTP; ++n) {
                    const float mk = mkc[rr[n] * PTW + qq[n]];
                    const bool ring = rr[n] == 0 || rr[n] == PTH - 1 || qq[n] == 0 || qq[n] == PTW - 1;
                    const size_t o0 = (size_t)(ty0 + rr[n]) * W + tx0 + qq[n];
                    unsigned long long* const xd = a.xch + (size_t)((t + 1) & 1) * a.xch_words + (size_t)tile * C * kRingCells +
                                                   (ring ? ring_index(rr[n], qq[n]) : 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = 4 * g + r;
                        if (ch < CP) {
                            const int zo = ch * PCS + (rr[n] + 1) * PRS + qq[n] + 4;
                            const float xn = Zc[zo] + acc2[n][r] * mk;
                            Zn[zo] = ch < C ? xn : 0.0f;
                            if (ch < C && !(a.dbg & 2)) {
                                if (last) dst[(size_t)ch * plane + o0] = xn;
                                else if (ring) st_pair(xd + ch * kRingCells, xn, (int)(tag0 + (unsigned)(t + 1)));
                            }
                        }
                    }
                }
            };
            bool stop;
            {   // the halo of state t is in the LDS tile (bounded poll of the sync wave's LDS counter)
                int spins = 0;
                while (__hip_atomic_load(lflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                stop = __hip_atomic_load(lflag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            }
            if (stop) break;
            // the tile's border ring first (+ 4 left-over interior cells): its stores are what the neighbours wait for
            if (!(a.dbg & 4)) phase(std::integral_constant<int, 1>{}, 12 + wave, false);
            // the next step's fire masks: one cell per compute thread (in the sync wave -- four Philox evaluations per lane -- they
            // cost the SIMD it shares with compute wave 0 a microsecond per step)
            if (!last && !(a.dbg & 8)) fill_mask(t + 1, tid, 256);
            // the interior needs no halo and hides the exchange: ring pairs stored -> fetched by the neighbours
            phase(std::integral_constant<int, 3>{}, 3 * wave, true);
            __syncthreads();                    // Zn complete, Zc free
        }
        __builtin_amdgcn_s_waitcnt(0);   // the final step's stores (the whole tile)
    }
}

// =====================================================================================================================
// Two-scale variant (perception_scales = [0, 1]: every shipped video model; dynca.py:75-115, WebGL twin docs/dynca.js:288-355).
// Per step a tile additionally needs the COARSE level around it: xc = the 2 x 2 means of the state on coarse cells -2 .. 9 of its
// 8 x 8 (pad mode resolved on the coarse grid, as dynca_coarse_perceive_kernel does), from which it computes the coarse perception
// pc on cells -1 .. 8 (index-clamped at the image border: what bilinear x2 up-sampling with align_corners = False reads) and blends
// (P_fine + up2(pc)) / 2 exactly as the per-step kernel (same expressions, same order: bit-identical).  Neighbours therefore exchange
// two things per channel: the 60 fine ring cells and the 2 x 2 means of their 48 coarse cells within 2 of the tile border -- i.e. the
// state of every fine cell within 4 of the border.  So the order of a step is: the 192-cell border BAND first (12 groups), barrier,
// the sync wave publishes the coarse means while the compute waves do the 8 x 8 centre (4 groups), and fetches the neighbours' data.
constexpr int XCR = 12, XCS = XCR * XCR;           // coarse x tile: coarse rows / cols -2 .. 9, per channel
constexpr int PCR2 = 10, PCS2 = 104;               // coarse perception tile: 10 x 10 per plane (row pitch 10, plane pitch 104)
constexpr int kCoarseRing = 48;                    // coarse cells of a tile within 2 of its border (what neighbours read)
constexpr int kCoarseHalo = XCS - 64;              // 80 coarse cells around the tile's own 8 x 8
constexpr int kXchMS = kRingCells + kCoarseRing;   // 108 exchanged pairs per channel and tile

__device__ __forceinline__ int coarse_ring_index(int i, int j) {   // (i, j) in 0..7, within 2 of the border
    if (i < 2) return i * 8 + j;
    if (i > 5) return 16 + (i - 6) * 8 + j;
    return j < 2 ? 32 + (i - 2) * 2 + j : 40 + (i - 2) * 2 + (j - 6);
}
// groups 0..11 = the 192 band cells (rows 0-3, rows 12-15, then columns 0-3 and 12-15 of rows 4-11), groups 12..15 = the 8 x 8 centre
__device__ __forceinline__ void group_cell_ms(int j, int ci, int& r, int& q) {
    int idx = 16 * j + ci;
    if (j >= 12) { idx -= 192; r = 4 + idx / 8; q = 4 + idx % 8; }
    else if (idx < 64) { r = idx / 16; q = idx % 16; }
    else if (idx < 128) { r = 12 + (idx - 64) / 16; q = idx % 16; }
    else if (idx < 160) { r = 4 + (idx - 128) / 4; q = (idx - 128) % 4; }
    else { r = 4 + (idx - 160) / 4; q = 12 + (idx - 160) % 4; }
}

template <int CP, int FC, bool HAS_COND>
struct PersistMsCfg {
    using B = PersistCfg<CP, FC, HAS_COND>;
    static constexpr int OFF_XC = B::LDS_FLOATS;                 // [CP][12 x 12]
    static constexpr int OFF_PCL = OFF_XC + CP * XCS;            // [4 CP][10 x 10]
    static constexpr int NLC = (kCoarseHalo * CP + 63) / 64;     // coarse halo items per sync-wave lane
    static constexpr int OFF_TBL = OFF_PCL + 4 * CP * PCS2;      // the sync wave's item table: [2 words][fine + coarse items][64 lanes]
    static constexpr int LDS_FLOATS = OFF_TBL + 2 * (B::NLD + NLC) * 64;
    static constexpr int NPB = (kCoarseRing * CP + 63) / 64;     // own coarse ring cells to publish per sync-wave lane
    static constexpr int NPC = (PCR2 * PCR2 * CP + 255) / 256;   // coarse perception cells per compute thread
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

template <int CP, int FC, bool HAS_COND>
__global__ __launch_bounds__(kPT, 1) void dynca_persist_ms_kernel(const NcaDyncaPersistArgs a) {
    using K = PersistCfg<CP, FC, HAS_COND>;
    using KM = PersistMsCfg<CP, FC, HAS_COND>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const W1L = smem + K::OFF_W1;
    float* const W2L = smem + K::OFF_W2;
    float* const B1L = smem + K::OFF_B1;
    float* const B2L = smem + K::OFF_B2;
    float* const MK = smem + K::OFF_MK;
    float* const CN = smem + K::OFF_CN;
    float* const XC = smem + KM::OFF_XC;
    float* const PCL = smem + KM::OFF_PCL;
    int* const lflag = reinterpret_cast<int*>(smem + K::OFF_FLAG);

    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, ci = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W, fc = a.fc, CC = a.c_cond, K1 = 4 * C + CC;
    const int Hc = H >> 1, Wc = W >> 1;
    const size_t plane = (size_t)H * W;
    const unsigned tag0 = a.epoch << 12;
    const int tiles_x = W / PTW, tiles_y = H / PTH;
    const int tile = blockIdx.x, txi = tile % tiles_x, tyi = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
    const int ty0 = tyi * PTH, tx0 = txi * PTW, cy0 = ty0 >> 1, cx0 = tx0 >> 1;
    const size_t cell0 = (size_t)b * plane + (size_t)ty0 * W + tx0;

    // ---- once per launch: weight images (as the single-scale kernel), conditioning, masks of step 0, the tile's cells -----------
    {
        constexpr int N1 = K::M1T * K::K1S * 64, N2 = K::K2S * 64, U1 = (N1 + kPT - 1) / kPT, U2 = (N2 + kPT - 1) / kPT;
        float v1[U1], v2[U2];
#pragma unroll
        for (int u = 0; u < U1; ++u) {
            const int idx = tid + kPT * u;
            const int l = idx & 63, s_ = (idx >> 6) % K::K1S, m = (idx >> 6) / K::K1S;
            const int gg = l >> 4, o = 16 * m + (l & 15);
            long src = -1;
            if (idx < N1 && o < fc) {
                if (s_ < CP) {
                    const int ch = (s_ & ~3) + gg;
                    if (ch < C) src = (long)o * K1 + (s_ & 3) * C + ch;
                } else if (gg < CC) src = (long)o * K1 + 4 * C + gg;
            }
            const float w = a.w1[src >= 0 ? src : 0];
            v1[u] = src >= 0 ? w : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int idx = tid + kPT * u;
            const int l = idx & 63, s_ = idx >> 6;
            const int gg = l >> 4, o = l & 15, k = 16 * (s_ >> 2) + 4 * gg + (s_ & 3);
            const bool ok = idx < N2 && o < C && k < fc;
            const float w = a.w2[ok ? (long)o * fc + k : 0];
            v2[u] = ok ? w : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U1; ++u)
            if (tid + kPT * u < N1) W1L[tid + kPT * u] = v1[u];
#pragma unroll
        for (int u = 0; u < U2; ++u)
            if (tid + kPT * u < N2) W2L[tid + kPT * u] = v2[u];
    }
    for (int idx = tid; idx < FC; idx += kPT) B1L[idx] = idx < fc ? a.b1[idx] : 0.0f;
    if (tid < 16) B2L[tid] = tid < C ? a.b2[tid] : 0.0f;
    if (tid < 4) lflag[tid] = 0;
    if (HAS_COND) {
        for (int i = tid; i < 4 * PTH * PTW; i += kPT) {
            const int cc = i / (PTH * PTW), c = i % (PTH * PTW);
            CN[i] = cc < CC ? a.cond[((size_t)b * CC + cc) * plane + (size_t)(ty0 + c / PTW) * W + tx0 + c % PTW] : 0.0f;
        }
    }
    auto fill_mask = [&](int t, int first, int stride) {
        float* const mk = MK + (t & 1) * (PTH * PTW);
        const size_t cells = (size_t)a.B * plane;
        for (int c = first; c < PTH * PTW; c += stride) {
            const size_t cell = cell0 + (size_t)(c / PTW) * W + c % PTW;
            float uu;
            if (a.u) {
                if (a.u_bits) uu = ((reinterpret_cast<const uint32_t*>(a.u)[(size_t)t * ((cells + 31) / 32) + (cell >> 5)] >> (unsigned)(cell & 31)) & 1u) ? 1.0f : 0.0f;
                else uu = a.u[(size_t)t * cells + cell];
            } else uu = nca_philox_cell(a.seed, a.step0 + (uint64_t)t, cell);
            mk[c] = floorf(uu + a.rate);
        }
    };
    fill_mask(0, tid, kPT);
    {
        float* const Z0 = smem + K::OFF_Z;
        const float* const xb = a.x_in + (size_t)b * C * plane;
        for (int i = tid; i < CP * PTH * (PTW / 4); i += kPT) {
            const int f4 = i % (PTW / 4), r = (i / (PTW / 4)) % PTH, ch = i / (PTH * (PTW / 4));
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ch < C) v = *reinterpret_cast<const f32x4*>(xb + (size_t)ch * plane + (size_t)(ty0 + r) * W + tx0 + 4 * f4);
            *reinterpret_cast<f32x4*>(Z0 + ch * PCS + (r + 1) * PRS + 4 + 4 * f4) = v;
        }
    }
    __syncthreads();
    int* const abort_w = a.flags;
    // 2 x 2 mean of tile-local coarse cell (i, j) of channel ch from an LDS state tile -- upsample_bilinear2d's own expression for the
    // exact x1/2 case, as dynca_coarse_perceive_kernel evaluates it
    auto mean4 = [&](const float* Zt, int ch, int i, int j) -> float {
        const float* const p0 = Zt + ch * PCS + (2 * i + 1) * PRS + 2 * j + 4;
        return 0.5f * (0.5f * p0[0] + 0.5f * p0[1]) + 0.5f * (0.5f * p0[PRS] + 0.5f * p0[PRS + 1]);
    };

    if (wave == 4) {
        // =================================== sync wave ============================================================================
        // Halo items of this lane -- NIT = fine (68 x CP) + coarse (80 x CP) cells over 64 lanes -- live as a two-word table in LDS
        // (six register arrays of 37 entries would not fit beside the compute waves' registers: all five waves share one allocation):
        //   word 0: destination offset (fine: in a Z buffer, coarse: in XC) | kind << 16 | coarse << 18
        //   word 1: kind 3 (from a neighbour): index of the pair in the exchange buffer; kind 2 (this tile): source LDS offset
        // kind 0 = no item, 1 = zero (constant padding), 2 = pad-resolved source inside this tile, 3 = a neighbour's cell.
        // The t = 0 values come straight from x_in while the table is built.
        constexpr int NIT = K::NLD + KM::NLC;
        int* const TBL = reinterpret_cast<int*>(smem + KM::OFF_TBL);
        {
            float* const Z0 = smem + K::OFF_Z;
            const float* const src = a.x_in + (size_t)b * C * plane;
#pragma unroll 1
            for (int k = 0; k < NIT; ++k) {
                const bool coarse = k >= K::NLD;
                const int j = lane + 64 * (coarse ? k - K::NLD : k);
                int w0 = 0, w1 = 0;
                if (!coarse) {
                    const int hc = j % kHalo, ch = j / kHalo;
                    int r, q;
                    if (hc < PTW + 2) { r = 0; q = hc; }
                    else if (hc < 2 * (PTW + 2)) { r = PROWS - 1; q = hc - (PTW + 2); }
                    else if (hc < 2 * (PTW + 2) + PTH) { r = hc - 2 * (PTW + 2) + 1; q = 0; }
                    else { r = hc - 2 * (PTW + 2) - PTH + 1; q = PTW + 1; }
                    if (j < kHalo * CP) {
                        const int dsto = ch * PCS + r * PRS + q + 3;
                        const int sy = nca_pad_index(ty0 - 1 + r, H, a.pad_mode), sx = nca_pad_index(tx0 - 1 + q, W, a.pad_mode);
                        int kind = 1;
                        if (ch < C && sy >= 0 && sx >= 0) {
                            if (sy >= ty0 && sy < ty0 + PTH && sx >= tx0 && sx < tx0 + PTW) {
                                kind = 2;
                                w1 = ch * PCS + (sy - ty0 + 1) * PRS + (sx - tx0) + 4;
                            } else {
                                kind = 3;
                                w1 = (((b * tiles_y + sy / PTH) * tiles_x + sx / PTW) * C + ch) * kXchMS + ring_index(sy % PTH, sx % PTW);
                                Z0[dsto] = src[(size_t)ch * plane + (size_t)sy * W + sx];
                            }
                        }
                        if (kind == 1) { Z0[dsto] = 0.0f; Z0[CP * PCS + dsto] = 0.0f; }      // constant padding / padded channels: zero once, both buffers
                        w0 = dsto | (kind << 16);
                    }
                } else {
                    const int hc = j % kCoarseHalo, ch = j / kCoarseHalo;
                    int ri, rj;     // XC coordinates 0..11
                    if (hc < 2 * XCR) { ri = hc / XCR; rj = hc % XCR; }
                    else if (hc < 4 * XCR) { ri = 10 + (hc - 2 * XCR) / XCR; rj = hc % XCR; }
                    else { const int h2 = hc - 4 * XCR; ri = 2 + h2 / 4; rj = (h2 % 4) < 2 ? (h2 % 4) : 8 + (h2 % 4); }
                    if (j < kCoarseHalo * CP) {
                        const int dsto = ch * XCS + ri * XCR + rj;
                        const int sy = nca_pad_index(cy0 - 2 + ri, Hc, a.pad_mode), sx = nca_pad_index(cx0 - 2 + rj, Wc, a.pad_mode);
                        int kind = 1;
                        if (ch < C && sy >= 0 && sx >= 0) {
                            if (sy >= cy0 && sy < cy0 + 8 && sx >= cx0 && sx < cx0 + 8) {
                                kind = 2;
                                w1 = ch * XCS + (sy - cy0 + 2) * XCR + (sx - cx0) + 2;
                            } else {
                                kind = 3;
                                w1 = (((b * tiles_y + sy / 8) * tiles_x + sx / 8) * C + ch) * kXchMS + kRingCells + coarse_ring_index(sy % 8, sx % 8);
                                const float* const p0 = src + (size_t)ch * plane + (size_t)(2 * sy) * W + 2 * sx;
                                XC[dsto] = 0.5f * (0.5f * p0[0] + 0.5f * p0[1]) + 0.5f * (0.5f * p0[W] + 0.5f * p0[W + 1]);
                            }
                        }
                        if (kind == 1) XC[dsto] = 0.0f;
                        w0 = dsto | (kind << 16) | (1 << 18);
                    }
                }
                TBL[(2 * k) * 64 + lane] = w0;
                TBL[(2 * k + 1) * 64 + lane] = w1;
            }
        }
        // the part of the halos of state tt that comes from THIS tile (needs the LDS tile of state tt complete): own coarse cells, then
        // the pad-resolved items
        auto stage_own = [&](int tt) {
            float* const Zt = smem + K::OFF_Z + (tt & 1) * (CP * PCS);
            for (int ch = 0; ch < CP; ++ch) XC[ch * XCS + ((lane >> 3) + 2) * XCR + (lane & 7) + 2] = ch < C ? mean4(Zt, ch, lane >> 3, lane & 7) : 0.0f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
            for (int k = 0; k < NIT; ++k) {
                const int w0 = TBL[(2 * k) * 64 + lane], w1 = TBL[(2 * k + 1) * 64 + lane];
                if (((w0 >> 16) & 3) == 2) {
                    if (w0 & (1 << 18)) XC[w0 & 0xffff] = XC[w1];
                    else Zt[w0 & 0xffff] = Zt[w1];
                }
            }
        };
        // the neighbours' part of the halos of state tt >= 1: tagged pairs, re-read until every one carries the tag (false = gave up)
        auto stage_global = [&](int tt) -> bool {
            float* const Zt = smem + K::OFF_Z + (tt & 1) * (CP * PCS);
            const unsigned long long* const xs = a.xch + (size_t)(tt & 1) * a.xch_words;
            const unsigned want = tag0 + (unsigned)tt;
            float hv[NIT];
            for (int spins = 0;;) {
                bool stale = false;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int w0 = TBL[(2 * k) * 64 + lane], w1 = TBL[(2 * k + 1) * 64 + lane];
                    unsigned long long w = (unsigned long long)want << 32;
                    if (((w0 >> 16) & 3) == 3) w = ld_pair(xs + w1);
                    hv[k] = __uint_as_float((unsigned)w);
                    stale = stale || (unsigned)(w >> 32) != want;
                }
                if (!__any(stale)) break;
                bool give_up = ++spins >= (1 << 15);
                if ((spins & 31) == 0) give_up = give_up || __any(lane == 8 && (unsigned)__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch);
                if (give_up) return false;
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                const int w0 = TBL[(2 * k) * 64 + lane];
                if (((w0 >> 16) & 3) == 3) {
                    if (w0 & (1 << 18)) XC[w0 & 0xffff] = hv[k];
                    else Zt[w0 & 0xffff] = hv[k];
                }
            }
            return true;
        };
        auto post_halo = [&](int value, bool stop_) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            if (lane == 0) {
                if (stop_) __hip_atomic_store(lflag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(lflag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };
        stage_own(0);
        post_halo(1, false);
        bool stop = false;
        for (int t = 0; t < a.T; ++t) {
            __syncthreads();      // barrier P: the compute waves have built the coarse perception of state t (XC is free again)
            __syncthreads();      // barrier A: the border band of state t + 1 is in the LDS tile
            if (t + 1 < a.T && !(a.dbg & 1)) {
                // this tile's coarse cells within 2 of its border, for the neighbours
                const float* const Zn = smem + K::OFF_Z + ((t + 1) & 1) * (CP * PCS);
                unsigned long long* const xd = a.xch + (size_t)((t + 1) & 1) * a.xch_words + (size_t)tile * C * kXchMS + kRingCells;
#pragma unroll
                for (int k = 0; k < KM::NPB; ++k) {
                    const int j = lane + 64 * k, kc = j % kCoarseRing, ch = j / kCoarseRing;
                    int i, jj;
                    if (kc < 16) { i = kc >> 3; jj = kc & 7; }
                    else if (kc < 32) { i = 6 + ((kc - 16) >> 3); jj = kc & 7; }
                    else if (kc < 40) { i = 2 + ((kc - 32) >> 1); jj = (kc - 32) & 1; }
                    else { i = 2 + ((kc - 40) >> 1); jj = 6 + ((kc - 40) & 1); }
                    if (j < kCoarseRing * CP && ch < C) st_pair(xd + ch * kXchMS + kc, mean4(Zn, ch, i, jj), (int)(tag0 + (unsigned)(t + 1)));
                }
                // ... and the neighbours' data of state t + 1 as it arrives (fine ring -> the other LDS buffer's halo, coarse -> XC)
                stop = !stage_global(t + 1);
                if (stop && lane == 0) {
                    if (a.err) __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(abort_w, (int)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            __syncthreads();      // barrier E: state t + 1 complete in LDS
            if (t + 1 < a.T) {
                if (!stop && !(a.dbg & 1)) stage_own(t + 1);
                post_halo(t + 2, stop);
            }
            if (stop) break;
        }
    } else {
        // =================================== compute waves =======================================================================
        for (int t = 0; t < a.T; ++t) {
            float* const dst = a.x_out + (size_t)b * C * plane;
            const float* const Zc = smem + K::OFF_Z + (t & 1) * (CP * PCS);
            float* const Zn = smem + K::OFF_Z + ((t + 1) & 1) * (CP * PCS);
            const float* const mkc = MK + (t & 1) * (PTH * PTW);
            const bool last = t + 1 == a.T;
            bool stop;
            {
                int spins = 0;
                while (__hip_atomic_load(lflag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < t + 1 && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                stop = __hip_atomic_load(lflag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0;
            }
            if (stop) break;
            // ---- coarse perception of state t on coarse cells -1 .. 8 (index-clamped at the image border), from xc -------------------
#pragma unroll
            for (int k = 0; k < KM::NPC; ++k) {
                const int it = tid + 256 * k;
                if (it < PCR2 * PCR2 * CP) {
                    const int cellc = it % (PCR2 * PCR2), ch = it / (PCR2 * PCR2), r = cellc / PCR2, c = cellc % PCR2;
                    const int yy = min(max(cy0 - 1 + r, 0), Hc - 1) - (cy0 - 2), xx = min(max(cx0 - 1 + c, 0), Wc - 1) - (cx0 - 2);
                    float av[3][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) av[dy][dx] = XC[ch * XCS + (yy + dy - 1) * XCR + xx + dx - 1];
                    float* const o = PCL + ch * PCS2 + r * PCR2 + c;
                    o[0] = av[1][1];
                    o[CP * PCS2] = nca_sobel_x(av);
                    o[2 * CP * PCS2] = nca_sobel_y(av);
                    o[3 * CP * PCS2] = nca_laplacian(av);
                }
            }
            __syncthreads();      // barrier P
            auto phase = [&](auto ntp_tag, int j0) {
                constexpr int NTP = decltype(ntp_tag)::value;
                int rr[NTP], qq[NTP];
#pragma unroll
                for (int n = 0; n < NTP; ++n) group_cell_ms(j0 + n, ci, rr[n], qq[n]);
                float P[NTP][K::K1S];
#pragma unroll
                for (int cq4 = 0; cq4 < CP / 4; ++cq4) {
                    const float* const zc = Zc + (4 * cq4 + g) * PCS + 3;
#pragma unroll
                    for (int n = 0; n < NTP; ++n) {
                        float nbv[3][3];
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) nbv[dy][dx] = zc[(rr[n] + dy) * PRS + qq[n] + dx];
                        P[n][4 * cq4 + 0] = nbv[1][1];
                        P[n][4 * cq4 + 1] = nca_sobel_x(nbv);
                        P[n][4 * cq4 + 2] = nca_sobel_y(nbv);
                        P[n][4 * cq4 + 3] = nca_laplacian(nbv);
                        // + bilinear x2 up-sampling (align_corners = False) of the coarse perception, then the mean over the two scales
                        // (dynca.py:98, :105-110): the per-step kernel's expressions, in its order
                        const int fr = rr[n], fq = qq[n];
                        const int kr = (fr >> 1) + ((fr & 1) ? 1 : 0), kq = (fq >> 1) + ((fq & 1) ? 1 : 0);
                        const float h1 = (fr & 1) ? 0.25f : 0.75f, w1 = (fq & 1) ? 0.25f : 0.75f, h0 = 1.0f - h1, w0 = 1.0f - w1;
                        const float* const pcp = PCL + (4 * cq4 + g) * PCS2 + kr * PCR2 + kq;
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const float* const q = pcp + f * CP * PCS2;
                            P[n][4 * cq4 + f] = nca_up2_blend(P[n][4 * cq4 + f], q[0], q[1], q[PCR2], q[PCR2 + 1], h0, h1, w0, w1);
                        }
                    }
                }
                if (HAS_COND) {
#pragma unroll
                    for (int n = 0; n < NTP; ++n) P[n][CP] = CN[g * PTH * PTW + rr[n] * PTW + qq[n]];
                }
                f32x4 acc2[NTP];
                {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(B2L + 4 * g);
#pragma unroll
                    for (int n = 0; n < NTP; ++n) acc2[n] = bias;
                }
                float wa1[K::K1S], wa2[4];
                f32x4 bias1;
                auto fetch = [&](int m) {
                    const float* const w1m = W1L + m * K::K1S * 64 + lane;
#pragma unroll
                    for (int s_ = 0; s_ < K::K1S; ++s_) wa1[s_] = w1m[s_ * 64];
                    const float* const w2m = W2L + (4 * m) * 64 + lane;
#pragma unroll
                    for (int r = 0; r < 4; ++r) wa2[r] = w2m[r * 64];
                    bias1 = *reinterpret_cast<const f32x4*>(B1L + 16 * m + 4 * g);
                };
                fetch(0);
#pragma unroll 1
                for (int m = 0; m < K::M1T; ++m) {
                    f32x4 acc1[NTP];
#pragma unroll
                    for (int n = 0; n < NTP; ++n) acc1[n] = bias1;
#pragma unroll
                    for (int s_ = 0; s_ < K::K1S; ++s_)
#pragma unroll
                        for (int n = 0; n < NTP; ++n) acc1[n] = nca_mfma(wa1[s_], P[n][s_], acc1[n]);
                    float w2c[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) w2c[r] = wa2[r];
                    __builtin_amdgcn_sched_barrier(0);
                    if (m + 1 < K::M1T) fetch(m + 1);
                    float h[NTP][4];
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[n][r] = __int_as_float(max(__float_as_int(acc1[n][r]), 0));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int n = 0; n < NTP; ++n) acc2[n] = nca_mfma(w2c[r], h[n][r], acc2[n]);
                }
#pragma unroll
                for (int n = 0; n < NTP; ++n) {
                    const float mk = mkc[rr[n] * PTW + qq[n]];
                    const bool ring = rr[n] == 0 || rr[n] == PTH - 1 || qq[n] == 0 || qq[n] == PTW - 1;
                    const size_t o0 = (size_t)(ty0 + rr[n]) * W + tx0 + qq[n];
                    unsigned long long* const xd = a.xch + (size_t)((t + 1) & 1) * a.xch_words + (size_t)tile * C * kXchMS +
                                                   (ring ? ring_index(rr[n], qq[n]) : 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ch = 4 * g + r;
                        if (ch < CP) {
                            const int zo = ch * PCS + (rr[n] + 1) * PRS + qq[n] + 4;
                            const float xn = Zc[zo] + acc2[n][r] * mk;
                            Zn[zo] = ch < C ? xn : 0.0f;
                            if (ch < C) {
                                if (last) dst[(size_t)ch * plane + o0] = xn;
                                else if (ring) st_pair(xd + ch * kXchMS, xn, (int)(tag0 + (unsigned)(t + 1)));
                            }
                        }
                    }
                }
            };
            phase(std::integral_constant<int, 3>{}, 3 * wave);      // the 192-cell border band: what the neighbours' next step needs
            if (!last) fill_mask(t + 1, tid, 256);
            __syncthreads();      // barrier A
            phase(std::integral_constant<int, 1>{}, 12 + wave);     // the 8 x 8 centre, under the exchange
            __syncthreads();      // barrier E
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
}

template <int CP, int FC, bool HAS_COND>
hipError_t launch_persist_ms(const NcaDyncaPersistArgs& a, hipStream_t st, bool query_only, bool* fits) {
    using KM = PersistMsCfg<CP, FC, HAS_COND>;
    auto kern = dynca_persist_ms_kernel<CP, FC, HAS_COND>;
    const size_t lds = (size_t)KM::LDS_FLOATS * sizeof(float) > 81 * 1024 ? (size_t)KM::LDS_FLOATS * sizeof(float) : (size_t)81 * 1024;
    static NcaLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int ntiles = a.B * (a.H / PTH) * (a.W / PTW);
    static std::atomic<int> occ[kNcaMaxDevices];
    int per_cu = occ[nca_device_index()].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kPT, lds); e != hipSuccess) return e;
        occ[nca_device_index()].store(per_cu > 0 ? per_cu : -1, std::memory_order_relaxed);
    }
    *fits = (long)per_cu * nca_cu_count() >= ntiles;
    if (!*fits || query_only) return hipSuccess;
    const int grid = g_drop_tiles > 0 && g_drop_tiles < ntiles ? ntiles - g_drop_tiles : ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPT), lds, st, a);
    return hipGetLastError();
}

template <int CP, int FC, bool HAS_COND>
hipError_t launch_persist(const NcaDyncaPersistArgs& a, hipStream_t st, bool query_only, bool* fits) {
    using K = PersistCfg<CP, FC, HAS_COND>;
    auto kern = dynca_persist_kernel<CP, FC, HAS_COND>;
    // more than half a CU's LDS: ONE workgroup per CU (two tiles on one CU would share its matrix pipes while another CU idles)
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float) > 81 * 1024 ? (size_t)K::LDS_FLOATS * sizeof(float) : (size_t)81 * 1024;
    static NcaLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int ntiles = a.B * (a.H / PTH) * (a.W / PTW);
    static std::atomic<int> occ[kNcaMaxDevices];     // workgroups per CU of this instantiation (queried once per device)
    int per_cu = occ[nca_device_index()].load(std::memory_order_relaxed);
    if (per_cu == 0) {
        if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kPT, lds); e != hipSuccess) return e;
        occ[nca_device_index()].store(per_cu > 0 ? per_cu : -1, std::memory_order_relaxed);
    }
    *fits = (long)per_cu * nca_cu_count() >= ntiles;      // every workgroup must be resident at once (neighbours wait for each other)
    if (!*fits || query_only) return hipSuccess;
    // test hook (ncahip_debug_persist_drop_tiles): launch only the first tiles -- the others' neighbours never hear from them, their
    // bounded polls expire, the launch drains and the sticky error word carries bit 1: what a non-resident workgroup looks like
    const int grid = g_drop_tiles > 0 && g_drop_tiles < ntiles ? ntiles - g_drop_tiles : ntiles;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kPT), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// Shapes the persistent kernel covers (the co-residency test needs the device: done at launch).
bool nca_dynca_persist_shape_ok(int B, int C, int H, int W, int fc, int c_cond) {
    return B >= 1 && C >= 1 && C <= 16 && fc >= 1 && fc <= 128 && c_cond >= 0 && c_cond <= 4 && H % PTH == 0 && W % PTW == 0 &&
           (long)B * (H / PTH) * (W / PTW) <= 4096 && (size_t)C * H * W < ((size_t)1 << 31);
}
int nca_dynca_persist_tiles(int B, int H, int W) { return B * (H / PTH) * (W / PTW); }

hipError_t nca_launch_dynca_persist_ms(const NcaDyncaPersistArgs& a_in, hipStream_t st, bool query_only, bool* fits) {
    NcaDyncaPersistArgs a = a_in;
    a.err = nca_error_word_device();
    static const int dbg = getenv("NCAHIP_PERSIST_DBG") ? atoi(getenv("NCAHIP_PERSIST_DBG")) : 0;
    a.dbg = dbg;
    const bool small = a.C <= 12 && a.fc <= 96;
    if (a.c_cond > 0) return small ? launch_persist_ms<12, 96, true>(a, st, query_only, fits) : launch_persist_ms<16, 128, true>(a, st, query_only, fits);
    return small ? launch_persist_ms<12, 96, false>(a, st, query_only, fits) : launch_persist_ms<16, 128, false>(a, st, query_only, fits);
}

hipError_t nca_launch_dynca_persist(const NcaDyncaPersistArgs& a_in, hipStream_t st, bool query_only, bool* fits) {
    NcaDyncaPersistArgs a = a_in;
    a.err = nca_error_word_device();
    static const int dbg = getenv("NCAHIP_PERSIST_DBG") ? atoi(getenv("NCAHIP_PERSIST_DBG")) : 0;
    a.dbg = dbg;
    const bool small = a.C <= 12 && a.fc <= 96;
    if (a.c_cond > 0) return small ? launch_persist<12, 96, true>(a, st, query_only, fits) : launch_persist<16, 128, true>(a, st, query_only, fits);
    return small ? launch_persist<12, 96, false>(a, st, query_only, fits) : launch_persist<16, 128, false>(a, st, query_only, fits);
}
