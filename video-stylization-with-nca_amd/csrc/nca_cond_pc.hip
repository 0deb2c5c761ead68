// nca_cond_pc.hip -- ConditionedNCA fused step, producer/consumer wave specialisation (gfx950, fp32 exact).
//
// Same math, MFMA mapping and tile staging code as nca_cond_wave.hip, different division of labour.  The
// exact-f32 MFMA runs at the vector rate and does not co-execute with VALU work of its own wave, and the mask
// resolution in front of it is a chain of dependent LDS round trips: a wave that does both leaves the matrix
// pipe idle half the time even with a second such wave on the SIMD.  Here each SIMD hosts ONE consumer wave
// (perception -> MFMA chain -> 16-byte stores; nothing else) and ONE producer wave (global loads, pending
// life-mask resolution, z tile, resolved-state copy, fire mask) that works one tile ahead through a
// double-buffered LDS tile.  One workgroup barrier per tile hands the buffers over.
//   workgroup = 8 waves = 4 pairs; pair p owns the 4x16 tiles at rows 4p..4p+3 of each 16x16 super-tile.
#include "nca_cond_tile.h"

namespace {

template <int CP>
struct PCfg {
    using F = WCfg<CP>;
    static constexpr int BUF_Z = 0;
    static constexpr int BUF_XR = BUF_Z + CP * CS;
    static constexpr int BUF_MK = BUF_XR + CP * XRS;
    static constexpr int BUF = BUF_MK + WTH * WTW;                 // one tile buffer (floats)
    static constexpr int SCR_A3 = 2 * BUF;                         // producer scratch, not double-buffered
    static constexpr int SCR_LIFE = SCR_A3 + (WTH + 6) * RS;
    static constexpr int SCR_A2 = SCR_LIFE + (WTH + 4) * RS;
    static constexpr int PAIR = SCR_A2 + (WTH + 4) * RS;
    static constexpr int LDS_FLOATS = F::SHARED + 4 * PAIR;
    static_assert(BUF % 4 == 0 && PAIR % 4 == 0 && BUF_XR % 4 == 0, "16-byte carve");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

template <int CP>
__global__ __launch_bounds__(kThreadsW, 2) void cond_step_fwd_pc_kernel(const NcaCondArgs a) {
    using K = WCfg<CP>;
    using PK = PCfg<CP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W, hid = a.hidden, K1 = 3 * C;
    const bool producer = wave >= 4;
    const int pair = wave & 3;

    // 16-byte A-operand images (same layouts as nca_cond_wave.hip)
    fill_image_w<4 * K::K1S4 * 256>(smem + K::OFF_W1, a.w1, tid, [&](int idx) -> long {
        const int j = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) % K::K1S4, m = (idx >> 8) / K::K1S4;
        const int s = 4 * q + j, gg = l >> 4, o = 16 * m + (l & 15);
        const int ch = 4 * (s / 3) + gg, f = s % 3;
        return (s < K::K1S && ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;
    });
    fill_image_w<4 * 16 * 64>(smem + K::OFF_W2, a.w2, tid, [&](int idx) -> long {
        const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m2 = idx >> 10;
        const int gg = l >> 4, o = 16 * m2 + (l & 15), k = 16 * m + 4 * gg + r;
        return (o < hid && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image_w<K::M3T * 16 * 64>(smem + K::OFF_W3, a.w3, tid, [&](int idx) -> long {
        const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m3 = idx >> 10;
        const int gg = l >> 4, o = 16 * m3 + (l & 15), k = 16 * m + 4 * gg + r;
        return (o < C && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image_w<K::HID>(smem + K::OFF_B1, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image_w<K::HID>(smem + K::OFF_B2, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image_w<CP * K::WPS>(smem + K::OFF_WP, a.wp, tid, [&](int idx) -> long {
        const int ch = idx / K::WPS, j = idx % K::WPS;
        return (ch < C && j < 27) ? (long)ch * 27 + j : -1;
    });

    float* const PR = smem + K::SHARED + pair * PK::PAIR;
    auto lds_of = [&](int which) -> TileLds {
        float* const B = PR + which * PK::BUF;
        return TileLds{B + PK::BUF_Z, B + PK::BUF_XR, PR + PK::SCR_A3, PR + PK::SCR_A3, PR + PK::SCR_LIFE, PR + PK::SCR_A2,
                       B + PK::BUF_MK};
    };

    // every pair walks the SAME super-tile sequence (uniform trip count: the barrier below is workgroup-wide)
    constexpr int PSTH = 16, PSTW = 16;
    const int st_x = (W + PSTW - 1) / PSTW, st_y = (H + PSTH - 1) / PSTH;
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    NcaTileWalk tw = nca_tile_walk(a.B * st_x * st_y);
    auto tile_at = [&](int t) -> WTile {
        WTile w{0, 0, 0, false, false};
        if (t < tw.end) {
            w.b = t / (st_x * st_y);
            w.ty0 = ((t / st_x) % st_y) * PSTH + pair * WTH;
            w.tx0 = (t % st_x) * PSTW;
            w.valid = w.ty0 < H && w.tx0 < W;
            w.inner = w.ty0 >= halo && w.ty0 + WTH + halo <= H && w.tx0 >= halo && w.tx0 + WTW + halo <= W;
        }
        return w;
    };
    auto produce = [&](const WTile& t, int which) {
        if (!t.valid) return;
        TileRegs<CP> R;
        issue_loads<CP, true, true>(a, t, lane, R);
        const TileLds L = lds_of(which);
        if (t.inner) stage_tile<CP, false>(a, t, L, lane, R, 0);
        else stage_tile<CP, true>(a, t, L, lane, R, 0);
    };
    auto consume = [&](const WTile& t, int which) {
        if (!t.valid) return;
        const TileLds L = lds_of(which);
        constexpr int NT = 2;
#pragma unroll 1
        for (int pass = 0; pass < WTH / NT; ++pass) {
            float P[NT][K::K1S];
            perceive_tile<CP, NT>(smem, L.Z, lane, pass * NT, P);
            mlp_tile<CP, NT>(a, smem, L.XR, L.MK, lane, pass * NT, P);
        }
        if (t.inner) store_tile<CP, false>(a, t, L.XR, lane);
        else store_tile<CP, true>(a, t, L.XR, lane);
    };

    int t = tw.t, which = 0;
    WTile cur = tile_at(t);
    if (producer) produce(cur, 0);   // overlaps the tail of the weight-image fill of the other waves
    __syncthreads();                  // weight image + first tile ready
    while (t < tw.end) {              // uniform over the workgroup
        const int tn = t + tw.stride;
        const WTile nxt = tile_at(tn);
        if (producer) produce(nxt, which ^ 1);
        else consume(cur, which);
        __syncthreads();              // tile buffers change hands
        cur = nxt;
        t = tn;
        which ^= 1;
    }
}

template <int CP>
hipError_t launch_cond_pc(const NcaCondArgs& a, hipStream_t st) {
    using PK = PCfg<CP>;
    auto kern = cond_step_fwd_pc_kernel<CP>;
    const size_t lds = (size_t)PK::LDS_FLOATS * sizeof(float);
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static thread_local int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
    }
    const int nst = a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    hipLaunchKernelGGL(kern, dim3(nst < cus ? nst : cus), dim3(kThreadsW), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// W % 4 == 0 and 16-byte aligned x_in / goal: caller (nca_step_fwd.hip dispatch) guarantees it.
hipError_t nca_launch_cond_step_fwd_pc(const NcaCondArgs& a, hipStream_t st) {
    if (a.C <= 12) return launch_cond_pc<12>(a, st);
    if (a.C <= 16) return launch_cond_pc<16>(a, st);
    return hipErrorInvalidValue;
}
