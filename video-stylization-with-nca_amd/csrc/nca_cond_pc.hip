// nca_cond_pc.hip -- ConditionedNCA fused step, producer/consumer wave specialisation (gfx950, fp32 exact).
//
// An exact-f32 MFMA does not co-execute with VALU work on this chip: every vector instruction that lands inside
// an MFMA stream first drains the matrix pipe (~25 cycles per isolated instruction, tools/micro/mlp_pass.hip), and
// with in-order issue every LDS/global load placed next to its use stalls it for the full latency.  So the step is
// split by *kind of instruction*, not by tile:
//   consumer wave (one per SIMD): ONLY the UpdateNet -- 512 MFMAs per 4x16-cell tile with all 128 A-operand registers
//       resident for the whole launch, ReLUs in fenced groups, B operands of layer 1 read ready-made from LDS, masked
//       residual, 16-byte write-through stores;
//   producer wave (one per SIMD): everything else, one tile ahead -- global loads, pending life-mask resolution,
//       z = x + goal*pre, the learned 3x3 depthwise perception (v_pk_fma), fire mask -- handing over the perception
//       output P in MFMA B-operand layout through a double-buffered LDS tile.  Its latencies are off the critical path.
// One workgroup barrier per tile swaps the buffers.  No shared LDS data: each consumer lane loads its UpdateNet
// operands straight from the weight tensors (the "g-major" feature order makes them contiguous 16-byte slices), each
// producer wave keeps a private copy of the perception weights.
//   workgroup = 8 waves = 4 pairs; pair p owns the 4x16 tiles at rows 4p..4p+3 of each 16x16 super-tile.
#include "nca_cond_tile.h"
#include <type_traits>

namespace {

#ifndef NCA_WT_STORE
#define NCA_WT_STORE 1
#endif
constexpr bool kNtStore = NCA_WT_STORE != 0;   // write-through output stores (see store_tile)

template <int CP>
struct PCfg {
    using F = WCfg<CP>;
    static constexpr int ZCS = CP == 16 ? 148 : 144;               // (CP/4)*ZCS % 32 == 16 (perceive_rows4)
    static constexpr int PBUF = WTH * F::K1S4 * 64 * 4;            // P[row][s/4][lane][4]; doubles as the producer's z tile
    static constexpr int XRB = 16 * F::M3T * XRS;                  // resolved state, whole MFMA row tiles
    static constexpr int OFF_PB = 0;                               // x2
    static constexpr int OFF_XR = OFF_PB + 2 * PBUF;               // x2
    static constexpr int OFF_MK = OFF_XR + 2 * XRB;                // x2
    static constexpr int OFF_A3 = OFF_MK + 2 * WTH * WTW;          // producer scratch: alpha' halo 3 / pre mask
    static constexpr int OFF_LIFE = OFF_A3 + (WTH + 6) * RS;
    static constexpr int OFF_A2 = OFF_LIFE + (WTH + 4) * RS;
    static constexpr int OFF_WP = OFF_A2 + (WTH + 4) * RS;         // producer's copy of the perception weights [CP][28]
    static constexpr int OFF_BL = OFF_WP + CP * 28;                // consumer's copy of the biases [b1 64][b2 64]
    static constexpr int PAIR = OFF_BL + 128;
    static constexpr int OFF_FLAG = 4 * PAIR;                       // workgroup-wide: grid-barrier verdict
    static constexpr int LDS_FLOATS = 4 * PAIR + 4;
    static_assert(CP * ZCS <= PBUF, "z tile fits the P buffer it aliases");
    static_assert(PBUF % 4 == 0 && XRB % 4 == 0 && OFF_A3 % 4 == 0 && OFF_WP % 4 == 0 && PAIR % 4 == 0, "16-byte carve");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
};

// Grid-wide barrier between two fused steps of a cooperative launch (every workgroup is resident).  No cache-wide
// write-back / invalidate (measured: +40 us per step when every wave issues buffer_wbl2 / buffer_inv): everything one
// step writes and the next reads -- state, pre mask -- is stored write-through and loaded coherently (sc1, see
// issue_loads / store_tile), so the barrier only has to wait for this wave's stores and count arrivals.  The spin is
// bounded: after ~2 s (or when another workgroup gave up) the launch aborts instead of hanging the device.
__device__ __forceinline__ bool grid_barrier(unsigned* sync, unsigned target, volatile int* flag) {
    __builtin_amdgcn_s_waitcnt(0);   // this wave's write-through stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sync[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        const unsigned long long t0 = wall_clock64();   // 100 MHz
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(&sync[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                wall_clock64() - t0 > 200000000ull) {
                __hip_atomic_store(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *flag = ok;
    }
    __syncthreads();
    return *flag != 0;
}

// g.T == 0: one step described by a0 (ordinary launch).  g.T >= 1: the grow loop's steps 0..T-1 in this one
// (cooperative) launch, see NcaGrowLoop.
template <int CP, bool EXACT>
__global__ __launch_bounds__(kThreadsW, 2) void cond_step_fwd_pc_kernel(const NcaCondArgs a0, const NcaGrowLoop g) {
    NcaCondArgs a = a0;
    using K = WCfg<CP>;
    using PK = PCfg<CP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = EXACT ? CP : a.C, H = a.H, W = a.W, hid = EXACT ? 64 : a.hidden;
    const bool producer = wave >= 4;
    const int pair = wave & 3;
    NCA_KSTAMP(0);

    float* const PR = smem + pair * PK::PAIR;
    // producer view of buffer `which`: the z tile lives in the P buffer it is about to fill
    auto lds_of = [&](int which) -> TileLds {
        return TileLds{PR + PK::OFF_PB + which * PK::PBUF, PR + PK::OFF_XR + which * PK::XRB, PR + PK::OFF_A3, PR + PK::OFF_A3,
                       PR + PK::OFF_LIFE, PR + PK::OFF_A2, PR + PK::OFF_MK + which * (WTH * WTW)};
    };

    // every pair walks the SAME super-tile sequence (uniform trip count: the barrier below is workgroup-wide)
    constexpr int PSTH = 16, PSTW = 16;
    const int st_x = (W + PSTW - 1) / PSTW, st_y = (H + PSTH - 1) / PSTH;
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    NcaTileWalk tw = nca_tile_walk(a.B * st_x * st_y);
    // super-tile coordinates advance incrementally (no integer division per tile)
    struct Pos { int t, sx, sy, b; };
    auto advance = [&](Pos p) -> Pos {
        p.t += tw.stride;
        p.sx += tw.stride;
        while (p.sx >= st_x) {
            p.sx -= st_x;
            if (++p.sy == st_y) { p.sy = 0; ++p.b; }
        }
        return p;
    };
    auto tile_of = [&](const Pos& p) -> WTile {
        WTile w{0, 0, 0, false, false};
        if (p.t < tw.end) {
            w.b = p.b;
            w.ty0 = p.sy * PSTH + pair * WTH;
            w.tx0 = p.sx * PSTW;
            w.valid = w.ty0 < H && w.tx0 < W;
            w.inner = w.ty0 >= halo && w.ty0 + WTH + halo <= H && w.tx0 >= halo && w.tx0 + WTW + halo <= W;
        }
        return w;
    };
    int tile_no = 0;
    int which = 0;
    const Pos pos0{tw.t, tw.t % st_x, (tw.t / st_x) % st_y, tw.t / (st_x * st_y)};   // one division per kernel
    Pos pos = pos0;
    WTile cur = tile_of(pos);
    // fused grow loop: ring slots of step t
    const int nsteps = g.T > 0 ? g.T : 1;
    const size_t slot = (size_t)a0.B * C * H * W, pslot = (size_t)a0.B * H * W;
    int ring_in = 0;
    auto set_step = [&](int t) {
        if (g.T > 0) {
            const int ring_out = ring_in + 1 == g.ring ? 0 : ring_in + 1;
            a.x_in = g.states + (size_t)ring_in * slot;
            a.pre_in = t == 0 ? nullptr : g.pre + (size_t)ring_in * pslot;
            a.x_out = g.states + (size_t)ring_out * slot;
            a.pre_out = g.pre + (size_t)ring_out * pslot;
            a.u = a0.u ? a0.u + (size_t)t * pslot : nullptr;
            a.step = a0.step + (uint64_t)t;
            ring_in = ring_out;
        }
        pos = pos0;
        cur = tile_of(pos);
        which = 0;
    };
    const int kst = nsteps > 2 ? 1 : 0;   // stamped step (light stamps build): a warm one when the launch is fused
    (void)kst;
    volatile int* const gflag = reinterpret_cast<volatile int*>(smem + PK::OFF_FLAG);

    // The two roles run separate loops with the same barrier count (every branch is wave-uniform), so neither role's
    // long-lived registers are live in the other's code.
    if (producer) {
        float* const WPL = PR + PK::OFF_WP;
        // private copy of the perception weights: all loads in flight together, landed after the first tile's own
        // loads have been issued (fill_wp below)
        constexpr int NWP = (CP * 28 + 63) / 64;
        float wpv[NWP];
#pragma unroll
        for (int u = 0; u < NWP; ++u) {   // raw loads only: any use of the value here would wait for the cold miss
            const int idx = lane + 64 * u, ch = idx / 28, j = idx % 28;
            const bool ok = idx < CP * 28 && ch < C && j < 27;
            wpv[u] = a.wp[ok ? ch * 27 + j : 0];          // [3c+f][3][3] == [c][f*9+tap] (nca.py:99-107)
        }
        bool wp_pending = true;
        auto fill_wp = [&]() {
#pragma unroll
            for (int u = 0; u < NWP; ++u) {
                const int idx = lane + 64 * u, ch = idx / 28, j = idx % 28;
                if (idx < CP * 28) WPL[idx] = (ch < C && j < 27) ? wpv[u] : 0.0f;
            }
        };
        auto produce = [&](const WTile& t, int wb) {
            if (!t.valid) return;   // (the perception weights stay pending in registers until the first valid tile)
            TileRegs<CP> R;
            const TileLds L = lds_of(wb);
            if (t.inner) {
                issue_loads<CP, true, true, 0, EXACT>(a, t, lane, R);
                if (wp_pending) { NCA_KSTAMP(4); fill_wp(); NCA_KSTAMP(5); }
                stage_tile<CP, false, EXACT, PK::ZCS>(a, t, L, lane, R, tile_no);
            } else {
                issue_loads<CP, true, true, 1, EXACT>(a, t, lane, R);
                if (wp_pending) { NCA_KSTAMP(4); fill_wp(); NCA_KSTAMP(5); }
                stage_tile<CP, true, EXACT, PK::ZCS>(a, t, L, lane, R, tile_no);
            }
            if (wp_pending) NCA_KSTAMP(6);
            wp_pending = false;
            float P[WTH][K::K1S];
            perceive_rows4<CP, PK::ZCS>(WPL, L.Z, lane, P);   // stage_tile ended with a wave_sync: the z tile is complete
            wave_sync();                                      // every lane's z reads are done: the region becomes P
            float* const PB = L.Z;
#pragma unroll
            for (int row = 0; row < WTH; ++row)
#pragma unroll
                for (int q = 0; q < K::K1S4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = 4 * q + j < K::K1S ? P[row][4 * q + j] : 0.0f;
                    st4(PB + ((row * K::K1S4 + q) * 64 + lane) * 4, v);
                }
        };
        // At equal priority the (older) consumer waves win every arbitration and the producer only issues in the gaps
        // of their MFMA stream; its instructions are few: let them go first.
        __builtin_amdgcn_s_setprio(3);
        for (int t = 0; t < nsteps; ++t) {
            set_step(t);
            produce(cur, 0);
            if (t == kst) NCA_KSTAMP(1);
            __syncthreads();              // first tile ready
            if (t == kst) NCA_KSTAMP(2);
            while (pos.t < tw.end) {      // uniform over the workgroup
                const Pos pn = advance(pos);
                const WTile nxt = tile_of(pn);
#ifdef NCA_STAMPS
                if (a.seed != 0xD1A6ull)  // diagnostic knob (stamps build only): idle producers
#endif
                produce(nxt, which ^ 1);
                __syncthreads();          // tile buffers change hands
                cur = nxt;
                pos = pn;
                which ^= 1;
                ++tile_no;
            }
            if (t == kst) NCA_KSTAMP(3);
            if (t + 1 < nsteps && !grid_barrier(g.sync, (unsigned)(t + 1) * gridDim.x, gflag)) break;
            if (t == kst) NCA_KSTAMP(7);
            if (t + 1 == kst) NCA_KSTAMP(0);
        }
    } else {
        MlpRegs<CP> Wr;
        mlp_load_regs_global<CP, EXACT && CP == 16>(a, lane, Wr);
        float* const BL = PR + PK::OFF_BL;
        {
            const float v1 = a.b1[lane < hid ? lane : 0], v2 = a.b2[lane < hid ? lane : 0];
            BL[lane] = lane < hid ? v1 : 0.0f;
            BL[64 + lane] = lane < hid ? v2 : 0.0f;
        }
        auto consume = [&](const WTile& t, int rb) {
            if (!t.valid) return;
            const float* const PB = PR + PK::OFF_PB + rb * PK::PBUF;
            float* const XR = PR + PK::OFF_XR + rb * PK::XRB;
            const float* const MK = PR + PK::OFF_MK + rb * (WTH * WTW);
            constexpr int NT = 2;
#pragma unroll 1
            for (int pass = 0; pass < WTH / NT; ++pass) {
                int lane_o = lane;
                asm volatile("" : "+v"(lane_o));
                const int g = lane_o >> 4;
                // this pass's B operands of layer 1 and the accumulator seeds: 14 LDS reads issued together, ahead of the
                // MFMA stream (one exposed latency per pass)
                f32x4 pv[NT][K::K1S4], b1[4], b2[4];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int q = 0; q < K::K1S4; ++q) pv[n][q] = ld4(PB + (((pass * NT + n) * K::K1S4 + q) * 64 + lane_o) * 4);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    b1[m] = ld4(BL + 16 * m + 4 * g);
                    b2[m] = ld4(BL + 64 + 16 * m + 4 * g);
                }
                __builtin_amdgcn_sched_barrier(0);
                float P[NT][K::K1S];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < K::K1S; ++s_) P[n][s_] = pv[n][s_ >> 2][s_ & 3];
                mlp_tile_regs<CP, NT>(Wr, b1, b2, XR, MK, lane, pass * NT, P);
            }
            if (t.inner) store_tile<CP, false, EXACT, kNtStore>(a, t, XR, lane);
            else store_tile<CP, true, EXACT, kNtStore>(a, t, XR, lane);
        };
        wave_sync();                  // this wave's bias copy
#ifdef NCA_STAMPS
        NCA_KSTAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NCA_KSTAMP(5);
#endif
        for (int t = 0; t < nsteps; ++t) {
            set_step(t);
            if (t == kst) NCA_KSTAMP(1);
            __syncthreads();
            if (t == kst) NCA_KSTAMP(2);
            while (pos.t < tw.end) {
                const Pos pn = advance(pos);
                const WTile nxt = tile_of(pn);
#ifdef NCA_STAMPS
                if (a.seed != 0xD1A7ull)  // diagnostic knob: idle consumers
#endif
                consume(cur, which);
                __syncthreads();
                cur = nxt;
                pos = pn;
                which ^= 1;
                ++tile_no;
            }
            if (t == kst) NCA_KSTAMP(3);
            if (t + 1 < nsteps && !grid_barrier(g.sync, (unsigned)(t + 1) * gridDim.x, gflag)) break;
            if (t == kst) NCA_KSTAMP(7);
            if (t + 1 == kst) NCA_KSTAMP(0);
        }
    }
}

template <int CP, bool EXACT>
hipError_t launch_cond_pc(const NcaCondArgs& a, const NcaGrowLoop& g, hipStream_t st) {
    using PK = PCfg<CP>;
    auto kern = cond_step_fwd_pc_kernel<CP, EXACT>;
    const size_t lds = (size_t)PK::LDS_FLOATS * sizeof(float);
    static thread_local bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    static thread_local int cus = 0, resident = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            cus = v;
        int per_cu = 0, coop = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kThreadsW, lds) == hipSuccess &&
            hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev) == hipSuccess && coop)
            resident = per_cu * cus;
    }
    const int nst = a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    const int grid = nst < cus ? nst : cus;
    if (g.T <= 0) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreadsW), lds, st, a, g);
        return hipGetLastError();
    }
    if (grid > resident) return hipErrorNotSupported;   // the grid barrier needs every workgroup resident
    NcaCondArgs a_ = a;
    NcaGrowLoop g_ = g;
    void* params[2] = {&a_, &g_};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kern), dim3(grid), dim3(kThreadsW), params, (unsigned)lds, st);
}

template <typename F>
hipError_t dispatch_cond_pc(const NcaCondArgs& a, F&& go) {
    const bool h64 = a.hidden == 64;
    // <16,true> loads its UpdateNet operands as 16-byte slices of the weight tensors
    const bool wal = (((uintptr_t)a.w1 | (uintptr_t)a.w2 | (uintptr_t)a.w3) & 15) == 0;
    if (a.C == 12 && h64) return go(std::integral_constant<int, 12>{}, std::true_type{});
    if (a.C == 16 && h64 && wal) return go(std::integral_constant<int, 16>{}, std::true_type{});
    if (a.C <= 12) return go(std::integral_constant<int, 12>{}, std::false_type{});
    if (a.C <= 16) return go(std::integral_constant<int, 16>{}, std::false_type{});
    return hipErrorInvalidValue;
}

}  // namespace

// W % 4 == 0 and 16-byte aligned x_in / goal: caller (nca_step_fwd.hip dispatch) guarantees it.
extern "C" void nca_debug_set_stamp_buffer_pc(void* p);
static unsigned long long* g_stamp_pc = nullptr;
extern "C" void nca_debug_set_stamp_buffer_pc(void* p) { g_stamp_pc = (unsigned long long*)p; }
hipError_t nca_launch_cond_step_fwd_pc(const NcaCondArgs& a_in, hipStream_t st) {
    NcaCondArgs a = a_in;
    a.dbg = g_stamp_pc;
    const NcaGrowLoop g{nullptr, nullptr, 0, 0, nullptr};
    return dispatch_cond_pc(a, [&](auto cp, auto ex) { return launch_cond_pc<decltype(cp)::value, decltype(ex)::value>(a, g, st); });
}
hipError_t nca_launch_cond_grow_fwd_pc(const NcaCondArgs& a_in, const NcaGrowLoop& g, hipStream_t st) {
    if (g.T < 1 || g.ring < 2 || !g.states || !g.pre || !g.sync) return hipErrorInvalidValue;
    NcaCondArgs a = a_in;
    a.dbg = g_stamp_pc;
    return dispatch_cond_pc(a, [&](auto cp, auto ex) { return launch_cond_pc<decltype(cp)::value, decltype(ex)::value>(a, g, st); });
}
