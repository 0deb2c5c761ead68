// nca_cond_pc.hip -- ConditionedNCA fused step, producer/consumer wave specialisation (gfx950, fp32 exact).
//
// Same math, MFMA mapping and tile staging code as nca_cond_wave.hip, different division of labour.  The
// exact-f32 MFMA runs at the vector rate and does not co-execute with VALU work of its own wave, and the mask
// resolution in front of it is a chain of dependent LDS round trips: a wave that does both leaves the matrix
// pipe idle half the time even with a second such wave on the SIMD.  Here each SIMD hosts ONE consumer wave
// (perception -> MFMA chain -> 16-byte stores; nothing else) and ONE producer wave (global loads, pending
// life-mask resolution, z tile, resolved-state copy, fire mask) that works one tile ahead through a
// double-buffered LDS tile; the two hand tiles over through round counters in LDS (no workgroup barrier in the loop).
//   workgroup = 8 waves = 4 pairs; pair p owns the 4x16 tiles at rows 4p..4p+3 of each 16x16 super-tile.
#include "nca_cond_tile.h"

namespace {

#ifndef NCA_WT_STORE
#define NCA_WT_STORE 1
#endif
constexpr bool kNtStore = NCA_WT_STORE != 0;   // write-through output stores (see store_tile)

// LDS carve of the pairs.  Offsets are floats from the start of the allocation; `which` is the tile buffer (0/1).
//   CP <= 16: [weight image F::SHARED][pair 0: buf0 buf1 scratch][pair 1] ...  with XR of 16*M3T rows (whole MFMA row tiles).
//   CP  > 16 (the reference's default C = 20): the straightforward carve is 217 KB.  Two things bring it to 159 KB:
//     * XR holds exactly CP rows (mlp_tile_regs guards the rows of the last output tile), and
//     * the W1 / W2 images are dead once the consumers have moved them into registers (mlp_load_regs), so the SECOND tile buffers
//       of pairs 0 and 1 live on top of them; the producers wait at one more workgroup barrier before they stage into a second
//       buffer for the first time.  W3, the biases and the perception taps stay resident (W3 is read from LDS per pass here: 160
//       operand registers would not fit beside the rest of the consumer).
template <int CP>
struct PCfg {
    using F = WCfg<CP>;
    static constexpr bool WIDE = CP > 16;
    static constexpr int XR_ROWS = WIDE ? CP : 16 * F::M3T;
    static constexpr int SZ_Z = CP * CS, SZ_XR = XR_ROWS * XRS, SZ_MK = WTH * WTW;
    static constexpr int BUF = SZ_Z + SZ_XR + SZ_MK;               // one tile buffer (floats)
    static constexpr int SCR_A3 = 0;                               // producer scratch of a pair, not double-buffered
    static constexpr int SCR_LIFE = SCR_A3 + (WTH + 6) * RS;
    static constexpr int SCR_A2 = SCR_LIFE + (WTH + 4) * RS;
    static constexpr int SCR_FLAG = SCR_A2 + (WTH + 4) * RS;       // [0] last round staged, [1] last round consumed (ints)
    static constexpr int SCR = SCR_FLAG + 4;
    static constexpr int PAIR = 2 * BUF + SCR;                     // narrow carve: one contiguous region per pair
    // wide carve
    static constexpr int ALIAS_END = F::OFF_W3;                    // [0, OFF_W3) = the W1 and W2 images
    static constexpr int W_BUF0 = F::SHARED;                       // 4 x buf0
    static constexpr int W_SCR = W_BUF0 + 4 * BUF;                 // 4 x scratch
    static constexpr int W_XR1P1 = W_SCR + 4 * SCR;                // XR of pair 1's second buffer (its Z and MK sit in the alias region)
    static constexpr int W_BUF1 = W_XR1P1 + SZ_XR;                 // second buffers of pairs 2 and 3
    static constexpr int LDS_FLOATS = WIDE ? W_BUF1 + 2 * BUF : F::SHARED + 4 * PAIR;
    static_assert(!WIDE || BUF + SZ_Z + SZ_MK <= ALIAS_END, "pair 0's second buffer + pair 1's second Z / MK fit on the dead images");
    static_assert(BUF % 4 == 0 && SCR % 4 == 0 && SZ_Z % 4 == 0 && SZ_XR % 4 == 0 && F::SHARED % 4 == 0, "16-byte carve");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    struct Offs { int z, xr, mk, scr; };
    static __device__ __forceinline__ Offs offs(int pair, int which) {
        if constexpr (!WIDE) {
            const int b = F::SHARED + pair * PAIR + which * BUF;
            return Offs{b, b + SZ_Z, b + SZ_Z + SZ_XR, F::SHARED + pair * PAIR + 2 * BUF};
        } else {
            const int scr = W_SCR + pair * SCR;
            if (which == 0) {
                const int b = W_BUF0 + pair * BUF;
                return Offs{b, b + SZ_Z, b + SZ_Z + SZ_XR, scr};
            }
            if (pair == 0) return Offs{0, SZ_Z, SZ_Z + SZ_XR, scr};
            if (pair == 1) return Offs{BUF, W_XR1P1, BUF + SZ_Z, scr};
            const int b = W_BUF1 + (pair - 2) * BUF;
            return Offs{b, b + SZ_Z, b + SZ_Z + SZ_XR, scr};
        }
    }
};

template <int CP, bool EXACT, typename ST, bool SPLIT = false>
__global__ __launch_bounds__(kThreadsW, 2) void cond_step_fwd_pc_kernel(const NcaCondArgs a) {
    static_assert(!SPLIT || ST::BYTES == 4, "bf16x3 emulation is an option of the fp32-storage kernel");
    constexpr bool BF = ST::BYTES == 2;   // bf16 storage: UpdateNet on bf16 MFMA (operands rounded from the same LDS image)
    using K = WCfg<CP>;
    using PK = PCfg<CP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = EXACT ? CP : a.C, H = a.H, W = a.W, hid = EXACT ? 64 : a.hidden, K1 = 3 * C;
    const bool producer = wave >= 4;
    const int pair = wave & 3;
    NCA_KSTAMP(0);

    // 16-byte A-operand images (same layouts as nca_cond_wave.hip), built by the four consumer waves while the
    // producers already stage the first tile
    if (!producer) {
        // all loads of all images in flight together, then the LDS writes (one cold round trip instead of eight)
        FillRegs<4 * K::K1S4 * 256, 256> f1;
        FillRegs<4 * 16 * 64, 256> f2;
        FillRegs<K::M3T * 16 * 64, 256> f3;
        FillRegs<K::HID, 256> fb1, fb2;
        FillRegs<CP * K::WPS, 256> fwp;
        {
            fill_load(f1, a.w1, tid, [&](int idx) -> long {
                const int j = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) % K::K1S4, m = (idx >> 8) / K::K1S4;
                const int s = 4 * q + j, gg = l >> 4, o = 16 * m + (l & 15);
                const int ch = 4 * (s / 3) + gg, f = s % 3;
                return (s < K::K1S && ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;
            });
            fill_load(f2, a.w2, tid, [&](int idx) -> long {
                const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m2 = idx >> 10;
                const int gg = l >> 4, o = 16 * m2 + (l & 15), k = 16 * m + 4 * gg + r;
                return (o < hid && k < hid) ? (long)o * hid + k : -1;
            });
            fill_load(f3, a.w3, tid, [&](int idx) -> long {
                const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m3 = idx >> 10;
                const int gg = l >> 4, o = 16 * m3 + (l & 15), k = 16 * m + 4 * gg + r;
                return (o < C && k < hid) ? (long)o * hid + k : -1;
            });
        }
        fill_load(fb1, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        fill_load(fb2, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        fill_load(fwp, a.wp, tid, [&](int idx) -> long {
            const int ch = idx / K::WPS, j = idx % K::WPS;
            return (ch < C && j < 27) ? (long)ch * 27 + j : -1;
        });
        {
            fill_store(f1, smem + K::OFF_W1, tid);
            fill_store(f2, smem + K::OFF_W2, tid);
            fill_store(f3, smem + K::OFF_W3, tid);
        }
        fill_store(fb1, smem + K::OFF_B1, tid);
        fill_store(fb2, smem + K::OFF_B2, tid);
        fill_store(fwp, smem + K::OFF_WP, tid);
    }

    auto lds_of = [&](int which) -> TileLds {
        const typename PK::Offs o = PK::offs(pair, which);
        float* const S = smem + o.scr;
        return TileLds{smem + o.z, smem + o.xr, S + PK::SCR_A3, S + PK::SCR_A3, S + PK::SCR_LIFE, S + PK::SCR_A2, smem + o.mk};
    };

    // every pair walks the same super-tile sequence
    constexpr int PSTH = 16, PSTW = 16;
    const int st_x = (W + PSTW - 1) / PSTW, st_y = (H + PSTH - 1) / PSTH;
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    NcaTileWalk tw = nca_tile_walk(a.B * st_x * st_y);
    // Round k of the XCD's chunk hands tile base + k*nloc + ((local + k) mod nloc) to this workgroup: the nloc workgroups
    // of an XCD still sweep nloc consecutive super-tiles together (shared halos in its L2), but the rotation walks every
    // workgroup across the columns of the image -- with the plain stride a workgroup whose first tile sits on the left or
    // right image border gets ONLY border tiles (nloc = 32 is a multiple of the 16 super-tile columns at 256^2) and
    // finishes ~8 % after the others.
    // Coordinates advance incrementally (no integer division per tile: on this target a runtime division is a ~40-instruction
    // vector sequence, and both roles walk the tiles): t_{k+1} - t_k is nloc + 1, or 1 when the rotation wraps.
    struct Pos { int k, r, t, sx, sy, b; };   // r = (local + k) mod nloc
    const int n_rounds = (tw.end - tw.base + tw.stride - 1) / tw.stride;
    auto pos_first = [&]() -> Pos {
        Pos p{0, tw.local, tw.base + tw.local, 0, 0, 0};
        p.sx = p.t % st_x;                      // the only divisions of the launch
        const int q = p.t / st_x;
        p.sy = q % st_y;
        p.b = q / st_y;
        return p;
    };
    auto advance = [&](Pos p) -> Pos {
        ++p.k;
        int delta = tw.stride + 1;
        if (++p.r == tw.stride) { p.r = 0; delta = 1; }
        p.t += delta;
        p.sx += delta;
        while (p.sx >= st_x) {
            p.sx -= st_x;
            if (++p.sy == st_y) { p.sy = 0; ++p.b; }
        }
        return p;
    };
    auto tile_of = [&](const Pos& p) -> WTile {
        WTile w{0, 0, 0, false, false};
        if (p.k < n_rounds && p.t < tw.end) {
            w.b = p.b;
            w.ty0 = p.sy * PSTH + pair * WTH;
            w.tx0 = p.sx * PSTW;
            w.valid = w.ty0 < H && w.tx0 < W;
            w.inner = w.ty0 >= halo && w.ty0 + WTH + halo <= H && w.tx0 >= halo && w.tx0 + WTW + halo <= W;
        }
        return w;
    };
    int tile_no = 0;
    // (A register-carried prefetch of the NEXT tile's loads was tried here and measured slower, 97 -> 102 us f32 and
    // 60 -> 64 us bf16: border tiles force waits inside the issue sequence, and the producer is off the critical path.)
    auto produce = [&](const WTile& t, int which) {
        if (!t.valid) return;
        TileRegs<CP, ST> R;
        const TileLds L = lds_of(which);
        if (t.inner) {
            issue_loads<CP, true, true, 0, EXACT, ST>(a, t, lane, R);
            stage_tile<CP, false, EXACT, ST>(a, t, L, lane, R, tile_no);
        } else {
            issue_loads<CP, true, true, 1, EXACT, ST>(a, t, lane, R);
            stage_tile<CP, true, EXACT, ST>(a, t, L, lane, R, tile_no);
        }
    };
    MlpRegs<CP> Wr;        // exact-f32 operands (f32 storage)
    MlpRegsBf<CP> Wb;      // bf16 operands (bf16 storage)
    MlpRegsSplit<CP> Ws;   // bf16 hi/lo operand pairs (fp32 storage, ncahip_cond_precision(1))
    auto consume = [&](const WTile& t, int which) {
        if (!t.valid) return;
        const TileLds L = lds_of(which);
        constexpr int NT = 2;
#pragma unroll 1
        for (int pass = 0; pass < WTH / NT; ++pass) {
            float P[NT][K::K1S];
            if (pass == 0) NCA_STAMP(4);
#ifdef NCA_STAMPS
            // diagnostic knobs (stamps build only): 0xD1A8 = no perception, 0xD1A9 = no MLP
            if (a.seed == 0xD1A8ull || a.seed == 0xD1AAull || a.seed == 0xD1ADull) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < K::K1S; ++s_) P[n][s_] = L.Z[lane + n];
            } else
#endif
            perceive_tile_pipe<CP, NT>(smem, L.Z, lane, pass * NT, P);   // next channel group's LDS reads in flight
            if (pass == 0) NCA_STAMP(5);
#ifdef NCA_STAMPS
            if (a.seed == 0xD1A9ull || a.seed == 0xD1ABull || a.seed == 0xD1ADull) {
                float acc_ = 0.0f;
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < K::K1S; ++s_) acc_ += P[n][s_];
                L.XR[lane] = acc_;
            } else
#endif
            if constexpr (SPLIT) mlp_tile_split<CP, NT>(Ws, smem + K::OFF_B1, smem + K::OFF_B2, L.XR, L.MK, lane, pass * NT, P);
            else if constexpr (BF) mlp_tile_bf16<CP, NT>(Wb, smem + K::OFF_B1, smem + K::OFF_B2, L.XR, L.MK, lane, pass * NT, P);
            else mlp_tile_regs<CP, NT>(Wr, smem, L.XR, L.MK, lane, pass * NT, P);
            if (pass == 0) NCA_STAMP(6);
        }
        NCA_STAMP(7);
#ifdef NCA_STAMPS
        if (a.seed == 0xD1ACull || a.seed == 0xD1ADull || a.seed == 0xD1AEull) return;   // diagnostic knob: no store (0xD1AE: nothing else changed)
#endif
        if (t.inner) store_tile<CP, false, EXACT, kNtStore && !BF, ST>(a, t, L.XR, lane);   // bf16 rows are 32-byte segments: let the L2 merge them
        else store_tile<CP, true, EXACT, kNtStore && !BF, ST>(a, t, L.XR, lane);
        NCA_STAMP(8);
    };

    int which = 0;
    Pos pos = pos_first();
    WTile cur = tile_of(pos);
    // Hand-off between the two waves of a pair: two monotonic round counters in LDS instead of a workgroup barrier per
    // tile.  A workgroup barrier keeps the four pairs of a CU in lock-step, so their perception phases (LDS-bandwidth-bound)
    // and their stores all collide; with pair-local hand-offs the pairs drift apart.  LDS operations of one wave execute in
    // order, so "data, then counter" on the writer side and "counter, then data" on the reader side is all the ordering
    // needed.  The polls are bounded (a broken hand-off never hangs the device; it sets the sticky error word, see await).
    int* const flags = reinterpret_cast<int*>(smem + PK::offs(pair, 0).scr + PK::SCR_FLAG);
    auto post = [&](int idx, int round) {
        wave_sync();
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS traffic for the round is done
        if (lane == 0) __hip_atomic_store(flags + idx, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto await = [&](int idx, int round) {
        int spins = 0;
        while (__hip_atomic_load(flags + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < round && ++spins < (1 << 20))
            __builtin_amdgcn_s_sleep(1);
        // an expired poll means the partner wave never posted: the tile about to be used is stale.  The launch still drains (no
        // hung device), but the failure is RECORDED in the sticky host-visible error word, which the grow drivers and
        // ncahip_check_errors turn into a non-zero return code -- never silent wrong numbers.
        if (spins >= (1 << 20) && lane == 0 && a.err) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        wave_sync();
    };
    // The two roles run separate loops (all branches are wave-uniform): the consumer's 128 weight registers are then not
    // live in the producer's code and vice versa.
    if (producer) {
        if (lane == 0) { flags[0] = -1; flags[1] = -1; }
        // At equal priority the (older) consumer waves win every arbitration and the producer only issues in the gaps
        // of their MFMA stream; its instructions are few: let them go first.
        __builtin_amdgcn_s_setprio(3);
        produce(cur, 0);              // overlaps the weight-image fill of the consumer waves
        post(0, 0);
        NCA_KSTAMP(1);
        __syncthreads();              // weight image complete, counters initialised
        if constexpr (PK::WIDE) __syncthreads();   // ... and moved into the consumers' registers: the second buffers may overwrite it
        NCA_KSTAMP(2);
        while (pos.k + 1 < n_rounds) {
            const Pos pn = advance(pos);
            const WTile nxt = tile_of(pn);
            __builtin_amdgcn_s_setprio(0);
            await(1, pn.k - 2);       // the consumer is done with the round that used this buffer
            __builtin_amdgcn_s_setprio(3);
#ifdef NCA_STAMPS
            if (a.seed != 0xD1A6ull && (a.seed < 0xD1AAull || a.seed > 0xD1ADull))  // diagnostic knob (stamps build only): idle producers
#endif
            produce(nxt, which ^ 1);
            post(0, pn.k);
            cur = nxt;
            pos = pn;
            which ^= 1;
            ++tile_no;
        }
        NCA_KSTAMP(3);
    } else {
        NCA_KSTAMP(1);
        __syncthreads();
        NCA_KSTAMP(2);
        if constexpr (SPLIT) load_weights_split_lds<CP>(smem, lane, Ws);
        else if constexpr (BF) load_weights_bf16_lds<CP>(smem, lane, Wb);   // same image, rounded to bf16 operand pairs
        else mlp_load_regs<CP>(smem, lane, Wr);
        if constexpr (PK::WIDE) {
            static_assert(!PK::WIDE || !SPLIT, "wide carve: the exact-f32 and the bf16 consumer");
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the image reads have landed
            __syncthreads();
        }
        while (pos.k < n_rounds) {
            const Pos pn = advance(pos);
            const WTile nxt = tile_of(pn);
            await(0, pos.k);          // this round's tile has been staged
            if (tile_no == 2) NCA_KSTAMP(4);            // light stamps around one steady-state tile
#ifdef NCA_STAMPS
            if (a.seed != 0xD1A7ull)  // diagnostic knob: idle consumers
#endif
            consume(cur, which);
            if (tile_no == 2) NCA_KSTAMP(5);
            post(1, pos.k);
            if (tile_no == 2) NCA_KSTAMP(6);
            if (tile_no == 3) NCA_KSTAMP(7);
            cur = nxt;
            pos = pn;
            which ^= 1;
            ++tile_no;
        }
        NCA_KSTAMP(3);
    }
}

template <int CP, bool EXACT, typename ST = StF32, bool SPLIT = false>
hipError_t launch_cond_pc(const NcaCondArgs& a, hipStream_t st) {
    using PK = PCfg<CP>;
    auto kern = cond_step_fwd_pc_kernel<CP, EXACT, ST, SPLIT>;
    const size_t lds = (size_t)PK::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;   // per instantiation; keyed by device inside
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int cus = nca_cu_count();
    const int nst = a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    hipLaunchKernelGGL(kern, dim3(nst < cus ? nst : cus), dim3(kThreadsW), lds, st, a);
    return hipGetLastError();
}

}  // namespace

// W % 4 == 0 and 16-byte aligned x_in / goal: caller (nca_step_fwd.hip dispatch) guarantees it.
extern "C" void nca_debug_set_stamp_buffer_pc(void* p);
static unsigned long long* g_stamp_pc = nullptr;
extern "C" void nca_debug_set_stamp_buffer_pc(void* p) { g_stamp_pc = (unsigned long long*)p; }
#if defined(NCA_STAMPS)
unsigned long long* nca_debug_stamp_ptr() { return g_stamp_pc; }
#endif
static int g_cond_precision = 0;
void nca_set_cond_precision(int mode) { g_cond_precision = mode; }
hipError_t nca_launch_cond_step_fwd_pc(const NcaCondArgs& a_in, hipStream_t st) {
    NcaCondArgs a = a_in;
    a.dbg = g_stamp_pc;
    a.err = nca_error_word_device();
    const bool h64 = a.hidden == 64;
    if (g_cond_precision == 1 && h64) {   // opt-in bf16x3 emulation of the fp32 products (exact shapes only)
        if (a.C == 12) return launch_cond_pc<12, true, StF32, true>(a, st);
        if (a.C == 16) return launch_cond_pc<16, true, StF32, true>(a, st);
    }
    if (a.C == 12 && h64) return launch_cond_pc<12, true>(a, st);
    if (a.C == 16 && h64) return launch_cond_pc<16, true>(a, st);
    if (a.C <= 12) return launch_cond_pc<12, false>(a, st);
    if (a.C <= 16) return launch_cond_pc<16, false>(a, st);
    // the reference's default model, C = 3 + 1 + 16 = 20 (nca.py:62-94)
    if (a.C == 20 && h64) return launch_cond_pc<20, true>(a, st);
    if (a.C <= 20) return launch_cond_pc<20, false>(a, st);
    return hipErrorInvalidValue;
}

// bf16 state / goal behind the float-typed pointers of NcaCondArgs.  Caller (nca_capi.hip) guarantees W % 4 == 0, 8-byte
// aligned x_in / x_out / goal, C <= 20, hidden <= 64, H*W < 2^24.
hipError_t nca_launch_cond_step_fwd_bf16(const NcaCondArgs& a_in, hipStream_t st) {
    NcaCondArgs a = a_in;
    a.dbg = g_stamp_pc;
    a.err = nca_error_word_device();
    const bool h64 = a.hidden == 64;
    if (a.C == 12 && h64) return launch_cond_pc<12, true, StBF16>(a, st);
    if (a.C == 16 && h64) return launch_cond_pc<16, true, StBF16>(a, st);
    if (a.C <= 12) return launch_cond_pc<12, false, StBF16>(a, st);
    if (a.C <= 16) return launch_cond_pc<16, false, StBF16>(a, st);
    if (a.C == 20 && h64) return launch_cond_pc<20, true, StBF16>(a, st);
    if (a.C <= 20) return launch_cond_pc<20, false, StBF16>(a, st);
    return hipErrorInvalidValue;
}
