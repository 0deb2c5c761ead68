// nca_cond_bwd_fm.hip -- backward kernel A of one ConditionedNCA step split by WORK instead of by wave (EncoderConditioning/
// nca.py:181-195 under autograd, conditioned_trainer.py:125-132), gfx950.
//
// nca_cond_bwd.hip's kernel A does everything for a tile in one wave at one wave per SIMD (the 128 weight-gradient
// accumulators pin it there): a third of its time is the FRONT of the tile -- 45 loads, the pending-mask chain, gate,
// perception -- an instruction- and latency-bound stream that nothing overlaps.  Here the front is its own launch:
//   F  cond_step_bwd_front_kernel   no accumulators, ~130 registers, several waves per SIMD: stages the tile exactly as the
//      forward does (s_t, pre_t, z_t, fire mask), writes z_t and dL/dx'_t = G * gate for kernel B, and leaves the two things the
//      matrix part needs in a scratch laid out in MFMA-operand order: the perception vector P (lane-major: the 16-byte value
//      a lane stores is the B operand it will load) and dO = dL/dx'_t * fire mask ([row tile][channel][cell]).
//   M  cond_step_bwd_mlp_kernel     the matrix part (forward recomputation from P, W3^T/W2^T/W1^T, the three weight-gradient
//      products, dL/dperception out): per-cell work with no halo, no masks, no Philox; its only loads are 4 coalesced
//      16-byte reads per 16-cell row.  Same products in the same per-wave order as the one-launch kernel: bitwise the same
//      weight gradients.
// Cost: P and dO cross HBM once (256 B/cell fp32, 128 B/cell bf16).  Kernel B (stencil adjoint) is unchanged.
#include <cstdlib>

#include "nca_cond_bwd_common.h"

#ifndef NCA_FM_NT32
#define NCA_FM_NT32 1   // 16-cell rows per pass of the matrix kernel at CP = 32 (two rows: 58 spilled vector registers)
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// F: front kernel.  One workgroup = one 16 x 16 super-tile, 4 waves, wave-private 4 x 16 tiles; three workgroups per CU.
constexpr int kFrontWaves = 4, kFrontThreads = 256;
template <int CP>
struct FrontCfg {
    using F = WCfg<CP>;
    // workgroups per CU: CP <= 16: 51 KB of LDS and <= 170 registers -> three; the wide instantiations (16 < C <= 32: the
    // reference's default model is C = 20) are LDS-bound at two (CP <= 24: 70 KB) / one (CP = 32: 89 KB)
    static constexpr int OCC = CP <= 16 ? 3 : (CP <= 24 ? 2 : 1);
    static constexpr int KQ = F::K1S4;                          // 16-byte groups of the perception vector per lane (scratch P: [row tile][KQ][64 lanes])
    static constexpr int DOS = 256 * F::M3T;                    // scratch dO: [row tile][16 M3T channels][16 cells]
    static constexpr int PW_Z = 0;                              // z halo 1: [CP][6][RS]
    static constexpr int PW_A3 = CP * CS;                       // alpha' halo 3 (10 rows); later rows 0-5 = PN, rows 6-9 = fire mask
    static constexpr int PW_LIFE = PW_A3 + (WTH + 6) * RS;
    static constexpr int PW_A2 = PW_LIFE + (WTH + 4) * RS;
    static constexpr int PW_A1 = PW_A2 + (WTH + 4) * RS;        // alpha of the pending x'_t, halo 1
    static constexpr int PW = PW_A1 + ZROWS * RS;
    static constexpr int OFF_WP = kFrontWaves * PW;             // behind the tiles: perceive_tile addresses it as WS + F::OFF_WP
    static_assert(OFF_WP >= F::OFF_WP, "WS = smem + OFF_WP - F::OFF_WP stays inside the allocation");
    static constexpr int LDS_FLOATS = OFF_WP + CP * F::WPS;
    static_assert(PW % 4 == 0 && PW_A3 % 4 == 0, "16-byte carve");
};

// EXACT: the launch guarantees C == CP (no channel-padding clamps in the 45 loads, no ch < C guards in the staging).
template <int CP, typename ST, bool BFM, bool EXACT>
__global__ __launch_bounds__(kFrontThreads, FrontCfg<CP>::OCC) void cond_step_bwd_front_kernel(const NcaCondBwdArgs ba) {
    using K = FrontCfg<CP>;
    using FK = WCfg<CP>;
    const NcaCondArgs& a = ba.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W;
    const unsigned plane = (unsigned)(H * W);
    // The perception taps are needed last (after staging, gate and the z / dO outputs): their gather is REQUESTED here, ahead of the tile's
    // own loads, and lands in LDS (with the workgroup barrier) right before the perception -- the kernel used to wait for this round trip
    // and a barrier before it requested anything of its tile.
    FillRegs<CP * FK::WPS, kFrontThreads> frwp;
    fill_load(frwp, a.wp, tid, [&](int idx) -> long {
        const int ch = idx / FK::WPS, j = idx % FK::WPS;
        return (ch < C && j < 27) ? (long)ch * 27 + j : -1;
    });
    const float* const WS = smem + K::OFF_WP - FK::OFF_WP;
    float* const PWR = smem + wave * K::PW;
    float* const Z = PWR + K::PW_Z;
    float* const PN = PWR + K::PW_A3;
    const float* const MK = PWR + K::PW_A3 + ZROWS * RS;
    float* const A1 = PWR + K::PW_A1;
    const TileLds L{Z, nullptr, PWR + K::PW_A3, PWR + K::PW_A3, PWR + K::PW_LIFE, PWR + K::PW_A2, PWR + K::PW_A3 + ZROWS * RS};

    // super-tile of this workgroup: consecutive super-tiles go to the same XCD (workgroups are dealt round-robin over 8 XCDs)
    const int st_x = (W + 15) / 16, st_y = (H + 15) / 16, nst = a.B * st_x * st_y;
    const int chunk = (nst + 7) / 8, sidx = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || sidx >= nst) return;   // (whole workgroup)
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    const bool use_alive = a.alive_ch >= 0;
    WTile t;
    t.b = sidx / (st_x * st_y);
    t.ty0 = ((sidx / st_x) % st_y) * 16 + wave * WTH;
    t.tx0 = (sidx % st_x) * 16;
    if (t.ty0 >= H || t.tx0 >= W) {   // this wave's rows lie below the image: it still delivers its share of the taps (a finished wave does
        fill_store(frwp, smem + K::OFF_WP, tid);   // not count at the barrier the others wait at)
        return;
    }
    t.valid = true;
    t.inner = t.ty0 >= halo && t.ty0 + WTH + halo <= H && t.tx0 >= halo && t.tx0 + WTW + halo <= W;
    const int ty0 = t.ty0, tx0 = t.tx0;
    const size_t rid0 = ((size_t)sidx * kFrontWaves + wave) * WTH;   // scratch row tiles of this wave (as kernel M computes them)

    // ---- every global load of the tile up front: pending x'_t (alpha halo 1 + interior), incoming gradient, forward operands
    const char* const xn = reinterpret_cast<const char*>(ba.x_next) + (size_t)t.b * C * plane * ST::BYTES;
    const float* const gn = ba.g_next + (size_t)t.b * C * plane;
    const int g = lane >> 4, hl = (lane >> 5) & 1, l5 = lane & 31;
    const int row = (lane >> 2) & 3, ff = lane & 3;
    const bool ok = ty0 + row < H && tx0 + 4 * ff + 3 < W;
    const unsigned off = ok ? (unsigned)((ty0 + row) * W + tx0 + 4 * ff) : 0u;
    typename ST::raw1 av[3];
    bool aok[3];
    if (use_alive) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int gy = ty0 - 1 + 2 * k + hl, gx = tx0 - 1 + l5;
            aok[k] = l5 < 18 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            av[k] = ST::gld1(xn, (unsigned)a.alive_ch * plane + (aok[k] ? (unsigned)(gy * W + gx) : 0u));
        }
    }
    typename ST::raw4 xv[CP / 4];
    f32x4 gv[CP / 4];
#pragma unroll
    for (int k = 0; k < CP / 4; ++k) {
        const unsigned ch = (unsigned)min(4 * k + g, C - 1);
        xv[k] = ST::gld4(xn, ch * plane + off);
        gv[k] = ld4(gn + ch * plane + off);
    }
    // ---- forward staging: pre_t (PN), z_t (Z), fire mask (MK); the resolved-state copy is not needed here
    {
        TileRegs<CP, ST> R;
        if (t.inner) {
            issue_loads<CP, true, true, 0, EXACT, ST>(a, t, lane, R);
            stage_tile<CP, false, EXACT, ST, false>(a, t, L, lane, R, 0);
        } else {
            issue_loads<CP, true, true, 1, EXACT, ST>(a, t, lane, R);
            stage_tile<CP, true, EXACT, ST, false>(a, t, L, lane, R, 0);
        }
    }
    // ---- alpha of the pending state -> A1 (post mask of this step);  z_t interior out (kernel B's perception-weight gradient)
    if (use_alive) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (l5 < 18) A1[(2 * k + hl) * RS + l5 + 3] = aok[k] ? ST::cv1(av[k]) : NCA_NEG_INF;
    }
    {
        float* const zo = ba.zbuf + (size_t)t.b * C * plane + off;
        uint16_t* const zo16 = reinterpret_cast<uint16_t*>(ba.zbuf) + (size_t)t.b * C * plane + off;
#pragma unroll
        for (int k = 0; k < CP / 4; ++k) {
            const int ch = 4 * k + g;
            const f32x4 zv = ld4(Z + ch * CS + (row + 1) * RS + 4 + 4 * ff);
            if (ok && ch < C) {
                if constexpr (BFM) *reinterpret_cast<u32x2*>(zo16 + (unsigned)ch * plane) = u32x2{pk_bf16(zv[0], zv[1]), pk_bf16(zv[2], zv[3])};
                else st4(zo + (unsigned)ch * plane, zv);
            }
        }
    }
    wave_sync();
    // ---- gate, in the layout the loads arrived in (lane = channel 4k+g, row, cells 4ff..4ff+3):
    //      dL/dx'_t = G * 1[lo <= x'*life <= hi] * life (nca.py:191-194) -> gx;  dO = . * fire mask -> scratch
    {
        const f32x4 pnv = ld4(PN + (row + 1) * RS + 4 + 4 * ff);
        const f32x4 mk = ld4(MK + row * WTW + 4 * ff);
        float lf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float life = pnv[j];
            if (use_alive) life = (life != 0.0f && max3x3(A1 + (row + 1) * RS + 4 * ff + j + 4) > a.thr) ? 1.0f : 0.0f;
            lf[j] = life;
        }
        float* const go = ba.gx + (size_t)t.b * C * plane + off;
        float* const dso = reinterpret_cast<float*>(ba.doscr) + (rid0 + row) * K::DOS + 4 * ff;
        uint16_t* const dso16 = reinterpret_cast<uint16_t*>(ba.doscr) + (rid0 + row) * K::DOS + 4 * ff;
#pragma unroll
        for (int k = 0; k < CP / 4; ++k) {
            const int ch = 4 * k + g;
            const bool live = ok && ch < C;
            const f32x4 x = ST::cv4(xv[k]);
            f32x4 gxv, dov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = x[j] * lf[j];
                gxv[j] = (live && y >= a.lo && y <= a.hi) ? gv[k][j] * lf[j] : 0.0f;
                dov[j] = gxv[j] * mk[j];
            }
            if (live) st4(go + (unsigned)ch * plane, gxv);
            if constexpr (BFM) *reinterpret_cast<u32x2*>(dso16 + ch * 16) = u32x2{pk_bf16(dov[0], dov[1]), pk_bf16(dov[2], dov[3])};
            else st4(dso + ch * 16, dov);
        }
    }
    // ---- perception of the four rows, stored as the B operands kernel M will load
    fill_store(frwp, smem + K::OFF_WP, tid);
    __syncthreads();
#pragma unroll 1
    for (int n0 = 0; n0 < WTH; n0 += 2) {
        float P[2][FK::K1S];
        perceive_tile<CP, 2>(WS, Z, lane, n0, P);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if constexpr (BFM) {
                bf_s16x4* const ps = reinterpret_cast<bf_s16x4*>(ba.pscr) + (rid0 + n0 + n) * (K::KQ * 64) + lane;
#pragma unroll
                for (int q = 0; q < K::KQ; ++q) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = 4 * q + r < FK::K1S ? P[n][4 * q + r] : 0.0f;
                    ps[q * 64] = pack4(v[0], v[1], v[2], v[3]);
                }
            } else {
                f32x4* const ps = reinterpret_cast<f32x4*>(ba.pscr) + (rid0 + n0 + n) * (K::KQ * 64) + lane;
#pragma unroll
                for (int q = 0; q < K::KQ; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = 4 * q + r < FK::K1S ? P[n][4 * q + r] : 0.0f;
                    ps[q * 64] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// M: the matrix part.  4 waves / workgroup / CU (one wave per SIMD: 128 persistent accumulators), persistent over the tiles.
template <int CP, int NW = 4>
struct MCfg {
    using F = WCfg<CP>;
    static constexpr int K1S = F::K1S;
    static constexpr int KQ = F::K1S4;                      // 16-byte groups of the perception vector per lane (front kernel's scratch)
    static constexpr int DOS = 256 * F::M3T;                // scratch dO: [row tile][16 M3T channels][16 cells]
    static constexpr int MJ = (3 * CP + 15) / 16;           // 16-row tiles of the perception index
    static constexpr int M3T = F::M3T;                      // 16-row tiles of the channel index (2 for 16 < CP <= 32)
    static constexpr int OFF_W3T = F::SHARED;               // [4 m][4 M3T s][64]
    static constexpr int OFF_W1T = OFF_W3T + 4 * 4 * M3T * 64;    // [MJ][16 s][64]
    static constexpr int SHARED = OFF_W1T + MJ * 16 * 64;
    // transposition buffer: [16 cells][TBW f32 rows] (rows: the two factors of one weight-gradient product: at most
    // max(16 M3T + 64, 128, 64 + 3 CP)); TBW % 32 == 20 keeps the 16-byte writes of a cell row conflict-free.  The same area
    // stages dL/dperception on the way out ([3C rows][NT x 16 cells], row stride 36).
    static constexpr int TBW = 64 + 3 * CP <= TBS ? TBS : 180;
    static_assert(TBW % 32 == 20 && TBW >= 64 + 3 * CP && TBW >= 16 * M3T + 64 && TBW >= 128, "transposition buffer rows");
    static constexpr int PW_F32 = 16 * TBW > 16 * MJ * 36 ? 16 * TBW : 16 * MJ * 36;
    static constexpr int PW = NW == 8 ? 16 * TBH / 2 : PW_F32;   // (NW = 8 is bf16-only: x 144 bf16)
    static constexpr int OFF_DB = SHARED + NW * PW;         // NW = 8: bias-gradient sums, [wave][32 values][64 lanes], accumulated with LDS adds
    static constexpr int DB = NW == 8 ? NW * 32 * 64 : 0;
    // the flush stages four partial slabs in their tile-major form, one section at a time (SlabTM)
    static constexpr int SLABS = SlabTM<MJ, M3T>::stage(NW);
    static constexpr int MERGE = 0;   // (the pairs' accumulators meet in the flush's sum: no separate merge)
    static constexpr int LDS_A = (SHARED + NW * PW + DB) > SLABS ? (SHARED + NW * PW + DB) : SLABS;
    static constexpr int LDS_FLOATS = LDS_A > MERGE ? LDS_A : MERGE;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(CP <= 16 || NW == 4, "wide channel counts: fp32 products, one wave per SIMD");
};

// NW = 4: one wave per SIMD, two 16-cell rows per pass (NT = 2) -- the one-launch kernel's arrangement.  NW = 8 (the bf16-MFMA
// default): TWO waves per SIMD; the waves 2p and 2p+1 split a 4 x 16 tile into its upper and lower two rows and walk them one
// row per pass (NT = 1), which with the ReLU gates taken from the stored bf16 activations instead of kept f32 pre-activations
// fits 256 registers beside the 128 accumulators; their partial accumulators are merged through LDS before the flush.
template <int CP, typename ST = StF32, bool BFM = false, int NW = 4, int NTW = 2>
__global__ __launch_bounds__(64 * NW, 1) void cond_step_bwd_mlp_kernel(const NcaCondBwdArgs ba) {
    using K = MCfg<CP, NW>;
    using FK = WCfg<CP>;
    constexpr int NT = NW == 8 ? 1 : NTW;              // 16-cell rows per pass
    constexpr int NPASS = (NW == 8 ? 2 : WTH) / NT;    // passes of a wave over its rows of a 4 x 16 tile
    constexpr int kThr = 64 * NW;
    constexpr int DPS = NT == 2 ? 36 : 20;   // row stride of the dL/dperception staging in TB ([3C rows][NT x 16 cells])
    static_assert(NW == 4 || (NW == 8 && BFM), "two waves per SIMD: bf16-MFMA form only");
    const NcaCondArgs& a = ba.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, lane_w = lane;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tslot = NW == 8 ? wave >> 1 : wave;   // which 4 x 16 tile of the super-tile;  NW == 8: wave & 1 = its upper / lower two rows
    const int C = a.C, H = a.H, W = a.W, hid = a.hidden, K1 = 3 * C;
    const unsigned plane = (unsigned)(H * W);
    const int g = lane >> 4, ci = lane & 15;

#if defined(NCA_STAMPS)
    unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last, ph_real0, ph_t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_real0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_last)::"memory");
    ph_t0 = ph_last;
#endif
    // ---- A-operand images.  Index maps: element idx of an image <- offset into the weight tensor (or -1: zero) -----------
    auto map_w1 = [&](int idx) -> long {          // forward W1: [4 m][K1S s][64]
        const int l = idx & 63, s = (idx >> 6) % FK::K1S, m = (idx >> 6) / FK::K1S;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int ch = 4 * (s / 3) + gg, f = s % 3;
        return (ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;
    };
    auto map_w2 = [&](int idx) -> long {          // forward W2: [4 m2][16 s][64]
        const int l = idx & 63, s = (idx >> 6) % 16, m = (idx >> 6) / 16;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int k = 16 * (s >> 2) + 4 * gg + (s & 3);
        return (o < hid && k < hid) ? (long)o * hid + k : -1;
    };
    auto map_w3t = [&](int idx) -> long {         // W3^T: lane (gg,i) of (m, s) holds W3[ch = 16(s/4)+4gg+s%4][h2 = 16m+i]
        const int l = idx & 63, s = (idx >> 6) % (4 * K::M3T), m = (idx >> 6) / (4 * K::M3T);
        const int ch = 16 * (s >> 2) + 4 * (l >> 4) + (s & 3), h2 = 16 * m + (l & 15);
        return (ch < C && h2 < hid) ? (long)ch * hid + h2 : -1;
    };
    auto map_w1t = [&](int idx) -> long {         // W1^T: lane (gg,i) of (mj, s) holds W1[h1 = 16(s/4)+4gg+s%4][j = 16mj+i]
        const int l = idx & 63, s = (idx >> 6) % 16, mj = (idx >> 6) / 16;
        const int h1 = 16 * (s >> 2) + 4 * (l >> 4) + (s & 3), j = 16 * mj + (l & 15);
        return (h1 < hid && j < K1) ? (long)h1 * K1 + j : -1;
    };
    // ba.opmode == 2: the image (operands + biases) was built once for this backward pass (a one-workgroup launch with opmode == 1) and
    // is copied from memory with 16-byte loads; the gathers below cost ~13 K cycles per launch (6 us of a 70 us kernel)
    constexpr int IMG = BFM ? FK::OFF_B2 + FK::HID : K::SHARED;   // floats of LDS the prologue fills
    static_assert(IMG % 4 == 0 && IMG * 4 <= (int)kNcaCondBwdOpimgBytes, "operand image fits its workspace slot");
    const bool import_img = ba.opmode == 2;
    if (import_img) {
        for (int i = 4 * tid; i < IMG; i += 4 * kThr) st4(smem + i, ld4(ba.opimg + i));
        __syncthreads();
    }
    if (!import_img) {
        FillRegs<FK::HID, kThr> fr2;
        FillRegs<FK::HID, kThr> fr3;
        fill_load(fr2, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        fill_load(fr3, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        if constexpr (!BFM) {
            // f32 images, two-phase: all gathers are requested before the first is waited for (one cold round trip instead of ten)
            FillRegs<4 * FK::K1S * 64, kThr> fr0;
            FillRegs<4 * 16 * 64, kThr> fr1;
            FillRegs<4 * 4 * K::M3T * 64, kThr> fr5;
            FillRegs<K::MJ * 16 * 64, kThr> fr6;
            fill_load(fr0, a.w1, tid, map_w1);
            fill_load(fr1, a.w2, tid, map_w2);
            fill_load(fr5, a.w3, tid, map_w3t);
            fill_load(fr6, a.w1, tid, map_w1t);
            fill_store(fr0, smem + FK::OFF_W1, tid);
            fill_store(fr1, smem + FK::OFF_W2, tid);
            fill_store(fr5, smem + K::OFF_W3T, tid);
            fill_store(fr6, smem + K::OFF_W1T, tid);
        }
        fill_store(fr2, smem + FK::OFF_B1, tid);
        fill_store(fr3, smem + FK::OFF_B2, tid);
    }
    if constexpr (!BFM) {
        if (!import_img) __syncthreads();
    }

    const float* const W1L = smem + FK::OFF_W1;
    const float* const W2L = smem + FK::OFF_W2;
    const float* const B1L = smem + FK::OFF_B1;
    const float* const B2L = smem + FK::OFF_B2;
    const float* const W3T = smem + K::OFF_W3T;
    const float* const W1T = smem + K::OFF_W1T;
    float* const TB = smem + K::SHARED + wave * K::PW;   // this wave's transposition buffer (the only per-wave LDS left)

    // BFM: bf16 A operands of every product, kept as ONE packed image in LDS, [operand][lane] x 8 bytes (120 operand registers
    // per lane would not fit beside the 128 weight-gradient accumulators).  Each wave gathers its share of the operands straight
    // from the weight tensors (through the index maps of the f32 images composed with the operand order: all loads in flight
    // together, one round trip), rounds and stores them: no f32 images, one barrier.
    constexpr int KS1 = (K::K1S + 3) / 4;                 // bf16 k-steps of layer 1 (slots q = 3*c4 + f, zero padded)
    constexpr int OP_W1 = 0, OP_W2 = OP_W1 + 4 * KS1, OP_W3T = OP_W2 + 16, OP_W2T = OP_W3T + 4 * K::M3T, OP_W1T = OP_W2T + 16, OP_N = OP_W1T + 4 * K::MJ;
    static_assert(OP_N * 64 * 2 <= FK::OFF_B1, "bf16 operand image fits below the bias vectors");
    const bf_s16x4* const BW = reinterpret_cast<const bf_s16x4*>(smem) + lane;      // operand o of this lane: BW[o * 64]
    if constexpr (BFM) if (!import_img) {
        constexpr int NOP = (OP_N + NW - 1) / NW;    // this wave's share of the operands (round robin)
        float raw[NOP][4];
        const int w2t_lane0 = (ci & 3) * 64 + (ci >> 2) * 16 + 4 * g;
        auto gw = [&](const float* w, long off) -> float {   // unconditional load (clamped), zeroed afterwards: no branch per element
            const float v = w[off < 0 ? 0 : off];
            return off < 0 ? 0.0f : v;
        };
#pragma unroll
        for (int k = 0; k < NOP; ++k) {
            const int o = k * NW + wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[k][r] = 0.0f;
            if (o < OP_W2) {                           // W1 forward: (m, s)
                const int m = o / KS1, s_ = o % KS1;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * s_ + r < K::K1S) raw[k][r] = gw(a.w1, map_w1((m * K::K1S + 4 * s_ + r) * 64 + lane));
            } else if (o < OP_W3T) {                   // W2 forward: (m2, kk)
                const int m2 = (o - OP_W2) >> 2, kk = (o - OP_W2) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) raw[k][r] = gw(a.w2, map_w2((m2 * 16 + 4 * kk + r) * 64 + lane));
            } else if (o < OP_W2T) {                   // W3^T: (m, k3): channels 16 k3 + 4g + r
                const int m = (o - OP_W3T) / K::M3T, k3 = (o - OP_W3T) % K::M3T;
#pragma unroll
                for (int r = 0; r < 4; ++r) raw[k][r] = gw(a.w3, map_w3t((m * 4 * K::M3T + 4 * k3 + r) * 64 + lane));
            } else if (o < OP_W1T) {                   // W2^T: (m, mp) = W2[h2 = 16mp+4g+r][h1 = 16m+ci]: four consecutive image elements
                const int m = (o - OP_W2T) >> 2, mp = (o - OP_W2T) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) raw[k][r] = gw(a.w2, map_w2((mp * 16 + 4 * m) * 64 + w2t_lane0 + r));
            } else if (o < OP_N) {                     // W1^T: (mj, kk)
                const int mj = (o - OP_W1T) >> 2, kk = (o - OP_W1T) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) raw[k][r] = gw(a.w1, map_w1t((mj * 16 + 4 * kk + r) * 64 + lane));
            }
        }
#pragma unroll
        for (int k = 0; k < NOP; ++k) {
            const int o = k * NW + wave;
            if (o < OP_N) *(reinterpret_cast<bf_s16x4*>(smem) + o * 64 + lane) = pack4(raw[k][0], raw[k][1], raw[k][2], raw[k][3]);
        }
        __syncthreads();
    }
    if (ba.opmode == 1) {   // export launch: the image is complete (barriers above) -- write it out, no tile work
        for (int i = 4 * tid; i < IMG; i += 4 * kThr) st4(ba.opimg + i, ld4(smem + i));
        return;
    }
    // persistent weight-gradient accumulators (D = A * B^T with the cell axis as K)
    f32x4 aW1[4][K::MJ], aW2[4][4], aW3[K::M3T][4];
    // bias-gradient sums: 32 registers per lane -- or, with two waves per SIMD (no registers to spare), 32 x 64 floats of LDS per
    // wave, [which][m][lane][4]: one 16-byte read-modify-write per gated tile (lane-private addresses; ds_add_f32 was tried: the
    // LDS's float atomics run a lane at a time and made the kernel five times slower)
    float db1[4][4], db2[4][4];
    float* const DBL = smem + K::OFF_DB + wave * (32 * 64) + lane * 4;
    if constexpr (NW == 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) st4(DBL + i * 256, f32x4{0.f, 0.f, 0.f, 0.f});
    }
    auto db_add4 = [&](int which, int m, const f32x4& d) {
        if constexpr (NW == 8) {
            float* const p_ = DBL + (which * 4 + m) * 256;
            st4(p_, ld4(p_) + d);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (which == 0) db1[m][r] += d[r];
                else db2[m][r] += d[r];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3) aW3[m3][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) { aW2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; db1[i][j] = 0.f; db2[i][j] = 0.f; }
#pragma unroll
        for (int j = 0; j < K::MJ; ++j) aW1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    NCA_BPHASE(10);  // start-up: weight images, accumulators
    // ---- tile walk: super-tiles of 16 x 16 (4 waves stacked vertically) ------------------------------------
    constexpr int BSTH = 16, BSTW = 16;
    const int st_x = (W + BSTW - 1) / BSTW, st_y = (H + BSTH - 1) / BSTH;
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    const bool use_alive = a.alive_ch >= 0;
    // one pass's scratch operands, requested a pass ahead (raw: 16-byte groups of P / 8-byte groups of bf16 P in the low half)
    f32x4 nP[NT][K::KQ];
    float ndO[NT][4 * K::M3T];
    long pre_rid = -1;
    auto fetch = [&](long r_) {
        const int gq = (lane_w >> 4) & 3, cq = lane_w & 15;
        if constexpr (BFM) {
            const u32x2* const ps = reinterpret_cast<const u32x2*>(ba.pscr) + (size_t)r_ * (K::KQ * 64) + lane_w;
            const uint16_t* const ds = reinterpret_cast<const uint16_t*>(ba.doscr) + (size_t)r_ * K::DOS + 64 * gq + cq;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int q = 0; q < K::KQ; ++q) {
                    const u32x2 v = ps[n * (K::KQ * 64) + q * 64];
                    nP[n][q] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), 0.0f, 0.0f};
                }
#pragma unroll
                for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // channel 16 m3 + 4 gq + r (rows >= CP of the scratch are never written)
                        if (CP % 16 == 0 || 16 * m3 + 4 * gq + r < CP) ndO[n][4 * m3 + r] = __uint_as_float((unsigned)ds[n * K::DOS + 256 * m3 + 16 * r] << 16);
                        else ndO[n][4 * m3 + r] = 0.0f;
                    }
            }
        } else {
            const f32x4* const ps = reinterpret_cast<const f32x4*>(ba.pscr) + (size_t)r_ * (K::KQ * 64) + lane_w;
            const float* const ds = reinterpret_cast<const float*>(ba.doscr) + (size_t)r_ * K::DOS + 64 * gq + cq;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int q = 0; q < K::KQ; ++q) nP[n][q] = ps[n * (K::KQ * 64) + q * 64];
#pragma unroll
                for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {   // channel 16 m3 + 4 gq + r (rows >= CP of the scratch are never written)
                        if (CP % 16 == 0 || 16 * m3 + 4 * gq + r < CP) ndO[n][4 * m3 + r] = ds[n * K::DOS + 256 * m3 + 16 * r];
                        else ndO[n][4 * m3 + r] = 0.0f;
                    }
            }
        }
        pre_rid = r_;
    };
    // Walk items: super-tiles -- or, on grids with fewer super-tiles than half the workgroup slots (ba.msplit), (super-tile, pass)
    // pairs, so that twice as many workgroups share the work (the launch is otherwise one super-tile per workgroup on half the chip,
    // behind a fixed start-up + flush of ~20 us).  Item i = super-tile i >> msplit, the first or second half of the passes by i & 1.
    const int msplit = ba.msplit;
    for (NcaTileWalk tw = nca_tile_walk((a.B * st_x * st_y) << msplit); tw.t < tw.end; tw.t += tw.stride) {
        int lane = lane_w;   // opaque per tile: lane-derived offsets are recomputed where used, not hoisted out of the tile loop and spilled
        asm volatile("" : "+v"(lane));
        const int g = (lane >> 4) & 3, ci = lane & 15;
        WTile t;
        const int sti = tw.t >> msplit;                    // super-tile of this item
        t.b = sti / (st_x * st_y);
        t.ty0 = ((sti / st_x) % st_y) * BSTH + tslot * WTH;
        t.tx0 = (sti % st_x) * BSTW;
        if (t.ty0 >= H || t.tx0 >= W) continue;
        const int ty0 = t.ty0, tx0 = t.tx0;
        NCA_BPHASE(0);   // loop overhead / previous tile's tail
        const int nbase = NW == 8 ? (wave & 1) * 2 : 0;   // first row of this wave inside the tile
        const size_t rid0 = ((size_t)sti * kBwdWaves + tslot) * WTH + nbase;   // 16-cell row tiles of the front kernel's scratch
        const int pass0 = msplit ? (tw.t & 1) * (NPASS / 2) : 0, pass1 = msplit ? pass0 + NPASS / 2 : NPASS;
#pragma unroll 1
        for (int pass = pass0; pass < pass1; ++pass) {
            const int n0 = nbase + pass * NT;
            // ---- forward recompute: P, h1, h2 kept in registers ------------------------------------------
            // ---- this pass's two 16-cell rows from the front kernel's scratch: the perception vector in B-operand order
            //      ([row tile][4-slot group][lane] x 16 B, or x 8 B of bf16) and the gated gradient dO = dL/dx' * fire mask
            //      ([row tile][channel][cell]).  They were requested a pass ago (fetch below); the request for the NEXT
            //      pass -- rows 2, 3 of this tile, or rows 0, 1 of the wave's next tile -- goes out before this pass's MFMAs.
            const long rid = (long)(rid0 + pass * NT);
            if (pre_rid != rid) fetch(rid);   // first tile of the wave, or the tile before this one was outside the image
            float P[NT][4 * K::KQ];
            float dOin[NT][4 * K::M3T];
            bf_s16x4 pbin[NT][K::KQ];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
                for (int q = 0; q < K::KQ; ++q) {
                    if constexpr (BFM) pbin[n][q] = __builtin_bit_cast(bf_s16x4, u32x2{__float_as_uint(nP[n][q][0]), __float_as_uint(nP[n][q][1])});
                    else { P[n][4 * q] = nP[n][q][0]; P[n][4 * q + 1] = nP[n][q][1]; P[n][4 * q + 2] = nP[n][q][2]; P[n][4 * q + 3] = nP[n][q][3]; }
                }
#pragma unroll
                for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dOin[n][4 * m3 + r] = 16 * m3 + 4 * g + r < CP ? ndO[n][4 * m3 + r] : 0.0f;
            }
            {
                long nrid = rid + NT;
                bool has = true;
                if (pass == pass1 - 1) {
                    const int tn = tw.t + tw.stride;
                    has = tn < tw.end;
                    nrid = (long)(((size_t)(tn >> msplit) * kBwdWaves + tslot) * WTH + nbase + (msplit ? (tn & 1) * (NPASS / 2) * NT : 0));
                }
                if (has) fetch(nrid);
                else pre_rid = -1;
            }
            __builtin_amdgcn_sched_barrier(0);   // the requests stay here
            NCA_BPHASE(4);   // scratch loads
            f32x4 dp[K::MJ][NT];
            if constexpr (BFM) {
                short* const tb16 = reinterpret_cast<short*>(TB);
                // ---- forward recompute on bf16 MFMA (rounding points of the bf16 forward kernel) ---------------------------
                bf_s16x4 pb[NT][KS1];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < KS1; ++s_) pb[n][s_] = pbin[n][s_];
                f32x4 h1f[4][NT], h2f[4][NT];              // pre-ReLU accumulators (gates), f32
                bf_s16x4 h1b[4][NT], h2b[4][NT];           // ReLU'd, bf16: operands of layer 2 / 3 and of the weight gradients
#pragma unroll
                for (int m2 = 0; m2 < 4; ++m2) {
                    const f32x4 b = ld4(B2L + 16 * m2 + 4 * g);
#pragma unroll
                    for (int n = 0; n < NT; ++n) h2f[m2][n] = b;
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 b = ld4(B1L + 16 * m + 4 * g);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 a1 = b;
#pragma unroll
                        for (int s_ = 0; s_ < KS1; ++s_) a1 = mfma_bf16(BW[(OP_W1 + m * KS1 + s_) * 64], pb[n][s_], a1);
                        h1f[m][n] = a1;
                        h1b[m][n] = pack4_relu(a1);
#pragma unroll
                        for (int m2 = 0; m2 < 4; ++m2) h2f[m2][n] = mfma_bf16(BW[(OP_W2 + m2 * 4 + m) * 64], h1b[m][n], h2f[m2][n]);
                    }
                }
#pragma unroll
                for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                    for (int n = 0; n < NT; ++n) h2b[m2][n] = pack4_relu(h2f[m2][n]);
                NCA_BPHASE(5);   // forward recompute
                bf_s16x4 dOb[NT][K::M3T];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3)   // (bf16 values: exact)
                        dOb[n][m3] = pack4(dOin[n][4 * m3], dOin[n][4 * m3 + 1], dOin[n][4 * m3 + 2], dOin[n][4 * m3 + 3]);
                bf_s16x4 d2b[4][NT], d1b[4][NT];
                // ---- layer 3: dW3 += dO x h2 (cells as K);  d2 = (W3^T dO) * 1[h2 > 0] ---------------------------------------
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wave_sync();
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3) tb_write(tb16, m3, lane, dOb[n][m3]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) tb_write(tb16, K::M3T + m, lane, h2b[m][n]);
                    wave_sync();
                    bf_s16x4 tbv[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tbv[nb] = tb_tr_read(tb16, K::M3T + nb, lane);
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3) {
                        const bf_s16x4 ta = tb_tr_read(tb16, m3, lane);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) aW3[m3][nb] = mfma_bf16(ta, tbv[nb], aW3[m3][nb]);
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int m3 = 0; m3 < K::M3T; ++m3) d = mfma_bf16(BW[(OP_W3T + m * K::M3T + m3) * 64], dOb[n][m3], d);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { d[r] = (NW == 8 ? bf_nz(h2b[m][n], r) : h2f[m][n][r] > 0.0f) ? d[r] : 0.0f; }
                        db_add4(1, m, d);
                        d2b[m][n] = pack4(d[0], d[1], d[2], d[3]);
                    }
                NCA_BPHASE(6);   // layer 3
                // ---- layer 2: dW2 += d2 x h1;  d1 = (W2^T d2) * 1[h1 > 0] ----------------------------------------------------
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 4; ++m) { tb_write(tb16, m, lane, d2b[m][n]); tb_write(tb16, 4 + m, lane, h1b[m][n]); }
                    wave_sync();
                    bf_s16x4 tbv[4];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) tbv[nb] = tb_tr_read(tb16, 4 + nb, lane);
#pragma unroll
                    for (int ma = 0; ma < 4; ++ma) {
                        const bf_s16x4 ta = tb_tr_read(tb16, ma, lane);
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) aW2[ma][nb] = mfma_bf16(ta, tbv[nb], aW2[ma][nb]);
                    }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int mp = 0; mp < 4; ++mp) d = mfma_bf16(BW[(OP_W2T + m * 4 + mp) * 64], d2b[mp][n], d);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { d[r] = (NW == 8 ? bf_nz(h1b[m][n], r) : h1f[m][n][r] > 0.0f) ? d[r] : 0.0f; }
                        db_add4(0, m, d);
                        d1b[m][n] = pack4(d[0], d[1], d[2], d[3]);
                    }
                NCA_BPHASE(7);   // layer 2
                // ---- layer 1: dW1 += d1 x P (P columns in slot order: column 12 g' + q, un-permuted at the slab flush);  dp = W1^T d1
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wave_sync();
#pragma unroll
                    for (int m = 0; m < 4; ++m) tb_write(tb16, m, lane, d1b[m][n]);
#pragma unroll
                    for (int s_ = 0; s_ < KS1; ++s_) tb_write_chunk(tb16, 16 + KS1 * g + s_, lane, pb[n][s_]);   // columns 4 KS1 g + 4s .. of tiles 4..4+MJ-1
                    wave_sync();
                    bf_s16x4 tbv[K::MJ];
#pragma unroll
                    for (int nb = 0; nb < K::MJ; ++nb) tbv[nb] = tb_tr_read(tb16, 4 + nb, lane);
#pragma unroll
                    for (int ma = 0; ma < 4; ++ma) {
                        const bf_s16x4 ta = tb_tr_read(tb16, ma, lane);
#pragma unroll
                        for (int nb = 0; nb < K::MJ; ++nb) aW1[ma][nb] = mfma_bf16(ta, tbv[nb], aW1[ma][nb]);
                    }
                }
#pragma unroll
                for (int mj = 0; mj < K::MJ; ++mj)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) d = mfma_bf16(BW[(OP_W1T + mj * 4 + kk) * 64], d1b[kk][n], d);
                        dp[mj][n] = d;
                    }
                NCA_BPHASE(8);   // layer 1
            } else {
            f32x4 h1[4][NT], h2[4][NT];
            constexpr int G1 = 3, NG1 = K::K1S / G1;   // layer-1 k-steps in groups of 3 (K1S = 9 or 12)
            static_assert(K::K1S % G1 == 0, "layer-1 operand groups");
            piped<4 * NG1>(
                [&](int i) {
                    const int m = i / NG1, sg = i % NG1;
                    OpN<G1> o;
#pragma unroll
                    for (int q = 0; q < G1; ++q) o.v[q] = W1L[(m * K::K1S + G1 * sg + q) * 64 + lane];
                    return o;
                },
                [&](int i, const OpN<G1>& o) {
                    const int m = i / NG1, sg = i % NG1;
                    if (sg == 0) {
                        const f32x4 bias = ld4(B1L + 16 * m + 4 * g);
#pragma unroll
                        for (int n = 0; n < NT; ++n) h1[m][n] = bias;
                    }
#pragma unroll
                    for (int q = 0; q < G1; ++q)
#pragma unroll
                        for (int n = 0; n < NT; ++n) h1[m][n] = nca_mfma(o.v[q], P[n][G1 * sg + q], h1[m][n]);
                    if (sg == NG1 - 1) {
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) h1[m][n][r] = relu(h1[m][n][r]);
                    }
                });
            piped<16>(
                [&](int i) {
                    const int m2 = i >> 2, m = i & 3;
                    OpN<4> o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.v[r] = W2L[(m2 * 16 + 4 * m + r) * 64 + lane];
                    return o;
                },
                [&](int i, const OpN<4>& o) {
                    const int m2 = i >> 2, m = i & 3;
                    if (m == 0) {
                        const f32x4 bias = ld4(B2L + 16 * m2 + 4 * g);
#pragma unroll
                        for (int n = 0; n < NT; ++n) h2[m2][n] = bias;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n) h2[m2][n] = nca_mfma(o.v[r], h1[m][n][r], h2[m2][n]);
                    if (m == 3) {
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) h2[m2][n][r] = relu(h2[m2][n][r]);
                    }
                });
            NCA_BPHASE(5);   // forward recompute
            float dO[NT][4 * K::M3T];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4 * K::M3T; ++r) dO[n][r] = dOin[n][r];
            // Backward data path and weight gradients, layer by layer from the output: each layer's weight-gradient product
            // is issued as soon as its two factors exist, so h2 dies after layer 3, d2 and h1 after layer 2, P and d1 after
            // layer 1 (all five tiles alive at once through a separate weight-gradient phase cost 36 spilled registers).
            // Weight gradients go per 16-cell tile through the cell-major buffer TB[cell][row]: lane (g,ci) writes its 4
            // accumulator rows of a tile with ONE 16-byte store; operand fragments A[i][k=g] = TB[4s+g][rowA+i],
            // B[k=g][j] = TB[4s+g][rowB+j] are 16 consecutive floats per lane group.
            constexpr int TBS = K::TBW;                          // (row pitch of this instantiation's transposition buffer)
            float* const tw_ = TB + ci * TBS + 4 * g;            // this lane's cell row, accumulator-row offset
            const float* const tr_ = TB + g * TBS + ci;           // operand reads: cell 4s+g -> + 4*s*TBS
            f32x4 d2[4][NT], d1[4][NT];
            // ---- layer 3: dW3 = dO (rows 0..16 M3T-1) x h2 (the next 64 rows);  d2 = (W3^T dO) * 1[h2 > 0] -------------------
            constexpr int M3 = K::M3T;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wave_sync();  // TB free (gradient tile consumed above / previous products done)
#pragma unroll
                for (int m3 = 0; m3 < M3; ++m3) st4(tw_ + 16 * m3, f32x4{dO[n][4 * m3], dO[n][4 * m3 + 1], dO[n][4 * m3 + 2], dO[n][4 * m3 + 3]});
#pragma unroll
                for (int m = 0; m < 4; ++m) st4(tw_ + 16 * M3 + 16 * m, h2[m][n]);
                wave_sync();
                piped<4>(
                    [&](int s_) {
                        OpN<M3 + 4> o;
#pragma unroll
                        for (int m3 = 0; m3 < M3; ++m3) o.v[m3] = tr_[4 * s_ * TBS + 16 * m3];
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) o.v[M3 + nb] = tr_[4 * s_ * TBS + 16 * M3 + 16 * nb];
                        return o;
                    },
                    [&](int, const OpN<M3 + 4>& o) {
#pragma unroll
                        for (int m3 = 0; m3 < M3; ++m3)
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) aW3[m3][nb] = nca_mfma(o.v[m3], o.v[M3 + nb], aW3[m3][nb]);
                    });
            }
            piped<4>(
                [&](int m) {
                    OpN<4 * M3> o;
#pragma unroll
                    for (int s_ = 0; s_ < 4 * M3; ++s_) o.v[s_] = W3T[(m * 4 * M3 + s_) * 64 + lane];
                    return o;
                },
                [&](int m, const OpN<4 * M3>& o) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) d2[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s_ = 0; s_ < 4 * M3; ++s_)
#pragma unroll
                        for (int n = 0; n < NT; ++n) d2[m][n] = nca_mfma(o.v[s_], dO[n][s_], d2[m][n]);
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) d2[m][n][r] = gate_pos(h2[m][n][r], d2[m][n][r]);
                });
            NCA_BPHASE(6);   // layer 3
            // ---- layer 2: dW2 = d2 (rows 0..63) x h1 (rows 64..127);  d1 = (W2^T d2) * 1[h1 > 0] ---------------------------
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wave_sync();
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    st4(tw_ + 16 * m, d2[m][n]);
                    st4(tw_ + 64 + 16 * m, h1[m][n]);
                }
                wave_sync();
                piped<4>(
                    [&](int s_) {
                        OpN<8> o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            o.v[k] = tr_[4 * s_ * TBS + 16 * k];
                            o.v[4 + k] = tr_[4 * s_ * TBS + 64 + 16 * k];
                        }
                        return o;
                    },
                    [&](int, const OpN<8>& o) {
#pragma unroll
                        for (int ma = 0; ma < 4; ++ma)
#pragma unroll
                            for (int nb = 0; nb < 4; ++nb) aW2[ma][nb] = nca_mfma(o.v[ma], o.v[4 + nb], aW2[ma][nb]);
                    });
            }
            const int w2t_lane = (ci & 3) * 64 + (ci >> 2) * 16 + 4 * g;  // transposed read of the forward W2 image
            piped<16>(
                [&](int i) {   // W2[h2 = 16mp+4g+r][h1 = 16m+ci], r = 0..3: one 16-byte read
                    const int m = i >> 2, mp = i & 3;
                    return ld4(W2L + (mp * 16 + 4 * m) * 64 + w2t_lane);
                },
                [&](int i, const f32x4& o) {
                    const int m = i >> 2, mp = i & 3;
                    if (mp == 0) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) d1[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n) d1[m][n] = nca_mfma(o[r], d2[mp][n][r], d1[m][n]);
                    if (mp == 3) {
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 4; ++r) d1[m][n][r] = gate_pos(h1[m][n][r], d1[m][n][r]);
                    }
                });
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int n = 0; n < NT; ++n) { db1[m][r] += d1[m][n][r]; db2[m][r] += d2[m][n][r]; }
            NCA_BPHASE(7);   // layer 2
            // ---- layer 1: dW1 = d1 (rows 0..63) x P (rows 64.., natural perception index j = 3c+f);  dp = W1^T d1 ---------
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wave_sync();
#pragma unroll
                for (int m = 0; m < 4; ++m) st4(tw_ + 16 * m, d1[m][n]);
#pragma unroll
                for (int c4 = 0; c4 < CP / 4; ++c4)
#pragma unroll
                    for (int f = 0; f < 3; ++f) TB[ci * TBS + 64 + 3 * (4 * c4 + g) + f] = P[n][3 * c4 + f];
                wave_sync();
                piped<4>(
                    [&](int s_) {
                        OpN<4 + K::MJ> o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) o.v[k] = tr_[4 * s_ * TBS + 16 * k];
#pragma unroll
                        for (int nb = 0; nb < K::MJ; ++nb) o.v[4 + nb] = tr_[4 * s_ * TBS + 64 + 16 * nb];
                        return o;
                    },
                    [&](int, const OpN<4 + K::MJ>& o) {
#pragma unroll
                        for (int ma = 0; ma < 4; ++ma)
#pragma unroll
                            for (int nb = 0; nb < K::MJ; ++nb) aW1[ma][nb] = nca_mfma(o.v[ma], o.v[4 + nb], aW1[ma][nb]);
                    });
            }
            piped<4 * K::MJ>(
                [&](int i) {
                    const int mj = i >> 2, mp = i & 3;
                    OpN<4> o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.v[r] = W1T[(mj * 16 + 4 * mp + r) * 64 + lane];
                    return o;
                },
                [&](int i, const OpN<4>& o) {
                    const int mj = i >> 2, mp = i & 3;
                    if (mp == 0) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) dp[mj][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int n = 0; n < NT; ++n) dp[mj][n] = nca_mfma(o.v[r], d1[mp][n][r], dp[mj][n]);
                });
            NCA_BPHASE(8);   // layer 1
            }
            // ---- dL/dperception out: [j][2 rows][16] via TB, 16-byte stores -----------------------------------
            wave_sync();
#pragma unroll
            for (int mj = 0; mj < K::MJ; ++mj)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) TB[(16 * mj + 4 * g + r) * DPS + n * 16 + ci] = dp[mj][n][r];
            wave_sync();
            {
                const int rr = NT == 2 ? (lane >> 2) & 1 : 0, ff = lane & 3, jl = NT == 2 ? lane >> 3 : lane >> 2;  // 8 (NT = 1: 16) rows of j per instruction
                const int gy = ty0 + n0 + rr, gx = tx0 + 4 * ff;
                const bool ok = gy < H && gx + 3 < W;
                float* const po = ba.dP + (size_t)t.b * 3 * C * plane + (ok ? (unsigned)(gy * W + gx) : 0u);
                uint16_t* const po16 = reinterpret_cast<uint16_t*>(ba.dP) + (size_t)t.b * 3 * C * plane + (ok ? (unsigned)(gy * W + gx) : 0u);
#pragma unroll
                for (int k = 0; k < (NT == 2 ? 2 : 1) * K::MJ; ++k) {
                    const int j = (NT == 2 ? 8 : 16) * k + jl;
                    const f32x4 v = ld4(TB + j * DPS + rr * 16 + 4 * ff);
                    if (ok && j < K1) {
                        if constexpr (BFM) *reinterpret_cast<u32x2*>(po16 + (unsigned)j * plane) = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
                        else st4(po + (unsigned)j * plane, v);
                    }
                }
            }
            NCA_BPHASE(9);   // dP out
        }
        wave_sync();
    }

#if defined(NCA_STAMPS)
    if (a.dbg && lane == 0) {
        for (int i = 0; i < 12; ++i) a.dbg[(size_t)(blockIdx.x * kBwdWaves + wave) * 16 + i] = ph_acc[i];
        unsigned long long r1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
        a.dbg[(size_t)(blockIdx.x * kBwdWaves + wave) * 16 + 12] = r1 - ph_real0;    // 100 MHz ticks
        a.dbg[(size_t)(blockIdx.x * kBwdWaves + wave) * 16 + 13] = ph_last - ph_t0;  // shader-clock ticks over the same span
    }
#endif
    // ---- weight-gradient partials: the four waves' accumulators are summed through LDS (tiles and weight images are dead
    //      by now) and added to the WORKGROUP's slab with coalesced accesses; fixed summation order (deterministic).
    using TM = SlabTM<K::MJ, K::M3T>;
    static_assert(TM::stage(NW) <= K::LDS_FLOATS, "slab staging fits the LDS carve");
    float* const slab = ba.slabs + (size_t)blockIdx.x * TM::SF;
    if constexpr (NW == 8) {
        // the bias sums leave their LDS accumulators (lane-private addresses: no barrier needed before, the flush's first barrier
        // separates these reads from the staging that reuses the carve).  The two waves of a tile are partials 2p and 2p + 1: the flush
        // adds them first (the association the pair merge through LDS had, which this replaces: 44 16-byte stores + loads per lane and
        // five workgroup barriers per launch).
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) { db1[m][r] = DBL[m * 256 + r]; db2[m][r] = DBL[(4 + m) * 256 + r]; }
    }
    slab_flush_tm<TM, kThr, NW>(smem, slab, tid, lane, wave, aW1, aW2, aW3, db1, db2);
#if defined(NCA_STAMPS)
    NCA_BPHASE(11);  // slab flush
    if (a.dbg && lane == 0) a.dbg[(size_t)(blockIdx.x * kBwdWaves + wave) * 16 + 11] = ph_acc[11];
#endif
}


bool g_fm_nosplit = false;   // test hook (ncahip_debug_force_generic bit 4): whole super-tiles per walk item on every grid

template <int CP, typename ST, bool BFM, int NTW = 2>
hipError_t launch_fm(const NcaCondBwdArgs& ba_in, hipStream_t st) {
    NcaCondBwdArgs ba = ba_in;
    ba.f.err = nca_error_word_device();
    const NcaCondArgs& a = ba.f;
    const int nst = a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    if (ba.opmode != 1) {
        using KF = FrontCfg<CP>;
        const size_t lds = (size_t)KF::LDS_FLOATS * sizeof(float);
        auto go = [&](auto kern, NcaLdsAttr& attr) -> hipError_t {
            if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(8 * ((nst + 7) / 8)), dim3(kFrontThreads), lds, st, ba);
            return hipGetLastError();
        };
        static NcaLdsAttr attr_x, attr_g;   // per instantiation; keyed by device inside
        const hipError_t e = a.C == CP ? go(cond_step_bwd_front_kernel<CP, ST, BFM, true>, attr_x)
                                       : go(cond_step_bwd_front_kernel<CP, ST, BFM, false>, attr_g);
        if (e != hipSuccess) return e;
    }
    constexpr int NW = (BFM && CP <= 16) ? 8 : 4;   // bf16 MFMA: two waves per SIMD (CP <= 16: 128 accumulators; wider: 160, one wave per SIMD)
    using KM = MCfg<CP, NW>;
    auto kern = cond_step_bwd_mlp_kernel<CP, ST, BFM, NW, NTW>;
    const size_t lds = (size_t)KM::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    ba.msplit = (!g_fm_nosplit && 2 * nst <= ba.nslab) ? 1 : 0;           // small grids: (super-tile, pass) items
    const int items = nst << ba.msplit;
    const int grid = ba.opmode == 1 ? 1 : (items < ba.nslab ? items : ba.nslab);   // one slab per workgroup (export launch: one workgroup)
#if defined(NCA_STAMPS)
    ba.f.dbg = nca_debug_stamp_ptr();
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, st, ba);
    return hipGetLastError();
}

}  // namespace

void nca_set_bwd_fm_nosplit(bool on) { g_fm_nosplit = on; }

// channel padding of the instantiation that serves C channels
static int fm_cp(int C) { return C <= 12 ? 12 : (C <= 16 ? 16 : (C <= 20 ? 20 : (C <= 24 ? 24 : 32))); }
// bytes of the two scratch areas for a B x C x H x W grid (fp32 sizes: the bf16 forms use half): per 16-cell row tile,
// [K1S4 16-byte groups][64 lanes] of perception values and [16 M3T channels][16 cells] of gated gradient
size_t nca_cond_bwd_fm_pscr_bytes(int B, int C, int H, int W) {
    const int kq = (3 * fm_cp(C) / 4 + 3) / 4;
    return (size_t)B * ((W + 15) / 16) * ((H + 15) / 16) * 16 * (size_t)kq * 64 * 16;
}
size_t nca_cond_bwd_fm_doscr_bytes(int B, int C, int H, int W) {
    return (size_t)B * ((W + 15) / 16) * ((H + 15) / 16) * 16 * (C <= 16 ? 256 : 512) * 4;
}

// front + matrix kernels (kernel B is launched by the caller).  mode 0 = f32 history, 1 = bf16 history / exact-f32 products,
// 2 = bf16 history with the products on bf16 MFMA.
hipError_t nca_launch_cond_step_bwd_fm(const NcaCondBwdArgs& ba, hipStream_t st, int mode) {
    if (ba.f.C > 32 || !ba.pscr || !ba.doscr) return hipErrorInvalidValue;
    if (ba.f.C > 16) {   // wide channel counts (the reference's default model is C = 20): fp32 products only; bf16 history up to C = 20
        if (mode == 1 && ba.f.C <= 20) return launch_fm<20, StBF16, false>(ba, st);
        if (mode == 2 && ba.f.C <= 20) return launch_fm<20, StBF16, true>(ba, st);
        if (mode != 0) return hipErrorInvalidValue;
        if (ba.f.C <= 20) return launch_fm<20, StF32, false>(ba, st);
        if (ba.f.C <= 24) return launch_fm<24, StF32, false>(ba, st);
        return launch_fm<32, StF32, false, NCA_FM_NT32>(ba, st);
    }
    const bool c12 = ba.f.C <= 12;
    switch (mode) {
        case 0: return c12 ? launch_fm<12, StF32, false>(ba, st) : launch_fm<16, StF32, false>(ba, st);
        case 1: return c12 ? launch_fm<12, StBF16, false>(ba, st) : launch_fm<16, StBF16, false>(ba, st);
        case 2: return c12 ? launch_fm<12, StBF16, true>(ba, st) : launch_fm<16, StBF16, true>(ba, st);
    }
    return hipErrorInvalidValue;
}
