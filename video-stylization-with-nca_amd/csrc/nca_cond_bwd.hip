// nca_cond_bwd.hip -- backward of one ConditionedNCA step (EncoderConditioning/nca.py:181-195 under
// autograd, conditioned_trainer.py:125-132), gfx950, fp32.
//
// Recomputation design: only the pending states x'_t and the 1-byte pre masks are kept by
// the forward pass.  Per step, two launches:
//   A  cond_step_bwd_kernel   wave-private 4x16 tiles, same staging as the forward kernel (resolve s_t, pre_t,
//      z_t in LDS).  The forward MLP is recomputed on MFMA with h1/h2 kept in registers; the incoming gradient
//      is gated by life_t and the clamp pass-band (closed interval, as torch.clamp) -> dL/dx'_t; the data path
//      runs back through W3^T, W2^T, W1^T on MFMA (accumulator tile == next B operand, as in the forward);
//      weight gradients are MFMA products with the CELL axis as K: each 16-cell tile's activations are
//      transposed through an 18-float-stride LDS buffer (conflict-free operand reads) and accumulated in 128
//      persistent accumulator registers, flushed once per launch into this wave's slab (deterministic).
//      Outputs dL/dperception [B,3C,H,W], dL/dx'_t and z_t.
//   B  cond_step_bwd_stencil_kernel   HBM-bound: dL/ds_t = dL/dx'_t + depthwise-stencil^T(dL/dperception),
//      dL/dgoal += dz * pre_t, per-block partials of the perception-weight gradient.
// One wave per SIMD (the persistent accumulators need the registers); 4 waves / workgroup / CU.
#include <cstdlib>

#include "nca_cond_bwd_common.h"

namespace {

template <int CP, typename ST = StF32, bool BFM = false>
__global__ __launch_bounds__(kBwdThreads, 1) void cond_step_bwd_kernel(const NcaCondBwdArgs ba) {
    using K = BCfg<CP>;
    using FK = WCfg<CP>;
    constexpr int NT = 2;
    const NcaCondArgs& a = ba.f;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, lane_w = lane;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W, hid = a.hidden, K1 = 3 * C;
    const unsigned plane = (unsigned)(H * W);
    const int g = lane >> 4, ci = lane & 15;

#if defined(NCA_STAMPS)
    unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_last, ph_real0, ph_t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_real0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_last)::"memory");
    ph_t0 = ph_last;
#endif
    // ---- forward A-operand images (identical to the forward kernel) + transposed images --------------------
    // Two-phase: all gathers are requested before the first is waited for (one cold round trip instead of ten).
    {
        FillRegs<4 * FK::K1S * 64, kBwdThreads> fr0;
        FillRegs<4 * 16 * 64, kBwdThreads> fr1;
        FillRegs<FK::HID, kBwdThreads> fr2;
        FillRegs<FK::HID, kBwdThreads> fr3;
        FillRegs<CP * FK::WPS, kBwdThreads> fr4;
        FillRegs<4 * 4 * 64, kBwdThreads> fr5;
        FillRegs<K::MJ * 16 * 64, kBwdThreads> fr6;
        fill_load(fr0, a.w1, tid, [&](int idx) -> long {
            const int l = idx & 63, s = (idx >> 6) % FK::K1S, m = (idx >> 6) / FK::K1S;
            const int gg = l >> 4, o = 16 * m + (l & 15);
            const int ch = 4 * (s / 3) + gg, f = s % 3;
            return (ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;
        });
        fill_load(fr1, a.w2, tid, [&](int idx) -> long {
            const int l = idx & 63, s = (idx >> 6) % 16, m = (idx >> 6) / 16;
            const int gg = l >> 4, o = 16 * m + (l & 15);
            const int k = 16 * (s >> 2) + 4 * gg + (s & 3);
            return (o < hid && k < hid) ? (long)o * hid + k : -1;
        });
        fill_load(fr2, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        fill_load(fr3, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
        fill_load(fr4, a.wp, tid, [&](int idx) -> long {
            const int ch = idx / FK::WPS, j = idx % FK::WPS;
            return (ch < C && j < 27) ? (long)ch * 27 + j : -1;
        });
        // W3^T: lane (gg,i) of (m, s) holds W3[ch = 4gg+s][h2 = 16m+i]
        fill_load(fr5, a.w3, tid, [&](int idx) -> long {
            const int l = idx & 63, s = (idx >> 6) % 4, m = (idx >> 6) / 4;
            const int ch = 4 * (l >> 4) + s, h2 = 16 * m + (l & 15);
            return (ch < C && h2 < hid) ? (long)ch * hid + h2 : -1;
        });
        // W1^T: lane (gg,i) of (mj, s) holds W1[h1 = 16(s/4)+4gg+s%4][j = 16mj+i]
        fill_load(fr6, a.w1, tid, [&](int idx) -> long {
            const int l = idx & 63, s = (idx >> 6) % 16, mj = (idx >> 6) / 16;
            const int h1 = 16 * (s >> 2) + 4 * (l >> 4) + (s & 3), j = 16 * mj + (l & 15);
            return (h1 < hid && j < K1) ? (long)h1 * K1 + j : -1;
        });
        fill_store(fr0, smem + FK::OFF_W1, tid);
        fill_store(fr1, smem + FK::OFF_W2, tid);
        fill_store(fr2, smem + FK::OFF_B1, tid);
        fill_store(fr3, smem + FK::OFF_B2, tid);
        fill_store(fr4, smem + FK::OFF_WP, tid);
        fill_store(fr5, smem + K::OFF_W3T, tid);
        fill_store(fr6, smem + K::OFF_W1T, tid);
    }
    __syncthreads();

    const float* const W1L = smem + FK::OFF_W1;
    const float* const W2L = smem + FK::OFF_W2;
    const float* const B1L = smem + FK::OFF_B1;
    const float* const B2L = smem + FK::OFF_B2;
    const float* const W3T = smem + K::OFF_W3T;
    const float* const W1T = smem + K::OFF_W1T;
    float* const PWR = smem + K::SHARED + wave * K::PW;
    float* const Z = PWR + FK::PW_Z;
    float* const XR = PWR + FK::PW_XR;
    float* const PN = PWR + FK::PW_A3;
    const float* const MK = PWR + FK::PW_A3 + ZROWS * RS;
    float* const TB = PWR + K::PW_TB;
    float* const A1 = PWR + K::PW_A1;

    // BFM: bf16 A operands of every product, built once per launch from the f32 LDS images (same k orders as the f32 path) and
    // kept as ONE packed image in LDS, [operand][lane] x 8 bytes, over the (then dead) f32 images: 120 operand registers per
    // lane would not fit beside the 128 weight-gradient accumulators.
    constexpr int KS1 = (K::K1S + 3) / 4;                 // bf16 k-steps of layer 1 (slots q = 3*c4 + f, zero padded)
    constexpr int OP_W1 = 0, OP_W2 = OP_W1 + 4 * KS1, OP_W3T = OP_W2 + 16, OP_W2T = OP_W3T + 4, OP_W1T = OP_W2T + 16, OP_N = OP_W1T + 4 * K::MJ;
    static_assert(OP_N * 64 * 2 <= FK::OFF_B1, "bf16 operand image fits over the f32 W1 | W2 | W3 images");
    const bf_s16x4* const BW = reinterpret_cast<const bf_s16x4*>(smem) + lane;      // operand o of this lane: BW[o * 64]
    if constexpr (BFM) {
        bf_s16x4 img[(OP_N + kBwdWaves - 1) / kBwdWaves];    // this wave's share of the operands (round robin)
        const int w2t_lane0 = (ci & 3) * 64 + (ci >> 2) * 16 + 4 * g;
        auto build = [&](int o) -> bf_s16x4 {
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (o < OP_W2) {                           // W1 forward: (m, s)
                const int m = o / KS1, s_ = o % KS1;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = 4 * s_ + r < K::K1S ? W1L[(m * K::K1S + 4 * s_ + r) * 64 + lane] : 0.0f;
            } else if (o < OP_W3T) {                   // W2 forward: (m2, kk)
                const int m2 = (o - OP_W2) >> 2, kk = (o - OP_W2) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = W2L[(m2 * 16 + 4 * kk + r) * 64 + lane];
            } else if (o < OP_W2T) {                   // W3^T: m
                const int m = o - OP_W3T;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = W3T[(m * 4 + r) * 64 + lane];
            } else if (o < OP_W1T) {                   // W2^T: (m, mp) = W2[h2 = 16mp+4g+r][h1 = 16m+ci]
                const int m = (o - OP_W2T) >> 2, mp = (o - OP_W2T) & 3;
                const f32x4 t = ld4(W2L + (mp * 16 + 4 * m) * 64 + w2t_lane0);
                v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
            } else {                                   // W1^T: (mj, kk)
                const int mj = (o - OP_W1T) >> 2, kk = (o - OP_W1T) & 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = W1T[(mj * 16 + 4 * kk + r) * 64 + lane];
            }
            return pack4(v[0], v[1], v[2], v[3]);
        };
#pragma unroll
        for (int k = 0; k < (OP_N + kBwdWaves - 1) / kBwdWaves; ++k) {
            const int o = k * kBwdWaves + wave;
            img[k] = o < OP_N ? build(o) : bf_s16x4{0, 0, 0, 0};
        }
        __syncthreads();                               // every wave has read what it needs of the f32 images
#pragma unroll
        for (int k = 0; k < (OP_N + kBwdWaves - 1) / kBwdWaves; ++k) {
            const int o = k * kBwdWaves + wave;
            if (o < OP_N) *(reinterpret_cast<bf_s16x4*>(smem) + o * 64 + lane) = img[k];
        }
        __syncthreads();
    }
    // persistent weight-gradient accumulators (D = A * B^T with the cell axis as K)
    f32x4 aW1[4][K::MJ], aW2[4][4], aW3[4];
    float db1[4][4], db2[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aW3[i] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    for (NcaTileWalk tw = nca_tile_walk(a.B * st_x * st_y); tw.t < tw.end; tw.t += tw.stride) {
        int lane = lane_w;   // opaque per tile: lane-derived offsets are recomputed where used, not hoisted out of the tile loop and spilled
        asm volatile("" : "+v"(lane));
        const int g = (lane >> 4) & 3, ci = lane & 15;
        WTile t;
        t.b = tw.t / (st_x * st_y);
        t.ty0 = ((tw.t / st_x) % st_y) * BSTH + wave * WTH;
        t.tx0 = (tw.t % st_x) * BSTW;
        if (t.ty0 >= H || t.tx0 >= W) continue;
        t.valid = true;
        t.inner = t.ty0 >= halo && t.ty0 + WTH + halo <= H && t.tx0 >= halo && t.tx0 + WTW + halo <= W;
        const int ty0 = t.ty0, tx0 = t.tx0;

        NCA_BPHASE(0);   // loop overhead / previous tile's tail
        // ---- all global loads of the tile are requested up front (one HBM round trip per tile instead of two: with one
        //      wave per SIMD nothing else hides it): forward operands, pending x'_t (alpha halo 1 + interior), incoming gradient
        TileRegs<CP, ST> R;
        issue_loads<CP, true, true, -1, false, ST>(a, t, lane, R);
        const char* const xn = reinterpret_cast<const char*>(ba.x_next) + (size_t)t.b * C * plane * ST::BYTES;
        const float* const gn = ba.g_next + (size_t)t.b * C * plane;
        const int hl = (lane >> 5) & 1, l5 = lane & 31;
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
        // ---- forward staging: s_t (XR), pre_t (PN), z_t (Z), fire mask (MK) -------------------------------
        const TileLds L = wave_private_lds<CP>(PWR);
        if (t.inner) stage_tile<CP, false, false, ST>(a, t, L, lane, R, 0);
        else stage_tile<CP, true, false, ST>(a, t, L, lane, R, 0);
        NCA_BPHASE(1);   // forward staging

        // ---- pending x'_t: alpha halo 1 -> A1 (post mask), interior -> XR; incoming gradient -> TB --------
        {
            if (use_alive) {
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (l5 < 18) A1[(2 * k + hl) * RS + l5 + 3] = aok[k] ? ST::cv1(av[k]) : NCA_NEG_INF;
            }
            // z_t interior out (kernel B needs it for the perception-weight gradient)
            // (BFM: the two scratch tensors kernel B streams, z_t and dL/dperception, are stored as bf16: half the bytes)
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
            wave_sync();
#pragma unroll
            for (int k = 0; k < CP / 4; ++k) {
                const int ch = 4 * k + g;
                const bool live = ok && ch < C;
                st4(XR + ch * XRS + row * WTW + 4 * ff, live ? ST::cv4(xv[k]) : f32x4{0.f, 0.f, 0.f, 0.f});
                st4(TB + ch * XRS + row * WTW + 4 * ff, live ? gv[k] : f32x4{0.f, 0.f, 0.f, 0.f});
            }
            wave_sync();
        }

        // ---- warm the L2 for the NEXT tile: one 4-byte read per row segment it will load (state and goal rows with halo 1,
        //      pending state and gradient rows).  There are no registers to hold the real loads a tile ahead (one wave per SIMD,
        //      256 + 200 in use), but these need only a handful of landing registers: they are issued here, return during the
        //      pass loop, and the tile's real requests then meet the L2 instead of HBM.  They are ordinary loads the compiler
        //      tracks (issue pinned by the scheduling fence, values "consumed" by an empty asm after the pass loop): an
        //      inline-asm load into a register the allocator may copy or re-use while the load is in flight is a race.
        constexpr int NW1 = (8 * 2 * CP + 63) / 64, NW2 = (8 * CP + 63) / 64;
        float warm[NW1 + NW2];
#pragma unroll
        for (int k = 0; k < NW1 + NW2; ++k) warm[k] = 0.0f;
        if (tw.t + tw.stride < tw.end) {
            const int tnx = tw.t + tw.stride, gch = a.goal_ch;
            const int nb = tnx / (st_x * st_y), ny0 = ((tnx / st_x) % st_y) * BSTH + wave * WTH, nx0 = (tnx % st_x) * BSTW;
            constexpr unsigned SB = ST::BYTES;      // state-type tensors are addressed in bytes; a 4-byte read of a bf16 row
            // covers two elements (even column: aligned), which is fine: the value is never used
            const char* const bx = reinterpret_cast<const char*>(a.x_in) + (size_t)nb * C * plane * SB;
            const char* const bg = gch ? reinterpret_cast<const char*>(a.goal) + (size_t)nb * gch * plane * SB : bx;
            const char* const bn = reinterpret_cast<const char*>(ba.x_next) + (size_t)nb * C * plane * SB;
            const float* const bq = ba.g_next + (size_t)nb * C * plane;
            const unsigned col = (unsigned)min(nx0, W - 1) & ~1u;
            // state + goal planes: 8 rows each (ty0-1 ..; two more than needed keeps the index arithmetic to shifts)
#pragma unroll
            for (int k = 0; k < NW1; ++k) {
                const int i = min(64 * k + lane, 8 * (C + gch) - 1), pl = i >> 3;
                const unsigned rowo = (unsigned)min(max(ny0 + (i & 7) - 1, 0), H - 1) * (unsigned)W + col;
                const char* const p = (pl < C ? bx + (size_t)((unsigned)pl * plane + rowo) * SB : bg + (size_t)((unsigned)(pl - C) * plane + rowo) * SB);
                warm[k] = *reinterpret_cast<const float*>(p);
            }
            // pending state + incoming gradient: 4 rows each
#pragma unroll
            for (int k = 0; k < NW2; ++k) {
                const int i = min(64 * k + lane, 8 * C - 1), pl = i >> 2;
                const unsigned rowo = (unsigned)min(ny0 + (i & 3), H - 1) * (unsigned)W + col;
                const void* const p = pl < C ? (const void*)(bn + (size_t)((unsigned)pl * plane + rowo) * SB)
                                             : (const void*)(bq + (unsigned)(pl - C) * plane + rowo);
                warm[NW1 + k] = *reinterpret_cast<const float*>(p);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // the requests stay HERE (not sunk to their "use" after the pass loop)
        NCA_BPHASE(2);   // x'/g loads, z out
        // ---- dL/dx'_t = G * 1[lo <= x'*life <= hi] * life (nca.py:191-194);  d out = . * fire mask.  All four
        //      rows now: TB (the staged incoming gradient) is reused for the transposes inside the pass loop.
        float dOall[WTH][4];
        {
            // reads of all four rows first, then the arithmetic and the XR write-back: with the write-back inside the row loop
            // every row's reads waited behind the previous row's stores (possible aliases) -- four exposed LDS round trips
            float lifev[WTH], mkv[WTH], xr[WTH][4], gr[WTH][4];
#pragma unroll
            for (int row = 0; row < WTH; ++row) {
                float life = PN[(row + 1) * RS + ci + 4];
                if (use_alive) life = (life != 0.0f && max3x3(A1 + (row + 1) * RS + ci + 4) > a.thr) ? 1.0f : 0.0f;
                lifev[row] = life;
                mkv[row] = MK[row * WTW + ci];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g + r;
                    xr[row][r] = ch < CP ? XR[ch * XRS + row * WTW + ci] : 0.0f;   // XR / TB hold CP channel planes
                    gr[row][r] = ch < CP ? TB[ch * XRS + row * WTW + ci] : 0.0f;
                }
            }
#pragma unroll
            for (int row = 0; row < WTH; ++row)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = 4 * g + r;
                    const float y = xr[row][r] * lifev[row];
                    const float gxv = (ch < C && y >= a.lo && y <= a.hi) ? gr[row][r] * lifev[row] : 0.0f;
                    if (ch < CP) XR[ch * XRS + row * WTW + ci] = gxv;  // XR now carries dL/dx'_t for the 16-byte store pass
                    dOall[row][r] = gxv * mkv[row];
                }
        }
        NCA_BPHASE(3);   // gate
#pragma unroll 1
        for (int pass = 0; pass < WTH / NT; ++pass) {
            const int n0 = pass * NT;
            // ---- forward recompute: P, h1, h2 kept in registers ------------------------------------------
            float P[NT][K::K1S];
            perceive_tile<CP, NT>(smem, Z, lane, n0, P);
            NCA_BPHASE(4);   // perception
            f32x4 dp[K::MJ][NT];
            if constexpr (BFM) {
                short* const tb16 = reinterpret_cast<short*>(TB);
                // ---- forward recompute on bf16 MFMA (rounding points of the bf16 forward kernel) ---------------------------
                bf_s16x4 pb[NT][KS1];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int s_ = 0; s_ < KS1; ++s_) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = 4 * s_ + r < K::K1S ? P[n][4 * s_ + r] : 0.0f;
                        pb[n][s_] = pack4(v[0], v[1], v[2], v[3]);
                    }
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
                bf_s16x4 dOb[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int rw = pass ? NT + n : n;
                    dOb[n] = pack4(dOall[rw][0], dOall[rw][1], dOall[rw][2], dOall[rw][3]);
                }
                bf_s16x4 d2b[4][NT], d1b[4][NT];
                // ---- layer 3: dW3 += dO x h2 (cells as K);  d2 = (W3^T dO) * 1[h2 > 0] ---------------------------------------
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wave_sync();
                    tb_write(tb16, 0, lane, dOb[n]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) tb_write(tb16, 1 + m, lane, h2b[m][n]);
                    wave_sync();
                    const bf_s16x4 ta = tb_tr_read(tb16, 0, lane);
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) aW3[nb] = mfma_bf16(ta, tb_tr_read(tb16, 1 + nb, lane), aW3[nb]);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        f32x4 d = mfma_bf16(BW[(OP_W3T + m) * 64], dOb[n], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int r = 0; r < 4; ++r) { d[r] = h2f[m][n][r] > 0.0f ? d[r] : 0.0f; db2[m][r] += d[r]; }
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
                        for (int r = 0; r < 4; ++r) { d[r] = h1f[m][n][r] > 0.0f ? d[r] : 0.0f; db1[m][r] += d[r]; }
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
                    for (int s_ = 0; s_ < KS1; ++s_) tb_write_chunk(tb16, 16 + 3 * g + s_, lane, pb[n][s_]);   // columns 12g + 4s .. of tiles 4..6
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
            float dO[NT][4];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) dO[n][r] = pass ? dOall[NT + n][r] : dOall[n][r];
            // Backward data path and weight gradients, layer by layer from the output: each layer's weight-gradient product
            // is issued as soon as its two factors exist, so h2 dies after layer 3, d2 and h1 after layer 2, P and d1 after
            // layer 1 (all five tiles alive at once through a separate weight-gradient phase cost 36 spilled registers).
            // Weight gradients go per 16-cell tile through the cell-major buffer TB[cell][row]: lane (g,ci) writes its 4
            // accumulator rows of a tile with ONE 16-byte store; operand fragments A[i][k=g] = TB[4s+g][rowA+i],
            // B[k=g][j] = TB[4s+g][rowB+j] are 16 consecutive floats per lane group.
            float* const tw_ = TB + ci * TBS + 4 * g;            // this lane's cell row, accumulator-row offset
            const float* const tr_ = TB + g * TBS + ci;           // operand reads: cell 4s+g -> + 4*s*TBS
            f32x4 d2[4][NT], d1[4][NT];
            // ---- layer 3: dW3 = dO (rows 0..15) x h2 (rows 16..79);  d2 = (W3^T dO) * 1[h2 > 0] ---------------------------
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                wave_sync();  // TB free (gradient tile consumed above / previous products done)
                st4(tw_, f32x4{dO[n][0], dO[n][1], dO[n][2], dO[n][3]});
#pragma unroll
                for (int m = 0; m < 4; ++m) st4(tw_ + 16 + 16 * m, h2[m][n]);
                wave_sync();
                piped<4>(
                    [&](int s_) {
                        OpN<5> o;
                        o.v[0] = tr_[4 * s_ * TBS];
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) o.v[1 + nb] = tr_[4 * s_ * TBS + 16 + 16 * nb];
                        return o;
                    },
                    [&](int, const OpN<5>& o) {
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb) aW3[nb] = nca_mfma(o.v[0], o.v[1 + nb], aW3[nb]);
                    });
            }
            piped<4>(
                [&](int m) {
                    OpN<4> o;
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_) o.v[s_] = W3T[(m * 4 + s_) * 64 + lane];
                    return o;
                },
                [&](int m, const OpN<4>& o) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) d2[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s_ = 0; s_ < 4; ++s_)
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
                    for (int r = 0; r < 4; ++r) TB[(16 * mj + 4 * g + r) * 36 + n * 16 + ci] = dp[mj][n][r];
            wave_sync();
            {
                const int rr = (lane >> 2) & 1, ff = lane & 3, jl = lane >> 3;  // 8 rows of j per instruction
                const int gy = ty0 + n0 + rr, gx = tx0 + 4 * ff;
                const bool ok = gy < H && gx + 3 < W;
                float* const po = ba.dP + (size_t)t.b * 3 * C * plane + (ok ? (unsigned)(gy * W + gx) : 0u);
                uint16_t* const po16 = reinterpret_cast<uint16_t*>(ba.dP) + (size_t)t.b * 3 * C * plane + (ok ? (unsigned)(gy * W + gx) : 0u);
#pragma unroll
                for (int k = 0; k < 2 * K::MJ; ++k) {
                    const int j = 8 * k + jl;
                    const f32x4 v = ld4(TB + j * 36 + rr * 16 + 4 * ff);
                    if (ok && j < K1) {
                        if constexpr (BFM) *reinterpret_cast<u32x2*>(po16 + (unsigned)j * plane) = u32x2{pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3])};
                        else st4(po + (unsigned)j * plane, v);
                    }
                }
            }
            NCA_BPHASE(9);   // dP out
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < NW1 + NW2; ++k) asm volatile("" ::"v"(warm[k]));   // the warm-up reads have long returned
        // ---- dL/dx'_t out (XR), 16-byte stores ---------------------------------------------------------------
        wave_sync();
        {
            const int row = (lane >> 2) & 3, ff = lane & 3, gy = ty0 + row, gx = tx0 + 4 * ff;
            const bool ok = gy < H && gx + 3 < W;
            float* const go = ba.gx + (size_t)t.b * C * plane + (ok ? (unsigned)(gy * W + gx) : 0u);
#pragma unroll
            for (int k = 0; k < CP / 4; ++k) {
                const int ch = 4 * k + g;
                if (ok && ch < C) st4(go + (unsigned)ch * plane, ld4(XR + ch * XRS + row * WTW + 4 * ff));
            }
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
    using TM = SlabTM<K::MJ, 1>;
    static_assert(TM::stage(4) <= K::LDS_FLOATS, "slab staging fits the LDS carve");
    slab_flush_tm<TM, kBwdThreads, 4>(smem, ba.slabs + (size_t)blockIdx.x * TM::SF, tid, lane, wave, aW1, aW2,
                                      reinterpret_cast<const f32x4 (&)[1][4]>(aW3), db1, db2);
#if defined(NCA_STAMPS)
    NCA_BPHASE(11);  // slab flush
    if (a.dbg && lane == 0) a.dbg[(size_t)(blockIdx.x * kBwdWaves + wave) * 16 + 11] = ph_acc[11];
#endif
}

// Kernel B: dL/ds_t = dL/dx'_t + stencil^T(dL/dP);  dL/dgoal += dz * pre_t;  perception-weight partials.
// One thread = 4 W-contiguous cells x a strip of SROWS rows of one (b, c) plane, walked top to bottom with a
// three-row sliding window held in registers: every row of z / dL/dP is loaded once per thread (16-byte loads,
// left/right neighbours by wavefront shuffle with a scalar fallback at row/wave edges), and the 27 perception-weight
// sums are reduced across the block once per strip.  A block never spans two channels.
constexpr int SROWS = 16;   // tallest strip.  The launcher picks 16, 8 or 4 rows per strip (nca_cond_bwd_srows): a launch whose wave count
                            // is not a multiple of the chip's resident waves (two per SIMD at 192-200 registers) pays for the last partial
                            // round in full, and shorter strips make that round cheaper (or, on small grids, fill the chip at all)
struct Row6 { float v[6]; };   // columns x0-1 .. x0+4

// Two-stage row fetch: issue (raw 16-byte group + the two edge cells that have no neighbour lane) and finish (shuffle the
// neighbours' edge values in).  Anything that touches the loaded value belongs to finish -- a shuffle at issue time waits
// for the load and defeats the prefetch.
struct RawRow {
    float4 c;
    float el, er;
};   // (whether the row lies inside the image is recomputed from its index at finish time: one register less per row in flight)
__device__ __forceinline__ RawRow issue_row6(const float* __restrict__ plane, int H, int W, int y, int x0, bool has_l, bool has_r) {
    RawRow r;
    const float* const row = plane + (size_t)((y >= 0 && y < H) ? y : 0) * W;
    r.c = *reinterpret_cast<const float4*>(row + x0);
    r.el = 0.0f;
    r.er = 0.0f;
    if (!has_l && x0 > 0) r.el = row[x0 - 1];
    if (!has_r && x0 + 4 < W) r.er = row[x0 + 4];
    return r;
}
// bf16 scratch (the BFM backward): 8-byte group + 2-byte edges, widened on arrival (the widening touches the loaded value, so a
// row issued two rows ahead pays no wait here either: the compiler places the wait at the first shift)
__device__ __forceinline__ RawRow issue_row6(const uint16_t* __restrict__ plane, int H, int W, int y, int x0, bool has_l, bool has_r) {
    RawRow r;
    const uint16_t* const row = plane + (size_t)((y >= 0 && y < H) ? y : 0) * W;
    const u32x2 c = *reinterpret_cast<const u32x2*>(row + x0);
    unsigned el = 0u, er = 0u;
    if (!has_l && x0 > 0) el = row[x0 - 1];
    if (!has_r && x0 + 4 < W) er = row[x0 + 4];
    r.c = make_float4(__uint_as_float(c[0] << 16), __uint_as_float(c[0] & 0xffff0000u), __uint_as_float(c[1] << 16), __uint_as_float(c[1] & 0xffff0000u));
    r.el = __uint_as_float(el << 16);
    r.er = __uint_as_float(er << 16);
    return r;
}
__device__ __forceinline__ Row6 finish_row6(const RawRow& q, bool in, bool has_l, bool has_r) {
    Row6 r;
    const float z = in ? 1.0f : 0.0f;   // rows outside the image read row 0 and are zeroed here
    float l = __shfl_up(q.c.w, 1), rr = __shfl_down(q.c.x, 1);
    if (!has_l) l = q.el;
    if (!has_r) rr = q.er;
    r.v[0] = l * z; r.v[1] = q.c.x * z; r.v[2] = q.c.y * z; r.v[3] = q.c.z * z; r.v[4] = q.c.w * z; r.v[5] = rr * z;
    return r;
}
template <typename ET>
__device__ __forceinline__ Row6 load_row6(const ET* __restrict__ plane, int H, int W, int y, int x0, bool has_l, bool has_r) {
    return finish_row6(issue_row6(plane, H, W, y, x0, has_l, has_r), y >= 0 && y < H, has_l, has_r);
}

// ET = element type of the two scratch tensors kernel A wrote (z_t, dL/dperception): float, or uint16_t (bf16) behind the BFM kernel A
template <typename ET, int SR>
__global__ __launch_bounds__(256, 2) void cond_step_bwd_stencil_kernel(const NcaCondBwdArgs ba) {
    const NcaCondArgs& a = ba.f;
    const int C = a.C, H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;
    static_assert(SR == 16 || SR == 8 || SR == 4, "strip heights the launcher chooses from");
    const int W4 = W / 4, strips = (H + SR - 1) / SR;
    const int per_plane = strips * W4, blocks_per_plane = (per_plane + 255) / 256;
    const int bc = blockIdx.x / blocks_per_plane, b = bc / C, c = bc % C;
    const int id = (blockIdx.x % blocks_per_plane) * 256 + threadIdx.x;
    const bool active = id < per_plane;
    const int ida = active ? id : per_plane - 1;      // inactive lanes shadow the last item (they still shuffle)
    const int x0 = (ida % W4) * 4, y0 = (ida / W4) * SR;
    const int lane = threadIdx.x & 63;
    const bool has_l = x0 > 0 && lane > 0, has_r = x0 + 4 < W && lane < 63 && id + 1 < per_plane;
    float wl[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) wl[i] = a.wp[(size_t)c * 27 + i];
    const ET* const zb = reinterpret_cast<const ET*>(ba.zbuf) + ((size_t)b * C + c) * plane;
    const ET* const p0 = reinterpret_cast<const ET*>(ba.dP) + ((size_t)b * 3 * C + 3 * c) * plane;
    float wsum[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) wsum[i] = 0.f;
    const int gch0 = C - a.goal_ch;
    // window rows: index 0 = y-1, 1 = y, 2 = y+1.  Prefetch depth TWO: the loads of row y+3 are requested in the iteration that
    // computes row y and enter the window at the end of the NEXT iteration (two register sets, A / B, the loop unrolled by two) -- one
    // iteration's ~1 K cycles of arithmetic do not cover a memory round trip, and with a single set every iteration waited for its own
    // request.  The row's own read-modify-write operands (dL/dx', dL/dgoal, pre mask) are requested one row ahead (a full iteration in flight).
    Row6 zw[3], pw[3][3];
    RawRow znA, pnA[3], znB, pnB[3];
    zw[0] = load_row6(zb, H, W, y0 - 1, x0, has_l, has_r);
    zw[1] = load_row6(zb, H, W, y0, x0, has_l, has_r);
    zw[2] = load_row6(zb, H, W, y0 + 1, x0, has_l, has_r);
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        pw[f][0] = load_row6(p0 + (size_t)f * plane, H, W, y0 - 1, x0, has_l, has_r);
        pw[f][1] = load_row6(p0 + (size_t)f * plane, H, W, y0, x0, has_l, has_r);
        pw[f][2] = load_row6(p0 + (size_t)f * plane, H, W, y0 + 1, x0, has_l, has_r);
    }
    const bool goal_ch = c >= gch0, use_pre = goal_ch && a.alive_ch >= 0;
    const float* const gxp = ba.gx + ((size_t)b * C + c) * plane + x0;
    float* const gop = ba.g_out + ((size_t)b * C + c) * plane + x0;
    float* const dgp = goal_ch ? ba.dgoal + ((size_t)b * a.goal_ch + (c - gch0)) * plane + x0 : nullptr;
    const uint8_t* const prp = use_pre ? ba.pre_t + (size_t)b * plane + x0 : nullptr;
    struct Rmw { float4 gx, dg; uchar4 pb; };
    Rmw rn{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_uchar4(1, 1, 1, 1)};
    auto issue_rmw = [&](Rmw& r, int y) {
        const size_t ro = (size_t)min(y, H - 1) * W;
        r.gx = *reinterpret_cast<const float4*>(gxp + ro);
        if (goal_ch) r.dg = *reinterpret_cast<const float4*>(dgp + ro);
        if (use_pre) r.pb = *reinterpret_cast<const uchar4*>(prp + ro);
    };
    auto issue_rows = [&](RawRow& zi, RawRow (&pi)[3], int y) {
        zi = issue_row6(zb, H, W, y, x0, has_l, has_r);
#pragma unroll
        for (int f = 0; f < 3; ++f) pi[f] = issue_row6(p0 + (size_t)f * plane, H, W, y, x0, has_l, has_r);
    };
    issue_rows(znA, pnA, y0 + 2);
    issue_rmw(rn, y0);
    // one row: request row y+3 into (zi, pi) and the read-modify-write operands of row y+1 (after taking this row's), compute row y,
    // then move the rows requested an iteration ago (zf, pf) into the window
    auto row = [&](int k, RawRow& zi, RawRow (&pi)[3], const RawRow& zf, const RawRow (&pf)[3]) {
        const int y = y0 + k;
        if (k + 3 <= SR) issue_rows(zi, pi, y + 3);        // rows up to y0 + SR are needed (the halo below the strip)
        const float4 gx = rn.gx, dgc = rn.dg;
        const uchar4 pb = rn.pb;
        if (k + 1 < SR) issue_rmw(rn, y + 1);
        if (active && y < H) {
            // dz[x] = sum_f sum_{ty,tx} Wp[f][ty][tx] * dP[f][y-(ty-1)][x-(tx-1)]      (transpose of the zero-padded correlation)
            float dz[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            dz[j] = fmaf(wl[9 * f + 3 * ty + tx], pw[f][2 - ty].v[j + 2 - tx], dz[j]);
            // dWp[f][ty][tx] += dP[f][y][x] * z[y+ty-1][x+tx-1]
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            wsum[9 * f + 3 * ty + tx] = fmaf(pw[f][1].v[j + 1], zw[ty].v[j + tx], wsum[9 * f + 3 * ty + tx]);
            const size_t ro = (size_t)y * W;
            *reinterpret_cast<float4*>(gop + ro) = make_float4(gx.x + dz[0], gx.y + dz[1], gx.z + dz[2], gx.w + dz[3]);
            if (goal_ch) {
                float4 o = dgc;
                o.x += dz[0] * (float)pb.x; o.y += dz[1] * (float)pb.y; o.z += dz[2] * (float)pb.z; o.w += dz[3] * (float)pb.w;
                *reinterpret_cast<float4*>(dgp + ro) = o;
            }
        }
        const bool in2 = y + 2 >= 0 && y + 2 < H;          // (zf, pf) hold row y + 2
        zw[0] = zw[1]; zw[1] = zw[2]; zw[2] = finish_row6(zf, in2, has_l, has_r);
#pragma unroll
        for (int f = 0; f < 3; ++f) { pw[f][0] = pw[f][1]; pw[f][1] = pw[f][2]; pw[f][2] = finish_row6(pf[f], in2, has_l, has_r); }
    };
    static_assert(SR % 2 == 0, "the row loop is unrolled by two (register sets A / B)");
#pragma unroll 1
    for (int k = 0; k < SR; k += 2) {
        row(k, znB, pnB, znA, pnA);
        row(k + 1, znA, pnA, znB, pnB);
    }
    // block reduction of the 27 partial sums, fixed order (deterministic)
    __shared__ float red[4][27];
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        float v = row16_sum(active ? wsum[i] : 0.0f);   // 16 lanes on the VALU (DPP), the four rows through two crossbar exchanges
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane == 0) red[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27)
        ba.wp_partials[(size_t)blockIdx.x * 27 + threadIdx.x] +=
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// dst[j] = sum_i src[i*m + j] (fixed order -> deterministic)
// dst[j] = sum_i src[i][j]: a block owns 16 columns, its 16 row groups take every sixteenth row each, partial sums combined
// in a fixed order (deterministic)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int m, int accumulate) {
    __shared__ float part[16][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4, j = blockIdx.x * 16 + c;
    float acc0 = 0.0f, acc1 = 0.0f;
    if (j < m) {
        int i = rg;
        for (; i + 16 < n; i += 32) {
            acc0 += src[(size_t)i * m + j];
            acc1 += src[(size_t)(i + 16) * m + j];
        }
        if (i < n) acc0 += src[(size_t)i * m + j];
    }
    part[rg][c] = acc0 + acc1;
    __syncthreads();
    if (rg == 0 && j < m) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += part[k][c];
        dst[j] = accumulate ? dst[j] + v : v;
    }
}
// Summed tile-major slab (SlabTM) -> gradients in the reference layouts: w1 [hid, 3C], w2 [hid, hid], w3 [C, hid], b1, b2.  Once per
// backward pass.  `slot_order`: the bf16-MFMA kernels accumulate dW1 with the perception index in operand-slot order (column
// 4 KS1 g' + q, q = 3 c4 + f  <->  channel 4 c4 + g', filter f).
__global__ __launch_bounds__(256) void cond_bwd_unpermute_kernel(const float* __restrict__ red, int C, int cp, int hid, int slot_order,
                                                                 float* __restrict__ g_w1, float* __restrict__ g_w2, float* __restrict__ g_w3,
                                                                 float* __restrict__ g_b1, float* __restrict__ g_b2) {
    const int K1 = 3 * C, mj = slab_mj(cp), m3t = slab_m3t(cp), ks1 = (3 * cp / 4 + 3) / 4;
    const int s1 = 1024 * mj, n1 = hid * K1, n2 = hid * hid, n3 = C * hid;
    const int i = blockIdx.x * 256 + threadIdx.x;
    auto tile_at = [&](int base, int tiles_per_row, int row, int col) -> float {   // element (row, col) of a [.][tiles_per_row] tile grid
        const int ma = row >> 4, g = (row >> 2) & 3, r = row & 3, nb = col >> 4, ci = col & 15;
        return red[base + ((ma * tiles_per_row + nb) * 64 + g * 16 + ci) * 4 + r];
    };
    if (i < n1) {
        const int o = i / K1, j = i - o * K1;
        int col = j;
        if (slot_order) {
            const int ch = j / 3, f = j - 3 * ch;
            col = 4 * ks1 * (ch & 3) + 3 * (ch >> 2) + f;
        }
        g_w1[i] = tile_at(0, mj, o, col);
    } else if (i < n1 + n2) {
        const int e = i - n1, o = e / hid, k = e - o * hid;
        g_w2[e] = tile_at(s1, 4, o, k);
    } else if (i < n1 + n2 + n3) {
        const int e = i - n1 - n2, ch = e / hid, k = e - ch * hid;
        g_w3[e] = tile_at(s1 + 4096, 4, ch, k);
    } else if (i < n1 + n2 + n3 + hid) {
        const int o = i - n1 - n2 - n3;
        g_b1[o] = red[s1 + 4096 + 1024 * m3t + o];
    } else if (i < n1 + n2 + n3 + 2 * hid) {
        const int o = i - n1 - n2 - n3 - hid;
        g_b2[o] = red[s1 + 4096 + 1024 * m3t + 64 + o];
    }
}

// perception-weight partials [B*C*bpp][27] -> grad [C][27]
__global__ __launch_bounds__(64) void reduce_wp_kernel(const float* __restrict__ part, float* __restrict__ dst, int B, int C, int bpp) {
    const int c = blockIdx.x, i = threadIdx.x;
    if (i >= 27) return;
    float acc = 0.0f;
    for (int b = 0; b < B; ++b)
        for (int k = 0; k < bpp; ++k) acc += part[((size_t)(b * C + c) * bpp + k) * 27 + i];
    dst[c * 27 + i] = acc;
}

int g_bwd_variant = [] { const char* e = getenv("NCAHIP_BWD_VARIANT"); return e ? atoi(e) : 0; }();

// rows per strip of kernel B: modelled cost = rounds of resident waves x (rows + 2 halo rows), smallest wins (ties: taller strips)
int stencil_srows(int B, int C, int H, int W) {
    const long slots = (long)nca_cu_count() * 8;   // two waves per SIMD
    int best = SROWS;
    long best_cost = -1;
    for (int sr = SROWS; sr >= 4; sr /= 2) {
        const long per_plane = (long)((H + sr - 1) / sr) * (W / 4), waves = (long)B * C * ((per_plane + 255) / 256) * 4;
        const long cost = ((waves + slots - 1) / slots) * (sr + 2);
        if (best_cost < 0 || cost < best_cost) { best = sr; best_cost = cost; }
    }
    return best;
}
int stencil_blocks(int B, int C, int H, int W, int sr) { return B * C * ((((H + sr - 1) / sr) * (W / 4) + 255) / 256); }

hipError_t launch_stencil(const NcaCondBwdArgs& ba_in, hipStream_t st, bool bf16_scratch) {
    NcaCondBwdArgs ba = ba_in;
    const NcaCondArgs& a = ba.f;
    ba.srows = stencil_srows(a.B, a.C, a.H, a.W);
    const int nblk = stencil_blocks(a.B, a.C, a.H, a.W, ba.srows);
    if (nblk > ba.nblk) return hipErrorInvalidValue;   // wp_partials holds ba.nblk rows (nca_cond_bwd_nblk)
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), 0, st, ba);
        return hipGetLastError();
    };
    if (bf16_scratch) return ba.srows == 16 ? go(cond_step_bwd_stencil_kernel<uint16_t, 16>) : ba.srows == 8 ? go(cond_step_bwd_stencil_kernel<uint16_t, 8>)
                                                                                                             : go(cond_step_bwd_stencil_kernel<uint16_t, 4>);
    return ba.srows == 16 ? go(cond_step_bwd_stencil_kernel<float, 16>) : ba.srows == 8 ? go(cond_step_bwd_stencil_kernel<float, 8>)
                                                                                         : go(cond_step_bwd_stencil_kernel<float, 4>);
}

// C <= 16: which form of kernel A runs (launch_bwd below)
bool narrow_is_fm(const NcaCondBwdArgs& ba, bool bfm) {
    const int nst_ = ba.f.B * ((ba.f.W + 15) / 16) * ((ba.f.H + 15) / 16);
    const bool default_fm = bfm || 2 * nst_ <= ba.nslab;
    return ba.pscr && ba.doscr && (g_bwd_variant == 2 || (g_bwd_variant == 0 && default_fm) || (g_bwd_variant == 3 && !default_fm));
}

template <int CP, typename ST, bool BFM = false>
hipError_t launch_bwd(const NcaCondBwdArgs& ba, hipStream_t st) {
    // Kernel A exists in two forms with the same results (the products run in the same per-wave order): ONE launch, everything
    // for a tile in one wave (this file), or TWO launches, front + matrix part (nca_cond_bwd_fm.hip).  Measured at 8 x 16 x 256^2:
    // fp32 products 360 vs 370 us per backward step, bf16 MFMA 196 vs 191 us -- each mode defaults to its faster form, the test
    // hook (ncahip_debug_force_generic bit 3) swaps them so that the parity suite checks both.
    // Small grids (fewer super-tiles than half the CUs): the matrix kernel splits each super-tile over two workgroups, which the one-launch
    // form cannot -- front + matrix is then the faster form for fp32 products as well (64 x 64, batch 8, C = 16: 53 -> ~40 us per step).
    const bool fm = narrow_is_fm(ba, BFM);
    if (fm) {
        if (hipError_t e = nca_launch_cond_step_bwd_fm(ba, st, BFM ? 2 : (ST::BYTES == 2 ? 1 : 0)); e != hipSuccess) return e;
        return ba.opmode == 1 ? hipSuccess : launch_stencil(ba, st, BFM);
    }
    if (ba.opmode == 1) return hipSuccess;   // the one-launch form builds its images itself: nothing to export
    using K = BCfg<CP>;
    auto kern = cond_step_bwd_kernel<CP, ST, BFM>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;   // per instantiation; keyed by device inside
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const NcaCondArgs& a = ba.f;
    const int nst = a.B * ((a.W + 15) / 16) * ((a.H + 15) / 16);
    const int grid = nst < ba.nslab ? nst : ba.nslab;   // one slab per workgroup
#if defined(NCA_STAMPS)
    NcaCondBwdArgs bd = ba;
    bd.f.dbg = nca_debug_stamp_ptr();
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), lds, st, bd);
#else
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), lds, st, ba);
#endif
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_stencil(ba, st, BFM);
}

}  // namespace

// channel padding of the backward kernels' instantiation that serves C channels (nca_cond_bwd_fm.hip: fm_cp)
static int bwd_cp(int C) { return C <= 12 ? 12 : (C <= 16 ? 16 : (C <= 20 ? 20 : (C <= 24 ? 24 : 32))); }
int nca_cond_bwd_slab_floats(int C, int hidden) { (void)hidden; return slab_floats_cp(bwd_cp(C)); }
int nca_cond_bwd_nslab() { return nca_cu_count(); }   // one persistent workgroup (and one slab) per CU
int nca_cond_bwd_nblk(int B, int C, int H, int W) { return stencil_blocks(B, C, H, W, stencil_srows(B, C, H, W)); }

static bool g_bwd_bf16_exact = getenv("NCAHIP_BWD_BF16_EXACT") != nullptr;
void nca_set_bwd_bf16_exact(bool on) { g_bwd_bf16_exact = on; }
void nca_set_bwd_variant(int v) { g_bwd_variant = v; }

// W % 4 == 0 and 16-byte aligned tensors required (checked by the C ABI).
hipError_t nca_launch_cond_step_bwd(const NcaCondBwdArgs& ba, hipStream_t st, bool bf16) {
    if (bf16) {   // history (f.x_in, x_next) and goal hold bf16; gradients and scratch stay f32
        if (g_bwd_bf16_exact) {   // test hook: exact-f32 recomputation from the widened history
            if (ba.f.C <= 12) return launch_bwd<12, StBF16>(ba, st);
            if (ba.f.C <= 16) return launch_bwd<16, StBF16>(ba, st);
        }
        if (ba.f.C <= 12) return launch_bwd<12, StBF16, true>(ba, st);
        if (ba.f.C <= 16) return launch_bwd<16, StBF16, true>(ba, st);
        if (ba.f.C <= 20) {   // the reference's default model over a bf16 history: front + matrix kernels at CP = 20 (bf16 MFMA, one wave per
                              // SIMD); with the exact hook: exact-f32 products of the widened history, as the fp32 form
            if (!ba.pscr || !ba.doscr) return hipErrorInvalidValue;
            if (hipError_t e = nca_launch_cond_step_bwd_fm(ba, st, g_bwd_bf16_exact ? 1 : 2); e != hipSuccess) return e;
            return ba.opmode == 1 ? hipSuccess : launch_stencil(ba, st, !g_bwd_bf16_exact);
        }
        return hipErrorInvalidValue;
    }
    if (ba.f.C <= 12) return launch_bwd<12, StF32>(ba, st);
    if (ba.f.C <= 16) return launch_bwd<16, StF32>(ba, st);
    if (ba.f.C <= 32) {   // 16 < C <= 32 (the reference's default model is C = 20, nca.py:62-94): front + matrix kernels only
        if (!ba.pscr || !ba.doscr) return hipErrorInvalidValue;
        if (hipError_t e = nca_launch_cond_step_bwd_fm(ba, st, 0); e != hipSuccess) return e;
        return ba.opmode == 1 ? hipSuccess : launch_stencil(ba, st, false);
    }
    return hipErrorInvalidValue;
}

bool nca_cond_bwd_is_fm(const NcaCondBwdArgs& ba, bool bf16) {
    if (ba.f.C > 16) return true;                                   // wide channel counts: front + matrix kernels only
    return narrow_is_fm(ba, bf16 && !g_bwd_bf16_exact);
}

hipError_t nca_launch_reduce_rows(const float* src, float* dst, int n, int m, hipStream_t st, bool accumulate) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((m + 15) / 16), dim3(256), 0, st, src, dst, n, m, accumulate ? 1 : 0);
    return hipGetLastError();
}
hipError_t nca_launch_cond_bwd_unpermute(const float* red, int C, int hidden, bool bf16_history, float* g_w1, float* g_w2, float* g_w3,
                                         float* g_b1, float* g_b2, hipStream_t st) {
    const int n = hidden * 3 * C + hidden * hidden + C * hidden + 2 * hidden;
    const int slot_order = (bf16_history && !g_bwd_bf16_exact) ? 1 : 0;   // the bf16-MFMA products accumulate dW1 in operand-slot order
    hipLaunchKernelGGL(cond_bwd_unpermute_kernel, dim3((n + 255) / 256), dim3(256), 0, st, red, C, bwd_cp(C), hidden, slot_order, g_w1, g_w2,
                       g_w3, g_b1, g_b2);
    return hipGetLastError();
}
hipError_t nca_launch_reduce_wp(const float* part, float* dst, int B, int C, int H, int W, hipStream_t st) {
    const int sr = stencil_srows(B, C, H, W), bpp = (((H + sr - 1) / sr) * (W / 4) + 255) / 256;
    hipLaunchKernelGGL(reduce_wp_kernel, dim3(C), dim3(64), 0, st, part, dst, B, C, bpp);
    return hipGetLastError();
}
