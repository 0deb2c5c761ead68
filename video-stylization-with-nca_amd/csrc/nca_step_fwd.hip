// nca_step_fwd.hip -- fused NCA step kernels for MI355X (gfx950), fp32 exact.
//
// One persistent launch per NCA step.  A workgroup (4 waves) owns TH x TW cell tiles:
//   1. the state tile (+1 halo, pad mode resolved at load time; ConditionedNCA: the pending
//      life mask of the previous step is resolved on an alpha halo of 3) is staged in LDS.
//      Every global load of a tile is issued up front (16-byte row loads, one position per
//      thread, channels in a register loop) so a tile exposes ONE memory latency, and the
//      loads are issued before the barrier that retires the previous tile;
//   2. each wave takes 4 groups of 16 W-contiguous cells; lane l = 16*g + i computes the
//      3x3 depthwise perception of channels {4c'+g} for cell i from LDS (the perception value
//      IS the MFMA B operand: B[k = g][col = i]);
//   3. the per-cell MLP runs on v_mfma_f32_16x16x4_f32 (exact f32 == fmaf chain) with cells on
//      the N/lane axis and channels on M/K: D = W * X.  A layer's accumulator tile
//      (rows 4g+r in reg r, guide sec.3) is the next layer's B operand without any lane
//      movement when that layer's k order is permuted to k(s,g) = 16*(s/4) + 4g + s%4 -- the
//      permutation is folded into the LDS weight image, built once per workgroup;
//   4. residual add + fire mask, stores straight from the accumulator layout.
// Weights live in LDS as the A-operand image [m-tile][k-step][lane]; hidden activations never
// leave registers.  Layer 1 is streamed m-tile by m-tile into layer 2's accumulators so only
// one layer-1 tile is live (keeps 2 waves/SIMD).
#include <cstdlib>

#include "nca_common.h"
#include "nca_kernels.h"

namespace {

constexpr int kThreads = 256;

constexpr int lds_cs(int rows, int rs) {  // channel stride with CS % 32 == 16 (conflict-free g/g+1 pairs)
    int cs = rows * rs;
    while (cs % 32 != 16) ++cs;
    return cs;
}

__device__ __forceinline__ float nca_clamp(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// Batched gather of a weight image: dst[idx] = map(idx) >= 0 ? src[map(idx)] : 0, 8 independent loads
// in flight per thread (a dependent-latency loop here costs tens of microseconds per launch).
template <int N, typename MapT>
__device__ __forceinline__ void fill_image(float* __restrict__ dst, const float* __restrict__ src, int tid, MapT map) {
    constexpr int U = 8;
    for (int base = tid; base < N; base += kThreads * U) {
        float v[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + kThreads * u;
            const long o = idx < N ? map(idx) : -1;
            ok[u] = o >= 0;
            v[u] = src[ok[u] ? o : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + kThreads * u;
            if (idx < N) dst[idx] = ok[u] ? v[u] : 0.0f;
        }
    }
}

// Two-phase form of the same gather: fill_issue requests EVERY element of an image (one register each), fill_commit writes
// them to LDS.  Issuing all images before committing any makes the workgroup's cold start ONE memory round trip instead of one
// per batch of 8 (at B = 1 -- the video loop -- a launch is one tile per workgroup and the prologue is most of it).
template <int N>
struct FillV { float v[(N + kThreads - 1) / kThreads]; };
template <int N, typename MapT>
__device__ __forceinline__ void fill_issue(FillV<N>& r, const float* __restrict__ src, int tid, MapT map) {
#pragma unroll
    for (int u = 0; u < (N + kThreads - 1) / kThreads; ++u) {
        const int idx = tid + kThreads * u;
        const long o = idx < N ? map(idx) : -1;
        r.v[u] = src[o >= 0 ? o : 0];
    }
}
template <int N, typename MapT>
__device__ __forceinline__ void fill_commit(const FillV<N>& r, float* __restrict__ dst, int tid, MapT map) {
#pragma unroll
    for (int u = 0; u < (N + kThreads - 1) / kThreads; ++u) {
        const int idx = tid + kThreads * u;
        if (idx < N) dst[idx] = map(idx) >= 0 ? r.v[u] : 0.0f;
    }
}

// ---- halo-1 tile staging -------------------------------------------------------------------
// LDS tile Z[ch][r][zq]: r = 0..TH+1 (image row ty0-1+r), zq = q+3 where q = 0..TW+1 is the halo-1
// column (image col tx0-1+q), so the interior starts 16-byte aligned at zq = 4.  Row stride TW+8.
// Thread t of each 128-thread half owns one POSITION: an interior 4-cell group (r, 4 cols) or one
// halo cell; the two halves split the channels.  Address math happens once per thread.
template <int TH, int TW>
struct TilePos {
    static constexpr int F4 = TW / 4, ROWS = TH + 2, RS = TW + 8;
    static constexpr int NINT = ROWS * F4, NPOS = NINT + 2 * ROWS;
    static_assert(NPOS <= 128, "positions must fit one 128-thread half");
    int r, q;          // halo-1 coordinates of the first element
    bool active, interior, vec;
    unsigned eo[4];    // plane offsets of the (up to) 4 elements, always dereferenceable
    bool ev[4];        // element is real (inside the image / pad-resolved)

    template <bool VEC>
    __device__ __forceinline__ void init(int tid, int ty0, int tx0, int H, int W, int pad) {
        const int p = tid & 127;
        active = p < NPOS;
        interior = p < NINT;
        if (interior) {
            r = p / F4;
            q = 1 + 4 * (p % F4);
        } else {
            const int h = p - NINT;
            r = (h >> 1) % ROWS;
            q = (h & 1) ? TW + 1 : 0;
        }
        const int sy = nca_pad_index(ty0 - 1 + r, H, pad);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sx = nca_pad_index(tx0 - 1 + q + j, W, pad);
            ev[j] = active && sy >= 0 && sx >= 0 && (interior || j == 0);
            eo[j] = ev[j] ? (unsigned)(sy * W + sx) : 0u;
        }
        vec = VEC && interior && ev[0] && (tx0 - 1 + q + 3 < W);
    }
};

// Loads planes clamp(ch0 + c - sub, 0, C-1), c = 0..CPH-1, of this thread's position (the caller
// zeroes/ignores slots whose unclamped index is out of range).  `base` is wave-uniform; offsets are
// 32-bit (one sample's planes stay below 2^31 elements) so the loads take the saddr+voffset form.
template <int CPH, typename PosT>
__device__ __forceinline__ void pos_load(const PosT& ps, const float* __restrict__ base, unsigned plane, int ch0, int sub,
                                         int C, float (&v)[CPH][4]) {
    if (ps.vec) {
#pragma unroll
        for (int c = 0; c < CPH; ++c) {
            const unsigned ch = (unsigned)min(max(ch0 + c - sub, 0), C - 1);
            const float4 t = *reinterpret_cast<const float4*>(base + (ch * plane + ps.eo[0]));
            v[c][0] = t.x; v[c][1] = t.y; v[c][2] = t.z; v[c][3] = t.w;
        }
    } else {  // halo cells, image-edge overhang, unaligned rows: per-element (masked elements re-read eo = 0)
#pragma unroll
        for (int c = 0; c < CPH; ++c) {
            const unsigned ch = (unsigned)min(max(ch0 + c - sub, 0), C - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[c][j] = base[ch * plane + ps.eo[j]];
        }
    }
}

// bf16 storage (ncahip_dynca_*_bf16): the same positions, 2-byte elements.  Raw bits are kept until staging consumes
// them (converting at load time would wait for the load): v4 holds the 8-byte group of the vector path, v1 the four
// halfwords of the per-element path.
typedef unsigned nca_u32x2 __attribute__((ext_vector_type(2)));
template <int CPH>
struct RawB16 {
    unsigned v1[CPH][4];   // vector path: [0..1] = the 8-byte group; per-element path: one halfword each
};
template <int CPH, typename PosT>
__device__ __forceinline__ void pos_load_b16(const PosT& ps, const uint16_t* __restrict__ base, unsigned plane, int ch0, int C,
                                             RawB16<CPH>& r) {
    if (ps.vec) {
#pragma unroll
        for (int c = 0; c < CPH; ++c) {
            const unsigned ch = (unsigned)min(ch0 + c, C - 1);
            const nca_u32x2 t = *reinterpret_cast<const nca_u32x2*>(base + (ch * plane + ps.eo[0]));
            r.v1[c][0] = t[0];
            r.v1[c][1] = t[1];
        }
    } else {
#pragma unroll
        for (int c = 0; c < CPH; ++c) {
            const unsigned ch = (unsigned)min(ch0 + c, C - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) r.v1[c][j] = base[ch * plane + ps.eo[j]];
        }
    }
}
__device__ __forceinline__ float nca_b16_to_f32(unsigned bits16) { return __uint_as_float(bits16 << 16); }
// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint16_t nca_f32_to_b16(float v) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return (uint16_t)(__builtin_bit_cast(unsigned, __builtin_convertvector(f2{v, 0.0f}, b2)) & 0xffffu);
}

template <typename PosT>
__device__ __forceinline__ void pos_store(const PosT& ps, float* __restrict__ Zc /* &Z[ch][0][0] */, const float (&v)[4]) {
    float* const d = Zc + ps.r * PosT::RS + ps.q + 3;
    if (ps.interior) *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]};
    else d[0] = v[0];
}

// =========================================================================================
// DyNCA step (ConditioneDyNCA/models/dynca.py:117-138)
// =========================================================================================
template <int CP, int FC, bool HAS_COND, int TH, int TW, int NT>
struct DyncaCfg {
    static constexpr int K1S = CP + (HAS_COND ? 1 : 0);  // k-steps of layer 1 (4 inputs each)
    static constexpr int M1T = FC / 16;                  // 16-row output tiles of layer 1
    static constexpr int K2S = FC / 4;
    static constexpr int M2T = (CP + 15) / 16;
    static constexpr int ROWS = TH + 2, RS = TW + 8;
    static constexpr int CS = lds_cs(ROWS, RS);
    static constexpr int TWN = TW / 16;
    static constexpr int NTILES16 = TH * TW / 16;
    static constexpr int ITERS = NTILES16 / (4 * NT);
    static constexpr int CPH = CP / 2;
    // LDS carve (floats)
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + M1T * K1S * 64;
    static constexpr int OFF_B1 = OFF_W2 + M2T * K2S * 64;
    static constexpr int OFF_B2 = OFF_B1 + FC;
    static constexpr int OFF_Z = OFF_B2 + M2T * 16;
    static constexpr int OFF_MK = OFF_Z + CP * CS;
    static constexpr int OFF_CN = OFF_MK + TH * TW;
    static constexpr int LDS_FLOATS = OFF_CN + (HAS_COND ? 4 * TH * TW : 0);
    // two-scale perception: coarse-level perception tile [4*CP planes][TH/2 + 2 rows][PCR], halo 1 (index-clamped: bilinear)
    static constexpr int PCR = TW / 2 + 4, PCS = (TH / 2 + 2) * PCR;
    static constexpr int OFF_PC = LDS_FLOATS;
    static constexpr int LDS_FLOATS_MS = OFF_PC + 4 * CP * PCS;
    // backward variant: the transposed operands (W2^T for dh, W1^T for dL/dy) are 16-byte reads of the FORWARD images (see the
    // kernel), so the backward needs no weight images of its own.  dL/dy is produced in tiles of 16 rows = 4 channels x 4
    // filters (row i <-> channel 4mj + (i & 3), filter i >> 2): MJ = CP / 4 tiles, no conditioning rows (dynca.py:123).
    static constexpr int MJ = CP / 4;
    static constexpr int LDS_FLOATS_BWD = OFF_CN;                     // the backward reads the conditioning straight from memory (no CN tile)
    static constexpr int TBS = M2T == 1 ? 20 : 36;                    // transposition tiles [16 cells][TBS] (16-byte rows)
    static constexpr int OFF_TB = LDS_FLOATS_BWD;                     // fused dW2: per wave NT tiles
    static constexpr int LDS_FLOATS_BWD_W2 = OFF_TB + 4 * NT * 16 * TBS;
    static_assert(FC % 16 == 0 && CP % 4 == 0 && TW % 16 == 0, "shape");
    static_assert(NTILES16 % (4 * NT) == 0, "tile must split evenly over 4 waves x NT");
    static_assert(OFF_Z % 4 == 0 && CS % 4 == 0 && TH * TW == kThreads, "16-byte carve; one cell per thread");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static constexpr bool kBwdFits = LDS_FLOATS_BWD * 4 <= 160 * 1024;   // checked where the backward variant is launched
};

// BWD = true: the backward data path of the same step (autograd through dynca.py:117-138).  Recomputes the
// hidden layer, then dh = (W2^T (G*mask)) * 1[h>0] and dL/dy = W1^T dh on MFMA (accumulator tile == next B
// operand, as in the forward); writes relu(h), dh and dL/dy[:4C].  The two weight-gradient GEMMs
// (dW2 = (G*mask) h^T, dW1 = dh y^T, K = all cells) are plain library GEMMs on those buffers.
template <int CP, int FC, bool HAS_COND, int TH, int TW, int NT, bool VEC, bool BWD = false, bool B16 = false, bool ACC = false, bool W2F = false,
          bool MS = false>
__global__ __launch_bounds__(kThreads, (CP > 16 || MS) ? 1 : 2) void dynca_step_fwd_kernel(const NcaDyncaArgs a) {   // CP > 16: 140 KB of LDS -> one workgroup per CU anyway: 512 registers
    static_assert(!MS || (!B16 && !ACC && TH % 2 == 0 && TW % 2 == 0 && (!BWD || W2F)), "the two-scale step: fp32, one hidden slice");
    static_assert(!W2F || BWD, "fused dW2 is an option of the backward kernel");
    static_assert(!ACC || !B16, "accumulating passes (fc slices beyond the first) read fp32 partial results");
    using K = DyncaCfg<CP, FC, HAS_COND, TH, TW, NT>;
    using Pos = TilePos<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const W1L = smem + K::OFF_W1;
    float* const W2L = smem + K::OFF_W2;
    float* const B1L = smem + K::OFF_B1;
    float* const B2L = smem + K::OFF_B2;
    float* const Z = smem + K::OFF_Z;
    float* const MK = smem + K::OFF_MK;
    float* const CN = smem + K::OFF_CN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ci = lane & 15;
    const int C = a.C, H = a.H, W = a.W, fc = a.fc, CC = a.c_cond;
    const int K1 = 4 * C + CC;

    // ---- A-operand weight images, once per workgroup -------------------------------------
    auto map_w1 = [&](int idx) -> long {
        const int l = idx & 63, s = (idx >> 6) % K::K1S, m = (idx >> 6) / K::K1S;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        if (o >= fc) return -1;
        if (s < CP) {  // k-step s = 4c'+f : channel 4c'+g, filter f (0 id, 1 sobel_x, 2 sobel_y, 3 lap)
            const int ch = (s & ~3) + gg;
            return ch < C ? (long)o * K1 + (s & 3) * C + ch : -1;  // blocked [x|Sx|Sy|L], dynca.py:92-95
        }
        return gg < CC ? (long)o * K1 + 4 * C + gg : -1;  // conditioning k-step
    };
    auto map_w2 = [&](int idx) -> long {
        const int l = idx & 63, s = (idx >> 6) % K::K2S, m = (idx >> 6) / K::K2S;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int k = 16 * (s >> 2) + 4 * gg + (s & 3);
        return (o < C && k < fc) ? (long)o * (a.w2_ld ? a.w2_ld : fc) + k : -1;
    };
    auto map_b1 = [&](int idx) -> long { return idx < fc ? idx : -1; };
    auto map_b2 = [&](int idx) -> long { return (idx < C && !ACC) ? idx : -1; };
    {
        FillV<K::M1T * K::K1S * 64> f1;
        FillV<K::M2T * K::K2S * 64> f2;
        FillV<FC> f3;
        FillV<K::M2T * 16> f4;
        fill_issue(f1, a.w1, tid, map_w1);
        fill_issue(f2, a.w2, tid, map_w2);
        fill_issue(f3, a.b1, tid, map_b1);
        fill_issue(f4, a.b2, tid, map_b2);
        fill_commit(f1, W1L, tid, map_w1);
        fill_commit(f2, W2L, tid, map_w2);
        fill_commit(f3, B1L, tid, map_b1);
        fill_commit(f4, B2L, tid, map_b2);
    }
    // Transposed operands of the backward, read from the forward images with ONE 16-byte LDS read each (the 64 lanes of a read
    // cover 1 KiB contiguously: conflict-free):
    //   W2^T, k-steps s = 0..3 of (m, m2):  W2[ch = 16 m2 + 4g + s][h = 16 m + ci]  = W2L[(m2*K2S + 4m + (ci&3))*64 + (ci>>2)*16 + 4g + s]
    //   W1^T, k-steps r = 0..3 of (m, mj):  W1[h = 16 m + 4g + r][row ci of tile mj] = W1L[(m*K1S + 4mj + (ci>>2))*64 + (ci&3)*16 + 4g + r]
    // with row ci of dL/dy tile mj standing for channel 4mj + (ci & 3), filter ci >> 2.
    const int w2t_lane = (ci & 3) * 64 + (ci >> 2) * 16 + 4 * g;
    const int w1t_lane = (ci >> 2) * 64 + (ci & 3) * 16 + 4 * g;

    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = a.B * tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    // fused dW2 (W2F): dW2[ch][hid] = sum_cells dO[ch][cell] * h[hid][cell] accumulated over the whole launch, one 16x16
    // tile per hidden tile m (rows = channels 4g+r, column = hidden 16m+ci), and db2 = sum_cells dO
    f32x4 w2acc[W2F ? K::M1T : 1][K::M2T];
    float b2acc[K::M2T][4];
#pragma unroll
    for (int m2 = 0; m2 < K::M2T; ++m2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) b2acc[m2][r] = 0.f;
#pragma unroll
        for (int m = 0; m < (W2F ? K::M1T : 1); ++m) w2acc[m][m2] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float* const TBW = smem + K::OFF_TB + wave * (NT * 16 * K::TBS);

    for (NcaTileWalk tw = nca_tile_walk(ntiles); tw.t < tw.end; tw.t += tw.stride) {
        const int txi = tw.t % tiles_x, tyi = (tw.t / tiles_x) % tiles_y, b = tw.t / (tiles_x * tiles_y);
        const int ty0 = tyi * TH, tx0 = txi * TW;
        const float* const xb = a.x_in + (size_t)b * C * plane;                                    // f32 storage
        const uint16_t* const xb16 = reinterpret_cast<const uint16_t*>(a.x_in) + (size_t)b * C * plane;   // bf16 storage

        // ---- issue every global load of the tile, then retire the previous tile ---------
        // (staging index made opaque per tile: its derived coordinates are recomputed here rather than
        //  hoisted out of the tile loop and kept live -- or spilled -- across the MFMA phase)
        int st = tid;
        asm volatile("" : "+v"(st));
        const int half = st >> 7;
        Pos ps;
        ps.template init<VEC>(st, ty0, tx0, H, W, a.pad_mode);
        float xv[K::CPH][4];
        RawB16<B16 ? K::CPH : 1> xr;
        if constexpr (B16) pos_load_b16<K::CPH>(ps, xb16, (unsigned)plane, half * K::CPH, C, xr);
        else pos_load<K::CPH>(ps, xb, (unsigned)plane, half * K::CPH, 0, C, xv);
        const int cr = st / TW, cq = st % TW, cgy = ty0 + cr, cgx = tx0 + cq;  // this thread's cell
        const bool cin = cgy < H && cgx < W;
        const size_t cell = (size_t)b * plane + (cin ? (size_t)cgy * W + cgx : 0);
        float uu = 0.0f, cnv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (a.u) {   // explicit uniforms, or bit-packed masks: bit 1 -> floor(1 + rate) = 1, bit 0 -> floor(0 + rate) = 0 (0 <= rate < 1)
            if (a.u_bits) uu = ((reinterpret_cast<const uint32_t*>(a.u)[cell >> 5] >> (unsigned)(cell & 31)) & 1u) ? 1.0f : 0.0f;
            else uu = a.u[cell];
        }
        if (HAS_COND && !BWD) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                cnv[cc] = a.cond[((size_t)b * CC + min(cc, CC - 1)) * plane + (cell - (size_t)b * plane)];
        }
        __syncthreads();  // previous tile fully consumed (also orders the weight image on iter 0)
        if (ps.active) {
#pragma unroll
            for (int c = 0; c < K::CPH; ++c) {
                const int ch = half * K::CPH + c;
                float v[4];
                if constexpr (B16) {   // widen now (exact): the loads have long landed
                    const unsigned lo = xr.v1[c][0], hi = xr.v1[c][1];
                    const float w4[4] = {__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u),
                                         __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[c][j] = ps.vec ? w4[j] : nca_b16_to_f32(xr.v1[c][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (ch < C && ps.ev[j]) ? xv[c][j] : 0.0f;
                pos_store(ps, Z + ch * K::CS, v);
            }
        }
        if (!a.u) uu = nca_philox_cell(a.seed, a.step, cell);
        MK[st] = cin ? floorf(uu + a.rate) : 0.0f;  // dynca.py:131
        if (HAS_COND && !BWD) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) CN[cc * TH * TW + st] = (cin && cc < CC) ? cnv[cc] : 0.0f;
        }
        if constexpr (MS) {
            // coarse perception tile: rows ty0/2 - 1 .. ty0/2 + TH/2, columns tx0/2 - 1 .. tx0/2 + TW/2, indices clamped into the
            // coarse image (what bilinear up-sampling does at the border, whatever the pad mode of the stencil was)
            constexpr int NR = TH / 2 + 2, NC = TW / 2 + 2, PER_PLANE = NR * NC;
            const int Hc = H >> 1, Wc = W >> 1;
            const float* const pcb = a.pc + (size_t)b * 4 * C * Hc * Wc;
            float* const PCL = smem + (BWD ? K::LDS_FLOATS_BWD_W2 : K::OFF_PC);
            for (int i = st; i < 4 * CP * PER_PLANE; i += kThreads) {
                const int pl = i / PER_PLANE, rc = i - pl * PER_PLANE, r = rc / NC, c = rc - r * NC;
                const int f = pl / CP, ch = pl - f * CP;
                const int sy = min(max((ty0 >> 1) - 1 + r, 0), Hc - 1), sx = min(max((tx0 >> 1) - 1 + c, 0), Wc - 1);
                PCL[pl * K::PCS + r * K::PCR + c] = ch < C ? pcb[((size_t)(f * C + ch) * Hc + sy) * Wc + sx] : 0.0f;
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int it = 0; it < K::ITERS; ++it) {
            int r0[NT], q0[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int j = (it * 4 + wave) * NT + n;
                r0[n] = j / K::TWN;
                q0[n] = (j % K::TWN) * 16 + ci;
            }
            // ---- perception: B operands of layer 1 ---------------------------------------
            float P[NT][K::K1S];
#pragma unroll
            for (int cq4 = 0; cq4 < CP / 4; ++cq4) {
                const float* const zc = Z + (4 * cq4 + g) * K::CS + 3;
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float nb[3][3];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) nb[dy][dx] = zc[(r0[n] + dy) * K::RS + q0[n] + dx];
                    P[n][4 * cq4 + 0] = nb[1][1];
                    P[n][4 * cq4 + 1] = nca_sobel_x(nb);
                    P[n][4 * cq4 + 2] = nca_sobel_y(nb);
                    P[n][4 * cq4 + 3] = nca_laplacian(nb);
                    if constexpr (MS) {
                        // + bilinear x2 up-sampling (align_corners = False) of the coarse perception, then the mean over the two
                        // scales (dynca.py:98, :105-110).  Fine row r = 2k: rows (k-1, k) with lambdas (0.25, 0.75); r = 2k+1:
                        // (k, k+1) with (0.75, 0.25); the tile has a coarse halo of 1, indices clamped at staging.
                        const int fr = r0[n], fq = q0[n];
                        const int kr = (fr >> 1) + ((fr & 1) ? 1 : 0), kq = (fq >> 1) + ((fq & 1) ? 1 : 0);   // first of the two coarse rows / cols (tile coordinates incl. halo)
                        const float h1 = (fr & 1) ? 0.25f : 0.75f, w1 = (fq & 1) ? 0.25f : 0.75f, h0 = 1.0f - h1, w0 = 1.0f - w1;
                        const float* const pcp = smem + (BWD ? K::LDS_FLOATS_BWD_W2 : K::OFF_PC) + (4 * cq4 + g) * K::PCS + kr * K::PCR + kq;
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            const float* const q = pcp + f * CP * K::PCS;
                            P[n][4 * cq4 + f] = nca_up2_blend(P[n][4 * cq4 + f], q[0], q[1], q[K::PCR], q[K::PCR + 1], h0, h1, w0, w1);
                        }
                    }
                }
            }
            if (HAS_COND) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (BWD) {   // conditioning channel g of the lane's cell, straight from memory (64-byte segments per lane group)
                        const int gy = min(ty0 + r0[n], H - 1), gx = min(tx0 + q0[n], W - 1);
                        const float cv = a.cond[((size_t)b * CC + min(g, CC - 1)) * plane + (size_t)gy * W + gx];
                        P[n][CP] = (g < CC && ty0 + r0[n] < H && tx0 + q0[n] < W) ? cv : 0.0f;
                    } else P[n][CP] = CN[g * TH * TW + r0[n] * TW + q0[n]];
                }
            }
            if constexpr (BWD) {
                // ---- backward data path -----------------------------------------------------------------
                float dO[NT][K::M2T][4];   // dL/d(out) = G * mask, accumulator layout (channel 16 m2 + 4g + r, cell ci)
                bool live[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int gy = ty0 + r0[n], gx = tx0 + q0[n];
                    live[n] = gy < H && gx < W;
                    const float mk = MK[r0[n] * TW + q0[n]];
                    const float* const gb = a.g_next + (size_t)b * C * plane + (live[n] ? (size_t)gy * W + gx : 0);
#pragma unroll
                    for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ch = 16 * m2 + 4 * g + r;
                            const float gv = gb[(size_t)min(ch, C - 1) * plane];
                            dO[n][m2][r] = (live[n] && ch < C) ? gv * mk : 0.0f;
                        }
                }
                // The perception just recomputed is also the B operand of the layer-1 weight-gradient product (dW1 = dh y^T, in
                // gram_rows_kernel): written out here -- slot 4c'+f of lane (g, cell) is channel 4c'+g, filter f, i.e. row
                // f*C + 4c'+g of perceive_torch's output; lane part of the address once per n, row part a scalar offset -- instead
                // of a perception (and, two-scale, a combine) launch of its own that re-reads x_t.
                if constexpr (!ACC) {
                    if (a.ybuf) {
                        const unsigned pl4 = (unsigned)plane * 4u;
                        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.ybuf + (size_t)b * 4 * C * plane, 0, -1, 0x00020000);
#pragma unroll
                        for (int n = 0; n < NT; ++n) {
                            if (!live[n]) continue;
                            const unsigned vy = (unsigned)((ty0 + r0[n]) * W + tx0 + q0[n]) * 4u + (unsigned)g * pl4;
#pragma unroll
                            for (int cq4 = 0; cq4 < CP / 4; ++cq4)
                                if (4 * cq4 + g < C) {
#pragma unroll
                                    for (int f = 0; f < 4; ++f)
                                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(P[n][4 * cq4 + f]), ry, (int)vy,
                                                                              (int)((unsigned)(f * C + 4 * cq4) * pl4), 0);
                                }
                        }
                    }
                }
                // Output addressing: the lane's part (cell, lane-dependent row group) is ONE vector offset per n, the remaining row
                // index of each store a scalar offset -- per-store 64-bit vector address arithmetic was ~1300 VALU instructions per
                // pass, in the same issue slots as the exact-f32 MFMAs.  (Launcher: fc*H*W*4 and 4C*H*W*4 below 4 GiB.)
                const unsigned plane4 = (unsigned)plane * 4u;
                const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hbuf + (size_t)b * fc * plane, 0, -1, 0x00020000);
                const __amdgpu_buffer_rsrc_t rdh = __builtin_amdgcn_make_buffer_rsrc(a.dhbuf + (size_t)b * fc * plane, 0, -1, 0x00020000);
                const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(a.dybuf + (size_t)b * 4 * C * plane, 0, -1, 0x00020000);
                unsigned vo[NT], voy[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const unsigned cellb = live[n] ? (unsigned)((ty0 + r0[n]) * W + tx0 + q0[n]) * 4u : 0u;
                    vo[n] = cellb + (unsigned)(4 * g) * plane4;             // h / dh rows 16m + 4g + r
                    voy[n] = cellb + (unsigned)(g * C) * plane4;            // dL/dy rows g*C + (4mj + r): filter g, channel 4mj + r
                }
                const int lim_h = fc - 4 * g;   // row 16m + 4g + r exists <=> 16m + r < lim_h
                // fused dW2: the A operands dO[ch = ci][cell 4s+g] of every n, transposed once per pass through the wave's
                // LDS tiles (lane (g,ci) writes its four channel rows of cell ci with one 16-byte store per channel tile)
                float doT[W2F ? NT : 1][K::M2T][4];
                if constexpr (W2F) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2) {
                            *reinterpret_cast<f32x4*>(TBW + n * 16 * K::TBS + ci * K::TBS + 16 * m2 + 4 * g) =
                                f32x4{dO[n][m2][0], dO[n][m2][1], dO[n][m2][2], dO[n][m2][3]};
#pragma unroll
                            for (int r = 0; r < 4; ++r) b2acc[m2][r] += dO[n][m2][r];
                        }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
                            for (int s_ = 0; s_ < 4; ++s_) doT[n][m2][s_] = TBW[n * 16 * K::TBS + (4 * s_ + g) * K::TBS + 16 * m2 + ci];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                }
                f32x4 dY[K::MJ][NT];
#pragma unroll
                for (int mj = 0; mj < K::MJ; ++mj)
#pragma unroll
                    for (int n = 0; n < NT; ++n) dY[mj][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                // A operands are read from LDS a whole section ahead of their MFMAs (one wave per SIMD: a read issued right
                // before its use costs the full LDS round trip each time): layer-1 operands of hidden tile m+1 during the
                // dL/dy products of tile m, the transposed operands of tile m during its layer-1 chain.  The fences keep the
                // compiler from sinking the reads back to their uses.
                float wa1[K::K1S];
                f32x4 wt2[K::M2T], wt1[K::MJ];
                f32x4 bias1;
                auto fetch1 = [&](int m) {
                    const float* const w1m = W1L + m * K::K1S * 64 + lane;
#pragma unroll
                    for (int s = 0; s < K::K1S; ++s) wa1[s] = w1m[s * 64];
                    bias1 = *reinterpret_cast<const f32x4*>(B1L + 16 * m + 4 * g);
                };
                auto fetch_t = [&](int m) {
#pragma unroll
                    for (int m2 = 0; m2 < K::M2T; ++m2) wt2[m2] = *reinterpret_cast<const f32x4*>(W2L + (m2 * K::K2S + 4 * m) * 64 + w2t_lane);
#pragma unroll
                    for (int mj = 0; mj < K::MJ; ++mj) wt1[mj] = *reinterpret_cast<const f32x4*>(W1L + (m * K::K1S + 4 * mj) * 64 + w1t_lane);
                };
                // later slices of a wide hidden layer add to the dL/dy the earlier launches wrote: those partial sums are requested
                // here, a whole MFMA phase ahead of their use (one wave per SIMD: a load issued next to its use costs its full latency)
                float prev[ACC ? NT : 1][ACC ? K::MJ : 1][4];
                if constexpr (ACC) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int mj = 0; mj < K::MJ; ++mj)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                prev[n][mj][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                    rdy, (int)voy[n], (int)((unsigned)min(4 * mj + r, C - 1) * plane4), 0));
                }
                fetch1(0);
                float hT[W2F ? NT : 1][4];
#pragma unroll (W2F ? K::M1T : 1)
                for (int m = 0; m < K::M1T; ++m) {
                    f32x4 acc1[NT], dacc[NT];
#pragma unroll
                    for (int n = 0; n < NT; ++n) { acc1[n] = bias1; dacc[n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
                    fetch_t(m);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 0; s < K::K1S; ++s) {
#pragma unroll
                        for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(wa1[s], P[n][s], acc1[n]);
                    }
#pragma unroll
                    for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
#pragma unroll
                            for (int n = 0; n < NT; ++n) dacc[n] = nca_mfma(wt2[m2][s], dO[n][m2][s], dacc[n]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    if (m + 1 < K::M1T) fetch1(m + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float hv = fmaxf(acc1[n][r], 0.0f);
                            const float dv = acc1[n][r] > 0.0f ? dacc[n][r] : 0.0f;   // relu' = 0 at exactly 0
                            dacc[n][r] = dv;
                            if (live[n] && 16 * m + r < lim_h) {
                                const int so = (int)((unsigned)(16 * m + r) * plane4);
                                if constexpr (!W2F) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), rh, (int)vo[n], so, 0);
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rdh, (int)vo[n], so, 0);
                            }
                            if constexpr (W2F) acc1[n][r] = (live[n] && 16 * m + r < lim_h) ? hv : 0.0f;   // h, zero outside the image / fc
                        }
                        if constexpr (W2F) *reinterpret_cast<f32x4*>(TBW + n * 16 * K::TBS + ci * K::TBS + 4 * g) = acc1[n];
                    }
                    if constexpr (W2F) {   // B operands h[cell 4s+g][hid ci]: requested here, consumed after the dL/dy products
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int s_ = 0; s_ < 4; ++s_) hT[n][s_] = TBW[n * 16 * K::TBS + (4 * s_ + g) * K::TBS + ci];
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mj = 0; mj < K::MJ; ++mj) {
#pragma unroll
                            for (int n = 0; n < NT; ++n) dY[mj][n] = nca_mfma(wt1[mj][r], dacc[n][r], dY[mj][n]);
                        }
                    if constexpr (W2F) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
                                for (int s_ = 0; s_ < 4; ++s_) w2acc[m][m2] = nca_mfma(doT[n][m2][s_], hT[n][s_], w2acc[m][m2]);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tiles are rewritten by the next hidden tile
                    }
                }
                // dL/dy out: tile mj, register r of lane (g, cell) = channel 4mj + r, filter g -> plane g*C + 4mj + r.  Later
                // slices of a wide hidden layer (ACC) add to what the earlier launches wrote.
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (!live[n]) continue;
#pragma unroll
                    for (int mj = 0; mj < K::MJ; ++mj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (4 * mj + r < C) {
                                float v = dY[mj][n][r];
                                if constexpr (ACC) v += prev[n][mj][r];
                                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rdy, (int)voy[n],
                                                                      (int)((unsigned)(4 * mj + r) * plane4), 0);
                            }
                        }
                }
            } else {
                // ---- MLP on MFMA: layer 1 streamed tile-by-tile into layer 2 -----------------
                // Issue order (see nca_cond_tile.h, mlp_tile_regs): the A operands of hidden tile m+1 are read from LDS
                // while tile m's MFMA chain runs (an LDS read next to its use stalls the in-order pipe for the whole
                // latency), and the ReLUs are issued as ONE fenced group per tile (a VALU instruction inside an f32 MFMA
                // stream drains the matrix pipe: ~25 cycles each when scattered).
                f32x4 acc2[K::M2T][NT];
    #pragma unroll
                for (int m2 = 0; m2 < K::M2T; ++m2) {
                    const f32x4 bias = *reinterpret_cast<const f32x4*>(B2L + 16 * m2 + 4 * g);
    #pragma unroll
                    for (int n = 0; n < NT; ++n) acc2[m2][n] = bias;
                }
                // later fc slice: the partial result of the earlier launches is requested now, a whole MLP ahead of its use
                float xprev[ACC ? NT : 1][K::M2T][4];
                if constexpr (ACC) {
    #pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        const int gy = min(ty0 + r0[n], H - 1), gx = min(tx0 + q0[n], W - 1);
                        const float* const ob = a.x_out + (size_t)b * C * plane + (size_t)gy * W + gx;
    #pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2)
    #pragma unroll
                            for (int r = 0; r < 4; ++r) xprev[n][m2][r] = ob[(size_t)min(16 * m2 + 4 * g + r, C - 1) * plane];
                    }
                }
                float wa1[K::K1S], wa2[4][K::M2T];
                f32x4 bias1;
                auto fetch = [&](int m) {
                    const float* const w1m = W1L + m * K::K1S * 64 + lane;
    #pragma unroll
                    for (int s = 0; s < K::K1S; ++s) wa1[s] = w1m[s * 64];
                    const float* const w2m = W2L + (4 * m) * 64 + lane;
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2) wa2[r][m2] = w2m[(m2 * K::K2S + r) * 64];
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
                    float w2c[4][K::M2T];
    #pragma unroll
                    for (int r = 0; r < 4; ++r)
    #pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2) w2c[r][m2] = wa2[r][m2];
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
                        for (int m2 = 0; m2 < K::M2T; ++m2)
    #pragma unroll
                            for (int n = 0; n < NT; ++n) acc2[m2][n] = nca_mfma(w2c[r][m2], h[n][r], acc2[m2][n]);
                }
                // ---- residual + stochastic mask (dynca.py:131-133) ---------------------------
    #pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int gy = ty0 + r0[n], gx = tx0 + q0[n];
                    if (gy < H && gx < W) {
                        const float mk = MK[r0[n] * TW + q0[n]];
                        const size_t o0 = (size_t)b * C * plane + (size_t)gy * W + gx;
                        const float* const ob = a.x_out + o0;
                        (void)ob;
                        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.x_out + (size_t)b * C * plane, 0, -1, 0x00020000);
                        const unsigned ooff = (unsigned)(gy * W + gx) * 4u;
                        uint16_t* const ob16 = reinterpret_cast<uint16_t*>(a.x_out) + o0;
    #pragma unroll
                        for (int m2 = 0; m2 < K::M2T; ++m2)
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int ch = 16 * m2 + 4 * g + r;
                                if (ch < C) {
                                    float xo = Z[ch * K::CS + (r0[n] + 1) * K::RS + q0[n] + 4];
                                    if constexpr (ACC) xo = xprev[n][m2][r];           // later fc slice: add to the partial result
                                    const float xn = xo + acc2[m2][n][r] * mk;
                                    if constexpr (B16) ob16[ch * plane] = nca_f32_to_b16(xn);
                                    else   // write-through (sc1): no dirty lines left for the end-of-kernel L2 write-back
                                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(xn), orsrc, (int)((unsigned)ch * (unsigned)plane * 4u + ooff), 0, 16);
                                }
                            }
                    }
                }

            }
        }
    }
    if constexpr (W2F) {
        // the four waves' dW2 / db2 partials are summed through LDS (weight images and tiles are dead) into the workgroup's slab
        constexpr int CH = 16 * K::M2T, SW = CH * FC + CH;
        __syncthreads();
        float* const sw = smem + wave * SW;
#pragma unroll
        for (int m = 0; m < K::M1T; ++m)
#pragma unroll
            for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
                for (int r = 0; r < 4; ++r) sw[(16 * m2 + 4 * g + r) * FC + 16 * m + ci] = w2acc[m][m2][r];
#pragma unroll
        for (int m2 = 0; m2 < K::M2T; ++m2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = b2acc[m2][r];
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d);
                if (ci == 0) sw[CH * FC + 16 * m2 + 4 * g + r] = v;
            }
        __syncthreads();
        float* const slab = a.gw2_ws + (size_t)blockIdx.x * ((size_t)C * fc + C);
        for (int i = tid; i < C * fc; i += kThreads) {
            const int ch = i / fc, hid = i - ch * fc, o = ch * FC + hid;
            slab[i] = (smem[o] + smem[SW + o]) + (smem[2 * SW + o] + smem[3 * SW + o]);
        }
        if (tid < C) {
            const int o = CH * FC + tid;
            slab[(size_t)C * fc + tid] = (smem[o] + smem[SW + o]) + (smem[2 * SW + o] + smem[3 * SW + o]);
        }
    }
}

// =========================================================================================
// ConditionedNCA step (EncoderConditioning/nca.py:181-195), pending-life-mask protocol
// =========================================================================================
template <int CP, int TH, int TW, int NT>
struct CondCfg {
    static constexpr int HID = 64;
    static constexpr int K1S = 3 * CP / 4;
    static constexpr int M3T = (CP + 15) / 16;
    static constexpr int ROWS = TH + 2, RS = TW + 8;
    static constexpr int CS = lds_cs(ROWS, RS);
    static constexpr int PNS = TW + 2;                // pre-mask tile, halo 1
    static constexpr int A3R = TH + 6, A3S = TW + 6;  // alpha', halo 3
    static constexpr int L2R = TH + 4, L2S = TW + 4;  // life / resolved alpha, halo 2
    static constexpr int NA3 = A3R * A3S, NL2 = L2R * L2S, NPN = ROWS * PNS;
    static constexpr int KA3 = (NA3 + kThreads - 1) / kThreads, KL2 = (NL2 + kThreads - 1) / kThreads,
                         KPN = (NPN + kThreads - 1) / kThreads;
    static constexpr int WPS = 28;  // 27 taps padded
    static constexpr int TWN = TW / 16;
    static constexpr int NTILES16 = TH * TW / 16;
    static constexpr int ITERS = NTILES16 / (4 * NT);
    static constexpr int CPH = CP / 2;
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + 4 * K1S * 64;
    static constexpr int OFF_W3 = OFF_W2 + 4 * 16 * 64;
    static constexpr int OFF_B1 = OFF_W3 + M3T * 16 * 64;
    static constexpr int OFF_B2 = OFF_B1 + HID;
    static constexpr int OFF_WP = OFF_B2 + HID;
    static constexpr int OFF_Z = OFF_WP + CP * WPS;
    static constexpr int OFF_A3 = OFF_Z + CP * CS;
    static constexpr int OFF_LIFE = OFF_A3 + NA3;
    static constexpr int OFF_A2 = OFF_LIFE + NL2;
    static constexpr int OFF_PN = OFF_A2 + NL2;
    static constexpr int OFF_MK = OFF_PN + NPN;
    static constexpr int LDS_FLOATS = OFF_MK + TH * TW;
    static_assert(CP % 4 == 0 && TW % 16 == 0 && NTILES16 % (4 * NT) == 0, "shape");
    static_assert(OFF_Z % 4 == 0 && OFF_WP % 4 == 0 && CS % 4 == 0 && TH * TW == kThreads, "16-byte carve");
};

__device__ __forceinline__ float nca_max3x3(const float* p, int stride) {
    float m = fmaxf(fmaxf(p[-stride - 1], p[-stride]), p[-stride + 1]);
    m = fmaxf(m, fmaxf(fmaxf(p[-1], p[0]), p[1]));
    return fmaxf(m, fmaxf(fmaxf(p[stride - 1], p[stride]), p[stride + 1]));
}

template <int CP, int TH, int TW, int NT, bool VEC>
__global__ __launch_bounds__(kThreads, CP > 16 ? 1 : 2) void cond_step_fwd_kernel(const NcaCondArgs a) {   // CP > 16: 84+ KB of LDS -> one workgroup per CU anyway: 512 registers per lane
    using K = CondCfg<CP, TH, TW, NT>;
    using Pos = TilePos<TH, TW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const W1L = smem + K::OFF_W1;
    float* const W2L = smem + K::OFF_W2;
    float* const W3L = smem + K::OFF_W3;
    float* const B1L = smem + K::OFF_B1;
    float* const B2L = smem + K::OFF_B2;
    float* const WPL = smem + K::OFF_WP;
    float* const Z = smem + K::OFF_Z;
    float* const A3 = smem + K::OFF_A3;
    float* const LIFE = smem + K::OFF_LIFE;
    float* const A2 = smem + K::OFF_A2;
    float* const PN = smem + K::OFF_PN;
    float* const MK = smem + K::OFF_MK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, ci = lane & 15;
    const int C = a.C, H = a.H, W = a.W, hid = a.hidden;
    const int K1 = 3 * C;
    const int gch0 = C - a.goal_ch;  // first channel the goal encoding is added to (nca.py:199-203)
    const bool pending = a.pre_in != nullptr;
    const bool use_alive = a.alive_ch >= 0;
    const bool has_goal = a.goal_ch > 0;

    fill_image<4 * K::K1S * 64>(W1L, a.w1, tid, [&](int idx) -> long {
        const int l = idx & 63, s = (idx >> 6) % K::K1S, m = (idx >> 6) / K::K1S;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int ch = 4 * (s / 3) + gg, f = s % 3;  // k-step s = 3c'+f : channel 4c'+g, filter f
        return (ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;  // out[3c+f], nca.py:99-107
    });
    fill_image<4 * 16 * 64>(W2L, a.w2, tid, [&](int idx) -> long {
        const int l = idx & 63, s = (idx >> 6) % 16, m = (idx >> 6) / 16;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int k = 16 * (s >> 2) + 4 * gg + (s & 3);
        return (o < hid && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image<K::M3T * 16 * 64>(W3L, a.w3, tid, [&](int idx) -> long {
        const int l = idx & 63, s = (idx >> 6) % 16, m = (idx >> 6) / 16;
        const int gg = l >> 4, o = 16 * m + (l & 15);
        const int k = 16 * (s >> 2) + 4 * gg + (s & 3);
        return (o < C && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image<K::HID>(B1L, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image<K::HID>(B2L, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image<CP * K::WPS>(WPL, a.wp, tid, [&](int idx) -> long {
        const int ch = idx / K::WPS, j = idx % K::WPS;
        return (ch < C && j < 27) ? (long)ch * 27 + j : -1;  // [3c+f][3][3] == [c][f*9+tap]
    });

    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = a.B * tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;

    for (NcaTileWalk tw = nca_tile_walk(ntiles); tw.t < tw.end; tw.t += tw.stride) {
        const int txi = tw.t % tiles_x, tyi = (tw.t / tiles_x) % tiles_y, b = tw.t / (tiles_x * tiles_y);
        const int ty0 = tyi * TH, tx0 = txi * TW;
        const float* const xb = a.x_in + (size_t)b * C * plane;

        // ---- issue every global load of the tile ------------------------------------------
        // (staging index made opaque per tile so its derived coordinates are recomputed, not kept
        //  live -- or spilled -- across the MFMA phase)
        int st = tid;
        asm volatile("" : "+v"(st));
        const int half = st >> 7;
        // alpha' (halo 3), pre_in (halo 2), u: needed first
        float a3v[K::KA3];
        bool a3in[K::KA3];
        if (use_alive) {
#pragma unroll
            for (int k = 0; k < K::KA3; ++k) {
                const int idx = st + k * kThreads;
                const int q = idx % K::A3S, r = idx / K::A3S, gy = ty0 - 3 + r, gx = tx0 - 3 + q;
                a3in[k] = idx < K::NA3 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                a3v[k] = xb[a.alive_ch * plane + (a3in[k] ? (size_t)gy * W + gx : 0)];
            }
        }
        float l2v[K::KL2];
        bool l2in[K::KL2];
#pragma unroll
        for (int k = 0; k < K::KL2; ++k) {
            const int idx = st + k * kThreads;
            const int q = idx % K::L2S, r = idx / K::L2S, gy = ty0 - 2 + r, gx = tx0 - 2 + q;
            l2in[k] = idx < K::NL2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            l2v[k] = 1.0f;
            if (pending && use_alive) l2v[k] = (float)a.pre_in[(size_t)b * plane + (l2in[k] ? (size_t)gy * W + gx : 0)];
        }
        const int cr = st / TW, cq = st % TW, cgy = ty0 + cr, cgx = tx0 + cq;  // this thread's cell
        const bool cin = cgy < H && cgx < W;
        const size_t cell = (size_t)b * plane + (cin ? (size_t)cgy * W + cgx : 0);
        float uu = 0.0f;
        if (a.u) {   // explicit uniforms, or bit-packed masks: bit 1 -> clamp(0) < rate, bit 0 -> clamp(2) = 1 < rate never (rate <= 1)
            if (a.u_bits) uu = ((reinterpret_cast<const uint32_t*>(a.u)[cell >> 5] >> (unsigned)(cell & 31)) & 1u) ? 0.0f : 2.0f;
            else uu = a.u[cell];
        }
        // state tile + goal encoding (halo 1): one position per thread, channels split over the two halves
        Pos ps;
        ps.template init<VEC>(st, ty0, tx0, H, W, NCA_PAD_ZERO);
        float xv[K::CPH][4], gv[K::CPH][4];
        pos_load<K::CPH>(ps, xb, (unsigned)plane, half * K::CPH, 0, C, xv);
        if (has_goal)  // slot c holds goal plane (half*CPH + c) - gch0 (used only when that index is >= 0)
            pos_load<K::CPH>(ps, a.goal + (size_t)b * a.goal_ch * plane, (unsigned)plane, half * K::CPH, gch0, a.goal_ch, gv);

        __syncthreads();  // previous tile fully consumed
        // ---- S1: alpha' with -inf outside the image (= max_pool2d padding), pre_in -----------
        if (use_alive) {
#pragma unroll
            for (int k = 0; k < K::KA3; ++k) {
                const int idx = st + k * kThreads;
                if (idx < K::NA3) A3[idx] = a3in[k] ? a3v[k] : NCA_NEG_INF;
            }
        }
#pragma unroll
        for (int k = 0; k < K::KL2; ++k) {
            const int idx = st + k * kThreads;
            if (idx < K::NL2) LIFE[idx] = l2in[k] ? l2v[k] : 0.0f;
        }
        __syncthreads();
        // ---- S2: life = pre & post of the PREVIOUS step; resolved alpha (nca.py:191-194) ----
        if (use_alive) {
#pragma unroll
            for (int k = 0; k < K::KL2; ++k) {
                const int idx = st + k * kThreads;
                if (idx < K::NL2) {
                    const int q = idx % K::L2S, r = idx / K::L2S;
                    const float* const ac = A3 + (r + 1) * K::A3S + q + 1;
                    float life = l2in[k] ? l2v[k] : 0.0f, av = NCA_NEG_INF;
                    if (l2in[k]) {
                        if (pending) {
                            life = (life != 0.0f && nca_max3x3(ac, K::A3S) > a.thr) ? 1.0f : 0.0f;
                            av = nca_clamp(ac[0] * life, a.lo, a.hi);
                        } else {
                            av = ac[0];
                        }
                    }
                    LIFE[idx] = life;
                    A2[idx] = av;
                }
            }
        }
        __syncthreads();
        // ---- S3: pre-life mask of THIS step (halo 1), fire mask --------------------------
#pragma unroll
        for (int k = 0; k < K::KPN; ++k) {
            const int idx = st + k * kThreads;
            if (idx < K::NPN) {
                const int q = idx % K::PNS, r = idx / K::PNS, gy = ty0 - 1 + r, gx = tx0 - 1 + q;
                float pn = 0.0f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    pn = (!use_alive || nca_max3x3(A2 + (r + 1) * K::L2S + q + 1, K::L2S) > a.thr) ? 1.0f : 0.0f;
                    if (r >= 1 && r <= TH && q >= 1 && q <= TW)
                        a.pre_out[(size_t)b * plane + (size_t)gy * W + gx] = (uint8_t)pn;
                }
                PN[idx] = pn;
            }
        }
        if (!a.u) uu = nca_philox_cell(a.seed, a.step, cell);
        MK[st] = (cin && nca_clamp(uu, 0.0f, 1.0f) < a.fire_rate) ? 1.0f : 0.0f;  // nca.py:171-174
        __syncthreads();
        // ---- S4: z = x + goal * pre (nca.py:177) on halo 1, zero outside the image -------
        if (ps.active) {
            const float* const lf = LIFE + (ps.r + 1) * K::L2S + ps.q + 1;
            const float* const pnp = PN + ps.r * K::PNS + ps.q;
            float lfv[4], pnv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool rd = ps.interior || j == 0;
                lfv[j] = rd ? lf[j] : 0.0f;
                pnv[j] = rd ? pnp[j] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < K::CPH; ++c) {
                const int ch = half * K::CPH + c;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = 0.0f;
                    if (ch < C && ps.ev[j]) {
                        t = xv[c][j];
                        if (pending) t = nca_clamp(t * lfv[j], a.lo, a.hi);
                    }
                    v[j] = t;
                }
                if (has_goal && ch >= gch0 && ch < C) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ps.ev[j]) v[j] = fmaf(gv[c][j], pnv[j], v[j]);
                }
                pos_store(ps, Z + ch * K::CS, v);
            }
        }
        __syncthreads();

#pragma unroll 1
        for (int it = 0; it < K::ITERS; ++it) {
            int r0[NT], q0[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int j = (it * 4 + wave) * NT + n;
                r0[n] = j / K::TWN;
                q0[n] = (j % K::TWN) * 16 + ci;
            }
            // ---- learned depthwise perception (nca.py:99-107): P[3c'+f] for channel 4c'+g
            float P[NT][K::K1S];
#pragma unroll
            for (int cq4 = 0; cq4 < CP / 4; ++cq4) {
                const float* const zc = Z + (4 * cq4 + g) * K::CS + 3;
                float wt[28];
#pragma unroll
                for (int j4 = 0; j4 < 7; ++j4) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(WPL + (4 * cq4 + g) * K::WPS + 4 * j4);
                    wt[4 * j4 + 0] = w4[0]; wt[4 * j4 + 1] = w4[1]; wt[4 * j4 + 2] = w4[2]; wt[4 * j4 + 3] = w4[3];
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    float nb[9];
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) nb[3 * dy + dx] = zc[(r0[n] + dy) * K::RS + q0[n] + dx];
#pragma unroll
                    for (int f = 0; f < 3; ++f) {
                        float acc = 0.0f;
#pragma unroll
                        for (int t = 0; t < 9; ++t) acc = fmaf(wt[9 * f + t], nb[t], acc);
                        P[n][3 * cq4 + f] = acc;
                    }
                }
            }
            // ---- UpdateNet (nca.py:40-46): 3C -> 64 -> 64 -> C ---------------------------
            f32x4 acc2[4][NT];
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) {
                const f32x4 bias = *reinterpret_cast<const f32x4*>(B2L + 16 * m2 + 4 * g);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc2[m2][n] = bias;
            }
#pragma unroll 1
            for (int m = 0; m < 4; ++m) {
                const float* const w1m = W1L + m * K::K1S * 64 + lane;
                const f32x4 bias = *reinterpret_cast<const f32x4*>(B1L + 16 * m + 4 * g);
                f32x4 acc1[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) acc1[n] = bias;
#pragma unroll
                for (int s = 0; s < K::K1S; ++s) {
                    const float wa = w1m[s * 64];
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(wa, P[n][s], acc1[n]);
                }
                const float* const w2m = W2L + (4 * m) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int m2 = 0; m2 < 4; ++m2) {
                        const float wa = w2m[(m2 * 16 + r) * 64];
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc2[m2][n] = nca_mfma(wa, fmaxf(acc1[n][r], 0.0f), acc2[m2][n]);
                    }
                }
            }
            f32x4 acc3[K::M3T][NT];
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc3[m3][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // out.4 has no bias
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3) {
                        const float wa = W3L[(m3 * 16 + 4 * m + r) * 64 + lane];
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc3[m3][n] = nca_mfma(wa, fmaxf(acc2[m][n][r], 0.0f), acc3[m3][n]);
                    }
                }
            }
            // ---- x' = x + rand_mask * out (nca.py:189); stays pending ---------------------
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int gy = ty0 + r0[n], gx = tx0 + q0[n];
                if (gy < H && gx < W) {
                    const float mk = MK[r0[n] * TW + q0[n]];
                    const float life = LIFE[(r0[n] + 2) * K::L2S + q0[n] + 2];
                    const size_t off = (size_t)gy * W + gx;
                    float* const ob = a.x_out + (size_t)b * C * plane + off;
                    float xo[K::M3T][4];
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                        for (int r = 0; r < 4; ++r) xo[m3][r] = xb[min(16 * m3 + 4 * g + r, C - 1) * plane + off];
#pragma unroll
                    for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ch = 16 * m3 + 4 * g + r;
                            if (ch < C) {
                                float x0 = xo[m3][r];
                                if (pending) x0 = nca_clamp(x0 * life, a.lo, a.hi);
                                ob[ch * plane] = x0 + mk * acc3[m3][n][r];
                            }
                        }
                }
            }
        }
    }
}

// ---- host-side launch helpers -------------------------------------------------------------
template <typename KernelT>
hipError_t set_lds(KernelT kern, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

int grid_for(int ntiles, int wg_per_cu) {
    const int cap = nca_cu_count() * wg_per_cu;
    return ntiles < cap ? ntiles : cap;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

template <int CP, int FC, bool HAS_COND, bool VEC, bool B16 = false, bool ACC = false>
hipError_t launch_dynca_v(const NcaDyncaArgs& a, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NT = 4;   // (C = 32 ran two rows per pass while it was compiled for 256 registers)
    using K = DyncaCfg<CP, FC, HAS_COND, TH, TW, NT>;
    auto kern = dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, VEC, false, B16, ACC>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;   // per instantiation; keyed by device inside
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int grid = grid_for(ntiles, lds * 2 <= 160 * 1024 ? 2 : 1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
    return hipGetLastError();
}

// two-scale perception (a.pc = coarse-level perception of x_in, H and W even): one workgroup per CU (the coarse tile is in LDS)
template <int CP, int FC, bool HAS_COND>
hipError_t launch_dynca_ms(const NcaDyncaArgs& a, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NT = 4;
    using K = DyncaCfg<CP, FC, HAS_COND, TH, TW, NT>;
    static_assert(K::LDS_FLOATS_MS * 4 <= 160 * 1024, "LDS budget (two-scale step)");
    const bool vec = (a.W % 4 == 0) && aligned16(a.x_in) && (((size_t)a.H * a.W) % 4 == 0);
    const size_t lds = (size_t)K::LDS_FLOATS_MS * sizeof(float);
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int grid = grid_for(ntiles, 1);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = set_lds(kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
        return hipGetLastError();
    };
    return vec ? go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, true, false, false, false, false, true>)
               : go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, false, false, false, false, false, true>);
}

template <int CP, int FC, bool HAS_COND, bool ACC = false>
hipError_t launch_dynca(const NcaDyncaArgs& a, hipStream_t st) {
    const bool vec = (a.W % 4 == 0) && aligned16(a.x_in) && (((size_t)a.H * a.W) % 4 == 0);
    return vec ? launch_dynca_v<CP, FC, HAS_COND, true, false, ACC>(a, st) : launch_dynca_v<CP, FC, HAS_COND, false, false, ACC>(a, st);
}

template <int CP, int FC, bool HAS_COND, bool ACC>
hipError_t launch_dynca_bwd(const NcaDyncaArgs& a, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NT = 2;       // two 16-cell groups per pass: 2 workgroups per CU at C <= 16 (registers and LDS)
    using K = DyncaCfg<CP, FC, HAS_COND, TH, TW, NT>;
    static_assert(K::LDS_FLOATS_BWD_W2 * 4 <= 160 * 1024 && 4 * (16 * K::M2T * FC + 16 * K::M2T) <= K::LDS_FLOATS_BWD_W2, "LDS budget (backward)");
    const bool vec = (a.W % 4 == 0) && aligned16(a.x_in);
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int grid = nca_dynca_bwd_grid_c(a.B, a.C, a.H, a.W);
    (void)ntiles;
    auto go = [&](auto kern, size_t lds) -> hipError_t {
        hipError_t e = set_lds(kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
        return hipGetLastError();
    };
    if (a.gw2_ws) {   // fused dW2 | db2: per-workgroup partials, h is not written
        const size_t lds2 = (size_t)K::LDS_FLOATS_BWD_W2 * sizeof(float);
        return vec ? go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, true, true, false, ACC, true>, lds2)
                   : go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, false, true, false, ACC, true>, lds2);
    }
    if constexpr (ACC) return hipErrorInvalidValue;   // hidden-layer slices exist for the fused-dW2 form only
    else {
        const size_t lds = (size_t)K::LDS_FLOATS_BWD * sizeof(float);
        return vec ? go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, true, true>, lds)
                   : go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, false, true>, lds);
    }
}

// dL/dx_t = G + dy[0:C] + adj(Sx)(dy[C:2C]) + adj(Sy)(dy[2C:3C]) + adj(L)(dy[3C:4C]), where adj is the adjoint of
// "F.pad(mode) then 3x3 cross-correlation" (dynca.py:83-86): a cell p collects w[t] * dy[q] from every (q, t) whose
// padded source index pad(q + t) equals p.  Candidates q lie in p's 3x3 neighbourhood (wrapped for 'circular').
__device__ float dynca_bwd_cell(const NcaDyncaArgs& a, int b, int c, int py, int px) {
    const int C = a.C, H = a.H, W = a.W, pad = a.pad_mode;
    const size_t plane = (size_t)H * W;
    const float* const dy = a.dybuf + (size_t)b * 4 * C * plane;
    const size_t off = (size_t)py * W + px;
    float base = a.g_next ? a.g_next[((size_t)b * C + c) * plane + off] : 0.0f;
    if (a.g_extra) base += a.g_extra[((size_t)b * C + c) * plane + off];   // cotangent of the intermediate state itself
    if (a.coarse_add)   // two-scale perception: adjoint of the 2x2 mean, 0.25 * dL/dx_coarse of the cell's coarse parent
        base += 0.25f * a.coarse_add[((size_t)b * C + c) * (size_t)(H >> 1) * (W >> 1) + (size_t)(py >> 1) * (W >> 1) + (px >> 1)];
    float acc = dy[(size_t)c * plane + off];
    // per axis: for candidate offset iq in {-1,0,1} and tap t in {-1,0,1}: does pad(q + t) land on p ?
    int qy[3], qx[3];
    bool hy[3][3], hx[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int q = py + i - 1;
        if (pad == NCA_PAD_CIRCULAR) q = ((q % H) + H) % H;
        bool ok = q >= 0 && q < H;
        for (int k = 0; k < i; ++k) ok = ok && !(qy[k] == q);   // wrapped duplicates when H < 3
        qy[i] = ok ? q : -1;
#pragma unroll
        for (int t = 0; t < 3; ++t) hy[i][t] = ok && nca_pad_index(q + t - 1, H, pad) == py;
        q = px + i - 1;
        if (pad == NCA_PAD_CIRCULAR) q = ((q % W) + W) % W;
        ok = q >= 0 && q < W;
        for (int k = 0; k < i; ++k) ok = ok && !(qx[k] == q);
        qx[i] = ok ? q : -1;
#pragma unroll
        for (int t = 0; t < 3; ++t) hx[i][t] = ok && nca_pad_index(q + t - 1, W, pad) == px;
    }
    const float SX[3][3] = {{-1.f, 0.f, 1.f}, {-2.f, 0.f, 2.f}, {-1.f, 0.f, 1.f}};
    const float LP[3][3] = {{1.f, 2.f, 1.f}, {2.f, -12.f, 2.f}, {1.f, 2.f, 1.f}};
#pragma unroll
    for (int iy = 0; iy < 3; ++iy)
#pragma unroll
        for (int ix = 0; ix < 3; ++ix) {
            if (qy[iy] < 0 || qx[ix] < 0) continue;
            float wsx = 0.f, wsy = 0.f, wl = 0.f;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx)
                    if (hy[iy][ty] && hx[ix][tx]) { wsx += SX[ty][tx]; wsy += SX[tx][ty]; wl += LP[ty][tx]; }
            const size_t qo = (size_t)qy[iy] * W + qx[ix];
            acc = fmaf(wsx, dy[(size_t)(C + c) * plane + qo], acc);
            acc = fmaf(wsy, dy[(size_t)(2 * C + c) * plane + qo], acc);
            acc = fmaf(wl, dy[(size_t)(3 * C + c) * plane + qo], acc);
        }
    return base + (a.dy_half ? 0.5f * acc : acc);   // two-scale perception: the fine level carries half of dL/dy
}

// any shape: one thread per cell
__global__ __launch_bounds__(256) void dynca_step_bwd_stencil_kernel(const NcaDyncaArgs a) {
    const int C = a.C, H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)a.B * C * plane) return;
    const int px = (int)(id % W), py = (int)((id / W) % H), c = (int)((id / plane) % C), b = (int)(id / (plane * C));
    a.g_out[id] = dynca_bwd_cell(a, b, c, py, px);
}

// W % 4 == 0, 16-byte aligned planes: one thread per 4 W-contiguous cells.  Away from the image border the padding plays no
// part and the adjoint is the correlation with the flipped filters (Sobel flips sign, the Laplacian is symmetric): three
// rows of three planes, 16-byte loads plus the two edge cells (L1 hits: the neighbouring lanes fetch those lines).
// The border band (two rows top and bottom, four columns left and right) is left to separate WORKGROUPS of the same launch
// (dynca_bwd_border_block below): a few lanes of every wave running the per-cell routine would hold the whole wave for its
// ~2000 instructions.
__device__ __forceinline__ void dynca_bwd_vec_block(const NcaDyncaArgs& a, unsigned blk) {
    const int C = a.C, H = a.H, W = a.W, W4 = W >> 2;
    const size_t plane = (size_t)H * W;
    const size_t id = (size_t)blk * blockDim.x + threadIdx.x;
    if (id >= (size_t)a.B * C * H * W4) return;
    const int x0 = (int)(id % W4) * 4, py = (int)((id / W4) % H), c = (int)((id / ((size_t)W4 * H)) % C),
              b = (int)(id / ((size_t)W4 * H * C));
    float* const out = a.g_out + ((size_t)b * C + c) * plane + (size_t)py * W + x0;
    // border band of two cells: with 'reflect' a border cell's out-of-range taps land one cell INSIDE the image
    if (py < 2 || py > H - 3 || x0 < 4 || x0 + 8 > W) return;   // the border band belongs to dynca_bwd_border_block
    const float* const dy = a.dybuf + (size_t)b * 4 * C * plane + (size_t)py * W + x0;
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.g_next) gv = *reinterpret_cast<const float4*>(a.g_next + ((size_t)b * C + c) * plane + (size_t)py * W + x0);
    const float4 d0 = *reinterpret_cast<const float4*>(dy + (size_t)c * plane);
    float base[4] = {gv.x, gv.y, gv.z, gv.w};
    float acc[4] = {d0.x, d0.y, d0.z, d0.w};
    if (a.g_extra) {
        const float4 ge = *reinterpret_cast<const float4*>(a.g_extra + ((size_t)b * C + c) * plane + (size_t)py * W + x0);
        base[0] += ge.x; base[1] += ge.y; base[2] += ge.z; base[3] += ge.w;
    }
    if (a.coarse_add) {   // W % 4 == 0: the 4 cells have the coarse parents x0/2 and x0/2 + 1 (8-byte aligned pair)
        const float2 cp = *reinterpret_cast<const float2*>(a.coarse_add + ((size_t)b * C + c) * (size_t)(H >> 1) * (W >> 1) +
                                                           (size_t)(py >> 1) * (W >> 1) + (x0 >> 1));
        base[0] += 0.25f * cp.x; base[1] += 0.25f * cp.x; base[2] += 0.25f * cp.y; base[3] += 0.25f * cp.y;
    }
    float v[3][3][6];   // [filter plane][row y-1..y+1][col x0-1..x0+4]
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* const row = dy + (size_t)((f + 1) * C + c) * plane + (ptrdiff_t)(r - 1) * W;
            const float4 m = *reinterpret_cast<const float4*>(row);
            v[f][r][0] = row[-1]; v[f][r][1] = m.x; v[f][r][2] = m.y; v[f][r][3] = m.z; v[f][r][4] = m.w; v[f][r][5] = row[4];
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // flipped Sobel-x: weight of dy1[p + (dyy, dxx)] is SX[1 - dyy][1 - dxx] = -SX[1 + dyy][1 + dxx]
        const float sx = (v[0][0][j] - v[0][0][j + 2]) + 2.0f * (v[0][1][j] - v[0][1][j + 2]) + (v[0][2][j] - v[0][2][j + 2]);
        const float sy = (v[1][0][j] + 2.0f * v[1][0][j + 1] + v[1][0][j + 2]) - (v[1][2][j] + 2.0f * v[1][2][j + 1] + v[1][2][j + 2]);
        const float lp = (v[2][0][j] + v[2][0][j + 2] + v[2][2][j] + v[2][2][j + 2]) +
                         2.0f * (v[2][0][j + 1] + v[2][1][j] + v[2][1][j + 2] + v[2][2][j + 1]) - 12.0f * v[2][1][j + 1];
        acc[j] += sx + sy + lp;
        acc[j] = base[j] + (a.dy_half ? 0.5f * acc[j] : acc[j]);
    }
    *reinterpret_cast<float4*>(out) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

template <int CP, bool VEC>
hipError_t launch_cond_v(const NcaCondArgs& a, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NT = 4;   // (C > 16 ran two rows per pass while it was compiled for 256 registers; one workgroup per CU has 512)
    using K = CondCfg<CP, TH, TW, NT>;
    auto kern = cond_step_fwd_kernel<CP, TH, TW, NT, VEC>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;   // per instantiation; keyed by device inside
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int grid = grid_for(ntiles, lds * 2 <= 160 * 1024 ? 2 : 1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
    return hipGetLastError();
}

template <int CP>
hipError_t launch_cond(const NcaCondArgs& a, hipStream_t st) {
    const bool vec = (a.W % 4 == 0) && aligned16(a.x_in) && (a.goal == nullptr || aligned16(a.goal));
    return vec ? launch_cond_v<CP, true>(a, st) : launch_cond_v<CP, false>(a, st);
}

}  // namespace

static bool g_force_generic = getenv("NCAHIP_FORCE_GENERIC") != nullptr;
static int g_cond_variant = getenv("NCAHIP_COND_VARIANT") ? atoi(getenv("NCAHIP_COND_VARIANT")) : 0;
void nca_set_force_generic(bool on) { g_force_generic = on; }
void nca_set_cond_variant(int v) { g_cond_variant = v; }

// ---- dispatch: smallest instantiation that covers (C, fc); padding lanes carry zero weights ---
hipError_t nca_launch_dynca_step_fwd(const NcaDyncaArgs& a, hipStream_t st) {
    const bool hc = a.c_cond > 0;
    if (a.pc) {   // two-scale perception: the shipped video models (C = 12 / fc = 96 and C = 16 / fc = 128, pos_emb conditioning)
        if ((a.H | a.W) & 1) return hipErrorInvalidValue;
        if (a.C <= 12 && a.fc <= 96) return hc ? launch_dynca_ms<12, 96, true>(a, st) : launch_dynca_ms<12, 96, false>(a, st);
        if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca_ms<16, 128, true>(a, st) : launch_dynca_ms<16, 128, false>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.C <= 12 && a.fc <= 96) return hc ? launch_dynca<12, 96, true>(a, st) : launch_dynca<12, 96, false>(a, st);
    if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca<16, 128, true>(a, st) : launch_dynca<16, 128, false>(a, st);
    if (a.C <= 32 && a.fc <= 128) return hc ? launch_dynca<32, 128, true>(a, st) : launch_dynca<32, 128, false>(a, st);   // configs[4]
    if (a.C <= 32) {
        // fc > 128: the A-operand image of w1 no longer fits the LDS beside the tile.  w2 relu(w1 y + b1) is a sum over hidden
        // units, so the step runs as one launch per 128-wide slice: the first writes x + mask*(slice + b2), the others add
        // their slice to x_out (same mask: same explicit uniforms / Philox key).  Each launch recomputes the perception.
        const int K1 = 4 * a.C + a.c_cond;
        const bool c16 = a.C <= 16;
        for (int h0 = 0; h0 < a.fc; h0 += 128) {
            NcaDyncaArgs s = a;
            s.w1 = a.w1 + (size_t)h0 * K1;
            s.b1 = a.b1 + h0;
            s.w2 = a.w2 + h0;
            s.fc = a.fc - h0 < 128 ? a.fc - h0 : 128;
            s.w2_ld = a.fc;
            hipError_t e;
            if (h0 == 0) e = c16 ? (hc ? launch_dynca<16, 128, true>(s, st) : launch_dynca<16, 128, false>(s, st))
                                 : (hc ? launch_dynca<32, 128, true>(s, st) : launch_dynca<32, 128, false>(s, st));
            else e = c16 ? (hc ? launch_dynca<16, 128, true, true>(s, st) : launch_dynca<16, 128, false, true>(s, st))
                         : (hc ? launch_dynca<32, 128, true, true>(s, st) : launch_dynca<32, 128, false, true>(s, st));
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    return hipErrorInvalidValue;
}

// bf16 state storage: x_in / x_out point at bf16 data (cond, uniforms, weights stay f32); same kernel, exact f32 compute,
// round-to-nearest-even on store.  The 8-byte vector path needs W % 4 == 0 and 8-byte aligned planes.
// the border band of every (b, c) plane, one thread per cell: 4 full rows + 8 columns of the H - 4 rows between them
__device__ __forceinline__ void dynca_bwd_border_block(const NcaDyncaArgs& a, unsigned blk) {
    const int C = a.C, H = a.H, W = a.W;
    const int per_plane = 4 * W + 8 * (H - 4);
    const size_t id = (size_t)blk * blockDim.x + threadIdx.x;
    if (id >= (size_t)a.B * C * per_plane) return;
    const int k = (int)(id % per_plane), c = (int)((id / per_plane) % C), b = (int)(id / ((size_t)per_plane * C));
    int py, px;
    if (k < 4 * W) {
        const int r = k / W;
        py = r < 2 ? r : H - 4 + r;
        px = k - r * W;
    } else {
        const int q = k - 4 * W, j = q & 7;
        py = 2 + (q >> 3);
        px = j < 4 ? j : W - 8 + j;
    }
    a.g_out[((size_t)b * C + c) * (size_t)H * W + (size_t)py * W + px] = dynca_bwd_cell(a, b, c, py, px);
}
// One launch for both: the first nborder workgroups take the border band (their threads are the slow ones: dispatched first,
// they run under the interior's memory traffic instead of after it -- as a launch of their own they were 19 us behind a 50 us
// interior pass), the rest the interior.
__global__ __launch_bounds__(256) void dynca_step_bwd_stencil_vec_kernel(const NcaDyncaArgs a, unsigned nborder) {
    if (blockIdx.x < nborder) dynca_bwd_border_block(a, blockIdx.x);
    else dynca_bwd_vec_block(a, blockIdx.x - nborder);
}

template <int CP, int FC, bool HAS_COND>
static hipError_t launch_dynca_b16(const NcaDyncaArgs& a, hipStream_t st) {
    const bool vec = (a.W % 4 == 0) && (((uintptr_t)a.x_in & 7u) == 0) && (((size_t)a.H * a.W) % 4 == 0);
    return vec ? launch_dynca_v<CP, FC, HAS_COND, true, true>(a, st) : launch_dynca_v<CP, FC, HAS_COND, false, true>(a, st);
}
hipError_t nca_launch_dynca_step_fwd_bf16(const NcaDyncaArgs& a, hipStream_t st) {
    const bool hc = a.c_cond > 0;
    if (a.C <= 12 && a.fc <= 96) return hc ? launch_dynca_b16<12, 96, true>(a, st) : launch_dynca_b16<12, 96, false>(a, st);
    if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca_b16<16, 128, true>(a, st) : launch_dynca_b16<16, 128, false>(a, st);
    if (a.C <= 32 && a.fc <= 128) return hc ? launch_dynca_b16<32, 128, true>(a, st) : launch_dynca_b16<32, 128, false>(a, st);
    return hipErrorInvalidValue;
}

int nca_dynca_bwd_grid_c(int B, int C, int H, int W) {
    const int ntiles = B * ((W + 31) / 32) * ((H + 7) / 8);
    return grid_for(ntiles, C > 16 ? 1 : 2);
}
int nca_dynca_bwd_grid(int B, int H, int W) { return nca_dynca_bwd_grid_c(B, 16, H, W); }   // upper bound (slab workspace sizing)

// two-scale perception in the backward: the same recomputation as the forward's (coarse tile in LDS); one workgroup per CU
template <int CP, int FC, bool HAS_COND>
hipError_t launch_dynca_bwd_ms(const NcaDyncaArgs& a, hipStream_t st) {
    constexpr int TH = 8, TW = 32, NT = 2;
    using K = DyncaCfg<CP, FC, HAS_COND, TH, TW, NT>;
    constexpr int LDSF = K::LDS_FLOATS_BWD_W2 + 4 * CP * K::PCS;
    static_assert(LDSF * 4 <= 160 * 1024, "LDS budget (two-scale backward)");
    if (!a.gw2_ws) return hipErrorInvalidValue;
    const bool vec = (a.W % 4 == 0) && aligned16(a.x_in);
    const size_t lds = (size_t)LDSF * sizeof(float);
    const int ntiles = a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH);
    const int grid = grid_for(ntiles, 1);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = set_lds(kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, a);
        return hipGetLastError();
    };
    return vec ? go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, true, true, false, false, true, true>)
               : go(dynca_step_fwd_kernel<CP, FC, HAS_COND, TH, TW, NT, false, true, false, false, true, true>);
}
int nca_dynca_bwd_ms_grid(int B, int H, int W) { return grid_for(B * ((W + 31) / 32) * ((H + 7) / 8), 1); }

// The MLP part of one backward step (writes dh, dL/dy and, with gw2_ws, the per-workgroup dW2 | db2 partials).  acc: a later
// 128-wide slice of a wide hidden layer -- dL/dy is added to what the earlier slices wrote.
hipError_t nca_launch_dynca_step_bwd_mlp(const NcaDyncaArgs& a, hipStream_t st, bool acc) {
    const bool hc = a.c_cond > 0;
    if (a.pc) {   // two-scale perception (C <= 16, fc <= 128, even sizes: checked by the C ABI)
        if (acc || ((a.H | a.W) & 1)) return hipErrorInvalidValue;
        if (a.C <= 12 && a.fc <= 96) return hc ? launch_dynca_bwd_ms<12, 96, true>(a, st) : launch_dynca_bwd_ms<12, 96, false>(a, st);
        if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca_bwd_ms<16, 128, true>(a, st) : launch_dynca_bwd_ms<16, 128, false>(a, st);
        return hipErrorInvalidValue;
    }
    if (!acc) {
        if (a.C <= 12 && a.fc <= 96) return hc ? launch_dynca_bwd<12, 96, true, false>(a, st) : launch_dynca_bwd<12, 96, false, false>(a, st);
        if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca_bwd<16, 128, true, false>(a, st) : launch_dynca_bwd<16, 128, false, false>(a, st);
        if (a.C <= 32 && a.fc <= 128) return hc ? launch_dynca_bwd<32, 128, true, false>(a, st) : launch_dynca_bwd<32, 128, false, false>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.C <= 16 && a.fc <= 128) return hc ? launch_dynca_bwd<16, 128, true, true>(a, st) : launch_dynca_bwd<16, 128, false, true>(a, st);
    if (a.C <= 32 && a.fc <= 128) return hc ? launch_dynca_bwd<32, 128, true, true>(a, st) : launch_dynca_bwd<32, 128, false, true>(a, st);
    return hipErrorInvalidValue;
}

// The stencil-adjoint part: g_out = g_next (+ g_extra) + adj(perception)(dybuf).
hipError_t nca_launch_dynca_step_bwd_stencil(const NcaDyncaArgs& a, hipStream_t st) {
    const size_t n = (size_t)a.B * a.C * a.H * a.W;
    const bool vec = (a.W % 4 == 0) && a.W >= 12 && a.H >= 3 && (a.g_next == nullptr || aligned16(a.g_next)) && aligned16(a.g_out) &&
                     aligned16(a.dybuf) && (a.g_extra == nullptr || aligned16(a.g_extra)) &&
                     (a.coarse_add == nullptr || (((uintptr_t)a.coarse_add & 7u) == 0 && a.H % 2 == 0));
    if (vec && a.H >= 5) {
        const size_t nb = (size_t)a.B * a.C * (4 * a.W + 8 * (a.H - 4));
        const unsigned nborder = (unsigned)((nb + 255) / 256), nvec = (unsigned)((n / 4 + 255) / 256);
        hipLaunchKernelGGL(dynca_step_bwd_stencil_vec_kernel, dim3(nborder + nvec), dim3(256), 0, st, a, nborder);
    } else {
        hipLaunchKernelGGL(dynca_step_bwd_stencil_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

hipError_t nca_launch_dynca_step_bwd(const NcaDyncaArgs& a, hipStream_t st) {
    hipError_t e = nca_launch_dynca_step_bwd_mlp(a, st, false);
    if (e != hipSuccess) return e;
    return nca_launch_dynca_step_bwd_stencil(a, st);
}

hipError_t nca_launch_cond_step_fwd(const NcaCondArgs& a, hipStream_t st) {
    // the tile kernels address with 32-bit byte offsets from the batch item's base (issue_loads): H*W < 2^24, C*H*W*4 < 2^32
    const bool small = (size_t)a.H * a.W < ((size_t)1 << 24) && (size_t)(a.C <= 16 ? 16 : 20) * a.H * a.W * 4 < ((size_t)1 << 32);
    const bool tiled = !g_force_generic && small && (a.W % 4 == 0) && aligned16(a.x_in) && aligned16(a.x_out) &&
                       (a.goal == nullptr || aligned16(a.goal));
    if (tiled && a.C <= 16) return g_cond_variant == 1 ? nca_launch_cond_step_fwd_wave(a, st) : nca_launch_cond_step_fwd_pc(a, st);
    if (tiled && a.C <= 20) return nca_launch_cond_step_fwd_pc(a, st);   // the reference's default C = 20: producer/consumer kernel, wide carve
    if (a.C <= 12) return launch_cond<12>(a, st);
    if (a.C <= 16) return launch_cond<16>(a, st);
    // the reference's default model is C = 3 + 1 + 16 = 20 (nca.py:62-94): the generic kernel family with two output tiles
    if (a.C <= 20) return launch_cond<20>(a, st);
    if (a.C <= 24) return launch_cond<24>(a, st);
    if (a.C <= 32) return launch_cond<32>(a, st);
    return hipErrorInvalidValue;
}
