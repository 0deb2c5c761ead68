// nca_cond_tile.h -- device code shared by the wave-private ConditionedNCA kernels (forward:
// nca_cond_wave.hip, backward: nca_cond_bwd.hip): LDS carve, tile walk, batched global loads,
// pending-life-mask resolution and staging, perception.  See nca_cond_wave.hip for the design notes.
#pragma once
#include "nca_common.h"
#include "nca_kernels.h"

namespace {


constexpr int kWaves = 8, kThreadsW = kWaves * 64;
constexpr int WTH = 4, WTW = 16;  // wave tile
constexpr int STH = 16, STW = 32; // super-tile of a workgroup: 4 x 2 wave tiles
constexpr int RS = 24;            // LDS row stride of every per-wave 2-D array; image col tx0+c <-> index c+4
constexpr int ZROWS = WTH + 2, CS = ZROWS * RS;  // 144 floats per channel (144 % 32 == 16)
constexpr int XRS = 68;           // resolved-state copy: channel stride (4*68 % 32 == 16)
static_assert(CS % 32 == 16 && (4 * XRS) % 32 == 16, "bank layout");

template <int CP>
struct WCfg {
    static constexpr int HID = 64;
    static constexpr int K1S = 3 * CP / 4;
    static constexpr int K1S4 = (K1S + 3) / 4;   // k-steps of layer 1 in groups of 4 (16-byte A-operand reads)
    static constexpr int M3T = (CP + 15) / 16;
    static constexpr int WPS = 28;
    // shared weight image (floats)
    static constexpr int OFF_W1 = 0;
    static constexpr int OFF_W2 = OFF_W1 + 4 * K1S4 * 4 * 64;
    static constexpr int OFF_W3 = OFF_W2 + 4 * 16 * 64;
    static constexpr int OFF_B1 = OFF_W3 + M3T * 16 * 64;
    static constexpr int OFF_B2 = OFF_B1 + HID;
    static constexpr int OFF_WP = OFF_B2 + HID;
    static constexpr int SHARED = OFF_WP + CP * WPS;
    // per-wave region (floats)
    static constexpr int PW_Z = 0;
    static constexpr int PW_XR = PW_Z + CP * CS;
    static constexpr int PW_A3 = PW_XR + CP * XRS;       // alpha' rows ty0-3.. (10 rows); after the life mask is
                                                         // resolved: rows 0-5 = PN, rows 6-9 = fire mask MK
    static constexpr int PW_LIFE = PW_A3 + (WTH + 6) * RS;
    static constexpr int PW_A2 = PW_LIFE + (WTH + 4) * RS;
    static constexpr int PW = PW_A2 + (WTH + 4) * RS;
    static constexpr int LDS_FLOATS = SHARED + kWaves * PW;
    static_assert(CP % 4 == 0 && SHARED % 4 == 0 && PW % 4 == 0 && PW_XR % 4 == 0 && PW_A3 % 4 == 0, "16-byte carve");
    // the 8-wave carve fits for CP <= 16 (checked where a kernel is launched with it); wider CP only borrows the weight-image
    // offsets and the per-wave sub-offsets (backward front / matrix kernels, 4 waves with their own carve)
    static constexpr bool kFits8 = LDS_FLOATS * 4 <= 160 * 1024;
};

#if defined(NCA_STAMPS)
// per-kernel stamps (cheap: entry / loop start / loop end), region behind the per-tile stamps
#define NCA_KSTAMP(i)                                                                                  \
    do {                                                                                               \
        if (a.dbg) {                                                                                   \
            unsigned long long t_;                                                                     \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
            if ((threadIdx.x & 63) == 0) a.dbg[(size_t)gridDim.x * kWaves * 8 * 16 + (size_t)(blockIdx.x * kWaves + (threadIdx.x >> 6)) * 8 + (i)] = t_; \
        }                                                                                              \
    } while (0)
#else
#define NCA_KSTAMP(i) do { } while (0)
#endif
#if defined(NCA_STAMPS) && NCA_STAMPS >= 2
// Diagnostic build only: phase stamps per wave tile -> a.dbg[((wg*8+wave)*kStampTiles + tile)*16 + i].
constexpr int kStampTiles = 8;  // stamped tiles per wave (buffer: [wg*8+wave][kStampTiles][16])
#define NCA_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (a.dbg && tile_no < kStampTiles) {                                                          \
            unsigned long long t_;                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            if (lane == 0) a.dbg[((size_t)(blockIdx.x * kWaves + (threadIdx.x >> 6)) * kStampTiles + tile_no) * 16 + (i)] = t_; \
        }                                                                                              \
    } while (0)
#else
#define NCA_STAMP(i) do { } while (0)
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
// clamp as one v_med3_f32 (lo <= hi; identical to fmin(fmax(v,lo),hi) for every non-NaN v)
__device__ __forceinline__ float wclamp(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float max3x3(const float* p) {
    float m = fmaxf(fmaxf(p[-RS - 1], p[-RS]), p[-RS + 1]);
    m = fmaxf(m, fmaxf(fmaxf(p[-1], p[0]), p[1]));
    return fmaxf(m, fmaxf(fmaxf(p[RS - 1], p[RS]), p[RS + 1]));
}
// LDS hand-off between lanes of ONE wave: DS ops of a wave execute in order, so only the compiler
// must be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N, int NTHR = kThreadsW, typename MapT>
__device__ __forceinline__ void fill_image_w(float* __restrict__ dst, const float* __restrict__ src, int tid, MapT map) {
    constexpr int U = 8;
    constexpr int kThreadsW = NTHR;
    for (int base = tid; base < N; base += kThreadsW * U) {
        float v[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + kThreadsW * u;
            const long o = idx < N ? map(idx) : -1;
            ok[u] = o >= 0;
            v[u] = src[ok[u] ? o : 0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + kThreadsW * u;
            if (idx < N) dst[idx] = ok[u] ? v[u] : 0.0f;
        }
    }
}

// Two-phase variant: every image's loads are issued before the first one is waited for (fill_image_w waits per batch
// of 8 -- eight cold round trips in a row at kernel start).  Validity travels as a bit mask in one VGPR.
template <int N, int NTHR>
struct FillRegs {
    static constexpr int PER = (N + NTHR - 1) / NTHR;
    static_assert(PER <= 32, "one mask word");
    float v[PER];
    unsigned ok;
};
template <int N, int NTHR, typename MapT>
__device__ __forceinline__ void fill_load(FillRegs<N, NTHR>& r, const float* __restrict__ src, int tid, MapT map) {
    r.ok = 0u;
#pragma unroll
    for (int u = 0; u < FillRegs<N, NTHR>::PER; ++u) {
        const int idx = tid + NTHR * u;
        const long o = idx < N ? map(idx) : -1;
        r.ok |= (o >= 0 ? 1u : 0u) << u;
        r.v[u] = src[o >= 0 ? o : 0];
    }
}
template <int N, int NTHR>
__device__ __forceinline__ void fill_store(const FillRegs<N, NTHR>& r, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int u = 0; u < FillRegs<N, NTHR>::PER; ++u) {
        const int idx = tid + NTHR * u;
        if (idx < N) dst[idx] = ((r.ok >> u) & 1u) ? r.v[u] : 0.0f;
    }
}

// LDS arrays of one tile (all rows RS floats wide, image col tx0+c <-> index c+4; see WCfg for the sizes)
struct TileLds {
    float* Z;     // z = x + goal*pre, halo 1: [CP][6][RS]
    float* XR;    // resolved state interior -> x' in place: [CP][XRS]
    float* A3;    // alpha' halo 3 (10 rows); dead after the life mask is resolved
    float* PN;    // pre-life mask of this step, halo 1 (6 rows); may alias A3
    float* LIFE;  // life mask of the previous step, halo 2 (8 rows)
    float* A2;    // resolved alpha, halo 2 (8 rows)
    float* MK;    // fire mask, [4][16]
};
template <int CP>
__device__ __forceinline__ TileLds wave_private_lds(float* PWR) {
    using K = WCfg<CP>;
    return TileLds{PWR + K::PW_Z, PWR + K::PW_XR, PWR + K::PW_A3, PWR + K::PW_A3, PWR + K::PW_LIFE, PWR + K::PW_A2,
                   PWR + K::PW_A3 + ZROWS * RS};
}

// ---- per-wave tile bookkeeping ------------------------------------------------------------------
struct WTile {
    int b, ty0, tx0;
    bool valid, inner;  // inner: the 3-cell (1-cell without alive channel) halo lies inside the image
};

// ---- storage type of the state / goal tensors in HBM ----------------------------------------------
// The tile code computes in f32; ST says how a value travels to and from memory.  Raw loaded registers are kept raw
// until staging consumes them (converting at load time would wait for the load).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct StF32 {
    static constexpr int BYTES = 4;
    typedef float raw1;
    typedef f32x4 raw4;
    static __device__ __forceinline__ float cv1(raw1 v) { return v; }
    static __device__ __forceinline__ f32x4 cv4(raw4 v) { return v; }
    template <int AUX>
    static __device__ __forceinline__ raw1 ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
    }
    template <int AUX>
    static __device__ __forceinline__ raw4 ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
    }
    template <int AUX>
    static __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)voff, (int)soff, AUX);
    }
    // plain global loads at an ELEMENT offset from a typeless base (backward kernel: pending state x'_t)
    static __device__ __forceinline__ raw1 gld1(const void* base, unsigned elem) { return static_cast<const float*>(base)[elem]; }
    static __device__ __forceinline__ raw4 gld4(const void* base, unsigned elem) {
        return *reinterpret_cast<const f32x4*>(static_cast<const float*>(base) + elem);
    }
};
// bf16 storage (round-to-nearest-even on store: v_cvt_pk_bf16_f32; widening on load is exact)
struct StBF16 {
    static constexpr int BYTES = 2;
    typedef unsigned raw1;   // low 16 bits
    typedef u32x2 raw4;      // 4 consecutive cells
    static __device__ __forceinline__ float cv1(raw1 v) { return __uint_as_float(v << 16); }
    static __device__ __forceinline__ f32x4 cv4(raw4 v) {
        return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                     __uint_as_float(v[1] & 0xffff0000u)};
    }
    template <int AUX>
    static __device__ __forceinline__ raw1 ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, (int)soff, AUX);
    }
    template <int AUX>
    static __device__ __forceinline__ raw4 ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
    }
    static __device__ __forceinline__ unsigned pk2(float lo, float hi) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{lo, hi}, bf16x2));
    }
    template <int AUX>
    static __device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pk2(v[0], v[1]), pk2(v[2], v[3])}, r, (int)voff, (int)soff, AUX);
    }
    static __device__ __forceinline__ raw1 gld1(const void* base, unsigned elem) { return (unsigned)static_cast<const uint16_t*>(base)[elem]; }
    static __device__ __forceinline__ raw4 gld4(const void* base, unsigned elem) {
        return *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(base) + elem);
    }
};
// channel index * plane bytes: a full 32-bit multiply -- plane bytes reach 2^24 at 2048 x 2048 fp32 cells, beyond v_mul_u32_u24
// (a handful per tile; the dispatch guards keep C * H * W * 4 below 2^32)
__device__ __forceinline__ unsigned nca_mul_u32(unsigned a, unsigned b) { return a * b; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t nca_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}

// Registers that carry one tile's global loads from issue (before the previous tile's MFMA chain)
// to staging (after it).
template <int CP, typename ST = StF32>
struct TileRegs {
    typename ST::raw1 a3v[5];   // alpha' halo 3
    unsigned prv[4];            // previous pre mask bytes, halo 2 (raw: converting at load time would force a wait)
    float uu, up;               // fire-mask uniform of the lane's cell: loaded (explicit uniforms) / in-kernel Philox
    typename ST::raw4 xf[CP / 2], gf[CP / 2];  // state / goal interior 4-cell groups
    typename ST::raw1 xh[CP / 4], gh[CP / 4];  // state / goal halo columns
};

// Lane geometry (all shifts of the lane id; recomputed where used, never carried across the MFMAs):
//   alpha' halo 3 : item k -> row 2k+hl (<10), col l5 (<22)      image (ty0-3+row, tx0-3+col)
//   pre    halo 2 : item k -> row 2k+hl (<8),  col l5 (<20)      image (ty0-2+row, tx0-2+col)
//   interior f4   : item k -> channel 2k+hl, slot l5 (<24): halo-1 row l5>>2, 4-cell group l5&3
//   halo columns  : item k -> channel 4k+q4, slot ci (<12): halo-1 row ci>>1, side ci&1
// CHK: -1 = decide from t.inner at run time, 0 = interior tile (no bounds logic), 1 = border tile.
// EXACT: the launch guarantees a.C == CP (no channel-padding guards).
// Addressing: raw buffer loads -- <buffer resource of the batch item (SGPRs)> + <32-bit VGPR byte offset shared by a
// whole group of loads> + <uniform SGPR offset that steps through channels / rows>: no per-load vector arithmetic at all
// on interior tiles.  Needs H*W < 2^24 and C*H*W*4 < 2^32 (checked by the dispatch in nca_step_fwd.hip; larger grids
// take the generic kernel).  kAuxCoherent selects the cache policy of the state-type loads (x, pre mask): 0 = ordinary
// cached loads (a step launch reads only what earlier launches wrote); 16 = sc1, agent-scope coherent -- what a fused
// multi-step launch with grid barriers needs (tried and measured slower: DESIGN.md section 4).
constexpr int kAuxCoherent = 0;
template <int CP, bool STATE, bool GOAL, int CHK = -1, bool EXACT = false, typename ST = StF32>
__device__ __forceinline__ void issue_loads(const NcaCondArgs& a, const WTile& t, int lane_in, TileRegs<CP, ST>& R) {
    constexpr unsigned SB = ST::BYTES;   // "4" in the names below = one storage element
    const int C = EXACT ? CP : a.C, H = a.H, W = a.W;
    const unsigned plane = (unsigned)(H * W);
    unsigned plane4 = plane * SB, W4 = (unsigned)W * SB;
    // opaque per call: the k * plane4 / k * W4 offsets below are then recomputed on the scalar ALU for every tile instead
    // of being hoisted out of the tile loop as ~40 loop-invariant SGPRs that get spilled to VGPR lanes and read back
    asm volatile("" : "+s"(plane4), "+s"(W4));
    const int gch0 = C - a.goal_ch;
    const bool pending = a.pre_in != nullptr, use_alive = a.alive_ch >= 0, has_goal = a.goal_ch > 0;
    const char* const xb = reinterpret_cast<const char*>(a.x_in) + (size_t)t.b * C * plane * SB;
    const size_t cell0 = (size_t)t.b * plane;
    const __amdgpu_buffer_rsrc_t rx = nca_rsrc(xb);
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int hl = (lane >> 5) & 1, l5 = lane & 31, q4 = (lane >> 4) & 3, ci = lane & 15;   // masked: the opaque lane id has no known range, and index * constant would become a quarter-rate 32-bit multiply
    const bool chk = CHK < 0 ? !t.inner : (CHK != 0);
    if (STATE && use_alive) {
        const __amdgpu_buffer_rsrc_t ra = nca_rsrc(xb + (size_t)a.alive_ch * plane4);
        if (!chk) {   // interior: one lane offset (columns >= 22 re-read column 21), rows step through the SGPR offset
            const unsigned vo = (unsigned)(__mul24(t.ty0 - 3 + hl, W) + t.tx0 - 3 + min(l5, 21)) * SB;
#pragma unroll
            for (int k = 0; k < 5; ++k) R.a3v[k] = ST::template ld1<kAuxCoherent>(ra, vo, 2u * k * W4);
        } else {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int gy = t.ty0 - 3 + 2 * k + hl, gx = t.tx0 - 3 + l5;
                const bool ok = l5 < 22 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                R.a3v[k] = ST::template ld1<kAuxCoherent>(ra, ok ? (unsigned)(__mul24(gy, W) + gx) * SB : 0u, 0u);
            }
        }
    }
    if (STATE) {
        // always load (from a valid address when there is no pending mask): no branch, no wait at issue
        const __amdgpu_buffer_rsrc_t rp = nca_rsrc((pending && use_alive) ? (const void*)(a.pre_in + cell0) : (const void*)xb);
        if (!chk) {
            const unsigned vo = (unsigned)(__mul24(t.ty0 - 2 + hl, W) + t.tx0 - 2 + min(l5, 19));
#pragma unroll
            for (int k = 0; k < 4; ++k) R.prv[k] = __builtin_amdgcn_raw_buffer_load_b8(rp, (int)vo, (int)(2u * k * (unsigned)W), kAuxCoherent);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int gy = t.ty0 - 2 + 2 * k + hl, gx = t.tx0 - 2 + l5;
                const bool ok = l5 < 20 && gy >= 0 && gy < H && gx >= 0 && gx < W;
                R.prv[k] = __builtin_amdgcn_raw_buffer_load_b8(rp, ok ? __mul24(gy, W) + gx : 0, 0, kAuxCoherent);
            }
        }
    }
    if (STATE) {
        const int cgy = t.ty0 + q4, cgx = t.tx0 + ci;
        const bool cin = !chk || (cgy < H && cgx < W);
        const unsigned pix = cin ? (unsigned)(__mul24(cgy, W) + cgx) : 0u;
        // both sources are produced without touching the loaded value (a select here would wait for every load issued
        // so far); stage_tile picks one
        // explicit uniforms: the cell's float; bit-packed masks (a.u_bits): the 32-bit word that holds the cell's bit
        R.uu = StF32::ld1<0>(nca_rsrc(a.u ? (a.u_bits ? (const void*)a.u : (const void*)(a.u + cell0)) : (const void*)xb),
                             a.u ? (a.u_bits ? (((unsigned)cell0 + pix) >> 5) * 4u : pix * 4u) : 0u, 0u);
        R.up = a.u ? 0.0f : nca_philox_cell(a.seed, a.step, cell0 + pix);
    }
    const __amdgpu_buffer_rsrc_t rg = nca_rsrc(has_goal ? reinterpret_cast<const char*>(a.goal) + (size_t)t.b * a.goal_ch * plane * SB : xb);
    {
        const int fr = l5 >> 2, ff = l5 & 3, fgy = t.ty0 - 1 + fr, fgx = t.tx0 + 4 * ff;
        const bool fok = l5 < 24 && (!chk || (fgy >= 0 && fgy < H && fgx + 3 < W));
        const unsigned pix4 = fok ? (unsigned)(__mul24(fgy, W) + fgx) * SB : 0u;
        const unsigned vox = pix4 + (hl ? plane4 : 0u);   // channel 2k + hl: the 2k part rides in the SGPR offset
        if (STATE) {
#pragma unroll
            for (int k = 0; k < CP / 2; ++k) {
                if (EXACT) R.xf[k] = ST::template ld4<kAuxCoherent>(rx, vox, 2u * k * plane4);
                else R.xf[k] = ST::template ld4<kAuxCoherent>(rx, pix4 + nca_mul_u32((unsigned)min(2 * k + hl, C - 1), plane4), 0u);
            }
        }
        if (GOAL && has_goal) {
#pragma unroll
            for (int k = 0; k < CP / 2; ++k) {
                // goal channel 2k + hl - gch0; lanes below gch0 never use the value and read channel 0 instead
                const int d = 2 * k - gch0;   // uniform
                if (EXACT) R.gf[k] = ST::template ld4<0>(rg, d >= 0 ? vox : pix4, d >= 0 ? (unsigned)d * plane4 : 0u);
                else   // C < CP: the padded channels 2k + hl >= C must not index past the goal tensor's last plane
                    R.gf[k] = ST::template ld4<0>(rg, pix4 + nca_mul_u32((unsigned)min(max(d + hl, 0), a.goal_ch - 1), plane4), 0u);
            }
        }
    }
    {
        const int hr = ci >> 1, hgy = t.ty0 - 1 + hr, hgx = (ci & 1) ? t.tx0 + WTW : t.tx0 - 1;
        const bool hok = ci < 12 && (!chk || (hgy >= 0 && hgy < H && hgx >= 0 && hgx < W));
        const unsigned pix4 = hok ? (unsigned)(__mul24(hgy, W) + hgx) * SB : 0u;
        const unsigned voh = pix4 + nca_mul_u32((unsigned)q4, plane4);   // channel 4k + q4
        if (STATE) {
#pragma unroll
            for (int k = 0; k < CP / 4; ++k) {
                if (EXACT) R.xh[k] = ST::template ld1<kAuxCoherent>(rx, voh, 4u * k * plane4);
                else R.xh[k] = ST::template ld1<kAuxCoherent>(rx, pix4 + nca_mul_u32((unsigned)min(4 * k + q4, C - 1), plane4), 0u);
            }
        }
        if (GOAL && has_goal) {
#pragma unroll
            for (int k = 0; k < CP / 4; ++k) {
                const int d = 4 * k - gch0;   // uniform
                if (EXACT && d >= 0) R.gh[k] = ST::template ld1<0>(rg, voh, (unsigned)d * plane4);
                else   // (C < CP: clamp to the goal tensor's last plane, as above)
                    R.gh[k] = ST::template ld1<0>(rg, pix4 + nca_mul_u32((unsigned)min(max(4 * k + q4 - gch0, 0), a.goal_ch - 1), plane4), 0u);
            }
        }
    }
}

// Resolve the pending life mask, build z = x + goal*pre in LDS (halo 1), keep the resolved state for
// the residual.  CHECK=false: no bounds logic.  KEEPX=false: the resolved-state copy XR is not written (callers that only
// want z, the masks and the fire mask: the backward's front kernel).
template <int CP, bool CHECK, bool EXACT = false, typename ST = StF32, bool KEEPX = true>
__device__ __forceinline__ void stage_tile(const NcaCondArgs& a, const WTile& t, const TileLds& L, int lane_in,
                                           const TileRegs<CP, ST>& R, int tile_no) {
    float* const Z = L.Z;
    float* const XR = L.XR;
    float* const A3 = L.A3;
    float* const PN = L.PN;
    float* const LIFE = L.LIFE;
    float* const A2 = L.A2;
    float* const MK = L.MK;

    const int C = EXACT ? CP : a.C, H = a.H, W = a.W;
    const unsigned plane = (unsigned)(H * W);
    const int gch0 = C - a.goal_ch, ty0 = t.ty0, tx0 = t.tx0;
    const bool pending = a.pre_in != nullptr, use_alive = a.alive_ch >= 0, has_goal = a.goal_ch > 0;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int hl = (lane >> 5) & 1, l5 = lane & 31, q4 = (lane >> 4) & 3, ci = lane & 15;   // masked: the opaque lane id has no known range, and index * constant would become a quarter-rate 32-bit multiply

    NCA_STAMP(8);
    // ---- S1: alpha' (-inf outside the image == max_pool2d padding) -----------------------------
    wave_sync();  // the previous tile's LDS reads are ordered before these writes
    bool l2ok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gy = ty0 - 2 + 2 * k + hl, gx = tx0 - 2 + l5;
        l2ok[k] = l5 < 20 && (!CHECK || (gy >= 0 && gy < H && gx >= 0 && gx < W));
    }
    if (use_alive) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int gy = ty0 - 3 + 2 * k + hl, gx = tx0 - 3 + l5;
            const bool ok = !CHECK || (gy >= 0 && gy < H && gx >= 0 && gx < W);
            if (l5 < 22) A3[(2 * k + hl) * RS + l5 + 1] = ok ? ST::cv1(R.a3v[k]) : NCA_NEG_INF;
        }
        wave_sync();
        NCA_STAMP(9);
        // ---- S2: life = pre & post of the PREVIOUS step, resolved alpha (nca.py:191-194) -------
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 2 * k + hl;
            const float* const ac = A3 + (r + 1) * RS + l5 + 2;
            float life = 0.0f, av = NCA_NEG_INF;
            if (l2ok[k]) {
                life = 1.0f;
                av = ac[0];
                if (pending) {
                    life = (R.prv[k] != 0u && max3x3(ac) > a.thr) ? 1.0f : 0.0f;
                    av = wclamp(av * life, a.lo, a.hi);
                }
            }
            if (l5 < 20) {
                LIFE[r * RS + l5 + 2] = life;
                A2[r * RS + l5 + 2] = av;
            }
        }
        wave_sync();
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (l5 < 20) LIFE[(2 * k + hl) * RS + l5 + 2] = l2ok[k] ? 1.0f : 0.0f;
    }
    NCA_STAMP(10);
    // ---- S3: pre-life mask of THIS step on halo 1; fire mask -------------------------------------
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int r = 2 * k + hl;  // halo-1 row, col q = l5 < 18
        const int gy = ty0 - 1 + r, gx = tx0 - 1 + l5;
        const bool in = l5 < 18 && (!CHECK || (gy >= 0 && gy < H && gx >= 0 && gx < W));
        float pn = 0.0f;
        if (in) pn = (!use_alive || max3x3(A2 + (r + 1) * RS + l5 + 3) > a.thr) ? 1.0f : 0.0f;
        if (l5 < 18) PN[r * RS + l5 + 3] = pn;
    }
    {
        const int cgy = ty0 + q4, cgx = tx0 + ci;
        const bool cin = !CHECK || (cgy < H && cgx < W);
        // the pick is a bit-mask the compiler cannot see through: a select would be sunk into the branches that
        // produce the two values, share one register, and force a vmcnt(0) in front of the Philox rounds
        unsigned um = a.u ? 0xFFFFFFFFu : 0u;
        asm volatile("" : "+v"(um));
        float uf = __uint_as_float(__float_as_uint(R.up) | (__float_as_uint(R.uu) & um));
        if (a.u_bits) {   // the cell's bit of the loaded word: 1 -> fires (0 < rate), 0 -> never (clamp(2) = 1 < rate is false)
            const unsigned cellg = (unsigned)((size_t)t.b * plane) + (cin ? (unsigned)(__mul24(cgy, W) + cgx) : 0u);
            uf = ((__float_as_uint(R.uu) >> (cellg & 31u)) & 1u) ? 0.0f : 2.0f;
        }
        MK[lane] = (cin && wclamp(uf, 0.0f, 1.0f) < a.fire_rate) ? 1.0f : 0.0f;  // nca.py:171-174
        wave_sync();
        if (cin && a.pre_out)
            __builtin_amdgcn_raw_buffer_store_b8((uint8_t)PN[(q4 + 1) * RS + ci + 4], nca_rsrc(a.pre_out + (size_t)t.b * plane),
                                                 __mul24(cgy, W) + cgx, 0, kAuxCoherent);
    }
    NCA_STAMP(11);
    // ---- S4: z = x + goal * pre (nca.py:177) on halo 1; resolved state kept for the residual -----
    if (l5 < 24) {
        const int fr = l5 >> 2, ff = l5 & 3;
        const bool fok = !CHECK || (ty0 - 1 + fr >= 0 && ty0 - 1 + fr < H && tx0 + 4 * ff + 3 < W);
        const f32x4 lf = ld4(LIFE + (fr + 1) * RS + 4 + 4 * ff);
        const f32x4 pn = ld4(PN + fr * RS + 4 + 4 * ff);
        const bool inner = fr >= 1 && fr <= WTH;
        f32x4 v[CP / 2];
#pragma unroll
        for (int k = 0; k < CP / 2; ++k) {  // resolved state (prefetched long ago): no memory wait here
            const int ch = 2 * k + hl;
            v[k] = ST::cv4(R.xf[k]);
            if (pending) {
                v[k] = v[k] * lf;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[k][j] = wclamp(v[k][j], a.lo, a.hi);
            }
            if ((CHECK && !fok) || ch >= C) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (KEEPX && inner) st4(XR + ch * XRS + (fr - 1) * WTW + 4 * ff, v[k]);
        }
        NCA_STAMP(12);
#pragma unroll
        for (int k = 0; k < CP / 2; ++k) {  // goal encoding (issued at the top of staging) consumed last
            const int ch = 2 * k + hl;
            if (has_goal && ch >= gch0 && ch < C && fok) v[k] = __builtin_elementwise_fma(ST::cv4(R.gf[k]), pn, v[k]);
            st4(Z + ch * CS + fr * RS + 4 + 4 * ff, v[k]);
        }
    }
    NCA_STAMP(13);
    if (ci < 12) {
        const int hr = ci >> 1, zq = (ci & 1) ? WTW + 4 : 3;
        const int hgy = ty0 - 1 + hr, hgx = (ci & 1) ? tx0 + WTW : tx0 - 1;
        const bool hok = !CHECK || (hgy >= 0 && hgy < H && hgx >= 0 && hgx < W);
        const float lf = LIFE[(hr + 1) * RS + zq], pn = PN[hr * RS + zq];
#pragma unroll
        for (int k = 0; k < CP / 4; ++k) {
            const int ch = 4 * k + q4;
            float v = ST::cv1(R.xh[k]);
            if (pending) v = wclamp(v * lf, a.lo, a.hi);
            if (!hok || ch >= C) v = 0.0f;
            else if (has_goal && ch >= gch0) v = fmaf(ST::cv1(R.gh[k]), pn, v);
            Z[ch * CS + hr * RS + zq] = v;
        }
    }
    wave_sync();
}

// Learned depthwise perception (nca.py:99-107) from the LDS tile: P[n][3c'+f] for channel 4c'+g, cell (row n, col ci).
template <int CP, int NT>
__device__ __forceinline__ void perceive_tile(const float* __restrict__ WS, const float* __restrict__ Z, int lane_in,
                                              int n0, float (&P)[NT][3 * CP / 4]) {
    using K = WCfg<CP>;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));  // per pass: re-read the 27 taps from LDS instead of holding 4x28 registers
    const float* const WPL = WS + K::OFF_WP;
    const int g = (lane >> 4) & 3, ci = lane & 15;
#pragma unroll
    for (int c4 = 0; c4 < CP / 4; ++c4) {
        const float* const zc = Z + (4 * c4 + g) * CS + n0 * RS + ci + 3;
        float wt[28];
#pragma unroll
        for (int j4 = 0; j4 < 7; ++j4) {
            const f32x4 w4 = ld4(WPL + (4 * c4 + g) * K::WPS + 4 * j4);
            wt[4 * j4 + 0] = w4[0]; wt[4 * j4 + 1] = w4[1]; wt[4 * j4 + 2] = w4[2]; wt[4 * j4 + 3] = w4[3];
        }
        static_assert(NT == 2, "row pairs: one v_pk_fma_f32 serves output rows n0 and n0+1");
        // tap (dy,dx) of output rows (n0, n0+1) reads tile rows (dy, dy+1): ONE ds_read2_b32 lands them in an aligned
        // register pair (the compiler otherwise assembles the pairs with v_mov's).  Inline-asm loads are not counted by
        // hipcc, so they are waited for explicitly before use (guide 5.7, form (ii)).
        const unsigned za = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)zc;
        f32x2 nb[9];
#define NCA_RD2(i, o) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(nb[i]) : "v"(za), "n"(o), "n"((o) + RS))
        NCA_RD2(0, 0); NCA_RD2(1, 1); NCA_RD2(2, 2);
        NCA_RD2(3, RS); NCA_RD2(4, RS + 1); NCA_RD2(5, RS + 2);
        NCA_RD2(6, 2 * RS); NCA_RD2(7, 2 * RS + 1); NCA_RD2(8, 2 * RS + 2);
#undef NCA_RD2
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]), "+v"(nb[4]), "+v"(nb[5]), "+v"(nb[6]), "+v"(nb[7]), "+v"(nb[8])
                     :: "memory");
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            f32x2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = __builtin_elementwise_fma(f32x2{wt[9 * f + t], wt[9 * f + t]}, nb[t], acc);
            P[0][3 * c4 + f] = acc[0];
            P[1][3 * c4 + f] = acc[1];
        }
        __builtin_amdgcn_sched_barrier(0);  // one channel group's 27 taps live at a time (else hipcc hoists all 4 x 28)
    }
}

// perceive_tile with the LDS operands of channel group c4+1 in flight while group c4 is computed (46 more registers:
// used where the wave has them, i.e. the bf16-operand consumer).  All 16 reads of a group are inline asm, so the one
// counter the hardware keeps is managed here: s_waitcnt lgkmcnt(16) = "everything but the group just issued has landed".
template <int CP, int NT>
__device__ __forceinline__ void perceive_tile_pipe(const float* __restrict__ WS, const float* __restrict__ Z, int lane_in,
                                                   int n0, float (&P)[NT][3 * CP / 4]) {
    using K = WCfg<CP>;
    static_assert(NT == 2, "row pairs");
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int g = (lane >> 4) & 3, ci = lane & 15;
    constexpr int NG = CP / 4;
    f32x4 wt[2][7];
    f32x2 nb[2][9];
    const unsigned wa0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(WS + K::OFF_WP + g * K::WPS);
    const unsigned za0 = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)(Z + g * CS + n0 * RS + ci + 3);
    auto issue = [&](int c4, int b) {
        const unsigned wa = wa0 + (unsigned)(4 * c4 * K::WPS * 4), za = za0 + (unsigned)(4 * c4 * CS * 4);
#define NCA_RDW(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wt[b][j]) : "v"(wa), "n"(16 * (j)))
        NCA_RDW(0); NCA_RDW(1); NCA_RDW(2); NCA_RDW(3); NCA_RDW(4); NCA_RDW(5); NCA_RDW(6);
#undef NCA_RDW
#define NCA_RD2(i, o) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(nb[b][i]) : "v"(za), "n"(o), "n"((o) + RS))
        NCA_RD2(0, 0); NCA_RD2(1, 1); NCA_RD2(2, 2);
        NCA_RD2(3, RS); NCA_RD2(4, RS + 1); NCA_RD2(5, RS + 2);
        NCA_RD2(6, 2 * RS); NCA_RD2(7, 2 * RS + 1); NCA_RD2(8, 2 * RS + 2);
#undef NCA_RD2
    };
    issue(0, 0);
#pragma unroll
    for (int c4 = 0; c4 < NG; ++c4) {
        const int b = c4 & 1;
        if (c4 + 1 < NG) {
            issue(c4 + 1, b ^ 1);
            asm volatile("s_waitcnt lgkmcnt(15)"   // 16 reads of the next group outstanding at most -> this group has landed
                         : "+v"(wt[b][0]), "+v"(wt[b][1]), "+v"(wt[b][2]), "+v"(wt[b][3]), "+v"(wt[b][4]), "+v"(wt[b][5]),
                           "+v"(wt[b][6]), "+v"(nb[b][0]), "+v"(nb[b][1]), "+v"(nb[b][2]), "+v"(nb[b][3]), "+v"(nb[b][4]),
                           "+v"(nb[b][5]), "+v"(nb[b][6]), "+v"(nb[b][7]), "+v"(nb[b][8])
                         :: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(wt[b][0]), "+v"(wt[b][1]), "+v"(wt[b][2]), "+v"(wt[b][3]), "+v"(wt[b][4]), "+v"(wt[b][5]),
                           "+v"(wt[b][6]), "+v"(nb[b][0]), "+v"(nb[b][1]), "+v"(nb[b][2]), "+v"(nb[b][3]), "+v"(nb[b][4]),
                           "+v"(nb[b][5]), "+v"(nb[b][6]), "+v"(nb[b][7]), "+v"(nb[b][8])
                         :: "memory");
        }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            f32x2 acc = {0.0f, 0.0f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float w = wt[b][(9 * f + t) >> 2][(9 * f + t) & 3];
                acc = __builtin_elementwise_fma(f32x2{w, w}, nb[b][t], acc);
            }
            P[0][3 * c4 + f] = acc[0];
            P[1][3 * c4 + f] = acc[1];
        }
    }
}

// relu as ONE integer max on the bit pattern (sign bit set <=> negative int): no NaN-canonicalising pre-pass.
__device__ __forceinline__ float relu(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

// UpdateNet (nca.py:40-46) 3C -> 64 -> 64 -> C on v_mfma_f32_16x16x4_f32 for rows n0..n0+NT-1, then
// x' = x + mask * out (nca.py:189) written back in place into the resolved-state copy XR.
template <int CP, int NT>
__device__ __forceinline__ void mlp_tile(const NcaCondArgs& a, const float* __restrict__ WS, float* __restrict__ XR,
                                         const float* __restrict__ MK, int lane_in, int n0,
                                         const float (&P)[NT][3 * CP / 4]) {
    using K = WCfg<CP>;
    int lane_o = lane_in;
    asm volatile("" : "+v"(lane_o));  // nothing below is hoisted out of the tile loop and kept live through staging
    // A-operand images in 16-byte form: element ((tile*S4 + s4)*64 + lane)*4 + j  <->  k-step 4*s4 + j
    const f32x4* const W1V = reinterpret_cast<const f32x4*>(WS + K::OFF_W1) + lane_o;   // [4 m][K1S4][64]
    const f32x4* const W2V = reinterpret_cast<const f32x4*>(WS + K::OFF_W2) + lane_o;   // [4 m2][4 m][64] (j = r)
    const f32x4* const W3V = reinterpret_cast<const f32x4*>(WS + K::OFF_W3) + lane_o;   // [M3T][4 m][64]
    const float* const B1L = WS + K::OFF_B1;
    const float* const B2L = WS + K::OFF_B2;
    const int g = (lane_o >> 4) & 3, ci = lane_o & 15;
    f32x4 acc2[4][NT];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x4 bias = ld4(B2L + 16 * m2 + 4 * g);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[m2][n] = bias;
    }
    // layer-1 operands of hidden tile m+1 are fetched while tile m computes; layer-2 operands of tile m are fetched
    // at the top of tile m (its layer-1 chain covers their latency); layer-3 operands during the last tile
    f32x4 a1[K::K1S4], a3[K::M3T][4];
#pragma unroll
    for (int q = 0; q < K::K1S4; ++q) a1[q] = W1V[q * 64];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 a1n[K::K1S4], a2[4];
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) a2[m2] = W2V[(m2 * 4 + m) * 64];
        if (m < 3) {
#pragma unroll
            for (int q = 0; q < K::K1S4; ++q) a1n[q] = W1V[((m + 1) * K::K1S4 + q) * 64];
        } else {
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) a3[m3][mm] = W3V[(m3 * 4 + mm) * 64];
        }
        const f32x4 bias = ld4(B1L + 16 * m + 4 * g);
        f32x4 acc1[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) acc1[n] = bias;
#pragma unroll
        for (int s = 0; s < K::K1S; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc1[n] = nca_mfma(a1[s >> 2][s & 3], P[n][s], acc1[n]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc2[m2][n] = nca_mfma(a2[m2][r], relu(acc1[n][r]), acc2[m2][n]);
        if (m < 3) {
#pragma unroll
            for (int q = 0; q < K::K1S4; ++q) a1[q] = a1n[q];
        }
    }
    f32x4 acc3[K::M3T][NT];
#pragma unroll
    for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc3[m3][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // out.4 has no bias
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc3[m3][n] = nca_mfma(a3[m3][m][r], relu(acc2[m][n][r]), acc3[m3][n]);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float mk = MK[(n0 + n) * WTW + ci];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = 16 * m3 + 4 * g + r;
                if (ch < CP) {  // same lane reads and rewrites the element: in place
                    float* const p = XR + ch * XRS + (n0 + n) * WTW + ci;
                    *p = fmaf(mk, acc3[m3][n][r], *p);
                }
            }
    }
}

// Register-resident A operands of the whole UpdateNet for one lane: 4*K1S4 + 16 + 4*M3T 16-byte values (128 VGPRs at
// CP=16).  A wave that only consumes tiles (nca_cond_pc.hip) loads them once per launch; its MFMA stream then contains no
// LDS reads at all -- with in-order issue every ds_read that lands next to its use stalls the matrix pipe for the full
// LDS latency (measured: 44.6 instead of 34.6 cycles per MFMA with LDS-fed operands).
template <int CP>
struct MlpRegs {
    using K = WCfg<CP>;
    static constexpr bool W3_LDS = CP > 16;   // wide channel counts: W3 stays in LDS (32 more operand registers do not fit)
    f32x4 w1[4][K::K1S4], w2[4][4], w3[W3_LDS ? 1 : K::M3T][4];
};
template <int CP>
__device__ __forceinline__ void mlp_load_regs(const float* __restrict__ WS, int lane, MlpRegs<CP>& R) {
    using K = WCfg<CP>;
    const f32x4* const W1V = reinterpret_cast<const f32x4*>(WS + K::OFF_W1) + lane;
    const f32x4* const W2V = reinterpret_cast<const f32x4*>(WS + K::OFF_W2) + lane;
    [[maybe_unused]] const f32x4* const W3V = reinterpret_cast<const f32x4*>(WS + K::OFF_W3) + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int q = 0; q < K::K1S4; ++q) R.w1[m][q] = W1V[(m * K::K1S4 + q) * 64];
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) R.w2[m2][m] = W2V[(m2 * 4 + m) * 64];
        if constexpr (!MlpRegs<CP>::W3_LDS) {
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3) R.w3[m3][m] = W3V[(m3 * 4 + m) * 64];
        }
    }
}
// mlp_tile with register-resident operands (same MFMA order per accumulator => bit-identical results).
// Issue order matters more than instruction count here: an exact-f32 MFMA does not co-execute with VALU work, so every
// VALU instruction that lands between two MFMAs drains the matrix pipe first (measured: ~25 cycles per isolated v_max,
// tools/micro/mlp_pass.hip).  The ReLUs are therefore issued as fenced groups of 8 / 32, and the layer-1 chain of hidden
// tile m+1 is issued BEFORE the ReLU group of tile m so that group never waits for the chain it reads.
// XR must provide 16*M3T channel rows (CP <= 16) / exactly CP rows (CP > 16: the last output tile's rows are guarded).
template <int CP, int NT>
__device__ __forceinline__ void mlp_tile_regs(const MlpRegs<CP>& Wr, const float* __restrict__ WS, float* __restrict__ XR,
                                              const float* __restrict__ MK, int lane_in, int n0,
                                              const float (&P)[NT][3 * CP / 4]) {
    using K = WCfg<CP>;
    constexpr bool WIDE = MlpRegs<CP>::W3_LDS;
    int lane_o = lane_in;
    asm volatile("" : "+v"(lane_o));
    const int g = (lane_o >> 4) & 3, ci = lane_o & 15;
    // accumulator seeds: eight LDS reads issued together before the MFMA stream (they live only during this phase, so
    // the perception before it has the registers to pipeline its own reads)
    f32x4 b1[4], b2[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        b1[m] = ld4(WS + K::OFF_B1 + 16 * m + 4 * g);
        b2[m] = ld4(WS + K::OFF_B2 + 16 * m + 4 * g);
    }
    // LDS offsets of the residual read-modify-write, computed (and pinned) here: left to the compiler their integer
    // multiplies land inside the MFMA stream next to the reads
    int xoff = 4 * g * XRS + n0 * WTW + ci, moff = n0 * WTW + ci;
    asm volatile("" : "+v"(xoff), "+v"(moff));
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc2[4][NT], acc1[NT], acc1n[NT];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[m2][n] = b2[m2];
    auto layer1 = [&](int m, f32x4 (&acc)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = b1[m];
#pragma unroll
        for (int s = 0; s < K::K1S; ++s)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = nca_mfma(Wr.w1[m][s >> 2][s & 3], P[n][s], acc[n]);
    };
    layer1(0, acc1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (m < 3) layer1(m + 1, acc1n);
        float h[NT][4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[n][r] = relu(acc1[n][r]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc2[m2][n] = nca_mfma(Wr.w2[m2][m][r], h[n][r], acc2[m2][n]);
        if (m < 3) {
#pragma unroll
            for (int n = 0; n < NT; ++n) acc1[n] = acc1n[n];
        }
    }
    // the residual operands (read-modify-write of the resolved state, same lane) are fetched before the last ReLU group
    float xr[NT][K::M3T][4], mk[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        mk[n] = MK[moff + n * WTW];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // CP <= 16: XR holds 16*M3T channel rows (rows >= CP are scratch): no per-lane guard, no exec masking
                if (!WIDE || 16 * m3 + 16 <= CP || 4 * g < CP - 16 * m3) xr[n][m3][r] = XR[xoff + (16 * m3 + r) * XRS + n * WTW];
                else xr[n][m3][r] = 0.0f;
            }
    }
    f32x4 w3l[K::M3T][4];   // WIDE: layer-3 operands from the resident image, requested here (they land under the ReLU group)
    if constexpr (WIDE) {
        const f32x4* const W3V = reinterpret_cast<const f32x4*>(WS + K::OFF_W3) + lane_o;
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int m = 0; m < 4; ++m) w3l[m3][m] = W3V[(m3 * 4 + m) * 64];
    }
    float h2[4][NT][4];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[m][n][r] = relu(acc2[m][n][r]);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc3[K::M3T][NT];
#pragma unroll
    for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc3[m3][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // out.4 has no bias
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if constexpr (WIDE) acc3[m3][n] = nca_mfma(w3l[m3][m][r], h2[m][n][r], acc3[m3][n]);
                    else acc3[m3][n] = nca_mfma(Wr.w3[m3][m][r], h2[m][n][r], acc3[m3][n]);
                }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!WIDE || 16 * m3 + 16 <= CP || 4 * g < CP - 16 * m3)
                    XR[xoff + (16 * m3 + r) * XRS + n * WTW] = fmaf(mk[n], acc3[m3][n][r], xr[n][m3][r]);
            }
}

// ---- UpdateNet on bf16 MFMA (bf16-storage kernels) -------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) { return StBF16::pk2(lo, hi); }
// relu on a packed bf16 pair: the sign bit of each half makes it a negative int16
__device__ __forceinline__ unsigned relu_pk(unsigned v) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), s16x2{0, 0}));
}
__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
    return __builtin_bit_cast(s16x4, u32x2{pk_bf16(a, b), pk_bf16(c, d)});
}
__device__ __forceinline__ s16x4 pack4_relu(f32x4 v) {
    return __builtin_bit_cast(s16x4, u32x2{relu_pk(pk_bf16(v[0], v[1])), relu_pk(pk_bf16(v[2], v[3]))});
}
__device__ __forceinline__ f32x4 mfma_bf16(s16x4 a, s16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}

template <int CP>
struct MlpRegsBf {
    static constexpr int K1S = 3 * CP / 4, KS1 = (K1S + 3) / 4, M3T = (CP + 15) / 16;   // k-steps of layer 1: slots q = 3*c4 + f
    s16x4 w1[4][KS1], w2[4][4], w3[M3T][4];
};
// A operands from the f32 weight tensors (nca.py:40-46 layouts), rounded to bf16.  Lane (g, i): row o = 16*tile + i,
// k = 4g + r.  Layer 1's k order follows perceive_tile: slot q = 4s + r = 3*c4 + f is channel 4*c4 + g, filter f.
// bf16 A operands from the workgroup's f32 A-operand image in LDS (built cooperatively: every weight is fetched from
// memory once per workgroup; loading them per wave straight from the tensors cost 19 K instead of 7 K cycles of cold start).  The image's k order of layer 1
// (k-step s = 4q + j <-> channel 4*(s/3) + g, filter s%3) is exactly the bf16 slot order q' = 4s' + r.
template <int CP>
__device__ __forceinline__ void load_weights_bf16_lds(const float* __restrict__ WS, int lane, MlpRegsBf<CP>& R) {
    using K = WCfg<CP>;
    using KB = MlpRegsBf<CP>;
    static_assert(KB::KS1 == K::K1S4 && KB::M3T == K::M3T, "image layout");
    const f32x4* const W1V = reinterpret_cast<const f32x4*>(WS + K::OFF_W1) + lane;
    const f32x4* const W2V = reinterpret_cast<const f32x4*>(WS + K::OFF_W2) + lane;
    const f32x4* const W3V = reinterpret_cast<const f32x4*>(WS + K::OFF_W3) + lane;
    auto pk = [](f32x4 v) { return pack4(v[0], v[1], v[2], v[3]); };
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int s = 0; s < KB::KS1; ++s) R.w1[m][s] = pk(W1V[(m * K::K1S4 + s) * 64]);
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) R.w2[m2][m] = pk(W2V[(m2 * 4 + m) * 64]);
#pragma unroll
        for (int m3 = 0; m3 < KB::M3T; ++m3) R.w3[m3][m] = pk(W3V[(m3 * 4 + m) * 64]);
    }
}

// UpdateNet for rows n0..n0+NT-1 of the tile, then x' = x + mask*out in place in XR (16*M3T channel rows).
template <int CP, int NT>
__device__ __forceinline__ void mlp_tile_bf16(const MlpRegsBf<CP>& Wr, const float* __restrict__ B1L, const float* __restrict__ B2L,
                                              float* __restrict__ XR, const float* __restrict__ MK, int lane_in, int n0,
                                              const float (&P)[NT][3 * CP / 4]) {
    using K = MlpRegsBf<CP>;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int g = (lane >> 4) & 3, ci = lane & 15;
    s16x4 pb[NT][K::KS1];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int s = 0; s < K::KS1; ++s) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 4 * s + r < K::K1S ? P[n][4 * s + r] : 0.0f;
            pb[n][s] = pack4(v[0], v[1], v[2], v[3]);
        }
    f32x4 acc2[4][NT];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x4 b = ld4(B2L + 16 * m2 + 4 * g);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[m2][n] = b;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 b = ld4(B1L + 16 * m + 4 * g);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 acc1 = b;
#pragma unroll
            for (int s = 0; s < K::KS1; ++s) acc1 = mfma_bf16(Wr.w1[m][s], pb[n][s], acc1);
            const s16x4 hb = pack4_relu(acc1);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2][n] = mfma_bf16(Wr.w2[m2][m], hb, acc2[m2][n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x4 acc3[K::M3T];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3) acc3[m3] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // out.4 has no bias
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const s16x4 hb = pack4_relu(acc2[m][n]);
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3) acc3[m3] = mfma_bf16(Wr.w3[m3][m], hb, acc3[m3]);
        }
        const float mk = MK[(n0 + n) * WTW + ci];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (CP <= 16 || 16 * m3 + 16 <= CP || 4 * g < CP - 16 * m3) {   // CP > 16: XR holds exactly CP rows
                    float* const p = XR + (16 * m3 + 4 * g + r) * XRS + (n0 + n) * WTW + ci;
                    *p = fmaf(mk, acc3[m3][r], *p);
                }
            }
    }
}


// ---- UpdateNet with fp32 operands emulated by bf16 pairs ("bf16x3") --------------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) keeps 16 significand bits; W likewise.  W*x ~ Whi*xhi + Whi*xlo +
// Wlo*xhi on v_mfma_f32_16x16x16_bf16 (products exact, f32 accumulation): relative error per product ~2^-17 (the dropped
// Wlo*xlo term and the two truncations), i.e. ~1e-5 -- inside the 1e-4 parity bar but NOT the exact-f32 step; opt-in
// (ncahip_cond_precision).  Three bf16 MFMAs cost 24-48 cycles against 128 for the four exact-f32 ones they replace, and
// they co-execute with the vector ALU.
template <int CP>
struct MlpRegsSplit {
    MlpRegsBf<CP> hi, lo;
};
__device__ __forceinline__ f32x4 widen4(s16x4 b) {
    const u32x2 v = __builtin_bit_cast(u32x2, b);
    return f32x4{__uint_as_float(v[0] << 16), __uint_as_float(v[0] & 0xffff0000u), __uint_as_float(v[1] << 16),
                 __uint_as_float(v[1] & 0xffff0000u)};
}
__device__ __forceinline__ void split4(f32x4 v, s16x4& hi, s16x4& lo) {
    hi = pack4(v[0], v[1], v[2], v[3]);
    const f32x4 d = v - widen4(hi);
    lo = pack4(d[0], d[1], d[2], d[3]);
}
template <int CP>
__device__ __forceinline__ void load_weights_split_lds(const float* __restrict__ WS, int lane, MlpRegsSplit<CP>& R) {
    using K = WCfg<CP>;
    using KB = MlpRegsBf<CP>;
    const f32x4* const W1V = reinterpret_cast<const f32x4*>(WS + K::OFF_W1) + lane;
    const f32x4* const W2V = reinterpret_cast<const f32x4*>(WS + K::OFF_W2) + lane;
    const f32x4* const W3V = reinterpret_cast<const f32x4*>(WS + K::OFF_W3) + lane;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int s = 0; s < KB::KS1; ++s) split4(W1V[(m * K::K1S4 + s) * 64], R.hi.w1[m][s], R.lo.w1[m][s]);
#pragma unroll
        for (int m2 = 0; m2 < 4; ++m2) split4(W2V[(m2 * 4 + m) * 64], R.hi.w2[m2][m], R.lo.w2[m2][m]);
#pragma unroll
        for (int m3 = 0; m3 < KB::M3T; ++m3) split4(W3V[(m3 * 4 + m) * 64], R.hi.w3[m3][m], R.lo.w3[m3][m]);
    }
}
__device__ __forceinline__ f32x4 mfma_split(s16x4 whi, s16x4 wlo, s16x4 xhi, s16x4 xlo, f32x4 c) {
    c = mfma_bf16(whi, xlo, c);   // small terms first
    c = mfma_bf16(wlo, xhi, c);
    return mfma_bf16(whi, xhi, c);
}
template <int CP, int NT>
__device__ __forceinline__ void mlp_tile_split(const MlpRegsSplit<CP>& Wr, const float* __restrict__ B1L, const float* __restrict__ B2L,
                                               float* __restrict__ XR, const float* __restrict__ MK, int lane_in, int n0,
                                               const float (&P)[NT][3 * CP / 4]) {
    using K = MlpRegsBf<CP>;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int g = (lane >> 4) & 3, ci = lane & 15;
    s16x4 ph[NT][K::KS1], pl[NT][K::KS1];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int s = 0; s < K::KS1; ++s) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 4 * s + r < K::K1S ? P[n][4 * s + r] : 0.0f;
            split4(v, ph[n][s], pl[n][s]);
        }
    f32x4 acc2[4][NT];
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2) {
        const f32x4 b = ld4(B2L + 16 * m2 + 4 * g);
#pragma unroll
        for (int n = 0; n < NT; ++n) acc2[m2][n] = b;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const f32x4 b = ld4(B1L + 16 * m + 4 * g);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 acc1 = b;
#pragma unroll
            for (int s = 0; s < K::KS1; ++s) acc1 = mfma_split(Wr.hi.w1[m][s], Wr.lo.w1[m][s], ph[n][s], pl[n][s], acc1);
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = relu(acc1[r]);
            s16x4 hh, hl;
            split4(h, hh, hl);
#pragma unroll
            for (int m2 = 0; m2 < 4; ++m2) acc2[m2][n] = mfma_split(Wr.hi.w2[m2][m], Wr.lo.w2[m2][m], hh, hl, acc2[m2][n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        f32x4 acc3[K::M3T];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3) acc3[m3] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};   // out.4 has no bias
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = relu(acc2[m][n][r]);
            s16x4 hh, hl;
            split4(h, hh, hl);
#pragma unroll
            for (int m3 = 0; m3 < K::M3T; ++m3) acc3[m3] = mfma_split(Wr.hi.w3[m3][m], Wr.lo.w3[m3][m], hh, hl, acc3[m3]);
        }
        const float mk = MK[(n0 + n) * WTW + ci];
#pragma unroll
        for (int m3 = 0; m3 < K::M3T; ++m3)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* const p = XR + (16 * m3 + 4 * g + r) * XRS + (n0 + n) * WTW + ci;
                *p = fmaf(mk, acc3[m3][r], *p);
            }
    }
}

// Pending state out: 16-byte stores, 4 per lane (item k -> channel 4k+q4, row (lane>>2)&3, group lane&3).
// WT: write-through at agent scope (sc1) -- the line goes to memory now instead of sitting dirty in the XCD's L2 until
// the end-of-kernel write-back (measured: -4 us launch cadence; a plain `nt` hint changes nothing).
template <int CP, bool CHECK, bool EXACT = false, bool WT = false, typename ST = StF32>
__device__ __forceinline__ void store_tile(const NcaCondArgs& a, const WTile& t, const float* __restrict__ XR, int lane_in) {
    constexpr unsigned SB = ST::BYTES;
    const int C = EXACT ? CP : a.C, H = a.H, W = a.W;
    const unsigned plane = (unsigned)(H * W);
    unsigned plane4 = plane * SB;
    asm volatile("" : "+s"(plane4));   // see issue_loads
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int q4 = (lane >> 4) & 3, row = (lane >> 2) & 3, ff = lane & 3;
    const int gy = t.ty0 + row, gx = t.tx0 + 4 * ff;
    const bool ok = !CHECK || (gy < H && gx + 3 < W);
    const __amdgpu_buffer_rsrc_t ro = nca_rsrc(reinterpret_cast<char*>(a.x_out) + (size_t)t.b * C * plane * SB);
    const unsigned vo = (ok ? (unsigned)(__mul24(gy, W) + gx) * SB : 0u) + nca_mul_u32((unsigned)q4, plane4);
    wave_sync();
#pragma unroll
    for (int k = 0; k < CP / 4; ++k) {
        const int ch = 4 * k + q4;
        const f32x4 v = ld4(XR + ch * XRS + row * WTW + 4 * ff);
        if (ok && ch < C) ST::template st4<WT ? 16 : 0>(ro, vo, 4u * k * plane4, v);
    }
}

}  // namespace
