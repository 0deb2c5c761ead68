// nca_cond_bwd_common.h -- pieces shared by the two forms of backward kernel A (nca_cond_bwd.hip: one launch; nca_cond_bwd_fm.hip:
// front kernel + matrix kernel): LDS carve, slab layout, operand streaming helper, the bf16 transposition buffer.
#pragma once
#include "nca_cond_tile.h"

#if defined(NCA_STAMPS)
unsigned long long* nca_debug_stamp_ptr();
#endif

namespace {

constexpr int kBwdWaves = 4, kBwdThreads = 256;
constexpr int TBS = 148;  // transposition buffer: [16 cells][TBS rows]; 148 % 32 == 20 -> conflict-free ds_write_b128 per cell

template <int CP>
struct BCfg {
    using F = WCfg<CP>;
    static constexpr int K1S = F::K1S;
    static constexpr int MJ = (3 * CP + 15) / 16;           // 16-row tiles of the perception index
    static constexpr int OFF_W3T = F::SHARED;               // [4 m][4 s][64]
    static constexpr int OFF_W1T = OFF_W3T + 4 * 4 * 64;    // [MJ][16 s][64]
    static constexpr int SHARED = OFF_W1T + MJ * 16 * 64;
    static constexpr int PW_TB = F::PW;                     // 16 cells x 148 (>= 128 activation rows)
    static constexpr int PW_A1 = PW_TB + 16 * TBS;          // alpha'_t halo 1: 6 x RS
    static constexpr int PW = PW_A1 + ZROWS * RS;
    static constexpr int LDS_FLOATS = SHARED + kBwdWaves * PW;
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS budget");
    static_assert(CP <= 16, "one 16-row output tile (M3T == 1)");
};

// Slab layout (floats) of one workgroup's weight-gradient partial -- TILE-MAJOR, i.e. the accumulator registers as they stand:
//   section 1:  W1 tiles [4 ma][MJ nb][64 lanes][4 r]           lane (g, ci) of tile (ma, nb) holds rows 16 ma + 4 g + r, column 16 nb + ci
//   section 2:  W2 tiles [4 ma][4 nb][64][4] | W3 tiles [M3T m3][4 nb][64][4] | b1 [64] | b2 [64]
// The flush writes every accumulator with ONE 16-byte LDS store, sums the four partials and read-modify-writes the slab with 16-byte
// accesses; nothing in it depends on C / hidden.  (The reference-layout version spent 24 K cycles per launch in scattered 4-byte LDS
// stores behind per-element index arithmetic and bounds tests -- 14 % of the bf16 matrix kernel.)  The permutation to the reference
// layouts happens ONCE per backward pass, after the slabs of all workgroups are summed (cond_bwd_unpermute_kernel).
template <int MJ_, int M3T_>
struct SlabTM {
    static constexpr int MJ = MJ_, M3T = M3T_;
    static constexpr int S1 = 1024 * MJ;
    static constexpr int OFF_W3 = 4096, OFF_B1 = OFF_W3 + 1024 * M3T, OFF_B2 = OFF_B1 + 64, S2 = OFF_B2 + 64;
    static constexpr int SF = S1 + S2;
    static constexpr int SECMAX = S1 > 4096 ? S1 : 4096;     // largest of the three flush sections (W1 | W2 | W3 + biases)
    static constexpr int stage(int np) { return np * SECMAX; }   // LDS floats the flush needs (np partials of one section)
};
__host__ __device__ inline int slab_mj(int cp) { return (3 * cp + 15) / 16; }
__host__ __device__ inline int slab_m3t(int cp) { return (cp + 15) / 16; }
__host__ __device__ inline int slab_floats_cp(int cp) { return 1024 * slab_mj(cp) + 4096 + 1024 * slab_m3t(cp) + 128; }

// Operand streaming for one wave per SIMD: with nobody to switch to, an LDS read issued right before its MFMAs costs the
// whole LDS round trip (and that is where the compiler's scheduler puts it, to save registers).  piped() runs N steps with
// the operands of step i+1 requested before the MFMAs of step i are issued; the scheduling fences keep that order, and the
// compiler's own wait insertion then only waits for the older request.
// d * 1[h > 0], with the compare pinned to the point of use: left free, the compiler evaluates all 64 compares of a layer
// where h is produced and carries the lane masks in SGPR pairs across the layer (spilled, one VALU op each way).
__device__ __forceinline__ float gate_pos(float h, float d) {
    asm volatile("" : "+v"(h));
    return h > 0.0f ? d : 0.0f;
}
template <int GSZ>
struct OpN { float v[GSZ]; };
#define NCA_FENCE() __builtin_amdgcn_sched_barrier(0)
// Sum over the 16 lanes of a DPP row (the cell lanes of one channel group), left in every lane: four vector adds with lane
// permutes done by the VALU's data-parallel primitives -- xor 1, xor 2 inside the quads, then the half-row and the row mirrored.
// Same tree, bit for bit, as the xor-butterfly over __shfl_xor it replaces (which is four dependent LDS-crossbar round trips).
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}
// LDS hand-off between the waves of a workgroup that does NOT wait for this wave's outstanding global loads / stores (the compiler's
// __syncthreads() drains vmcnt as well: at the flush that is the slab prefetch and the last tile's stores, ~6 us of pure latency).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The flush of one workgroup: NP waves hold partial accumulators (NP = 4: one wave per SIMD; NP = 8: two waves per SIMD, waves 2p and
// 2p + 1 share a tile); wave `part` stages its registers, every thread takes part in the read-modify-write.  Three sections (W1 | W2 |
// W3 + biases) so that NP partials of a section fit the LDS.  Sums in a fixed order (deterministic), element by element:
// slab += (p0 + p1) + (p2 + p3)   [+ ((p4 + p5) + (p6 + p7)) as the second operand of the outer sum for NP = 8: the pair sums first].
template <typename TM, int THREADS, int NP>
__device__ __forceinline__ void slab_flush_tm(float* __restrict__ smem, float* __restrict__ slab, int tid, int lane, int part,
                                              const f32x4 (&aW1)[4][TM::MJ], const f32x4 (&aW2)[4][4], const f32x4 (&aW3)[TM::M3T][4],
                                              const float (&db1)[4][4], const float (&db2)[4][4]) {
    static_assert(NP == 4 || NP == 8, "partials per workgroup");
    constexpr int N4 = TM::SF / 4, PER4 = (N4 + THREADS - 1) / THREADS;
    constexpr int SEC_OFF[3] = {0, TM::S1, TM::S1 + 4096}, SEC_LEN[3] = {TM::S1, 4096, TM::S2 - 4096};
    f32x4* const slab4 = reinterpret_cast<f32x4*>(slab);
    // the read half of the read-modify-write goes out first: one memory round trip, under the LDS staging below
    f32x4 cur[PER4];
#pragma unroll
    for (int k = 0; k < PER4; ++k) {
        const int i4 = tid + THREADS * k;
        cur[k] = i4 < N4 ? slab4[i4] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sec = 0; sec < 3; ++sec) {
        constexpr int dummy = 0; (void)dummy;
        const int LEN = SEC_LEN[sec], OFF = SEC_OFF[sec];
        lds_barrier();   // first section: tiles and weight images are dead in every wave; later: the previous section has been summed
        float* const sp = smem + part * LEN;
        if (sec == 0) {
#pragma unroll
            for (int ma = 0; ma < 4; ++ma)
#pragma unroll
                for (int nb = 0; nb < TM::MJ; ++nb) st4(sp + ((ma * TM::MJ + nb) * 64 + lane) * 4, aW1[ma][nb]);
        } else if (sec == 1) {
#pragma unroll
            for (int ma = 0; ma < 4; ++ma)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) st4(sp + ((ma * 4 + nb) * 64 + lane) * 4, aW2[ma][nb]);
        } else {
#pragma unroll
            for (int m3 = 0; m3 < TM::M3T; ++m3)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) st4(sp + ((m3 * 4 + nb) * 64 + lane) * 4, aW3[m3][nb]);
            const int g = (lane >> 4) & 3, ci = lane & 15;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t1 = row16_sum(db1[m][r]), t2 = row16_sum(db2[m][r]);
                    if (ci == 0) {
                        sp[1024 * TM::M3T + 16 * m + 4 * g + r] = t1;
                        sp[1024 * TM::M3T + 64 + 16 * m + 4 * g + r] = t2;
                    }
                }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < PER4; ++k) {
            const int i4 = tid + THREADS * k, j4 = i4 - OFF / 4;
            if (j4 >= 0 && j4 < LEN / 4) {
                const float* const q = smem + 4 * j4;
                f32x4 v = (ld4(q) + ld4(q + LEN)) + (ld4(q + 2 * LEN) + ld4(q + 3 * LEN));
                if constexpr (NP == 8) v = v + ((ld4(q + 4 * LEN) + ld4(q + 5 * LEN)) + (ld4(q + 6 * LEN) + ld4(q + 7 * LEN)));
                slab4[i4] = cur[k] + v;
            }
        }
    }
}

template <int N, typename LD, typename MM>
__device__ __forceinline__ void piped(LD&& ld, MM&& mm) {
    auto cur = ld(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        auto nxt = cur;
        if (i + 1 < N) nxt = ld(i + 1);
        NCA_FENCE();
        mm(i, cur);
        NCA_FENCE();
        cur = nxt;
    }
}

#if defined(NCA_STAMPS)
// diagnostic build: cycles per phase, summed over the wave's tiles -> dbg[(wg*4+wave)*16 + phase]
#define NCA_BPHASE(i)                                                                       \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        ph_acc[i] += t_ - ph_last;                                                          \
        ph_last = t_;                                                                       \
    } while (0)
#else
#define NCA_BPHASE(i) do { } while (0)
#endif

// ST = storage type of the history (states / pending states) and of the goal encoding: StF32, or StBF16 for a bf16 pool
// (BASELINE configs[2]): the values are widened exactly on load and the whole recomputation and every gradient stay f32.
// BFM: the matrix products (forward recomputation, data path, weight gradients) on bf16 MFMA (v_mfma_f32_16x16x16_bf16, f32
// accumulation) with the rounding points of the bf16 forward kernel (perception vector, hidden activations, weights; see
// include/ncahip.h): 92 bf16 MFMAs of 8 cycles per 16 cells instead of 368 exact-f32 ones of 32.  Cell-axis-as-K operands
// are transposed through LDS with ds_read_b64_tr_b16.  Everything that is not a matrix product (staging, gating, masks, the
// stencil kernel B, every stored gradient) stays f32.
typedef short bf_s16x4 __attribute__((ext_vector_type(4)));
// transposition buffer of the BFM path: [16 cells][TBH halfwords], 8 tiles of 16 features (32 B) per row + 16 B pad: pitch
// 72 dwords makes the transposed reads conflict-free; the 8-byte chunk of a tile is XOR-swizzled by (cell >> 2) so the
// 8-byte accumulator-layout writes are conflict-free too.
constexpr int TBH = 144;
__device__ __forceinline__ bf_s16x4 tb_tr_read(const short* tb, int tile, int lane) {
    // operand [feature i][cells 4g .. 4g+3] of a 16-feature tile: lane 4q+p of group g supplies row (cell) 4g+q, chunk p
    const int g = (lane >> 4) & 3, i = lane & 15, q = i >> 2, p = i & 3;
    const short* a = tb + (4 * g + q) * TBH + tile * 16 + 4 * (p ^ g);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf_s16x4 __attribute__((address_space(3)))*)(a));
}
// element r of a packed bf16x4 is non-zero (ReLU'd activations: non-zero <=> the f32 pre-activation was positive, but for values
// below the bf16 denormal range)
__device__ __forceinline__ bool bf_nz(bf_s16x4 v, int r) {
    const u32x2 w = __builtin_bit_cast(u32x2, v);
    return (w[r >> 1] & ((r & 1) ? 0x7fff0000u : 0x00007fffu)) != 0u;
}
__device__ __forceinline__ void tb_write(short* tb, int tile, int lane, bf_s16x4 v) {
    // accumulator layout: lane (g, cell c) holds features 4g .. 4g+3 of the tile for its cell: chunk g of row c
    const int g = (lane >> 4) & 3, c = lane & 15;
    *reinterpret_cast<bf_s16x4*>(tb + c * TBH + tile * 16 + 4 * (g ^ ((c >> 2) & 3))) = v;
}
__device__ __forceinline__ void tb_write_chunk(short* tb, int chunk, int lane, bf_s16x4 v) {   // chunk = 4 * tile + position
    const int c = lane & 15;
    *reinterpret_cast<bf_s16x4*>(tb + c * TBH + (chunk >> 2) * 16 + 4 * ((chunk & 3) ^ ((c >> 2) & 3))) = v;
}


}  // namespace
