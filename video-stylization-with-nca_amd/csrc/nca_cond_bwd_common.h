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

// slab layout (floats), runtime C / hidden: [w1 hid*3C | w2 hid*hid | w3 C*hid | b1 hid | b2 hid]
__host__ __device__ inline int slab_off_w2(int C, int hid) { return hid * 3 * C; }
__host__ __device__ inline int slab_off_w3(int C, int hid) { return slab_off_w2(C, hid) + hid * hid; }
__host__ __device__ inline int slab_off_b1(int C, int hid) { return slab_off_w3(C, hid) + C * hid; }
__host__ __device__ inline int slab_off_b2(int C, int hid) { return slab_off_b1(C, hid) + hid; }
__host__ __device__ inline int slab_floats(int C, int hid) { return slab_off_b2(C, hid) + hid; }

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
