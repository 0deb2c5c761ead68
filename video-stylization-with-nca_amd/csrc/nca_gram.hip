// nca_gram.hip -- weight-gradient products of the DyNCA backward with the CELL axis as K (gfx950, exact fp32 MFMA).
//
// One backward step of DyNCA (autograd through ConditioneDyNCA/models/dynca.py:117-138) needs
//     dW1 = dh y^T   [fc, 4C+c_cond]      db1 = sum_cells dh          (dh, y: per-cell columns, K = B*H*W cells)
//     dW2 = dO h^T   [C, fc]              db2 = sum_cells dO
// i.e. products out[i][j] = sum_n A[i][n] * B[j][n] of two row sets stored channel-major ([B, rows, H*W], exactly as the
// step kernels write them).  Library GEMMs see an M x N of ~128 x 67 with K = 524 288 and need transposed copies; here the
// operands stream through LDS once, coalesced along the cell axis, and every wave keeps its share of the out tiles in
// accumulator registers for the whole launch (per-workgroup partials, summed in fixed order afterwards: deterministic).
// The B rows may come from two tensors (the perception [B,4C,HW] and the conditioning map [B,c_cond,HW]): no concatenation.
#include <cstdlib>

#include "nca_common.h"
#include "nca_kernels.h"

namespace {

constexpr int kGramThreads = 256, kGramChunk = 64, kGramLS = 66;   // 64 cells per chunk; LDS row stride 66: operand reads A[row ci][k = g] hit bank (2 ci + g) % 32 -- conflict-free per 32-lane half (68 was 2-way)

struct GramArgs {
    const float* a;    // [B, ma, HW]
    const float* b1;   // [B, nb1, HW]
    const float* b2;   // [B, nb2, HW] or null
    float* ws;         // [grid, ma*nb + ma]
    int ma, nb1, nb2, B, HW;
};

// ROWSPLIT: wave w owns out row tiles [w*RT, (w+1)*RT) x all CT column tiles; else all RT row tiles x column tiles
// [w*CT, (w+1)*CT).  MA_T x NB_T 16-row tiles are staged per chunk.
template <int RT, int CT, bool ROWSPLIT, bool ONES>
__global__ __launch_bounds__(kGramThreads, 2) void gram_rows_kernel(const GramArgs a) {
    constexpr int MA_T = ROWSPLIT ? 4 * RT : RT, NB_T = ROWSPLIT ? CT : 4 * CT;
    constexpr int ROWS = 16 * (MA_T + NB_T), PER = ROWS / 4;   // rows staged per chunk; rows per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];   // ROWS * kGramLS floats (74 KB at CT = 9: dynamic)
    const int tid = threadIdx.x, lane = tid & 63, g = (lane >> 4) & 3, ci = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = a.nb1 + a.nb2, HW = a.HW;
    const int cpb = (HW + kGramChunk - 1) / kGramChunk, total = a.B * cpb;

    // Row sums of A (the bias gradient): either vector adds beside the MFMAs, or ONE MORE B COLUMN of ones riding on the MFMAs
    // (needs nb < 16 * NB_T).  Which is faster depends on the shape (see launch_gram).
    constexpr bool ones_col = ONES;      // the launcher picks ONES = (nb < 16 * NB_T)
    f32x4 acc[RT][CT];
    float rs[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        rs[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // row r of the staged set: r < 16*MA_T -> A row r, else B row r - 16*MA_T (rows beyond the real counts read as zero)
    float pre[PER];
    // Loads are unconditional (row and cell indices clamped into the tensors) -- per-row predicates cost an exec-mask dance or
    // a branch per row.
    // Buffer loads: the lane's cell offset is the only vector operand, every row's offset is a scalar (no vector address
    // arithmetic per load: it would sit in the same issue slots as the exact-f32 MFMAs).  Needs each batch item's rows
    // within 4 GiB, which the launcher checks.
    auto issue = [&](int c) {
        const int bi = c / cpb;
        const unsigned vo = (unsigned)min((c - bi * cpb) * kGramChunk + lane, HW - 1) * 4u;
        // opaque per chunk: the row offsets r*HW are recomputed with a scalar multiply each instead of being hoisted out of
        // the chunk loop as ~100 SGPRs of loop invariants (and spilled)
        unsigned hw4 = (unsigned)HW * 4u;
        asm volatile("" : "+s"(hw4));
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.a + (size_t)bi * a.ma * HW), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.b1 + (size_t)bi * a.nb1 * HW), 0, -1, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.b2 ? a.b2 + (size_t)bi * a.nb2 * HW : a.b1 + (size_t)bi * a.nb1 * HW), 0, -1, 0x00020000);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int r = 4 * k + wave;
            if (k < 4 * MA_T) {                       // compile-time split: the first 16*MA_T staged rows are A rows
                pre[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)vo, (int)((unsigned)min(r, a.ma - 1) * hw4), 0));
            } else {
                const int rb = r - 16 * MA_T;
                if (rb < a.nb1)
                    pre[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb1, (int)vo, (int)((unsigned)rb * hw4), 0));
                else
                    pre[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rb2, (int)vo, (int)((unsigned)max(min(rb, nb - 1) - a.nb1, 0) * hw4), 0));
            }
        }
    };
    int c = blockIdx.x;
    if (c < total) issue(c);
    for (; c < total; c += gridDim.x) {
        // Rows beyond the real counts hold copies of the last real row: they only reach output rows / columns that are never
        // written.  Cells beyond HW (last chunk of a batch item) would reach every output: zeroed here.
        const int cell0 = (c % cpb) * kGramChunk;
        // (ONES: staged B row nb is the ones column -- a scalar select at store time, the load of that slot is simply unused)
        const int ones_row = ONES ? 16 * MA_T + nb : -1;
        if (cell0 + kGramChunk <= HW) {
#pragma unroll
            for (int k = 0; k < PER; ++k) lds[(4 * k + wave) * kGramLS + lane] = (4 * k + wave == ones_row) ? 1.0f : pre[k];
        } else {
            const bool in = cell0 + lane < HW;
#pragma unroll
            for (int k = 0; k < PER; ++k) lds[(4 * k + wave) * kGramLS + lane] = in ? ((4 * k + wave == ones_row) ? 1.0f : pre[k]) : 0.0f;
        }
        __syncthreads();
        if (c + (int)gridDim.x < total) issue(c + gridDim.x);   // next chunk's rows fly during this chunk's products
        const float* const ar = lds + (16 * (ROWSPLIT ? wave * RT : 0) + ci) * kGramLS + g;
        const float* const br = lds + (16 * (MA_T + (ROWSPLIT ? 0 : wave * CT)) + ci) * kGramLS + g;
        // operand reads one k-step ahead of their MFMAs (an LDS read issued right before its use stalls the in-order pipe
        // for the whole round trip; the fences keep the compiler from sinking the reads back down)
        float av[2][RT], bv[2][CT];
        auto fetch = [&](int s, int q) {
#pragma unroll
            for (int i = 0; i < RT; ++i) av[q][i] = ar[16 * i * kGramLS + 4 * s];
#pragma unroll
            for (int j = 0; j < CT; ++j) bv[q][j] = br[16 * j * kGramLS + 4 * s];
        };
        fetch(0, 0);
#pragma unroll
        for (int s = 0; s < kGramChunk / 4; ++s) {
            const int q = s & 1;
            if (s + 1 < kGramChunk / 4) fetch(s + 1, q ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < RT; ++i) {
#pragma unroll
                for (int j = 0; j < CT; ++j) acc[i][j] = nca_mfma(av[q][i], bv[q][j], acc[i][j]);
            }
            if (!ones_col) {
#pragma unroll
                for (int i = 0; i < RT; ++i) rs[i] += av[q][i];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    // ---- this workgroup's partial: [ma x nb] products, then ma row sums ----------------------------------------------
    float* const ws = a.ws + (size_t)blockIdx.x * ((size_t)a.ma * nb + a.ma);
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int rt = ROWSPLIT ? wave * RT + i : i;
#pragma unroll
        for (int j = 0; j < CT; ++j) {
            const int col = 16 * (ROWSPLIT ? j : wave * CT + j) + ci;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * rt + 4 * g + r;
                if (o < a.ma && col < nb) ws[(size_t)o * nb + col] = acc[i][j][r];
                if (ones_col && o < a.ma && col == nb) ws[(size_t)a.ma * nb + o] = acc[i][j][r];   // row sums = the ones column
            }
        }
        if (!ones_col) {
            float v = rs[i];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int o = 16 * rt + ci;
            if (g == 0 && o < a.ma && (ROWSPLIT || wave == 0)) ws[(size_t)a.ma * nb + o] = v;
        }
    }
}

template <int RT, int CT, bool ROWSPLIT, bool ONES>
hipError_t launch_gram_o(const GramArgs& a, int grid, hipStream_t st) {
    constexpr int MA_T = ROWSPLIT ? 4 * RT : RT, NB_T = ROWSPLIT ? CT : 4 * CT;
    constexpr size_t lds = (size_t)16 * (MA_T + NB_T) * kGramLS * sizeof(float);
    auto kern = gram_rows_kernel<RT, CT, ROWSPLIT, ONES>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kGramThreads), lds, st, a);
    return hipGetLastError();
}

template <int RT, int CT, bool ROWSPLIT>
hipError_t launch_gram(const GramArgs& a, int grid, hipStream_t st) {
    constexpr int NB_T = ROWSPLIT ? CT : 4 * CT;
    // Measured in one process, interleaved (tools/ab_gram.py): the ones column wins at 9 column tiles (ma = 128, nb = 131: 270 -> 230 us)
    // and loses at 5 (nb = 67: 142 -> 150 us; nb = 51: 125 -> 134 us), so it is the default only for the wide shape.
    const char* const e = getenv("NCAHIP_GRAM_ONES");      // A/B hook: 0 / 1 force a form
    const bool want = e ? e[0] == '1' : CT >= 9;
    return (want && a.nb1 + a.nb2 < 16 * NB_T) ? launch_gram_o<RT, CT, ROWSPLIT, true>(a, grid, st) : launch_gram_o<RT, CT, ROWSPLIT, false>(a, grid, st);
}

}  // namespace

int nca_gram_grid(int B, int HW) {
    const int cus = nca_cu_count();
    const long total = (long)B * ((HW + kGramChunk - 1) / kGramChunk);
    return (int)(total < 2L * cus ? total : 2L * cus);
}

// out[ma*nb + ma] (+)= [products | row sums of A]; ws holds nca_gram_grid(B, HW) partials of that size.
// Shapes: ma <= 128 with nb <= 144 (dW1-like: rows split over the waves), or ma <= 32 with nb <= 128 (dW2-like).
hipError_t nca_launch_gram_rows(const float* a, int ma, const float* b1, int nb1, const float* b2, int nb2, int B, int HW,
                                float* out, float* ws, hipStream_t st, bool accumulate) {
    const int nb = nb1 + nb2, grid = nca_gram_grid(B, HW);
    const GramArgs ga{a, b1, b2, ws, ma, nb1, nb2, B, HW};
    hipError_t e = hipErrorInvalidValue;
    const size_t lim = (size_t)1 << 32;   // scalar byte offsets inside one batch item
    if ((size_t)ma * HW * 4 >= lim || (size_t)nb1 * HW * 4 >= lim || (size_t)nb2 * HW * 4 >= lim) return hipErrorInvalidValue;
    if (ma <= 32 && nb <= 128) e = ma <= 16 ? launch_gram<1, 2, false>(ga, grid, st) : launch_gram<2, 2, false>(ga, grid, st);
    else if (ma <= 128 && nb <= 80) e = ma <= 64 ? launch_gram<1, 5, true>(ga, grid, st) : launch_gram<2, 5, true>(ga, grid, st);
    else if (ma <= 128 && nb <= 144) e = ma <= 64 ? launch_gram<1, 9, true>(ga, grid, st) : launch_gram<2, 9, true>(ga, grid, st);   // C = 32: 4C + c_cond = 131
    if (e != hipSuccess) return e;
    return nca_launch_reduce_rows(ws, out, grid, ma * nb + ma, st, accumulate);
}
