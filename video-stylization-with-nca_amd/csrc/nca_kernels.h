// nca_kernels.h -- internal launch interface between the C ABI (nca_capi.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// ---- per-DEVICE launch state (the C ABI is legal from any host thread on any device: caches are keyed by the device that is
// current at the call, never by the calling thread) -----------------------------------------------------------------------
constexpr int kNcaMaxDevices = 64;
inline int nca_device_index() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kNcaMaxDevices) d = 0;
    return d;
}
// CU count of the current device (cached per device)
inline int nca_cu_count() {
    static std::atomic<int> cus[kNcaMaxDevices];
    const int d = nca_device_index();
    int v = cus[d].load(std::memory_order_relaxed);
    if (v == 0) {
        int q = 0;
        v = 256;
        if (hipDeviceGetAttribute(&q, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess && q > 0) v = q;
        cus[d].store(v, std::memory_order_relaxed);
    }
    return v;
}
// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel instantiation, device): one static instance per launcher template
struct NcaLdsAttr {
    std::atomic<bool> done[kNcaMaxDevices];
    hipError_t ensure(const void* kern, size_t bytes) {
        const int d = nca_device_index();
        if (done[d].load(std::memory_order_acquire)) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) done[d].store(true, std::memory_order_release);
        return e;
    }
};
// Sticky device-side error word of the current device (host-mapped, so the host reads it without synchronising): bit 0 = a
// producer/consumer hand-off poll expired (nca_cond_pc.hip).  nullptr when the allocation failed (errors are then not recorded).
unsigned* nca_error_word_device();        // device-visible pointer for kernel arguments
unsigned nca_error_word_read(bool clear); // host view of the current device's word

struct NcaDyncaArgs {
    const float* x_in;
    float* x_out;
    const float* cond;  // [B,c_cond,H,W] or null
    const float* u;     // [B,1,H,W] or null (Philox)
    const float *w1, *b1, *w2, *b2;
    int B, C, H, W, fc, c_cond, pad_mode;
    float rate;
    uint64_t seed, step;
    // backward variant only (nca_launch_dynca_step_bwd): dL/dx_{t+1} in, recomputed hidden layer / its gradient /
    // dL/dperception out, dL/dx_t out (stencil-adjoint kernel)
    const float* g_next;   // [B,C,H,W]
    float* hbuf;           // relu(w1 y + b1)                [B,fc,H,W]
    float* dhbuf;          // dL/d(pre-activation)           [B,fc,H,W]
    float* dybuf;          // dL/dy, first 4C rows           [B,4C,H,W]
    float* g_out;          // dL/dx_t                        [B,C,H,W]
    // fc > 128 runs as several launches over 128-wide slices of the hidden layer (nca_launch_dynca_step_fwd):
    int w2_ld;             // row stride of w2 (0: = fc); later slices run the accumulating instantiation (x_out += mask * slice)
    float* gw2_ws;         // backward, fused dW2: per-workgroup partials [grid][C*fc + C] (dW2 | db2); hbuf is then not written
    const float* g_extra;  // backward stencil: optional cotangent of x_t itself, added to g_out (forward_nsteps' middle features)
    const float* pc;       // two-scale perception (perception_scales = [0, 1]): coarse-level perception [B,4C,H/2,W/2], or null
    const float* coarse_add;  // backward stencil, two-scale: dL/dx of the coarse level [B,C,H/2,W/2]; 0.25 * parent is added to g_out
    int dy_half;              // backward stencil, two-scale: the fine level carries 0.5 * dL/dy
    float* ybuf;              // backward MLP kernel: if set, the recomputed perception y (two-scale: the combined one) is written here,
                              // [B,4C,H,W] in perceive_torch's row order -- the B rows of the layer-1 weight-gradient product
    int u_bits;               // u points at bit-packed fire masks (uint32 words, cell i -> bit i & 31 of word i >> 5; B*H*W < 2^32)
};

// T DyNCA steps in one launch (nca_dynca_persist.hip): one workgroup per 16 x 16 tile for all steps, neighbour counters in `flags`
struct NcaDyncaPersistArgs {
    const float* x_in;   // [B,C,H,W] input state (read at step 0 only)
    float* x_out;        // [B,C,H,W] state after T steps (written at the last step only; may alias x_in? no: tiles read neighbours' x_in cells)
    int T;               // 1 <= T < 4096
    const float* cond;   // [B,c_cond,H,W] or null
    const float* u;      // [T][B*H*W] uniforms / [T][ceil(B*H*W/32)] mask words (u_bits) / null: Philox (seed, step0 + t)
    const float *w1, *b1, *w2, *b2;
    int B, C, H, W, fc, c_cond, pad_mode;
    float rate;
    uint64_t seed, step0;
    int* flags;          // [0] = abort word: holds the epoch of a launch that gave up
    unsigned epoch;            // strictly increasing per workspace (1 .. 2^20 - 1): tags of the exchanged pairs are epoch * 4096 + step
    unsigned long long* xch;   // ring exchange: [2 parities][tile][C][60 ring cells] (value, tag) pairs; zero once, never cleared between launches
    size_t xch_words;          // pairs per parity
    unsigned* err;       // sticky error word (bit 1: a neighbour poll expired)
    int u_bits;
    int dbg;             // diagnostic knobs (NCAHIP_PERSIST_DBG, timing experiments only -- results are then NOT valid): bit 0 no neighbour
                         // polls / halo loads, bit 1 no state stores, bit 2 no ring phase, bit 3 no mask refill, bit 4 no MFMA chains
};
void nca_set_persist_drop_tiles(int n);   // test hook: the persistent launch leaves out its last n tiles (their neighbours' polls expire)
bool nca_dynca_persist_shape_ok(int B, int C, int H, int W, int fc, int c_cond);
int nca_dynca_persist_tiles(int B, int H, int W);
// query_only: only decide whether every workgroup can be co-resident on the current device (*fits)
hipError_t nca_launch_dynca_persist(const NcaDyncaPersistArgs& a, hipStream_t st, bool query_only, bool* fits);
// two-scale perception (perception_scales = [0, 1]): exchanges 108 pairs per channel and tile (60 fine ring cells + 48 coarse means)
hipError_t nca_launch_dynca_persist_ms(const NcaDyncaPersistArgs& a, hipStream_t st, bool query_only, bool* fits);

struct NcaCondArgs {
    const float* x_in;
    const uint8_t* pre_in;  // null: x_in is a true state; else x_in is pending with this pre mask
    float* x_out;
    uint8_t* pre_out;
    const float* goal;  // [B,goal_ch,H,W]
    const float* u;
    const float *wp, *w1, *b1, *w2, *b2, *w3;
    int B, C, H, W, hidden, goal_ch, alive_ch;
    float thr, fire_rate, lo, hi;
    uint64_t seed, step;
    unsigned long long* dbg;  // diagnostic builds (-DNCA_STAMPS) only: per-wave phase time stamps
    unsigned* err;            // sticky error word (nca_error_word_device()), or null
    int u_bits;               // u points at bit-packed fire masks (uint32 words, cell i -> bit i & 31 of word i >> 5; B*H*W < 2^32)
};

// One backward step of the ConditionedNCA grow loop (nca_cond_bwd.hip).  f describes forward step t exactly as
// it was launched (x_in = states[t], pre_in = pre[t] or null for t == 0, goal, u / seed+step, weights).
struct NcaCondBwdArgs {
    NcaCondArgs f;
    const float* x_next;    // states[t+1] = pending x'_t                      [B,C,H,W]
    const uint8_t* pre_t;   // pre[t+1]    = alive(s_t)                        [B,H,W]
    const float* g_next;    // dL/d s_{t+1}                                    [B,C,H,W]
    float* g_out;           // dL/d s_t  (written by the stencil kernel)        [B,C,H,W]
    float* gx;              // scratch: direct-path gradient dL/dx'_t           [B,C,H,W]
    float* dP;              // scratch: dL/d perception                         [B,3C,H,W]
    float* zbuf;            // scratch: z_t = s_t + goal*pre_t                  [B,C,H,W]
    float* dgoal;           // accumulated over steps                           [B,goal_ch,H,W]
    float* slabs;           // per-workgroup weight-gradient partials, accumulated   [nslab, nca_cond_bwd_slab_floats] (tile-major: SlabTM)
    float* wp_partials;     // per-block perception-weight partials             [nblk, 27] accumulated
    int nslab, nblk;
    int srows;              // rows per strip of the stencil-adjoint kernel (set by its launcher: nca_cond_bwd_srows)
    float* opimg;           // matrix kernel: its LDS operand image kept in memory across the steps of one backward pass (workspace), see opmode
    int opmode;             // 0 = every launch gathers the image from the weight tensors; 1 = ONE workgroup builds it and writes it to opimg
                            // (no tile work); 2 = the launch copies it from opimg (16-byte loads instead of ~100 scattered 4-byte gathers per thread)
    int msplit;             // matrix kernel: 1 = a workgroup takes ONE pass (half the rows) of a super-tile per walk item (small grids: twice the workgroups)
    void* pscr;             // front/matrix form: perception vectors in MFMA-operand order  (nca_cond_bwd_fm_pscr_bytes)
    void* doscr;            // front/matrix form: dL/dx'_t * fire mask, [row tile][channel][cell] (nca_cond_bwd_fm_doscr_bytes)
};
int nca_cond_bwd_slab_floats(int C, int hidden);
hipError_t nca_launch_cond_bwd_unpermute(const float* red, int C, int hidden, bool bf16_history, float* g_w1, float* g_w2, float* g_w3,
                                         float* g_b1, float* g_b2, hipStream_t st);
int nca_cond_bwd_nslab();
int nca_cond_bwd_nblk(int B, int C, int H, int W);
void nca_set_bwd_fm_nosplit(bool on);
bool nca_cond_bwd_is_fm(const NcaCondBwdArgs& ba, bool bf16);   // the form nca_launch_cond_step_bwd will run for these arguments
constexpr size_t kNcaCondBwdOpimgBytes = 128 * 1024;            // upper bound of the matrix kernel's operand image (CP = 32: 100 KB)
hipError_t nca_launch_cond_step_bwd(const NcaCondBwdArgs& a, hipStream_t st, bool bf16 = false);   // bf16: f.x_in / x_next / f.goal hold bf16
// kernel A as two launches (nca_cond_bwd_fm.hip); mode 0 = f32 history, 1 = bf16 history / exact-f32 products, 2 = bf16 MFMA
hipError_t nca_launch_cond_step_bwd_fm(const NcaCondBwdArgs& a, hipStream_t st, int mode);
size_t nca_cond_bwd_fm_pscr_bytes(int B, int C, int H, int W);
size_t nca_cond_bwd_fm_doscr_bytes(int B, int C, int H, int W);
void nca_set_bwd_variant(int v);   // kernel A: 0 = the faster form per mode (bf16 MFMA: front + matrix kernels; fp32 products: one launch), 1 = one launch always, 2 = front + matrix always, 3 = the slower form per mode (cross-check)
hipError_t nca_launch_reduce_rows(const float* src, float* dst, int n, int m, hipStream_t st, bool accumulate = false);   // dst (+)= column sums
// nca_gram.hip: out[ma*nb + ma] = [sum_n a[i][n] * b[j][n] | sum_n a[i][n]] over all B*HW cells; b rows from two tensors
int nca_dynca_bwd_grid(int B, int H, int W);   // upper bound over C (workspace sizing)
int nca_dynca_bwd_grid_c(int B, int C, int H, int W);   // the grid the backward kernel actually runs with = number of slabs it writes   // workgroups of the DyNCA backward kernel (= partial slabs of its fused dW2)
int nca_gram_grid(int B, int HW);
hipError_t nca_launch_gram_rows(const float* a, int ma, const float* b1, int nb1, const float* b2, int nb2, int B, int HW,
                                float* out, float* ws, hipStream_t st, bool accumulate = false);
hipError_t nca_launch_reduce_wp(const float* part, float* dst, int B, int C, int H, int W, hipStream_t st);

// fused steps (nca_step_fwd.hip); hipErrorInvalidValue when no instantiation covers the shape
hipError_t nca_launch_dynca_step_fwd(const NcaDyncaArgs& a, hipStream_t st);
hipError_t nca_launch_cond_step_fwd(const NcaCondArgs& a, hipStream_t st);
hipError_t nca_launch_dynca_step_bwd(const NcaDyncaArgs& a, hipStream_t st);
hipError_t nca_launch_dynca_step_bwd_mlp(const NcaDyncaArgs& a, hipStream_t st, bool acc);      // MLP part only (hidden-layer slices)
hipError_t nca_launch_dynca_step_bwd_stencil(const NcaDyncaArgs& a, hipStream_t st);            // stencil adjoint + residual
hipError_t nca_launch_dynca_step_fwd_bf16(const NcaDyncaArgs& a, hipStream_t st);   // x_in / x_out hold bf16
// wave-private-tile variant (nca_cond_wave.hip); needs W % 4 == 0 and 16-byte aligned x_in / goal
hipError_t nca_launch_cond_step_fwd_wave(const NcaCondArgs& a, hipStream_t st);
// producer/consumer wave-specialised variant (nca_cond_pc.hip); same preconditions
hipError_t nca_launch_cond_step_fwd_pc(const NcaCondArgs& a, hipStream_t st);
// bf16 state storage + bf16 MFMA (nca_cond_bf16.hip): x_in / x_out / goal point at bf16 data
hipError_t nca_launch_cond_step_fwd_bf16(const NcaCondArgs& a, hipStream_t st);
hipError_t nca_launch_cond_finalize_bf16(const uint16_t* x, const uint8_t* pre, uint16_t* out, int B, int C, int H, int W,
                                         int alive_ch, float thr, float lo, float hi, hipStream_t st);
void nca_set_cond_precision(int mode);   // 0 = exact-f32 MFMA (default), 1 = bf16x3 emulation in the producer/consumer kernel
void nca_set_bwd_bf16_exact(bool on);  // test hook: the bf16-history backward recomputes in exact f32 instead of on bf16 MFMA
void nca_set_cond_variant(int v);  // 0 = producer/consumer (default), 1 = symmetric wave-private

// diagnostic build hook (-DNCA_STAMPS): buffer that receives s_memtime stamps, [wave][tile][8]
void nca_debug_set_stamp_buffer(unsigned long long* p);

// test hook: route every fused step through the generic (any-shape) kernels
void nca_set_force_generic(bool on);

// stencils and small kernels (nca_stencil.hip)
hipError_t nca_launch_dynca_perceive(const float* x, float* y, int B, int C, int H, int W, int pad, hipStream_t st);
hipError_t nca_launch_image_encoder_front(const float* img, const float* k3, const float* k5, float* feat, int B, int ch, int H, int W,
                                          hipStream_t st);
hipError_t nca_launch_edge_extractor(const float* img, const float* k3, float* out, int B, int H, int W, int do_tanh, hipStream_t st);
hipError_t nca_launch_widen_bf16(const uint16_t* in, float* out, size_t n, hipStream_t st);
hipError_t nca_launch_dynca_ms_combine(float* y, const float* pc, int B, int C, int H, int W, hipStream_t st);   // y <- (y + up2(pc)) / 2
hipError_t nca_launch_dynca_ms_upT(const float* dy, float* dpc, int B, int C, int H, int W, hipStream_t st);    // dpc = 0.5 * up2^T(dy)
int nca_dynca_bwd_ms_grid(int B, int H, int W);
hipError_t nca_launch_dynca_coarse_perceive(const float* x, float* pc, int B, int C, int H, int W, int pad, hipStream_t st);
hipError_t nca_launch_cond_perceive(const float* z, const float* wp, float* y, int B, int C, int H, int W, hipStream_t st);
hipError_t nca_launch_cond_finalize(const float* x, const uint8_t* pre, float* out, int B, int C, int H, int W,
                                    int alive_ch, float thr, float lo, float hi, hipStream_t st);
hipError_t nca_launch_cond_alive(const float* x, uint8_t* out, int B, int C, int H, int W, int alive_ch, float thr,
                                 hipStream_t st);
hipError_t nca_launch_philox_uniform(float* u, int B, int H, int W, uint64_t seed, uint64_t step, hipStream_t st);
// fire masks of T steps, bit-packed: mode 0 = clamp(u,0,1) < rate (nca.py:171-174), 1 = floorf(u + rate) >= 1 (dynca.py:131)
hipError_t nca_launch_pack_fire_mask(const float* u, uint32_t* bits, int T, size_t cells, float rate, int mode, hipStream_t st);
hipError_t nca_launch_selftest(int* result, hipStream_t st);
