/* ncahip.h -- C ABI of libncahip.so: the MI355X (gfx950) NCA step hot path.
 *
 * The reference (smehra34/Video-Stylization-with-NCA) is 100 % Python/PyTorch and has no FFI
 * layer; its "plugin surface" for this path is the Python class API.  This header
 * is the boundary a maintainer binds with ctypes from those classes (INTEGRATION.md shows the
 * stub).  Every entry point names the reference code it replaces (file:line in the reference).
 *
 * Conventions (all functions):
 *   - plain C types only; every pointer is a DEVICE pointer (tensor.data_ptr()), contiguous NCHW;
 *   - the caller owns every buffer, nothing is allocated, freed or retained across calls;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t; NULL = default stream), the
 *     call never synchronises (ncahip_selftest and ncahip_check_errors excepted) and is safe from several host
 *     threads on distinct streams and devices: launch state is cached per DEVICE (the device current at the call),
 *     the two process-wide switches (ncahip_cond_precision, ncahip_debug_force_generic) are test / tuning hooks;
 *   - return 0 on success, a negative NCAHIP_E* on an argument error (nothing was launched),
 *     or a positive hipError_t if the launch failed; ncahip_last_error() describes the last
 *     failure of the calling thread.
 *   - in/out buffers of one call must not alias unless stated.
 */
#ifndef NCAHIP_H
#define NCAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCAHIP_VERSION 300 /* major*10000 + minor*100 + patch : 0.3.0 */

/* argument errors */
#define NCAHIP_EINVAL (-1)   /* null pointer / non-positive size / bad enum            */
#define NCAHIP_ERANGE (-2)   /* shape outside what the kernels are instantiated for     */
#define NCAHIP_EDEVICE 100001 /* a device-side failure was recorded (sticky error word)  */

/* F.pad modes used by DyNCA perception (ConditioneDyNCA/models/dynca.py:85, default :31) */
#define NCAHIP_PAD_ZERO      0   /* 'constant'  */
#define NCAHIP_PAD_REPLICATE 1
#define NCAHIP_PAD_CIRCULAR  2
#define NCAHIP_PAD_REFLECT   3

typedef void *ncahip_stream_t;   /* hipStream_t */

int ncahip_version(void);
const char *ncahip_last_error(void);

/* Device-side failures.  The producer/consumer step kernels hand tiles over through bounded polls; a poll that expires (a
 * stalled partner wave) cannot hang the device, but the launch then worked on a stale tile.  Such a launch sets a sticky,
 * host-visible error word of its device.  The grow drivers (ncahip_cond_grow_fwd_*) refuse to enqueue with NCAHIP_EDEVICE
 * while the word is set (checked without synchronising: it reflects launches that have completed), and
 * ncahip_check_errors synchronises `stream`, returns NCAHIP_EDEVICE if the word is set and clears it when `clear` != 0.
 * Callers that need certainty for a particular result call ncahip_check_errors after it (the Python layer does so wherever
 * it synchronises anyway).  ncahip_debug_inject_error sets bits of the word (test hook for the return-code plumbing).       */
int ncahip_check_errors(ncahip_stream_t stream, int clear);
int ncahip_debug_inject_error(unsigned bits);

/* Largest shapes the fused step kernels accept (C <= max_c, fc <= max_fc, hidden <= 64).  The fp32 DyNCA *forward* entry
 * points and ncahip_dynca_nsteps_bwd_f32 additionally take 16 < C <= 32 (BASELINE configs[4]) and fc up to 1024 (one launch
 * per 128-wide slice of the hidden layer, the later ones accumulating); the single-step DyNCA backward entry points take
 * C <= 32 with fc <= max_fc; the ConditionedNCA bf16 backward covers C <= max_c, the ConditionedNCA forward fast paths (fp32 and
 * bf16 storage) C <= 20 -- the reference's default model --, the fp32 forward any-shape kernels and the fused fp32 backward
 * C <= 32.                                                                                                                  */
int ncahip_limits(int *max_c, int *max_fc, int *max_hidden);

/* Arithmetic of the UpdateNet products in ncahip_cond_step_fwd_f32 / ncahip_cond_grow_fwd_f32 (process-wide; C in {12,16},
 * hidden = 64, aligned shapes -- other shapes always compute exactly):
 *   0  exact fp32 MFMA (default; bit-compatible with an fmaf chain -- what every parity claim and bench.py's `value` use)
 *   1  "bf16x3": each fp32 operand as a bf16 pair, three bf16 MFMAs per product block, f32 accumulation; relative error
 *      ~1e-5 per step (inside the 1e-4 bar, not exact), about twice the throughput.  The backward always recomputes exactly. */
int ncahip_cond_precision(int mode);

/* Test hook (process-wide) selecting which kernel family serves the fused steps, so every variant can be checked
 * against the oracle on the same inputs: bit 0 = generic any-shape kernels instead of the aligned fast paths;
 * bit 1 = symmetric wave-private ConditionedNCA kernel instead of the default producer/consumer one;
 * bit 2 = ncahip_cond_grow_bwd_bf16 evaluates its matrix products in exact fp32 instead of on bf16 MFMA;
 * bit 3 = the ConditionedNCA backward runs its main kernel in the other of its two forms (one launch <-> front kernel +
 * matrix kernel; same results, each mode defaults to the faster one: an independent implementation to cross-check with);
 * bit 4 = the matrix kernel walks whole super-tiles on small grids as well (by default a grid with fewer super-tiles than half
 * the CUs gives each workgroup half a super-tile; with this bit its summation order equals the one-launch form's bit for bit). */
int ncahip_debug_force_generic(int on);

/* Device-side check that the MFMA operand/accumulator lane maps the kernels assume hold on
 * this GPU (exact integer data, asymmetric B).  `scratch` >= 4096 bytes of device memory;
 * returns 0 and writes 1 to ((int*)scratch)[0] when the maps hold.  Synchronises `stream`. */
int ncahip_selftest(void *scratch, ncahip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Standalone perception stencils (HBM-bound; the kernels the HBM roofline is measured on)
 * ---------------------------------------------------------------------------------------- */

/* DyNCA.perceive_torch(x, scale=0)          ConditioneDyNCA/models/dynca.py:75-100
 *   y[B,4C,H,W] = [ x | Sx*x | Sy*x | L*x ]  fixed filters :67-73, F.pad mode `pad_mode`.    */
int ncahip_dynca_perceive_f32(const float *x, float *y, int B, int C, int H, int W,
                              int pad_mode, ncahip_stream_t stream);

/* ConditionedNCA.perception_net(z)          EncoderConditioning/nca.py:99-107,177
 *   y[B,3C,H,W], y[3c+k] = wp[3c+k] (3x3, zero pad) * z[c];  wp = perception_net.weight [3C,1,3,3]. */
int ncahip_cond_perceive_f32(const float *z, const float *wp, float *y, int B, int C, int H, int W,
                             ncahip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Conditioning front ends: the fixed-filter part of the encoders that produce the step's other input
 * ---------------------------------------------------------------------------------------- */

/* ImageEncoder.forward up to `embed`         EncoderConditioning/encoder.py:37-52
 *   feat[B,3+ch,H,W] = [ sobel_x(gray) | sobel_y(gray) | laplacian(gray) | gaussian_blur(img[c]) for c < ch ],
 *   gray = mean over the ch image channels; 3x3 filters k3 [3][9] = the module's sobel_x / sobel_y / laplacian weights,
 *   5x5 blur k5 [25] = gaussian_blur.weight (frozen parameters, :12-27, :60-64); zero padding.  ch <= 8.
 *   The learned convolutions (embed, :30-34) stay with the caller (MIOpen through PyTorch: their weights train).          */
int ncahip_image_encoder_front_f32(const float *img, const float *k3, const float *k5, float *feat,
                                   int B, int ch, int H, int W, ncahip_stream_t stream);

/* EdgeExtractor.forward                      ConditioneDyNCA/models/dynca.py:204-213
 *   out[B,3,H,W] = transform([ sobel_x | sobel_y | laplacian ](img[B,1,H,W])), zero padding, transform = tanh iff
 *   apply_tanh != 0 (edge_transform == 'tanh'); k3 [3][9] = the module's three frozen filters.                           */
int ncahip_edge_extractor_f32(const float *img, const float *k3, float *out, int B, int H, int W,
                              int apply_tanh, ncahip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DyNCA fused step                          ConditioneDyNCA/models/dynca.py:117-138
 *   (ExtraChannels/models/dynca.py:113-128 is the same step with c_cond = 2 or 0)
 *   x_out = x_in + (w2 relu(w1 [perc(x_in) | cond] + b1) + b2) * floor(u + update_rate)
 *   x_in,x_out [B,C,H,W]; cond [B,c_cond,H,W] = already-extracted conditioning (EdgeExtractor
 *   :204-213 or CPE2D :226-253 output; NULL iff c_cond == 0); u [B,1,H,W] = the torch.rand draw
 *   of :131, or NULL to draw it in-kernel (Philox4x32-10 keyed (seed, step, cell)).
 *   w1 [fc, 4C+c_cond], b1 [fc], w2 [C, fc], b2 [C]  (reference layouts, 1x1 dims squeezed).
 * ---------------------------------------------------------------------------------------- */
int ncahip_dynca_step_fwd_f32(const float *x_in, float *x_out, const float *cond, const float *u,
                              const float *w1, const float *b1, const float *w2, const float *b2,
                              int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                              float update_rate, uint64_t seed, uint64_t step,
                              ncahip_stream_t stream);

/* DyNCA.forward_nsteps                      ConditioneDyNCA/models/dynca.py:168-178
 *   `states` holds `ring` slots of B*C*H*W floats; slot 0 is the input state, step t reads slot
 *   t % ring and writes slot (t+1) % ring.  ring = 2: ping-pong (inference); ring = T+1: every
 *   state is kept (what the backward pass re-reads).  u: [T,B,1,H,W] or NULL (Philox, step0+t). */
int ncahip_dynca_nsteps_fwd_f32(float *states, int ring, int T, const float *cond, const float *u,
                                const float *w1, const float *b1, const float *w2, const float *b2,
                                int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                float update_rate, uint64_t seed, uint64_t step0,
                                ncahip_stream_t stream);

/* The same step / loop with TWO-SCALE perception (DyNCA(perception_scales=[0, 1]): dynca.py:75-115 -- every video model the
 * reference ships, docs/data/video_models/{small,large}/*.json, has n_perception_scales = 2; WebGL twin docs/dynca.js:288-355):
 *     y = ( perc(x) + up2( perc( down2(x) ) ) ) / 2,   down2 / up2 = F.interpolate(bilinear, align_corners=False) by 2.
 * Per step: one coarse pass (2x2 mean + fixed filters with F.pad(mode) on the coarse grid -> pc_scratch [B,4C,H/2,W/2]) and
 * the fused step kernel, which up-samples the coarse tile from LDS on the fly.  H and W even, C <= 16, fc <= 128
 * (NCAHIP_ERANGE otherwise: the Python layer then composes stencil + torch resampling).                                   */
int ncahip_dynca_step_fwd_ms_f32(const float *x_in, float *x_out, const float *cond, const float *u,
                                 const float *w1, const float *b1, const float *w2, const float *b2,
                                 int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step, float *pc_scratch,
                                 ncahip_stream_t stream);
int ncahip_dynca_nsteps_fwd_ms_f32(float *states, int ring, int T, const float *cond, const float *u,
                                   const float *w1, const float *b1, const float *w2, const float *b2,
                                   int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                   float update_rate, uint64_t seed, uint64_t step0, float *pc_scratch,
                                   ncahip_stream_t stream);

/* Backward of ONE DyNCA step (autograd through dynca.py:117-138; dynca.py:123: no gradient into cond).
 *   In : x_t (the step's input state), the same cond / u (or seed, step) / weights, g_next = dL/dx_{t+1}.
 *   Out: g_x = dL/dx_t (data path through W2^T, relu', W1^T on MFMA, then the adjoint of "F.pad(mode) + fixed 3x3
 *        filters", plus the residual path), and the two operand pairs of the weight-gradient products, which the
 *        caller evaluates over all cells with ncahip_gram_rows_f32 (below):
 *            h_out  = relu(w1 y + b1)            [B,fc,H,W]     dW2 += (g_next*mask) h^T ,  db2 += sum(g_next*mask)
 *            dh_out = dL/d(w1 y + b1)            [B,fc,H,W]     dW1 += dh y^T            ,  db1 += sum(dh)
 *        (y = [ncahip_dynca_perceive_f32(x_t) | cond], mask = floor(u + rate)).  dy_scratch: [B,4C,H,W] floats. */
int ncahip_dynca_step_bwd_f32(const float *x_t, const float *cond, const float *u,
                              const float *w1, const float *b1, const float *w2, const float *b2,
                              int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                              float update_rate, uint64_t seed, uint64_t step,
                              const float *g_next, float *g_x, float *h_out, float *dh_out,
                              float *dy_scratch, ncahip_stream_t stream);

/* The same backward step with the layer-2 weight gradient fused in: h is kept in registers and never written; the kernel
 * accumulates dW2 = (g_next*mask) h^T and db2 = sum(g_next*mask) on MFMA (cell axis as K, operands transposed through LDS)
 * and gw2_out [C*fc + C] receives [dW2 | db2] of THIS step (accumulate = 0: overwritten, 1: added; per-workgroup partials
 * summed in fixed order).
 * dh_out / dy_scratch / g_x as above; dW1 | db1 = ncahip_gram_rows_f32(dh_out, perception, cond).                     */
size_t ncahip_dynca_step_bwd_w2_workspace(int B, int C, int H, int W, int fc);
int ncahip_dynca_step_bwd_w2_f32(const float *x_t, const float *cond, const float *u,
                                 const float *w1, const float *b1, const float *w2, const float *b2,
                                 int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step,
                                 const float *g_next, float *g_x, float *dh_out, float *dy_scratch,
                                 float *gw2_out, int accumulate, void *workspace, size_t workspace_bytes,
                                 ncahip_stream_t stream);

/* Backward of ncahip_dynca_nsteps_fwd_f32      autograd through dynca.py:168-178 (experiments.py:226,254)
 *   The whole T-step loop is enqueued on the stream with caller-owned scratch (no allocation, no host round trip).  states:
 *   the T+1 slots the forward kept with ring = T+1; cond / u (or seed, step0) / weights as in the forward.  g_final = dL/dx_T
 *   (INCLUDING any cotangent of x_T itself); g_states (nullable): [T+1 slots] cotangents of the intermediate states x_t, slots
 *   0..T-1 are added where x_t's gradient is formed (forward_nsteps' return_middle_feature).  Writes dL/dx_0 and the weight
 *   gradients in the reference layouts g_w1 [fc, 4C+c_cond], g_b1 [fc], g_w2 [C, fc], g_b2 [C] (overwritten).  C <= 32,
 *   fc <= 1024: hidden layers wider than 128 run as 128-wide slices (w2 relu(w1 y + b1) is a sum over hidden units; dL/dy is
 *   accumulated over the slices, the weight gradients of different slices are independent).  Per step: perception of x_t,
 *   per slice {fused MLP-backward kernel (dh, dL/dy, dW2|db2 on MFMA), dW1|db1 = ncahip_gram_rows_f32(dh, perception, cond)},
 *   stencil adjoint.  Deterministic (fixed-order partial sums, no float atomics).                                          */
size_t ncahip_dynca_nsteps_bwd_workspace(int B, int C, int H, int W, int fc, int c_cond);
int ncahip_dynca_nsteps_bwd_f32(const float *states, int T, const float *cond, const float *u,
                                const float *w1, const float *b1, const float *w2, const float *b2,
                                int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                float update_rate, uint64_t seed, uint64_t step0,
                                const float *g_final, const float *g_states,
                                float *g_x0, float *g_w1, float *g_b1, float *g_w2, float *g_b2,
                                void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* The same backward over a bf16 history (the [T+1 slots] ncahip_dynca_nsteps_fwd_bf16 kept with ring = T+1): storage format only, as in
 * the forward -- x_t is widened exactly into scratch and the step's backward runs in fp32; every gradient is fp32.
 * B*C*H*W % 4 == 0, states 8-byte aligned; fc <= 128 (the bf16 forward's limit).                                           */
size_t ncahip_dynca_nsteps_bwd_bf16_workspace(int B, int C, int H, int W, int fc, int c_cond);
int ncahip_dynca_nsteps_bwd_bf16(const uint16_t *states, int T, const float *cond, const float *u,
                                 const float *w1, const float *b1, const float *w2, const float *b2,
                                 int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step0,
                                 const float *g_final, const float *g_states,
                                 float *g_x0, float *g_w1, float *g_b1, float *g_w2, float *g_b2,
                                 void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* The same backward through the TWO-SCALE steps of ncahip_dynca_nsteps_fwd_ms_f32 (training with perception_scales = [0, 1]:
 * ExtraChannels/fit_video_motion.py:129-130 defaults to it).  dL/dy splits evenly over the two levels: the fine level goes
 * through the stencil adjoint as before; the coarse level through the adjoint of the bilinear x2 up-sampling, the stencil
 * adjoint on the COARSE grid (pad mode resolved there) and the adjoint of the 2x2 mean.  Even H and W, C <= 16, fc <= 128.     */
size_t ncahip_dynca_nsteps_bwd_ms_workspace(int B, int C, int H, int W, int fc, int c_cond);
int ncahip_dynca_nsteps_bwd_ms_f32(const float *states, int T, const float *cond, const float *u,
                                   const float *w1, const float *b1, const float *w2, const float *b2,
                                   int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                   float update_rate, uint64_t seed, uint64_t step0,
                                   const float *g_final, const float *g_states,
                                   float *g_x0, float *g_w1, float *g_b1, float *g_w2, float *g_b2,
                                   void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* Weight-gradient products of the DyNCA backward with the cell axis as K (replaces the library GEMMs over transposed
 * copies that autograd through dynca.py:127-128 amounts to):
 *     out[i*nb + j] = sum over all B*HW cells of a[., i, .] * b[., j, .]     i < ma, j < nb = nb1 + nb2
 *     out[ma*nb + i] = sum over all cells of a[., i, .]                       (the bias gradient of the same layer)
 * a [B, ma, HW]; the b rows come from two tensors, b1 [B, nb1, HW] then b2 [B, nb2, HW] (b2 may be NULL with nb2 = 0):
 * dW1 | db1 = gram(dh_out, perception, cond), dW2 | db2 = gram(g_next * mask, h_out).  Exact fp32 MFMA, per-workgroup
 * partials summed in a fixed order (deterministic).  Shapes: ma <= 128 with nb <= 80, or ma <= 32 with nb <= 128
 * -- since 0.2.0: ma <= 128 with nb <= 144 --
 * (NCAHIP_ERANGE otherwise).  out holds ma*nb + ma floats; accumulate = 0 overwrites it, 1 adds to it (the sum over the
 * steps of a backward pass without a separate add per step).                                                          */
size_t ncahip_gram_rows_workspace(int ma, int nb, int B, int HW);
int ncahip_gram_rows_f32(const float *a, int ma, const float *b1, int nb1, const float *b2, int nb2, int B, int HW,
                         float *out, int accumulate, void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * ConditionedNCA fused step                 EncoderConditioning/nca.py:181-195
 *
 * The step's post-life mask (:191-193) needs the NEW alpha on a 3x3 neighbourhood, i.e. a
 * grid-wide exchange after the update.  The kernel therefore emits the state in PENDING form:
 *     x_out   = x + rand_mask * update(x)                 (:189, before :191-194)
 *     pre_out = alive(x)                                  (:185, uint8 [B,H,W])
 * and the NEXT step (or ncahip_cond_finalize_f32) resolves
 *     x'' = clamp(x_out * (pre_out & alive(x_out)), lo, hi)   (:191-194)
 * while it loads its tile (alpha halo of 3 cells).  Pass pre_in = NULL when x_in is a true
 * state (first step), or the previous call's pre_out when x_in is pending.
 *
 *   goal  [B,goal_ch,H,W] = encoder(goal) UNPADDED (nca.py:198); it is added to the LAST
 *         goal_ch channels (:199-203 pads zeros in front).  goal_ch may equal C.
 *   u     [B,1,H,W] rand_like draw (:172) or NULL (in-kernel Philox); mask = u < fire_rate.
 *   wp [3C,9]; w1 [hidden,3C], b1 [hidden]; w2 [hidden,hidden], b2 [hidden]; w3 [C,hidden].
 *   alive_ch < 0  <=>  use_living_channel=False (:153-154): every cell alive.
 * ---------------------------------------------------------------------------------------- */
int ncahip_cond_step_fwd_f32(const float *x_in, const uint8_t *pre_in, float *x_out, uint8_t *pre_out,
                             const float *goal, int goal_ch, const float *u,
                             const float *wp, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *w3,
                             int B, int C, int H, int W, int hidden,
                             int alive_ch, float alive_thr, float fire_rate,
                             float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step,
                             ncahip_stream_t stream);

/* Resolve a pending state: x_out = clamp(x_pend * (pre & alive(x_pend)), lo, hi)  nca.py:191-194.
 * x_out may alias x_pend only if alive_ch < 0.                                              */
int ncahip_cond_finalize_f32(const float *x_pend, const uint8_t *pre, float *x_out,
                             int B, int C, int H, int W, int alive_ch, float alive_thr,
                             float clamp_lo, float clamp_hi, ncahip_stream_t stream);

/* ---- bf16 state storage, DyNCA ---------------------------------------------------------------
 * DyNCA.forward / forward_nsteps (dynca.py:117-138, :168-178) with the state stored as bf16 (uint16_t bit patterns);
 * cond, uniforms and weights stay fp32.  The step computes exactly as the _f32 entry points on the widened state
 * (exact fp32 MFMA) and rounds x + dx*mask to bf16 (RNE) on store: storage format only, same kernel.               */
int ncahip_dynca_step_fwd_bf16(const uint16_t *x_in, uint16_t *x_out, const float *cond, const float *u,
                               const float *w1, const float *b1, const float *w2, const float *b2,
                               int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                               float update_rate, uint64_t seed, uint64_t step, ncahip_stream_t stream);
int ncahip_dynca_nsteps_fwd_bf16(uint16_t *states, int ring, int T, const float *cond, const float *u,
                                 const float *w1, const float *b1, const float *w2, const float *b2,
                                 int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step0, ncahip_stream_t stream);

/* ---- bf16 state storage ---------------------------------------------------------------------
 * The same step / finalize / grow loop with the state and the goal encoding stored as bf16 (uint16_t bit patterns,
 * [B,C,H,W] contiguous) and the UpdateNet on bf16 MFMA; weights, biases and explicit uniforms stay fp32.  Rounding
 * points (restated by oracle.cond_step_bf16): perception in f32 from the widened state; the perception vector, both
 * hidden activations and the weights are rounded to bf16 (RNE) as matrix operands, accumulation in f32; x' = x + mask*out
 * in f32, rounded to bf16 on store.  Replaces the same reference code as the _f32 entry points (nca.py:181-195, :207-208)
 * for callers that keep the pool in bf16 (BASELINE configs[2]).  Needs W % 4 == 0, 8-byte aligned tensors and C <= 20 -- the
 * reference's default model (nca.py:62-94) included; ncahip_cond_grow_bwd_bf16 takes the same range -- (NCAHIP_ERANGE otherwise;
 * there is no any-shape bf16 kernel).                                                                                 */
int ncahip_cond_step_fwd_bf16(const uint16_t *x_in, const uint8_t *pre_in, uint16_t *x_out, uint8_t *pre_out,
                              const uint16_t *goal, int goal_ch, const float *u,
                              const float *wp, const float *w1, const float *b1,
                              const float *w2, const float *b2, const float *w3,
                              int B, int C, int H, int W, int hidden,
                              int alive_ch, float alive_thr, float fire_rate,
                              float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step,
                              ncahip_stream_t stream);
int ncahip_cond_finalize_bf16(const uint16_t *x_pend, const uint8_t *pre, uint16_t *x_out,
                              int B, int C, int H, int W, int alive_ch, float alive_thr,
                              float clamp_lo, float clamp_hi, ncahip_stream_t stream);
int ncahip_cond_grow_fwd_bf16(uint16_t *states, uint8_t *pre, int ring, int T, uint16_t *x_final,
                              const uint16_t *goal, int goal_ch, const float *u,
                              const float *wp, const float *w1, const float *b1,
                              const float *w2, const float *b2, const float *w3,
                              int B, int C, int H, int W, int hidden,
                              int alive_ch, float alive_thr, float fire_rate,
                              float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step0,
                              ncahip_stream_t stream);

/* ConditionedNCA.alive(x)                   EncoderConditioning/nca.py:152-163 -> uint8 [B,H,W] */
int ncahip_cond_alive_u8(const float *x, uint8_t *out, int B, int C, int H, int W,
                         int alive_ch, float alive_thr, ncahip_stream_t stream);

/* ConditionedNCA.grow's loop                EncoderConditioning/nca.py:207-208
 *   T fused steps + one finalize.  states: `ring` slots of B*C*H*W floats, slot 0 = input (true
 *   state); slot k (k>=1) receives the PENDING output of step k-1; pre: `ring` slots of B*H*W
 *   bytes, slot k pairs with states slot k (slot 0 unused).  x_final receives grow()'s result.
 *   ring = 2 ping-pong, ring = T+1 keeps every pending state for the backward pass.          */
int ncahip_cond_grow_fwd_f32(float *states, uint8_t *pre, int ring, int T, float *x_final,
                             const float *goal, int goal_ch, const float *u,
                             const float *wp, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *w3,
                             int B, int C, int H, int W, int hidden,
                             int alive_ch, float alive_thr, float fire_rate,
                             float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step0,
                             ncahip_stream_t stream);

/* Backward of ncahip_cond_grow_fwd_f32      autograd through nca.py:207-208 (conditioned_trainer.py:125-132)
 *   Recomputation: needs only what the forward kept with ring = T+1 -- states [T+1 slots] (slot 0 the
 *   input, slot k the pending output of step k-1) and pre [T+1 slots] -- plus the same goal / u (or seed,
 *   step0) / weights.  g_final = dL/d x_final.  Writes dL/dx0 [B,C,H,W], dL/dgoal [B,goal_ch,H,W] (summed over
 *   the T steps, nca.py:207-208 reuse the same encoding), and the weight gradients in the reference layouts
 *   (g_wp [3C,9], g_w1 [hidden,3C], g_b1, g_w2 [hidden,hidden], g_b2, g_w3 [C,hidden]); masks carry no
 *   gradient; clamp passes gradient on the closed interval (torch.clamp).  Deterministic (no float atomics).
 *   Requires W % 4 == 0 and 16-byte aligned buffers.  `workspace`: ncahip_cond_grow_bwd_workspace() bytes.
 *   T = 1 is the backward of one ncahip_cond_step_fwd_f32 + ncahip_cond_finalize_f32 pair (slot 0 = x_t, slot 1 = the
 *   pending x'_t the step wrote): there is no separate per-step entry point.                                       */
size_t ncahip_cond_grow_bwd_workspace(int B, int C, int H, int W, int hidden);
int ncahip_cond_grow_bwd_f32(const float *states, const uint8_t *pre, int T,
                             const float *goal, int goal_ch, const float *u,
                             const float *wp, const float *w1, const float *b1,
                             const float *w2, const float *b2, const float *w3,
                             int B, int C, int H, int W, int hidden,
                             int alive_ch, float alive_thr, float fire_rate,
                             float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step0,
                             const float *g_final, float *g_x0, float *g_goal,
                             float *g_wp, float *g_w1, float *g_b1, float *g_w2, float *g_b2, float *g_w3,
                             void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* The same backward over a bf16 history: states [T+1 slots] and goal as stored by ncahip_cond_grow_fwd_bf16 with ring = T+1
 * (BASELINE configs[2]: a bf16 pool halves the saved-for-backward set).  Mixed precision as in the forward: the stored values
 * are widened exactly; perception, masks, gating and every stored gradient (g_final, outputs, scratch) are fp32; the matrix
 * products -- the recomputed UpdateNet with the forward's rounding points, the data path through W3^T / W2^T / W1^T, and
 * the three weight-gradient products over the cell axis -- run on v_mfma_f32_16x16x16_bf16 with fp32 accumulation (92 MFMAs
 * of 8 cycles per 16 cells instead of 368 exact-f32 ones of 32).  Straight-through with respect to the storage rounding;
 * bound against the exact-fp32 products of the same history and against the fp32 gradients in tests/.
 * W % 4 == 0, states / goal 8-byte aligned.                                                                             */
int ncahip_cond_grow_bwd_bf16(const uint16_t *states, const uint8_t *pre, int T,
                              const uint16_t *goal, int goal_ch, const float *u,
                              const float *wp, const float *w1, const float *b1,
                              const float *w2, const float *b2, const float *w3,
                              int B, int C, int H, int W, int hidden,
                              int alive_ch, float alive_thr, float fire_rate,
                              float clamp_lo, float clamp_hi, uint64_t seed, uint64_t step0,
                              const float *g_final, float *g_x0, float *g_goal,
                              float *g_wp, float *g_w1, float *g_b1, float *g_w2, float *g_b2, float *g_w3,
                              void *workspace, size_t workspace_bytes, ncahip_stream_t stream);

/* ---- T DyNCA steps in ONE launch (B = 1 video inference) --------------------------------------------------------------
 * ConditioneDyNCA/utils/misc/video_utils.py:50-82 (forward_nsteps(h, step_n, cond_img=frame) per frame, step_n = 8 by default;
 * WebGL twin docs/dynca.js:1057-1132).  x_out = the state after T steps from x_in (x_in != x_out), bit for bit what
 * ncahip_dynca_nsteps_fwd_f32 computes, but every 16 x 16 tile is owned by one workgroup for all T steps: weights and
 * conditioning staged once, the tile's state kept in LDS, only the one-cell halo exchanged per step -- as (value, tag) pairs in
 * `workspace`, tag = epoch * 4096 + step, so neighbours need no counters and the call is ONE launch (no copy, no memset).
 * Workspace contract: ncahip_dynca_nsteps_persist_workspace bytes, 256-byte aligned, ZEROED ONCE by the caller when allocated;
 * every call on it passes a strictly larger `epoch` than the one before (1, 2, ... < 2^20; zero it again and restart at 1 when
 * that runs out) -- stale pairs of earlier launches then never match.  Covered: C <= 16, fc <= 128, H % 16 == 0, W % 16 == 0,
 * T < 4096, and every tile's workgroup resident at once (occupancy x CUs >= B * H/16 * W/16: 1 x 256 x 256 is exactly one per CU
 * of an MI355X); NCAHIP_ERANGE otherwise -- the caller then runs ncahip_dynca_nsteps_fwd_f32.  The workspace query returns 0 for
 * shapes that are never covered.  A neighbour that never becomes resident (another process holding CUs) makes a bounded poll
 * expire: the launch still drains, bit 1 of the sticky device error word is set and ncahip_check_errors / the next driver call
 * report NCAHIP_EDEVICE (x_out is then not valid). */
size_t ncahip_dynca_nsteps_persist_workspace(int B, int C, int H, int W, int fc, int c_cond);
int ncahip_dynca_nsteps_fwd_persist_f32(const float *x_in, float *x_out, int T, const float *cond, const float *u,
                                        const float *w1, const float *b1, const float *w2, const float *b2,
                                        int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                        float update_rate, uint64_t seed, uint64_t step0,
                                        void *workspace, size_t workspace_bytes, unsigned epoch, ncahip_stream_t stream);
/* The same for perception_scales = [0, 1] (every shipped video model; dynca.py:75-115): bit for bit what
 * ncahip_dynca_nsteps_fwd_ms_f32 computes.  Tiles additionally exchange the 2 x 2 means of their coarse cells within 2 of the
 * border; same workspace, epoch and coverage rules. */
int ncahip_dynca_nsteps_fwd_persist_ms_f32(const float *x_in, float *x_out, int T, const float *cond, const float *u,
                                           const float *w1, const float *b1, const float *w2, const float *b2,
                                           int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                           float update_rate, uint64_t seed, uint64_t step0,
                                           void *workspace, size_t workspace_bytes, unsigned epoch, ncahip_stream_t stream);
/* Test hook: the next persistent launches leave out their last n tiles -- what a workgroup that never becomes resident looks
 * like to its neighbours (their bounded polls expire; the launch drains; NCAHIP_EDEVICE).  0 restores normal launches. */
int ncahip_debug_persist_drop_tiles(int n);

/* ---- fire masks as bits --------------------------------------------------------------------------------------------
 * Every entry point above that takes `u` (the per-step uniform draws of nca.py:172 / dynca.py:131) also accepts the fire
 * masks ALREADY EVALUATED and bit-packed: pass a non-NULL `u` that points at uint32_t words together with
 * seed == NCAHIP_SEED_U_IS_BITS (seed is otherwise unused when u != NULL).  Layout: per step ceil(B*H*W / 32) words, cell
 * i = (b*H + y)*W + x  <->  bit (i & 31) of word (i >> 5); the multi-step drivers advance by that many words per step.  A
 * set bit = the cell updates (ConditionedNCA: clamp(u,0,1) < fire_rate; DyNCA: floor(u + update_rate) = 1, which needs
 * 0 <= update_rate < 1).  32x smaller than the float draws: what the drop-in classes keep for the backward pass (BASELINE
 * configs[2]: 805 MB of uniforms -> 25 MB).  Needs B*H*W < 2^32.  ncahip_pack_fire_mask_u32 evaluates T steps of float
 * draws u [T][B*H*W] into bits [T][ceil(B*H*W/32)] with exactly the kernels' predicates: mode 0 = ConditionedNCA
 * (nca.py:171-174), mode 1 = DyNCA (dynca.py:131). */
#define NCAHIP_SEED_U_IS_BITS 0x5354494255ull
int ncahip_pack_fire_mask_u32(const float *u, uint32_t *bits, int T, int B, int H, int W, float rate, int mode,
                              ncahip_stream_t stream);

/* The [B,1,H,W] uniforms the kernels draw for (seed, step) when u == NULL (for tests/tools). */
int ncahip_philox_uniform_f32(float *u, int B, int H, int W, uint64_t seed, uint64_t step,
                              ncahip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NCAHIP_H */
