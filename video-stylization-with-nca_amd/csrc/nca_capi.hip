// nca_capi.hip -- extern "C" entry points of libncahip.so (declared in include/ncahip.h).
// Validates arguments on the host (shapes the kernels and their grids assume), then enqueues.
#include "../../include/ncahip.h"

#include <cstdarg>
#include <cstdio>

#include "nca_kernels.h"

namespace {

constexpr int kMaxC = 16, kMaxCCondFwd = 32, kMaxCCondFwdBf16 = 20, kMaxCDynca = 32, kMaxFc = 128, kMaxFcFwd = 1024, kMaxHidden = 64, kMaxCond = 4;   // DyNCA forward: C <= 32 (configs[4]), fc in 128-wide slices

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_result(hipError_t e, const char* what) {
    if (e == hipSuccess) return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

// `u` holds bit-packed fire masks instead of uniforms (include/ncahip.h: NCAHIP_SEED_U_IS_BITS)
bool u_is_bits(const void* u, uint64_t seed) { return u != nullptr && seed == NCAHIP_SEED_U_IS_BITS; }
// u of step t: floats are [T][B*H*W]; bits are [T][ceil(B*H*W / 32)] words
const float* u_at(const float* u, bool bits, int t, size_t cells) {
    if (!u) return nullptr;
    return bits ? reinterpret_cast<const float*>(reinterpret_cast<const uint32_t*>(u) + (size_t)t * ((cells + 31) / 32)) : u + (size_t)t * cells;
}
int check_bits(bool bits, int B, int H, int W, float rate, bool dynca) {
    if (!bits) return 0;
    if ((size_t)B * H * W >= (((size_t)1 << 32) - 64)) return fail(NCAHIP_ERANGE, "bit-packed fire masks: B*H*W must stay below 2^32");
    if (dynca && !(rate >= 0.0f && rate < 1.0f)) return fail(NCAHIP_ERANGE, "bit-packed fire masks (DyNCA): 0 <= update_rate < 1 required (floor(u + rate) must be 0 or 1)");
    return 0;
}

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

bool dims_ok(int B, int C, int H, int W) {
    return B > 0 && C > 0 && H > 0 && W > 0 && (size_t)B * C * H * W < ((size_t)1 << 40);
}

int check_dynca(const void* x_in, const void* x_out, const void* cond, const void* w1, const void* b1, const void* w2,
                const void* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode, int max_fc = kMaxFc) {
    if (!x_in || !x_out || !w1 || !b1 || !w2 || !b2) return fail(NCAHIP_EINVAL, "dynca step: null pointer");
    if (!dims_ok(B, C, H, W) || fc <= 0 || c_cond < 0) return fail(NCAHIP_EINVAL, "dynca step: bad size");
    if ((c_cond > 0) != (cond != nullptr)) return fail(NCAHIP_EINVAL, "dynca step: cond pointer / c_cond mismatch");
    if (pad_mode < 0 || pad_mode > 3) return fail(NCAHIP_EINVAL, "dynca step: bad pad_mode %d", pad_mode);
    if (pad_mode == NCAHIP_PAD_REFLECT && (H < 2 || W < 2)) return fail(NCAHIP_EINVAL, "reflect pad needs H,W >= 2");
    if (C > kMaxCDynca || fc > max_fc || c_cond > kMaxCond)
        return fail(NCAHIP_ERANGE, "dynca step: C=%d fc=%d c_cond=%d exceeds (%d,%d,%d)", C, fc, c_cond, kMaxCDynca, max_fc,
                    kMaxCond);
    if (x_in == x_out) return fail(NCAHIP_EINVAL, "dynca step: x_in and x_out must not alias (halo reads)");
    return 0;
}

int check_cond(const void* x_in, const void* x_out, const void* pre_out, const void* goal, const void* wp,
               const void* w1, const void* b1, const void* w2, const void* b2, const void* w3, int B, int C, int H,
               int W, int hidden, int goal_ch, int alive_ch, int max_c = kMaxC) {
    if (!x_in || !x_out || !pre_out || !wp || !w1 || !b1 || !w2 || !b2 || !w3)
        return fail(NCAHIP_EINVAL, "cond step: null pointer");
    if (!dims_ok(B, C, H, W) || hidden <= 0 || goal_ch < 0) return fail(NCAHIP_EINVAL, "cond step: bad size");
    if ((goal_ch > 0) != (goal != nullptr)) return fail(NCAHIP_EINVAL, "cond step: goal pointer / goal_ch mismatch");
    if (goal_ch > C || alive_ch >= C) return fail(NCAHIP_EINVAL, "cond step: goal_ch/alive_ch outside C=%d", C);
    if (C > max_c || hidden > kMaxHidden)
        return fail(NCAHIP_ERANGE, "cond step: C=%d hidden=%d exceeds (%d,%d)", C, hidden, max_c, kMaxHidden);
    if (x_in == x_out) return fail(NCAHIP_EINVAL, "cond step: x_in and x_out must not alias (halo reads)");
    return 0;
}

}  // namespace

// ---- sticky device error word: one host-mapped word per device, written by kernels (system-scope atomic OR), read by the host
// without synchronising -------------------------------------------------------------------------------------------------
namespace {
struct ErrWord {
    std::atomic<int> state{0};   // 0 = not allocated, 1 = ready, 2 = allocation failed
    unsigned* host = nullptr;
    unsigned* dev = nullptr;
};
ErrWord g_errw[kNcaMaxDevices];
ErrWord& errw() {
    ErrWord& w = g_errw[nca_device_index()];
    if (w.state.load(std::memory_order_acquire) == 0) {
        static std::atomic_flag lock = ATOMIC_FLAG_INIT;
        while (lock.test_and_set(std::memory_order_acquire)) {}
        if (w.state.load(std::memory_order_relaxed) == 0) {
            void *h = nullptr, *d = nullptr;
            int st = 2;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                *(volatile unsigned*)h = 0u;
                w.host = (unsigned*)h;
                w.dev = (unsigned*)d;
                st = 1;
            }
            w.state.store(st, std::memory_order_release);
        }
        lock.clear(std::memory_order_release);
    }
    return w;
}
}  // namespace
unsigned* nca_error_word_device() { return errw().dev; }
unsigned nca_error_word_read(bool clear) {
    ErrWord& w = errw();
    if (!w.host) return 0u;
    const unsigned v = *(volatile unsigned*)w.host;
    if (clear && v) *(volatile unsigned*)w.host = 0u;
    return v;
}
// positive return code of a recorded device-side failure (distinct from every hipError_t in use)
static int device_error_rc(const char* where) {
    const unsigned v = nca_error_word_read(false);
    if (!v) return 0;
    return fail(NCAHIP_EDEVICE, "%s: a device-side failure was recorded earlier on this device (error word 0x%x: bit 0 = "
                "producer/consumer hand-off poll expired, bit 1 = neighbour poll of the persistent DyNCA kernel expired); results since "
                "the last ncahip_check_errors are not valid", where, v);
}

extern "C" {

int ncahip_check_errors(ncahip_stream_t stream, int clear) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_result(e, "check_errors sync");
    const unsigned v = nca_error_word_read(clear != 0);
    if (!v) return 0;
    return fail(NCAHIP_EDEVICE, "device error word 0x%x (bit 0: a producer/consumer hand-off poll expired -- the affected launch "
                "produced stale tiles; bit 1: a neighbour poll of the persistent DyNCA kernel expired -- that launch stopped early)", v);
}

int ncahip_debug_inject_error(unsigned bits) {   // test hook: what a kernel does when a poll expires
    unsigned* h = errw().host;
    if (!h) return fail(NCAHIP_EINVAL, "no error word on this device");
    *(volatile unsigned*)h |= bits;
    return 0;
}

int ncahip_debug_persist_drop_tiles(int n) {   // test hook: see nca_set_persist_drop_tiles
    nca_set_persist_drop_tiles(n < 0 ? 0 : n);
    return 0;
}

int ncahip_version(void) { return NCAHIP_VERSION; }
const char* ncahip_last_error(void) { return g_err; }

int ncahip_cond_precision(int mode) {
    if (mode != 0 && mode != 1) return fail(NCAHIP_EINVAL, "cond precision: 0 (exact fp32) or 1 (bf16x3)");
    nca_set_cond_precision(mode);
    return 0;
}

int ncahip_debug_force_generic(int on) {
    nca_set_force_generic((on & 1) != 0);    // bit 0: generic any-shape kernels
    nca_set_cond_variant((on >> 1) & 1);     // bit 1: symmetric wave-private ConditionedNCA kernel instead of producer/consumer
    nca_set_bwd_bf16_exact((on & 4) != 0);   // bit 2: bf16-history backward with exact-f32 products instead of bf16 MFMA
    nca_set_bwd_variant((on & 8) ? 3 : 0);   // bit 3: ConditionedNCA backward kernel A in the form that is NOT the mode's default (one launch <-> front + matrix)
    nca_set_bwd_fm_nosplit((on & 16) != 0);  // bit 4: the matrix kernel walks whole super-tiles on small grids too (its summation order then equals the one-launch form's)
    return 0;
}

int ncahip_limits(int* max_c, int* max_fc, int* max_hidden) {
    if (max_c) *max_c = kMaxC;
    if (max_fc) *max_fc = kMaxFc;
    if (max_hidden) *max_hidden = kMaxHidden;
    return 0;
}

int ncahip_selftest(void* scratch, ncahip_stream_t stream) {
    if (!scratch) return fail(NCAHIP_EINVAL, "selftest: null scratch");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, 64, st);
    if (e != hipSuccess) return hip_result(e, "selftest memset");
    e = nca_launch_selftest((int*)scratch, st);
    if (e != hipSuccess) return hip_result(e, "selftest launch");
    int host = 0;
    e = hipMemcpyAsync(&host, scratch, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_result(e, "selftest sync");
    return host == 1 ? 0 : fail(NCAHIP_ERANGE, "selftest: MFMA 16x16x4 f32 lane map differs from the one assumed");
}

int ncahip_dynca_perceive_f32(const float* x, float* y, int B, int C, int H, int W, int pad_mode,
                              ncahip_stream_t stream) {
    if (!x || !y || x == y) return fail(NCAHIP_EINVAL, "dynca_perceive: null or aliased pointer");
    if (!dims_ok(B, C, H, W) || pad_mode < 0 || pad_mode > 3) return fail(NCAHIP_EINVAL, "dynca_perceive: bad argument");
    if (pad_mode == NCAHIP_PAD_REFLECT && (H < 2 || W < 2)) return fail(NCAHIP_EINVAL, "reflect pad needs H,W >= 2");
    return hip_result(nca_launch_dynca_perceive(x, y, B, C, H, W, pad_mode, (hipStream_t)stream), "dynca_perceive");
}

int ncahip_cond_perceive_f32(const float* z, const float* wp, float* y, int B, int C, int H, int W,
                             ncahip_stream_t stream) {
    if (!z || !wp || !y || z == y) return fail(NCAHIP_EINVAL, "cond_perceive: null or aliased pointer");
    if (!dims_ok(B, C, H, W)) return fail(NCAHIP_EINVAL, "cond_perceive: bad size");
    return hip_result(nca_launch_cond_perceive(z, wp, y, B, C, H, W, (hipStream_t)stream), "cond_perceive");
}

int ncahip_dynca_step_fwd_f32(const float* x_in, float* x_out, const float* cond, const float* u, const float* w1,
                              const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                              int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step,
                              ncahip_stream_t stream) {
    if (int rc = check_dynca(x_in, x_out, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, kMaxFcFwd)) return rc;
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, update_rate, true)) return rc;
    NcaDyncaArgs a{x_in, x_out, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step};
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_dynca_step_fwd(a, (hipStream_t)stream), "dynca_step_fwd");
}

int ncahip_dynca_nsteps_fwd_f32(float* states, int ring, int T, const float* cond, const float* u, const float* w1,
                                const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                                int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step0,
                                ncahip_stream_t stream) {
    if (ring < 2 || T < 0) return fail(NCAHIP_EINVAL, "dynca nsteps: ring >= 2 and T >= 0 required");
    if (int rc = check_dynca(states, states + 1, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, kMaxFcFwd)) return rc;
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, update_rate, true)) return rc;
    const size_t slot = (size_t)B * C * H * W, uslot = (size_t)B * H * W;
    for (int t = 0; t < T; ++t) {
        NcaDyncaArgs a{states + (size_t)(t % ring) * slot, states + (size_t)((t + 1) % ring) * slot, cond,
                       u_at(u, ubits, t, uslot), w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode,
                       update_rate, seed, step0 + (uint64_t)t};
        a.u_bits = ubits;
        if (int rc = hip_result(nca_launch_dynca_step_fwd(a, (hipStream_t)stream), "dynca_nsteps_fwd")) return rc;
    }
    return 0;
}

// ---- T steps in ONE launch for small grids (B = 1 video inference), nca_dynca_persist.hip -------------------------------------
size_t ncahip_dynca_nsteps_persist_workspace(int B, int C, int H, int W, int fc, int c_cond) {
    if (!dims_ok(B, C, H, W) || !nca_dynca_persist_shape_ok(B, C, H, W, fc, c_cond)) return 0;
    // abort word + the exchange: 2 parities x tiles x C x (60 fine ring cells + 48 coarse means) (value, tag) pairs
    return 256 + align256((size_t)2 * nca_dynca_persist_tiles(B, H, W) * C * 108 * sizeof(unsigned long long));      // (sized for the two-scale exchange)
}

static int dynca_persist_impl(bool two_scale, const float* x_in, float* x_out, int T, const float* cond, const float* u, const float* w1,
                                        const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond,
                                        int pad_mode, float update_rate, uint64_t seed, uint64_t step0, void* workspace,
                                        size_t workspace_bytes, unsigned epoch, ncahip_stream_t stream) {
    if (T < 1 || !workspace) return fail(NCAHIP_EINVAL, "dynca nsteps persist: T >= 1 and a workspace required");
    if (epoch < 1 || epoch >= (1u << 20)) return fail(NCAHIP_EINVAL, "dynca nsteps persist: epoch must be in [1, 2^20) (zero the workspace and restart at 1 when it runs out)");
    if (int rc = check_dynca(x_in, x_out, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;   // (refuses x_in == x_out)
    const size_t need = ncahip_dynca_nsteps_persist_workspace(B, C, H, W, fc, c_cond);
    if (need == 0 || T >= 4096)
        return fail(NCAHIP_ERANGE, "dynca nsteps persist: shape not covered (C <= 16, fc <= 128, H %% 16 == 0, W %% 16 == 0, T < 4096); use ncahip_dynca_nsteps_fwd_f32");
    if (workspace_bytes < need) return fail(NCAHIP_EINVAL, "dynca nsteps persist: workspace too small");
    if ((((uintptr_t)x_in | (uintptr_t)x_out) & 15) != 0) return fail(NCAHIP_ERANGE, "dynca nsteps persist: 16-byte aligned states required");
    if (((uintptr_t)workspace & 255) != 0) return fail(NCAHIP_ERANGE, "dynca nsteps persist: 256-byte aligned workspace required");
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, update_rate, true)) return rc;
    if (int rc = device_error_rc("dynca nsteps persist")) return rc;
    hipStream_t st = (hipStream_t)stream;
    NcaDyncaPersistArgs a{x_in, x_out, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0,
                          (int*)workspace, epoch, (unsigned long long*)((char*)workspace + 256),
                          (size_t)nca_dynca_persist_tiles(B, H, W) * C * (two_scale ? 108 : 60), nullptr, ubits ? 1 : 0};
    bool fits = false;
    auto launch = two_scale ? nca_launch_dynca_persist_ms : nca_launch_dynca_persist;
    if (int rc = hip_result(launch(a, st, true, &fits), "dynca nsteps persist (occupancy)")) return rc;
    if (!fits) return fail(NCAHIP_ERANGE, "dynca nsteps persist: %d tiles cannot all be resident on this device; use ncahip_dynca_nsteps_fwd_f32",
                           nca_dynca_persist_tiles(B, H, W));
    return hip_result(launch(a, st, false, &fits), "dynca_nsteps_fwd_persist");     // ONE launch: no copy, no memset
}

int ncahip_dynca_nsteps_fwd_persist_f32(const float* x_in, float* x_out, int T, const float* cond, const float* u, const float* w1,
                                        const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond,
                                        int pad_mode, float update_rate, uint64_t seed, uint64_t step0, void* workspace,
                                        size_t workspace_bytes, unsigned epoch, ncahip_stream_t stream) {
    return dynca_persist_impl(false, x_in, x_out, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0,
                              workspace, workspace_bytes, epoch, stream);
}
int ncahip_dynca_nsteps_fwd_persist_ms_f32(const float* x_in, float* x_out, int T, const float* cond, const float* u, const float* w1,
                                           const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                                           int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step0, void* workspace,
                                           size_t workspace_bytes, unsigned epoch, ncahip_stream_t stream) {
    return dynca_persist_impl(true, x_in, x_out, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0,
                              workspace, workspace_bytes, epoch, stream);
}

// ---- conditioning front ends (fixed-filter part of the encoders) ---------------------------------------------------------
int ncahip_image_encoder_front_f32(const float* img, const float* k3, const float* k5, float* feat, int B, int ch, int H, int W,
                                   ncahip_stream_t stream) {
    if (!img || !k3 || !k5 || !feat || img == feat) return fail(NCAHIP_EINVAL, "image_encoder_front: null or aliased pointer");
    if (!dims_ok(B, ch, H, W)) return fail(NCAHIP_EINVAL, "image_encoder_front: bad size");
    if (ch > 8) return fail(NCAHIP_ERANGE, "image_encoder_front: %d image channels (at most 8)", ch);
    return hip_result(nca_launch_image_encoder_front(img, k3, k5, feat, B, ch, H, W, (hipStream_t)stream), "image_encoder_front");
}

int ncahip_edge_extractor_f32(const float* img, const float* k3, float* out, int B, int H, int W, int apply_tanh, ncahip_stream_t stream) {
    if (!img || !k3 || !out || img == out) return fail(NCAHIP_EINVAL, "edge_extractor: null or aliased pointer");
    if (!dims_ok(B, 1, H, W)) return fail(NCAHIP_EINVAL, "edge_extractor: bad size");
    return hip_result(nca_launch_edge_extractor(img, k3, out, B, H, W, apply_tanh != 0, (hipStream_t)stream), "edge_extractor");
}

// ---- two-scale perception (perception_scales = [0, 1]): coarse pass + fused step with on-the-fly bilinear up-sampling --------
static int check_ms(int C, int H, int W, int fc, const void* pc) {
    if (!pc) return fail(NCAHIP_EINVAL, "dynca two-scale step: pc_scratch required");
    if ((H | W) & 1) return fail(NCAHIP_ERANGE, "dynca two-scale step: H and W must be even (the x2 resampling is then the exact 2x2 mean / (0.25, 0.75) blend)");
    if (C > kMaxC || fc > kMaxFc) return fail(NCAHIP_ERANGE, "dynca two-scale step: C=%d fc=%d exceeds (%d,%d)", C, fc, kMaxC, kMaxFc);
    return 0;
}

int ncahip_dynca_step_fwd_ms_f32(const float* x_in, float* x_out, const float* cond, const float* u, const float* w1,
                                 const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                                 int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step, float* pc_scratch,
                                 ncahip_stream_t stream) {
    if (int rc = check_dynca(x_in, x_out, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    if (int rc = check_ms(C, H, W, fc, pc_scratch)) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (int rc = hip_result(nca_launch_dynca_coarse_perceive(x_in, pc_scratch, B, C, H, W, pad_mode, st), "dynca coarse perceive")) return rc;
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, update_rate, true)) return rc;
    NcaDyncaArgs a{x_in, x_out, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step};
    a.pc = pc_scratch;
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_dynca_step_fwd(a, st), "dynca_step_fwd_ms");
}

int ncahip_dynca_nsteps_fwd_ms_f32(float* states, int ring, int T, const float* cond, const float* u, const float* w1,
                                   const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                                   int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step0, float* pc_scratch,
                                   ncahip_stream_t stream) {
    if (ring < 2 || T < 0) return fail(NCAHIP_EINVAL, "dynca nsteps: ring >= 2 and T >= 0 required");
    if (int rc = check_dynca(states, states + 1, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    if (int rc = check_ms(C, H, W, fc, pc_scratch)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, update_rate, true)) return rc;
    const size_t slot = (size_t)B * C * H * W, uslot = (size_t)B * H * W;
    for (int t = 0; t < T; ++t) {
        const float* const xi = states + (size_t)(t % ring) * slot;
        if (int rc = hip_result(nca_launch_dynca_coarse_perceive(xi, pc_scratch, B, C, H, W, pad_mode, st), "dynca coarse perceive")) return rc;
        NcaDyncaArgs a{xi, states + (size_t)((t + 1) % ring) * slot, cond, u_at(u, ubits, t, uslot), w1, b1, w2, b2, B, C, H,
                       W, fc, c_cond, pad_mode, update_rate, seed, step0 + (uint64_t)t};
        a.pc = pc_scratch;
        a.u_bits = ubits;
        if (int rc = hip_result(nca_launch_dynca_step_fwd(a, st), "dynca_nsteps_fwd_ms")) return rc;
    }
    return 0;
}

int ncahip_cond_step_fwd_f32(const float* x_in, const uint8_t* pre_in, float* x_out, uint8_t* pre_out,
                             const float* goal, int goal_ch, const float* u, const float* wp, const float* w1,
                             const float* b1, const float* w2, const float* b2, const float* w3, int B, int C, int H,
                             int W, int hidden, int alive_ch, float alive_thr, float fire_rate, float clamp_lo,
                             float clamp_hi, uint64_t seed, uint64_t step, ncahip_stream_t stream) {
    if (int rc = check_cond(x_in, x_out, pre_out, goal, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch, alive_ch, kMaxCCondFwd))
        return rc;
    if (pre_in && pre_in == pre_out) return fail(NCAHIP_EINVAL, "cond step: pre_in and pre_out must not alias");
    NcaCondArgs a{x_in, pre_in, x_out, pre_out, goal, u, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch,
                  alive_ch, alive_thr, fire_rate, clamp_lo, clamp_hi, seed, step};
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, fire_rate, false)) return rc;
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_cond_step_fwd(a, (hipStream_t)stream), "cond_step_fwd");
}

int ncahip_cond_finalize_f32(const float* x_pend, const uint8_t* pre, float* x_out, int B, int C, int H, int W,
                             int alive_ch, float alive_thr, float clamp_lo, float clamp_hi, ncahip_stream_t stream) {
    if (!x_pend || !x_out || (alive_ch >= 0 && !pre)) return fail(NCAHIP_EINVAL, "cond_finalize: null pointer");
    if (!dims_ok(B, C, H, W) || alive_ch >= C) return fail(NCAHIP_EINVAL, "cond_finalize: bad size");
    if (alive_ch >= 0 && x_pend == x_out) return fail(NCAHIP_EINVAL, "cond_finalize: in-place needs alive_ch < 0");
    return hip_result(nca_launch_cond_finalize(x_pend, pre, x_out, B, C, H, W, alive_ch, alive_thr, clamp_lo, clamp_hi,
                                               (hipStream_t)stream), "cond_finalize");
}

// ---- bf16 state storage (nca_cond_bf16.hip) ---------------------------------------------------------------
static int check_bf16_shape(const void* x_in, const void* x_out, const void* goal, int H, int W) {
    if (W % 4 != 0 || (((uintptr_t)x_in | (uintptr_t)x_out | (uintptr_t)goal) & 7) != 0)
        return fail(NCAHIP_ERANGE, "bf16 cond step: needs W %% 4 == 0 and 8-byte aligned state / goal tensors");
    if ((size_t)H * W >= ((size_t)1 << 24)) return fail(NCAHIP_ERANGE, "bf16 cond step: H*W must be below 2^24");
    return 0;
}

int ncahip_cond_step_fwd_bf16(const uint16_t* x_in, const uint8_t* pre_in, uint16_t* x_out, uint8_t* pre_out,
                              const uint16_t* goal, int goal_ch, const float* u, const float* wp, const float* w1,
                              const float* b1, const float* w2, const float* b2, const float* w3, int B, int C, int H,
                              int W, int hidden, int alive_ch, float alive_thr, float fire_rate, float clamp_lo,
                              float clamp_hi, uint64_t seed, uint64_t step, ncahip_stream_t stream) {
    if (int rc = check_cond(x_in, x_out, pre_out, goal, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch, alive_ch, kMaxCCondFwdBf16))
        return rc;
    if (int rc = check_bf16_shape(x_in, x_out, goal, H, W)) return rc;
    if (pre_in && pre_in == pre_out) return fail(NCAHIP_EINVAL, "cond step: pre_in and pre_out must not alias");
    NcaCondArgs a{reinterpret_cast<const float*>(x_in), pre_in, reinterpret_cast<float*>(x_out), pre_out,
                  reinterpret_cast<const float*>(goal), u, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch,
                  alive_ch, alive_thr, fire_rate, clamp_lo, clamp_hi, seed, step};
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, fire_rate, false)) return rc;
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_cond_step_fwd_bf16(a, (hipStream_t)stream), "cond_step_fwd_bf16");
}

int ncahip_cond_finalize_bf16(const uint16_t* x_pend, const uint8_t* pre, uint16_t* x_out, int B, int C, int H, int W,
                              int alive_ch, float alive_thr, float clamp_lo, float clamp_hi, ncahip_stream_t stream) {
    if (!x_pend || !x_out || (alive_ch >= 0 && !pre)) return fail(NCAHIP_EINVAL, "cond_finalize: null pointer");
    if (!dims_ok(B, C, H, W) || alive_ch >= C) return fail(NCAHIP_EINVAL, "cond_finalize: bad size");
    if (alive_ch >= 0 && x_pend == x_out) return fail(NCAHIP_EINVAL, "cond_finalize: in-place needs alive_ch < 0");
    return hip_result(nca_launch_cond_finalize_bf16(x_pend, pre, x_out, B, C, H, W, alive_ch, alive_thr, clamp_lo,
                                                    clamp_hi, (hipStream_t)stream), "cond_finalize_bf16");
}

int ncahip_cond_grow_fwd_bf16(uint16_t* states, uint8_t* pre, int ring, int T, uint16_t* x_final, const uint16_t* goal,
                              int goal_ch, const float* u, const float* wp, const float* w1, const float* b1,
                              const float* w2, const float* b2, const float* w3, int B, int C, int H, int W, int hidden,
                              int alive_ch, float alive_thr, float fire_rate, float clamp_lo, float clamp_hi,
                              uint64_t seed, uint64_t step0, ncahip_stream_t stream) {
    if (ring < 2 || T < 1 || !pre || !x_final) return fail(NCAHIP_EINVAL, "cond grow: ring >= 2, T >= 1, buffers required");
    if (int rc = check_cond(states, states + 1, pre, goal, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch, alive_ch, kMaxCCondFwdBf16))
        return rc;
    if (int rc = check_bf16_shape(states, states, goal, H, W)) return rc;
    if (int rc = device_error_rc("cond grow (bf16)")) return rc;
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, fire_rate, false)) return rc;
    const size_t slot = (size_t)B * C * H * W, pslot = (size_t)B * H * W;
    if ((slot * sizeof(uint16_t)) % 8 != 0) return fail(NCAHIP_ERANGE, "bf16 cond grow: state slots must stay 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int sl = T % ring;
    if (x_final == states + (size_t)sl * slot && alive_ch >= 0)
        return fail(NCAHIP_EINVAL, "cond grow: x_final must not alias the last state slot");
    for (int t = 0; t < T; ++t) {
        const int si = t % ring, so = (t + 1) % ring;
        NcaCondArgs a{reinterpret_cast<const float*>(states + (size_t)si * slot), t == 0 ? nullptr : pre + (size_t)si * pslot,
                      reinterpret_cast<float*>(states + (size_t)so * slot), pre + (size_t)so * pslot,
                      reinterpret_cast<const float*>(goal), u_at(u, ubits, t, pslot), wp, w1, b1, w2, b2, w3, B, C, H,
                      W, hidden, goal_ch, alive_ch, alive_thr, fire_rate, clamp_lo, clamp_hi, seed, step0 + (uint64_t)t};
        a.u_bits = ubits;
        if (int rc = hip_result(nca_launch_cond_step_fwd_bf16(a, st), "cond_grow_fwd_bf16")) return rc;
    }
    return hip_result(nca_launch_cond_finalize_bf16(states + (size_t)sl * slot, pre + (size_t)sl * pslot, x_final, B, C, H, W,
                                                    alive_ch, alive_thr, clamp_lo, clamp_hi, st), "cond_grow finalize (bf16)");
}

int ncahip_cond_alive_u8(const float* x, uint8_t* out, int B, int C, int H, int W, int alive_ch, float alive_thr,
                         ncahip_stream_t stream) {
    if (!x || !out) return fail(NCAHIP_EINVAL, "cond_alive: null pointer");
    if (!dims_ok(B, C, H, W) || alive_ch >= C) return fail(NCAHIP_EINVAL, "cond_alive: bad size");
    return hip_result(nca_launch_cond_alive(x, out, B, C, H, W, alive_ch, alive_thr, (hipStream_t)stream), "cond_alive");
}

int ncahip_cond_grow_fwd_f32(float* states, uint8_t* pre, int ring, int T, float* x_final, const float* goal,
                             int goal_ch, const float* u, const float* wp, const float* w1, const float* b1,
                             const float* w2, const float* b2, const float* w3, int B, int C, int H, int W, int hidden,
                             int alive_ch, float alive_thr, float fire_rate, float clamp_lo, float clamp_hi,
                             uint64_t seed, uint64_t step0, ncahip_stream_t stream) {
    if (ring < 2 || T < 1 || !pre || !x_final) return fail(NCAHIP_EINVAL, "cond grow: ring >= 2, T >= 1, buffers required");
    if (int rc = check_cond(states, states + 1, pre, goal, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch, alive_ch, kMaxCCondFwd))
        return rc;
    if (int rc = device_error_rc("cond grow")) return rc;
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, fire_rate, false)) return rc;
    const size_t slot = (size_t)B * C * H * W, pslot = (size_t)B * H * W;
    hipStream_t st = (hipStream_t)stream;
    const int sl = T % ring;
    if (x_final == states + (size_t)sl * slot && alive_ch >= 0)
        return fail(NCAHIP_EINVAL, "cond grow: x_final must not alias the last state slot");
    for (int t = 0; t < T; ++t) {
        const int si = t % ring, so = (t + 1) % ring;
        NcaCondArgs a{states + (size_t)si * slot, t == 0 ? nullptr : pre + (size_t)si * pslot,
                      states + (size_t)so * slot, pre + (size_t)so * pslot, goal,
                      u_at(u, ubits, t, pslot), wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch,
                      alive_ch, alive_thr, fire_rate, clamp_lo, clamp_hi, seed, step0 + (uint64_t)t};
        a.u_bits = ubits;
        if (int rc = hip_result(nca_launch_cond_step_fwd(a, st), "cond_grow_fwd")) return rc;
    }
    return hip_result(nca_launch_cond_finalize(states + (size_t)sl * slot, pre + (size_t)sl * pslot, x_final, B, C, H, W,
                                               alive_ch, alive_thr, clamp_lo, clamp_hi, st), "cond_grow finalize");
}

// ---- bf16 state storage for the DyNCA step (same kernel, exact f32 compute, RNE on store) ----------------------
int ncahip_dynca_step_fwd_bf16(const uint16_t* x_in, uint16_t* x_out, const float* cond, const float* u, const float* w1,
                               const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                               int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step,
                               ncahip_stream_t stream) {
    if (int rc = check_dynca(x_in, x_out, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    NcaDyncaArgs a{reinterpret_cast<const float*>(x_in), reinterpret_cast<float*>(x_out), cond, u, w1, b1, w2, b2, B, C, H, W,
                   fc, c_cond, pad_mode, update_rate, seed, step};
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, update_rate, true)) return rc;
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_dynca_step_fwd_bf16(a, (hipStream_t)stream), "dynca_step_fwd_bf16");
}

int ncahip_dynca_nsteps_fwd_bf16(uint16_t* states, int ring, int T, const float* cond, const float* u, const float* w1,
                                 const float* b1, const float* w2, const float* b2, int B, int C, int H, int W, int fc,
                                 int c_cond, int pad_mode, float update_rate, uint64_t seed, uint64_t step0,
                                 ncahip_stream_t stream) {
    if (ring < 2 || T < 0) return fail(NCAHIP_EINVAL, "dynca nsteps: ring >= 2 and T >= 0 required");
    if (int rc = check_dynca(states, states + 1, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, update_rate, true)) return rc;
    const size_t slot = (size_t)B * C * H * W, uslot = (size_t)B * H * W;
    for (int t = 0; t < T; ++t) {
        NcaDyncaArgs a{reinterpret_cast<const float*>(states + (size_t)(t % ring) * slot),
                       reinterpret_cast<float*>(states + (size_t)((t + 1) % ring) * slot), cond,
                       u_at(u, ubits, t, uslot), w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode,
                       update_rate, seed, step0 + (uint64_t)t};
        a.u_bits = ubits;
        if (int rc = hip_result(nca_launch_dynca_step_fwd_bf16(a, (hipStream_t)stream), "dynca_nsteps_fwd_bf16")) return rc;
    }
    return 0;
}

int ncahip_dynca_step_bwd_f32(const float* x_t, const float* cond, const float* u, const float* w1, const float* b1,
                              const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                              float update_rate, uint64_t seed, uint64_t step, const float* g_next, float* g_x, float* h_out,
                              float* dh_out, float* dy_scratch, ncahip_stream_t stream) {
    if (!g_next || !g_x || !h_out || !dh_out || !dy_scratch) return fail(NCAHIP_EINVAL, "dynca step bwd: null pointer");
    if (int rc = check_dynca(x_t, g_x, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    if (g_next == g_x) return fail(NCAHIP_EINVAL, "dynca step bwd: g_next and g_x must not alias");
    if ((size_t)(fc > 4 * C ? fc : 4 * C) * H * W * sizeof(float) >= ((size_t)1 << 32))
        return fail(NCAHIP_ERANGE, "dynca step bwd: fc*H*W*4 must stay below 4 GiB (32-bit store offsets inside a batch item)");
    NcaDyncaArgs a{x_t, nullptr, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step,
                   g_next, h_out, dh_out, dy_scratch, g_x};
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, update_rate, true)) return rc;
    a.u_bits = u_is_bits(u, seed);
    return hip_result(nca_launch_dynca_step_bwd(a, (hipStream_t)stream), "dynca_step_bwd");
}

// ---- backward of one DyNCA step with the layer-2 weight gradient fused in (no h buffer) ---------------------------------
size_t ncahip_dynca_step_bwd_w2_workspace(int B, int C, int H, int W, int fc) {
    if (!dims_ok(B, C, H, W) || fc <= 0) return 0;
    return (size_t)nca_dynca_bwd_grid(B, H, W) * ((size_t)C * fc + C) * sizeof(float);
}

int ncahip_dynca_step_bwd_w2_f32(const float* x_t, const float* cond, const float* u, const float* w1, const float* b1,
                                 const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step, const float* g_next, float* g_x,
                                 float* dh_out, float* dy_scratch, float* gw2_out, int accumulate, void* workspace,
                                 size_t workspace_bytes, ncahip_stream_t stream) {
    if (!g_next || !g_x || !dh_out || !dy_scratch || !gw2_out || !workspace) return fail(NCAHIP_EINVAL, "dynca step bwd_w2: null pointer");
    if (int rc = check_dynca(x_t, g_x, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode)) return rc;
    if (g_next == g_x) return fail(NCAHIP_EINVAL, "dynca step bwd_w2: g_next and g_x must not alias");
    if ((size_t)(fc > 4 * C ? fc : 4 * C) * H * W * sizeof(float) >= ((size_t)1 << 32))
        return fail(NCAHIP_ERANGE, "dynca step bwd_w2: fc*H*W*4 must stay below 4 GiB (32-bit store offsets inside a batch item)");
    if (workspace_bytes < ncahip_dynca_step_bwd_w2_workspace(B, C, H, W, fc)) return fail(NCAHIP_EINVAL, "dynca step bwd_w2: workspace too small");
    NcaDyncaArgs a{x_t, nullptr, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step,
                   g_next, nullptr, dh_out, dy_scratch, g_x};
    a.gw2_ws = (float*)workspace;
    if (int rc = check_bits(u_is_bits(u, seed), B, H, W, update_rate, true)) return rc;
    a.u_bits = u_is_bits(u, seed);
    if (int rc = hip_result(nca_launch_dynca_step_bwd(a, (hipStream_t)stream), "dynca_step_bwd_w2")) return rc;
    return hip_result(nca_launch_reduce_rows((const float*)workspace, gw2_out, nca_dynca_bwd_grid_c(B, C, H, W), C * fc + C,
                                             (hipStream_t)stream, accumulate != 0), "dynca_step_bwd_w2 reduce");
}

// ---- backward of ncahip_dynca_nsteps_fwd_f32: the whole T-step loop on the stream, caller-owned workspace ----------------
namespace {
struct DyncaBwdPlan {
    int nsl, fs;                 // hidden-layer slices of at most 128 units (fs = width of the full slices)
    size_t n, off_g[2], off_y, off_dy, off_dh, off_ws2, off_wsg, off_acc1, off_acc2, off_pc, off_dpc, off_dxc, off_x32, total;
    int grid2, gridg, K1;
};
DyncaBwdPlan dynca_bwd_plan(int B, int C, int H, int W, int fc, int c_cond, bool two_scale = false, bool bf16 = false) {
    DyncaBwdPlan p{};
    p.nsl = (fc + 127) / 128;
    p.fs = fc < 128 ? fc : 128;
    p.K1 = 4 * C + c_cond;
    p.n = (size_t)B * C * H * W;
    p.grid2 = two_scale ? nca_dynca_bwd_ms_grid(B, H, W) : nca_dynca_bwd_grid_c(B, C, H, W);
    p.gridg = nca_gram_grid(B, H * W);
    size_t o = 0;
    auto take = [&](size_t floats) { size_t at = o; o += (floats * sizeof(float) + 255) & ~(size_t)255; return at; };
    p.off_g[0] = take(p.n);
    p.off_g[1] = take(p.n);
    p.off_y = take(4 * p.n);
    p.off_dy = take(4 * p.n);
    p.off_dh = take((size_t)B * p.fs * H * W);
    p.off_ws2 = take((size_t)p.grid2 * ((size_t)C * p.fs + C));
    p.off_wsg = take((size_t)p.gridg * ((size_t)p.fs * p.K1 + p.fs));
    p.off_acc1 = take((size_t)p.nsl * ((size_t)p.fs * p.K1 + p.fs));
    p.off_acc2 = take((size_t)p.nsl * ((size_t)C * p.fs + C));
    if (two_scale) {   // coarse-level perception, its gradient, and dL/dx of the coarse level
        p.off_pc = take(p.n);      // 4C * (H/2) * (W/2) = C*H*W floats
        p.off_dpc = take(p.n);
        p.off_dxc = take(p.n / 4);
    }
    if (bf16) p.off_x32 = take(p.n);   // x_t widened to fp32
    p.total = o;
    return p;
}
}  // namespace

size_t ncahip_dynca_nsteps_bwd_workspace(int B, int C, int H, int W, int fc, int c_cond) {
    if (!dims_ok(B, C, H, W) || fc <= 0 || c_cond < 0) return 0;
    return dynca_bwd_plan(B, C, H, W, fc, c_cond).total;
}
size_t ncahip_dynca_nsteps_bwd_ms_workspace(int B, int C, int H, int W, int fc, int c_cond) {
    if (!dims_ok(B, C, H, W) || fc <= 0 || c_cond < 0) return 0;
    return dynca_bwd_plan(B, C, H, W, fc, c_cond, true).total;
}

static int dynca_nsteps_bwd_impl(bool two_scale, const void* states_v, int sb, int T, const float* cond, const float* u, const float* w1, const float* b1,
                                const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                float update_rate, uint64_t seed, uint64_t step0, const float* g_final, const float* g_states,
                                float* g_x0, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                                size_t workspace_bytes, ncahip_stream_t stream) {
    const char* const states = (const char*)states_v;
    const bool bf16 = sb == 2;
    if (T < 1 || !states || !g_final || !g_x0 || !g_w1 || !g_b1 || !g_w2 || !g_b2 || !workspace)
        return fail(NCAHIP_EINVAL, "dynca nsteps bwd: null pointer or T < 1");
    if (int rc = check_dynca(states, g_x0, cond, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, kMaxFcFwd)) return rc;
    if (bf16 && ((((size_t)B * C * H * W) & 3) != 0 || ((uintptr_t)states & 7) != 0))
        return fail(NCAHIP_ERANGE, "dynca nsteps bwd (bf16): B*C*H*W %% 4 == 0 and 8-byte aligned states required");
    if ((size_t)(128 > 4 * C ? 128 : 4 * C) * H * W * sizeof(float) >= ((size_t)1 << 32))
        return fail(NCAHIP_ERANGE, "dynca nsteps bwd: 4C*H*W*4 must stay below 4 GiB (32-bit store offsets inside a batch item)");
    if (two_scale) {
        if (int rc = check_ms(C, H, W, fc, workspace)) return rc;
    }
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, update_rate, true)) return rc;
    const DyncaBwdPlan p = dynca_bwd_plan(B, C, H, W, fc, c_cond, two_scale, bf16);
    if (workspace_bytes < p.total) return fail(NCAHIP_EINVAL, "dynca nsteps bwd: workspace too small");
    if (((uintptr_t)workspace & 15) != 0) return fail(NCAHIP_ERANGE, "dynca nsteps bwd: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* const ws = (char*)workspace;
    float* const gbuf[2] = {(float*)(ws + p.off_g[0]), (float*)(ws + p.off_g[1])};
    float* const y = (float*)(ws + p.off_y);
    float* const dy = (float*)(ws + p.off_dy);
    float* const dh = (float*)(ws + p.off_dh);
    float* const ws2 = (float*)(ws + p.off_ws2);
    float* const wsg = (float*)(ws + p.off_wsg);
    float* const acc1 = (float*)(ws + p.off_acc1);
    float* const acc2 = (float*)(ws + p.off_acc2);
    float* const pcb = two_scale ? (float*)(ws + p.off_pc) : nullptr;
    float* const dpc = two_scale ? (float*)(ws + p.off_dpc) : nullptr;
    float* const dxc = two_scale ? (float*)(ws + p.off_dxc) : nullptr;
    const size_t a1n = (size_t)p.fs * p.K1 + p.fs, a2n = (size_t)C * p.fs + C;
    hipError_t e = hipMemsetAsync(acc1, 0, (size_t)p.nsl * a1n * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(acc2, 0, (size_t)p.nsl * a2n * sizeof(float), st);
    if (e != hipSuccess) return hip_result(e, "dynca nsteps bwd memset");
    const size_t slot = p.n, uslot = (size_t)B * H * W;
    const float* gcur = g_final;
    for (int t = T - 1; t >= 0; --t) {
        const float* x_t = reinterpret_cast<const float*>(states + (size_t)t * slot * sb);
        if (bf16) {   // bf16 history (storage format only: the DyNCA step computes in fp32 on the widened state)
            float* const x32 = (float*)(ws + p.off_x32);
            if (int rc = hip_result(nca_launch_widen_bf16(reinterpret_cast<const uint16_t*>(x_t), x32, slot, st), "dynca nsteps bwd widen")) return rc;
            x_t = x32;
        }
        float* const g_out = t == 0 ? g_x0 : gbuf[t & 1];
        // y (the B rows of the layer-1 weight-gradient product) is written by the first slice's step kernel, which recomputes it anyway
        if (two_scale) {   // coarse level of the two-scale perception: input of the step kernel
            if (int rc = hip_result(nca_launch_dynca_coarse_perceive(x_t, pcb, B, C, H, W, pad_mode, st), "dynca nsteps bwd coarse perceive")) return rc;
        }
        for (int sl = 0; sl < p.nsl; ++sl) {
            const int h0 = sl * 128, fs = fc - h0 < 128 ? fc - h0 : 128;
            NcaDyncaArgs a{x_t, nullptr, cond, u_at(u, ubits, t, uslot), w1 + (size_t)h0 * p.K1, b1 + h0, w2 + h0, b2, B, C, H, W,
                           fs, c_cond, pad_mode, update_rate, seed, step0 + (uint64_t)t, gcur, nullptr, dh, dy, g_out};
            a.w2_ld = fc;
            a.u_bits = ubits;
            a.gw2_ws = ws2;
            a.pc = pcb;
            a.ybuf = sl == 0 ? y : nullptr;
            if (int rc = hip_result(nca_launch_dynca_step_bwd_mlp(a, st, sl > 0), "dynca nsteps bwd step")) return rc;
            // slabs hold [C x fs | C] of THIS slice (compact); slices narrower than p.fs use the front of their accumulator
            if (int rc = hip_result(nca_launch_reduce_rows(ws2, acc2 + (size_t)sl * a2n, p.grid2, C * fs + C, st, true), "dynca nsteps bwd reduce")) return rc;
            if (int rc = hip_result(nca_launch_gram_rows(dh, fs, y, 4 * C, cond, c_cond, B, H * W, acc1 + (size_t)sl * a1n, wsg, st, true),
                                    "dynca nsteps bwd gram")) return rc;
        }
        NcaDyncaArgs s{x_t, nullptr, cond, nullptr, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0 + (uint64_t)t,
                       gcur, nullptr, dh, dy, g_out};
        s.g_extra = g_states ? g_states + (size_t)t * slot : nullptr;
        if (two_scale) {
            // dL/dy splits evenly over the two levels: coarse level = up2^T, then the coarse grid's stencil adjoint (pad mode
            // resolved there), then the adjoint of the 2x2 mean inside the fine-level kernel
            if (int rc = hip_result(nca_launch_dynca_ms_upT(dy, dpc, B, C, H, W, st), "dynca nsteps bwd upT")) return rc;
            NcaDyncaArgs cs{nullptr, nullptr, nullptr, nullptr, w1, b1, w2, b2, B, C, H / 2, W / 2, fc, c_cond, pad_mode, update_rate, seed, 0,
                            nullptr, nullptr, nullptr, dpc, dxc};
            if (int rc = hip_result(nca_launch_dynca_step_bwd_stencil(cs, st), "dynca nsteps bwd coarse stencil")) return rc;
            s.coarse_add = dxc;
            s.dy_half = 1;
        }
        if (int rc = hip_result(nca_launch_dynca_step_bwd_stencil(s, st), "dynca nsteps bwd stencil")) return rc;
        gcur = g_out;
    }
    // accumulators -> gradients in the reference layouts: w1 [fc, K1] rows of a slice are contiguous, w2 [C, fc] columns are not
    for (int sl = 0; sl < p.nsl && e == hipSuccess; ++sl) {
        const int h0 = sl * 128, fs = fc - h0 < 128 ? fc - h0 : 128;
        const float* const s1 = acc1 + (size_t)sl * a1n;
        const float* const s2 = acc2 + (size_t)sl * a2n;
        e = hipMemcpyAsync(g_w1 + (size_t)h0 * p.K1, s1, (size_t)fs * p.K1 * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(g_b1 + h0, s1 + (size_t)fs * p.K1, fs * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(g_w2 + h0, (size_t)fc * sizeof(float), s2, (size_t)fs * sizeof(float), (size_t)fs * sizeof(float), C,
                                 hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess && sl == 0) e = hipMemcpyAsync(g_b2, s2 + (size_t)C * fs, C * sizeof(float), hipMemcpyDeviceToDevice, st);
    }
    return hip_result(e, "dynca nsteps bwd copy");
}

int ncahip_dynca_nsteps_bwd_f32(const float* states, int T, const float* cond, const float* u, const float* w1, const float* b1,
                                const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                float update_rate, uint64_t seed, uint64_t step0, const float* g_final, const float* g_states,
                                float* g_x0, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                                size_t workspace_bytes, ncahip_stream_t stream) {
    return dynca_nsteps_bwd_impl(false, states, 4, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0, g_final,
                                 g_states, g_x0, g_w1, g_b1, g_w2, g_b2, workspace, workspace_bytes, stream);
}
int ncahip_dynca_nsteps_bwd_ms_f32(const float* states, int T, const float* cond, const float* u, const float* w1, const float* b1,
                                   const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                   float update_rate, uint64_t seed, uint64_t step0, const float* g_final, const float* g_states,
                                   float* g_x0, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                                   size_t workspace_bytes, ncahip_stream_t stream) {
    return dynca_nsteps_bwd_impl(true, states, 4, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0, g_final,
                                 g_states, g_x0, g_w1, g_b1, g_w2, g_b2, workspace, workspace_bytes, stream);
}

size_t ncahip_dynca_nsteps_bwd_bf16_workspace(int B, int C, int H, int W, int fc, int c_cond) {
    if (!dims_ok(B, C, H, W) || fc <= 0 || c_cond < 0) return 0;
    return dynca_bwd_plan(B, C, H, W, fc, c_cond, false, true).total;
}
int ncahip_dynca_nsteps_bwd_bf16(const uint16_t* states, int T, const float* cond, const float* u, const float* w1, const float* b1,
                                 const float* w2, const float* b2, int B, int C, int H, int W, int fc, int c_cond, int pad_mode,
                                 float update_rate, uint64_t seed, uint64_t step0, const float* g_final, const float* g_states,
                                 float* g_x0, float* g_w1, float* g_b1, float* g_w2, float* g_b2, void* workspace,
                                 size_t workspace_bytes, ncahip_stream_t stream) {
    return dynca_nsteps_bwd_impl(false, states, 2, T, cond, u, w1, b1, w2, b2, B, C, H, W, fc, c_cond, pad_mode, update_rate, seed, step0, g_final,
                                 g_states, g_x0, g_w1, g_b1, g_w2, g_b2, workspace, workspace_bytes, stream);
}

// ---- weight-gradient products of the DyNCA backward (cell axis as K) -------------------------------------------------
size_t ncahip_gram_rows_workspace(int ma, int nb, int B, int HW) {
    if (ma <= 0 || nb <= 0 || B <= 0 || HW <= 0) return 0;
    return (size_t)nca_gram_grid(B, HW) * ((size_t)ma * nb + ma) * sizeof(float);
}

int ncahip_gram_rows_f32(const float* a, int ma, const float* b1, int nb1, const float* b2, int nb2, int B, int HW,
                         float* out, int accumulate, void* workspace, size_t workspace_bytes, ncahip_stream_t stream) {
    if (!a || !b1 || !out || !workspace || (nb2 > 0) != (b2 != nullptr)) return fail(NCAHIP_EINVAL, "gram_rows: null pointer");
    if (ma <= 0 || nb1 <= 0 || nb2 < 0 || B <= 0 || HW <= 0) return fail(NCAHIP_EINVAL, "gram_rows: bad size");
    const int nb = nb1 + nb2;
    if (!((ma <= 32 && nb <= 128) || (ma <= 128 && nb <= 144)))
        return fail(NCAHIP_ERANGE, "gram_rows: ma=%d nb=%d outside (<=32 x <=128) / (<=128 x <=144)", ma, nb);
    if (workspace_bytes < ncahip_gram_rows_workspace(ma, nb, B, HW)) return fail(NCAHIP_EINVAL, "gram_rows: workspace too small");
    return hip_result(nca_launch_gram_rows(a, ma, b1, nb1, b2, nb2, B, HW, out, (float*)workspace, (hipStream_t)stream,
                                           accumulate != 0), "gram_rows");
}

size_t ncahip_cond_grow_bwd_workspace(int B, int C, int H, int W, int hidden) {
    if (!dims_ok(B, C, H, W) || hidden <= 0) return 0;
    const size_t n = (size_t)B * C * H * W * sizeof(float);
    return 4 * align256(n) + align256(3 * n) +
           align256((size_t)(nca_cond_bwd_nslab() + 1) * nca_cond_bwd_slab_floats(C, hidden) * sizeof(float)) +
           align256((size_t)nca_cond_bwd_nblk(B, C, H, W) * 27 * sizeof(float)) +
           align256(nca_cond_bwd_fm_pscr_bytes(B, C, H, W)) + align256(nca_cond_bwd_fm_doscr_bytes(B, C, H, W)) + align256(kNcaCondBwdOpimgBytes);
}

// states / goal: fp32 or bf16 (sb = bytes per element); everything else fp32
static int cond_grow_bwd_impl(const void* states_v, int sb, const uint8_t* pre, int T, const void* goal_v, int goal_ch,
                              const float* u, const float* wp, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* w3, int B, int C, int H, int W, int hidden, int alive_ch,
                              float alive_thr, float fire_rate, float clamp_lo, float clamp_hi, uint64_t seed,
                              uint64_t step0, const float* g_final, float* g_x0, float* g_goal, float* g_wp, float* g_w1,
                              float* g_b1, float* g_w2, float* g_b2, float* g_w3, void* workspace, size_t workspace_bytes,
                              ncahip_stream_t stream) {
    const char* const states = (const char*)states_v;
    const bool bf16 = sb == 2;
    if (T < 1 || !states || !pre || !g_final || !g_x0 || !g_wp || !g_w1 || !g_b1 || !g_w2 || !g_b2 || !g_w3 || !workspace)
        return fail(NCAHIP_EINVAL, "cond grow bwd: null pointer or T < 1");
    // fp32 history: C <= 32 (16 < C <= 32 on the front + matrix kernels); bf16 history: C <= 20 (as the bf16 forward)
    if (int rc = check_cond(states, g_x0, pre, goal_v, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch, alive_ch, bf16 ? kMaxCCondFwdBf16 : kMaxCCondFwd)) return rc;
    if (goal_ch > 0 && !g_goal) return fail(NCAHIP_EINVAL, "cond grow bwd: g_goal required when goal_ch > 0");
    const uintptr_t amask = bf16 ? 7 : 15;   // state-type tensors: 4-cell groups (16 bytes fp32, 8 bytes bf16)
    if (W % 4 != 0 || (((uintptr_t)states | (uintptr_t)goal_v) & amask) != 0 ||
        ((uintptr_t)g_final | (uintptr_t)g_x0 | (uintptr_t)workspace) % 16 != 0)
        return fail(NCAHIP_ERANGE, "cond grow bwd: needs W %% 4 == 0 and 16-byte aligned buffers (8-byte for bf16 states / goal)");
    if ((size_t)H * W >= ((size_t)1 << 24) || (size_t)(C > 16 ? C : 16) * H * W * 4 >= ((size_t)1 << 32))
        return fail(NCAHIP_ERANGE, "cond grow bwd: grid too large for the tile kernels' 32-bit addressing (H*W < 2^24)");
    if (workspace_bytes < ncahip_cond_grow_bwd_workspace(B, C, H, W, hidden))
        return fail(NCAHIP_EINVAL, "cond grow bwd: workspace too small");
    const bool ubits = u_is_bits(u, seed);
    if (int rc = check_bits(ubits, B, H, W, fire_rate, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t slot = (size_t)B * C * H * W, pslot = (size_t)B * H * W, nb = slot * sizeof(float);
    if (bf16 && (slot * 2) % 8 != 0) return fail(NCAHIP_ERANGE, "cond grow bwd (bf16): state slots must stay 8-byte aligned");
    char* p = (char*)workspace;
    float* gbuf[2] = {(float*)p, (float*)(p + align256(nb))};
    p += 2 * align256(nb);
    float* gx = (float*)p; p += align256(nb);
    float* zbuf = (float*)p; p += align256(nb);
    float* dP = (float*)p; p += align256(3 * nb);
    const int nslab = nca_cond_bwd_nslab(), sf = nca_cond_bwd_slab_floats(C, hidden), nblk = nca_cond_bwd_nblk(B, C, H, W);
    float* slabs = (float*)p;
    float* red = slabs + (size_t)nslab * sf;  // one extra slab: the reduced gradients
    p += align256((size_t)(nslab + 1) * sf * sizeof(float));
    float* wpp = (float*)p; p += align256((size_t)nblk * 27 * sizeof(float));
    void* pscr = p; p += align256(nca_cond_bwd_fm_pscr_bytes(B, C, H, W));   // front kernel -> matrix kernel scratch (operand order)
    void* doscr = p; p += align256(nca_cond_bwd_fm_doscr_bytes(B, C, H, W));
    float* opimg = (float*)p;   // the matrix kernel's operand image, built once per call (the weights do not change between the steps)
    hipError_t e = hipMemsetAsync(slabs, 0, (size_t)nslab * sf * sizeof(float), st);
    if (e == hipSuccess) e = hipMemsetAsync(wpp, 0, (size_t)nblk * 27 * sizeof(float), st);
    if (e == hipSuccess && goal_ch > 0) e = hipMemsetAsync(g_goal, 0, (size_t)B * goal_ch * H * W * sizeof(float), st);
    if (e != hipSuccess) return hip_result(e, "cond grow bwd memset");
    const float* gcur = g_final;
    auto step_args = [&](int t, const float* g_in) {
        NcaCondBwdArgs ba{};
        ba.f = NcaCondArgs{reinterpret_cast<const float*>(states + (size_t)t * slot * sb), t == 0 ? nullptr : pre + (size_t)t * pslot,
                           nullptr, nullptr, reinterpret_cast<const float*>(goal_v),
                           u_at(u, ubits, t, pslot), wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, goal_ch,
                           alive_ch, alive_thr, fire_rate, clamp_lo, clamp_hi, seed, step0 + (uint64_t)t, nullptr};
        ba.f.u_bits = ubits;
        ba.x_next = reinterpret_cast<const float*>(states + (size_t)(t + 1) * slot * sb);
        ba.pre_t = pre + (size_t)(t + 1) * pslot;
        ba.g_next = g_in;
        ba.g_out = t == 0 ? g_x0 : gbuf[t & 1];
        ba.gx = gx; ba.dP = dP; ba.zbuf = zbuf; ba.dgoal = g_goal; ba.slabs = slabs; ba.wp_partials = wpp;
        ba.nslab = nslab; ba.nblk = nblk;
        ba.pscr = pscr; ba.doscr = doscr;
        ba.opimg = opimg;
        return ba;
    };
    // The weights do not change between the steps: in the front + matrix form ONE workgroup builds the matrix kernel's operand image
    // (opmode 1, no tile work) and the T step launches copy it (opmode 2) instead of gathering it again each (~6 us per launch).
    int opmode = 0;
    if (T >= 2) {
        NcaCondBwdArgs pa = step_args(T - 1, gcur);
        if (nca_cond_bwd_is_fm(pa, bf16)) {
            pa.opmode = 1;
            if (int rc = hip_result(nca_launch_cond_step_bwd(pa, st, bf16), "cond_grow_bwd operand image")) return rc;
            opmode = 2;
        }
    }
    for (int t = T - 1; t >= 0; --t) {
        NcaCondBwdArgs ba = step_args(t, gcur);
        ba.opmode = opmode;
        if (int rc = hip_result(nca_launch_cond_step_bwd(ba, st, bf16), "cond_grow_bwd step")) return rc;
        gcur = ba.g_out;
    }
    // slabs -> one summed slab (tile-major) -> gradients in the reference layouts
    if (int rc = hip_result(nca_launch_reduce_rows(slabs, red, nslab, sf, st), "cond_grow_bwd reduce")) return rc;
    if (int rc = hip_result(nca_launch_cond_bwd_unpermute(red, C, hidden, bf16, g_w1, g_w2, g_w3, g_b1, g_b2, st), "cond_grow_bwd unpermute")) return rc;
    return hip_result(nca_launch_reduce_wp(wpp, g_wp, B, C, H, W, st), "cond_grow_bwd reduce wp");
}

int ncahip_cond_grow_bwd_f32(const float* states, const uint8_t* pre, int T, const float* goal, int goal_ch,
                             const float* u, const float* wp, const float* w1, const float* b1, const float* w2,
                             const float* b2, const float* w3, int B, int C, int H, int W, int hidden, int alive_ch,
                             float alive_thr, float fire_rate, float clamp_lo, float clamp_hi, uint64_t seed,
                             uint64_t step0, const float* g_final, float* g_x0, float* g_goal, float* g_wp, float* g_w1,
                             float* g_b1, float* g_w2, float* g_b2, float* g_w3, void* workspace, size_t workspace_bytes,
                             ncahip_stream_t stream) {
    return cond_grow_bwd_impl(states, 4, pre, T, goal, goal_ch, u, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, alive_ch, alive_thr,
                              fire_rate, clamp_lo, clamp_hi, seed, step0, g_final, g_x0, g_goal, g_wp, g_w1, g_b1, g_w2, g_b2,
                              g_w3, workspace, workspace_bytes, stream);
}

int ncahip_cond_grow_bwd_bf16(const uint16_t* states, const uint8_t* pre, int T, const uint16_t* goal, int goal_ch,
                              const float* u, const float* wp, const float* w1, const float* b1, const float* w2,
                              const float* b2, const float* w3, int B, int C, int H, int W, int hidden, int alive_ch,
                              float alive_thr, float fire_rate, float clamp_lo, float clamp_hi, uint64_t seed,
                              uint64_t step0, const float* g_final, float* g_x0, float* g_goal, float* g_wp, float* g_w1,
                              float* g_b1, float* g_w2, float* g_b2, float* g_w3, void* workspace, size_t workspace_bytes,
                              ncahip_stream_t stream) {
    return cond_grow_bwd_impl(states, 2, pre, T, goal, goal_ch, u, wp, w1, b1, w2, b2, w3, B, C, H, W, hidden, alive_ch, alive_thr,
                              fire_rate, clamp_lo, clamp_hi, seed, step0, g_final, g_x0, g_goal, g_wp, g_w1, g_b1, g_w2, g_b2,
                              g_w3, workspace, workspace_bytes, stream);
}

int ncahip_pack_fire_mask_u32(const float* u, uint32_t* bits, int T, int B, int H, int W, float rate, int mode, ncahip_stream_t stream) {
    if (!u || !bits || T <= 0 || B <= 0 || H <= 0 || W <= 0 || (mode != 0 && mode != 1)) return fail(NCAHIP_EINVAL, "pack_fire_mask: bad argument");
    if (int rc = check_bits(true, B, H, W, rate, mode == 1)) return rc;
    return hip_result(nca_launch_pack_fire_mask(u, bits, T, (size_t)B * H * W, rate, mode, (hipStream_t)stream), "pack_fire_mask");
}

int ncahip_philox_uniform_f32(float* u, int B, int H, int W, uint64_t seed, uint64_t step, ncahip_stream_t stream) {
    if (!u || B <= 0 || H <= 0 || W <= 0) return fail(NCAHIP_EINVAL, "philox_uniform: bad argument");
    return hip_result(nca_launch_philox_uniform(u, B, H, W, seed, step, (hipStream_t)stream), "philox_uniform");
}

}  // extern "C"
