// nca_stencil.hip -- HBM-bound kernels: standalone perception stencils, pending-state finalize,
// alive mask, Philox uniforms, MFMA lane-map self-test.  gfx950 only.
//
// Perception: one thread produces 4 W-contiguous cells of one (b, channel) plane over R consecutive rows, marching down with a
// three-row window in registers: one 16-byte row load per row (pad mode resolved per row), left/right neighbours taken from the
// adjacent lanes by wavefront shuffle (the row edges and the wave edges fall back to one scalar load that hits L1/L2), 16-byte
// NONTEMPORAL stores of each output plane (written once, never re-read by the kernel).  Algorithmic traffic: read C, write 4C (or
// 3C) floats per cell -- 20*C (16*C) bytes/cell; that figure over the launch time is what bench.py prices against the 8 TB/s HBM
// roof.
#include "nca_common.h"
#include "nca_kernels.h"

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Loads rows y-1,y,y+1 of one plane at columns 4*x4-1 .. 4*x4+4 into nb[3][6] (pad-resolved).
__device__ __forceinline__ void load_rows_vec(const float* plane, int H, int W, int y, int x4, int pad,
                                              bool lane_has_left, bool lane_has_right, float (&nb)[3][6]) {
    const int xl = nca_pad_index(4 * x4 - 1, W, pad), xr = nca_pad_index(4 * x4 + 4, W, pad);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int sy = nca_pad_index(y + dy - 1, H, pad);
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* row = plane + (size_t)(sy < 0 ? 0 : sy) * W;
        if (sy >= 0) c = ld4(row + 4 * x4);
        // neighbours from the adjacent lanes (same row when lane_has_* holds)
        float l = __shfl_up(c.w, 1), r = __shfl_down(c.x, 1);
        if (!lane_has_left) l = (sy >= 0 && xl >= 0) ? row[xl] : 0.0f;
        if (!lane_has_right) r = (sy >= 0 && xr >= 0) ? row[xr] : 0.0f;
        nb[dy][0] = l; nb[dy][1] = c.x; nb[dy][2] = c.y; nb[dy][3] = c.z; nb[dy][4] = c.w; nb[dy][5] = r;
    }
}

// Scalar form (W not a multiple of 4, or unaligned pointers): one thread per cell.
__global__ __launch_bounds__(256) void dynca_perceive_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                                    int H, int W, int pad) {
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)B * C * plane;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= total) return;
    const int xx = (int)(id % W), yy = (int)((id / W) % H);
    const int c = (int)((id / plane) % C), b = (int)(id / (plane * C));
    const float* const p = x + ((size_t)b * C + c) * plane;
    float a[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int sy = nca_pad_index(yy + dy - 1, H, pad), sx = nca_pad_index(xx + dx - 1, W, pad);
            a[dy][dx] = (sy >= 0 && sx >= 0) ? p[(size_t)sy * W + sx] : 0.0f;
        }
    float* const yb = y + (size_t)b * 4 * C * plane + (size_t)yy * W + xx;  // blocked order, dynca.py:92-95
    __builtin_nontemporal_store(a[1][1], yb + (size_t)c * plane);
    __builtin_nontemporal_store(nca_sobel_x(a), yb + (size_t)(C + c) * plane);
    __builtin_nontemporal_store(nca_sobel_y(a), yb + (size_t)(2 * C + c) * plane);
    __builtin_nontemporal_store(nca_laplacian(a), yb + (size_t)(3 * C + c) * plane);
}

// vec4 path, rolling rows: one thread owns 4 W-contiguous cells of R consecutive rows and marches down them with a
// three-row window in registers, so each state row is loaded once per thread (R + 2 row loads per R rows produced instead of 3 R) and the
// halo rows shared with the chunks above and below are the only re-reads (R = 1 is the plain one-row form, used when the launch is too
// small for R = 8 to fill the chip).  The four output planes are written once and not read again by this kernel: nontemporal stores
// keep them from displacing the state rows in L2 / MALL -- at B = 64 (537 MB written per launch) that alone is 5.6 -> 6.5 TB/s.
typedef float nca_f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load_row6(const float* plane, int H, int W, int y, int x4, int xl, int xr, int pad, bool has_left,
                                          bool has_right, float (&o)[6]) {
    const int sy = nca_pad_index(y, H, pad);
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* row = plane + (size_t)(sy < 0 ? 0 : sy) * W;
    if (sy >= 0) c = ld4(row + 4 * x4);
    float l = __shfl_up(c.w, 1), r = __shfl_down(c.x, 1);
    if (!has_left) l = (sy >= 0 && xl >= 0) ? row[xl] : 0.0f;
    if (!has_right) r = (sy >= 0 && xr >= 0) ? row[xr] : 0.0f;
    o[0] = l; o[1] = c.x; o[2] = c.y; o[3] = c.z; o[4] = c.w; o[5] = r;
}

template <int R>
__global__ __launch_bounds__(256) void dynca_perceive_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C,
                                                                  int H, int W, int pad) {
    const size_t plane = (size_t)H * W;
    const int W4 = W >> 2, HC = (H + R - 1) / R;
    const size_t total = (size_t)B * C * HC * W4;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = gid < total;
    const size_t id = active ? gid : total - 1;  // keep every lane in the shuffles
    const int x4 = (int)(id % W4);
    const int yc = (int)((id / W4) % HC);
    const int c = (int)((id / ((size_t)W4 * HC)) % C);
    const int b = (int)(id / ((size_t)W4 * HC * C));
    const int lane = threadIdx.x & 63;
    const bool hl = x4 > 0 && lane > 0, hr = x4 < W4 - 1 && lane < 63 && gid + 1 < total;
    const int xl = nca_pad_index(4 * x4 - 1, W, pad), xr = nca_pad_index(4 * x4 + 4, W, pad);
    const float* const p = x + ((size_t)b * C + c) * plane;
    float* const yb = y + (size_t)b * 4 * C * plane + 4 * x4;
    const int y0 = yc * R;
    float nb[3][6];
    load_row6(p, H, W, y0 - 1, x4, xl, xr, pad, hl, hr, nb[0]);
    load_row6(p, H, W, y0, x4, xl, xr, pad, hl, hr, nb[1]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int yy = y0 + r;
        load_row6(p, H, W, yy + 1, x4, xl, xr, pad, hl, hr, nb[2]);
        if (active && yy < H) {
            nca_f4v o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a[3][3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) a[dy][dx] = nb[dy][j + dx];
                o[0][j] = a[1][1];
                o[1][j] = nca_sobel_x(a);
                o[2][j] = nca_sobel_y(a);
                o[3][j] = nca_laplacian(a);
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {  // blocked order, dynca.py:92-95
                nca_f4v* const dst = reinterpret_cast<nca_f4v*>(yb + (size_t)(f * C + c) * plane + (size_t)yy * W);
                __builtin_nontemporal_store(o[f], dst);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { nb[0][k] = nb[1][k]; nb[1][k] = nb[2][k]; }
    }
}

// Coarse level of the two-scale perception (dynca.py:75-100 with scale = 1, H and W even): the state is bilinearly reduced by
// 2 -- for even sizes F.interpolate(align_corners=False) samples at 2i + 0.5, i.e. the 2x2 mean with lambdas 0.5 -- and the
// fixed filters run on the coarse grid with F.pad(mode) resolved THERE.  One thread = one coarse cell of one (b, c) plane;
// the 36 fine values it touches are 8-byte loads that hit L1/L2 (neighbouring threads share them).  Output: pc [B,4C,H/2,W/2]
// in the blocked order [xc | Sx*xc | Sy*xc | L*xc]; the fused multi-scale step up-samples it on the fly.
__global__ __launch_bounds__(256) void dynca_coarse_perceive_kernel(const float* __restrict__ x, float* __restrict__ pc, int B, int C,
                                                                    int H, int W, int pad) {
    const int Hc = H >> 1, Wc = W >> 1;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * C * Hc * Wc) return;
    const int xc = (int)(id % Wc), yc = (int)((id / Wc) % Hc), c = (int)((id / ((size_t)Wc * Hc)) % C), b = (int)(id / ((size_t)Wc * Hc * C));
    const float* const pl = x + ((size_t)b * C + c) * (size_t)H * W;
    float a[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int sy = nca_pad_index(yc + dy - 1, Hc, pad);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int sx = nca_pad_index(xc + dx - 1, Wc, pad);
            float v = 0.0f;
            if (sy >= 0 && sx >= 0) {
                const float2 r0 = *reinterpret_cast<const float2*>(pl + (size_t)(2 * sy) * W + 2 * sx);
                const float2 r1 = *reinterpret_cast<const float2*>(pl + (size_t)(2 * sy + 1) * W + 2 * sx);
                v = 0.5f * (0.5f * r0.x + 0.5f * r0.y) + 0.5f * (0.5f * r1.x + 0.5f * r1.y);   // upsample_bilinear2d's own expression
            }
            a[dy][dx] = v;
        }
    }
    const size_t cp = (size_t)Hc * Wc;
    float* const o = pc + ((size_t)b * 4 * C + c) * cp + (size_t)yc * Wc + xc;
    o[0] = a[1][1];
    o[(size_t)C * cp] = nca_sobel_x(a);
    o[(size_t)2 * C * cp] = nca_sobel_y(a);
    o[(size_t)3 * C * cp] = nca_laplacian(a);
}

// bf16 -> f32 (exact), 4 elements per thread: the DyNCA backward over a bf16 history widens x_t into scratch and then runs fp32
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint16_t* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n4) return;
    const uint2 v = reinterpret_cast<const uint2*>(in)[id];
    reinterpret_cast<float4*>(out)[id] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                                                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}

// Two-scale perception, pieces of the backward pass (forward: dynca_coarse_perceive_kernel + the fused step).
// bilinear x2 up-sampling weights of fine index y (align_corners = False, even sizes): rows (r0, r1) with lambdas (1 - l1, l1)
__device__ __forceinline__ void nca_up2_taps(int y, int nc, int& r0, int& r1, float& l1) {
    const int k = y >> 1;
    if (y & 1) { r0 = k; r1 = min(k + 1, nc - 1); l1 = 0.25f; }
    else { r0 = max(k - 1, 0); r1 = k; l1 = 0.75f; }
}
// y <- (y + up2(pc)) / 2 in place: the two-scale perception [B,4C,H,W] the layer-1 weight-gradient product needs (dynca.py:98-110)
__global__ __launch_bounds__(256) void dynca_ms_combine_kernel(float* __restrict__ y, const float* __restrict__ pc, int B, int C4, int H, int W) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * C4 * H * W) return;
    const int px = (int)(id % W), py = (int)((id / W) % H);
    const size_t pl = id / ((size_t)H * W);
    const int Hc = H >> 1, Wc = W >> 1;
    int r0, r1, c0, c1;
    float h1, w1;
    nca_up2_taps(py, Hc, r0, r1, h1);
    nca_up2_taps(px, Wc, c0, c1, w1);
    const float* const q = pc + pl * (size_t)Hc * Wc;
    const float h0 = 1.0f - h1, w0 = 1.0f - w1;
    const float up = h0 * (w0 * q[(size_t)r0 * Wc + c0] + w1 * q[(size_t)r0 * Wc + c1]) + h1 * (w0 * q[(size_t)r1 * Wc + c0] + w1 * q[(size_t)r1 * Wc + c1]);
    y[id] = (y[id] + up) / 2.0f;
}
// dpc = 0.5 * up2^T(dy): gradient of the coarse-level perception [B,4C,H/2,W/2] from dL/d(two-scale perception) [B,4C,H,W].
// Gather form: coarse cell k receives from the fine cells 2k-1 .. 2k+2 of each axis whatever weight their taps put on k
// (border clamping included, so the weights of a fine row always sum to 1 over the coarse rows).
__global__ __launch_bounds__(256) void dynca_ms_upT_kernel(const float* __restrict__ dy, float* __restrict__ dpc, int B, int C4, int H, int W) {
    const int Hc = H >> 1, Wc = W >> 1;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * C4 * Hc * Wc) return;
    const int kx = (int)(id % Wc), ky = (int)((id / Wc) % Hc);
    const size_t pl = id / ((size_t)Hc * Wc);
    float wy[4], wx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = 2 * ky - 1 + i, x = 2 * kx - 1 + i;
        int r0, r1;
        float l1;
        wy[i] = 0.0f;
        wx[i] = 0.0f;
        if (y >= 0 && y < H) { nca_up2_taps(y, Hc, r0, r1, l1); wy[i] = (r0 == ky ? 1.0f - l1 : 0.0f) + (r1 == ky ? l1 : 0.0f); }
        if (x >= 0 && x < W) { nca_up2_taps(x, Wc, r0, r1, l1); wx[i] = (r0 == kx ? 1.0f - l1 : 0.0f) + (r1 == kx ? l1 : 0.0f); }
    }
    const float* const src = dy + pl * (size_t)H * W;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = min(max(2 * ky - 1 + i, 0), H - 1);
        float row = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) row = fmaf(wx[j], src[(size_t)y * W + min(max(2 * kx - 1 + j, 0), W - 1)], row);
        acc = fmaf(wy[i], row, acc);
    }
    dpc[id] = 0.5f * acc;
}

// ---- conditioning front ends (the fixed-filter part of the encoders that feed the step) ----------------------------------
// ImageEncoder (EncoderConditioning/encoder.py:37-52): gray = mean over channels; [sobel_x, sobel_y, laplacian](gray), 3x3, zero
// pad; per-channel 5x5 blur, zero pad 2.  One pass: one thread = one pixel of one batch item, every output plane of it; the
// <= 25*ch + 9 inputs are L1/L2 hits (neighbouring threads share them).  feat [B, 3+ch, H, W] = [sx | sy | lap | blur(ch)],
// the input of the learned `embed` convolutions.  k3: the three 3x3 filters [3][9]; k5: the blur taps [25] (the module's
// frozen parameters, so a loaded state_dict is honoured).  MAXCH image channels (3 or 4 in the reference's targets).
template <int MAXCH>
__global__ __launch_bounds__(256) void image_encoder_front_kernel(const float* __restrict__ img, const float* __restrict__ k3,
                                                                  const float* __restrict__ k5, float* __restrict__ feat, int B, int ch,
                                                                  int H, int W) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)H * W;
    if (id >= (size_t)B * plane) return;
    const int px = (int)(id % W), py = (int)((id / W) % H), b = (int)(id / plane);
    const float* const ib = img + (size_t)b * ch * plane;
    float blur[MAXCH], g[3][3];
#pragma unroll
    for (int c = 0; c < MAXCH; ++c) blur[c] = 0.0f;
    const float inv = 1.0f / (float)ch;
#pragma unroll
    for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
        for (int dx = -2; dx <= 2; ++dx) {
            const int y = py + dy, x = px + dx;
            const bool in = y >= 0 && y < H && x >= 0 && x < W;
            const float kw = k5[(dy + 2) * 5 + dx + 2];
            float sum = 0.0f;
#pragma unroll
            for (int c = 0; c < MAXCH; ++c) {
                const float v = (in && c < ch) ? ib[(size_t)c * plane + (size_t)y * W + x] : 0.0f;
                blur[c] = fmaf(kw, v, blur[c]);
                sum += v;
            }
            if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1) g[dy + 1][dx + 1] = sum * inv;   // torch.mean: sum / ch
        }
    float* const ob = feat + (size_t)b * (3 + ch) * plane + (size_t)py * W + px;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(k3[f * 9 + t], g[t / 3][t % 3], acc);
        ob[(size_t)f * plane] = acc;
    }
#pragma unroll
    for (int c = 0; c < MAXCH; ++c)
        if (c < ch) ob[(size_t)(3 + c) * plane] = blur[c];
}

// EdgeExtractor (ConditioneDyNCA/models/dynca.py:182-213): [sobel_x, sobel_y, laplacian] of a 1-channel image, zero pad, then tanh
// when edge_transform == 'tanh'.  out [B,3,H,W] is the step's conditioning input.
__global__ __launch_bounds__(256) void edge_extractor_kernel(const float* __restrict__ img, const float* __restrict__ k3,
                                                             float* __restrict__ out, int B, int H, int W, int do_tanh) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)H * W;
    if (id >= (size_t)B * plane) return;
    const int px = (int)(id % W), py = (int)((id / W) % H), b = (int)(id / plane);
    const float* const ib = img + (size_t)b * plane;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int y = py + t / 3 - 1, x = px + t % 3 - 1;
        g[t] = (y >= 0 && y < H && x >= 0 && x < W) ? ib[(size_t)y * W + x] : 0.0f;
    }
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(k3[f * 9 + t], g[t], acc);
        out[((size_t)b * 3 + f) * plane + (size_t)py * W + px] = do_tanh ? tanhf(acc) : acc;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void cond_perceive_kernel(const float* __restrict__ z, const float* __restrict__ wp,
                                                            float* __restrict__ y, int B, int C, int H, int W) {
    const size_t plane = (size_t)H * W;
    constexpr int V = VEC ? 4 : 1;
    const int WV = W / V;
    const size_t total = (size_t)B * C * H * WV;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = gid < total;
    const size_t id = active ? gid : total - 1;
    const int xv = (int)(id % WV);
    const int yy = (int)((id / WV) % H);
    const int c = (int)((id / ((size_t)WV * H)) % C);
    const int b = (int)(id / ((size_t)WV * H * C));
    const float* const p = z + ((size_t)b * C + c) * plane;
    float nb[3][V + 2];
    if (VEC) {
        const int lane = threadIdx.x & 63;
        float t[3][6];
        load_rows_vec(p, H, W, yy, xv, NCA_PAD_ZERO, xv > 0 && lane > 0, xv < WV - 1 && lane < 63 && gid + 1 < total, t);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < V + 2; ++j) nb[dy][j] = t[dy][j];
    } else {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int sy = yy + dy - 1, sx = xv + dx - 1;
                nb[dy][dx] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? p[(size_t)sy * W + sx] : 0.0f;
            }
    }
    if (!active) return;
    const float* const w = wp + (size_t)c * 27;
    float* const yb = y + ((size_t)b * 3 * C + 3 * c) * plane + (size_t)yy * W + V * xv;
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        float o[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc = fmaf(w[9 * f + 3 * dy + dx], nb[dy][j + dx], acc);
            o[j] = acc;
        }
        // written once, not re-read here: nontemporal, as in the DyNCA stencil above
        if (VEC) {
            const nca_f4v v = {o[0], o[V > 1 ? 1 : 0], o[V > 2 ? 2 : 0], o[V > 3 ? 3 : 0]};
            __builtin_nontemporal_store(v, reinterpret_cast<nca_f4v*>(yb + (size_t)f * plane));
        } else {
            __builtin_nontemporal_store(o[0], yb + (size_t)f * plane);
        }
    }
}

__device__ __forceinline__ float alpha_max3x3(const float* ap, int H, int W, int y, int x) {
    float m = NCA_NEG_INF;  // max_pool2d pads with -inf (nca.py:156-161)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int sy = y + dy, sx = x + dx;
            if (sy >= 0 && sy < H && sx >= 0 && sx < W) m = fmaxf(m, ap[(size_t)sy * W + sx]);
        }
    return m;
}

__global__ __launch_bounds__(256) void cond_finalize_kernel(const float* __restrict__ x, const uint8_t* __restrict__ pre,
                                                            float* __restrict__ out, int B, int C, int H, int W,
                                                            int alive_ch, float thr, float lo, float hi) {
    const size_t plane = (size_t)H * W;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * plane) return;
    const int xx = (int)(id % W), yy = (int)((id / W) % H), b = (int)(id / plane);
    const float* const xb = x + (size_t)b * C * plane;
    float life = 1.0f;
    if (alive_ch >= 0)
        life = (pre[id] != 0 && alpha_max3x3(xb + (size_t)alive_ch * plane, H, W, yy, xx) > thr) ? 1.0f : 0.0f;
    const size_t off = (size_t)yy * W + xx;
    float* const ob = out + (size_t)b * C * plane + off;
    for (int c = 0; c < C; ++c) ob[(size_t)c * plane] = fminf(fmaxf(xb[(size_t)c * plane + off] * life, lo), hi);
}

__global__ __launch_bounds__(256) void cond_alive_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int B, int C,
                                                         int H, int W, int alive_ch, float thr) {
    const size_t plane = (size_t)H * W;
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (size_t)B * plane) return;
    const int xx = (int)(id % W), yy = (int)((id / W) % H), b = (int)(id / plane);
    out[id] = alive_ch < 0 ? 1 : (alpha_max3x3(x + ((size_t)b * C + alive_ch) * plane, H, W, yy, xx) > thr ? 1 : 0);
}

__global__ __launch_bounds__(256) void philox_uniform_kernel(float* __restrict__ u, size_t n, uint64_t seed, uint64_t step) {
    const size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) u[id] = nca_philox_cell(seed, step, id);
}

// Fire masks as bits: one wave evaluates 64 cells, the ballot IS two packed words.  Step t's words start at t * ceil(cells/32).
__global__ __launch_bounds__(256) void pack_fire_mask_kernel(const float* __restrict__ u, uint32_t* __restrict__ bits, size_t cells,
                                                            size_t words, float rate, int mode) {
    const size_t t = blockIdx.y, id = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool fire = false;
    if (id < cells) {
        const float v = u[t * cells + id];
        fire = mode == 0 ? (fminf(fmaxf(v, 0.0f), 1.0f) < rate) : (floorf(v + rate) >= 1.0f);
    }
    const unsigned long long m = __ballot(fire);
    const int lane = threadIdx.x & 63;
    const size_t w0 = id >> 5;          // lanes 0 and 32 of a wave write its two words
    if ((lane & 31) == 0 && w0 < words) bits[t * words + w0] = (uint32_t)(lane ? (m >> 32) : m);
}

// MFMA lane-map check with exact integers and an ASYMMETRIC B: D = A*B, A[i][k] = i + 16k + 1,
// B[k][j] = 3j + 7k + 2 (k = 0..3).  Every lane verifies its 4 accumulator values.
__global__ void selftest_kernel(int* result) {
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const float a = (float)(i + 16 * g + 1), bq = (float)(3 * i + 7 * g + 2);
    f32x4 d = nca_mfma(a, bq, f32x4{0.f, 0.f, 0.f, 0.f});
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r, col = i;
        float ref = 0.f;
        for (int k = 0; k < 4; ++k) ref += (float)(row + 16 * k + 1) * (float)(3 * col + 7 * k + 2);
        ok = ok && (d[r] == ref);
    }
    const unsigned long long bad = __ballot(!ok);
    if (lane == 0) result[0] = bad == 0ull ? 1 : 0;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

hipError_t nca_launch_dynca_perceive(const float* x, float* y, int B, int C, int H, int W, int pad, hipStream_t st) {
    const bool vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) % 16 == 0);
    const size_t elems = (size_t)B * C * H * W;
    if (vec && elems >= ((size_t)8 << 20))   // >= 4 waves of 8-row chunks per SIMD on 256 CUs
        hipLaunchKernelGGL(dynca_perceive_rows_kernel<8>, dim3(blocks_for((size_t)B * C * ((H + 7) / 8) * (W / 4))), dim3(256), 0, st, x, y,
                           B, C, H, W, pad);
    else if (vec)
        hipLaunchKernelGGL(dynca_perceive_rows_kernel<1>, dim3(blocks_for((size_t)B * C * H * (W / 4))), dim3(256), 0, st, x, y, B, C, H, W,
                           pad);
    else
        hipLaunchKernelGGL(dynca_perceive_scalar_kernel, dim3(blocks_for(elems)), dim3(256), 0, st, x, y, B, C, H, W, pad);
    return hipGetLastError();
}

hipError_t nca_launch_image_encoder_front(const float* img, const float* k3, const float* k5, float* feat, int B, int ch, int H, int W,
                                          hipStream_t st) {
    const size_t n = (size_t)B * H * W;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (ch <= 4) hipLaunchKernelGGL(image_encoder_front_kernel<4>, grid, dim3(256), 0, st, img, k3, k5, feat, B, ch, H, W);
    else if (ch <= 8) hipLaunchKernelGGL(image_encoder_front_kernel<8>, grid, dim3(256), 0, st, img, k3, k5, feat, B, ch, H, W);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t nca_launch_edge_extractor(const float* img, const float* k3, float* out, int B, int H, int W, int do_tanh, hipStream_t st) {
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(edge_extractor_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, img, k3, out, B, H, W, do_tanh);
    return hipGetLastError();
}

hipError_t nca_launch_widen_bf16(const uint16_t* in, float* out, size_t n, hipStream_t st) {   // n % 4 == 0, 8-byte aligned input
    hipLaunchKernelGGL(widen_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, in, out, n / 4);
    return hipGetLastError();
}
hipError_t nca_launch_dynca_ms_combine(float* y, const float* pc, int B, int C, int H, int W, hipStream_t st) {
    const size_t n = (size_t)B * 4 * C * H * W;
    hipLaunchKernelGGL(dynca_ms_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, pc, B, 4 * C, H, W);
    return hipGetLastError();
}
hipError_t nca_launch_dynca_ms_upT(const float* dy, float* dpc, int B, int C, int H, int W, hipStream_t st) {
    const size_t n = (size_t)B * 4 * C * (H / 2) * (W / 2);
    hipLaunchKernelGGL(dynca_ms_upT_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, dpc, B, 4 * C, H, W);
    return hipGetLastError();
}

hipError_t nca_launch_dynca_coarse_perceive(const float* x, float* pc, int B, int C, int H, int W, int pad, hipStream_t st) {
    const size_t n = (size_t)B * C * (H / 2) * (W / 2);
    hipLaunchKernelGGL(dynca_coarse_perceive_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, pc, B, C, H, W, pad);
    return hipGetLastError();
}

hipError_t nca_launch_cond_perceive(const float* z, const float* wp, float* y, int B, int C, int H, int W, hipStream_t st) {
    const bool vec = (W % 4 == 0) && (((uintptr_t)z | (uintptr_t)y) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(cond_perceive_kernel<true>, dim3(blocks_for((size_t)B * C * H * (W / 4))), dim3(256), 0, st,
                           z, wp, y, B, C, H, W);
    else
        hipLaunchKernelGGL(cond_perceive_kernel<false>, dim3(blocks_for((size_t)B * C * H * W)), dim3(256), 0, st, z,
                           wp, y, B, C, H, W);
    return hipGetLastError();
}

hipError_t nca_launch_cond_finalize(const float* x, const uint8_t* pre, float* out, int B, int C, int H, int W,
                                    int alive_ch, float thr, float lo, float hi, hipStream_t st) {
    hipLaunchKernelGGL(cond_finalize_kernel, dim3(blocks_for((size_t)B * H * W)), dim3(256), 0, st, x, pre, out, B, C,
                       H, W, alive_ch, thr, lo, hi);
    return hipGetLastError();
}

hipError_t nca_launch_cond_alive(const float* x, uint8_t* out, int B, int C, int H, int W, int alive_ch, float thr,
                                 hipStream_t st) {
    hipLaunchKernelGGL(cond_alive_kernel, dim3(blocks_for((size_t)B * H * W)), dim3(256), 0, st, x, out, B, C, H, W,
                       alive_ch, thr);
    return hipGetLastError();
}

hipError_t nca_launch_philox_uniform(float* u, int B, int H, int W, uint64_t seed, uint64_t step, hipStream_t st) {
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(philox_uniform_kernel, dim3(blocks_for(n)), dim3(256), 0, st, u, n, seed, step);
    return hipGetLastError();
}

hipError_t nca_launch_pack_fire_mask(const float* u, uint32_t* bits, int T, size_t cells, float rate, int mode, hipStream_t st) {
    const size_t words = (cells + 31) / 32, cover = words * 32;      // every word is written: the grid covers words * 32 lanes
    hipLaunchKernelGGL(pack_fire_mask_kernel, dim3((unsigned)((cover + 255) / 256), (unsigned)T), dim3(256), 0, st, u, bits, cells, words, rate, mode);
    return hipGetLastError();
}

hipError_t nca_launch_selftest(int* result, hipStream_t st) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, result);
    return hipGetLastError();
}
