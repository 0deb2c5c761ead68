// nca_cond_wave.hip -- ConditionedNCA fused step, wave-private tiles (gfx950, fp32 exact).
//
// Same math and MFMA mapping as nca_step_fwd.hip (see its header), different work decomposition:
// every WAVE owns a 4 x 16 cell tile and a private LDS region (state tile +1 halo, resolved-state
// copy for the residual, alpha/life/pre masks); the 8 waves of a workgroup share only the
// A-operand weight image.  There is no workgroup barrier inside the tile loop -- LDS hand-offs are
// between lanes of one wave (in-order DS queue; only compiler ordering is needed) -- so the two
// waves resident on a SIMD drift out of phase and one wave's loads / mask resolution / perception
// VALU overlap the other's MFMA chain.  Index math is shifts of the lane id; tiles whose 3-cell
// halo lies inside the image (85 % at 256^2) skip every bounds check.
//
// Requires W % 4 == 0 and 16-byte aligned tensors (16-byte row loads); other shapes take the
// generic kernel in nca_step_fwd.hip.
#include "nca_cond_tile.h"

namespace {

template <int CP>
__global__ __launch_bounds__(kThreadsW, 2) void cond_step_fwd_wave_kernel(const NcaCondArgs a) {
    using K = WCfg<CP>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, H = a.H, W = a.W, hid = a.hidden, K1 = 3 * C;

    // ---- tile walk of this wave: super-tiles XCD-chunked, wave -> (row-of-4, column-half) ------
    const int st_x = (W + STW - 1) / STW, st_y = (H + STH - 1) / STH;
    const int halo = a.alive_ch >= 0 ? 3 : 1;
    NcaTileWalk tw = nca_tile_walk(a.B * st_x * st_y);
    auto next_tile = [&]() -> WTile {
        WTile t{0, 0, 0, false, false};
        while (tw.t < tw.end) {
            const int sxi = tw.t % st_x, syi = (tw.t / st_x) % st_y;
            t.b = tw.t / (st_x * st_y);
            t.ty0 = syi * STH + (wave >> 1) * WTH;
            t.tx0 = sxi * STW + (wave & 1) * WTW;
            tw.t += tw.stride;
            if (t.ty0 < H && t.tx0 < W) {
                t.valid = true;
                t.inner = t.ty0 >= halo && t.ty0 + WTH + halo <= H && t.tx0 >= halo && t.tx0 + WTW + halo <= W;
                break;
            }
        }
        return t;
    };
    WTile cur = next_tile();
    TileRegs<CP> R;
    if (cur.valid) issue_loads<CP, true, false>(a, cur, lane, R);  // first tile's loads fly while the weight image is built

    // 16-byte A-operand images: element ((tile*S4 + s4)*64 + lane)*4 + j holds the weight of k-step s = 4*s4 + j
    fill_image_w<4 * K::K1S4 * 256>(smem + K::OFF_W1, a.w1, tid, [&](int idx) -> long {
        const int j = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) % K::K1S4, m = (idx >> 8) / K::K1S4;
        const int s = 4 * q + j, gg = l >> 4, o = 16 * m + (l & 15);
        const int ch = 4 * (s / 3) + gg, f = s % 3;  // k-step s = 3c'+f : channel 4c'+g, filter f
        return (s < K::K1S && ch < C && o < hid) ? (long)o * K1 + 3 * ch + f : -1;  // out[3c+f], nca.py:99-107
    });
    fill_image_w<4 * 16 * 64>(smem + K::OFF_W2, a.w2, tid, [&](int idx) -> long {
        const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m2 = idx >> 10;   // [m2][m][lane][r]
        const int gg = l >> 4, o = 16 * m2 + (l & 15);
        const int k = 16 * m + 4 * gg + r;                                                // k-step 4m+r of tile m2
        return (o < hid && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image_w<K::M3T * 16 * 64>(smem + K::OFF_W3, a.w3, tid, [&](int idx) -> long {
        const int r = idx & 3, l = (idx >> 2) & 63, m = (idx >> 8) & 3, m3 = idx >> 10;   // [m3][m][lane][r]
        const int gg = l >> 4, o = 16 * m3 + (l & 15);
        const int k = 16 * m + 4 * gg + r;
        return (o < C && k < hid) ? (long)o * hid + k : -1;
    });
    fill_image_w<K::HID>(smem + K::OFF_B1, a.b1, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image_w<K::HID>(smem + K::OFF_B2, a.b2, tid, [&](int idx) -> long { return idx < hid ? idx : -1; });
    fill_image_w<CP * K::WPS>(smem + K::OFF_WP, a.wp, tid, [&](int idx) -> long {
        const int ch = idx / K::WPS, j = idx % K::WPS;
        return (ch < C && j < 27) ? (long)ch * 27 + j : -1;  // [3c+f][3][3] == [c][f*9+tap]
    });
    __syncthreads();  // the only workgroup barrier: weight image complete

    float* const PWR = smem + K::SHARED + wave * K::PW;
    const TileLds L = wave_private_lds<CP>(PWR);
    int tile_no = 0;
    while (cur.valid) {
        // issue priority: the non-MFMA phases of a wave outrank its SIMD partner's MFMA chain (which only needs
        // one issue slot per 32 cycles), so staging finishes quickly and the pipe stays fed
        __builtin_amdgcn_s_setprio(3);
        NCA_STAMP(0);
        issue_loads<CP, false, true>(a, cur, lane, R);  // goal encoding: consumed last in staging (S4)
        if (cur.inner) stage_tile<CP, false>(a, cur, L, lane, R, tile_no);
        else stage_tile<CP, true>(a, cur, L, lane, R, tile_no);
        NCA_STAMP(1);
        const WTile nxt = next_tile();
        constexpr int NT = 2;  // rows per MFMA pass: 2 independent accumulator chains already pace the pipe
#pragma unroll 1
        for (int pass = 0; pass < WTH / NT; ++pass) {
            float P[NT][K::K1S];
            perceive_tile<CP, NT>(smem, L.Z, lane, pass * NT, P);
            if (pass == 0) {
                NCA_STAMP(2);
                if (nxt.valid) issue_loads<CP, true, false>(a, nxt, lane, R);  // in flight across this tile's MFMA chains
                NCA_STAMP(3);
            }
            __builtin_amdgcn_s_setprio(0);
            mlp_tile<CP, NT>(a, smem, L.XR, L.MK, lane, pass * NT, P);
            __builtin_amdgcn_s_setprio(3);
        }
        // The prefetch was issued a whole MFMA chain ago; telling the compiler so (an s_waitcnt it can see) keeps it
        // from draining the goal loads of the next tile at that tile's first use of a prefetched register.
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
        NCA_STAMP(4);
        if (cur.inner) store_tile<CP, false>(a, cur, L.XR, lane);
        else store_tile<CP, true>(a, cur, L.XR, lane);
        NCA_STAMP(5);
        cur = nxt;
        ++tile_no;
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int CP>
hipError_t launch_cond_wave(const NcaCondArgs& a, hipStream_t st) {
    using K = WCfg<CP>;
    static_assert(K::kFits8, "LDS budget (8-wave carve)");
    auto kern = cond_step_fwd_wave_kernel<CP>;
    const size_t lds = (size_t)K::LDS_FLOATS * sizeof(float);
    static NcaLdsAttr attr;   // per instantiation; keyed by device inside
    if (hipError_t e = attr.ensure(reinterpret_cast<const void*>(kern), lds); e != hipSuccess) return e;
    const int cus = nca_cu_count();
    const int nst = a.B * ((a.W + STW - 1) / STW) * ((a.H + STH - 1) / STH);
    hipLaunchKernelGGL(kern, dim3(nst < cus ? nst : cus), dim3(kThreadsW), lds, st, a);
    return hipGetLastError();
}

}  // namespace

static unsigned long long* g_stamp_buffer = nullptr;
void nca_debug_set_stamp_buffer(unsigned long long* p) { g_stamp_buffer = p; }
extern "C" void nca_debug_set_stamp_buffer_c(void* p) { g_stamp_buffer = (unsigned long long*)p; }

// W % 4 == 0 and 16-byte aligned x_in / goal: caller (nca_step_fwd.hip dispatch) guarantees it.
hipError_t nca_launch_cond_step_fwd_wave(const NcaCondArgs& a_in, hipStream_t st) {
    NcaCondArgs a = a_in;
    a.dbg = g_stamp_buffer;
    if (a.C <= 12) return launch_cond_wave<12>(a, st);
    if (a.C <= 16) return launch_cond_wave<16>(a, st);
    return hipErrorInvalidValue;
}
