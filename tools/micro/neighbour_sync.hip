// neighbour_sync.hip -- micro-benchmark for a persistent multi-step NCA kernel at B = 1: 256 workgroups, one
// 8 x 32 tile each (a 16 x 16 tile grid with wrap-around neighbours), T steps inside ONE launch.  Per step a workgroup
//   waits until its 8 neighbours have published step t-1 (per-tile monotonic flags in global memory, bounded polls),
//   reads their border cells (write-through / sc1-coherent accesses: the tiles live in different XCDs' L2s),
//   runs the MFMA work of one DyNCA step of the small video model (408 exact-f32 MFMAs per wave),
//   writes its own tile (write-through) and publishes step t.
// Ping-pong buffers: a neighbour finished READING buffer (t-1) % 2 before it published step t-1 ... so one condition covers both
// hazards.  Variants: compute only / sync + exchange only / both; flag store with release semantics or relaxed after a vmcnt(0).
// The exchanged values are checked (every halo word must carry the step it was written in).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/neighbour_sync.hip -o /tmp/neighbour_sync   run: /tmp/neighbour_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TILE_WORDS = 8 * 32 * 12;   // one tile's state: 8 x 32 cells x 12 channels (12 KB)
constexpr int GRID = 16;                  // 16 x 16 tiles

__global__ __launch_bounds__(256, 1) void persist_kernel(unsigned* buf0, unsigned* buf1, int* flags, int T, int n_mfma, int do_sync,
                                                         int release, unsigned* err, float* sink) {
    const int tile = blockIdx.x, ty = tile / GRID, tx = tile % GRID, tid = threadIdx.x;
    int nb[8];
    {
        int k = 0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx)
                if (dy || dx) nb[k++] = ((ty + dy + GRID) % GRID) * GRID + (tx + dx + GRID) % GRID;
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = 1.0f + tid * 1e-6f, b = 0.5f;
    unsigned bad = 0;
    for (int t = 0; t < T; ++t) {
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // a poll expired somewhere: drain
        unsigned* const src = (t & 1) ? buf1 : buf0;   // holds step t-1 (t >= 1)
        unsigned* const dst = (t & 1) ? buf0 : buf1;
        if (do_sync && t > 0) {
            if (tid < 8) {
                int spins = 0;
                while (__hip_atomic_load(flags + nb[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < t && ++spins < (1 << 16))
                    __builtin_amdgcn_s_sleep(2);
                if (spins >= (1 << 16)) atomicOr(err, 1u);   // (every workgroup leaves the step loop once this is set: bounded run time)
            }
            __syncthreads();
            // halo: 64 words from each of the 8 neighbours (a 1-cell ring of 12 channels is ~1 K words; same order of traffic)
            for (int k = tid; k < 8 * 128; k += 256) {
                const unsigned v = __hip_atomic_load(src + (size_t)nb[k >> 7] * TILE_WORDS + (k & 127), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (unsigned)t) ++bad;           // written in step t-1 as "t"
            }
        }
        for (int i = 0; i < n_mfma / 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        // own tile out: 12 words per thread, write-through at agent scope
        for (int k = tid; k < TILE_WORDS; k += 256)
            __hip_atomic_store(dst + (size_t)tile * TILE_WORDS + k, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (do_sync) {
            __builtin_amdgcn_s_waitcnt(0);             // this thread's stores have been acknowledged
            __syncthreads();
            if (tid == 0) {
                if (release) __hip_atomic_store(flags + tile, t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(flags + tile, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (bad) atomicAdd(err + 1, bad);
    sink[blockIdx.x * 256 + tid] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

int main() {
    unsigned *b0, *b1, *err;
    int* flags;
    float* sink;
    const size_t nb = (size_t)GRID * GRID * TILE_WORDS * sizeof(unsigned);
    hipMalloc(&b0, nb); hipMalloc(&b1, nb); hipMalloc(&flags, GRID * GRID * sizeof(int)); hipMalloc(&err, 8); hipMalloc(&sink, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int T = 400;
    struct V { const char* name; int n_mfma, sync, release; } vs[] = {
        {"compute only (408 MFMAs / wave / step)", 408, 0, 0},
        {"exchange only, relaxed flag after vmcnt(0)", 0, 1, 0},
        {"exchange only, release flag", 0, 1, 1},
        {"compute + exchange, relaxed flag", 408, 1, 0},
        {"compute + exchange, release flag", 408, 1, 1},
    };
    for (const V& v : vs) {
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(flags, 0, GRID * GRID * sizeof(int)); hipMemset(err, 0, 8); hipMemset(b0, 0, nb); hipMemset(b1, 0, nb);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(persist_kernel, dim3(GRID * GRID), dim3(256), 0, 0, b0, b1, flags, T, v.n_mfma, v.sync, v.release, err, sink);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned h[2];
            hipMemcpy(h, err, 8, hipMemcpyDeviceToHost);
            if (rep) printf("%-48s %7.2f us/step   poll-expired %u   stale halo words %u\n", v.name, ms * 1e3 / T, h[0], h[1]);
        }
    }
    return 0;
}
