// Store-pattern microbenchmark for the compaction write of the shade kernel.
// Writes ~the same number of bytes (9 f64 + 4 u32 arrays) with different
// placement of each wave's / block's run of surviving elements.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/store_bench tools/store_bench.hip && /tmp/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int BLOCK = 256;
struct Arrays { double* f[9]; uint32_t* u[4]; };

// mode 0: every lane stores to its own index (aligned, full waves)         -- generate-like
// mode 1: wave w stores `keep` lanes at w*keep (unaligned runs per wave)  -- current compaction
// mode 2: block b stores 4*keep consecutive elements at b*4*keep by its first lanes -- block-staged
// mode 3: like 2 but block base rounded up to a multiple of 16 elements
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_store(Arrays a, uint32_t n_rounds, uint32_t keep) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t r = 0; r < n_rounds; ++r) {
        const uint32_t blk = r * gridDim.x + blockIdx.x;
        uint32_t k;
        bool on;
        if (MODE == 0) {
            k = blk * BLOCK + threadIdx.x;
            on = true;
        } else if (MODE == 1) {
            k = (blk * 4u + wave) * keep + lane;
            on = lane < keep;
        } else if (MODE == 2) {
            k = blk * 4u * keep + threadIdx.x;
            on = threadIdx.x < 4u * keep;
        } else {
            const uint32_t span = (4u * keep + 15u) & ~15u;
            k = blk * span + threadIdx.x;
            on = threadIdx.x < 4u * keep;
        }
        if (on) {
            const double v = (double)k;
#pragma unroll
            for (int j = 0; j < 9; ++j) a.f[j][k] = v + j;
#pragma unroll
            for (int j = 0; j < 4; ++j) a.u[j][k] = k + j;
        }
    }
}

template <int MODE>
static void run(const char* name, Arrays a, uint32_t slots, uint32_t keep, uint32_t grid) {
    // total elements written ~ slots * keep / 64 (mode 0: slots)
    const uint32_t per_block_round = MODE == 0 ? BLOCK : 4u * keep;
    const uint32_t span = MODE == 3 ? ((4u * keep + 15u) & ~15u) : per_block_round;
    const uint32_t target = MODE == 0 ? slots : (uint32_t)((uint64_t)slots * keep / 64u);
    const uint32_t n_rounds = target / (grid * span);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_store<MODE><<<grid, BLOCK>>>(a, n_rounds, keep);
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k_store<MODE><<<grid, BLOCK>>>(a, n_rounds, keep);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n_rounds * grid * per_block_round * 88.0;
    std::printf("%-34s keep %2u: %7.1f us/launch  %6.1f MB  %6.2f TB/s\n", name, keep, 1e3 * ms / reps, bytes / 1e6, bytes / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const uint32_t slots = 1u << 22;
    const size_t stride_pad = argc > 1 ? (size_t)std::atoi(argv[1]) : 0;   // extra elements between arrays
    Arrays a;
    const size_t n = slots + 4096 + stride_pad;
    double* fbuf; uint32_t* ubuf;
    CK(hipMalloc(&fbuf, 9 * n * sizeof(double)));
    CK(hipMalloc(&ubuf, 4 * n * sizeof(uint32_t)));
    for (int j = 0; j < 9; ++j) a.f[j] = fbuf + j * n;
    for (int j = 0; j < 4; ++j) a.u[j] = ubuf + j * n;
    const uint32_t grid = 2048;
    std::printf("array stride %zu elements\n", n);
    run<0>("aligned full waves", a, slots, 64, grid);
    for (uint32_t keep : {40u, 39u, 32u}) {
        run<1>("per-wave runs (compaction today)", a, slots, keep, grid);
        run<2>("per-block runs", a, slots, keep, grid);
        run<3>("per-block runs, base 16-aligned", a, slots, keep, grid);
    }
    return 0;
}
