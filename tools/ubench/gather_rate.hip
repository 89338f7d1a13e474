// Micro-benchmark (debug aid): what the memory system sustains when every lane reads its own small record at an
// unrelated address, the access pattern of a closest-hit walk over a tree far larger than the caches.
// Every lane walks a pseudo-random sequence of record indices (a multiplicative generator, no two lanes alike) and
// reads REC bytes per step as 16-byte loads.  DEP = 1 makes the next index depend on the data read (one read in flight
// per lane, like a node walk); DEP = 0 leaves the reads independent.
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate [table MiB] [waves per SIMD]
// Under `rocprofv3 --pmc FETCH_SIZE` (or TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum) the bytes the fabric delivered per read follow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int REC, int DEP>
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ table, unsigned long long n_recs, unsigned iters, unsigned* out) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long x = 0x9E3779B97F4A7C15ull * (tid + 1u);
    unsigned acc = 0;
    for (unsigned it = 0; it < iters; ++it) {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        const unsigned long long idx = ((x >> 20) + (DEP ? (acc & 1u) : 0u)) % n_recs;
        const uint4* p = table + idx * (REC / 16);
#pragma unroll
        for (int k = 0; k < REC / 16; ++k) {
            const uint4 v = p[k];
            acc += v.x ^ v.w;
        }
    }
    out[tid] = acc;
}

template <int REC, int DEP>
void run(const uint4* table, size_t table_bytes, unsigned waves_per_simd, unsigned* out) {
    const unsigned blocks = 256u * waves_per_simd;   // 256 CUs x 4 SIMDs: one 256-thread block is one wave per SIMD of a CU
    const unsigned iters = 2048;
    const unsigned long long n_recs = table_bytes / REC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<REC, DEP>), dim3(blocks), dim3(256), 0, 0, table, n_recs, 64u, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<REC, DEP>), dim3(blocks), dim3(256), 0, 0, table, n_recs, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double reads = (double)blocks * 256.0 * iters;
    std::printf("rec %3d B %s table %6zu MiB %u waves/SIMD: %7.3f ms, %7.1f G reads/s, %7.1f GB/s useful, %7.1f GB/s if each read moves a 128-B line, %6.1f if 64 B\n",
                REC, DEP ? "dependent  " : "independent", table_bytes >> 20, waves_per_simd, ms, reads / ms * 1e-6, reads * REC / ms * 1e-6,
                reads * 128.0 / ms * 1e-6, reads * 64.0 / ms * 1e-6);
    std::fflush(stdout);
}

int main(int argc, char** argv) {
    const size_t only_mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 0;
    const unsigned only_waves = argc > 2 ? (unsigned)std::strtoul(argv[2], nullptr, 10) : 0;
    const int only_rec = argc > 3 ? std::atoi(argv[3]) : 0;
    const size_t max_bytes = 4096ull << 20;
    uint4* table = nullptr;
    unsigned* out = nullptr;
    if (hipMalloc(&table, max_bytes) != hipSuccess || hipMalloc(&out, 256u * 8u * 256u * sizeof(unsigned)) != hipSuccess) {
        std::fprintf(stderr, "hipMalloc failed\n");
        return 1;
    }
    hipMemset(table, 1, max_bytes);
    for (size_t mib : {64ull, 4096ull}) {
        if (only_mib && mib != only_mib) continue;
        for (unsigned w : {4u, 8u}) {
            if (only_waves && w != only_waves) continue;
            const size_t bytes = mib << 20;
            if (!only_rec || only_rec == 32) { run<32, 1>(table, bytes, w, out); run<32, 0>(table, bytes, w, out); }
            if (!only_rec || only_rec == 64) { run<64, 1>(table, bytes, w, out); run<64, 0>(table, bytes, w, out); }
            if (!only_rec || only_rec == 128) { run<128, 1>(table, bytes, w, out); run<128, 0>(table, bytes, w, out); }
            if (!only_rec || only_rec == 16) { run<16, 1>(table, bytes, w, out); }
        }
    }
    hipFree(table);
    hipFree(out);
    return 0;
}
