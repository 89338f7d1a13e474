// Micro-benchmark (debug aid): how many vector-ALU instructions of one kind a SIMD of this GPU retires per cycle, by waves per SIMD.
// Each wave runs LOOPS x 64 independent instructions of the kind (8 accumulators, unrolled) and reports cycles from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int LOOPS = 4096;
#ifndef PACKED
#define PACKED 0   // 1 when built without -fno-slp-vectorize: the compiler then pairs the f32 FMAs into v_pk_fma_f32
#endif

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cycles, long long* ticks, float seed) {
    float a[8];
    double d[8];
    int n[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i; d[i] = seed + i; n[i] = (int)seed + i; }
    const float m = seed * 0.5f, c = seed * 0.25f;
    const double dm = m, dc = c;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
                if (KIND == 1) d[i] = __builtin_fma(d[i], dm, dc);
                if (KIND == 2) a[i] = __builtin_fminf(__builtin_fminf(a[i], m), c + a[(i + 1) & 7]);   // v_min3_f32 + v_add
                if (KIND == 3) n[i] = n[i] * 3 + (n[(i + 1) & 7] ^ it);
                if (KIND == 4) d[i] = d[i] * dm;
                if (KIND == 5) d[i] = d[i] + dc;
            }
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i] + (float)n[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
        ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = w1 - w0;   // 100 MHz
    }
}

template <int KIND>
void run(const char* name, double insts_per_step) {
    for (int waves_per_simd : {1, 2, 4, 5, 8}) {
        const int blocks = 256 * waves_per_simd;   // 256 CUs x waves_per_simd blocks of 4 waves
        float* out;
        long long *cyc, *tck;
        hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
        hipMalloc(&cyc, (size_t)blocks * 4 * sizeof(long long));
        hipMalloc(&tck, (size_t)blocks * 4 * sizeof(long long));
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, tck, 1.5f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_rate<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, tck, 1.5f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h((size_t)blocks * 4), hw((size_t)blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        hipMemcpy(hw.data(), tck, hw.size() * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0, avg_ticks = 0;
        for (long long v : h) avg += (double)v;
        for (long long v : hw) avg_ticks += (double)v;
        avg /= (double)h.size();
        avg_ticks /= (double)hw.size();
        const double ghz = avg / (avg_ticks * 10.0);                  // shader cycles per ns
        const double insts = (double)LOOPS * 64.0 * insts_per_step;   // per wave
        const double kernel_cycles = ms * 1e6 * ghz;
        printf("%-22s %d waves per SIMD: %.2f cycles per instruction seen by a wave, %.3f instructions per cycle and SIMD over the launch (%.3f ms at %.2f GHz)\n",
               name, waves_per_simd, avg / insts, insts * waves_per_simd / kernel_cycles, ms, ghz);
        hipFree(out);
        hipFree(cyc);
        hipFree(tck);
    }
}

int main() {
    run<0>(PACKED ? "v_pk_fma_f32" : "v_fma_f32", PACKED ? 0.5 : 1);
    run<1>("v_fma_f64", 1);
    run<4>("v_mul_f64", 1);
    run<5>("v_add_f64", 1);
    run<2>("v_min3_f32+v_add_f32", 2);
    run<3>("int mul/xor/add", 3);
    return 0;
}
