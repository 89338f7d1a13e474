// rtx_scan.h -- block-wide and device-wide inclusive scans for any associative
// operator (segmented min/max of boxes in the BVH builder, prefix sums of
// flags and of text lengths).  Three launches per scan: tile totals, exclusive
// scan of the totals (one block per row), tiles again with their carry.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace rtx {
namespace scan {

constexpr uint32_t BLOCK = 256;
constexpr uint32_t WAVES = BLOCK / 64;
constexpr uint32_t ITEMS = 4;
constexpr uint32_t TILE = BLOCK * ITEMS;

__device__ __forceinline__ uint32_t lane() { return threadIdx.x & 63u; }

// shfl_up_t(x, d): x of the lane d below.  Overloads for other element types live next to the type.
__device__ __forceinline__ uint32_t shfl_up_t(const uint32_t& x, uint32_t d) { return __shfl_up(x, d); }
__device__ __forceinline__ uint4 shfl_up_t(const uint4& x, uint32_t d) {
    return make_uint4(__shfl_up(x.x, d), __shfl_up(x.y, d), __shfl_up(x.z, d), __shfl_up(x.w, d));
}

// Inclusive scan of BLOCK*N values held N per thread, seeded with `carry`.
// On return x[] holds the inclusive results, `total` = carry combined with the whole tile.
template <uint32_t N, class Op, class T>
__device__ __forceinline__ void block_scan(const Op& op, T (&x)[N], const T& carry, T* lds, T& total) {
#pragma unroll
    for (uint32_t i = 1; i < N; ++i) x[i] = op.combine(x[i - 1], x[i]);
    T inc = x[N - 1];
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const T up = shfl_up_t(inc, d);
        if (lane() >= d) inc = op.combine(up, inc);
    }
    const uint32_t wave = threadIdx.x >> 6;
    if (lane() == 63) lds[wave] = inc;
    __syncthreads();
    T prefix = carry;
    total = carry;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) {
        if (w < wave) prefix = op.combine(prefix, lds[w]);
        total = op.combine(total, lds[w]);
    }
    const T before = shfl_up_t(inc, 1);
    if (lane() > 0) prefix = op.combine(prefix, before);
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) x[i] = op.combine(prefix, x[i]);
    __syncthreads();
}

// The same scan in either direction.  REV: the scan runs from the block's last element to its first (thread BLOCK-1 before
// thread 0, x[N-1] before x[0]); `carry` is what lies beyond the block on the side the scan comes from.
__device__ __forceinline__ uint32_t shfl_down_t(const uint32_t& x, uint32_t d) { return __shfl_down(x, d); }
template <bool REV, uint32_t N, class Op, class T>
__device__ __forceinline__ void block_scan_dir(const Op& op, T (&x)[N], const T& carry, T* lds, T& total) {
    if (!REV) {
#pragma unroll
        for (uint32_t i = 1; i < N; ++i) x[i] = op.combine(x[i - 1], x[i]);
    } else {
#pragma unroll
        for (uint32_t i = N - 1; i > 0; --i) x[i - 1] = op.combine(x[i], x[i - 1]);
    }
    T inc = REV ? x[0] : x[N - 1];
    const uint32_t rank = REV ? 63u - lane() : lane();   // position of the lane in scan order
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const T prev = REV ? shfl_down_t(inc, d) : shfl_up_t(inc, d);
        if (rank >= d) inc = op.combine(prev, inc);
    }
    const uint32_t wave = threadIdx.x >> 6, wrank = REV ? WAVES - 1u - wave : wave;
    if (rank == 63) lds[wrank] = inc;
    __syncthreads();
    T prefix = carry;
    total = carry;
#pragma unroll
    for (uint32_t w = 0; w < WAVES; ++w) {
        if (w < wrank) prefix = op.combine(prefix, lds[w]);
        total = op.combine(total, lds[w]);
    }
    const T before = REV ? shfl_down_t(inc, 1) : shfl_up_t(inc, 1);
    if (rank > 0) prefix = op.combine(prefix, before);
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) x[i] = op.combine(prefix, x[i]);
    __syncthreads();
}

// What a segmented scan carries into tile `tile` of a row of tile totals: the fold of the totals before it (REV: after it),
// from the nearest total that holds a segment start on.  T::f != 0 marks such a total.  Whole block; the result is uniform.
template <bool REV, class Op, class T>
__device__ __forceinline__ T look_back(const Op& op, const T* row, uint32_t tile, uint32_t n_tiles, T* lds) {
    T acc = op.identity();
    const uint32_t avail = REV ? n_tiles - 1u - tile : tile;
    if (avail == 0u) return acc;
    acc = row[REV ? tile + 1u : tile - 1u];   // deep in the tree every tile holds a segment start: the neighbour is all there is
    if (acc.f) return acc;
    acc = op.identity();
    for (uint32_t done = 0; done < avail; done += BLOCK) {
        const uint32_t j = done + threadIdx.x;   // distance from the tile, less one
        T x[1];
        x[0] = j < avail ? row[REV ? tile + 1u + j : tile - 1u - j] : op.identity();
        T total;
        block_scan_dir<true, 1>(op, x, op.identity(), lds, total);   // the farthest first
        acc = op.combine(total, acc);
        if (acc.f) break;
    }
    return acc;
}

// Three launches: tile totals, exclusive scan of the totals (one block per
// row y), tiles again with their carry.  Op: T, identity(), load(i, y),
// combine(a, b), store(i, y, inclusive).
template <class Op>
__global__ void __launch_bounds__(BLOCK) k_scan_totals(Op op, uint32_t n, typename Op::T* totals) {
    using T = typename Op::T;
    __shared__ T lds[WAVES];
    T x[ITEMS];
    const uint32_t base = blockIdx.x * TILE + threadIdx.x * ITEMS;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) x[i] = base + i < n ? op.load(base + i, blockIdx.y) : op.identity();
    T total;
    block_scan<ITEMS>(op, x, op.identity(), lds, total);
    if (threadIdx.x == 0) totals[blockIdx.y * gridDim.x + blockIdx.x] = total;
}

template <class Op>
__global__ void __launch_bounds__(BLOCK) k_scan_spine(Op op, uint32_t n_tiles, typename Op::T* totals) {
    using T = typename Op::T;
    __shared__ T lds[WAVES];
    T* row = totals + (size_t)blockIdx.x * n_tiles;
    T carry = op.identity();
    for (uint32_t first = 0; first < n_tiles; first += BLOCK) {
        const uint32_t i = first + threadIdx.x;
        T x[1];
        x[0] = i < n_tiles ? row[i] : op.identity();
        T total;
        block_scan<1>(op, x, carry, lds, total);
        // exclusive: what precedes tile i
        const T before = shfl_up_t(x[0], 1);
        __shared__ T edge[WAVES];
        if (lane() == 63) edge[threadIdx.x >> 6] = x[0];
        __syncthreads();
        T excl = lane() > 0 ? before : ((threadIdx.x >> 6) > 0 ? edge[(threadIdx.x >> 6) - 1] : carry);
        if (i < n_tiles) row[i] = excl;
        carry = total;
        __syncthreads();
    }
}

template <class Op>
__global__ void __launch_bounds__(BLOCK) k_scan_apply(Op op, uint32_t n, const typename Op::T* totals) {
    using T = typename Op::T;
    __shared__ T lds[WAVES];
    T x[ITEMS];
    const uint32_t base = blockIdx.x * TILE + threadIdx.x * ITEMS;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) x[i] = base + i < n ? op.load(base + i, blockIdx.y) : op.identity();
    T total;
    block_scan<ITEMS>(op, x, totals[blockIdx.y * gridDim.x + blockIdx.x], lds, total);
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i)
        if (base + i < n) op.store(base + i, blockIdx.y, x[i]);
}

// Runs the scan of `rows` independent rows of n elements; totals: rows * ceil(n / TILE) elements of scratch.
template <class Op>
void run_scan(const Op& op, uint32_t n, uint32_t rows, typename Op::T* totals, hipStream_t st) {
    const uint32_t n_tiles = (n + TILE - 1u) / TILE;
    hipLaunchKernelGGL(k_scan_totals<Op>, dim3(n_tiles, rows), dim3(BLOCK), 0, st, op, n, totals);
    hipLaunchKernelGGL(k_scan_spine<Op>, dim3(rows), dim3(BLOCK), 0, st, op, n_tiles, totals);
    hipLaunchKernelGGL(k_scan_apply<Op>, dim3(n_tiles, rows), dim3(BLOCK), 0, st, op, n, totals);
}

}  // namespace scan
}  // namespace rtx
