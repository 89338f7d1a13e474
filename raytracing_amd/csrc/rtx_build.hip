// rtx_build.hip -- `BVH::from_slice` (src/bvh/mod.rs:76-131) on the device,
// for every PartitionBy strategy (src/bvh/partition.rs:58-266).  Citations are into the reference.
//
// The reference recurses depth first and, at every node, sorts the node's
// surfaces three times (splits_cache, partition.rs:119-171).  Here the tree is
// built breadth first, one level per pass over all surfaces, and the three
// orders by centroid are made ONCE (radix sort) and kept sorted by splitting
// them stably whenever a node is split.  The result is the reference's tree,
// node for node and bit for bit:
//
// * The leaves of a subtree are a contiguous range ("segment") of the current
//   surface order, and a subtree over m surfaces has 2m-1 nodes, so with the
//   left child at i+1 the right child of node i is i + 2*n_left: pre-order
//   indices are known top down.
// * `partition_in_place` (partition.rs:42-56) swaps the j-th failing element
//   of the left part with the j-th passing element of the right part counted
//   from the end; prefix counts of the predicate reproduce that permutation.
// * left / right boxes of a split are min/max folds, which do not depend on
//   the order of folding; the cost (partition.rs:87-106) is evaluated with the
//   reference's sequence of unfused f64 operations (-ffp-contract=off).
// * `min_by` keeps the first of equal minima (partition.rs:218): candidates
//   carry their position in the reference's iteration order (surface major /
//   axis minor for PerSurface, axis major for EqualSize) and ties on the cost
//   go to the lowest position.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/rtx.h"
#include "rtx_kernels.h"
#include "rtx_scan.h"

namespace rtx {
namespace build {

using scan::BLOCK;
using scan::TILE;
using scan::run_scan;
using scan::block_scan_dir;
constexpr unsigned long long NO_COST = ~0ull;

enum : uint32_t { ERR_NO_PLANE = 1, ERR_EMPTY_SIDE = 2, ERR_INTERNAL = 3 };
// Control words, 256 bytes apart: the count a level adds to with atomics must not share a line with words its threads read.
// ctl[ctl_count(level)] = segments of a level.
constexpr uint32_t CTL_STRIDE = 64, CTL_ERR = 4 * CTL_STRIDE, CTL_WORDS = 5 * CTL_STRIDE, REPORT_WORDS = 4;
__host__ __device__ constexpr uint32_t ctl_count(uint32_t level) { return (level & 3u) * CTL_STRIDE; }
// cost of a position whose key equals the next position's: the cost at the end of that run of equal keys
constexpr unsigned long long TIED = ~0ull - 1ull;

struct DBuild {
    uint32_t n, strategy, buckets, n_nodes;
    uint32_t n_pad;          // n rounded up to a multiple of four: row stride of cand_cost
    const double* box;       // [n][6] leaf boxes by surface: min xyz, max xyz
    double* cen;             // [3][n] centroid coordinate by surface
    uint32_t* order[2];      // surface at each position of the current order (this level / next level)
    uint32_t* pos_of;        // position of each surface in order[cur]
    uint32_t* sid[2][3];     // surfaces of every segment sorted by centroid, per axis
    double* skey[2][3];      // their centroid coordinates
    uint32_t* seg_at;        // [n] segment of a position at this level, NONE once the position is a leaf
    uint32_t* seg_a[2];      // segments (nodes still to split) of this level / the next: first position,
    uint32_t* seg_b[2];      //   one past the last,
    uint32_t* seg_node[2];   //   pre-order node index
    uint4* seg_rec[2];       // per segment of this level / the next: {first position, size, area factor of its box (a double)}
    uint4* seg_cut;          // decision per segment of this level: {n_left | strict << 31, axis, plane (a double)}
    uint32_t* seg_axis;      // the same, field by field
    double* seg_plane;
    uint32_t* seg_nleft;     // bit 31: predicate is `<` (midpoint) instead of `<=`
    uint32_t* seg_child[2];  // segment index of the left / right child at the next level, NONE for a leaf
    unsigned long long* best_cost[2];   // by segment of this level / the next
    unsigned long long* best_pick[2];   // the winning candidate: its place in the reference's order of candidates << 32 | surfaces left of its plane
    unsigned long long* cand_cost;   // [3][n]
    double* L;               // [3][n][6] box of the segment's sorted surfaces up to and including a position (not PerSurface)
    double* R;               // [3][n][6] ... from a position to the segment's end (EqualSize only)
    uint8_t* flags;          // bit k: predicate of the element at a position of {order, sid[0], sid[1], sid[2]}[k]
    uint32_t* before;        // [n] passing elements of the position's segment before it in the surface order
    uint32_t* F;             // positions of the failing elements of a segment's left part, in order
    uint32_t* G;             // positions of the passing elements of its right part, from the end
    uint32_t* ctl;           // control words, see ctl_count / CTL_ERR
    uint32_t* report;        // [REPORT_WORDS] what the host reads after a level: its segments, the next level's, error
    double* seg_box;         // [n][6] box of every segment of this level (PerSurface; the SAH passes fold it)
    uint32_t* leaf_node;     // [n] node index of the leaf at a position
    uint32_t* node_right;    // outputs, [n_nodes]
    uint32_t* node_prim;
    double* node_bmin;       // [3][n_nodes]
    double* node_bmax;
};

using scan::lane;

// f64::total_cmp as an unsigned key
__device__ __forceinline__ unsigned long long ordered_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// ---------------------------------------------------------------- scans
struct SBox {
    double v[6];
    uint32_t f;   // a segment starts at this element
};

__device__ __forceinline__ SBox shfl_up_t(const SBox& x, uint32_t d) {
    SBox r;
#pragma unroll
    for (int k = 0; k < 6; ++k) r.v[k] = __shfl_up(x.v[k], d);
    r.f = __shfl_up(x.f, d);
    return r;
}

// AABB::merge (src/aabb.rs:40-53) folded along a list, starting again wherever a segment starts
struct BoxOp {
    using T = SBox;
    __device__ SBox identity() const {
        SBox r;
        r.v[0] = r.v[1] = r.v[2] = __longlong_as_double(0x7FF0000000000000ll);
        r.v[3] = r.v[4] = r.v[5] = __longlong_as_double((long long)0xFFF0000000000000ull);
        r.f = 0;
        return r;
    }
    __device__ SBox combine(const SBox& a, const SBox& b) const {
        SBox r;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.v[k] = b.f ? b.v[k] : fmin(a.v[k], b.v[k]);
            r.v[3 + k] = b.f ? b.v[3 + k] : fmax(a.v[3 + k], b.v[3 + k]);
        }
        r.f = a.f | b.f;
        return r;
    }
};

// The segmented scan of boxes over a block, built to need few registers and no trip through LDS per step.
//  * Over a wave: shifts within rows of 16 lanes (DPP row_shr / row_shl), then the row totals handed on by readlane.
//    The shuffle scan of rtx_scan.h gives the same result (the fold is associative) but waits for an LDS permute of
//    thirteen registers at each of its steps.
//  * Where a lane takes in its neighbour's value depends on the segment starts only, not on the values: those decisions
//    are made once (SegPlan) and the six coordinates are then scanned one after the other, a pair of registers at a time.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double((int)dpp_u32<CTRL>((uint32_t)__double2hiint(v)), (int)dpp_u32<CTRL>((uint32_t)__double2loint(v)));
}
__device__ __forceinline__ double lane_f64(double v, int from) {   // the value of one lane, to all
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), from), __builtin_amdgcn_readlane(__double2loint(v), from));
}
// AABB::merge of one coordinate (src/aabb.rs:40-53): minima for the lower corner (c < 3), maxima for the upper
__device__ __forceinline__ double merge_coordinate(int c, double a, double b) { return c < 3 ? fmin(a, b) : fmax(a, b); }

template <bool REV>
struct SegPlan {
    uint32_t rank, wrank;   // place of the lane in its wave and of the wave in the block, in scan order
    bool take1, take2, take4, take8, take_a, take_b;   // the lane takes in: the lane 1, 2, 4, 8 before it in its row; the rows before
    bool closed;            // a segment starts in this wave at or before this lane
    bool before_closed;     // ... at or before the lane before this one
    bool wave_closed[scan::WAVES];   // ... anywhere in wave w (scan order)
    // flags: whether a segment starts at this thread's items (any of them)
    __device__ __forceinline__ void make(bool starts, uint32_t* lds_f) {
        const uint32_t l = lane();
        rank = REV ? 63u - l : l;
        wrank = REV ? scan::WAVES - 1u - (threadIdx.x >> 6) : (threadIdx.x >> 6);
        const uint32_t in_row = rank & 15u, row = rank >> 4;
        uint32_t f = starts ? 1u : 0u, p;
        p = REV ? dpp_u32<0x101>(f) : dpp_u32<0x111>(f);
        take1 = in_row >= 1u && !f;
        if (in_row >= 1u) f |= p;
        p = REV ? dpp_u32<0x102>(f) : dpp_u32<0x112>(f);
        take2 = in_row >= 2u && !f;
        if (in_row >= 2u) f |= p;
        p = REV ? dpp_u32<0x104>(f) : dpp_u32<0x114>(f);
        take4 = in_row >= 4u && !f;
        if (in_row >= 4u) f |= p;
        p = REV ? dpp_u32<0x108>(f) : dpp_u32<0x118>(f);
        take8 = in_row >= 8u && !f;
        if (in_row >= 8u) f |= p;
        // the last lane of a row in scan order holds the row's flag
        const uint32_t f15 = (uint32_t)__builtin_amdgcn_readlane((int)f, REV ? 48 : 15), f47 = (uint32_t)__builtin_amdgcn_readlane((int)f, REV ? 16 : 47);
        take_a = (row == 1u || row == 3u) && !f;
        if (row == 1u) f |= f15;
        if (row == 3u) f |= f47;
        const uint32_t f31 = (uint32_t)__builtin_amdgcn_readlane((int)f, 32 - (REV ? 0 : 1));   // rank 31: lane 31, REV lane 32
        take_b = row >= 2u && !f;
        if (row >= 2u) f |= f31;
        closed = f != 0u;
        before_closed = (REV ? __shfl_down(f, 1) : __shfl_up(f, 1)) != 0u;
        if (rank == 63u) lds_f[wrank] = f;
        __syncthreads();
#pragma unroll
        for (uint32_t w = 0; w < scan::WAVES; ++w) wave_closed[w] = lds_f[w] != 0u;
    }
    // inclusive scan of one coordinate of the threads' folds over the wave
    __device__ __forceinline__ double wave(int c, double v) const {
        double p;
        p = REV ? dpp_f64<0x101>(v) : dpp_f64<0x111>(v);
        if (take1) v = merge_coordinate(c, p, v);
        p = REV ? dpp_f64<0x102>(v) : dpp_f64<0x112>(v);
        if (take2) v = merge_coordinate(c, p, v);
        p = REV ? dpp_f64<0x104>(v) : dpp_f64<0x114>(v);
        if (take4) v = merge_coordinate(c, p, v);
        p = REV ? dpp_f64<0x108>(v) : dpp_f64<0x118>(v);
        if (take8) v = merge_coordinate(c, p, v);
        const double r15 = lane_f64(v, REV ? 48 : 15), r47 = lane_f64(v, REV ? 16 : 47);
        if (take_a) v = merge_coordinate(c, (rank >> 4) == 1u ? r15 : r47, v);
        const double r31 = lane_f64(v, REV ? 32 : 31);
        if (take_b) v = merge_coordinate(c, r31, v);
        return v;
    }
};

// Inclusive segmented scan of BLOCK*N boxes held N per thread as coordinates x[i][0..5]; REV: from the block's last box to its
// first.  starts: bit i = a segment starts at item i (REV: ends there).  carry (may be null: nothing) = the box of what
// lies before the block.  On return x holds the scan; if total is given, thread 0 stores the fold of carry and block
// there, flagged if a segment starts in the block.  lds_v / lds_f: scan::WAVES entries each.  Three barriers.
template <bool REV, uint32_t N>
__device__ __forceinline__ void block_scan_boxes(double (&x)[N][6], uint32_t starts, const double* carry, SBox* total, double (*lds_v)[6], uint32_t* lds_f) {
    // what of the thread's own items lies before item i in scan order, and whether item i is open to what comes before the thread
    bool open[N];
    {
        bool any = false;
#pragma unroll
        for (uint32_t k = 0; k < N; ++k) {
            const uint32_t i = REV ? N - 1u - k : k;
            any = any || ((starts >> i) & 1u);
            open[i] = !any;
        }
    }
    SegPlan<REV> plan;
    plan.make(starts != 0u, lds_f);
    double inc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (uint32_t k = 1; k < N; ++k) {
            const uint32_t i = REV ? N - 1u - k : k, prev = REV ? i + 1u : i - 1u;
            if (!((starts >> i) & 1u)) x[i][c] = merge_coordinate(c, x[prev][c], x[i][c]);
        }
        inc[c] = plan.wave(c, x[REV ? 0 : N - 1u][c]);
        if (plan.rank == 63u) lds_v[plan.wrank][c] = inc[c];
    }
    __syncthreads();
    bool total_closed = false;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double identity = __longlong_as_double(c < 3 ? 0x7FF0000000000000ll : (long long)0xFFF0000000000000ull);
        double prefix = carry ? carry[c] : identity, all = prefix;
#pragma unroll
        for (uint32_t w = 0; w < scan::WAVES; ++w) {
            const double t = lds_v[w][c];
            if (w < plan.wrank) prefix = plan.wave_closed[w] ? t : merge_coordinate(c, prefix, t);
            all = plan.wave_closed[w] ? t : merge_coordinate(c, all, t);
        }
        const double before = REV ? __shfl_down(inc[c], 1) : __shfl_up(inc[c], 1);
        if (plan.rank > 0u) prefix = plan.before_closed ? before : merge_coordinate(c, prefix, before);
#pragma unroll
        for (uint32_t i = 0; i < N; ++i)
            if (open[i]) x[i][c] = merge_coordinate(c, prefix, x[i][c]);
        if (total && threadIdx.x == 0) total->v[c] = all;
    }
#pragma unroll
    for (uint32_t w = 0; w < scan::WAVES; ++w) total_closed = total_closed || plan.wave_closed[w];
    if (total && threadIdx.x == 0) total->f = total_closed ? 1u : 0u;   // (the carry's own flag plays no part in a tile total)
    __syncthreads();
}

// Row y = axis + 3*direction.  Direction 0 folds the boxes of a segment's
// sorted surfaces from its first position on (left_of, partition.rs:136-150),
// direction 1 from its last position back (right_of, partition.rs:152-166).
struct BoxScan {
    using T = SBox;
    DBuild B;
    uint32_t cur;
    __device__ T identity() const { return BoxOp{}.identity(); }
    __device__ T load(uint32_t i, uint32_t y) const {
        const uint32_t axis = y % 3u, back = y / 3u;
        const uint32_t pos = back ? B.n - 1u - i : i;
        const uint32_t s = B.seg_at[pos];
        T r = identity();
        r.f = 1;
        // a finished leaf: a segment of its own, value never read.  After a failed level the lists of this one were never
        // written: nothing is read through them (the host stops two levels late).
        if (s == NONE || B.ctl[CTL_ERR] != 0u) return r;
        const bool edge = back ? pos == B.n - 1u : pos == 0u;
        r.f = (edge || B.seg_at[back ? pos + 1u : pos - 1u] != s) ? 1u : 0u;
        // bisection keeps no sorted lists: the fold over the surface order gives the same box
        const uint32_t* list = B.strategy == RTX_BISECT ? B.order[cur] : B.sid[cur][axis];
        const double* b = B.box + 6ull * list[pos];
#pragma unroll
        for (int k = 0; k < 6; ++k) r.v[k] = b[k];
        return r;
    }
    __device__ T combine(const T& a, const T& b) const { return BoxOp{}.combine(a, b); }
    __device__ void store(uint32_t i, uint32_t y, const T& inc) const {
        const uint32_t axis = y % 3u, back = y / 3u;
        const uint32_t pos = back ? B.n - 1u - i : i;
        if (B.seg_at[pos] == NONE || B.ctl[CTL_ERR] != 0u) return;
        double* out = (back ? B.R : B.L) + ((size_t)axis * B.n + pos) * 6u;
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = inc.v[k];
    }
};

// Passing elements of a segment up to a position, for the four lists at once.
struct SCount {
    uint32_t c[4];
    uint32_t f;   // a segment starts at this element
};
__device__ __forceinline__ SCount shfl_up_t(const SCount& x, uint32_t d) {
    SCount r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.c[k] = __shfl_up(x.c[k], d);
    r.f = __shfl_up(x.f, d);
    return r;
}
__device__ __forceinline__ SCount shfl_down_t(const SCount& x, uint32_t d) {
    SCount r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r.c[k] = __shfl_down(x.c[k], d);
    r.f = __shfl_down(x.f, d);
    return r;
}
struct CountOp {
    __device__ SCount identity() const { return SCount{{0, 0, 0, 0}, 0}; }
    __device__ SCount combine(const SCount& a, const SCount& b) const {
        SCount r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.c[k] = b.f ? b.c[k] : a.c[k] + b.c[k];
        r.f = a.f | b.f;
        return r;
    }
};

// ---------------------------------------------------------------- split search
// split_at, partition.rs:182-197: the number of a segment's surfaces whose
// centroid is on the left of plane `at`, 0 when the plane separates nothing.
// `key` = the segment's sorted centroid coordinates.
__device__ __forceinline__ uint32_t n_left_for(const double* key, uint32_t n, double at) {
    if (at < key[0] || at >= key[n - 1u]) return 0u;
    uint32_t lo = 0, hi = n - 1u;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (at >= key[mid + 1u]) lo = mid + 1u; else hi = mid;
    }
    return lo + 1u;
}
// number of keys below `at` (LongestAxisMidpoint's predicate, partition.rs:70-78)
__device__ __forceinline__ uint32_t n_below(const double* key, uint32_t n, double at) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (key[mid] < at) lo = mid + 1u; else hi = mid;
    }
    return lo;
}

// surface_area_factor, partition.rs:87-90
__device__ __forceinline__ double area_factor(const double* b) {
    const double dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dx * dz + dy * dz;
}
// surface_area_heuristic, partition.rs:92-106, in the reference's order of operations
__device__ __forceinline__ double split_cost_areas(double area_left, uint32_t n_left, double area_right, uint32_t n_right, double parent) {
    const double p_left = area_left / parent;
    const double p_right = area_right / parent;
    const double cost = 1.0 + p_left * (double)n_left + p_right * (double)n_right;
    // A parent box without area (zero-radius spheres on a line) makes the cost 0/0.  total_cmp orders NaNs by their sign,
    // and the default NaN of the hosts the reference and the host builder run on (x86-64) is the negative one, which
    // min_by ranks lowest; this device's default NaN is the positive one.  Every NaN cost takes the host's form, so
    // that both builders pick the same plane for such a segment.
    return cost != cost ? __longlong_as_double((long long)0xFFF8000000000000ull) : cost;
}
__device__ __forceinline__ double split_cost(const double* left, uint32_t n_left, const double* right, uint32_t n_right, const double* bb) {
    return split_cost_areas(area_factor(left), n_left, area_factor(right), n_right, area_factor(bb));
}

__device__ __forceinline__ double cost_of_split(const DBuild& B, uint32_t axis, uint32_t a, uint32_t b, uint32_t k) {
    const double* left = B.L + ((size_t)axis * B.n + a + k - 1u) * 6u;
    const double* right = B.R + ((size_t)axis * B.n + a + k) * 6u;
    const double* bb = B.L + (size_t)(b - 1u) * 6u;   // the segment's box, from the x order
    return split_cost(left, k, right, (b - a) - k, bb);
}

// An atomic minimum that first looks whether it would change anything.  Near the root thousands of waves report
// to the same few words, and atomics on one address are served one after the other: the look (past the compute unit's
// cache, which may hold the word as it was before the launch) lets all but the first few go without.
__device__ __forceinline__ void lazy_min(unsigned long long* p, unsigned long long v) {
    if (v < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(p, v);
}
// Where the cheapest cost of segment s is kept: the first segments of a level, which whole blocks report to, have a
// 128-byte line each.
constexpr uint32_t BEST_APART = 4096, BEST_EXTRA = BEST_APART * 16u;
__host__ __device__ __forceinline__ uint32_t best_slot(uint32_t s) { return s < BEST_APART ? s * 16u : BEST_EXTRA + (s - BEST_APART); }
// atomicMin(&table[best_slot(s)], v) for the lanes with s != NONE.  Lanes hold consecutive positions, so the lanes of a
// segment are a run: a segmented scan leaves the minimum of every run in its last lane, which reports for the run.
__device__ __forceinline__ void segment_min(unsigned long long* table, uint32_t s, unsigned long long v) {
    if (__ballot(s != NONE && v != NO_COST) == 0ull) return;
    const uint32_t l = lane();
    unsigned long long m = s != NONE ? v : NO_COST;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const unsigned long long o = __shfl_up(m, d);
        const uint32_t os = __shfl_up(s, d);
        if (l >= d && os == s) m = o < m ? o : m;   // (a run is contiguous: if the lane d below is of this segment, so are all between)
    }
    const uint32_t next = __shfl_down(s, 1);
    if (s != NONE && m != NO_COST && (l == 63u || next != s)) lazy_min(&table[best_slot(s)], m);
}

// N consecutive 32-bit entries of a table from `base` on, a multiple of N; every table is padded to 256 bytes, so they come
// in one load even where the table's last entries are not all there (what lies beyond n is not used).
template <uint32_t N>
__device__ __forceinline__ void load_run(const uint32_t* table, uint32_t base, uint32_t n, uint32_t (&out)[N]) {
    static_assert(N == 1 || N == 2 || N == 4, "one load per thread");
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) out[i] = NONE;
    if (base >= n) return;
    if (N == 4) {
        const uint4 v = *reinterpret_cast<const uint4*>(table + base);
        out[0] = v.x; out[1 % N] = v.y; out[2 % N] = v.z; out[3 % N] = v.w;
    } else if (N == 2) {
        const uint2 v = *reinterpret_cast<const uint2*>(table + base);
        out[0] = v.x; out[1 % N] = v.y;
    } else {
        out[0] = table[base];
    }
}
// Segment of the positions base-1 .. base+N of a thread's items, NONE outside the list.
template <uint32_t N>
__device__ __forceinline__ void load_segments(const DBuild& B, uint32_t base, uint32_t (&sg)[N + 2]) {
    const uint32_t n = B.n;
    uint32_t own[N];
    load_run<N>(B.seg_at, base, n, own);
    sg[0] = base - 1u < n ? B.seg_at[base - 1u] : NONE;
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) sg[i + 1u] = base + i < n ? own[i] : NONE;
    sg[N + 1u] = base + N < n ? B.seg_at[base + N] : NONE;
}

#ifndef RTX_SAH_WAVES
#define RTX_SAH_WAVES 4
#endif
#ifndef RTX_SAH_ITEMS
#define RTX_SAH_ITEMS 2
#endif
constexpr int SAH_WAVES = RTX_SAH_WAVES;   // waves per SIMD the register budget of the kernels below allows
constexpr uint32_t SAH_ITEMS = RTX_SAH_ITEMS, SAH_TILE = BLOCK * SAH_ITEMS;   // positions per thread and per block

// exclusive scan of every row of tile totals (one block per row, eight totals per thread); only the boxes of the result are used
constexpr uint32_t SPINE_ITEMS = 8, SPINE_TILE = BLOCK * SPINE_ITEMS;
__global__ void __launch_bounds__(BLOCK) k_box_spine(uint32_t n_tiles, SBox* totals) {
    __shared__ double lds_v[scan::WAVES][6];
    __shared__ uint32_t lds_f[scan::WAVES];
    __shared__ double edge[scan::WAVES][6];
    __shared__ SBox running[2];   // what precedes the chunk / the chunk after this one
    SBox* row = totals + (size_t)blockIdx.x * n_tiles;
    const uint32_t wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running[0] = BoxOp{}.identity();
    __syncthreads();
    uint32_t which = 0;
    for (uint32_t first = 0; first < n_tiles; first += SPINE_TILE, which ^= 1u) {
        const uint32_t base = first + threadIdx.x * SPINE_ITEMS;
        double x[SPINE_ITEMS][6];
        uint32_t starts = 0;
#pragma unroll
        for (uint32_t i = 0; i < SPINE_ITEMS; ++i) {
            const SBox t = base + i < n_tiles ? row[base + i] : BoxOp{}.identity();
#pragma unroll
            for (int c = 0; c < 6; ++c) x[i][c] = t.v[c];
            starts |= (t.f ? 1u : 0u) << i;
        }
        block_scan_boxes<false, SPINE_ITEMS>(x, starts, running[which].v, &running[which ^ 1u], lds_v, lds_f);
        // exclusive: the scan one total back (the thread's own, the thread before's last, the wave before's, the carry)
        if (lane() == 63)
#pragma unroll
            for (int c = 0; c < 6; ++c) edge[wave][c] = x[SPINE_ITEMS - 1u][c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double before = __shfl_up(x[SPINE_ITEMS - 1u][c], 1);
            double excl = lane() > 0 ? before : (wave > 0 ? edge[wave - 1u][c] : running[which].v[c]);
#pragma unroll
            for (uint32_t i = 0; i < SPINE_ITEMS; ++i) {
                if (base + i < n_tiles) row[base + i].v[c] = excl;
                excl = x[i][c];
            }
        }
        __syncthreads();
    }
}

// PerSurface (partition.rs:254-266): every surface's centroid coordinate on every axis is a candidate plane.  The cost of
// the plane at a position of an axis' sorted order needs the box of the segment's surfaces up to that position and the box
// of those behind it (left_of / right_of, partition.rs:136-166): a forward and a backward segmented scan of the boxes.
// Neither scan is written to memory.  A block scans its tile both ways; a first pass (APPLY = false) leaves every tile's
// totals and folds the box of every segment (the parent in the cost, partition.rs:93, and the node's box), the second
// takes what the scans carry into the tile (the totals, scanned in between by k_box_spine) and prices its positions on the
// spot.  Positions whose key equals the next one's share the plane, and so the cost, of the last position of that run
// of equal keys (split_at, partition.rs:182-197): they are marked TIED and looked up when the winner is picked.
// totals: rows 0-2 forward per axis by tile, rows 3-5 backward per axis by tile counted from the end.
template <bool APPLY>
__global__ void __launch_bounds__(BLOCK, SAH_WAVES) k_sah_per_surface(DBuild B, uint32_t cur, uint32_t level, SBox* totals) {
    __shared__ double lds_v[scan::WAVES][6];
    __shared__ uint32_t lds_f[scan::WAVES];
    __shared__ double edge[scan::WAVES];
    if (B.ctl[ctl_count(level)] == 0u || B.ctl[CTL_ERR] != 0u) return;
    constexpr uint32_t ITEMS = SAH_ITEMS;
    const uint32_t n = B.n, tile = blockIdx.x, axis = blockIdx.y, n_tiles = gridDim.x, tile_b = n_tiles - 1u - tile;
    const uint32_t base = tile * SAH_TILE + threadIdx.x * ITEMS;
    uint32_t sg[ITEMS + 2];
    load_segments<ITEMS>(B, base, sg);
    uint32_t list[ITEMS];
    load_run<ITEMS>(B.sid[cur][axis], base, n, list);
    double raw[ITEMS][6], x[ITEMS][6];
    uint32_t starts = 0, ends = 0;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t pos = base + i, s = sg[i + 1u];
#pragma unroll
        for (int c = 0; c < 6; ++c) raw[i][c] = __longlong_as_double(c < 3 ? 0x7FF0000000000000ll : (long long)0xFFF0000000000000ull);
        if (pos < n) {
            if (s == NONE) {   // a finished leaf: a segment of its own
                starts |= 1u << i;
                ends |= 1u << i;
            } else {
                starts |= (uint32_t)(sg[i] != s) << i;
                ends |= (uint32_t)(sg[i + 2u] != s) << i;
                const double2* b = reinterpret_cast<const double2*>(B.box + 6ull * list[i]);   // 48 bytes apart: 16-byte aligned
                const double2 b0 = b[0], b1 = b[1], b2 = b[2];
                raw[i][0] = b0.x; raw[i][1] = b0.y; raw[i][2] = b1.x; raw[i][3] = b1.y; raw[i][4] = b2.x; raw[i][5] = b2.y;
            }
        }
    }
    SBox* const row_f = totals + (size_t)axis * n_tiles;
    SBox* const row_b = totals + (size_t)(3u + axis) * n_tiles;
    if (!APPLY) {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i)
#pragma unroll
            for (int c = 0; c < 6; ++c) x[i][c] = raw[i][c];
        block_scan_boxes<true, ITEMS>(x, ends, nullptr, &row_b[tile_b], lds_v, lds_f);
        block_scan_boxes<false, ITEMS>(raw, starts, nullptr, &row_f[tile], lds_v, lds_f);
        if (axis != 0u) return;
        // The segment's box: at a segment's last position the scan holds the fold since the segment's start, or since the
        // tile's if the segment started before it; the rest follows from the tile totals (k_segment_areas).
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) {
            const uint32_t pos = base + i, s = sg[i + 1u];
            if (pos >= n || s == NONE || !((ends >> i) & 1u)) continue;
#pragma unroll
            for (int k = 0; k < 6; ++k) B.seg_box[6ull * s + k] = raw[i][k];
        }
        return;
    }
    double area_behind[ITEMS];   // area factor of the surfaces behind a position: the backward scan one position on
    {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i)
#pragma unroll
            for (int c = 0; c < 6; ++c) x[i][c] = raw[i][c];
        const double* carry = row_b[tile_b].v;
        block_scan_boxes<true, ITEMS>(x, ends, carry, nullptr, lds_v, lds_f);
        // ... the next item, the next thread's first, the carry
        const double own = area_factor(x[0]);
#pragma unroll
        for (uint32_t i = 0; i + 1u < ITEMS; ++i) area_behind[i] = area_factor(x[i + 1u]);
        const uint32_t wave = threadIdx.x >> 6;
        if (lane() == 0) edge[wave] = own;
        __syncthreads();
        double nx = __shfl_down(own, 1);
        if (lane() == 63) {
            if (wave + 1u < scan::WAVES) {
                nx = edge[wave + 1u];
            } else {
                nx = area_factor(carry);
            }
        }
        area_behind[ITEMS - 1u] = nx;
    }
    {
        const double* carry = row_f[tile].v;
        block_scan_boxes<false, ITEMS>(raw, starts, carry, nullptr, lds_v, lds_f);
    }

    const double* key = B.skey[cur][axis];
    double keys[ITEMS + 1];
    if (ITEMS > 1 && base + ITEMS < n) {
#pragma unroll
        for (uint32_t i = 0; i + 1u < ITEMS; i += 2) {
            const double2 kk = *reinterpret_cast<const double2*>(key + base + i);
            keys[i] = kk.x;
            keys[i + 1u] = kk.y;
        }
        keys[ITEMS] = key[base + ITEMS];
    } else {
#pragma unroll
        for (uint32_t i = 0; i <= ITEMS; ++i) keys[i] = base + i < n ? key[base + i] : 0.0;
    }
    uint4 rec[ITEMS];   // first position, size and area factor of the position's segment
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) rec[i] = base + i < n && sg[i + 1u] != NONE ? B.seg_rec[cur][sg[i + 1u]] : make_uint4(0, 0, 0, 0);
    unsigned long long ck[ITEMS], out[ITEMS];
    uint32_t ss[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t pos = base + i, s = sg[i + 1u];
        ck[i] = out[i] = NO_COST;   // a segment's last position separates nothing
        ss[i] = NONE;
        if (pos < n && s != NONE) {
            ss[i] = s;
            if (sg[i + 2u] == s) {
                if (keys[i + 1u] > keys[i]) {
                    const uint32_t k = pos + 1u - rec[i].x;
                    const double parent = __hiloint2double((int)rec[i].w, (int)rec[i].z);
                    out[i] = ck[i] = ordered_key(split_cost_areas(area_factor(raw[i]), k, area_behind[i], rec[i].y - k, parent));
                } else {
                    out[i] = TIED;
                }
            }
        }
    }
    unsigned long long* const cand = B.cand_cost + (size_t)axis * B.n_pad + base;   // rows start at multiples of four
    if (ITEMS > 1 && base + ITEMS <= n) {
#pragma unroll
        for (uint32_t i = 0; i + 1u < ITEMS; i += 2) reinterpret_cast<ulonglong2*>(cand)[i / 2u] = make_ulonglong2(out[i], out[i + 1u]);
    } else {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i)
            if (base + i < n) cand[i] = out[i];
    }
    // The cheapest per segment.  Near the root a whole block prices positions of one segment: then the block reports once.
    // Otherwise one call for the threads whose positions share a segment, and the other threads position by position.
    bool same = true;
    unsigned long long m = ck[0];
#pragma unroll
    for (uint32_t i = 1; i < ITEMS; ++i) {
        same = same && ss[i] == ss[0];
        m = ck[i] < m ? ck[i] : m;
    }
    __shared__ unsigned long long wave_best[scan::WAVES];
    __shared__ uint32_t wave_segment[scan::WAVES];
    constexpr uint32_t MIXED = NONE - 1u;
    {
        const unsigned long long with = __ballot(same && ss[0] != NONE);
        uint32_t s0 = NONE;
        if (with) s0 = __shfl(ss[0], (int)__ffsll((long long)with) - 1);
        const bool one = __ballot(!same || (ss[0] != NONE && ss[0] != s0)) == 0ull;
        unsigned long long w = same && ss[0] != NONE ? m : NO_COST;
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(w, d);
            w = o < w ? o : w;
        }
        if (lane() == 0) {
            wave_segment[threadIdx.x >> 6] = one ? s0 : MIXED;
            wave_best[threadIdx.x >> 6] = w;
        }
    }
    __syncthreads();
    uint32_t block_segment = NONE;
    bool block_one = true;
    unsigned long long block_best = NO_COST;
#pragma unroll
    for (uint32_t w = 0; w < scan::WAVES; ++w) {
        const uint32_t sw = wave_segment[w];
        if (sw == MIXED || (sw != NONE && block_segment != NONE && sw != block_segment)) block_one = false;
        if (sw != NONE && sw != MIXED) block_segment = sw;
        block_best = wave_best[w] < block_best ? wave_best[w] : block_best;
    }
    if (block_one) {
        if (threadIdx.x == 0 && block_segment != NONE && block_best != NO_COST) lazy_min(&B.best_cost[cur][best_slot(block_segment)], block_best);
        return;
    }
    segment_min(B.best_cost[cur], same ? ss[0] : NONE, same ? m : NO_COST);
    if (__ballot(!same) != 0ull) {
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; ++i) segment_min(B.best_cost[cur], same ? NONE : ss[i], same ? NO_COST : ck[i]);
    }
}
// Completes the box of every segment and notes its area factor (the parent in surface_area_heuristic, partition.rs:93)
// next to the segment's first position and size: one 16-byte record for the pricing pass.  The first SAH pass left the
// fold over the segment's positions in its last tile; what precedes that tile in the segment is the tile's entry in the
// scanned totals of the x order.
__global__ void __launch_bounds__(BLOCK) k_segment_areas(DBuild B, uint32_t cur, uint32_t level, const SBox* row_f) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= B.ctl[ctl_count(level)] || B.ctl[CTL_ERR] != 0u) return;
    const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s], tile_a = a / SAH_TILE, tile_e = (b - 1u) / SAH_TILE;
    double bb[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) bb[k] = B.seg_box[6ull * s + k];
    if (tile_a < tile_e) {
        const SBox& before = row_f[tile_e];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bb[k] = fmin(before.v[k], bb[k]);
            bb[3 + k] = fmax(before.v[3 + k], bb[3 + k]);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) B.seg_box[6ull * s + k] = bb[k];
    }
    const double area = area_factor(bb);
    B.seg_rec[cur][s].z = (uint32_t)__double2loint(area);
    B.seg_rec[cur][s].w = (uint32_t)__double2hiint(area);
}
// min_by keeps the first of equal minima in surface-major, axis-minor order (partition.rs:218,254-266)
__global__ void __launch_bounds__(BLOCK) k_pick_per_surface(DBuild B, uint32_t cur, uint32_t level) {
    if (B.ctl[ctl_count(level)] == 0u || B.ctl[CTL_ERR] != 0u) return;
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x, axis = blockIdx.y;
    if (pos >= B.n) return;
    const uint32_t s = B.seg_at[pos];
    if (s == NONE) return;
    unsigned long long ck = B.cand_cost[(size_t)axis * B.n_pad + pos];
    uint32_t k = 0;   // surfaces left of the plane, once it is needed
    if (ck == TIED) {
        const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s];
        const double* key = B.skey[cur][axis];
        k = n_left_for(key + a, b - a, key[pos]);
        if (!k) return;
        ck = B.cand_cost[(size_t)axis * B.n_pad + a + k - 1u];
    }
    if (ck == NO_COST || ck != B.best_cost[cur][best_slot(s)]) return;
    if (!k) k = pos + 1u - B.seg_a[cur][s];
    atomicMin(&B.best_pick[cur][s], (unsigned long long)(B.pos_of[B.sid[cur][axis][pos]] * 3u + axis) << 32 | k);
}

// EqualSize (partition.rs:229-248): buckets-1 planes per axis, evenly spaced
// over the segment's box.  One thread per (segment, axis, plane).
template <int PASS>
__global__ void __launch_bounds__(BLOCK) k_candidates_equal_size(DBuild B, uint32_t cur, uint32_t level) {
    const uint32_t n_seg = B.ctl[ctl_count(level)];
    if (B.ctl[CTL_ERR] != 0u) return;
    const uint32_t per_axis = B.buckets - 1u, per_seg = 3u * per_axis;
    const unsigned long long t = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (t >= (unsigned long long)n_seg * per_seg) return;
    const uint32_t s = (uint32_t)(t / per_seg), idx = (uint32_t)(t % per_seg);
    const uint32_t axis = idx / per_axis, i = idx % per_axis + 1u;
    const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s];
    const double* bb = B.L + (size_t)(b - 1u) * 6u;
    const double start = bb[axis];
    const double step = (bb[3 + axis] - bb[axis]) / (double)B.buckets;
    const double at = start + ((double)i * step);
    const uint32_t k = n_left_for(B.skey[cur][axis] + a, b - a, at);
    if (!k) return;
    const unsigned long long ck = ordered_key(cost_of_split(B, axis, a, b, k));
    if (PASS == 0) atomicMin(&B.best_cost[cur][best_slot(s)], ck);
    else if (ck == B.best_cost[cur][best_slot(s)]) atomicMin(&B.best_pick[cur][s], (unsigned long long)idx << 32 | k);
}

// longest_axis, partition.rs:28-40: max_by returns the last of equal maxima
__device__ __forceinline__ uint32_t longest_axis(const double* bb) {
    uint32_t pick = 0;
    for (uint32_t axis = 1; axis < 3; ++axis) {
        const double cur = bb[3 + pick] - bb[pick], cand = bb[3 + axis] - bb[axis];
        if (!(ordered_key(cur) > ordered_key(cand))) pick = axis;
    }
    return pick;
}

// One thread per segment: fixes the split, writes the internal node and
// opens the children (build_tree_rec, src/bvh/mod.rs:92-131).  The number of segments of a level lives on the device
// (ctl[ctl_count(level)]); the host launches enough threads for the most a level can hold.
__global__ void __launch_bounds__(BLOCK) k_decide(DBuild B, uint32_t cur, uint32_t level) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n_seg = B.ctl[ctl_count(level)];
    if (s == 0u) B.ctl[ctl_count(level + 2u)] = 0u;   // neither read nor counted up at this level
    if (s >= n_seg || B.ctl[CTL_ERR] != 0u) return;
    const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s], node = B.seg_node[cur][s], m = b - a;
    // the segment's box: from the scan of the x order, or as the SAH passes folded it
    const double* bb = B.L ? B.L + (size_t)(b - 1u) * 6u : B.seg_box + 6ull * s;
    uint32_t axis = 0, n_left = 0, strict = 0;
    double at = 0.0;
    // until decided otherwise: nothing moves and the segment has no children (the passes after this one also stop on an error)
    B.seg_axis[s] = 0;
    B.seg_plane[s] = 0.0;
    B.seg_nleft[s] = m;
    B.seg_cut[s] = make_uint4(m, 0u, 0u, 0u);
    B.seg_child[0][s] = NONE;
    B.seg_child[1][s] = NONE;
    if (B.strategy == RTX_BISECT) {   // longest_axis_bisect_slice, partition.rs:58-68: sort on the axis, split in the middle
        axis = longest_axis(bb);
        n_left = m / 2u;
    } else if (B.strategy == RTX_MIDPOINT) {
        axis = longest_axis(bb);
        at = bb[axis] + (bb[3 + axis] - bb[axis]) * 0.5;   // AABB::centroid, src/aabb.rs:29-31
        strict = 1;
        n_left = n_below(B.skey[cur][axis] + a, m, at);
    } else {
        const unsigned long long pick = B.best_pick[cur][s];
        if (pick == ~0ull) {
            atomicMax(&B.ctl[CTL_ERR], (uint32_t)ERR_NO_PLANE);
            return;
        }
        const uint32_t idx = (uint32_t)(pick >> 32);
        n_left = (uint32_t)pick;   // split_at for the winner's plane (partition.rs:182-197), worked out when it was priced
        if (B.strategy == RTX_SAH_PER_SURFACE) {
            axis = idx % 3u;
            // the winner's centroid coordinate: the key of the last surface on the left (equal keys share the plane)
            at = n_left >= 1u && n_left < m ? B.skey[cur][axis][a + n_left - 1u] : 0.0;
        } else {
            const uint32_t per_axis = B.buckets - 1u;
            axis = idx / per_axis;
            const double step = (bb[3 + axis] - bb[axis]) / (double)B.buckets;
            at = bb[axis] + ((double)(idx % per_axis + 1u) * step);
        }
    }
    if (n_left == 0u || n_left == m) {
        atomicMax(&B.ctl[CTL_ERR], (uint32_t)ERR_EMPTY_SIDE);
        return;
    }
    const uint32_t node_r = node + 2u * n_left;
    B.node_right[node] = node_r;
    B.node_prim[node] = NONE;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        B.node_bmin[(size_t)k * B.n_nodes + node] = bb[k];
        B.node_bmax[(size_t)k * B.n_nodes + node] = bb[3 + k];
    }
    const uint32_t nxt = cur ^ 1u;
    uint32_t child[2] = {NONE, NONE};
    const uint32_t lo[2] = {a, a + n_left}, hi[2] = {a + n_left, b}, idx_of[2] = {node + 1u, node_r};
    // segments of the next level: one atomic per wave
    const bool opens[2] = {hi[0] - lo[0] >= 2u, hi[1] - lo[1] >= 2u};
    const unsigned long long m0 = __ballot(opens[0]), m1 = __ballot(opens[1]), here = __ballot(1);
    const uint32_t leader = (uint32_t)__ffsll((long long)here) - 1u, wanted = (uint32_t)(__popcll(m0) + __popcll(m1));
    uint32_t slot = 0;
    if (lane() == leader && wanted) slot = atomicAdd(&B.ctl[ctl_count(level + 1u)], wanted);
    slot = __shfl(slot, leader);
    const unsigned long long below = (1ull << lane()) - 1ull;
    slot += (uint32_t)(__popcll(m0 & below) + __popcll(m1 & below));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (opens[c]) {
            const uint32_t j = slot++;
            B.seg_a[nxt][j] = lo[c];
            B.seg_b[nxt][j] = hi[c];
            B.seg_rec[nxt][j] = make_uint4(lo[c], hi[c] - lo[c], 0u, 0u);
            B.seg_node[nxt][j] = idx_of[c];
            B.best_cost[nxt][best_slot(j)] = NO_COST;
            B.best_pick[nxt][j] = ~0ull;
            child[c] = j;
        } else {
            B.leaf_node[lo[c]] = idx_of[c];
        }
    }
    B.seg_axis[s] = axis;
    B.seg_plane[s] = at;
    B.seg_nleft[s] = n_left | (strict << 31);
    B.seg_cut[s] = make_uint4(n_left | (strict << 31), axis, (uint32_t)__double2loint(at), (uint32_t)__double2hiint(at));
    B.seg_child[0][s] = child[0];
    B.seg_child[1][s] = child[1];
}

__device__ __forceinline__ bool passes(const DBuild& B, uint32_t axis, double at, uint32_t strict, uint32_t id) {
    const double c = B.cen[(size_t)axis * B.n + id];
    return strict ? c < at : c <= at;
}

// Splits the three sorted orders stably and lists the pairs that partition_in_place swaps in the surface order.  Where
// an element goes follows from the number of passing elements of its segment before it: a segmented count over the four
// lists in two launches (APPLY = false: predicates and tile totals; true: what the tiles before carry in, folded back to
// the nearest segment start, then counts and moves).
template <bool APPLY>
__global__ void __launch_bounds__(BLOCK) k_split(DBuild B, uint32_t cur, uint32_t level, SCount* totals) {
    __shared__ SCount lds[scan::WAVES];
    if (B.ctl[ctl_count(level)] == 0u || B.ctl[CTL_ERR] != 0u) return;
    constexpr uint32_t ITEMS = scan::ITEMS;
    const CountOp op;
    const uint32_t n = B.n, tile = blockIdx.x, n_tiles = gridDim.x, nxt = cur ^ 1u;
    const uint32_t base = tile * TILE + threadIdx.x * ITEMS;
    uint32_t sg[ITEMS + 2];
    load_segments<ITEMS>(B, base, sg);
    SCount x[ITEMS];
    uint32_t bits[ITEMS];
    static_assert(scan::ITEMS == 4, "the four flags of a thread's positions travel as one word");
    uint32_t ids[4][ITEMS];   // the thread's entries of {order, sid[0], sid[1], sid[2]}
    if (!APPLY) load_run<ITEMS>(B.order[cur], base, n, ids[0]);
#pragma unroll
    for (uint32_t ax = 0; ax < 3; ++ax) load_run<ITEMS>(B.sid[cur][ax], base, n, ids[ax + 1u]);
    uint32_t packed = 0;
    if (APPLY && base < n) packed = *reinterpret_cast<const uint32_t*>(B.flags + base);
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t pos = base + i, s = sg[i + 1u];
        bits[i] = 0;
        x[i] = op.identity();
        if (pos >= n) continue;
        if (s == NONE) {
            x[i].f = 1;
            continue;
        }
        if (!APPLY) {
            const uint4 cut = B.seg_cut[s];
            const uint32_t axis = cut.y, strict = cut.x >> 31;
            const double at = __hiloint2double((int)cut.w, (int)cut.z);
            bits[i] = (uint32_t)passes(B, axis, at, strict, ids[0][i]);
            // the list sorted on the split axis holds its key next to the surface: no look-up of the centroid
            const double own_key = B.skey[cur][axis][pos];
#pragma unroll
            for (uint32_t ax = 0; ax < 3; ++ax) {
                const bool pass = ax == axis ? (strict ? own_key < at : own_key <= at) : passes(B, axis, at, strict, ids[ax + 1u][i]);
                bits[i] |= (uint32_t)pass << (ax + 1u);
            }
            packed |= bits[i] << (8u * i);
        } else {
            bits[i] = (packed >> (8u * i)) & 0xFFu;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) x[i].c[k] = (bits[i] >> k) & 1u;
        x[i].f = sg[i] != s;
    }
    SCount total;
    if (!APPLY) {
        if (base < n) *reinterpret_cast<uint32_t*>(B.flags + base) = packed;   // (padded like every table)
        block_scan_dir<false, ITEMS>(op, x, op.identity(), lds, total);
        if (threadIdx.x == 0) totals[tile] = total;
        return;
    }
    const SCount carry = scan::look_back<false>(op, totals, tile, n_tiles, lds);
    block_scan_dir<false, ITEMS>(op, x, carry, lds, total);
    double keys[3][ITEMS];
#pragma unroll
    for (uint32_t ax = 0; ax < 3; ++ax) {
        const double* key = B.skey[cur][ax];
        if (base + ITEMS <= n) {
            const double2 k01 = *reinterpret_cast<const double2*>(key + base), k23 = *reinterpret_cast<const double2*>(key + base + 2u);
            keys[ax][0] = k01.x; keys[ax][1] = k01.y; keys[ax][2] = k23.x; keys[ax][3] = k23.y;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; ++i) keys[ax][i] = base + i < n ? key[base + i] : 0.0;
        }
    }
    uint32_t passed[ITEMS] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; ++i) {
        const uint32_t pos = base + i, s = sg[i + 1u];
        if (pos >= n || s == NONE) continue;
        const uint4 rec = B.seg_rec[cur][s];
        const uint32_t a = rec.x, b = a + rec.y, n_left = B.seg_cut[s].x & 0x7FFFFFFFu, r = pos - a;
        // The split position comes from a search in the sorted keys, the counts from the predicate: they agree for
        // finite keys.  Nothing is written outside the segment if they ever do not; the build then fails.
        if (sg[i + 2u] != s && (x[i].c[0] != n_left || x[i].c[1] != n_left || x[i].c[2] != n_left || x[i].c[3] != n_left))
            atomicMax(&B.ctl[CTL_ERR], (uint32_t)ERR_INTERNAL);
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            const uint32_t passing = (bits[i] >> (ax + 1)) & 1u, before = x[i].c[ax + 1] - passing;
            uint32_t to = passing ? a + before : a + n_left + (r - before);
            to = to < b ? to : b - 1u;
            B.sid[nxt][ax][to] = ids[ax + 1][i];
            B.skey[nxt][ax][to] = keys[ax][i];
        }
        const uint32_t passing = bits[i] & 1u, passed_before = x[i].c[0] - passing;
        passed[i] = passed_before;
        if (pos < a + n_left) {
            if (!passing) B.F[a + (r - passed_before < b - a ? r - passed_before : 0u)] = pos;
        } else if (passing) {
            const uint32_t j = n_left - (passed_before + 1u);
            B.G[a + (j < b - a ? j : 0u)] = pos;
        }
    }
    if (base < n) *reinterpret_cast<uint4*>(B.before + base) = make_uint4(passed[0], passed[1], passed[2], passed[3]);
}

// what the host reads after a level
__device__ __forceinline__ void report_level(const DBuild& B, uint32_t level) {
    B.report[0] = B.ctl[ctl_count(level)];
    B.report[1] = B.ctl[ctl_count(level + 1u)];
    B.report[2] = B.ctl[CTL_ERR];
}
// The surface order after partition_in_place, and the segment of every position at the next level.  Runs at every level
// the host enqueues, also one past the last: the order moves to the other buffer either way.
__global__ void __launch_bounds__(BLOCK) k_reorder(DBuild B, uint32_t cur, uint32_t level) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos == 0u) report_level(B, level);
    if (pos >= B.n || B.ctl[CTL_ERR] != 0u) return;
    const uint32_t nxt = cur ^ 1u;
    const uint32_t s = B.seg_at[pos];
    uint32_t from = pos;
    if (s != NONE) {
        const uint32_t a = B.seg_a[cur][s], n_left = B.seg_nleft[s] & 0x7FFFFFFFu, r = pos - a;
        const uint32_t passed_before = B.before[pos], f = B.flags[pos] & 1u;
        if (pos < a + n_left) {
            if (!f) from = B.G[a + (r - passed_before)];
        } else if (f) {
            from = B.F[a + (n_left - (passed_before + 1u))];
        }
        B.seg_at[pos] = B.seg_child[pos < a + n_left ? 0 : 1][s];
    }
    const uint32_t id = B.order[cur][from];
    B.order[nxt][pos] = id;
    B.pos_of[id] = pos;
}

// LongestAxisBisectSlice: every segment is sorted by the lower box corner on
// its longest axis (partition.rs:58-68), ties keeping their current order.
// Two stable radix sorts over all positions do that for every segment at
// once: by the coordinate, then by the first position of the element's
// segment (a finished leaf counts as a segment of its own), which puts every
// segment back onto its own positions.
__global__ void __launch_bounds__(BLOCK) k_bisect_coordinate_keys(DBuild B, uint32_t cur, unsigned long long* keys) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n) return;
    const uint32_t s = B.ctl[CTL_ERR] != 0u ? NONE : B.seg_at[pos];
    keys[pos] = s == NONE ? 0ull : ordered_key(B.box[6ull * B.order[cur][pos] + B.seg_axis[s]]);
}
__global__ void __launch_bounds__(BLOCK) k_bisect_segment_keys(DBuild B, uint32_t cur, const uint32_t* by_coordinate, uint32_t* keys) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= B.n) return;
    if (B.ctl[CTL_ERR] != 0u) {
        keys[j] = j;
        return;
    }
    const uint32_t pos = B.pos_of[by_coordinate[j]];
    const uint32_t s = B.seg_at[pos];
    keys[j] = s == NONE ? pos : B.seg_a[cur][s];
}
__global__ void __launch_bounds__(BLOCK) k_bisect_finish(DBuild B, uint32_t cur, uint32_t level) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos == 0u) report_level(B, level);
    if (pos >= B.n || B.ctl[CTL_ERR] != 0u) return;
    B.pos_of[B.order[cur ^ 1u][pos]] = pos;
    const uint32_t s = B.seg_at[pos];
    if (s != NONE) B.seg_at[pos] = B.seg_child[pos < B.seg_a[cur][s] + (B.seg_nleft[s] & 0x7FFFFFFFu) ? 0 : 1][s];
}

__global__ void __launch_bounds__(BLOCK) k_init(DBuild B, unsigned long long* sort_keys, uint32_t* sort_vals) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B.n) return;
    const double* b = B.box + 6ull * i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double c = b[k] + (b[3 + k] - b[k]) * 0.5;   // AABB::centroid, src/aabb.rs:29-31
        B.cen[(size_t)k * B.n + i] = c;
        sort_keys[(size_t)k * B.n + i] = ordered_key(c);
    }
    sort_vals[i] = i;
    B.order[0][i] = i;
    B.pos_of[i] = i;
    B.seg_at[i] = B.n >= 2u ? 0u : NONE;
    B.leaf_node[i] = 0;
    if (i == 0) {
        B.seg_a[0][0] = 0;
        B.seg_b[0][0] = B.n;
        B.seg_rec[0][0] = make_uint4(0u, B.n, 0u, 0u);
        B.seg_node[0][0] = 0;
        B.best_cost[0][0] = NO_COST;
        B.best_pick[0][0] = ~0ull;
        for (uint32_t k = 0; k < CTL_WORDS; ++k) B.ctl[k] = 0;
        B.ctl[ctl_count(0)] = B.n >= 2u ? 1u : 0u;   // the root is the one segment of level 0
    }
}
__global__ void __launch_bounds__(BLOCK) k_sorted_keys(DBuild B) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x, axis = blockIdx.y;
    if (pos < B.n) B.skey[0][axis][pos] = B.cen[(size_t)axis * B.n + B.sid[0][axis][pos]];
}

// Node::Leaf for every position of the final order (src/bvh/mod.rs:97-98,104-113)
__global__ void __launch_bounds__(BLOCK) k_leaves(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n) return;
    const uint32_t node = B.leaf_node[pos];
    B.node_right[node] = NONE;
    B.node_prim[node] = pos;
    const double* b = B.box + 6ull * B.order[cur][pos];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        B.node_bmin[(size_t)k * B.n_nodes + node] = b[k];
        B.node_bmax[(size_t)k * B.n_nodes + node] = b[3 + k];
    }
}
// ---------------------------------------------------------------- host side
struct Arena {
    char* base = nullptr;
    size_t size = 0, used = 0;
    template <class T>
    T* take(size_t count) {
        used = (used + 255u) & ~(size_t)255u;
        T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
        used += count * sizeof(T);
        return p;
    }
};

struct Plan {
    DBuild B;
    unsigned long long* sort_keys_in;
    unsigned long long* sort_keys_out;
    uint32_t* sort_vals;
    SBox* box_totals;
    SCount* count_totals;
    void* sort_tmp;
    size_t sort_tmp_bytes;
    size_t zeroed_bytes;   // the tables at the head of the arena that start as zeros
};

// Carves every table out of `ar`; with ar.base == nullptr only adds up the sizes.
void carve(Arena& ar, Plan& p, uint32_t n, size_t sort_tmp_bytes) {
    DBuild& B = p.B;
    const uint32_t n_tiles = (n + TILE - 1u) / TILE, sah_tiles = (n + SAH_TILE - 1u) / SAH_TILE;
    const bool per_surface = B.strategy == RTX_SAH_PER_SURFACE, equal_size = B.strategy == RTX_SAH_EQUAL_SIZE;
    // first the tables whose entries are used as positions, surface or segment numbers: they start as zeros (see the caller)
    for (int c = 0; c < 2; ++c) {
        B.order[c] = ar.take<uint32_t>(n);
        for (int ax = 0; ax < 3; ++ax) B.sid[c][ax] = ar.take<uint32_t>(n);
        B.seg_a[c] = ar.take<uint32_t>(n);
        B.seg_b[c] = ar.take<uint32_t>(n);
        B.seg_node[c] = ar.take<uint32_t>(n);
        B.seg_child[c] = ar.take<uint32_t>(n);
        B.seg_rec[c] = ar.take<uint4>(n);
        B.best_pick[c] = ar.take<unsigned long long>(n);
    }
    B.pos_of = ar.take<uint32_t>(n);
    B.seg_at = ar.take<uint32_t>(n);
    B.seg_cut = ar.take<uint4>(n);
    B.seg_axis = ar.take<uint32_t>(n);
    B.seg_nleft = ar.take<uint32_t>(n);
    B.flags = ar.take<uint8_t>(n);
    B.before = ar.take<uint32_t>(n);
    B.F = ar.take<uint32_t>(n);
    B.G = ar.take<uint32_t>(n);
    B.ctl = ar.take<uint32_t>(CTL_WORDS);
    B.report = ar.take<uint32_t>(REPORT_WORDS);
    B.leaf_node = ar.take<uint32_t>(n);
    p.sort_vals = ar.take<uint32_t>(n);
    p.zeroed_bytes = ar.used;
    // then coordinates, costs and boxes
    B.box = ar.take<double>(6ull * n);
    B.cen = ar.take<double>(3ull * n);
    for (int c = 0; c < 2; ++c) {
        for (int ax = 0; ax < 3; ++ax) B.skey[c][ax] = ar.take<double>(n);
        B.best_cost[c] = ar.take<unsigned long long>((size_t)n + BEST_EXTRA);
    }
    B.seg_plane = ar.take<double>(n);
    B.cand_cost = ar.take<unsigned long long>(3ull * B.n_pad);
    // scanned boxes in memory: the x order forward for the strategies that split by the segment's box, all six for the
    // planes of EqualSize, none for PerSurface
    B.L = per_surface ? nullptr : ar.take<double>((equal_size ? 18ull : 6ull) * n);
    B.R = equal_size ? ar.take<double>(18ull * n) : nullptr;
    B.seg_box = per_surface ? ar.take<double>(6ull * n) : nullptr;
    B.node_right = ar.take<uint32_t>(B.n_nodes);
    B.node_prim = ar.take<uint32_t>(B.n_nodes);
    B.node_bmin = ar.take<double>(3ull * B.n_nodes);
    B.node_bmax = ar.take<double>(3ull * B.n_nodes);
    p.sort_keys_in = ar.take<unsigned long long>(3ull * n);
    p.sort_keys_out = ar.take<unsigned long long>(n);
    p.box_totals = ar.take<SBox>(6ull * (sah_tiles > n_tiles ? sah_tiles : n_tiles));
    p.count_totals = ar.take<SCount>(n_tiles);
    p.sort_tmp = ar.take<char>(sort_tmp_bytes);
    p.sort_tmp_bytes = sort_tmp_bytes;
}

#define BUILD_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            msg = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            code = e_ == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP;                 \
            goto done;                                                                    \
        }                                                                                 \
    } while (0)

}  // namespace build

// How many levels the host enqueues beyond the last one whose segment count it has seen.  The device keeps the segment
// counts; the host reads a level's report two levels late, so the stream never runs dry while it waits, and a tree that
// is complete costs one level of launches that find nothing to do.
constexpr uint32_t LEVELS_AHEAD = 2, REPORT_RING = 8;

int build_bvh_device(const double* leaf_boxes, uint32_t n, int strategy, uint32_t buckets, uint32_t* leaf_order, double* node_bmin,
                     double* node_bmax, uint32_t* node_right, uint32_t* node_prim, rtx_build_stats* stats, std::string& msg) {
    using namespace build;
    int code = RTX_OK;
    Plan plan;
    std::memset(&plan, 0, sizeof plan);
    Arena ar;
    hipStream_t st = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipEvent_t level_done[REPORT_RING] = {};
    uint32_t* report = nullptr;   // pinned, REPORT_RING copies of the level report
    std::vector<uint32_t> level_segments;   // segments (= internal nodes) of every level whose report has been read
    uint32_t cur = 0, level = 0, err = 0;
    const uint32_t grid = (n + BLOCK - 1u) / BLOCK, n_tiles = (n + TILE - 1u) / TILE;
    size_t sort_tmp_bytes = 0;
    DBuild& B = plan.B;
    bool complete = n < 2u;
    // waits for the report of the next level not yet read; false once that level turns out to be the last (or failed)
    auto read_report = [&](hipError_t& e) -> bool {
        const uint32_t seen = (uint32_t)level_segments.size();
        e = hipEventSynchronize(level_done[seen % REPORT_RING]);
        if (e != hipSuccess) return false;
        const uint32_t* r = report + (size_t)(seen % REPORT_RING) * REPORT_WORDS;
        level_segments.push_back(r[0]);
        err = r[2];
        complete = r[1] == 0u;
        return err == 0u && !complete;
    };

    B.n = n;
    B.n_pad = (n + 3u) & ~3u;
    B.strategy = (uint32_t)strategy;
    B.buckets = buckets;
    B.n_nodes = 2u * n - 1u;
    BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(nullptr, sort_tmp_bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr,
                                                   (uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64, st));
    {
        size_t bytes32 = 0;
        BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(nullptr, bytes32, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                                       (uint32_t*)nullptr, n, 0, 32, st));
        if (bytes32 > sort_tmp_bytes) sort_tmp_bytes = bytes32;
    }
    carve(ar, plan, n, sort_tmp_bytes);   // sizes only
    ar.size = ar.used + 256u;
    BUILD_TRY(hipMalloc((void**)&ar.base, ar.size));
    ar.used = 0;
    carve(ar, plan, n, sort_tmp_bytes);
    BUILD_TRY(hipHostMalloc((void**)&report, REPORT_RING * REPORT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    for (auto& e : ev) BUILD_TRY(hipEventCreate(&e));
    for (auto& e : level_done) BUILD_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    BUILD_TRY(hipEventRecord(ev[0], st));
    // The tables of positions and numbers start as zeros: a level that the host enqueues after one that failed finds tables
    // the failed level never wrote, and although its kernels look at the error word first, nothing that could serve as an
    // index may ever come from stale memory.
    BUILD_TRY(hipMemsetAsync(ar.base, 0, plan.zeroed_bytes, st));
    BUILD_TRY(hipMemcpyAsync(const_cast<double*>(B.box), leaf_boxes, 6ull * n * sizeof(double), hipMemcpyHostToDevice, st));

    hipLaunchKernelGGL(k_init, dim3(grid), dim3(BLOCK), 0, st, B, plan.sort_keys_in, plan.sort_vals);
    if (strategy != RTX_BISECT) {
        for (int ax = 0; ax < 3; ++ax)
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, plan.sort_keys_in + (size_t)ax * n,
                                                           plan.sort_keys_out, plan.sort_vals, B.sid[0][ax], n, 0, 64, st));
        hipLaunchKernelGGL(k_sorted_keys, dim3(grid, 3), dim3(BLOCK), 0, st, B);
    }
    BUILD_TRY(hipEventRecord(ev[1], st));

    while (!complete && err == 0u) {
        if (level > n) {   // a tree over n surfaces has fewer levels than that
            msg = "rtx_bvh_build: the device never reported the last level";
            code = RTX_ERR_BUILD;
            goto done;
        }
        if (level >= LEVELS_AHEAD) {
            hipError_t e = hipSuccess;
            const bool go_on = read_report(e);
            BUILD_TRY(e);
            if (!go_on) break;
        }
        // a level holds at most 2^level segments, and no more than n/2 (a segment has two surfaces or more)
        const uint32_t seg_bound = level < 31u && (1u << level) < n / 2u ? 1u << level : n / 2u;
        const uint32_t seg_grid = (seg_bound + BLOCK - 1u) / BLOCK;
        if (strategy == RTX_SAH_PER_SURFACE) {
            const uint32_t sah_tiles = (n + SAH_TILE - 1u) / SAH_TILE;
            hipLaunchKernelGGL(k_sah_per_surface<false>, dim3(sah_tiles, 3), dim3(BLOCK), 0, st, B, cur, level, plan.box_totals);
            hipLaunchKernelGGL(k_box_spine, dim3(6), dim3(BLOCK), 0, st, sah_tiles, plan.box_totals);
            hipLaunchKernelGGL(k_segment_areas, dim3(seg_grid), dim3(BLOCK), 0, st, B, cur, level, plan.box_totals);
            hipLaunchKernelGGL(k_sah_per_surface<true>, dim3(sah_tiles, 3), dim3(BLOCK), 0, st, B, cur, level, plan.box_totals);
            hipLaunchKernelGGL(k_pick_per_surface, dim3(grid, 3), dim3(BLOCK), 0, st, B, cur, level);
        } else {
            // a midpoint split needs the segment boxes only (x order, forward); the planes of EqualSize need all six folds
            run_scan(BoxScan{B, cur}, n, strategy == RTX_SAH_EQUAL_SIZE ? 6u : 1u, plan.box_totals, st);
            if (strategy == RTX_SAH_EQUAL_SIZE) {
                const unsigned long long threads = (unsigned long long)seg_bound * 3ull * (buckets - 1u);
                const uint32_t cand_grid = (uint32_t)((threads + BLOCK - 1u) / BLOCK);
                hipLaunchKernelGGL(k_candidates_equal_size<0>, dim3(cand_grid), dim3(BLOCK), 0, st, B, cur, level);
                hipLaunchKernelGGL(k_candidates_equal_size<1>, dim3(cand_grid), dim3(BLOCK), 0, st, B, cur, level);
            }
        }
        hipLaunchKernelGGL(k_decide, dim3(seg_grid), dim3(BLOCK), 0, st, B, cur, level);
        if (strategy == RTX_BISECT) {
            hipLaunchKernelGGL(k_bisect_coordinate_keys, dim3(grid), dim3(BLOCK), 0, st, B, cur, plan.sort_keys_in);
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, plan.sort_keys_in, plan.sort_keys_out, B.order[cur],
                                                           plan.sort_vals, n, 0, 64, st));
            hipLaunchKernelGGL(k_bisect_segment_keys, dim3(grid), dim3(BLOCK), 0, st, B, cur, plan.sort_vals, B.F);
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, B.F, B.G, plan.sort_vals, B.order[cur ^ 1u], n, 0,
                                                           32, st));
            hipLaunchKernelGGL(k_bisect_finish, dim3(grid), dim3(BLOCK), 0, st, B, cur, level);
        } else {
            hipLaunchKernelGGL(k_split<false>, dim3(n_tiles), dim3(BLOCK), 0, st, B, cur, level, plan.count_totals);
            hipLaunchKernelGGL(k_split<true>, dim3(n_tiles), dim3(BLOCK), 0, st, B, cur, level, plan.count_totals);
            hipLaunchKernelGGL(k_reorder, dim3(grid), dim3(BLOCK), 0, st, B, cur, level);
        }
        BUILD_TRY(hipMemcpyAsync(report + (size_t)(level % REPORT_RING) * REPORT_WORDS, B.report, REPORT_WORDS * sizeof(uint32_t),
                                 hipMemcpyDeviceToHost, st));
        BUILD_TRY(hipEventRecord(level_done[level % REPORT_RING], st));
        cur ^= 1u;
        ++level;
    }
    // the levels enqueued after the last report read: they hold no segments, unless the tree is shallower than LEVELS_AHEAD
    while (err == 0u && level_segments.size() < level) {
        hipError_t e = hipSuccess;
        (void)read_report(e);
        BUILD_TRY(e);
    }
    if (err != 0u) {
        // the reference panics with this message (partition.rs:219) or, for an empty side, indexes out of range / never returns
        msg = err == ERR_NO_PLANE     ? "No valid splitting plane"
              : err == ERR_EMPTY_SIDE ? "partition left one side empty"
                                      : "rtx_bvh_build: split position and predicate disagree";
        code = RTX_ERR_BUILD;
        goto done;
    }
    hipLaunchKernelGGL(k_leaves, dim3(grid), dim3(BLOCK), 0, st, B, cur);
    BUILD_TRY(hipEventRecord(ev[2], st));
    BUILD_TRY(hipMemcpyAsync(leaf_order, B.order[cur], (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipMemcpyAsync(node_right, B.node_right, (size_t)B.n_nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipMemcpyAsync(node_prim, B.node_prim, (size_t)B.n_nodes * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipMemcpyAsync(node_bmin, B.node_bmin, 3ull * B.n_nodes * sizeof(double), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipMemcpyAsync(node_bmax, B.node_bmax, 3ull * B.n_nodes * sizeof(double), hipMemcpyDeviceToHost, st));
    BUILD_TRY(hipStreamSynchronize(st));
    BUILD_TRY(hipGetLastError());
    if (stats) {
        float ms_sort = 0.f, ms_levels = 0.f;
        BUILD_TRY(hipEventElapsedTime(&ms_sort, ev[0], ev[1]));
        BUILD_TRY(hipEventElapsedTime(&ms_levels, ev[1], ev[2]));
        uint32_t levels = 0;
        for (uint32_t c : level_segments) levels += c != 0u;
        stats->levels = levels;
        stats->n_nodes = B.n_nodes;
        stats->ms_sort = ms_sort;
        stats->ms_levels = ms_levels;
        stats->device_bytes = ar.size;
    }
done:
    if (ar.base) (void)hipStreamSynchronize(st);   // nothing of this call is still running when its tables go
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : level_done)
        if (e) (void)hipEventDestroy(e);
    if (report) (void)hipHostFree(report);
    if (ar.base) (void)hipFree(ar.base);
    return code;
}

}  // namespace rtx
