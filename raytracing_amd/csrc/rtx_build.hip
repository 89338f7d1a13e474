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

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/rtx.h"
#include "rtx_kernels.h"
#include "rtx_scan.h"

namespace rtx {
namespace build {

using scan::BLOCK;
using scan::TILE;
using scan::run_scan;
constexpr unsigned long long NO_COST = ~0ull;

enum : uint32_t { ERR_NO_PLANE = 1, ERR_EMPTY_SIDE = 2, ERR_INTERNAL = 3 };

struct DBuild {
    uint32_t n, strategy, buckets, n_nodes;
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
    uint32_t* seg_axis;      // decision per segment of this level
    double* seg_plane;
    uint32_t* seg_nleft;     // bit 31: predicate is `<` (midpoint) instead of `<=`
    uint32_t* seg_child[2];  // segment index of the left / right child at the next level, NONE for a leaf
    unsigned long long* best_cost;
    uint32_t* best_idx;
    unsigned long long* cand_cost;   // [3][n]
    double* L;               // [3][n][6] box of the segment's sorted surfaces up to and including a position
    double* R;               // [3][n][6] ... from a position to the segment's end
    uint4* flags;            // predicate of the element at a position of {order, sid[0], sid[1], sid[2]}
    uint4* P;                // [n+1] exclusive prefix sums of flags
    uint32_t* F;             // positions of the failing elements of a segment's left part, in order
    uint32_t* G;             // positions of the passing elements of its right part, from the end
    uint32_t* ctl;           // [0] segments of the next level, [1] error
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

// Row y = axis + 3*direction.  Direction 0 folds the boxes of a segment's
// sorted surfaces from its first position on (left_of, partition.rs:136-150),
// direction 1 from its last position back (right_of, partition.rs:152-166).
struct BoxScan {
    using T = SBox;
    DBuild B;
    uint32_t cur;
    __device__ T identity() const {
        T r;
        r.v[0] = r.v[1] = r.v[2] = __longlong_as_double(0x7FF0000000000000ll);
        r.v[3] = r.v[4] = r.v[5] = __longlong_as_double((long long)0xFFF0000000000000ull);
        r.f = 0;
        return r;
    }
    __device__ T load(uint32_t i, uint32_t y) const {
        const uint32_t axis = y % 3u, back = y / 3u;
        const uint32_t pos = back ? B.n - 1u - i : i;
        const uint32_t s = B.seg_at[pos];
        T r = identity();
        r.f = 1;
        if (s == NONE) return r;   // a finished leaf: a segment of its own, value never read
        const bool edge = back ? pos == B.n - 1u : pos == 0u;
        r.f = (edge || B.seg_at[back ? pos + 1u : pos - 1u] != s) ? 1u : 0u;
        // bisection keeps no sorted lists: the fold over the surface order gives the same box
        const uint32_t* list = B.strategy == RTX_BISECT ? B.order[cur] : B.sid[cur][axis];
        const double* b = B.box + 6ull * list[pos];
#pragma unroll
        for (int k = 0; k < 6; ++k) r.v[k] = b[k];
        return r;
    }
    // AABB::merge, src/aabb.rs:40-53
    __device__ T combine(const T& a, const T& b) const {
        T r;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.v[k] = b.f ? b.v[k] : fmin(a.v[k], b.v[k]);
            r.v[3 + k] = b.f ? b.v[3 + k] : fmax(a.v[3 + k], b.v[3 + k]);
        }
        r.f = a.f | b.f;
        return r;
    }
    __device__ void store(uint32_t i, uint32_t y, const T& inc) const {
        const uint32_t axis = y % 3u, back = y / 3u;
        const uint32_t pos = back ? B.n - 1u - i : i;
        if (B.seg_at[pos] == NONE) return;
        double* out = (back ? B.R : B.L) + ((size_t)axis * B.n + pos) * 6u;
#pragma unroll
        for (int k = 0; k < 6; ++k) out[k] = inc.v[k];
    }
};

struct FlagScan {
    using T = uint4;
    DBuild B;
    __device__ T identity() const { return make_uint4(0, 0, 0, 0); }
    __device__ T load(uint32_t i, uint32_t) const { return B.flags[i]; }
    __device__ T combine(const T& a, const T& b) const { return make_uint4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
    __device__ void store(uint32_t i, uint32_t, const T& inc) const {
        const uint4 f = B.flags[i];
        B.P[i] = make_uint4(inc.x - f.x, inc.y - f.y, inc.z - f.z, inc.w - f.w);
        if (i == B.n - 1u) B.P[B.n] = inc;
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
__device__ __forceinline__ double split_cost(const double* left, uint32_t n_left, const double* right, uint32_t n_right, const double* bb) {
    const double parent = area_factor(bb);
    const double p_left = area_factor(left) / parent;
    const double p_right = area_factor(right) / parent;
    const double cost = 1.0 + p_left * (double)n_left + p_right * (double)n_right;
    // A parent box without area (zero-radius spheres on a line) makes the cost 0/0.  total_cmp orders NaNs by their sign,
    // and the default NaN of the hosts the reference and the host builder run on (x86-64) is the negative one, which
    // min_by ranks lowest; this device's default NaN is the positive one.  Every NaN cost takes the host's form, so
    // that both builders pick the same plane for such a segment.
    return cost != cost ? __longlong_as_double((long long)0xFFF8000000000000ull) : cost;
}

__device__ __forceinline__ double cost_of_split(const DBuild& B, uint32_t axis, uint32_t a, uint32_t b, uint32_t k) {
    const double* left = B.L + ((size_t)axis * B.n + a + k - 1u) * 6u;
    const double* right = B.R + ((size_t)axis * B.n + a + k) * 6u;
    const double* bb = B.L + (size_t)(b - 1u) * 6u;   // the segment's box, from the x order
    return split_cost(left, k, right, (b - a) - k, bb);
}

// atomicMin(&table[s], v) for the lanes with s != NONE; one atomic per wave when all of them share s
__device__ __forceinline__ void segment_min(unsigned long long* table, uint32_t s, unsigned long long v) {
    const bool live = s != NONE && v != NO_COST;
    const unsigned long long mask = __ballot(live);
    if (mask == 0ull) return;
    const uint32_t first = (uint32_t)__ffsll((long long)mask) - 1u;
    const uint32_t s0 = __shfl(s, first);
    if (__ballot(live && s != s0) == 0ull) {
        unsigned long long m = live ? v : NO_COST;
#pragma unroll
        for (uint32_t d = 32; d >= 1; d >>= 1) {
            const unsigned long long o = __shfl_xor(m, d);
            m = o < m ? o : m;
        }
        if (lane() == first) atomicMin(&table[s0], m);
    } else if (live) {
        atomicMin(&table[s], v);
    }
}

// PerSurface (partition.rs:254-266): every surface's centroid coordinate on
// every axis is a candidate plane.  One thread per (axis, position of the
// axis' sorted order).
__global__ void __launch_bounds__(BLOCK) k_candidates_per_surface(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x, axis = blockIdx.y;
    const uint32_t s = pos < B.n ? B.seg_at[pos] : NONE;
    unsigned long long ck = NO_COST;
    if (s != NONE) {
        const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s];
        const double* key = B.skey[cur][axis];
        const double at = key[pos];
        uint32_t k = 0;
        if (pos + 1u < b) k = key[pos + 1u] > at ? pos + 1u - a : n_left_for(key + a, b - a, at);
        if (k) ck = ordered_key(cost_of_split(B, axis, a, b, k));
        B.cand_cost[(size_t)axis * B.n + pos] = ck;
    }
    segment_min(B.best_cost, s, ck);
}
__global__ void __launch_bounds__(BLOCK) k_pick_per_surface(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x, axis = blockIdx.y;
    if (pos >= B.n) return;
    const uint32_t s = B.seg_at[pos];
    if (s == NONE) return;
    const unsigned long long ck = B.cand_cost[(size_t)axis * B.n + pos];
    if (ck == NO_COST || ck != B.best_cost[s]) return;
    atomicMin(&B.best_idx[s], B.pos_of[B.sid[cur][axis][pos]] * 3u + axis);
}

// EqualSize (partition.rs:229-248): buckets-1 planes per axis, evenly spaced
// over the segment's box.  One thread per (segment, axis, plane).
template <int PASS>
__global__ void __launch_bounds__(BLOCK) k_candidates_equal_size(DBuild B, uint32_t cur, uint32_t n_seg) {
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
    if (PASS == 0) atomicMin(&B.best_cost[s], ck);
    else if (ck == B.best_cost[s]) atomicMin(&B.best_idx[s], idx);
}

__global__ void __launch_bounds__(BLOCK) k_reset_best(DBuild B, uint32_t n_seg) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s < n_seg) {
        B.best_cost[s] = NO_COST;
        B.best_idx[s] = NONE;
    }
    if (s == 0) B.ctl[0] = 0;
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
// opens the children (build_tree_rec, src/bvh/mod.rs:92-131).
__global__ void __launch_bounds__(BLOCK) k_decide(DBuild B, uint32_t cur, uint32_t n_seg) {
    const uint32_t s = blockIdx.x * BLOCK + threadIdx.x;
    if (s >= n_seg) return;
    const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s], node = B.seg_node[cur][s], m = b - a;
    const double* bb = B.L + (size_t)(b - 1u) * 6u;
    uint32_t axis = 0, n_left = 0, strict = 0;
    double at = 0.0;
    // until decided otherwise: nothing moves and the segment has no children (the passes after this one also stop on an error)
    B.seg_axis[s] = 0;
    B.seg_plane[s] = 0.0;
    B.seg_nleft[s] = m;
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
        const uint32_t idx = B.best_idx[s];
        if (idx == NONE) {
            atomicMax(&B.ctl[1], (uint32_t)ERR_NO_PLANE);
            return;
        }
        if (B.strategy == RTX_SAH_PER_SURFACE) {
            axis = idx % 3u;
            at = B.cen[(size_t)axis * B.n + B.order[cur][idx / 3u]];
        } else {
            const uint32_t per_axis = B.buckets - 1u;
            axis = idx / per_axis;
            const double step = (bb[3 + axis] - bb[axis]) / (double)B.buckets;
            at = bb[axis] + ((double)(idx % per_axis + 1u) * step);
        }
        n_left = n_left_for(B.skey[cur][axis] + a, m, at);
    }
    if (n_left == 0u || n_left == m) {
        atomicMax(&B.ctl[1], (uint32_t)ERR_EMPTY_SIDE);
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
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (hi[c] - lo[c] >= 2u) {
            const uint32_t j = atomicAdd(&B.ctl[0], 1u);
            B.seg_a[nxt][j] = lo[c];
            B.seg_b[nxt][j] = hi[c];
            B.seg_node[nxt][j] = idx_of[c];
            child[c] = j;
        } else {
            B.leaf_node[lo[c]] = idx_of[c];
        }
    }
    B.seg_axis[s] = axis;
    B.seg_plane[s] = at;
    B.seg_nleft[s] = n_left | (strict << 31);
    B.seg_child[0][s] = child[0];
    B.seg_child[1][s] = child[1];
}

__device__ __forceinline__ bool passes(const DBuild& B, uint32_t axis, double at, uint32_t strict, uint32_t id) {
    const double c = B.cen[(size_t)axis * B.n + id];
    return strict ? c < at : c <= at;
}

__global__ void __launch_bounds__(BLOCK) k_flags(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n) return;
    const uint32_t s = B.seg_at[pos];
    uint4 f = make_uint4(0, 0, 0, 0);
    if (s != NONE && B.ctl[1] == 0u) {
        const uint32_t axis = B.seg_axis[s], strict = B.seg_nleft[s] >> 31;
        const double at = B.seg_plane[s];
        f.x = passes(B, axis, at, strict, B.order[cur][pos]);
        f.y = passes(B, axis, at, strict, B.sid[cur][0][pos]);
        f.z = passes(B, axis, at, strict, B.sid[cur][1][pos]);
        f.w = passes(B, axis, at, strict, B.sid[cur][2][pos]);
    }
    B.flags[pos] = f;
}

// Splits the three sorted orders stably and lists the pairs that
// partition_in_place swaps in the surface order.
__global__ void __launch_bounds__(BLOCK) k_scatter(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n || B.ctl[1] != 0u) return;
    const uint32_t nxt = cur ^ 1u;
    const uint32_t s = B.seg_at[pos];
    if (s == NONE) {
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            B.sid[nxt][ax][pos] = B.sid[cur][ax][pos];
            B.skey[nxt][ax][pos] = B.skey[cur][ax][pos];
        }
        return;
    }
    const uint32_t a = B.seg_a[cur][s], b = B.seg_b[cur][s], n_left = B.seg_nleft[s] & 0x7FFFFFFFu, r = pos - a;
    const uint4 f = B.flags[pos], p = B.P[pos], pa = B.P[a], pb = B.P[b];
    // The split position comes from a search in the sorted keys, the flags from the predicate: they agree for
    // finite keys.  Nothing is written outside the segment if they ever do not; the build then fails.
    if (pb.x - pa.x != n_left || pb.y - pa.y != n_left || pb.z - pa.z != n_left || pb.w - pa.w != n_left) {
        atomicMax(&B.ctl[1], (uint32_t)ERR_INTERNAL);
        return;
    }
    const uint32_t fl[3] = {f.y, f.z, f.w}, before[3] = {p.y - pa.y, p.z - pa.z, p.w - pa.w};
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const uint32_t to = fl[ax] ? a + before[ax] : a + n_left + (r - before[ax]);
        B.sid[nxt][ax][to] = B.sid[cur][ax][pos];
        B.skey[nxt][ax][to] = B.skey[cur][ax][pos];
    }
    const uint32_t passed_before = p.x - pa.x;
    if (pos < a + n_left) {
        if (!f.x) B.F[a + (r - passed_before)] = pos;
    } else if (f.x) {
        B.G[a + (n_left - (passed_before + 1u))] = pos;
    }
}

// The surface order after partition_in_place, and the segment of every position at the next level.
__global__ void __launch_bounds__(BLOCK) k_reorder(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n || B.ctl[1] != 0u) return;
    const uint32_t nxt = cur ^ 1u;
    const uint32_t s = B.seg_at[pos];
    uint32_t from = pos;
    if (s != NONE) {
        const uint32_t a = B.seg_a[cur][s], n_left = B.seg_nleft[s] & 0x7FFFFFFFu, r = pos - a;
        const uint32_t passed_before = B.P[pos].x - B.P[a].x, f = B.flags[pos].x;
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
    const uint32_t s = B.seg_at[pos];
    keys[pos] = s == NONE ? 0ull : ordered_key(B.box[6ull * B.order[cur][pos] + B.seg_axis[s]]);
}
__global__ void __launch_bounds__(BLOCK) k_bisect_segment_keys(DBuild B, uint32_t cur, const uint32_t* by_coordinate, uint32_t* keys) {
    const uint32_t j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= B.n) return;
    const uint32_t pos = B.pos_of[by_coordinate[j]];
    const uint32_t s = B.seg_at[pos];
    keys[j] = s == NONE ? pos : B.seg_a[cur][s];
}
__global__ void __launch_bounds__(BLOCK) k_bisect_finish(DBuild B, uint32_t cur) {
    const uint32_t pos = blockIdx.x * BLOCK + threadIdx.x;
    if (pos >= B.n) return;
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
        B.seg_node[0][0] = 0;
        B.ctl[0] = 0;
        B.ctl[1] = 0;
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
    uint4* flag_totals;
    void* sort_tmp;
    size_t sort_tmp_bytes;
};

// Carves every table out of `ar`; with ar.base == nullptr only adds up the sizes.
void carve(Arena& ar, Plan& p, uint32_t n, size_t sort_tmp_bytes) {
    DBuild& B = p.B;
    const uint32_t n_tiles = (n + TILE - 1u) / TILE;
    B.box = ar.take<double>(6ull * n);
    B.cen = ar.take<double>(3ull * n);
    for (int c = 0; c < 2; ++c) {
        B.order[c] = ar.take<uint32_t>(n);
        for (int ax = 0; ax < 3; ++ax) {
            B.sid[c][ax] = ar.take<uint32_t>(n);
            B.skey[c][ax] = ar.take<double>(n);
        }
        B.seg_a[c] = ar.take<uint32_t>(n);
        B.seg_b[c] = ar.take<uint32_t>(n);
        B.seg_node[c] = ar.take<uint32_t>(n);
        B.seg_child[c] = ar.take<uint32_t>(n);
    }
    B.pos_of = ar.take<uint32_t>(n);
    B.seg_at = ar.take<uint32_t>(n);
    B.seg_axis = ar.take<uint32_t>(n);
    B.seg_plane = ar.take<double>(n);
    B.seg_nleft = ar.take<uint32_t>(n);
    B.best_cost = ar.take<unsigned long long>(n);
    B.best_idx = ar.take<uint32_t>(n);
    B.cand_cost = ar.take<unsigned long long>(3ull * n);
    B.L = ar.take<double>(18ull * n);
    B.R = ar.take<double>(18ull * n);
    B.flags = ar.take<uint4>(n);
    B.P = ar.take<uint4>((size_t)n + 1u);
    B.F = ar.take<uint32_t>(n);
    B.G = ar.take<uint32_t>(n);
    B.ctl = ar.take<uint32_t>(2);
    B.leaf_node = ar.take<uint32_t>(n);
    B.node_right = ar.take<uint32_t>(B.n_nodes);
    B.node_prim = ar.take<uint32_t>(B.n_nodes);
    B.node_bmin = ar.take<double>(3ull * B.n_nodes);
    B.node_bmax = ar.take<double>(3ull * B.n_nodes);
    p.sort_keys_in = ar.take<unsigned long long>(3ull * n);
    p.sort_keys_out = ar.take<unsigned long long>(n);
    p.sort_vals = ar.take<uint32_t>(n);
    p.box_totals = ar.take<SBox>(6ull * n_tiles);
    p.flag_totals = ar.take<uint4>(n_tiles);
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

int build_bvh_device(const double* leaf_boxes, uint32_t n, int strategy, uint32_t buckets, uint32_t* leaf_order, double* node_bmin,
                     double* node_bmax, uint32_t* node_right, uint32_t* node_prim, rtx_build_stats* stats, std::string& msg) {
    using namespace build;
    int code = RTX_OK;
    Plan plan;
    std::memset(&plan, 0, sizeof plan);
    Arena ar;
    hipStream_t st = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    uint32_t cur = 0, n_seg = n >= 2u ? 1u : 0u, levels = 0;
    const uint32_t grid = (n + BLOCK - 1u) / BLOCK;
    size_t sort_tmp_bytes = 0;
    DBuild& B = plan.B;
    uint32_t host_ctl[2] = {0, 0};

    B.n = n;
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
    for (auto& e : ev) BUILD_TRY(hipEventCreate(&e));
    BUILD_TRY(hipEventRecord(ev[0], st));
    BUILD_TRY(hipMemcpyAsync(const_cast<double*>(B.box), leaf_boxes, 6ull * n * sizeof(double), hipMemcpyHostToDevice, st));

    hipLaunchKernelGGL(k_init, dim3(grid), dim3(BLOCK), 0, st, B, plan.sort_keys_in, plan.sort_vals);
    if (strategy != RTX_BISECT) {
        for (int ax = 0; ax < 3; ++ax)
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, plan.sort_keys_in + (size_t)ax * n,
                                                           plan.sort_keys_out, plan.sort_vals, B.sid[0][ax], n, 0, 64, st));
        hipLaunchKernelGGL(k_sorted_keys, dim3(grid, 3), dim3(BLOCK), 0, st, B);
    }
    BUILD_TRY(hipEventRecord(ev[1], st));

    while (n_seg > 0u) {
        const uint32_t seg_grid = (n_seg + BLOCK - 1u) / BLOCK;
        BoxScan box_scan{B, cur};
        // a midpoint split needs the segment boxes only (x order, forward); the SAH needs all six folds
        run_scan(box_scan, n, strategy == RTX_MIDPOINT || strategy == RTX_BISECT ? 1u : 6u, plan.box_totals, st);
        hipLaunchKernelGGL(k_reset_best, dim3(seg_grid), dim3(BLOCK), 0, st, B, n_seg);
        if (strategy == RTX_SAH_PER_SURFACE) {
            hipLaunchKernelGGL(k_candidates_per_surface, dim3(grid, 3), dim3(BLOCK), 0, st, B, cur);
            hipLaunchKernelGGL(k_pick_per_surface, dim3(grid, 3), dim3(BLOCK), 0, st, B, cur);
        } else if (strategy == RTX_SAH_EQUAL_SIZE) {
            const unsigned long long threads = (unsigned long long)n_seg * 3ull * (buckets - 1u);
            const uint32_t cand_grid = (uint32_t)((threads + BLOCK - 1u) / BLOCK);
            hipLaunchKernelGGL(k_candidates_equal_size<0>, dim3(cand_grid), dim3(BLOCK), 0, st, B, cur, n_seg);
            hipLaunchKernelGGL(k_candidates_equal_size<1>, dim3(cand_grid), dim3(BLOCK), 0, st, B, cur, n_seg);
        }
        hipLaunchKernelGGL(k_decide, dim3(seg_grid), dim3(BLOCK), 0, st, B, cur, n_seg);
        if (strategy == RTX_BISECT) {
            hipLaunchKernelGGL(k_bisect_coordinate_keys, dim3(grid), dim3(BLOCK), 0, st, B, cur, plan.sort_keys_in);
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, plan.sort_keys_in, plan.sort_keys_out, B.order[cur],
                                                           plan.sort_vals, n, 0, 64, st));
            hipLaunchKernelGGL(k_bisect_segment_keys, dim3(grid), dim3(BLOCK), 0, st, B, cur, plan.sort_vals, B.F);
            BUILD_TRY((hipError_t)rocprim::radix_sort_pairs(plan.sort_tmp, plan.sort_tmp_bytes, B.F, B.G, plan.sort_vals, B.order[cur ^ 1u], n, 0,
                                                           32, st));
            hipLaunchKernelGGL(k_bisect_finish, dim3(grid), dim3(BLOCK), 0, st, B, cur);
        } else {
            hipLaunchKernelGGL(k_flags, dim3(grid), dim3(BLOCK), 0, st, B, cur);
            run_scan(FlagScan{B}, n, 1u, plan.flag_totals, st);
            hipLaunchKernelGGL(k_scatter, dim3(grid), dim3(BLOCK), 0, st, B, cur);
            hipLaunchKernelGGL(k_reorder, dim3(grid), dim3(BLOCK), 0, st, B, cur);
        }
        BUILD_TRY(hipMemcpyAsync(host_ctl, B.ctl, sizeof host_ctl, hipMemcpyDeviceToHost, st));
        BUILD_TRY(hipStreamSynchronize(st));
        if (host_ctl[1] != 0u) {
            // the reference panics with this message (partition.rs:219) or, for an empty side, indexes out of range / never returns
            msg = host_ctl[1] == ERR_NO_PLANE     ? "No valid splitting plane"
                  : host_ctl[1] == ERR_EMPTY_SIDE ? "partition left one side empty"
                                                  : "rtx_bvh_build: split position and predicate disagree";
            code = RTX_ERR_BUILD;
            goto done;
        }
        n_seg = host_ctl[0];
        cur ^= 1u;
        ++levels;
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
        stats->levels = levels;
        stats->n_nodes = B.n_nodes;
        stats->ms_sort = ms_sort;
        stats->ms_levels = ms_levels;
        stats->device_bytes = ar.size;
    }
done:
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    if (ar.base) (void)hipFree(ar.base);
    return code;
}

}  // namespace rtx
