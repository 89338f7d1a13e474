// rtx_device.h -- device-side data layout and arithmetic of the render path.
//
// Everything here is f64 and must be compiled with -ffp-contract=off: the
// reference (Rust) never fuses a*b+c, and closest-hit ids are compared
// bit-for-bit against a CPU restatement of it.  Citations "file:line" are
// into the reference repository seridescent/raytracing.
#pragma once

#include <hip/hip_runtime.h>
#include <cstddef>
#include <stdint.h>

namespace rtx {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// ---------------------------------------------------------------- layout in HBM
// One BVH node = one 64-byte record, so a lane's node fetch touches exactly
// one cache line whatever node it is on.  `miss` threads the pre-order array:
// it is the next node in pre-order outside this node's subtree, i.e. exactly
// the node the reference's explicit stack (src/bvh/mod.rs:135-156) would pop
// next after this subtree is finished or culled.
struct __attribute__((aligned(64))) DNode {
    double bmin[3];   // the traverse kernel reads the record as 3 x double2 + 1 x uint4
    double bmax[3];
    uint32_t miss;   // next node when the box is missed or the leaf is done
    uint32_t prim;   // leaf: prim index; internal: NONE
    uint32_t kind;   // leaf: geometry kind
    uint32_t slot;   // leaf: index into the sphere / planar table
};
static_assert(sizeof(DNode) == 64, "node record must be one 64-byte line");

// Single-precision copy of a node for the first tier of the box test: the box
// planes rounded to f32, interleaved per axis so that one packed multiply-add
// yields both slab distances of an axis, and the two links.  32 bytes: two
// 16-byte loads per visit, half of the f64 record.
struct __attribute__((aligned(32))) DNode32 {
    float plane[6];   // lo.x, hi.x, lo.y, hi.y, lo.z, hi.z
    uint32_t miss;    // DNode::miss as the byte offset of that node's record (index * 32)
    uint32_t down;    // where a ray whose box test passes goes: the next record (own offset + 32), with NODE32_LEAF set in a
                      // leaf's (the ray then waits there for the surface test; a leaf's successor is the next record as well).
                      // The step "hit ? down : miss" needs no test of what the node is, and no addition.
};
// LDS bytes of a tree's f32 records and, behind them, its leaf codes (padded to a 16-byte boundary)
__host__ __device__ constexpr size_t node_tables_lds_bytes(uint32_t n_nodes) {
    return (size_t)n_nodes * 32u + (((size_t)n_nodes * 4u + 15u) & ~(size_t)15u);
}
constexpr uint32_t NODE32_LEAF = 0x80000000u;   // records lie below 2^30 bytes (checked when the scene is created)
// The surface code of a leaf is read after the walk's node loop, once per waiting lane, from DScene::leaf_codes (by node index).
static_assert(sizeof(DNode32) == 32, "f32 node record must be 32 bytes");
// Half-size copy of a node for trees that are walked from global memory (far larger than LDS and L2).  There the
// walk is bound by how many lines the memory system can deliver per second (one request per visit that misses L2,
// whatever its size), so the records are made small enough for eight of them to share a 128-byte line: the box planes
// as 16-bit steps of a per-scene grid (plane = q_org + q * q_delta per axis, rounded to the nearest step) and ONE link
// word.  In pre-order a leaf's successor is the next record, so a leaf stores its surface code and an internal node
// (tag 3 in the two top bits, which no surface code has) the index of its successor.
struct __attribute__((aligned(16))) DNode16 {
    uint16_t q[6];    // lo.x, hi.x, lo.y, hi.y, lo.z, hi.z
    uint32_t link;    // leaf: surface code (make_code); internal: 0xC0000000 | index of the successor record
};
static_assert(sizeof(DNode16) == 16, "quantised node record must be 16 bytes");
constexpr uint32_t NODE16_INTERNAL = 0xC0000000u;
// The exact box of a node, for the rare visits the f32 tier cannot decide.
struct __attribute__((aligned(16))) DBox64 {
    double bmin[3];
    double bmax[3];
};

struct __attribute__((aligned(32))) DSphere {   // Geometry::Sphere, src/geometry.rs:27-30
    double c[3];
    double r;
};
struct __attribute__((aligned(128))) DPlanar {  // Geometry::Quadrilateral / Triangle, src/geometry.rs:31-46
    double q[3], u[3], v[3], norm[3], w[3];
    double d;
};
struct DPrimInfo {
    uint32_t kind, slot, material, user_id;
};
struct DMaterial {                              // enum Material, src/material.rs:9-17
    double rgb[3];
    double param;
    uint32_t kind;
    uint32_t _pad;
};
// Everything the shade kernel needs about the surface that won, in one
// 64-byte record per prim so it arrives with one dependent round trip:
//   sphere: a = center, s = 1.0 / radius  (`(p - c) / r` is `(p - c) * (1.0 / r)`, src/vector.rs:179-190)
//   planar: a = norm,   s unused
// plus the material's colour and parameter.  Geometry and material KIND do not
// need a load at all: they travel in the surface code.
//   dielectric: rgb = (1 / ri, schlick_r0(1 / ri), schlick_r0(ri)): the two divisions of a glass bounce, done once
__host__ __device__ inline double schlick_r0(double eta) {   // src/material.rs:135-136
    double r0 = (1.0 - eta) / (1.0 + eta);
    return r0 * r0;
}
struct DShadeRec {
    double a[3];
    double s;
    double rgb[3];
    double param;
};
// Surface code: how leaves name their surface and what the closest-hit
// kernels write as the hit.  Geometry kind in bits 31-30, material kind in
// bits 29-27, slot in the table of its geometry kind (spheres / planars) in
// the low 27 bits.  NONE (all ones) cannot collide: geometry kinds stop at 2.
// The shade kernel therefore knows both kinds without a load, and finds the
// surface's shade record at index slot (spheres) or n_spheres + slot (planars).
constexpr uint32_t CODE_SLOT_MASK = 0x07FFFFFFu;
__host__ __device__ inline uint32_t make_code(uint32_t geom_kind, uint32_t mat_kind, uint32_t slot) {
    return (geom_kind << 30) | (mat_kind << 27) | slot;
}
__host__ __device__ inline uint32_t code_geom(uint32_t c) { return c >> 30; }
__host__ __device__ inline uint32_t code_mat(uint32_t c) { return (c >> 27) & 7u; }
__host__ __device__ inline uint32_t code_slot(uint32_t c) { return c & CODE_SLOT_MASK; }

struct DScene {
    const DNode* nodes;
    const DSphere* spheres;
    const DPlanar* planars;
    uint32_t n_nodes;
    uint32_t filter_ok;   // every box coordinate is finite and small enough for the filtered box test
    uint32_t batch_undecided;   // box tests the f32 tier leaves open are resolved for many lanes at once, after the node walk
    uint32_t pend_at;           // the walk runs the waiting lanes' surface tests once this many lanes wait
    uint32_t refill_at;         // ... and gives idle lanes their next rays once this many lanes are idle
    uint32_t bounce_at;         // kernel with the paths in registers: one bounce once this many lanes have finished their walk or hold no path
    const DNode32* nodes32;
    const uint32_t* leaf_codes;    // per node: the surface code of a leaf (make_code), NONE for an internal node
    const DBox64* boxes64;
    uint32_t n_spheres, n_planars;
    const DShadeRec* shade_recs;   // spheres by slot, then planars by slot
    const uint32_t* rec_user_id;   // same order: index of the surface in the caller's array
    const DNode16* nodes16;        // quantised records, or nullptr when the scene has none
    double q_org[3];               // grid of the quantised planes: plane = q_org + q * q_delta
    float q_delta[3];
    uint32_t use16;                // the global-memory walk reads nodes16 instead of nodes32
    uint32_t assume_in_count;      // counting instances of the walk take the shortcuts of the product's walk (they count its tests, not the reference's)
};

// State of the paths in flight, struct-of-arrays.  Entry i of every array
// belongs to the i-th live path of the current round; the shade kernel writes
// the survivors of a round densely into the other buffer (stream compaction),
// so every kernel reads and writes these arrays with consecutive lanes on
// consecutive addresses.
struct DState {
    double *ox, *oy, *oz, *dx, *dy, *dz;   // current ray (src/ray.rs:4-7)
    double *bx, *by, *bz;                  // product of attenuations so far
    uint32_t *pixel, *sample, *draw, *depth;
};
// Closest-hit record of ray i of the current round.
struct DHits {
    uint32_t* prim;   // surface code (make_code) of the hit, or NONE
    double* t;
};

struct DCamera {   // InitializedCamera, src/camera.rs:28-43
    uint32_t image_width, image_height, samples_per_pixel, max_depth;
    double defocus_angle;
    double background[3];
    double center[3], pixel00_loc[3], pixel_du[3], pixel_dv[3], defocus_disk_u[3], defocus_disk_v[3];
};

// The path pool is split into one private sub-pool per wave of the (fixed)
// launch grid: wave w of every kernel works on slots
// [w * sub_capacity, w * sub_capacity + counts[w]).  A wave compacts its own
// survivors into its own sub-pool (of the other buffer in the round loop, in place in the single launch), in order, so no two
// waves ever write the same counter or reserve slots from a shared one: the
// round loop has no atomics besides the radiance accumulation.  (A returning
// atomic on a single address sustains fewer than 100 operations per
// microsecond on this part; shared counters throttled every earlier layout.)
// Order-preserving compaction also keeps rays that started at neighbouring
// pixels next to each other, which the closest-hit kernel rewards.
struct DPool {
    const uint32_t* counts;   // rays per sub-pool; nullptr: one flat batch of flat_total rays cut into sub_capacity pieces
    uint32_t n_sub;           // == number of waves of every launch on this pool
    uint32_t sub_capacity;    // multiple of 64
    uint32_t flat_total;
};

// Control block of one render call (device memory).  Words that one kernel
// reads while another word of the same meaning is written are kept twice,
// indexed by round parity.
struct DRenderCtl {
    unsigned long long next_path[2];   // next unstarted (pixel, sample) index
    unsigned long long total_paths;
    unsigned long long rays;           // closest-hit queries issued so far
    unsigned long long node_visits, sphere_tests, planar_tests;
    unsigned long long box_fallbacks;  // box tests the f32 tier left undecided
    uint32_t live_total;               // rays of the current round over all sub-pools (the host polls this)
    uint32_t rounds_max;               // single-launch render: most rounds any wave went through
    unsigned long long band_next[8];   // single-launch render with the image cut into bands (one per XCD): next unstarted path of each band
};

// Everything the single-launch render kernel needs, in device memory: the
// kernel and its phases take one pointer and load what they use, instead of
// holding some 120 scalar registers of launch arguments for their whole life.
struct DRenderParams {
    DScene sc;
    DCamera cam;
    DState st[2];
    DHits hits;
    DRenderCtl* ctl;
    double* accum;
    unsigned long long seed;
    double tmin, tmax0;
    uint32_t n_sub, cap, spp_begin, lds_surfaces;
    uint32_t lds_recs;      // the shade records are in LDS as well (behind the surface tables)
    uint32_t tile_order;    // path indices name pixels block by block (tiled_pixel) instead of row by row
    uint32_t tail_in_registers;   // the last row of rays of a wave finishes in registers (phase_tail)
    uint32_t resident_batch;      // resident form: primary rays a wave makes and stages at a time (64, or fewer where LDS is short)
    uint32_t xcd_bands;           // pooled form: the image in this many bands with a path counter each (0 or 1: one counter for the whole image)
};

// ---------------------------------------------------------------- Vector3 (src/vector.rs)
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return a * s; }
__device__ __forceinline__ V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
// Vector3 / f64 multiplies by the reciprocal (src/vector.rs:179-190)
__device__ __forceinline__ V3 operator/(V3 a, double s) {
    double inv = 1.0 / s;
    return v3(a.x * inv, a.y * inv, a.z * inv);
}
__device__ __forceinline__ double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ double length(V3 a) { return sqrt(length_squared(a)); }
__device__ __forceinline__ V3 to_unit(V3 a) { return a / length(a); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ bool is_near_zero(V3 a) { return fabs(a.x) < 1e-8 && fabs(a.y) < 1e-8 && fabs(a.z) < 1e-8; }
// f64::clamp: a NaN argument is returned unchanged
__device__ __forceinline__ double clampd(double x, double lo, double hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }   // src/vector.rs:254-256
__device__ __forceinline__ V3 refract(V3 r_in, V3 n, double eta) {                       // src/vector.rs:258-264
    double cos_theta = clampd(dot(-r_in, n), -1.0, 1.0);
    V3 r_out_perp = eta * (r_in + cos_theta * n);
    V3 r_out_par = -sqrt(fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_par;
}
__device__ __forceinline__ V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }

// ---------------------------------------------------------------- RNG
// Philox4x32 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3") with seven rounds: the fewest for which
// the paper reports the generator Crush-resistant (its default of ten adds a safety margin that a renderer's sample
// streams do not need: 1.3 % of the frame).  The 32x32->64 products are written as one 64-bit multiply each so that a
// single v_mad_u64_u32 yields both halves.
constexpr int PHILOX_ROUNDS = 7;
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < PHILOX_ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // a ^ b ^ c in one instruction (v_bitop3_b32, truth table 0x96); the compiler emits two v_xor_b32 otherwise
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Random numbers of one path.  Stream = (seed, pixel index, sample index):
// block b is Philox4x32-7 with counter (pixel, sample, b, 0) and the 64-bit
// seed as key; it yields the 64-bit words w0 = out[1]:out[0], w1 = out[3]:out[2].
//   * a uniform double takes one word: (w >> 11) * 2^-53 (draw i is word
//     i & 1 of block i >> 1);
//   * one trial of the unit-vector rejection sampler takes one whole block:
//     three 42-bit uniforms (top 42 bits of w0, top 42 bits of w1, and the
//     low 22 bits of w0 followed by the low 20 bits of w1).
// Every sampling event (the camera sample, each scatter call) starts on a
// block boundary, so the draw index is the only RNG state a path carries
// between kernels.
struct PathRng {
    uint32_t k0, k1, pixel, sample, draw, cached_block;
    double cached[2];
    __device__ __forceinline__ PathRng(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t draw_)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), pixel(pixel_), sample(sample_), draw(draw_), cached_block(NONE) {
        cached[0] = cached[1] = 0.0;
    }
    __device__ __forceinline__ void begin_event() { draw = (draw + 1u) & ~1u; }
    __device__ __forceinline__ double next() {
        uint32_t blk = draw >> 1;
        if (blk != cached_block) {
            uint32_t o[4];
            philox4x32(pixel, sample, blk, 0u, k0, k1, o);
            uint64_t w0 = ((uint64_t)o[1] << 32) | o[0];
            uint64_t w1 = ((uint64_t)o[3] << 32) | o[2];
            cached[0] = (double)(w0 >> 11) * 0x1.0p-53;
            cached[1] = (double)(w1 >> 11) * 0x1.0p-53;
            cached_block = blk;
        }
        double r = (draw & 1u) ? cached[1] : cached[0];
        ++draw;
        return r;
    }
    __device__ __forceinline__ double range(double lo, double hi) { return next() * (hi - lo) + lo; }
    // three uniforms in [0,1) with 42-bit resolution from one block
    __device__ __forceinline__ void next3(double& a, double& b, double& c) {
        begin_event();
        uint32_t o[4];
        philox4x32(pixel, sample, draw >> 1, 0u, k0, k1, o);
        draw += 2u;
        const uint64_t w0 = ((uint64_t)o[1] << 32) | o[0];
        const uint64_t w1 = ((uint64_t)o[3] << 32) | o[2];
        a = (double)(w0 >> 22) * 0x1.0p-42;
        b = (double)(w1 >> 22) * 0x1.0p-42;
        c = (double)(((w0 & 0x3FFFFFull) << 20) | (w1 & 0xFFFFFull)) * 0x1.0p-42;
    }
};

// src/vector.rs:59-81: rejection from the cube, one Philox block per trial
__device__ __forceinline__ V3 random_unit(PathRng& rng) {
    for (;;) {
        double a, b, c;
        rng.next3(a, b, c);
        V3 candidate = v3(a * 2.0 + -1.0, b * 2.0 + -1.0, c * 2.0 + -1.0);   // random_range(-1.0..1.0)
        double lensq = length_squared(candidate);
        if (1e-160 < lensq && lensq <= 1.0) return candidate / sqrt(lensq);
    }
}
// The same two samplers on a bare draw index (which must be even: a sampling event starts on a block
// boundary), for callers that keep no PathRng: candidate of one trial, and the first uniform of a block.
__device__ __forceinline__ V3 random_unit_candidate(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk) {
    uint32_t o[4];
    philox4x32(pixel, sample, blk, 0u, k0, k1, o);
    const uint64_t w0 = ((uint64_t)o[1] << 32) | o[0];
    const uint64_t w1 = ((uint64_t)o[3] << 32) | o[2];
    // u = m * 2^-42 for a 42-bit integer m, then random_range(-1.0..1.0) = u * 2.0 + -1.0.  Both steps are exact, so
    // the same number comes from placing m in the mantissa of d = 1 + u (in [1, 2)) and one fused 2 d - 3:
    // four instructions per coordinate instead of nine (integer-to-double conversions of 64-bit values are slow).
    const uint64_t ma = w0 >> 22, mb = w1 >> 22, mc = ((w0 & 0x3FFFFFull) << 20) | (w1 & 0xFFFFFull);
    const double da = __longlong_as_double((long long)(0x3FF0000000000000ull | (ma << 10)));
    const double db = __longlong_as_double((long long)(0x3FF0000000000000ull | (mb << 10)));
    const double dc = __longlong_as_double((long long)(0x3FF0000000000000ull | (mc << 10)));
    return v3(fma(da, 2.0, -3.0), fma(db, 2.0, -3.0), fma(dc, 2.0, -3.0));
}
__device__ __forceinline__ double first_uniform_of_block(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk) {
    uint32_t o[4];
    philox4x32(pixel, sample, blk, 0u, k0, k1, o);
    const uint64_t w0 = ((uint64_t)o[1] << 32) | o[0];
    return (double)(w0 >> 11) * 0x1.0p-53;
}
// Both uniforms of block `blk` of a stream (draws 2*blk and 2*blk + 1)
__device__ __forceinline__ void uniforms_of_block(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk, double& u0, double& u1) {
    uint32_t o[4];
    philox4x32(pixel, sample, blk, 0u, k0, k1, o);
    const uint64_t w0 = ((uint64_t)o[1] << 32) | o[0];
    const uint64_t w1 = ((uint64_t)o[3] << 32) | o[2];
    u0 = (double)(w0 >> 11) * 0x1.0p-53;
    u1 = (double)(w1 >> 11) * 0x1.0p-53;
}
// src/vector.rs:83-95
__device__ __forceinline__ V3 random_in_unit_disk(PathRng& rng) {
    for (;;) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        V3 candidate = v3(x, y, 0.0);
        if (length_squared(candidate) < 1.0) return candidate;
    }
}

// ---------------------------------------------------------------- box test
// AABB::hit, src/aabb.rs:55-76, with six true divisions.  The guard at :62
// uses `==` against NaN and can never fire, so it has no counterpart here.
__device__ __forceinline__ bool box_fold(double t0x, double t1x, double t0y, double t1y, double t0z, double t1z,
                                         double tmin, double tmax, double& lowers_max, double& uppers_min) {
    double lx = clampd(fmin(t0x, t1x), tmin, tmax), ux = clampd(fmax(t0x, t1x), tmin, tmax);
    double ly = clampd(fmin(t0y, t1y), tmin, tmax), uy = clampd(fmax(t0y, t1y), tmin, tmax);
    double lz = clampd(fmin(t0z, t1z), tmin, tmax), uz = clampd(fmax(t0z, t1z), tmin, tmax);
    // fold(NaN, max) / fold(NaN, min): NaN operands drop out
    lowers_max = fmax(fmax(lx, ly), lz);
    uppers_min = fmin(fmin(ux, uy), uz);
    return lowers_max < uppers_min;
}

__device__ __forceinline__ bool box_hit_exact(const double bmin[3], const double bmax[3], V3 o, V3 d, double tmin, double tmax) {
    double lo, hi;
    return box_fold((bmin[0] - o.x) / d.x, (bmax[0] - o.x) / d.x, (bmin[1] - o.y) / d.y, (bmax[1] - o.y) / d.y,
                    (bmin[2] - o.z) / d.z, (bmax[2] - o.z) / d.z, tmin, tmax, lo, hi);
}
// out-of-line copy for the filtered test's rare fallback, so the six
// divisions stay out of the hot loop's instruction stream and registers
// (box passed by value: pointer arguments would force the caller's copy into scratch memory)
__device__ __noinline__ bool box_hit_exact_slow(double b0, double b1, double b2, double b3, double b4, double b5, V3 o, V3 d,
                                                double tmin, double tmax) {
    const double bmin[3] = {b0, b1, b2}, bmax[3] = {b3, b4, b5};
    return box_hit_exact(bmin, bmax, o, d, tmin, tmax);
}

// Filtered form of the same predicate.
//
// With NaN-free operands clamp commutes with max/min (it is monotone), so the
// reference's  max_i clamp(lower_i) < min_i clamp(upper_i)  equals, for
// lo* = max_i lower_i and hi* = min_i upper_i and tmin < tmax,
//     hi* > tmin  &&  lo* < tmax  &&  lo* < hi*.
// The filter evaluates lo, hi with q~ = (b - o) * RN(1/d) instead of
// RN((b - o)/d).  q~ differs from the reference quotient by at most
// 3 * 2^-53 relative (two roundings against one) and max/min preserve that
// bound, so each of the three comparisons is decided for certain when its two
// sides are further apart than the 2^-50 relative margin; if any comparison
// that matters is inside the margin (about one test in 1e13) the six true
// divisions are done instead.  Preconditions, checked once per ray by
// filter_ray_ok: finite origin, every component of d finite, nonzero and
// within [2^-300, 2^300] (so no quotient is NaN or infinite), box and origin
// coordinates at most 2^300, 2^-900 <= tmin < tmax.  Rays outside them take
// the division form for every node.
constexpr double FILTER_EPS = 0x1.0p-50;

struct FilterBounds {   // tmin, tmax widened / narrowed by the margin
    double tmin_lo, tmin_hi, tmax_lo, tmax_hi;
};
__device__ __forceinline__ void filter_set_tmin(FilterBounds& f, double tmin) {
    f.tmin_lo = tmin * (1.0 - FILTER_EPS);
    f.tmin_hi = tmin * (1.0 + FILTER_EPS);
}
__device__ __forceinline__ void filter_set_tmax(FilterBounds& f, double tmax) {
    f.tmax_lo = tmax * (1.0 - FILTER_EPS);
    f.tmax_hi = tmax * (1.0 + FILTER_EPS);
}
// 1 = hit for certain, 0 = miss for certain, -1 = undecided
__device__ __forceinline__ int box_hit_filtered(const double bmin[3], const double bmax[3], V3 o, V3 inv, const FilterBounds& f) {
    double t0x = (bmin[0] - o.x) * inv.x, t1x = (bmax[0] - o.x) * inv.x;
    double t0y = (bmin[1] - o.y) * inv.y, t1y = (bmax[1] - o.y) * inv.y;
    double t0z = (bmin[2] - o.z) * inv.z, t1z = (bmax[2] - o.z) * inv.z;
    double lo = fmax(fmax(fmin(t0x, t1x), fmin(t0y, t1y)), fmin(t0z, t1z));
    double hi = fmin(fmin(fmax(t0x, t1x), fmax(t0y, t1y)), fmax(t0z, t1z));
    double diff = hi - lo;
    double margin = FILTER_EPS * (fabs(hi) + fabs(lo));
    bool hit = hi > f.tmin_hi && lo < f.tmax_lo && diff > margin;
    bool miss = hi < f.tmin_lo || lo > f.tmax_hi || -diff > margin;
    return hit ? 1 : (miss ? 0 : -1);
}

__device__ __forceinline__ bool filter_comp_ok(double d) {
    double a = fabs(d);
    return a >= 0x1.0p-300 && a <= 0x1.0p300;
}
__device__ __forceinline__ bool filter_ray_ok(V3 o, V3 d, double tmin, double tmax) {
    return filter_comp_ok(d.x) && filter_comp_ok(d.y) && filter_comp_ok(d.z) && fabs(o.x) <= 0x1.0p300 &&
           fabs(o.y) <= 0x1.0p300 && fabs(o.z) <= 0x1.0p300 && tmin >= 0x1.0p-900 && tmin < tmax;
}

// ---------------------------------------------------------------- f32 first tier
// The same predicate (hi* > tmin, lo* < tmax, lo* < hi*), first tried in
// single precision.  Write u = 2^-24.  With
//     b32 = RN32(b),  inv32 = recip32(RN32(d)),  oi32 = RN32(RN32(o) * inv32)
// (recip32: within (1 + 2^-18)u of the reciprocal, see there; so inv32 = (1/d)(1 + di), |di| <= 2.001u, and
// oi32 = (o/d)(1 + do)(1 + di)(1 + dm)),
// the slab distance t~ = RN32(b32 * inv32 - oi32), one fused operation, satisfies
//     |t~ - Q| <= |Q| 3.002u + |b/d| 1.001u + |o/d| 2.002u,        Q = (b - o)/d,
// and with |b/d| <= |Q| + |o/d|, |Q| <= |t~| + |t~ - Q|, |o/d| <= |oi32|(1 + 3.01u):
//     |t~ - Q| <= 4.01u (|t~| + |oi32|).
// The reference's own quotient is RN64 of the difference over d, within
// 2^-52 |Q| of Q.  Max and min over such values keep the bound of the value they
// return: with c = 4.01u and OI >= every |oi32| of the ray, x + c|x| and x - c|x|
// grow with x, so  max_i (t~_i + c|t~_i|) = lo~ + c|lo~|  and the exact lower
// fold lies within c (|lo~| + OI) of lo~ = max of the three near distances
// (each near distance, a min of two, by the same argument), whatever the other
// distances of the node are; the same for hi~.  Clamping at tmin32 / tmax32 only
// removes error (a fold beyond the clamp by more than its error is the clamp
// exactly, one within it has the clamp's magnitude).  So with
// M = max(|lo~ clamped|, |hi~ clamped|) the folds are off by at most
//     E = 4.05u (M + OI) + 1e-30
// (the absolute term covers planes and products below the f32 normal range).
// (Until the bound was taken from lo~ and hi~ themselves M was the largest of the
// node's six |t~|: two instructions more per visit and a wider margin.)
// Let s1 = hi - tmin32, s2 = tmax32 - lo, s3 = hi - lo and
// m = RN32(min(hi, tmax32) - max(lo, tmin32)) = RN32(min(s1, s2, s3, tmax32 - tmin32));
// the fourth term is never negative (tmin <= tmax), so it cannot make m
// negative and only makes "true" harder to prove for intervals shorter than T.
// Rounding tmin, tmax to f32 and the subtraction cost at most
// 2^-22 max(tmin, tmax) more, so with M' = max(M, tg), tg = tmax32 if finite
// else tmin32, and
//     T = 12.2u M' + (8.1u OI + 2e-30)   >=   2E + 2^-22 tg (1 + small),
// m > T proves the predicate true and m < -T proves it false (some s_k is then
// negative beyond its error).  Anything else, including any NaN or infinity
// on the way, leaves the visit undecided and it is repeated in f64.  To keep
// every f32 operation finite and the rounding model valid, a ray takes this
// tier only if 1e-30 <= |inv32| <= 1e15 and |oi32| <= 1e15 on every axis and
// 1e-30 <= tmin <= 1e15, and a scene only if every box plane is at most 1e15
// in magnitude; otherwise K is +infinity and every visit is undecided.
// 1/x for the tier's rays: the hardware's approximate reciprocal (v_rcp_f32, documented to 1 ulp; 2^-21 relative would
// do) and one Newton step.  With e0 = 1 - x r0, |e0| <= 2^-21:  e = RN32(1 - x r0) (one fused operation) = e0 (1 + d1),
// r1 = RN32(r0 + e r0) = RN32((1/x)(1 - e0)(1 + e0 (1 + d1))) = RN32((1/x)(1 - e0^2 + e0 d1 (1 - e0))), so
// |r1 - 1/x| <= (1/x)(u + 2^-42 + 2^-45) for x and 1/x in the normal range (the tier admits 1e-30 <= |1/x| <= 1e15 only;
// zero, infinite and NaN arguments give 0, infinity or NaN, which that check rejects).  Three instructions per
// component instead of the ten of a correctly rounded division, at the refill of a wave's idle lanes, where they
// are issued for half a wave.
__device__ __forceinline__ float recip32(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
struct RayF32 {
    float inv[3], oi[3];   // oi holds +o/d: t~ = fma(b32, inv, -oi)
    float K;               // 8.1u * OI + 2e-30, or +inf when the ray skips this tier
    float tmin32, tmax32, tg;
};
__device__ __forceinline__ void rayf32_set_tmax(RayF32& r, double tmax) {
    const float t = (float)tmax;
    r.tmax32 = t;
    r.tg = t <= 3.0e38f ? t : r.tmin32;   // tmax >= tmin always
}
__device__ __forceinline__ void rayf32_init(RayF32& r, V3 o, V3 d, double tmin, double tmax) {
    const float dv[3] = {(float)d.x, (float)d.y, (float)d.z};
    const float ov[3] = {(float)o.x, (float)o.y, (float)o.z};
    bool ok = true;
    float oi_max = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        r.inv[a] = recip32(dv[a]);
        r.oi[a] = ov[a] * r.inv[a];
        const float ai = fabsf(r.inv[a]), ao = fabsf(r.oi[a]);
        ok = ok && ai >= 1e-30f && ai <= 1e15f && ao <= 1e15f;   // NaN fails every comparison
        oi_max = fmaxf(oi_max, ao);
    }
    const float t = (float)tmin;
    ok = ok && t >= 1e-30f && t <= 1e15f && tmin <= tmax;   // box_hit_f32 relies on tmin <= tmax
    r.tmin32 = t;
    r.K = ok ? oi_max * (8.1f * 0x1.0p-24f) + 2e-30f : __builtin_inff();
    rayf32_set_tmax(r, tmax);
}
// The same tier on the quantised records (DNode16).  The plane of a record is b = q_org + q delta + eps with
// |eps| <= delta / 2 (1 + 1e-4) (checked plane by plane when the scene is created), so with o' = o - q_org
//     Q = (b - o) / d = (q delta - o') / d + eps / d.
// Per ray  s = RN32(RN32(delta) inv32)  and  c = RN32(RN32(o') inv32)  (relative errors at most 4.01u each: see the
// f32 tier for inv32), and the slab distance is the single fused t~ = RN32(q s - c); q < 2^16 is exact in f32.  Then
//     |t~ - (q delta - o')/d| <= u |t~| + 4.01u |q delta / d| + 4.01u |o'/d|  <=  5.03u |t~| + 8.04u |c|,
// using |q delta / d| <= |t~| + |t~ - ..| + |o'/d| and |o'/d| <= |c| (1 + 4.01u), and so with M = the larger of the two
// clamped folds' magnitudes (a fold keeps the relative bound of the value it returns, as in the f32 tier),
// C >= every |c| of the ray and D = max over the axes of RN32(delta) |inv32| (>= 2 |eps / d| / 1.0003):
//     E = 5.05u M + 8.05u C + 0.5002 D + 1e-30     bounds the error of lo and hi.
// The rest is the f32 tier's argument with this E: T = 14.2u M' + (16.2u C + 1.001 D + 2e-30) >= 2E + the rounding of
// tmin, tmax and of the subtractions (4.1u M' as there).  m > T proves the reference's predicate true, m < -T false.
// The test itself is the f32 tier's (box_hit_tier1) with the steps converted to float as "planes", s as inv, c as oi, and this T.
constexpr float TIER_F32_COEF = 12.2f * 0x1.0p-24f;
constexpr float TIER_Q16_COEF = 14.2f * 0x1.0p-24f;
__device__ __forceinline__ void rayq16_init(RayF32& r, V3 o, V3 d, double tmin, double tmax, const double q_org[3], const float q_delta[3]) {
    const float dv[3] = {(float)d.x, (float)d.y, (float)d.z};
    const float ov[3] = {(float)(o.x - q_org[0]), (float)(o.y - q_org[1]), (float)(o.z - q_org[2])};   // RN32 of the f64 difference
    bool ok = true;
    float c_max = 0.0f, d_max = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = recip32(dv[a]);
        r.inv[a] = q_delta[a] * inv;              // s
        r.oi[a] = ov[a] * inv;                    // c
        const float ai = fabsf(inv), ac = fabsf(r.oi[a]), as = fabsf(r.inv[a]);
        // every product in the normal range (a grid without extent on an axis has delta = 0 exactly: s = 0 is then exact)
        ok = ok && ai >= 1e-30f && ai <= 1e15f && ac <= 1e15f && (q_delta[a] == 0.0f || (as >= 1e-30f && as <= 1e30f));
        c_max = fmaxf(c_max, ac);
        d_max = fmaxf(d_max, as);
    }
    const float t = (float)tmin;
    ok = ok && t >= 1e-30f && t <= 1e15f && tmin <= tmax;
    r.tmin32 = t;
    r.K = ok ? fmaf(c_max, 16.2f * 0x1.0p-24f, fmaf(d_max, 1.001f, 2e-30f)) : __builtin_inff();
    rayf32_set_tmax(r, tmax);
}
// Returns whether the answer is certain; `hit` is the answer when it is.  p: the six planes as f32 (Q16 = false) or
// the six grid steps converted to f32 (Q16 = true), in the record's order.
template <bool Q16>
__device__ __forceinline__ bool box_hit_tier1(const float p[6], const RayF32& r, bool& hit) {
    const float t0x = fmaf(p[0], r.inv[0], -r.oi[0]), t1x = fmaf(p[1], r.inv[0], -r.oi[0]);
    const float t0y = fmaf(p[2], r.inv[1], -r.oi[1]), t1y = fmaf(p[3], r.inv[1], -r.oi[1]);
    const float t0z = fmaf(p[4], r.inv[2], -r.oi[2]), t1z = fmaf(p[5], r.inv[2], -r.oi[2]);
    const float nx = fminf(t0x, t1x), ny = fminf(t0y, t1y), nz = fminf(t0z, t1z);
    const float fx = fmaxf(t0x, t1x), fy = fmaxf(t0y, t1y), fz = fmaxf(t0z, t1z);
    const float lo = fmaxf(fmaxf(fmaxf(nx, ny), nz), r.tmin32), hi = fminf(fminf(fminf(fx, fy), fz), r.tmax32);   // the folds, clamped
    const float M = fmaxf(fmaxf(fabsf(lo), fabsf(hi)), r.tg);
    const float T = fmaf(M, Q16 ? TIER_Q16_COEF : TIER_F32_COEF, r.K);
    const float m = hi - lo;   // = min(s1, s2, s3, tmax32 - tmin32), one subtraction
    hit = m > T;
    return fabsf(m) > T;   // T >= 0 (M >= 0: -mn or mx is; K >= 0), so this is m > T || m < -T in one comparison
}

__device__ __forceinline__ bool box_hit_f32(const float p[6], const RayF32& r, bool& hit) { return box_hit_tier1<false>(p, r, hit); }

// ---------------------------------------------------------------- primitive tests
// sphere::hit, src/geometry.rs:204-230 (the t part; the rest of Hit is completed for the winner only)
__device__ __forceinline__ bool sphere_t(const DSphere& s, V3 o, V3 d, double tmin, double tmax, double& t) {
    V3 oc = v3(s.c[0], s.c[1], s.c[2]) - o;
    double a = length_squared(d);
    double h = dot(d, oc);
    double c = length_squared(oc) - s.r * s.r;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = sqrt(discriminant);
    double root = (h - sqrtd) / a;
    if (!(tmin < root && root < tmax)) {       // Interval::surrounds, open
        root = (h + sqrtd) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t = root;
    return true;
}
// The same test for a ray whose a = d.d and 1/a were prepared when the ray was taken up (ray_div_init): the walk tests
// two spheres per ray on average, for a quarter of a wave at a time, and a test is mostly its square root and its two
// divisions by a.
//
// Division and square root here are the compiler's own correctly rounded sequences with the range scaffolding left out:
//   n / x:   r = rcp(x), two Newton steps on r (refined_rcp, once per ray), q0 = n r, rem = fma(-x, q0, n), q = fma(rem, r, q0);
//   sqrt(x): y = rsq(x), g = x y, h = y / 2, one coupled step on (g, h), two corrections g += (x - g g) h.
// The compiler wraps these in v_div_scale / v_div_fmas / v_div_fixup and ldexp / class tests, which move operands of
// extreme magnitude into range by powers of two (exact) and patch zeros, infinities and NaNs.  With every operand and
// intermediate in the normal range the bare sequence computes the same roundings, hence the same bits.  That holds
// when 2^-300 <= a <= 2^300 (else ra = 0 marks the ray), 2^-500 <= disc < 2^500 and |h| < 2^300: then sqrtd is in
// [2^-250, 2^250], h -+ sqrtd is zero or at least 2^-53 max(|h|, sqrtd) >= 2^-303 in magnitude, and quotients, residuals
// and their products stay between 2^-700 and 2^700.  A test outside those bounds takes sphere_t's library forms
// (a GPU test compares both sequences with the host's division and square root bit for bit).
struct RayDiv {
    double a, ra;   // d.d and its refined reciprocal, or ra = 0: use the library division
};
__device__ __forceinline__ double refined_rcp(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e0 = fma(-x, r0, 1.0);
    const double r1 = fma(r0, e0, r0);
    const double e1 = fma(-x, r1, 1.0);
    return fma(r1, e1, r1);
}
__device__ __forceinline__ double div_refined(double n, double x, double rx) {
    const double q0 = n * rx;
    const double rem = fma(-x, q0, n);
    return fma(rem, rx, q0);
}
__device__ __forceinline__ double sqrt_normal_range(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, x);
    g = fma(e, h, g);
    e = fma(-g, g, x);
    return fma(e, h, g);
}
__device__ __forceinline__ void ray_div_init(RayDiv& rd, V3 d) {
    rd.a = length_squared(d);
    rd.ra = (rd.a >= 0x1.0p-300 && rd.a <= 0x1.0p300) ? refined_rcp(rd.a) : 0.0;
}
__device__ __forceinline__ bool sphere_t_ray(const DSphere& s, V3 o, V3 d, const RayDiv& rd, double tmin, double tmax, double& t) {
    V3 oc = v3(s.c[0], s.c[1], s.c[2]) - o;
    const double a = rd.a;
    double h = dot(d, oc);
    double c = length_squared(oc) - s.r * s.r;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    // The bounds as integer tests on the high words (biased exponents 523 .. 1523 hold 2^-500 .. 2^500; a NaN or a negative
    // number fails the first, an infinite or NaN h the second), and one choice for all the lanes that are here: the library
    // forms are right for every operand, so a wave with a single lane outside the bounds takes them for all.
    const uint32_t disc_hi = (uint32_t)__double2hiint(discriminant), h_hi = (uint32_t)__double2hiint(h) & 0x7FFFFFFFu;
    const bool bare = (uint32_t)__double2hiint(rd.ra) != 0u && disc_hi - (523u << 20) < (1000u << 20) && h_hi < (1323u << 20);
    double sqrtd, root;
    if (__builtin_amdgcn_ballot_w64(!bare) == 0ull) {   // wave-uniform
        sqrtd = sqrt_normal_range(discriminant);
        root = div_refined(h - sqrtd, a, rd.ra);
        if (!(tmin < root && root < tmax)) {       // Interval::surrounds, open
            root = div_refined(h + sqrtd, a, rd.ra);
            if (!(tmin < root && root < tmax)) return false;
        }
    } else {
        sqrtd = sqrt(discriminant);
        root = (h - sqrtd) / a;
        if (!(tmin < root && root < tmax)) {
            root = (h + sqrtd) / a;
            if (!(tmin < root && root < tmax)) return false;
        }
    }
    t = root;
    return true;
}
// plane_intersection + uv_hit + quad/triangle::hit, src/geometry.rs:147-190,264-327
__device__ __forceinline__ bool planar_t(const DPlanar& g, uint32_t kind, V3 o, V3 d, double tmin, double tmax,
                                         double& t, double& alpha, double& beta) {
    V3 norm = ld3(g.norm);
    double denominator = dot(norm, d);
    if (fabs(denominator) < 1e-10) return false;
    double tt = (g.d - dot(norm, o)) / denominator;
    if (!(tmin <= tt && tt <= tmax)) return false;   // Interval::contains, closed
    V3 p = o + d * tt;
    V3 qp = p - ld3(g.q);
    V3 w = ld3(g.w);
    double al = dot(w, cross(qp, ld3(g.v)));
    double be = dot(w, cross(ld3(g.u), qp));
    if (kind == 1u) {   // quad: Interval::UNIT.contains on both
        if (!(0.0 <= al && al <= 1.0) || !(0.0 <= be && be <= 1.0)) return false;
    } else {            // triangle
        if (!(al >= 0.0 && be >= 0.0 && al + be <= 1.0)) return false;
    }
    t = tt; alpha = al; beta = be;
    return true;
}

}  // namespace rtx
