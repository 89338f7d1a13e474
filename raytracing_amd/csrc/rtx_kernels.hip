// rtx_kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
// One round of the render loop is three launches on one stream:
//   k_generate      start new paths (sample_square + get_ray) behind the
//                   survivors of the previous round until every sub-pool is full
//   k_traverse_f32  closest hit of every live ray over the threaded pre-order
//   / k_traverse    BVH (BVH::hit)
//   k_shade         complete the hit, emitted + scatter, accumulate finished
//                   paths, and write the surviving paths densely and in order
//                   into the other state buffer (64-wide ballot + prefix count)
// Every wave of a launch owns one sub-pool of the path pool (DPool): the
// three kernels use the same grid, and nothing is shared between waves.
// k_scale / k_quantize are the output stage.
//
// Citations "file:line" are into the reference repository.
#include "rtx_device.h"
#include "rtx_kernels.h"

#include <cstdio>

namespace rtx {

constexpr int BLOCK = 256;
// Block size of the closest-hit kernel with the tree in LDS.  Small blocks interleave best with the other
// lane's kernels; five of them (5 waves per SIMD at this kernel's register count) share a CU's LDS.
#ifndef RTX_TRAVERSE_LDS_BLOCK
#define RTX_TRAVERSE_LDS_BLOCK 256
#endif
constexpr int TRAVERSE_LDS_BLOCK = RTX_TRAVERSE_LDS_BLOCK;
constexpr uint32_t MAX_BLOCK_WAVES = TRAVERSE_LDS_BLOCK / 64 > BLOCK / 64 ? TRAVERSE_LDS_BLOCK / 64 : BLOCK / 64;   // waves per block of the largest closest-hit launch

#ifndef RTX_TRAVERSE_WAVES
#define RTX_TRAVERSE_WAVES 1
#endif
#ifndef RTX_SHADE_WAVES
#define RTX_SHADE_WAVES 4
#endif
#ifndef RTX_PERSISTENT_WAVES
#define RTX_PERSISTENT_WAVES 5
#endif
#ifndef RTX_PERSISTENT_BLOCK
#define RTX_PERSISTENT_BLOCK 256
#endif

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
// Ray state and hit records stream through once per bounce: their loads are marked non-temporal, so the lines they bring do not
// displace the surface and material tables (the frame +2 %; non-temporal stores cost it 6 %).
template <class T> __device__ __forceinline__ T ld_stream(const T* p) { return __builtin_nontemporal_load(p); }
template <class T, class U> __device__ __forceinline__ void st_stream(T* p, U v) { *p = (T)v; }
// 64-wide ballot straight from the condition mask (the generic __ballot first turns the flag into an integer)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// readfirstlane as unsigned (the builtin returns int, which would sign-extend when widened) and for 64-bit values
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)uniform32((uint32_t)(v >> 32)) << 32) | (unsigned long long)uniform32((uint32_t)v);
}
__device__ __forceinline__ double uniform_f64(double v) {
    return __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(v)));
}
__device__ __forceinline__ uint32_t rank_below(unsigned long long mask, uint32_t lane) {
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// The sub-pool of this wave: its index, ray count and first slot (wave-uniform).
struct WaveSlot {
    uint32_t sub, count, base;
};
__device__ __forceinline__ WaveSlot wave_slot(const DPool& pool) {
    WaveSlot w;
    w.sub = uniform32((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    w.base = w.sub * pool.sub_capacity;
    if (w.sub >= pool.n_sub) {
        w.count = 0u;
    } else if (pool.counts) {
        w.count = pool.counts[w.sub];
    } else {
        const uint32_t rest = pool.flat_total > w.base ? pool.flat_total - w.base : 0u;
        w.count = rest < pool.sub_capacity ? rest : pool.sub_capacity;
    }
    return w;
}
__device__ __forceinline__ uint32_t pool_count(const DPool& pool, uint32_t sub) {
    if (sub >= pool.n_sub) return 0u;
    if (pool.counts) return pool.counts[sub];
    const uint32_t base = sub * pool.sub_capacity;
    const uint32_t rest = pool.flat_total > base ? pool.flat_total - base : 0u;
    return rest < pool.sub_capacity ? rest : pool.sub_capacity;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- rejection samplers, as a wave
// The reference's rejection loops (Vector3::random_unit, src/vector.rs:59-81,
// accepts 52 % of its candidates; random_in_unit_disk, src/vector.rs:83-95,
// 79 %) run by 64 lanes in lockstep wait for the unluckiest lane (6-7 rounds
// for 1.9 useful trials of random_unit).  Here every lane with `need` first
// tries its own trial 0; after that the lanes between two unresolved lanes all
// work for the upper one, each on a different one of its next trials, and the
// lowest accepted trial wins.  Trial k of a ray is Philox block blk0 + k of its
// stream whichever lane computes it, so the accepted candidate and the number
// of trials are exactly those of the sequential loop.  Must be called by all
// 64 lanes.  Returns the accepted candidate and the number of trials used.
// Sampler: static V3 candidate(k0, k1, pixel, sample, blk); static bool accept(V3).
struct UnitVectorTrial {   // one trial of Vector3::random_unit (the candidate is not normalised)
    static constexpr uint32_t OWN_TRIALS = 2u;   // accepts 52 % of its candidates
    static __device__ __forceinline__ V3 candidate(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk) {
        return random_unit_candidate(k0, k1, pixel, sample, blk);
    }
    static __device__ __forceinline__ bool accept(const V3& c) {
        const double l = length_squared(c);
        return 1e-160 < l && l <= 1.0;
    }
};
struct UnitDiskTrial {   // one trial of Vector3::random_in_unit_disk: x, y = random_range(-1.0..1.0) from the two words of a block
    static constexpr uint32_t OWN_TRIALS = 1u;   // accepts 79 %
    static __device__ __forceinline__ V3 candidate(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk) {
        double u0, u1;
        uniforms_of_block(k0, k1, pixel, sample, blk, u0, u1);
        return v3(u0 * 2.0 + -1.0, u1 * 2.0 + -1.0, 0.0);
    }
    static __device__ __forceinline__ bool accept(const V3& c) { return length_squared(c) < 1.0; }
};
template <class Sampler>
__device__ __forceinline__ void rejection_wave(bool need, uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk0,
                                               V3& cand, uint32_t& used) {
    const uint32_t lane = lane_id();
    cand = v3(1.0, 0.0, 0.0);
    used = 0u;
    uint32_t tried = 0;   // trials of this lane's own ray that have failed so far
    bool unresolved = need;
    if (need) {
        const V3 c = Sampler::candidate(k0, k1, pixel, sample, blk0);
        tried = 1u;
        if (Sampler::accept(c)) {
            cand = c;
            used = 1u;
            unresolved = false;
        }
        if (Sampler::OWN_TRIALS == 2u) {
            // A sampler that rejects about half of its candidates tries its second one in the same lane as well (the two
            // blocks are independent instruction streams): a quarter of the lanes is then left for the shared rounds below, each of
            // which costs more than a trial (nine cross-lane reads and the bookkeeping of who helps whom) and the wave's time
            // to wait for them.
            const V3 c1 = Sampler::candidate(k0, k1, pixel, sample, blk0 + 1u);
            const bool ok1 = Sampler::accept(c1);
            if (unresolved) {
                tried = 2u;
                if (ok1) {
                    cand = c1;
                    used = 2u;
                    unresolved = false;
                }
            }
        }
    }
    unsigned long long U = ballot64(unresolved);
    while (U != 0ull) {
        // the unresolved lane at or above this one, and how many lanes lie between it and the unresolved lane below
        const unsigned long long at_or_above = U >> lane;
        const bool helping = at_or_above != 0ull;
        const uint32_t target = helping ? lane + (uint32_t)__builtin_ctzll(at_or_above) : lane;
        const unsigned long long below = U & ((1ull << lane) - 1ull);                      // unresolved lanes under this one
        const uint32_t first_helper = below ? 64u - (uint32_t)__builtin_clzll(below) : 0u;   // lane after the unresolved lane below
        const uint32_t q = lane - first_helper;                                              // this lane's trial offset for the target
        const uint32_t t_pixel = __shfl(pixel, (int)target, 64), t_sample = __shfl(sample, (int)target, 64);
        const uint32_t t_blk = __shfl(blk0 + tried, (int)target, 64);
        bool ok = false;
        V3 c = v3(0.0, 0.0, 0.0);
        if (helping) {
            c = Sampler::candidate(k0, k1, t_pixel, t_sample, t_blk + q);
            ok = Sampler::accept(c);
        }
        const unsigned long long okmask = ballot64(ok);
        // an unresolved lane looks at its helpers: lanes first_helper .. lane
        const unsigned long long mine = unresolved ? (okmask >> first_helper) & (~0ull >> (63u - (lane - first_helper))) : 0ull;
        const bool got = mine != 0ull;
        const uint32_t win_q = got ? (uint32_t)__builtin_ctzll(mine) : 0u;
        const int src = (int)(first_helper + win_q);
        const double cx = __shfl(c.x, src, 64), cy = __shfl(c.y, src, 64), cz = __shfl(c.z, src, 64);
        if (unresolved) {
            if (got) {
                cand = v3(cx, cy, cz);
                used = tried + win_q + 1u;
                unresolved = false;
            } else {
                tried += lane - first_helper + 1u;   // all of this round's helpers failed
            }
        }
        U = ballot64(unresolved);
    }
}
__device__ __forceinline__ void random_unit_wave(bool need, uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t blk0,
                                                 V3& cand, uint32_t& used) {
    rejection_wave<UnitVectorTrial>(need, k0, k1, pixel, sample, blk0, cand, used);
}

// ---------------------------------------------------------------- primary rays
// sample_square + get_ray, src/camera.rs:164-182.
__device__ __forceinline__ void start_path(const DCamera& cam, uint64_t seed, uint32_t pixel, uint32_t sample,
                                           V3& origin, V3& direction, uint32_t& draw) {
    PathRng rng(seed, pixel, sample, 0u);
    rng.begin_event();
    uint32_t row = pixel / cam.image_width;
    uint32_t col = pixel - row * cam.image_width;
    double ox = rng.next() - 0.5;
    double oy = rng.next() - 0.5;
    V3 pixel_sample = ld3(cam.pixel00_loc) + (((double)col + ox) * ld3(cam.pixel_du)) + (((double)row + oy) * ld3(cam.pixel_dv));
    if (cam.defocus_angle <= 0.0) {
        origin = ld3(cam.center);
    } else {
        V3 p = random_in_unit_disk(rng);
        origin = ld3(cam.center) + (p.x * ld3(cam.defocus_disk_u)) + (p.y * ld3(cam.defocus_disk_v));
    }
    direction = pixel_sample - origin;
    draw = rng.draw;
}

// The same for the 64 paths of a wave at once (lanes without `valid` only help): the defocus disk's
// rejection loop is shared out over the wave.  Bit-identical to start_path.  Must be called by all 64 lanes.
__device__ __forceinline__ void start_path_wave(const DCamera& cam, uint64_t seed, bool valid, uint32_t pixel, uint32_t sample,
                                                V3& origin, V3& direction, uint32_t& draw) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    double u0, u1;
    uniforms_of_block(k0, k1, pixel, sample, 0u, u0, u1);   // sample_square: draws 0 and 1
    const uint32_t row = pixel / cam.image_width;
    const uint32_t col = pixel - row * cam.image_width;
    const double ox = u0 - 0.5, oy = u1 - 0.5;
    const V3 pixel_sample = ld3(cam.pixel00_loc) + (((double)col + ox) * ld3(cam.pixel_du)) + (((double)row + oy) * ld3(cam.pixel_dv));
    origin = ld3(cam.center);
    draw = 2u;
    if (cam.defocus_angle > 0.0) {   // wave-uniform
        V3 p;
        uint32_t trials;
        rejection_wave<UnitDiskTrial>(valid, k0, k1, pixel, sample, 1u, p, trials);   // a new sampling event: block 1 on
        origin = ld3(cam.center) + (p.x * ld3(cam.defocus_disk_u)) + (p.y * ld3(cam.defocus_disk_v));
        draw = 2u + 2u * trials;
    }
    direction = pixel_sample - origin;
}

// Path g of a render call is sample (spp_begin + g / npix) of the pixel that
// position g % npix of the pixel order names.  The single launch takes pixels in
// image order: the rays of a sub-pool then come from neighbouring pixels, stay
// together through the compactions and walk the tree together (+2.6 % on the
// headline frame against the scrambled order); its waves do not wait for each
// other, so a sub-pool over an expensive part of the image just takes fewer
// paths.  The round loop (SCRAMBLE) keeps runs of 64 consecutive pixels
// together but sends consecutive runs to far-apart parts of the image: run r
// of the order is run (r * RUN_STEP) mod n_runs of the image.  There a launch
// ends when its slowest wave does, and with runs in image order the waves
// whose sub-pools hold an expensive region take several times longer than the
// rest.  The pixels after the last whole run keep their place.  Which pixel a
// path index names does not change the image: every (pixel, sample) is still
// rendered exactly once.
constexpr unsigned long long RUN_STEP = 1000003ull;   // prime; a bijection on runs unless n_runs is a multiple of it
template <bool SCRAMBLE>
__device__ __forceinline__ void split_path_index(unsigned long long g, unsigned long long npix, uint32_t& pixel, uint32_t& sample_off) {
    unsigned long long q;
    if (g < (1ull << 52)) {
        // one f64 division instead of a software 64-bit one; exact in f64, the
        // rounded quotient is off by at most one
        q = (unsigned long long)((double)g / (double)npix);
        if (q * npix > g) --q;
        else if ((q + 1) * npix <= g) ++q;
    } else {
        q = g / npix;
    }
    sample_off = (uint32_t)q;
    const uint32_t pos = (uint32_t)(g - q * npix);
    const uint32_t n_runs = (uint32_t)(npix >> 6);
    uint32_t run = pos >> 6;
    if (SCRAMBLE && run < n_runs && n_runs % RUN_STEP != 0ull) {
        const unsigned long long prod = (unsigned long long)run * (RUN_STEP % n_runs);   // < 2^32 * 2^32
        // prod mod n_runs by one f64 division (prod < 2^52 for every image of at most 2^32 pixels)
        unsigned long long qq = (unsigned long long)((double)prod / (double)n_runs);
        if (qq * n_runs > prod) --qq;
        else if ((qq + 1) * n_runs <= prod) ++qq;
        run = (uint32_t)(prod - qq * n_runs);
    }
    pixel = (run << 6) | (pos & 63u);
}

// Image order in blocks: position `pos` of the pixel order names the pixel at (pos & 7, (pos >> 3) & 7) of an 8 x 8 tile, tiles
// run 4 x 2 inside blocks of 32 x 16 pixels and the blocks run along the image's rows.  A run of 64 paths then covers a square
// of the image and the ~800 paths of a sub-pool one or two blocks, instead of strips one pixel high: rays that start together
// walk the same part of a tree that is read from global memory.  Needs a width that is a multiple of 32 and a height that
// is a multiple of 16 (checked by the caller); any bijection on the pixels renders the same image.
__device__ __forceinline__ uint32_t tiled_pixel(uint32_t pos, uint32_t width) {
    const uint32_t k = pos & 63u, run = pos >> 6, t = run & 7u, blk = run >> 3, per_row = width >> 5;
    uint32_t by = (uint32_t)((float)blk / (float)per_row);   // blk < 2^23: off by at most one
    if (by * per_row > blk) --by;
    else if ((by + 1u) * per_row <= blk) ++by;
    const uint32_t bx = blk - by * per_row;
    const uint32_t x = (bx << 5) + ((t & 3u) << 3) + (k & 7u), y = (by << 4) + ((t >> 2) << 3) + (k >> 3);
    return y * width + x;
}

// The wave starts paths first .. first + n_new - 1 in slots slot0 .. of `st`.
// region_base / region_npix: path indices count through the positions [region_base, region_base + region_npix) of the pixel
// order only (region_npix = 0: through all of them).
template <bool SCRAMBLE>
__device__ __forceinline__ void generate_wave(const DCamera& cam, const DState& st, uint32_t slot0, unsigned long long first, uint32_t n_new,
                                              uint64_t seed, uint32_t spp_begin, uint32_t tiled = 0u, uint32_t region_base = 0u, uint32_t region_npix = 0u) {
    const unsigned long long npix = region_npix ? (unsigned long long)region_npix : (unsigned long long)cam.image_width * cam.image_height;
    for (uint32_t i0 = 0; i0 < n_new; i0 += 64u) {   // whole rows: the lanes past the end help the others
        const uint32_t i = i0 + lane_id();
        const bool valid = i < n_new;
        uint32_t pixel, sample_off;
        split_path_index<SCRAMBLE>(first + (valid ? i : 0u), npix, pixel, sample_off);
        if (!SCRAMBLE) pixel += region_base;
        if (!SCRAMBLE && tiled) pixel = tiled_pixel(pixel, cam.image_width);   // wave-uniform choice
        const uint32_t sample = spp_begin + sample_off;
        V3 o, d;
        uint32_t draw;
        start_path_wave(cam, seed, valid, pixel, sample, o, d, draw);
        if (!valid) continue;
        const uint32_t k = slot0 + i;
        st_stream(&st.ox[k], o.x); st_stream(&st.oy[k], o.y); st_stream(&st.oz[k], o.z);
        st_stream(&st.dx[k], d.x); st_stream(&st.dy[k], d.y); st_stream(&st.dz[k], d.z);
        st_stream(&st.bx[k], 1.0); st_stream(&st.by[k], 1.0); st_stream(&st.bz[k], 1.0);
        st_stream(&st.pixel[k], pixel);
        st_stream(&st.sample[k], sample);
        st_stream(&st.draw[k], draw);
        st_stream(&st.depth[k], 0u);
    }
}

// Appends new paths behind the survivors of every sub-pool of state buffer
// `st` until the sub-pool is full or every path of the call has been started,
// and publishes the round's ray counts.  Sub-pool s takes the path indices
// that follow those of sub-pools 0..s-1, so each block first sums what the
// sub-pools before its own will start (plain loads, no communication between
// blocks).  When fewer paths are left than there is room, they are shared out
// evenly.  The last block also publishes the totals of the round.
__global__ __launch_bounds__(BLOCK) void k_generate(DCamera cam, DState st, DRenderCtl* ctl, const uint32_t* surv, uint32_t* live,
                                                    uint32_t n_sub, uint32_t cap, uint32_t parity, uint64_t seed, uint32_t spp_begin) {
    __shared__ unsigned long long red_new[BLOCK / 64], red_surv[BLOCK / 64];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long next = ctl->next_path[parity];
    const unsigned long long left = ctl->total_paths - next;
    uint32_t share = cap;
    if (left < (unsigned long long)n_sub * cap) share = (uint32_t)((left + n_sub - 1u) / n_sub);

    const uint32_t first_sub = blockIdx.x * (BLOCK / 64);
    unsigned long long sum_new = 0, sum_surv = 0;
    for (uint32_t j = threadIdx.x; j < first_sub; j += BLOCK) {
        const uint32_t sv = surv[j];
        const uint32_t room = cap - sv;
        sum_new += room < share ? room : share;
        sum_surv += sv;
    }
    sum_new = wave_sum(sum_new);
    sum_surv = wave_sum(sum_surv);
    if (lane == 0) {
        red_new[wave] = sum_new;
        red_surv[wave] = sum_surv;
    }
    __syncthreads();
    unsigned long long before = 0, surv_total = 0;   // over the sub-pools before this block's
#pragma unroll
    for (int q = 0; q < BLOCK / 64; ++q) {
        before += red_new[q];
        surv_total += red_surv[q];
    }
    // this block's own sub-pools, in order; never past the last path of the call
    uint32_t my_surv = 0, my_new = 0;
    unsigned long long my_first = next, block_new = 0;
#pragma unroll
    for (uint32_t q = 0; q < BLOCK / 64; ++q) {
        const uint32_t s2 = first_sub + q;
        uint32_t sv = 0, want = 0;
        if (s2 < n_sub) {
            sv = surv[s2];
            const uint32_t room = cap - sv;
            want = room < share ? room : share;
        }
        const unsigned long long at = before + block_new;
        const unsigned long long rest = left > at ? left - at : 0ull;
        const uint32_t n_new = rest < want ? (uint32_t)rest : want;
        if (q == wave) {
            my_surv = sv;
            my_new = n_new;
            my_first = next + at;
        }
        block_new += n_new;
        surv_total += sv;
    }
    const uint32_t w = first_sub + wave;
    if (lane == 0 && w < n_sub) live[w] = my_surv + my_new;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        unsigned long long started = before + block_new;
        if (started > left) started = left;
        ctl->next_path[parity ^ 1u] = next + started;
        ctl->live_total = (uint32_t)(surv_total + started);
        ctl->rays += surv_total + started;
    }
    generate_wave<true>(cam, st, w * cap + my_surv, my_first, my_new, seed, spp_begin);
}

// Primary rays only, for the parity API (rays laid out [n][3]).
__global__ __launch_bounds__(BLOCK) void k_primary_rays(DCamera cam, uint64_t seed, uint32_t sample, uint32_t npix,
                                                        double* org, double* dir) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < npix; i += gridDim.x * BLOCK) {
        V3 o, d;
        uint32_t draw;
        start_path(cam, seed, i, sample, o, d, draw);
        org[3 * (size_t)i + 0] = o.x; org[3 * (size_t)i + 1] = o.y; org[3 * (size_t)i + 2] = o.z;
        dir[3 * (size_t)i + 0] = d.x; dir[3 * (size_t)i + 1] = d.y; dir[3 * (size_t)i + 2] = d.z;
    }
}

// Second and third tier of the box test, out of line: the f64 filter on the
// exact box and, if that is undecided too or the ray is outside the filter's
// preconditions, the reference's division form.
__device__ __noinline__ bool tier2_box(double2 q0, double2 q1, double2 q2, V3 o, V3 d, double tmin, double tmax, bool scene_filter) {
    const double bmin[3] = {q0.x, q0.y, q1.x};
    const double bmax[3] = {q1.y, q2.x, q2.y};
    if (scene_filter && filter_ray_ok(o, d, tmin, tmax)) {
        const V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        FilterBounds fb;
        filter_set_tmin(fb, tmin);
        filter_set_tmax(fb, tmax);
        const int r = box_hit_filtered(bmin, bmax, o, inv, fb);
        if (r >= 0) return r > 0;
    }
    return box_hit_exact(bmin, bmax, o, d, tmin, tmax);
}

// ---------------------------------------------------------------- closest hit
// impl Hittable for BVH::hit, src/bvh/mod.rs:134-174.
//
// Order of visits.  The reference pops node indices from an explicit stack,
// pushing the right child and then the left child (i+1) of every internal
// node whose box is hit.  Because children are always visited left first,
// the node popped after a subtree is finished or culled depends only on where
// that subtree ends in the pre-order array.  That successor is stored per
// node (`miss`), so the walk needs no stack: box missed -> miss; internal hit
// -> i+1; leaf -> test the surface, then miss.  The visit order, the
// shrinking interval and the tie rule (`hit.t > nearest.t` keeps the old hit,
// :159-167) are exactly the reference's.
//
// Shape of the kernel.  Each wave works through tiles of 64 consecutive rays
// and alternates two phases: all lanes walk nodes until they stand on a leaf
// whose box is hit (or run out of tree), then the lanes that found one test
// their surface together.  A ray's own sequence of box and surface tests is
// unchanged by this; only when the wave executes them is.
template <bool COUNT, bool EXACT>
__global__ __launch_bounds__(BLOCK, RTX_TRAVERSE_WAVES) void k_traverse(DScene sc, DState rays, DHits hits, DPool pool,
                                                                        double tmin, double tmax0, DRenderCtl* ctl) {
    const WaveSlot ws = wave_slot(pool);
    const uint32_t count = ws.count;
    const uint32_t lane = lane_id();
    const uint32_t tile_stride = 64u;
    uint32_t tile_base = 0u;   // first ray (offset in the wave's sub-pool) of the current tile
    const uint32_t end = sc.n_nodes;
    const bool scene_filter = !EXACT && sc.filter_ok != 0u;

    uint32_t idx = NONE;   // ray of this lane, NONE = idle
    V3 o = v3(0, 0, 0), d = v3(1, 1, 1), inv = v3(1, 1, 1);
    double tmax = tmax0, best_t = tmax0;
    uint32_t best_prim = NONE, node = 0;
    bool filtered = false;
    FilterBounds fb;
    filter_set_tmin(fb, tmin);
    filter_set_tmax(fb, tmax0);
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0;

    for (;;) {
        // ---- refill: when every lane has retired, the wave takes the next 64 rays of its sub-pool
        if (ballot64(idx != NONE) == 0ull) {
            if (tile_base >= count) break;
            const uint32_t off = tile_base + lane;
            tile_base += tile_stride;
            if (off < count) {
                const uint32_t mine = ws.base + off;
                idx = mine;
                o = v3(rays.ox[mine], rays.oy[mine], rays.oz[mine]);
                d = v3(rays.dx[mine], rays.dy[mine], rays.dz[mine]);
                tmax = tmax0;
                best_t = tmax0;
                best_prim = NONE;
                node = 0;
                filtered = scene_filter && filter_ray_ok(o, d, tmin, tmax0);
                inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                filter_set_tmax(fb, tmax0);
            }
        }

        // ---- phase 1: walk to the next leaf whose box is hit
        bool searching = idx != NONE && node < end;
        bool pending = false;
        uint32_t leaf_kind = 0, leaf_slot = 0, leaf_prim = NONE;
        while (ballot64(searching) != 0ull) {
            if (searching) {
                // the whole 64-byte record in four 16-byte loads issued back to back
                const double2* np = reinterpret_cast<const double2*>(sc.nodes + node);
                const double2 q0 = np[0], q1 = np[1], q2 = np[2];
                const uint4 link = *reinterpret_cast<const uint4*>(np + 3);   // miss, prim, kind, slot
                const double bmin[3] = {q0.x, q0.y, q1.x};
                const double bmax[3] = {q1.y, q2.x, q2.y};
                if (COUNT) ++c_nodes;
                bool hit;
                if (EXACT) {
                    hit = box_hit_exact(bmin, bmax, o, d, tmin, tmax);
                } else {
                    const int r = box_hit_filtered(bmin, bmax, o, inv, fb);
                    hit = r > 0;
                    if (!filtered || r < 0) hit = box_hit_exact_slow(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, o, d, tmin, tmax);
                }
                const bool leaf = link.y != NONE;
                const bool found = hit && leaf;
                node = (hit && !leaf) ? node + 1u : link.x;   // a leaf's successor is link.x as well
                if (found) {
                    pending = true;
                    leaf_prim = link.y;
                    leaf_kind = link.z;
                    leaf_slot = link.w;
                }
                searching = !found && node < end;
            }
        }

        // ---- phase 2: surface tests of the lanes standing on a leaf
        if (pending) {
            double t;
            bool ph;
            if (leaf_kind == 0u) {
                if (COUNT) ++c_sph;
                const double2* spp = reinterpret_cast<const double2*>(sc.spheres + leaf_slot);
                const double2 sph_a = spp[0], sph_b = spp[1];
                DSphere sp;
                sp.c[0] = sph_a.x; sp.c[1] = sph_a.y; sp.c[2] = sph_b.x; sp.r = sph_b.y;
                ph = sphere_t(sp, o, d, tmin, tmax, t);
            } else {
                if (COUNT) ++c_pln;
                double al, be;
                ph = planar_t(sc.planars[leaf_slot], leaf_kind, o, d, tmin, tmax, t, al, be);
            }
            // `if let Some(nearest) = acc && hit.t > nearest.t { keep } else { replace }`
            if (ph && !(best_prim != NONE && t > best_t)) {
                best_prim = leaf_prim;
                best_t = t;
                tmax = t;   // shrunken_ray_t.max = hit.t
                filter_set_tmax(fb, t);
            }
        }

        // ---- retire rays that have run out of tree
        if (idx != NONE && node >= end) {
            hits.prim[idx] = best_prim;
            hits.t[idx] = best_t;
            idx = NONE;
        }
    }
    if (COUNT && ctl) {
        atomicAdd(&ctl->node_visits, (unsigned long long)c_nodes);
        atomicAdd(&ctl->sphere_tests, (unsigned long long)c_sph);
        atomicAdd(&ctl->planar_tests, (unsigned long long)c_pln);
    }
}

// The same walk with the box test in three tiers: single precision from a
// 32-byte node record (box_hit_f32); if that cannot decide, the f64 filter on
// the exact box (box_hit_filtered); if that cannot decide either, the
// reference's six divisions.  Every tier either proves what the reference's
// predicate returns or hands the visit on, so the sequence of nodes visited
// and of surfaces tested is the reference's for every ray.  With LDS_NODES
// the f32 records of the whole tree are first copied into LDS (scenes up to a
// few thousand nodes), so a visit costs two LDS reads and no trip through the
// vector cache.
// Work list and scene staging of one wave of the closest-hit walk.
// LDS address (the 32-bit offset inside the block's LDS) of a pointer into LDS, and the way back
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x4* lds_u4_ptr;
__device__ __forceinline__ uint4 lds_read_u4(uint32_t lds_addr, uint32_t index) {   // 16 bytes at LDS address lds_addr + 16 * index
    const u32x4 v = ((lds_u4_ptr)(uintptr_t)lds_addr)[index];
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
typedef __attribute__((address_space(3))) u32x4* lds_u4_wptr;
__device__ __forceinline__ void lds_write_u4(uint32_t lds_addr, uint32_t index, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    u32x4 v;
    v.x = x; v.y = y; v.z = z; v.w = w;
    ((lds_u4_wptr)(uintptr_t)lds_addr)[index] = v;
}
__device__ __forceinline__ void lds_write_d2(uint32_t lds_addr, uint32_t index, double a, double b) {
    lds_write_u4(lds_addr, index, (uint32_t)__double2loint(a), (uint32_t)__double2hiint(a), (uint32_t)__double2loint(b), (uint32_t)__double2hiint(b));
}
__device__ __forceinline__ void lds_read_d2(uint32_t lds_addr, uint32_t index, double& a, double& b) {
    const uint4 v = lds_read_u4(lds_addr, index);
    a = __hiloint2double((int)v.y, (int)v.x);
    b = __hiloint2double((int)v.w, (int)v.z);
}
constexpr uint32_t RESIDENT_STAGE_BYTES = 64u;   // a staged primary ray of the resident render kernel: origin, direction (6 f64), pixel, sample, draw count (+ pad)
// Copies the f32 node records into LDS.  In the copy a record's link (byte offset of its successor in the table) becomes
// the successor's LDS address: the walk then keeps LDS addresses, not offsets, and the address of a visit's two reads is
// the lane's state itself instead of table + offset (one vector instruction of a visit's thirty).  The table starts on a
// 16-byte boundary, so the flag that BATCH scenes keep in bit 0 of a link survives the addition.
__device__ __forceinline__ void stage_nodes_in_lds(uint4* lds_nodes, const DNode32* nodes32, const uint32_t* leaf_codes, uint32_t n_nodes) {
    const uint4* src = reinterpret_cast<const uint4*>(nodes32);
    const uint32_t bias = lds_address(lds_nodes);
    for (uint32_t i = threadIdx.x; i < 2u * n_nodes; i += blockDim.x) {
        uint4 v = src[i];
        if (i & 1u) {   // second half of a record: planes z, the two links (NODE32_LEAF in the top bit of `down` is not touched: records lie below 2^30)
            v.z += bias;
            v.w += bias;
        }
        lds_nodes[i] = v;
    }
    uint32_t* codes = reinterpret_cast<uint32_t*>(lds_nodes + 2u * n_nodes);   // the leaf codes follow the records
    for (uint32_t i = threadIdx.x; i < n_nodes; i += blockDim.x) codes[i] = leaf_codes[i];
}
// Where the tables that follow the tree start in LDS, in 16-byte units from the start of the node records
__device__ __forceinline__ uint32_t lds_after_tree(uint32_t n_nodes) { return (uint32_t)(node_tables_lds_bytes(n_nodes) >> 4); }
struct TraverseWork {
    const uint4* lds_nodes;     // LDS copy of the f32 node records (LDS_NODES)
    uint32_t lds_base;          // its LDS address: with LDS_NODES the walk's node positions are LDS addresses
    const uint4* lds_spheres;   // LDS copies of the surface tables (lds_surfaces)
    const uint4* lds_planars;
    uint32_t lds_surfaces;
    // SHARED: the block's sub-pools end to end, drawn through an LDS cursor
    const uint32_t* s_first;
    uint32_t* s_cursor;
    uint32_t first_pool, sub_capacity;
    // !SHARED: `count` rays from slot `base` on, the wave's own
    uint32_t base, count;
};
struct WorkCounters {
    unsigned long long nodes = 0, spheres = 0, planars = 0, fallbacks = 0;
};

// Counters of an instrumented build (-DRTX_PROFILE_PHASES): where a wave's cycles go in the closest-hit walk
#ifdef RTX_PROFILE_PHASES
struct WalkProfile {
    long long fb_iters = 0, fb_cycles = 0, refill = 0, p1 = 0, p2 = 0, retire = 0, iters = 0, outer = 0, mark = 0, stage = 0, take = 0, stagings = 0, loop = 0, wave_steps = 0;
};
#define PT(field) do { const long long now_ = clock64(); prof.field += now_ - prof.mark; prof.mark = now_; } while (0)
#else
struct WalkProfile {};
#define PT(field) do {} while (0)
#endif

// The node loop of walk_pass for f32 records in LDS, written out: the loop a wave goes round 25 times per ray.  The compiler's
// form of it spends 43 instructions on a step: its exit logic keeps three masks (lanes that left, lanes that leave now, the
// threshold), and it waits for the two reads one by one.  This one spends 35 (24 of them vector instructions instead of 26): both reads, one wait, the box test exactly as
// box_hit_tier1<false> compiles (the same operations in the same order, hence the same bits), the link select, and an exit
// that needs one mask; lanes on an undecided box are remembered in a scalar mask.  Worth 1.3 % of the headline frame.  (Measured
// on this loop, same frame: the margin taken as a constant, wrongly, -1.4 %; four more register moves +1.6 %; one more read of the record per step +0.7 %; the next record requested
// before the margin and the exit test are evaluated, which takes two more vector instructions, +1.4 %.)
// A lane's `at` is the LDS address of its record; it ends at `end` or beyond (out of tree), with WAIT (0x80000000, part of a
// leaf's `down` link as stored) once it stands on a leaf whose box is hit, or with OPEN (0x40000000) on a box the tier cannot
// decide.  Lanes with at >= end take no part.  The loop ends when no lane walks any more or `pend_at` lanes have started waiting
// on a leaf.  For kernels built for 128 vector registers: the record and the temporaries sit in v104-v112.
// Wait states: a vector instruction that reads a mask written by a vector compare comes at least two instructions after it.
typedef uint32_t u32x4_reg __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void node_loop_lds_f32(uint32_t& at, const RayF32& rf, uint32_t end, uint32_t pend_at) {
    uint32_t n_pending = 0, cnt, tmp_a;
    u32x4_reg rec_a, rec_b;   // lo.x hi.x lo.y hi.y | lo.z hi.z miss down
    unsigned long long saved, mask, undecided, open = 0;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_gt_u32_e32 vcc, %[end], %[at]\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "s_cbranch_execz .Lrtx_nl_done_%=\n"
        ".Lrtx_nl_loop_%=:\n\t"
        "ds_read_b128 v[104:107], %[at]\n\t"
        "ds_read_b128 v[108:111], %[at] offset:16\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fma_f32 v104, v104, %[ix], -%[ox]\n\t"
        "v_fma_f32 v105, v105, %[ix], -%[ox]\n\t"
        "v_fma_f32 v106, v106, %[iy], -%[oy]\n\t"
        "v_fma_f32 v107, v107, %[iy], -%[oy]\n\t"
        "v_fma_f32 v108, v108, %[iz], -%[oz]\n\t"
        "v_fma_f32 v109, v109, %[iz], -%[oz]\n\t"
        "v_min_f32_e32 v112, v104, v105\n\t"          // near x
        "v_max_f32_e32 v104, v104, v105\n\t"          // far x
        "v_min_f32_e32 v105, v106, v107\n\t"          // near y
        "v_max_f32_e32 v106, v106, v107\n\t"          // far y
        "v_min_f32_e32 v107, v108, v109\n\t"          // near z
        "v_max_f32_e32 v108, v108, v109\n\t"          // far z
        "v_max_f32_e32 v112, v112, v105\n\t"
        "v_min_f32_e32 v104, v104, v106\n\t"
        "v_max3_f32 v112, v112, v107, %[tmin]\n\t"    // lo, clamped
        "v_min3_f32 v104, v104, v108, %[tmax]\n\t"    // hi, clamped
        "v_max3_f32 v105, |v112|, |v104|, %[tg]\n\t"  // M
        "v_sub_f32_e32 v106, v104, v112\n\t"          // m = hi - lo
        "v_fmamk_f32 v105, v105, 0x35433333, %[K]\n\t"   // T = M * 12.2u + K
        "v_cmp_gt_f32_e64 %[mk], v106, v105\n\t"      // hit
        "v_cmp_gt_f32_e64 vcc, |v106|, v105\n\t"      // decided
        "s_andn2_b64 %[un], exec, vcc\n\t"            // the lanes this step leaves undecided (one visit in 25 000) ...
        "s_or_b64 %[op], %[op], %[un]\n\t"            // ... are remembered in a mask and leave the loop below
        "v_cndmask_b32_e64 %[at], v110, v111, %[mk]\n\t"   // hit ? down : miss
        "v_cmp_gt_i32_e32 vcc, 0, %[at]\n\t"          // starts waiting on a leaf (WAIT is the sign bit)
        "v_cmp_gt_u32_e64 %[mk], %[end], %[at]\n\t"   // walks on
        "s_bcnt1_i32_b64 %[cnt], vcc\n\t"
        "s_add_u32 %[np], %[np], %[cnt]\n\t"
        "s_andn2_b64 exec, %[mk], %[un]\n\t"
        "s_cbranch_execz .Lrtx_nl_done_%=\n\t"
        "s_cmp_lt_u32 %[np], %[pend]\n\t"
        "s_cbranch_scc1 .Lrtx_nl_loop_%=\n"
        ".Lrtx_nl_done_%=:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [at] "+v"(at), [np] "+s"(n_pending), [op] "+s"(open), [cnt] "=&s"(cnt), [sv] "=&s"(saved), [mk] "=&s"(mask), [un] "=&s"(undecided),
          "=&{v[104:107]}"(rec_a), "=&{v[108:111]}"(rec_b), "=&{v112}"(tmp_a)
        : [ix] "v"(rf.inv[0]), [iy] "v"(rf.inv[1]), [iz] "v"(rf.inv[2]), [ox] "v"(rf.oi[0]), [oy] "v"(rf.oi[1]), [oz] "v"(rf.oi[2]), [tmin] "v"(rf.tmin32),
          [tmax] "v"(rf.tmax32), [tg] "v"(rf.tg), [K] "v"(rf.K), [end] "s"(end), [pend] "s"(pend_at)
        : "vcc", "scc", "memory");
    // An undecided lane has left the loop with the wrong link in `at`: its node is the one whose `down` link it read last (an
    // internal node's is the next record, a leaf's the same with WAIT), and the exact tiers expect it there with OPEN set.
    if (open != 0ull && ((open >> lane_id()) & 1ull) != 0ull) at = ((rec_b.w & 0x7FFFFFFFu) - 32u) | 0x40000000u;
}

// One pass of the closest-hit walk over the rays the lanes of a wave hold (`have`: this lane holds one; `node`: the record it
// stands on, `end` or beyond once it has run out of tree): phase 1 walks nodes until enough lanes stand on a leaf whose box is
// hit, phase 1b resolves the boxes the f32 tier left undecided, phase 2 tests the surfaces of the waiting lanes.
// REC16 (trees in global memory only): the 16-byte quantised records (DNode16) instead of the 32-byte f32 ones.
// ASM: the node loop as node_loop_lds_f32 (callers built for 128 vector registers; trees in LDS, f32 records, no counting)
template <bool COUNT, bool LDS_NODES, bool BATCH, bool REC16, bool ASM = false>
__device__ __forceinline__ void walk_pass(const DScene& sc, const TraverseWork& w, bool have, const V3& o, const V3& d, const RayDiv& rdiv, RayF32& rf, double tmin,
                                          double& tmax, uint32_t& best_leaf, uint32_t& node, uint32_t end, uint32_t PEND_AT, bool scene_filter,
                                          uint32_t& c_nodes, uint32_t& c_sph, uint32_t& c_pln, uint32_t& c_fallback, WalkProfile& prof) {
    constexpr bool batch_undecided = BATCH;   // a separate instance of the kernel: with both forms in one loop the common one is 3 % slower
    constexpr uint32_t REC_BYTES = REC16 ? 16u : 32u, REC_SHIFT = REC16 ? 4u : 5u;
    bool pending = false;    // standing on a leaf whose box was hit, surface not tested yet
    uint32_t leaf = NONE;
    // ---- phase 1: walk to the next leaf whose box is hit
    // A lane that found its leaf waits; so does a lane whose box test the f32 tier could not decide, at that node (the
    // exact tiers run after the loop, for all such lanes together).  The loop also ends as soon as PEND_AT lanes are
    // waiting, so a few long searches do not stall them.  The lane's state is one register: `at` is the record it stands
    // on; it is `end` or beyond once the lane has nothing to walk (idle, out of tree), and carries WAIT or OPEN in its
    // top bits while the lane waits on a leaf or on an undecided box.  With the state in flags the loop spent as many
    // scalar instructions on who walks, waits or is done as vector instructions on the box test, and the scalar chain
    // is what a step waits for.
    constexpr uint32_t WAIT = 0x80000000u, OPEN = 0x40000000u;   // (records lie below 2^30: checked when the scene is created)
    uint32_t n_pending = 0;   // wave-uniform
    bool undecided = false;
    uint32_t at = have ? node : end;
#if !defined(RTX_PROFILE_PHASES) && !defined(RTX_NO_ASM_LOOP)
    constexpr bool asm_loop = ASM && LDS_NODES && !REC16 && !COUNT && !BATCH;
#else
    constexpr bool asm_loop = false;
#endif
    if (asm_loop) {
        node_loop_lds_f32(at, rf, end, PEND_AT);
    } else if (at < end) for (;;) {   // lanes leave the loop one by one (test at the bottom); the wave stays until all have left
        // the record: six planes (f32, or grid steps as f32), whether it is a leaf, its surface code if so,
        // and the byte offset of the record that follows its subtree
        float planes[6];
        bool is_leaf, contained = false;
        uint32_t succ, down;   // where the lane goes if the box is missed / hit (a leaf's `down` carries WAIT)
        if (REC16) {
            const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes16) + at);
            planes[0] = (float)(r.x & 0xFFFFu); planes[1] = (float)(r.x >> 16);
            planes[2] = (float)(r.y & 0xFFFFu); planes[3] = (float)(r.y >> 16);
            planes[4] = (float)(r.z & 0xFFFFu); planes[5] = (float)(r.z >> 16);
            is_leaf = r.w < NODE16_INTERNAL;
            leaf = r.w;   // only read once the lane waits on this leaf; a lane that walks on overwrites it
            succ = is_leaf ? at + 16u : r.w << 4;   // the tag bits shift out; a leaf is followed by the next record
            down = is_leaf ? (succ | WAIT) : at + REC_BYTES;
        } else {
            uint4 n0, n1;
            if (LDS_NODES) {
                n0 = lds_read_u4(at, 0u);
                n1 = lds_read_u4(at, 1u);
            } else {
                const uint4* np = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes32) + at);
                n0 = np[0];
                n1 = np[1];
            }
            planes[0] = __uint_as_float(n0.x); planes[1] = __uint_as_float(n0.y); planes[2] = __uint_as_float(n0.z);
            planes[3] = __uint_as_float(n0.w); planes[4] = __uint_as_float(n1.x); planes[5] = __uint_as_float(n1.y);
            static_assert(NODE32_LEAF == WAIT, "a leaf's `down` link is the waiting lane's state as it stands");
            down = n1.w;
            is_leaf = (int32_t)n1.w < 0;
            succ = BATCH ? (n1.z & ~15u) : n1.z;   // scenes of the BATCH form keep a flag in the (otherwise zero) low bits of a leaf's link
            contained = BATCH && (n1.z & 1u) != 0u;
        }
        if (COUNT) ++c_nodes;
        bool hit;
        bool decided = box_hit_tier1<REC16>(planes, rf, hit);
        if (BATCH && !REC16 && (!COUNT || sc.assume_in_count)) {
            // A node whose surfaces all lie inside its box with a margin (flagged when the scene was created): a hit of one
            // of them inside [tmin, tmax] implies that the box test passes, so where the f32 tier cannot decide the box the walk
            // goes on as if it had passed.  If the reference culls the node, nothing below it is hit and the outcome is the
            // reference's; only the numbers of box and surface tests can differ, so the instances that count the reference's work
            // do not take this path.
            const bool assume = !decided && contained && rf.tmin32 < rf.tmax32 && rf.K <= 3.0e38f;
            hit = hit || assume;
            decided = decided || assume;
        }
        // The lanes that start waiting in this step are counted from the comparisons' own masks (the ballot of a
        // combined flag costs two vector instructions per step): f32 says hit (that implies decided) on a leaf ...
        n_pending += (uint32_t)__popcll(ballot64(hit) & ballot64(is_leaf));
        if (batch_undecided) n_pending += (uint32_t)__popcll(ballot64(true) & ~ballot64(decided));   // ... or, where there are many, undecided
        const uint32_t next = hit ? down : succ;
        at = decided ? next : (at | OPEN);
#ifdef RTX_PROFILE_PHASES
        ++prof.iters;
        if (lane_id() == (uint32_t)__builtin_ctzll(ballot64(true))) ++prof.wave_steps;   // (kept by whichever lane is the first of the step)
#endif
        if (n_pending >= PEND_AT || at >= end) break;   // the first is wave-uniform
    }
    PT(loop);
    uint32_t leaf_at = 0;   // f32 records: position of the leaf a waiting lane stands on (its surface code is read below)
    if (have) {
        pending = (at & WAIT) != 0u;
        undecided = (at & OPEN) != 0u;
        node = at & ~(WAIT | OPEN);
        leaf_at = node - REC_BYTES;   // a waiting lane has moved on to the leaf's successor, the next record
    }
    // ---- phase 1b: the boxes left undecided, for all such lanes together
    if (undecided) {
#ifdef RTX_PROFILE_PHASES
        ++prof.fb_iters;
#endif
        bool is_leaf;
        uint32_t leaf_code, succ;
        if (REC16) {
            const uint32_t link = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes16) + node)->w;
            is_leaf = link < NODE16_INTERNAL;
            leaf_code = link;
            succ = is_leaf ? node + 16u : link << 4;
        } else {
            uint4 n1;
            if (LDS_NODES) n1 = lds_read_u4(node, 1u);
            else n1 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(sc.nodes32) + node)[1];
            is_leaf = (int32_t)n1.w < 0;
            leaf_code = leaf;
            succ = BATCH ? (n1.z & ~15u) : n1.z;
        }
        const double2* bp = reinterpret_cast<const double2*>(sc.boxes64 + ((node - (LDS_NODES ? w.lds_base : 0u)) >> REC_SHIFT));
        const double2 q0 = bp[0], q1 = bp[1], q2 = bp[2];
        if (COUNT) ++c_fallback;
        const bool hit = tier2_box(q0, q1, q2, o, d, tmin, tmax, scene_filter);
        pending = hit && is_leaf;
        leaf_at = node;
        node = (hit && !is_leaf) ? node + REC_BYTES : succ;
        leaf = pending ? leaf_code : leaf;
        undecided = false;
    }
    if (!REC16 && pending) {   // the waiting lanes' surface codes, by node index
        const uint32_t i = (leaf_at - (LDS_NODES ? w.lds_base : 0u)) >> REC_SHIFT;
        if (LDS_NODES) {
            typedef __attribute__((address_space(3))) const uint32_t* lds_u32_ptr;
            leaf = ((lds_u32_ptr)(uintptr_t)end)[i];   // the codes follow the records: `end` is the LDS address behind the last one
        } else {
            leaf = sc.leaf_codes[i];
        }
    }
    PT(p1);

    // ---- phase 2: surface tests of the lanes standing on a leaf
    if (pending) {
        const uint32_t kind = code_geom(leaf), slot = code_slot(leaf);
        double t;
        bool ph;
        if (kind == 0u) {
            if (COUNT) ++c_sph;
            double2 sa, sb;
            if (LDS_NODES && w.lds_surfaces) {   // wave-uniform: a branch, not a per-lane choice between two address spaces
                typedef __attribute__((address_space(3))) const double* lds_ptr;   // keeps the two paths from being merged into one flat load
                const lds_ptr spp = (lds_ptr)reinterpret_cast<const double*>(w.lds_spheres + 2u * slot);
                sa = make_double2(spp[0], spp[1]);
                sb = make_double2(spp[2], spp[3]);
            } else {
                const double2* spp = reinterpret_cast<const double2*>(sc.spheres + slot);
                sa = spp[0];
                sb = spp[1];
            }
            DSphere sp;
            sp.c[0] = sa.x; sp.c[1] = sa.y; sp.c[2] = sb.x; sp.r = sb.y;
            // (trees walked from global memory wait for memory, not for the ALUs: there the prepared division only costs registers)
            ph = REC16 ? sphere_t(sp, o, d, tmin, tmax, t) : sphere_t_ray(sp, o, d, rdiv, tmin, tmax, t);
        } else {
            if (COUNT) ++c_pln;
            double al, be;
            if (LDS_NODES && w.lds_surfaces) ph = planar_t(*reinterpret_cast<const DPlanar*>(w.lds_planars + 8u * slot), kind, o, d, tmin, tmax, t, al, be);
            else ph = planar_t(sc.planars[slot], kind, o, d, tmin, tmax, t, al, be);
        }
        // `if let Some(nearest) = acc && hit.t > nearest.t { keep } else { replace }`: tmax is nearest.t once there is a nearest,
        // and a hit lies inside the interval, so t <= tmax here and the candidate always replaces (ties included)
        if (ph) {
            best_leaf = leaf;
            tmax = t;   // shrunken_ray_t.max = hit.t
            rayf32_set_tmax(rf, t);
        }
        pending = false;
    }

    PT(p2);
}

template <bool COUNT, bool LDS_NODES, bool SHARED, bool BATCH, bool REC16 = false>
__device__ __forceinline__ void traverse_f32_wave(const DScene& sc, const DState& rays, const DHits& hits, const TraverseWork& w, double tmin,
                                                  double tmax0, DRenderCtl* ctl, WorkCounters& wc) {
    const uint32_t lane = lane_id();
    const uint32_t REFILL_AT = sc.refill_at;        // refill once this many lanes are idle (wave-uniform)
    const uint32_t PEND_AT = sc.pend_at;            // test surfaces once this many lanes stand on a leaf (wave-uniform)
    const uint32_t p_end = SHARED ? w.s_first[MAX_BLOCK_WAVES] : w.count;
    bool list_done = p_end == 0u;
    uint32_t next_p = 0;   // !SHARED: next position of the wave's own list
    const bool scene_filter = sc.filter_ok != 0u;
    static_assert(!(REC16 && LDS_NODES), "the quantised records are for trees that do not fit in LDS");
    // nodes are addressed by the byte offset of their record in the table, or (LDS_NODES) by the record's LDS address
    const uint32_t root = LDS_NODES ? w.lds_base : 0u;
    const uint32_t end = root + sc.n_nodes * (REC16 ? 16u : 32u);

    uint32_t idx = NONE;   // ray of this lane, NONE = idle
    V3 o = v3(0, 0, 0), d = v3(1, 1, 1);
    double tmax = tmax0;   // == t of the best hit so far once there is one
    uint32_t best_leaf = NONE, node = root;
    RayF32 rf;
    if (REC16) rayq16_init(rf, o, d, tmin, tmax0, sc.q_org, sc.q_delta);
    else rayf32_init(rf, o, d, tmin, tmax0);
    RayDiv rdiv;
    ray_div_init(rdiv, d);
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0, c_fallback = 0;
    WalkProfile prof;
#ifdef RTX_PROFILE_PHASES
    prof.mark = clock64();
    const long long pt_begin = prof.mark, pt_wall_begin = wall_clock64();
#endif
    for (;;) {
        // ---- refill idle lanes from the wave's work list once enough of them are idle
        const unsigned long long idle = ballot64(idx == NONE);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u && list_done) break;
        if (!list_done && (n_idle >= REFILL_AT || n_idle == 64u)) {
            uint32_t start = next_p;
            if (SHARED) {
                if (lane == 0) start = atomicAdd(w.s_cursor, n_idle);
                start = uniform32(start);
            } else {
                next_p += n_idle;
            }
            list_done = start + n_idle >= p_end;
            const uint32_t p = start + rank_below(idle, lane);
            if (idx == NONE && p < p_end) {
                uint32_t mine = w.base + p;
                if (SHARED) {
                    uint32_t j = 0;   // which of the block's sub-pools position p falls in
#pragma unroll
                    for (uint32_t q = 1; q < MAX_BLOCK_WAVES; ++q) j += p >= w.s_first[q] ? 1u : 0u;
                    mine = (w.first_pool + j) * w.sub_capacity + (p - w.s_first[j]);
                }
                idx = mine;
                o = v3(ld_stream(&rays.ox[mine]), ld_stream(&rays.oy[mine]), ld_stream(&rays.oz[mine]));
                d = v3(ld_stream(&rays.dx[mine]), ld_stream(&rays.dy[mine]), ld_stream(&rays.dz[mine]));
                tmax = tmax0;
                best_leaf = NONE;
                node = root;
                if (REC16) rayq16_init(rf, o, d, tmin, tmax0, sc.q_org, sc.q_delta);
                else rayf32_init(rf, o, d, tmin, tmax0);
                if (!REC16 && sc.n_spheres) ray_div_init(rdiv, d);   // wave-uniform: scenes without spheres skip it
            }
        }

        PT(refill);
#ifdef RTX_PROFILE_PHASES
        ++prof.outer;
#endif
        walk_pass<COUNT, LDS_NODES, BATCH, REC16>(sc, w, idx != NONE, o, d, rdiv, rf, tmin, tmax, best_leaf, node, end, PEND_AT, scene_filter, c_nodes, c_sph,
                                                  c_pln, c_fallback, prof);
        // ---- retire rays that have run out of tree
        if (idx != NONE && node >= end) {
            st_stream(&hits.prim[idx], best_leaf);
            st_stream(&hits.t[idx], tmax);
            idx = NONE;
        }
        PT(retire);
    }
#ifdef RTX_PROFILE_PHASES
    if (lane == 0 && ctl && p_end > 0) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(ctl + 1);   // debug area behind the control block
        atomicAdd(pr + 0, (unsigned long long)prof.refill);
        atomicAdd(pr + 1, (unsigned long long)(prof.p1 + prof.loop));   // (the loop proper and what surrounds it)
        atomicAdd(pr + 2, (unsigned long long)prof.p2);
        atomicAdd(pr + 3, (unsigned long long)prof.retire);
        atomicAdd(pr + 4, (unsigned long long)prof.iters);
        atomicAdd(pr + 5, (unsigned long long)prof.outer);
        atomicAdd(pr + 6, (unsigned long long)(clock64() - pt_begin));
        atomicAdd(pr + 7, 1ull);
        atomicAdd(pr + 8, (unsigned long long)(wall_clock64() - pt_wall_begin));
        atomicMax(pr + 9, (unsigned long long)(wall_clock64() - pt_wall_begin));
        atomicMax(pr + 10, (unsigned long long)prof.iters);
        atomicMax(pr + 11, (unsigned long long)prof.p1);
        atomicAdd(pr + 12, (unsigned long long)prof.fb_iters);
        atomicAdd(pr + 13, (unsigned long long)prof.fb_cycles);
    }
#endif
    if (COUNT) {
        wc.nodes += c_nodes;
        wc.spheres += c_sph;
        wc.planars += c_pln;
        wc.fallbacks += c_fallback;
    }
}

template <bool COUNT, bool LDS_NODES, bool BATCH, bool REC16 = false>
__global__ __launch_bounds__(LDS_NODES ? TRAVERSE_LDS_BLOCK : BLOCK) void k_traverse_f32(DScene sc, DState rays, DHits hits, DPool pool,
                                                                                       double tmin, double tmax0, DRenderCtl* ctl,
                                                                                       uint32_t lds_surfaces) {
    // LDS_NODES: the f32 node records of the whole tree, then (lds_surfaces) the sphere and planar tables
    extern __shared__ uint4 lds_nodes[];
    const uint4* lds_spheres = lds_nodes + lds_after_tree(sc.n_nodes);
    const uint4* lds_planars = lds_spheres + 2u * sc.n_spheres;
    if (LDS_NODES) {
        stage_nodes_in_lds(lds_nodes, sc.nodes32, sc.leaf_codes, sc.n_nodes);
        if (lds_surfaces) {
            uint4* dst_s = lds_nodes + lds_after_tree(sc.n_nodes);
            const uint4* src_s = reinterpret_cast<const uint4*>(sc.spheres);
            for (uint32_t i = threadIdx.x; i < 2u * sc.n_spheres; i += blockDim.x) dst_s[i] = src_s[i];
            uint4* dst_p = dst_s + 2u * sc.n_spheres;
            const uint4* src_p = reinterpret_cast<const uint4*>(sc.planars);
            for (uint32_t i = threadIdx.x; i < 8u * sc.n_planars; i += blockDim.x) dst_p[i] = src_p[i];
        }
    }
    // The rays of the block's sub-pools (one per wave of the block), end to
    // end, are the block's work list.  Its waves draw from the list through
    // one LDS cursor, so a wave that meets expensive rays simply draws less
    // often; hits go to the ray's own slot, so it does not matter which wave
    // traced a ray.  (Each wave tracing only its own sub-pool leaves most of a
    // block idle while its slowest wave finishes.)
    __shared__ uint32_t s_first[MAX_BLOCK_WAVES + 1];   // position of the first ray of each of the block's sub-pools
    __shared__ uint32_t s_cursor;
    const uint32_t block_waves = blockDim.x >> 6;
    const uint32_t first_pool = blockIdx.x * block_waves;
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t j = 0; j < block_waves; ++j) {
            s_first[j] = acc;
            acc += pool_count(pool, first_pool + j);
        }
        for (uint32_t j = block_waves; j <= MAX_BLOCK_WAVES; ++j) s_first[j] = acc;
        s_cursor = 0u;
    }
    __syncthreads();
    TraverseWork w{};
    w.lds_nodes = lds_nodes;
    w.lds_base = lds_address(lds_nodes);
    w.lds_spheres = lds_spheres;
    w.lds_planars = lds_planars;
    w.lds_surfaces = lds_surfaces;
    w.s_first = s_first;
    w.s_cursor = &s_cursor;
    w.first_pool = first_pool;
    w.sub_capacity = pool.sub_capacity;
    WorkCounters wc;
    traverse_f32_wave<COUNT, LDS_NODES, true, BATCH, REC16>(sc, rays, hits, w, tmin, tmax0, ctl, wc);
    if (COUNT && ctl) {
        atomicAdd(&ctl->node_visits, wc.nodes);
        atomicAdd(&ctl->sphere_tests, wc.spheres);
        atomicAdd(&ctl->planar_tests, wc.planars);
        atomicAdd(&ctl->box_fallbacks, wc.fallbacks);
    }
}

// ---------------------------------------------------------------- shade
// Schlick, src/material.rs:134-137; powi(5) = x * ((x*x)*(x*x))
// (r0 = schlick_r0(refraction_index) comes with the surface's record)
__device__ __forceinline__ double reflectance(double cos_theta, double r0) {
    double x = 1.0 - cos_theta;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1.0 - r0) * x5;
}

// Sphere u,v (geometry.rs:236-240), from the world-space point as the reference has it.  Out of line: no
// example scene puts UVGradient on a sphere, and inlined the acos/atan2 polynomials keep two dozen
// registers of coefficients alive through the whole shade loop.
// (The pair is returned by value: reference arguments of an out-of-line function live in scratch memory.)
__device__ __noinline__ double2 sphere_uv(double px, double py, double pz) {
    const double PI = 3.14159265358979323846264338327950288;
    const double theta = acos(-py);
    const double phi = atan2(-pz, px) + PI;
    return make_double2(phi / (2.0 * PI), theta / PI);
}

// ---- memory operations of the shade phase in the kernel form that reads a whole row ahead (shade_wave<true>)
// A wave's memory operations complete in the order they were issued, and `s_waitcnt vmcnt(N)` waits until at most N are
// outstanding.  A row's loads are issued one row ahead; between them and their first use lie the previous row's atomics and
// stores.  Written in C++ those sit in conditional blocks, the compiler must assume them skipped, waits for `vmcnt(0)` and with
// that for the round trip of stores issued a moment ago, once per row.  Here every row issues exactly ROW_TAIL_OPS operations
// after its loads (lanes that have nothing to add or store are masked out, the instruction is still issued and counted), so the
// wait for the row's loads is `vmcnt(ROW_TAIL_OPS)` and leaves the stores in flight.  The compiler does not know of these loads:
// their results may only be used after row_wait().
// This rests on two properties: (1) a vector memory instruction issued with an empty EXEC mask (a row without survivors, or
// without anything to add) still counts up vmcnt and counts it down again in order -- the ISA's counters are per instruction
// issued, not per lane served; a GPU test renders frames whose rows are all of that kind -- and (2) the compiler never reads,
// copies or spills the destination registers of a row in flight, which tools/check_row_regs.py verifies on the compiler's
// assembly output as a step of the build.
constexpr int ROW_TAIL_OPS = 16;   // 3 atomics + 13 stores
struct RowAhead {
    uint32_t handle, pixel, sample, draw, depth;
    double t;
    V3 o, d, beta;
};
// issues the 15 loads of ray `i`'s state and hit record, for the lanes of `mask`
__device__ __forceinline__ void row_request(RowAhead& r, unsigned long long mask, uint32_t i, const DState& cur, const DHits& hits) {
    unsigned long long addr, saved;
    const unsigned long long i64 = i;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mk]\n\t"
        "v_lshl_add_u64 %[ad], %[i], 2, %[p0]\n\t"
        "global_load_dword %[a0], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 2, %[p1]\n\t"
        "global_load_dword %[a1], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 2, %[p2]\n\t"
        "global_load_dword %[a2], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 2, %[p3]\n\t"
        "global_load_dword %[a3], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 2, %[p4]\n\t"
        "global_load_dword %[a4], %[ad], off nt\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a0] "+v"(r.handle), [a1] "+v"(r.pixel), [a2] "+v"(r.sample), [a3] "+v"(r.draw), [a4] "+v"(r.depth), [ad] "=&v"(addr), [sv] "=&s"(saved)
        : [mk] "s"(mask), [i] "v"(i64), [p0] "s"(hits.prim), [p1] "s"(cur.pixel), [p2] "s"(cur.sample), [p3] "s"(cur.draw), [p4] "s"(cur.depth)
        : "memory");
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mk]\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p0]\n\t"
        "global_load_dwordx2 %[a0], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p1]\n\t"
        "global_load_dwordx2 %[a1], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p2]\n\t"
        "global_load_dwordx2 %[a2], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p3]\n\t"
        "global_load_dwordx2 %[a3], %[ad], off nt\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a0] "+v"(r.t), [a1] "+v"(r.o.x), [a2] "+v"(r.o.y), [a3] "+v"(r.o.z), [ad] "=&v"(addr), [sv] "=&s"(saved)
        : [mk] "s"(mask), [i] "v"(i64), [p0] "s"(hits.t), [p1] "s"(cur.ox), [p2] "s"(cur.oy), [p3] "s"(cur.oz)
        : "memory");
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mk]\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p0]\n\t"
        "global_load_dwordx2 %[a0], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p1]\n\t"
        "global_load_dwordx2 %[a1], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p2]\n\t"
        "global_load_dwordx2 %[a2], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p3]\n\t"
        "global_load_dwordx2 %[a3], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p4]\n\t"
        "global_load_dwordx2 %[a4], %[ad], off nt\n\t"
        "v_lshl_add_u64 %[ad], %[i], 3, %[p5]\n\t"
        "global_load_dwordx2 %[a5], %[ad], off nt\n\t"
        "s_mov_b64 exec, %[sv]"
        : [a0] "+v"(r.d.x), [a1] "+v"(r.d.y), [a2] "+v"(r.d.z), [a3] "+v"(r.beta.x), [a4] "+v"(r.beta.y), [a5] "+v"(r.beta.z), [ad] "=&v"(addr),
          [sv] "=&s"(saved)
        : [mk] "s"(mask), [i] "v"(i64), [p0] "s"(cur.dx), [p1] "s"(cur.dy), [p2] "s"(cur.dz), [p3] "s"(cur.bx), [p4] "s"(cur.by), [p5] "s"(cur.bz)
        : "memory");
}
// Waits for the loads of row_request() and hands the row over in registers of its own, so that `ahead` can be requested again at
// once.  The copies are part of the statement that waits: a copy the compiler made by itself could be scheduled before the wait,
// while the registers are still being written.  TAIL: a row_tail() has been issued since the request, and may stay in flight.
template <bool TAIL>
__device__ __forceinline__ RowAhead row_wait(const RowAhead& ahead) {
    static_assert(ROW_TAIL_OPS == 16, "the waits below name the number of operations of row_tail()");
    RowAhead r;
#define RTX_ROW_WAIT_A(WAIT)                                                                                                            \
    asm volatile(WAIT "\n\tv_mov_b32 %0, %9\n\tv_mov_b32 %1, %10\n\tv_mov_b32 %2, %11\n\tv_mov_b32 %3, %12\n\tv_mov_b32 %4, %13\n\t" \
                      "v_mov_b64 %5, %14\n\tv_mov_b64 %6, %15\n\tv_mov_b64 %7, %16\n\tv_mov_b64 %8, %17"                             \
                 : "=&v"(r.handle), "=&v"(r.pixel), "=&v"(r.sample), "=&v"(r.draw), "=&v"(r.depth), "=&v"(r.t), "=&v"(r.o.x), "=&v"(r.o.y),     \
                   "=&v"(r.o.z)                                                                                                         \
                 : "v"(ahead.handle), "v"(ahead.pixel), "v"(ahead.sample), "v"(ahead.draw), "v"(ahead.depth), "v"(ahead.t), "v"(ahead.o.x), \
                   "v"(ahead.o.y), "v"(ahead.o.z)                                                                                       \
                 : "memory")
#define RTX_ROW_WAIT_B(WAIT)                                                                                                            \
    asm volatile(WAIT "\n\tv_mov_b64 %0, %6\n\tv_mov_b64 %1, %7\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %9\n\tv_mov_b64 %4, %10\n\t"      \
                      "v_mov_b64 %5, %11"                                                                                               \
                 : "=&v"(r.d.x), "=&v"(r.d.y), "=&v"(r.d.z), "=&v"(r.beta.x), "=&v"(r.beta.y), "=&v"(r.beta.z)                              \
                 : "v"(ahead.d.x), "v"(ahead.d.y), "v"(ahead.d.z), "v"(ahead.beta.x), "v"(ahead.beta.y), "v"(ahead.beta.z)                  \
                 : "memory")
    if (TAIL) {
        RTX_ROW_WAIT_A("s_waitcnt vmcnt(16)");
        RTX_ROW_WAIT_B("s_waitcnt vmcnt(16)");
    } else {
        RTX_ROW_WAIT_A("s_waitcnt vmcnt(0)");
        RTX_ROW_WAIT_B("s_waitcnt vmcnt(0)");
    }
#undef RTX_ROW_WAIT_A
#undef RTX_ROW_WAIT_B
    return r;
}
// A row's atomics (radiance of the finished paths into the pixel at `px`, one mask per component) and the 13 stores of the
// surviving rays' state to slot `k` (lanes of `alive`): always ROW_TAIL_OPS operations.
__device__ __forceinline__ void row_tail(unsigned long long add_x, unsigned long long add_y, unsigned long long add_z, double* px, V3 radiance,
                                         unsigned long long alive, uint32_t k, const DState& nxt, V3 p, V3 nd, V3 beta, uint32_t pixel,
                                         uint32_t sample, uint32_t draw, uint32_t depth) {
    unsigned long long addr, saved;
    const unsigned long long k64 = k;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mx]\n\t"
        "global_atomic_add_f64 %[px], %[rx], off\n\t"
        "s_mov_b64 exec, %[my]\n\t"
        "global_atomic_add_f64 %[px], %[ry], off offset:8\n\t"
        "s_mov_b64 exec, %[mz]\n\t"
        "global_atomic_add_f64 %[px], %[rz], off offset:16\n\t"
        "s_mov_b64 exec, %[sv]"
        : [sv] "=&s"(saved)
        : [mx] "s"(add_x), [my] "s"(add_y), [mz] "s"(add_z), [px] "v"(px), [rx] "v"(radiance.x), [ry] "v"(radiance.y), [rz] "v"(radiance.z)
        : "memory");
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mk]\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p0]\n\t"
        "global_store_dwordx2 %[ad], %[a0], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p1]\n\t"
        "global_store_dwordx2 %[ad], %[a1], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p2]\n\t"
        "global_store_dwordx2 %[ad], %[a2], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p3]\n\t"
        "global_store_dwordx2 %[ad], %[a3], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p4]\n\t"
        "global_store_dwordx2 %[ad], %[a4], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p5]\n\t"
        "global_store_dwordx2 %[ad], %[a5], off\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ad] "=&v"(addr), [sv] "=&s"(saved)
        : [mk] "s"(alive), [k] "v"(k64), [p0] "s"(nxt.ox), [p1] "s"(nxt.oy), [p2] "s"(nxt.oz), [p3] "s"(nxt.dx), [p4] "s"(nxt.dy), [p5] "s"(nxt.dz),
          [a0] "v"(p.x), [a1] "v"(p.y), [a2] "v"(p.z), [a3] "v"(nd.x), [a4] "v"(nd.y), [a5] "v"(nd.z)
        : "memory");
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mk]\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p0]\n\t"
        "global_store_dwordx2 %[ad], %[a0], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p1]\n\t"
        "global_store_dwordx2 %[ad], %[a1], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 3, %[p2]\n\t"
        "global_store_dwordx2 %[ad], %[a2], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 2, %[p3]\n\t"
        "global_store_dword %[ad], %[a3], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 2, %[p4]\n\t"
        "global_store_dword %[ad], %[a4], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 2, %[p5]\n\t"
        "global_store_dword %[ad], %[a5], off\n\t"
        "v_lshl_add_u64 %[ad], %[k], 2, %[p6]\n\t"
        "global_store_dword %[ad], %[a6], off\n\t"
        "s_mov_b64 exec, %[sv]"
        : [ad] "=&v"(addr), [sv] "=&s"(saved)
        : [mk] "s"(alive), [k] "v"(k64), [p0] "s"(nxt.bx), [p1] "s"(nxt.by), [p2] "s"(nxt.bz), [p3] "s"(nxt.pixel), [p4] "s"(nxt.sample),
          [p5] "s"(nxt.draw), [p6] "s"(nxt.depth), [a0] "v"(beta.x), [a1] "v"(beta.y), [a2] "v"(beta.z), [a3] "v"(pixel), [a4] "v"(sample),
          [a5] "v"(draw), [a6] "v"(depth)
        : "memory");
}

// Square roots and reciprocals of a bounce, for the lanes that are here: the bare sequences of rtx_device.h (the library's own
// operations without the scaffolding for operands of extreme magnitude: a third of the instructions) where every lane's
// operand is between 2^-500 and 2^500, the library forms for all lanes otherwise.  The same roundings either way.
__device__ __forceinline__ bool in_bare_range(double x) {   // 2^-500 <= x < 2^500; zero, negative numbers, infinities and NaNs are not
    return (uint32_t)__double2hiint(x) - (523u << 20) < (1000u << 20);
}
__device__ __forceinline__ double sqrt_here(double x) {
    if (__builtin_amdgcn_ballot_w64(!in_bare_range(x)) == 0ull) return sqrt_normal_range(x);
    return sqrt(x);
}
// v / sqrt(v.v) = v * (1.0 / sqrt(v.v)): Vector3::to_unit and the end of Vector3::random_unit (src/vector.rs:39-41, :59-81, :179-190).
// The root of an operand in range lies between 2^-250 and 2^250, where div_refined's bounds hold with room to spare.
__device__ __forceinline__ V3 to_unit_here(V3 v) {
    const double l = length_squared(v);
    double inv;
    if (__builtin_amdgcn_ballot_w64(!in_bare_range(l)) == 0ull) {
        const double root = sqrt_normal_range(l);
        inv = div_refined(1.0, root, refined_rcp(root));
    } else {
        inv = 1.0 / sqrt(l);
    }
    return v * inv;
}
__device__ __forceinline__ V3 refract_here(V3 r_in, V3 n, double eta) {   // refract of rtx_device.h (src/vector.rs:258-264) with sqrt_here
    const double cos_theta = clampd(dot(-r_in, n), -1.0, 1.0);
    const V3 r_out_perp = eta * (r_in + cos_theta * n);
    const V3 r_out_par = -sqrt_here(fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_par;
}

// What one bounce does to the rays a wave holds (`active`: this lane holds one): emitted light and the sky for the rays that
// end here (`add`, `radiance`: to be added to the ray's pixel) and Material::scatter for the others (`alive`: the path goes on
// from `p` along `nd` with `beta`, `draw` and `depth`).  Must be called by all 64 lanes: Vector3::random_unit is run for the wave.
struct Bounce {
    bool alive, add;
    V3 p, nd, beta, radiance;
    uint32_t draw, depth;
};
__device__ __forceinline__ Bounce bounce_wave(const DScene& sc, const uint4* lds_recs, uint64_t seed, uint32_t max_depth, V3 background, bool active,
                                              uint32_t handle, double t, V3 o, V3 d, V3 beta, uint32_t pixel, uint32_t sample, uint32_t draw0,
                                              uint32_t depth0) {
    bool alive = false;
    V3 p = v3(0, 0, 0), nd = v3(0, 0, 0);
    uint32_t draw = 0, depth = 0;
    // the record of the surface that was hit: requested as soon as the hit is known, so that it arrives during the sampling below
    double2 r0 = make_double2(0, 0), r1 = r0, r2 = r0, r3 = r0;
    if (active && handle != NONE) {
        const uint32_t rec = code_geom(handle) != 0u ? sc.n_spheres + code_slot(handle) : code_slot(handle);
        if (lds_recs) {   // wave-uniform
            typedef __attribute__((address_space(3))) const double* lds_ptr;
            const lds_ptr rp = (lds_ptr)reinterpret_cast<const double*>(lds_recs + 4u * rec);
            r0 = make_double2(rp[0], rp[1]); r1 = make_double2(rp[2], rp[3]); r2 = make_double2(rp[4], rp[5]); r3 = make_double2(rp[6], rp[7]);
        } else {
            const double2* rp = reinterpret_cast<const double2*>(sc.shade_recs + rec);
            r0 = rp[0]; r1 = rp[1]; r2 = rp[2]; r3 = rp[3];
        }
    }
    // Material::scatter (material.rs:20-33) is one sampling event: it starts on a block boundary of the path's
    // stream.  Lambertian and Metal both start with Vector3::random_unit() (material.rs:62, :86), run here for the wave.
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t draw_ev = (draw0 + 1u) & ~1u;
    V3 ru_cand;
    uint32_t ru_trials;
    random_unit_wave(active && handle != NONE && code_mat(handle) <= 1u, k0, k1, pixel, sample, draw_ev >> 1, ru_cand, ru_trials);
#ifdef RTX_PROBE_RANDOM_UNIT_TWICE   // timing probe (same image): the sampler run a second time on another part of the stream (+11.7 % on the headline frame)
    {
        V3 again;
        uint32_t trials_again;
        random_unit_wave(active && handle != NONE && code_mat(handle) <= 1u, k0, k1, pixel, sample, (draw_ev >> 1) + 100000u, again, trials_again);
        if (trials_again == 0xFFFFFFFFu) ru_cand = again;   // (never: keeps the second run alive)
    }
#endif
    V3 radiance = v3(0.0, 0.0, 0.0);
    bool add = false;
    if (active) {
        if (handle == NONE) {
            radiance = beta * background;   // camera.rs:207
            add = true;
        } else {
            const uint32_t mkind = code_mat(handle), slot = code_slot(handle);
            const bool planar = code_geom(handle) != 0u;
            const V3 rec_a = v3(r0.x, r0.y, r1.x);
            const V3 rgb = v3(r2.x, r2.y, r3.x);
            const double param = r3.y;
            // complete the Hit for the winning surface (geometry.rs:232-249, 280-288)
            p = o + d * t;
            V3 outward;
            double alpha = 0.0, beta_uv = 0.0;
            if (!planar) {
                outward = (p - rec_a) * r1.y;   // (p - center) / radius
                if (mkind == 4u) {   // u,v only feed UVGradient
                    const double2 uv = sphere_uv(p.x, p.y, p.z);
                    alpha = uv.x;
                    beta_uv = uv.y;
                }
            } else {
                outward = rec_a;
                if (mkind == 4u) {
                    const DPlanar& g = sc.planars[slot];
                    V3 qp = p - ld3(g.q);
                    V3 w = ld3(g.w);
                    alpha = dot(w, cross(qp, ld3(g.v)));
                    beta_uv = dot(w, cross(ld3(g.u), qp));
                }
            }
            const bool front_face = dot(d, outward) < 0.0;           // geometry.rs:135-145
            const V3 n = front_face ? outward : -outward;

            // Material::emitted, material.rs:35-54
            if (mkind == 3u) {
                radiance = beta * rgb;
                add = true;
            } else if (mkind == 4u) {
                double a0 = 0.0 - alpha, a1 = 1.0 - alpha, b0 = 0.0 - beta_uv, b1 = 1.0 - beta_uv;
                V3 e = v3(1.0 - sqrt(a0 * a0 + b0 * b0), 1.0 - sqrt(a1 * a1 + b0 * b0), 1.0 - sqrt(a0 * a0 + b1 * b1)) * param;
                radiance = beta * e;
                add = true;
            }

            // Material::scatter, material.rs:20-33
            bool scattered = false;
            V3 att = v3(1.0, 1.0, 1.0);
            // One normalisation serves three materials: Vector3::random_unit's `candidate / lensq.sqrt()` (vector.rs:59-81) for
            // Lambertian and Metal and Dielectric's `ray.direction.to_unit()` (material.rs:110) are the same operations,
            // v * (1 / sqrt(v.v)), on different vectors, and a row of rays usually holds all three materials.
            V3 unit = v3(0.0, 0.0, 0.0);
            if (mkind <= 2u) {
                const V3 v = mkind == 2u ? d : ru_cand;
                unit = to_unit_here(v);
            }
            if (mkind <= 1u) {
                const V3 ru = unit;
                draw_ev += 2u * ru_trials;
                att = rgb;
                if (mkind == 0u) {        // lambertian, material.rs:61-73
                    V3 direction = n + ru;
                    if (is_near_zero(direction)) direction = n;
                    nd = direction;
                    scattered = true;
                } else {                   // metal, material.rs:84-96
                    V3 reflected = reflect(d, n);
                    V3 fuzz = ru * param;
                    V3 fuzzed = to_unit_here(reflected) + fuzz;
                    nd = fuzzed;
                    scattered = dot(fuzzed, n) > 0.0;
                }
            } else if (mkind == 2u) {     // dielectric, material.rs:108-130
                V3 r_in = unit;
                double eta = front_face ? rgb.x : param;        // 1.0 / ri : ri
                double cos_theta = clampd(dot(-r_in, n), -1.0, 1.0);
                double sin_theta = sqrt_here(1.0 - cos_theta * cos_theta);
                bool refl = eta * sin_theta > 1.0;
                if (!refl) {   // no draw on total internal reflection
                    refl = reflectance(cos_theta, front_face ? rgb.y : rgb.z) > first_uniform_of_block(k0, k1, pixel, sample, draw_ev >> 1);
                    ++draw_ev;
                }
                nd = refl ? reflect(r_in, n) : refract_here(r_in, n, eta);
                scattered = true;
            }
            depth = depth0 + 1u;
            if (scattered && depth < max_depth) {
                beta = beta * att;
                draw = draw_ev;
                alive = true;
            }
        }
    }
    Bounce out;
    out.alive = alive; out.add = add; out.p = p; out.nd = nd; out.beta = beta; out.radiance = radiance; out.draw = draw; out.depth = depth;
    return out;
}

// One step of ray_color (src/camera.rs:184-208) in its iterative form:
//   miss            -> add beta * background, path ends
//   hit             -> add beta * emitted; scatter ? (beta *= attenuation, new ray) : path ends
//   depth exhausted -> path ends contributing nothing more
// The reference nests `emitted + ray_color(..) * attenuation`; with emission
// only on surfaces that do not scatter this is the same product of
// attenuations, associated from the other end.
// The wave walks `count` rays from slot `base` on, 64 at a time, and appends
// the survivors, in order, from slot `base` of `nxt` on (the other buffer, or the
// same one: a row's survivors never go behind the row's own slots).  Returns
// the number of survivors.
template <bool DEEP>
__device__ __forceinline__ uint32_t shade_wave(const DScene& sc, const DCamera& cam, const DState& cur, const DHits& hits, uint32_t base,
                                               uint32_t count, const DState& nxt, uint64_t seed, double* accum, const uint4* lds_recs = nullptr) {
    const uint32_t lane = lane_id();
    uint32_t n_out = 0;   // survivors written so far (wave-uniform)
    // What the sampling at the top of a row needs (hit, pixel, sample, draw count) is read one row ahead, so a row starts
    // computing at once while the rest of its state and the record of its surface are on their way.
    uint32_t handle_n = NONE, pixel_n = 0, sample_n = 0, draw_n = 0;
    // DEEP (the kernel form with registers to spare): the whole state of the next row is read one row ahead, by the row_* helpers
    RowAhead ahead;
    ahead.handle = NONE; ahead.pixel = 0; ahead.sample = 0; ahead.draw = 0; ahead.depth = 0; ahead.t = 0.0;
    ahead.o = v3(0, 0, 0); ahead.d = v3(0, 0, 0); ahead.beta = v3(0, 0, 0);
    // what every row reads of the camera, once (through the reference these were loads in every row)
    const uint32_t max_depth = uniform32(cam.max_depth);
    const V3 background = v3(uniform_f64(cam.background[0]), uniform_f64(cam.background[1]), uniform_f64(cam.background[2]));
    if (DEEP) {
        row_request(ahead, ballot64(lane < count), base + lane, cur, hits);
    } else if (lane < count) {
        handle_n = ld_stream(&hits.prim[base + lane]);
        pixel_n = ld_stream(&cur.pixel[base + lane]);
        sample_n = ld_stream(&cur.sample[base + lane]);
        draw_n = ld_stream(&cur.draw[base + lane]);
    }
    for (uint32_t t0 = 0; t0 < count; t0 += 64u) {
        const uint32_t off = t0 + lane;
        const uint32_t i = base + off;
        V3 beta = v3(0, 0, 0);
        uint32_t handle, pixel, sample, draw0, depth0 = 0;
        double t = 0.0;
        V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
        if (DEEP) {
            // (lanes beyond `count` carry what an earlier row left: everything below asks off < count first)
            const RowAhead row = t0 == 0u ? row_wait<false>(ahead) : row_wait<true>(ahead);   // every row but the first follows a row_tail()
            handle = row.handle; pixel = row.pixel; sample = row.sample; draw0 = row.draw; depth0 = row.depth;
            t = row.t; o = row.o; d = row.d; beta = row.beta;
            row_request(ahead, ballot64(off + 64u < count), i + 64u, cur, hits);
        } else {
            handle = handle_n; pixel = pixel_n; sample = sample_n; draw0 = draw_n;
            if (off < count) {
                t = ld_stream(&hits.t[i]);
                o = v3(ld_stream(&cur.ox[i]), ld_stream(&cur.oy[i]), ld_stream(&cur.oz[i]));
                d = v3(ld_stream(&cur.dx[i]), ld_stream(&cur.dy[i]), ld_stream(&cur.dz[i]));
                beta = v3(ld_stream(&cur.bx[i]), ld_stream(&cur.by[i]), ld_stream(&cur.bz[i]));
                depth0 = ld_stream(&cur.depth[i]);
            }
            handle_n = NONE;
            if (off + 64u < count) {
                handle_n = ld_stream(&hits.prim[i + 64u]);
                pixel_n = ld_stream(&cur.pixel[i + 64u]);
                sample_n = ld_stream(&cur.sample[i + 64u]);
                draw_n = ld_stream(&cur.draw[i + 64u]);
            }
        }
        const Bounce bo = bounce_wave(sc, lds_recs, seed, max_depth, background, off < count, handle, t, o, d, beta, pixel, sample, draw0, depth0);
        const bool alive = bo.alive, add = bo.add;
        const V3 p = bo.p, nd = bo.nd, radiance = bo.radiance;
        const uint32_t draw = bo.draw, depth = bo.depth;
        beta = bo.beta;
        if (off < count) {
            if (!DEEP && add) {
                double* px = accum + 3 * (size_t)pixel;
                if (radiance.x != 0.0) unsafeAtomicAdd(px + 0, radiance.x);
                if (radiance.y != 0.0) unsafeAtomicAdd(px + 1, radiance.y);
                if (radiance.z != 0.0) unsafeAtomicAdd(px + 2, radiance.z);
            }
        }
        // compaction: 64-wide ballot + prefix popcount; the wave is the only writer of its sub-pool
        const unsigned long long amask = ballot64(alive);
        if (DEEP) {
            row_tail(ballot64(add && radiance.x != 0.0), ballot64(add && radiance.y != 0.0), ballot64(add && radiance.z != 0.0),
                     accum + 3 * (size_t)pixel, radiance, amask, base + n_out + rank_below(amask, lane), nxt, p, nd, beta, pixel, sample, draw, depth);
        } else if (alive) {
            const uint32_t k = base + n_out + rank_below(amask, lane);
            st_stream(&nxt.ox[k], p.x); st_stream(&nxt.oy[k], p.y); st_stream(&nxt.oz[k], p.z);
            st_stream(&nxt.dx[k], nd.x); st_stream(&nxt.dy[k], nd.y); st_stream(&nxt.dz[k], nd.z);
            st_stream(&nxt.bx[k], beta.x); st_stream(&nxt.by[k], beta.y); st_stream(&nxt.bz[k], beta.z);
            st_stream(&nxt.pixel[k], pixel);
            st_stream(&nxt.sample[k], sample);
            st_stream(&nxt.draw[k], draw);
            st_stream(&nxt.depth[k], depth);
        }
        n_out += (uint32_t)__popcll(amask);
    }
    return n_out;
}

// Every wave is on its own sub-pool.
__global__ __launch_bounds__(BLOCK, RTX_SHADE_WAVES) void k_shade(DScene sc, DCamera cam, DState cur, DHits hits, DPool pool,
                                                                  DState nxt, uint32_t* counts_out, uint64_t seed, double* accum) {
    const WaveSlot ws = wave_slot(pool);
    const uint32_t n_out = shade_wave<false>(sc, cam, cur, hits, ws.base, ws.count, nxt, seed, accum);
    if (lane_id() == 0 && ws.sub < pool.n_sub) counts_out[ws.sub] = n_out;
}

// ---------------------------------------------------------------- whole render in one launch
// Every wave owns a sub-pool and cycles through top-up -> closest hit -> shade
// on it until the call's paths are used up and its sub-pool is empty.  Waves
// never wait for each other (a slow wave just completes fewer rounds), the
// host is not involved between rounds, and at any moment the waves of a SIMD
// are in different phases, so the closest-hit walk (bound by VALU issue)
// overlaps shading (bound by memory latency) without any scheduling effort.
// The only shared words are the path counter (one returning atomic per wave
// and round) and the accumulation buffer.  A sub-pool is written and read by
// its own wave only; workgroup-scope fences order the phases.
// The three phases of the single-launch kernel.  They are inlined into it: as separate functions each saved and restored
// 28 callee-saved vector registers per call (7 KB of scratch traffic per wave and call, 8 % of the frame's memory traffic;
// inlined the frame is 4 % faster).  -DRTX_CALL_PHASES builds them as calls again.
#ifdef RTX_CALL_PHASES
#define RTX_PHASE_INLINE __noinline__
#else
#define RTX_PHASE_INLINE __forceinline__
#endif
__device__ __forceinline__ const DRenderParams* uniform_ptr(const DRenderParams* p) {
    typedef __attribute__((address_space(1))) const DRenderParams* global_ptr;
    unsigned long long v = uniform64((unsigned long long)p);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    // A phase reads its tables through a pointer the compiler cannot see through: otherwise the table pointers of all three
    // phases are loaded once before the round loop and stay in (or are spilled from) scalar registers for the whole launch.
    asm volatile("" : "+s"(lo), "+s"(hi));
    v = ((unsigned long long)hi << 32) | lo;
    return (const DRenderParams*)(global_ptr)v;
}
// Copies of the parameter record's tables with every pointer in scalar registers.  Loaded through the
// (uniform) parameter pointer they would otherwise sit in vector registers, two per pointer, and spill.
// (The round trip through the global address space tells the compiler what a pointer read from memory cannot:
// that it points to device memory, so accesses become global_* with a scalar base instead of flat_*.)
template <class T>
__device__ __forceinline__ T* uniform_p(T* p) {
    typedef __attribute__((address_space(1))) T* global_ptr;
    return (T*)(global_ptr)uniform64((unsigned long long)p);
}
__device__ __forceinline__ DState uniform_state(const DState& s) {
    DState u;
    u.ox = uniform_p(s.ox); u.oy = uniform_p(s.oy); u.oz = uniform_p(s.oz);
    u.dx = uniform_p(s.dx); u.dy = uniform_p(s.dy); u.dz = uniform_p(s.dz);
    u.bx = uniform_p(s.bx); u.by = uniform_p(s.by); u.bz = uniform_p(s.bz);
    u.pixel = uniform_p(s.pixel); u.sample = uniform_p(s.sample); u.draw = uniform_p(s.draw); u.depth = uniform_p(s.depth);
    return u;
}
__device__ __forceinline__ DHits uniform_hits(const DHits& h) {
    DHits u;
    u.prim = uniform_p(h.prim);
    u.t = uniform_p(h.t);
    return u;
}
__device__ __forceinline__ DScene uniform_scene(const DScene& s) {
    DScene u = s;
    u.nodes = uniform_p(s.nodes); u.spheres = uniform_p(s.spheres); u.planars = uniform_p(s.planars);
    u.nodes32 = uniform_p(s.nodes32); u.leaf_codes = uniform_p(s.leaf_codes);
    u.nodes16 = uniform_p(s.nodes16);
    u.boxes64 = uniform_p(s.boxes64); u.shade_recs = uniform_p(s.shade_recs); u.rec_user_id = uniform_p(s.rec_user_id);
    u.n_nodes = uniform32(s.n_nodes); u.filter_ok = uniform32(s.filter_ok); u.batch_undecided = uniform32(s.batch_undecided); u.pend_at = uniform32(s.pend_at); u.refill_at = uniform32(s.refill_at); u.bounce_at = uniform32(s.bounce_at);
    u.n_spheres = uniform32(s.n_spheres); u.n_planars = uniform32(s.n_planars); u.assume_in_count = uniform32(s.assume_in_count);
    return u;
}
__device__ RTX_PHASE_INLINE void phase_generate(const DRenderParams* pp, uint32_t parity, uint32_t slot0, unsigned long long first, uint32_t got,
                                                uint32_t region_base, uint32_t region_npix) {
    const DRenderParams* P = uniform_ptr(pp);
    parity = uniform32(parity);
    slot0 = uniform32(slot0);
    got = uniform32(got);
    first = uniform64(first);
    generate_wave<false>(P->cam, uniform_state(P->st[parity]), slot0, first, got, uniform64(P->seed), uniform32(P->spp_begin), uniform32(P->tile_order),
                         uniform32(region_base), uniform32(region_npix));
}
template <bool COUNT, bool LDS_NODES, bool BATCH, bool REC16>
__device__ RTX_PHASE_INLINE void phase_traverse(const DRenderParams* pp, uint32_t parity, uint32_t base, uint32_t count, WorkCounters* wc) {
    extern __shared__ uint4 lds_nodes[];
    const DRenderParams* P = uniform_ptr(pp);
    parity = uniform32(parity);
    TraverseWork w{};
    w.lds_nodes = lds_nodes;
    w.lds_base = uniform32(lds_address(lds_nodes));
    w.lds_spheres = lds_nodes + lds_after_tree(P->sc.n_nodes);
    w.lds_planars = w.lds_spheres + 2u * P->sc.n_spheres;
    w.lds_surfaces = P->lds_surfaces;
    w.base = uniform32(base);
    w.count = uniform32(count);
    WorkCounters local;
    traverse_f32_wave<COUNT, LDS_NODES, false, BATCH, REC16>(uniform_scene(P->sc), uniform_state(P->st[parity]), uniform_hits(P->hits), w,
                                               __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(P->tmin))),
                                               __longlong_as_double((long long)uniform64((unsigned long long)__double_as_longlong(P->tmax0))), uniform_p(P->ctl), local);
    if (COUNT) {
        wc->nodes += local.nodes;
        wc->spheres += local.spheres;
        wc->planars += local.planars;
        wc->fallbacks += local.fallbacks;
    }
}
template <bool DEEP>
__device__ RTX_PHASE_INLINE uint32_t phase_shade(const DRenderParams* pp, uint32_t parity, uint32_t base, uint32_t count) {
    const DRenderParams* P = uniform_ptr(pp);
    parity = uniform32(parity);
    base = uniform32(base);
    count = uniform32(count);
    // The survivors are compacted in place: a row's survivors go to slots at or before the row's own, which no later row reads
    // (the row header read one row ahead lies behind them as well).  One buffer instead of two halves the state's footprint in the
    // last-level cache and the scalar registers that hold table pointers (+1.5 % on the frame).
    const DState st = uniform_state(P->st[parity]);
    extern __shared__ uint4 lds_nodes[];
    const uint4* lds_recs = uniform32(P->lds_recs) ? lds_nodes + lds_after_tree(P->sc.n_nodes) + 2u * P->sc.n_spheres + 8u * P->sc.n_planars : nullptr;
    return shade_wave<DEEP>(uniform_scene(P->sc), P->cam, st, uniform_hits(P->hits), base, count, st, uniform64(P->seed), uniform_p(P->accum), lds_recs);
}

// The end of a call, for one wave: the call has no more paths to start and at most a row of rays is left in the wave's
// sub-pool.  Those rays stay in the lanes' registers from here to the end of their paths: closest hit, bounce, closest hit ...
// with nothing written to the pool in between.  Through the pool a round of a nearly empty sub-pool is five trips to memory
// (rays in, hits out, hits and state in, survivors out, each fenced), and the deepest path's rounds, one after the other, are the
// tail every call ends with.  Returns the number of rays traced; results are those of the pooled rounds bit for bit (the same
// walk_pass and bounce_wave on the same rays).
template <bool COUNT, bool LDS_NODES, bool BATCH, bool REC16>
__device__ RTX_PHASE_INLINE unsigned long long phase_tail(const DRenderParams* pp, uint32_t parity, uint32_t base, uint32_t count, WorkCounters* wc) {
    extern __shared__ uint4 lds_nodes[];
    const DRenderParams* P = uniform_ptr(pp);
    parity = uniform32(parity);
    base = uniform32(base);
    count = uniform32(count);
    const DScene sc = uniform_scene(P->sc);
    const DState st = uniform_state(P->st[parity]);
    TraverseWork w{};
    w.lds_nodes = lds_nodes;
    w.lds_base = uniform32(lds_address(lds_nodes));
    w.lds_spheres = lds_nodes + lds_after_tree(sc.n_nodes);
    w.lds_planars = w.lds_spheres + 2u * sc.n_spheres;
    w.lds_surfaces = uniform32(P->lds_surfaces);
    const uint4* lds_recs = (LDS_NODES && uniform32(P->lds_recs)) ? lds_nodes + lds_after_tree(sc.n_nodes) + 2u * sc.n_spheres + 8u * sc.n_planars : nullptr;
    const double tmin = uniform_f64(P->tmin), tmax0 = uniform_f64(P->tmax0);
    const uint64_t seed = uniform64(P->seed);
    const uint32_t max_depth = uniform32(P->cam.max_depth);
    const V3 background = v3(uniform_f64(P->cam.background[0]), uniform_f64(P->cam.background[1]), uniform_f64(P->cam.background[2]));
    double* const accum = uniform_p(P->accum);
    const uint32_t root = LDS_NODES ? w.lds_base : 0u;
    const uint32_t end = root + sc.n_nodes * (REC16 ? 16u : 32u);
    const bool scene_filter = sc.filter_ok != 0u;
    const uint32_t lane = lane_id();

    bool have = lane < count;
    const uint32_t k = base + (have ? lane : 0u);
    V3 o = v3(ld_stream(&st.ox[k]), ld_stream(&st.oy[k]), ld_stream(&st.oz[k]));
    V3 d = v3(ld_stream(&st.dx[k]), ld_stream(&st.dy[k]), ld_stream(&st.dz[k]));
    V3 beta = v3(ld_stream(&st.bx[k]), ld_stream(&st.by[k]), ld_stream(&st.bz[k]));
    const uint32_t pixel = ld_stream(&st.pixel[k]), sample = ld_stream(&st.sample[k]);
    uint32_t draw = ld_stream(&st.draw[k]), depth = ld_stream(&st.depth[k]);
    unsigned long long n_rays = 0;
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0, c_fallback = 0;
    WalkProfile prof;
    while (ballot64(have) != 0ull) {
        double tmax = tmax0;
        uint32_t best_leaf = NONE, node = root;
        RayF32 rf;
        if (REC16) rayq16_init(rf, o, d, tmin, tmax0, sc.q_org, sc.q_delta);
        else rayf32_init(rf, o, d, tmin, tmax0);
        RayDiv rdiv;
        rdiv.a = rdiv.ra = 0.0;
        if (!REC16 && sc.n_spheres) ray_div_init(rdiv, d);
        n_rays += (unsigned long long)__popcll(ballot64(have));
        while (ballot64(have && node < end) != 0ull)
            walk_pass<COUNT, LDS_NODES, BATCH, REC16>(sc, w, have, o, d, rdiv, rf, tmin, tmax, best_leaf, node, end, sc.pend_at, scene_filter, c_nodes, c_sph,
                                                      c_pln, c_fallback, prof);
        const Bounce bo = bounce_wave(sc, lds_recs, seed, max_depth, background, have, best_leaf, tmax, o, d, beta, pixel, sample, draw, depth);
        if (have) {
            if (bo.add) {
                double* px = accum + 3 * (size_t)pixel;
                if (bo.radiance.x != 0.0) unsafeAtomicAdd(px + 0, bo.radiance.x);
                if (bo.radiance.y != 0.0) unsafeAtomicAdd(px + 1, bo.radiance.y);
                if (bo.radiance.z != 0.0) unsafeAtomicAdd(px + 2, bo.radiance.z);
            }
            if (bo.alive) {
                o = bo.p; d = bo.nd; beta = bo.beta; draw = bo.draw; depth = bo.depth;
            } else {
                have = false;
            }
        }
    }
    if (COUNT) {
        wc->nodes += c_nodes;
        wc->spheres += c_sph;
        wc->planars += c_pln;
        wc->fallbacks += c_fallback;
    }
    return n_rays;
}

// The block's LDS copy of the scene: the f32 node records, then (lds_surfaces) the sphere and planar tables, then (lds_recs) the
// shade records.  Called by every thread of the block; ends with a barrier.
__device__ __forceinline__ void stage_scene_in_lds(const DRenderParams* P) {
    extern __shared__ uint4 lds_nodes[];
    const DScene& sc = P->sc;
    stage_nodes_in_lds(lds_nodes, sc.nodes32, sc.leaf_codes, sc.n_nodes);
    if (P->lds_surfaces) {
        uint4* dst_s = lds_nodes + lds_after_tree(sc.n_nodes);
        const uint4* src_s = reinterpret_cast<const uint4*>(sc.spheres);
        for (uint32_t i = threadIdx.x; i < 2u * sc.n_spheres; i += blockDim.x) dst_s[i] = src_s[i];
        uint4* dst_p = dst_s + 2u * sc.n_spheres;
        const uint4* src_p = reinterpret_cast<const uint4*>(sc.planars);
        for (uint32_t i = threadIdx.x; i < 8u * sc.n_planars; i += blockDim.x) dst_p[i] = src_p[i];
        if (P->lds_recs) {
            uint4* dst_r = dst_p + 8u * sc.n_planars;
            const uint4* src_r = reinterpret_cast<const uint4*>(sc.shade_recs);
            for (uint32_t i = threadIdx.x; i < 4u * (sc.n_spheres + sc.n_planars); i += blockDim.x) dst_r[i] = src_r[i];
        }
    }
    __syncthreads();
}

// WIDE: one block of 1024 threads per CU (four waves per SIMD instead of five, 128 registers each instead of 96) so that one LDS
// copy of the tree AND the surface tables serves sixteen waves; for scenes whose tables fit in LDS that way but not five times over.
// The fifth wave is given up for surface tests without a trip to memory and for the registers that let the shade phase read a whole
// row ahead: +1.9 % on the headline frame (42.3 against 43.1 ms per 100 spp).
constexpr uint32_t WIDE_BLOCK = 1024;
// First position (of the pixel order) of band `b` of `n`: whole blocks of 32 x 16 pixels, so that the block-by-block order stays intact
__device__ __forceinline__ uint32_t band_start(uint32_t b, uint32_t n, uint32_t npix) {
    if (b >= n) return npix;
    return (uint32_t)(((unsigned long long)npix * b / n) >> 9) << 9;
}
template <bool COUNT, bool LDS_NODES, bool BATCH, bool WIDE = false, bool REC16 = false>
__global__ __launch_bounds__(WIDE ? WIDE_BLOCK : RTX_PERSISTENT_BLOCK, WIDE ? 4 : RTX_PERSISTENT_WAVES) void k_render_persistent(const DRenderParams* P) {
    if (LDS_NODES) stage_scene_in_lds(P);
    const uint32_t wave = uniform32((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_sub = uniform32(P->n_sub);
    if (wave >= n_sub) return;
    const uint32_t lane = lane_id();
    const uint32_t cap = uniform32(P->cap);
    const uint32_t base = wave * cap;
    DRenderCtl* const ctl = uniform_p(P->ctl);
    const unsigned long long total = uniform64(ctl->total_paths);
    WorkCounters wc;
    unsigned long long n_rays = 0;
    uint32_t rounds = 0, count = 0, parity = 0;
    bool exhausted = false;
    // bands of the image with a path counter each (trees walked from global memory: see the top-up below)
    const uint32_t n_regions = uniform32(P->xcd_bands);
    const uint32_t npix_all = uniform32(P->cam.image_width) * uniform32(P->cam.image_height);
    const uint32_t spp_count = n_regions > 1u ? (uint32_t)(total / npix_all) : 0u;
    uint32_t band = n_regions > 1u ? uniform32(blockIdx.x % n_regions) : 0u, bands_dry = 0u;
#ifdef RTX_PROFILE_PHASES
    long long pp_gen = 0, pp_trav = 0, pp_shade = 0, pp_mark = clock64();
    const long long pp_begin = pp_mark;
#define PP(acc) do { const long long now_ = clock64(); acc += now_ - pp_mark; pp_mark = now_; } while (0)
#else
#define PP(acc) do {} while (0)
#endif
    for (;;) {
        // top up the sub-pool with the next paths of the call
        if (!exhausted && count < cap) {
            if (n_regions <= 1u) {
                const uint32_t need = cap - count;
                unsigned long long first = 0;
                if (lane == 0) first = atomicAdd(&ctl->next_path[0], (unsigned long long)need);
                first = uniform64(first);
                uint32_t got = 0;
                if (first >= total) {
                    exhausted = true;
                } else {
                    const unsigned long long rest = total - first;
                    got = rest < need ? (uint32_t)rest : need;
                    exhausted = rest <= need;
                }
                if (got) phase_generate(P, parity, base + count, first, got, 0u, 0u);
                count += got;
            } else {
                // The image is cut into as many bands as there are XCDs and every band has a path counter of its own.  A wave starts
                // in the band of its block (blocks b and b + 8 run on the same XCD) and moves on to the next band when one runs dry,
                // so an XCD's L2 mostly sees the part of the tree the rays of one band wander through.
                while (count < cap && bands_dry < n_regions) {
                    const uint32_t need = cap - count;
                    const uint32_t band_base = band_start(band, n_regions, npix_all), band_npix = band_start(band + 1u, n_regions, npix_all) - band_base;
                    const unsigned long long band_total = (unsigned long long)band_npix * spp_count;
                    unsigned long long first = band_total;
                    if (lane == 0 && band_total != 0ull) first = atomicAdd(&ctl->band_next[band], (unsigned long long)need);
                    first = uniform64(first);
                    if (first >= band_total) {
                        ++bands_dry;
                        band = band + 1u == n_regions ? 0u : band + 1u;
                        continue;
                    }
                    const unsigned long long rest = band_total - first;
                    const uint32_t got = rest < need ? (uint32_t)rest : need;
                    phase_generate(P, parity, base + count, first, got, band_base, band_npix);
                    count += got;
                }
                exhausted = bands_dry >= n_regions;
            }
        }
        if (count == 0u) break;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        PP(pp_gen);
        if (exhausted && count <= 64u && uniform32(P->tail_in_registers)) {   // the last rays of this wave
            n_rays += phase_tail<COUNT, LDS_NODES, BATCH, REC16>(P, parity, base, count, &wc);
            rounds += 1u;
            break;
        }
        phase_traverse<COUNT, LDS_NODES, BATCH, REC16>(P, parity, base, count, &wc);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        PP(pp_trav);
        n_rays += count;
        ++rounds;
        count = uniform32(phase_shade<WIDE>(P, parity, base, count));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        PP(pp_shade);
    }
#ifdef RTX_PROFILE_PHASES
    if (lane == 0) {
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(ctl + 1) + 16;   // debug area behind the control block
        atomicAdd(pr + 0, (unsigned long long)pp_gen);
        atomicAdd(pr + 1, (unsigned long long)pp_trav);
        atomicAdd(pr + 2, (unsigned long long)pp_shade);
        atomicAdd(pr + 6, (unsigned long long)(clock64() - pp_begin));
        atomicAdd(pr + 7, 1ull);
        atomicMax(pr + 9, (unsigned long long)(clock64() - pp_begin));
    }
#endif
    if (COUNT) {   // the work counters are per lane
        wc.nodes = wave_sum(wc.nodes);
        wc.spheres = wave_sum(wc.spheres);
        wc.planars = wave_sum(wc.planars);
        wc.fallbacks = wave_sum(wc.fallbacks);
    }
    if (lane == 0) {
        atomicAdd(&ctl->rays, n_rays);
        atomicMax(&ctl->rounds_max, rounds);
        if (COUNT) {
            atomicAdd(&ctl->node_visits, wc.nodes);
            atomicAdd(&ctl->sphere_tests, wc.spheres);
            atomicAdd(&ctl->planar_tests, wc.planars);
            atomicAdd(&ctl->box_fallbacks, wc.fallbacks);
        }
    }
}

// ---------------------------------------------------------------- whole render in one launch, paths resident in registers
// The same sample loop with nothing of a path in memory between bounces: a lane keeps its path (ray, attenuation, pixel, sample,
// draw count, depth) in registers from the primary ray to the path's end and then starts the next path of the call in the same
// registers.  The wave alternates between the closest-hit pass (walk_pass, as in the kernels above) and, once enough lanes have
// finished their walk, one bounce for those lanes (bounce_wave) followed by new primary rays for the lanes whose paths ended.
// There is no pool, no compaction and no hit record: per ray the kernel touches memory only to add finished paths into the
// image, against ~300 bytes per ray of state the pooled form moves.  What it gives up: the bounce and the primary-ray code run for
// the lanes that are ready (half to all of a wave) instead of for dense rows, and lanes wait for the batch to fill.  For scenes
// whose tree, surface tables and shade records all sit in LDS (the WIDE form's case).
template <bool BATCH>
__global__ __launch_bounds__(WIDE_BLOCK, 4) void k_render_resident(const DRenderParams* Pin) {
    extern __shared__ uint4 lds_nodes[];
    __shared__ DCamera cam;   // read whenever lanes start new paths: from LDS, not through a pointer to memory
    static_assert(sizeof(DCamera) % 4 == 0, "copied word by word");
    for (uint32_t i = threadIdx.x; i < sizeof(DCamera) / 4; i += blockDim.x)
        reinterpret_cast<uint32_t*>(&cam)[i] = reinterpret_cast<const uint32_t*>(&Pin->cam)[i];
    stage_scene_in_lds(Pin);   // ends with a barrier
    const DRenderParams* P = uniform_ptr(Pin);
    const uint32_t wave = uniform32((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (wave >= uniform32(P->n_sub)) return;
#ifdef RTX_PROBE_WAVES
    if ((threadIdx.x >> 6) >= RTX_PROBE_WAVES) return;   // measurement only (-DRTX_PROBE_WAVES=4 / 8 / 12): the frame's time with one, two or three waves per SIMD
#endif
    const uint32_t lane = lane_id();
    const DScene sc = uniform_scene(P->sc);
    DRenderCtl* const ctl = uniform_p(P->ctl);
    const unsigned long long total = uniform64(ctl->total_paths);
    const unsigned long long npix = (unsigned long long)uniform32(P->cam.image_width) * uniform32(P->cam.image_height);
    const uint64_t seed = uniform64(P->seed);
    const uint32_t spp_begin = uniform32(P->spp_begin);
    const uint32_t max_depth = uniform32(P->cam.max_depth);
    const V3 background = v3(uniform_f64(P->cam.background[0]), uniform_f64(P->cam.background[1]), uniform_f64(P->cam.background[2]));
    const double tmin = uniform_f64(P->tmin), tmax0 = uniform_f64(P->tmax0);
    double* const accum = uniform_p(P->accum);
    const uint4* lds_recs = lds_nodes + lds_after_tree(sc.n_nodes) + 2u * sc.n_spheres + 8u * sc.n_planars;
    TraverseWork w{};
    w.lds_nodes = lds_nodes;
    w.lds_base = uniform32(lds_address(lds_nodes));
    w.lds_spheres = lds_nodes + lds_after_tree(sc.n_nodes);
    w.lds_planars = w.lds_spheres + 2u * sc.n_spheres;
    w.lds_surfaces = 1u;
    const uint32_t root = w.lds_base, end = root + sc.n_nodes * 32u;   // node positions are LDS addresses
    // This wave's staged primary rays (`batch` of them, RESIDENT_STAGE_BYTES each) lie behind the scene's tables
    const uint32_t batch = uniform32(P->resident_batch);
    const uint32_t stage_lds = lds_address(lds_recs + 4u * (sc.n_spheres + sc.n_planars)) + uniform32(threadIdx.x >> 6) * (batch * RESIDENT_STAGE_BYTES);
    uint32_t st_head = 0, st_cnt = 0;   // entries [st_head, st_cnt) are staged and not yet taken (wave-uniform)
    const uint32_t PEND_AT = sc.pend_at, BOUNCE_AT = sc.bounce_at;
    const bool scene_filter = sc.filter_ok != 0u;

    // the lane's path and the ray it is on
    bool have = false;
    V3 o = v3(0, 0, 0), d = v3(1, 1, 1), beta = v3(1, 1, 1);
    uint32_t pixel = 0, sample = 0, draw = 0, depth = 0;
    double tmax = tmax0;
    uint32_t best_leaf = NONE, node = end;
    RayF32 rf;
    rayf32_init(rf, o, d, tmin, tmax0);
    RayDiv rdiv;
    ray_div_init(rdiv, d);
    bool exhausted = false;   // every path of the call has been started or belongs to another wave (wave-uniform, as what follows)
    bool drained = false;     // the call's counter has nothing left for this wave
    unsigned long long chunk_next = 0, chunk_end = 0, left_estimate = total;   // the wave's chunk of path indices
    const uint32_t n_waves = uniform32(P->n_sub);
    unsigned long long n_rays = 0;
    uint32_t events = 0;
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0, c_fallback = 0;
    WalkProfile prof;
#ifdef RTX_PROFILE_PHASES
    prof.mark = clock64();
    const long long pt_begin = prof.mark;
#endif
#ifdef RTX_PROBE_NO_ATOMICS
    V3 probe_sum = v3(0, 0, 0);
#endif
    for (;;) {
        const bool walked = have && node >= end;   // the ray has its closest hit (or none): the path's next bounce is due
        const uint32_t n_walked = (uint32_t)__popcll(ballot64(walked));
        const unsigned long long empty = ballot64(!have);
        const uint32_t n_empty = (uint32_t)__popcll(empty);
        const uint32_t n_walking = 64u - n_walked - n_empty;
        if (n_walking == 0u && n_walked == 0u && exhausted) break;
        // A bounce is worth its instructions once enough lanes take part: those whose walk is over plus, while the call has paths
        // left, the empty ones; towards the end of the call, half of the lanes that still hold a path.
        const uint32_t ready = n_walked + (exhausted ? 0u : n_empty);
        const uint32_t enough = exhausted ? (n_walked + n_walking + 1u) / 2u : BOUNCE_AT;
        if (ready >= enough || n_walking == 0u) {
            ++events;
            bool fresh = false;   // the lane starts on a new ray
            // ---- one bounce of the paths whose walk is over
            const Bounce bo = bounce_wave(sc, lds_recs, seed, max_depth, background, walked, best_leaf, tmax, o, d, beta, pixel, sample, draw, depth);
            if (walked) {
                if (bo.add) {
#ifdef RTX_PROBE_NO_ATOMICS
                    // Pricing probe (wrong image, same work): what do the frame's scattered atomics cost?
                    probe_sum = probe_sum + bo.radiance;
#else
                    double* px = accum + 3 * (size_t)pixel;
                    if (bo.radiance.x != 0.0) unsafeAtomicAdd(px + 0, bo.radiance.x);
                    if (bo.radiance.y != 0.0) unsafeAtomicAdd(px + 1, bo.radiance.y);
                    if (bo.radiance.z != 0.0) unsafeAtomicAdd(px + 2, bo.radiance.z);
#endif
                }
                if (bo.alive) {
                    o = bo.p; d = bo.nd; beta = bo.beta; draw = bo.draw; depth = bo.depth;
                    fresh = true;
                } else {
                    have = false;
                }
            }
            PT(retire);   // (instrumented build: the bounce)
            // ---- the next paths of the call for the lanes without one
            // Primary rays are made 64 at a time (fewer where LDS is short), by the whole wave, and staged in LDS; a lane whose path has ended takes the next
            // staged one (four LDS reads).  Made on demand they cost their ~400 instructions (path index, sample_square, the defocus
            // disk's rejection loop, get_ray) for the third of a wave whose paths ended at this bounce.
            if (!exhausted) {
                unsigned long long vacant = ballot64(!have);
                uint32_t need = (uint32_t)__popcll(vacant);
                for (uint32_t pass = 0; pass < 6u && need != 0u; ++pass) {   // (a batch of 16 serves at most 16 lanes)
                    if (st_head == st_cnt) {   // stage the wave's next paths
                        // The wave draws path indices from the call's counter a chunk at a time (one returning atomic on a single word
                        // per bounce of every wave would be the whole chip's atomic rate several times over); chunks shrink towards the
                        // end of the call so that no wave is left with a long tail of its own.
                        if (chunk_next == chunk_end && !drained) {
                            unsigned long long want = left_estimate / ((unsigned long long)n_waves * 8ull);
                            want = want < 64ull ? 64ull : want > 4096ull ? 4096ull : want;
                            unsigned long long first = 0;
                            if (lane == 0) first = atomicAdd(&ctl->next_path[0], want);
                            first = uniform64(first);
                            if (first >= total) {
                                drained = true;
                            } else {
                                const unsigned long long rest = total - first;
                                chunk_next = first;
                                chunk_end = first + (rest < want ? rest : want);
                                drained = rest <= want;
                                left_estimate = rest;
                            }
                        }
                        const unsigned long long in_chunk = chunk_end - chunk_next;
                        const uint32_t got = in_chunk < (unsigned long long)batch ? (uint32_t)in_chunk : batch;
                        if (got == 0u) break;   // nothing left for this wave
                        const unsigned long long first = chunk_next;
                        chunk_next += got;
                        const bool makes = lane < got;
                        uint32_t new_pixel, sample_off;
                        split_path_index<false>(first + (makes ? lane : 0u), npix, new_pixel, sample_off);
                        const uint32_t new_sample = spp_begin + sample_off;
                        V3 no, nd;
                        uint32_t new_draw;
                        start_path_wave(cam, seed, makes, new_pixel, new_sample, no, nd, new_draw);
                        if (makes) {
                            const uint32_t entry = stage_lds + lane * RESIDENT_STAGE_BYTES;
                            lds_write_d2(entry, 0u, no.x, no.y);
                            lds_write_d2(entry, 1u, no.z, nd.x);
                            lds_write_d2(entry, 2u, nd.y, nd.z);
                            lds_write_u4(entry, 3u, new_pixel, new_sample, new_draw, 0u);
                        }
                        st_head = 0u;
                        st_cnt = got;
#ifdef RTX_PROFILE_PHASES
                        ++prof.stagings;
#endif
                        PT(stage);
                    }
                    const uint32_t avail = st_cnt - st_head;
                    const uint32_t take = need < avail ? need : avail;
                    const uint32_t r = rank_below(vacant, lane);
                    if (!have && r < take) {
                        const uint32_t entry = stage_lds + (st_head + r) * RESIDENT_STAGE_BYTES;
                        lds_read_d2(entry, 0u, o.x, o.y);
                        lds_read_d2(entry, 1u, o.z, d.x);
                        lds_read_d2(entry, 2u, d.y, d.z);
                        const uint4 ids = lds_read_u4(entry, 3u);
                        pixel = ids.x; sample = ids.y; draw = ids.z; depth = 0u;
                        beta = v3(1.0, 1.0, 1.0);
                        have = true;
                        fresh = true;
                    }
                    st_head += take;
                    vacant = ballot64(!have);
                    need = (uint32_t)__popcll(vacant);
                }
                exhausted = drained && chunk_next == chunk_end && st_head == st_cnt;
                PT(take);
            }
            if (fresh) {
                tmax = tmax0;
                best_leaf = NONE;
                node = root;
                rayf32_init(rf, o, d, tmin, tmax0);
                if (sc.n_spheres) ray_div_init(rdiv, d);
            }
            n_rays += (unsigned long long)__popcll(ballot64(fresh));
        }
        PT(refill);   // (instrumented build: the set-up of the lanes' new rays and the head of the loop)
#ifdef RTX_PROFILE_PHASES
        ++prof.outer;
#endif
        walk_pass<false, true, BATCH, false, true>(sc, w, have, o, d, rdiv, rf, tmin, tmax, best_leaf, node, end, PEND_AT, scene_filter, c_nodes, c_sph, c_pln,
                                                   c_fallback, prof);
    }
#ifdef RTX_PROFILE_PHASES
    if (lane == 0) {   // (per-lane counts below are lane 0's: a lane's share of the wave's work)
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(ctl + 1);   // debug area behind the control block
        atomicAdd(pr + 0, (unsigned long long)prof.refill);
        atomicAdd(pr + 1, (unsigned long long)prof.p1);
        atomicAdd(pr + 2, (unsigned long long)prof.p2);
        atomicAdd(pr + 3, (unsigned long long)prof.retire);
        atomicAdd(pr + 4, (unsigned long long)prof.iters);
        atomicAdd(pr + 5, (unsigned long long)prof.outer);
        atomicAdd(pr + 6, (unsigned long long)(clock64() - pt_begin));
        atomicAdd(pr + 7, 1ull);
        atomicAdd(pr + 12, (unsigned long long)events);
        atomicAdd(pr + 13, (unsigned long long)prof.stage);
        atomicAdd(pr + 14, (unsigned long long)prof.take);
        atomicAdd(pr + 15, (unsigned long long)prof.stagings);
    }
    {   // counts kept by whichever lane was the first active one of a step
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(ctl + 1);
        const unsigned long long ws = wave_sum((unsigned long long)prof.wave_steps), ls = wave_sum((unsigned long long)prof.iters);
        if (lane == 0) {
            atomicAdd(pr + 8, (unsigned long long)prof.loop);
            atomicAdd(pr + 9, ws);
            atomicAdd(pr + 10, ls);
        }
    }
#endif
#ifdef RTX_PROBE_NO_ATOMICS
    unsafeAtomicAdd(accum + 3 * (size_t)lane + 0, probe_sum.x);
    unsafeAtomicAdd(accum + 3 * (size_t)lane + 1, probe_sum.y);
    unsafeAtomicAdd(accum + 3 * (size_t)lane + 2, probe_sum.z);
#endif
    if (lane == 0) {
        atomicAdd(&ctl->rays, n_rays);
        atomicMax(&ctl->rounds_max, events);
    }
}

// ---------------------------------------------------------------- output stage
__global__ __launch_bounds__(BLOCK) void k_scale(const double* in, uint64_t n, double scale, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) out[i] = in[i] * scale;
}
// out[i] += in[i]: the sum of two accumulation buffers on one device (the rehearsal form of the multi-GPU frame)
__global__ __launch_bounds__(BLOCK) void k_add(const double* in, uint64_t n, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) out[i] += in[i];
}
// linear_to_gamma + ppm_pixel, src/camera.rs:248-266
__global__ __launch_bounds__(BLOCK) void k_quantize(const double* in, uint64_t n, uint8_t* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        double lin = in[i];
        double g = lin > 0.0 ? sqrt(lin) : 0.0;
        double v = 255.999 * clampd(g, 0.0, 0.999);
        out[i] = (uint8_t)(v > 0.0 ? (v >= 255.0 ? 255.0 : v) : 0.0);
    }
}

// transposes rays [n][3] (API layout) into per-component arrays
__global__ __launch_bounds__(BLOCK) void k_rays_to_state(const double* org, const double* dir, uint64_t n, DState st) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        st.ox[i] = org[3 * i + 0]; st.oy[i] = org[3 * i + 1]; st.oz[i] = org[3 * i + 2];
        st.dx[i] = dir[3 * i + 0]; st.dy[i] = dir[3 * i + 1]; st.dz[i] = dir[3 * i + 2];
    }
}
__global__ __launch_bounds__(BLOCK) void k_user_ids(const uint32_t* rec_user_id, uint32_t n_spheres, uint32_t* ids, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        uint32_t p = ids[i];
        if (p != NONE) ids[i] = rec_user_id[(code_geom(p) != 0u ? n_spheres : 0u) + code_slot(p)];
    }
}

// ---------------------------------------------------------------- self tests
__global__ void k_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (op == 0) out[i] = a[i] / b[i];
        else if (op == 1) out[i] = sqrt(a[i]);
        else if (op == 3) out[i] = (double)((float)a[i] / (float)b[i]);
        else if (op == 4) out[i] = (double)recip32((float)b[i]);   // the reciprocal the first box tier relies on
        else if (op == 5) out[i] = div_refined(a[i], b[i], refined_rcp(b[i]));   // the sphere test's bare division and square root
        else if (op == 6) out[i] = sqrt_normal_range(a[i]);
        else out[i] = a[i] * (1.0 / b[i]);
    }
}
__global__ void k_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PathRng rng(seed, pixel, sample, 0u);
        for (uint32_t i = 0; i < n; ++i) out[i] = rng.next();
    }
}

// ---------------------------------------------------------------- launchers
static inline uint32_t grid_for(uint64_t n, uint32_t max_blocks) {
    uint64_t b = (n + BLOCK - 1) / BLOCK;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (uint32_t)b;
}

// One wave per sub-pool.
static inline uint32_t grid_for_pool(const uint32_t n_sub) { return (n_sub + BLOCK / 64 - 1) / (BLOCK / 64); }

void launch_generate(const DCamera& cam, const DState& st, DRenderCtl* ctl, const uint32_t* surv, uint32_t* live, uint32_t n_sub,
                     uint32_t sub_capacity, uint32_t parity, uint64_t seed, uint32_t spp_begin, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for_pool(n_sub)), dim3(BLOCK), 0, s, cam, st, ctl, surv, live, n_sub, sub_capacity, parity,
                       seed, spp_begin);
}
void launch_primary_rays(const DCamera& cam, uint64_t seed, uint32_t sample, uint32_t npix, double* org, double* dir,
                         uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_primary_rays, dim3(grid_for(npix, max_blocks)), dim3(BLOCK), 0, s, cam, seed, sample, npix, org, dir);
}
void launch_traverse(const DScene& sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                     DRenderCtl* ctl, bool count_work, bool exact, hipStream_t s) {
    dim3 g(grid_for_pool(pool.n_sub)), b(BLOCK);
    if (count_work) {
        if (exact) hipLaunchKernelGGL((k_traverse<true, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<true, false>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl);
    } else {
        if (exact) hipLaunchKernelGGL((k_traverse<false, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<false, false>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl);
    }
}
void launch_traverse_f32(const DScene& sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                         DRenderCtl* ctl, bool count_work, uint32_t lds_mode, hipStream_t s) {
    // lds_mode 0: tree in global memory; 1: tree in LDS; 2: tree and surface tables in LDS
    if (lds_mode) {
        const uint32_t waves_per_block = TRAVERSE_LDS_BLOCK / 64;
        dim3 g((pool.n_sub + waves_per_block - 1) / waves_per_block), b(TRAVERSE_LDS_BLOCK);
        size_t lds = node_tables_lds_bytes(sc.n_nodes);
        const uint32_t surf = lds_mode == 2u ? 1u : 0u;
        if (surf) lds += (size_t)sc.n_spheres * sizeof(DSphere) + (size_t)sc.n_planars * sizeof(DPlanar);
#ifdef RTX_PROFILE_PHASES
        static bool once = false;
        if (!once) {
            once = true;
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_traverse_f32<false, true, false>, TRAVERSE_LDS_BLOCK, lds);
            std::fprintf(stderr, "OCCUPANCY k_traverse_f32<false,true>: %d blocks of %d threads per CU with %zu B LDS; grid %u\n", nb, TRAVERSE_LDS_BLOCK, lds, g.x);
        }
#endif
        const bool batch = sc.batch_undecided != 0u;
        if (count_work && batch) hipLaunchKernelGGL((k_traverse_f32<true, true, true>), g, b, lds, s, sc, rays, hits, pool, tmin, tmax, ctl, surf);
        else if (count_work) hipLaunchKernelGGL((k_traverse_f32<true, true, false>), g, b, lds, s, sc, rays, hits, pool, tmin, tmax, ctl, surf);
        else if (batch) hipLaunchKernelGGL((k_traverse_f32<false, true, true>), g, b, lds, s, sc, rays, hits, pool, tmin, tmax, ctl, surf);
        else hipLaunchKernelGGL((k_traverse_f32<false, true, false>), g, b, lds, s, sc, rays, hits, pool, tmin, tmax, ctl, surf);
    } else {
        dim3 g(grid_for_pool(pool.n_sub)), b(BLOCK);
        const bool batch = sc.batch_undecided != 0u;
        if (sc.use16 && sc.nodes16) {   // the half-size records of a large tree
            if (count_work && batch) hipLaunchKernelGGL((k_traverse_f32<true, false, true, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
            else if (count_work) hipLaunchKernelGGL((k_traverse_f32<true, false, false, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
            else if (batch) hipLaunchKernelGGL((k_traverse_f32<false, false, true, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
            else hipLaunchKernelGGL((k_traverse_f32<false, false, false, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
            return;
        }
        if (count_work && batch) hipLaunchKernelGGL((k_traverse_f32<true, false, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
        else if (count_work) hipLaunchKernelGGL((k_traverse_f32<true, false, false>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
        else if (batch) hipLaunchKernelGGL((k_traverse_f32<false, false, true>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
        else hipLaunchKernelGGL((k_traverse_f32<false, false, false>), g, b, 0, s, sc, rays, hits, pool, tmin, tmax, ctl, 0u);
    }
}
void launch_shade(const DScene& sc, const DCamera& cam, const DState& cur, const DHits& hits, const DPool& pool, const DState& nxt,
                  uint32_t* counts_out, uint64_t seed, double* accum, hipStream_t s) {
    hipLaunchKernelGGL(k_shade, dim3(grid_for_pool(pool.n_sub)), dim3(BLOCK), 0, s, sc, cam, cur, hits, pool, nxt, counts_out, seed, accum);
}
// lds_mode: 0 tree in global memory; 1 tree in LDS; 2 tree and surface tables in LDS; 3 and 4: modes 2 and 1 in the WIDE form of the kernel
size_t persistent_lds_bytes(const DScene& sc, uint32_t lds_mode, bool recs) {
    size_t lds = lds_mode ? node_tables_lds_bytes(sc.n_nodes) : 0;
    if (lds_mode == 2u || lds_mode == 3u) {
        lds += (size_t)sc.n_spheres * sizeof(DSphere) + (size_t)sc.n_planars * sizeof(DPlanar);
        if (recs) lds += ((size_t)sc.n_spheres + sc.n_planars) * sizeof(DShadeRec);
    }
    return lds;
}
uint32_t persistent_waves_per_cu(const DScene& sc, uint32_t lds_mode, bool recs) {
    int nb = 0;
    const size_t lds = persistent_lds_bytes(sc, lds_mode, recs);
    const uint32_t block = lds_mode >= 3u ? WIDE_BLOCK : RTX_PERSISTENT_BLOCK;
    hipError_t e = lds_mode >= 3u ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_persistent<false, true, false, true>, block, lds)
                   : lds_mode  ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_persistent<false, true, false>, block, lds)
                   : (sc.use16 && sc.nodes16) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_persistent<false, false, false, false, true>, block, lds)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_persistent<false, false, false>, block, lds);
    if (e != hipSuccess || nb < 1) nb = 1;
    return (uint32_t)nb * (block / 64);   // waves
}
// `params` is device memory; host_params is the same record on the host (for the grid and LDS size).
void launch_render_persistent(const DRenderParams* params, const DRenderParams& host_params, bool count_work, uint32_t lds_mode, hipStream_t s) {
    const uint32_t block = lds_mode >= 3u ? WIDE_BLOCK : RTX_PERSISTENT_BLOCK;
    const uint32_t block_waves = block / 64;
    dim3 g((host_params.n_sub + block_waves - 1) / block_waves), b(block);
    const size_t lds = persistent_lds_bytes(host_params.sc, lds_mode, host_params.lds_recs != 0u);
    const bool batch = host_params.sc.batch_undecided != 0u;
    if (lds_mode >= 3u) {
        if (count_work && batch) hipLaunchKernelGGL((k_render_persistent<true, true, true, true>), g, b, lds, s, params);
        else if (count_work) hipLaunchKernelGGL((k_render_persistent<true, true, false, true>), g, b, lds, s, params);
        else if (batch) hipLaunchKernelGGL((k_render_persistent<false, true, true, true>), g, b, lds, s, params);
        else hipLaunchKernelGGL((k_render_persistent<false, true, false, true>), g, b, lds, s, params);
    } else if (lds_mode) {
        if (count_work && batch) hipLaunchKernelGGL((k_render_persistent<true, true, true>), g, b, lds, s, params);
        else if (count_work) hipLaunchKernelGGL((k_render_persistent<true, true, false>), g, b, lds, s, params);
        else if (batch) hipLaunchKernelGGL((k_render_persistent<false, true, true>), g, b, lds, s, params);
        else hipLaunchKernelGGL((k_render_persistent<false, true, false>), g, b, lds, s, params);
    } else if (host_params.sc.use16 && host_params.sc.nodes16) {
        if (count_work && batch) hipLaunchKernelGGL((k_render_persistent<true, false, true, false, true>), g, b, 0, s, params);
        else if (count_work) hipLaunchKernelGGL((k_render_persistent<true, false, false, false, true>), g, b, 0, s, params);
        else if (batch) hipLaunchKernelGGL((k_render_persistent<false, false, true, false, true>), g, b, 0, s, params);
        else hipLaunchKernelGGL((k_render_persistent<false, false, false, false, true>), g, b, 0, s, params);
    } else {
        if (count_work && batch) hipLaunchKernelGGL((k_render_persistent<true, false, true>), g, b, 0, s, params);
        else if (count_work) hipLaunchKernelGGL((k_render_persistent<true, false, false>), g, b, 0, s, params);
        else if (batch) hipLaunchKernelGGL((k_render_persistent<false, false, true>), g, b, 0, s, params);
        else hipLaunchKernelGGL((k_render_persistent<false, false, false>), g, b, 0, s, params);
    }
}
// The resident form (k_render_resident): one 1024-thread block per CU like the WIDE form, for scenes it can take (see
// resident_form_fits); 0 if the kernel cannot be resident with that much LDS.
// LDS of the resident form: the scene's tables and, per wave, `batch` staged primary rays
static size_t resident_lds_bytes(const DScene& sc, uint32_t batch) { return persistent_lds_bytes(sc, 3u, true) + (size_t)(WIDE_BLOCK / 64u) * batch * RESIDENT_STAGE_BYTES; }
uint32_t resident_waves_per_cu(const DScene& sc, uint32_t batch) {
    int nb = 0;
    const size_t lds = resident_lds_bytes(sc, batch);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_render_resident<false>, WIDE_BLOCK, lds) != hipSuccess || nb < 1) return 0u;
    return (uint32_t)nb * (WIDE_BLOCK / 64);
}
void launch_render_resident(const DRenderParams* params, const DRenderParams& host_params, hipStream_t s) {
    const uint32_t block_waves = WIDE_BLOCK / 64;
    dim3 g((host_params.n_sub + block_waves - 1) / block_waves), b(WIDE_BLOCK);
    const size_t lds = resident_lds_bytes(host_params.sc, host_params.resident_batch);
    if (host_params.sc.batch_undecided != 0u) hipLaunchKernelGGL((k_render_resident<true>), g, b, lds, s, params);
    else hipLaunchKernelGGL((k_render_resident<false>), g, b, lds, s, params);
}
void launch_scale(const double* in, uint64_t n, double scale, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, scale, out);
}
void launch_add(const double* in, uint64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_add, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, out);
}
void launch_quantize(const double* in, uint64_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, out);
}
void launch_rays_to_state(const double* org, const double* dir, uint64_t n, const DState& st, hipStream_t s) {
    hipLaunchKernelGGL(k_rays_to_state, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, org, dir, n, st);
}
void launch_user_ids(const uint32_t* rec_user_id, uint32_t n_spheres, uint32_t* ids, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_user_ids, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, rec_user_id, n_spheres, ids, n);
}
void launch_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_math, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, op, a, b, n, out);
}
void launch_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_rng, dim3(1), dim3(64), 0, s, seed, pixel, sample, n, out);
}

}  // namespace rtx
