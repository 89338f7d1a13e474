// rtx_kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
// One round of the render loop is three launches on one stream:
//   k_generate   start new paths (sample_square + get_ray) behind the
//                survivors of the previous round, filling the state buffer
//   k_traverse   closest hit of every live ray over the threaded pre-order
//                BVH (BVH::hit); persistent waves that refill idle lanes
//   k_shade      complete the hit, emitted + scatter, accumulate finished
//                paths, and write the surviving paths densely into the other
//                state buffer (64-wide ballot + prefix count, one atomic per wave)
// k_scale / k_quantize are the output stage.
//
// Citations "file:line" are into the reference repository.
#include "rtx_device.h"
#include "rtx_kernels.h"

namespace rtx {

constexpr int BLOCK = 256;

#ifndef RTX_SPHERE_PREFETCH
#define RTX_SPHERE_PREFETCH 0
#endif
#ifndef RTX_TRAVERSE_WAVES
#define RTX_TRAVERSE_WAVES 1
#endif
#ifndef RTX_SHADE_WAVES
#define RTX_SHADE_WAVES 1
#endif
#ifndef RTX_REFILL_AT
#define RTX_REFILL_AT 32
#endif
#ifndef RTX_PEND_AT
#define RTX_PEND_AT 16
#endif

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint32_t rank_below(unsigned long long mask, uint32_t lane) {
    return (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
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

// Path g of a render call is sample (spp_begin + g / npix) of pixel g % npix,
// so consecutive lanes start neighbouring pixels of one sample index.
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
    pixel = (uint32_t)(g - q * npix);
}

// Appends new paths behind the survivors in state buffer `st` until the
// buffer is full or every path of the call has been started, and publishes
// the round's ray count.  Thread 0 also resets the counters that the
// following traverse and shade kernels bump.
__global__ __launch_bounds__(BLOCK) void k_generate(DCamera cam, DState st, uint32_t capacity, DRenderCtl* ctl, uint32_t parity,
                                                    uint64_t seed, uint32_t spp_begin) {
    const uint32_t n_surv = ctl->count_surv[parity];
    const unsigned long long next = ctl->next_path[parity];
    const unsigned long long left = ctl->total_paths - next;
    const uint32_t room = capacity - n_surv;
    const uint32_t n_new = left < room ? (uint32_t)left : room;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctl->next_path[parity ^ 1u] = next + n_new;
        ctl->count_live = n_surv + n_new;
        ctl->rays += n_surv + n_new;
        ctl->next_ray = 0u;
        ctl->count_surv[parity ^ 1u] = 0u;
    }
    const unsigned long long npix = (unsigned long long)cam.image_width * cam.image_height;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_new; i += gridDim.x * BLOCK) {
        uint32_t pixel, sample_off;
        split_path_index(next + i, npix, pixel, sample_off);
        const uint32_t sample = spp_begin + sample_off;
        V3 o, d;
        uint32_t draw;
        start_path(cam, seed, pixel, sample, o, d, draw);
        const uint32_t k = n_surv + i;
        st.ox[k] = o.x; st.oy[k] = o.y; st.oz[k] = o.z;
        st.dx[k] = d.x; st.dy[k] = d.y; st.dz[k] = d.z;
        st.bx[k] = 1.0; st.by[k] = 1.0; st.bz[k] = 1.0;
        st.pixel[k] = pixel;
        st.sample[k] = sample;
        st.draw[k] = draw;
        st.depth[k] = 0u;
    }
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
__device__ __noinline__ bool tier2_box(double2 q0, double2 q1, double2 q2, V3 o, V3 d, double tmin, double tmax, bool use_filter) {
    const double bmin[3] = {q0.x, q0.y, q1.x};
    const double bmax[3] = {q1.y, q2.x, q2.y};
    if (use_filter) {
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
__global__ __launch_bounds__(BLOCK, RTX_TRAVERSE_WAVES) void k_traverse(DScene sc, DState rays, DHits hits, const uint32_t* count_ptr,
                                                                        double tmin, double tmax0, DRenderCtl* ctl) {
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    const uint32_t tile_stride = gridDim.x * BLOCK;
    uint32_t tile_base = (blockIdx.x * BLOCK + threadIdx.x) & ~63u;   // first ray of this wave's current tile
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
        // ---- refill: when every lane has retired, the wave takes its next tile of 64 rays.
        // Tiles are dealt round-robin over the waves of the grid; a shared work
        // counter would serialise on one L2 word (one atomic per 64 rays).
        if (__ballot(idx != NONE) == 0ull) {
            if (tile_base >= count) break;
            const uint32_t mine = tile_base + lane;
            tile_base += tile_stride;
            if (mine < count) {
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
        double2 sph_a = make_double2(0.0, 0.0), sph_b = make_double2(0.0, 0.0);
        while (__ballot(searching) != 0ull) {
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
#if RTX_SPHERE_PREFETCH
                    if (link.z == 0u) {   // start fetching the sphere while other lanes keep walking
                        const double2* sp = reinterpret_cast<const double2*>(sc.spheres + link.w);
                        sph_a = sp[0];
                        sph_b = sp[1];
                    }
#endif
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
#if !RTX_SPHERE_PREFETCH
                const double2* spp = reinterpret_cast<const double2*>(sc.spheres + leaf_slot);
                sph_a = spp[0];
                sph_b = spp[1];
#endif
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
template <bool COUNT, bool LDS_NODES>
__global__ __launch_bounds__(BLOCK) void k_traverse_f32(DScene sc, DState rays, DHits hits, const uint32_t* count_ptr,
                                                        double tmin, double tmax0, DRenderCtl* ctl) {
    extern __shared__ uint4 lds_nodes[];   // LDS_NODES: 2 x uint4 per node
    const uint32_t end = sc.n_nodes;
    if (LDS_NODES) {
        const uint4* src = reinterpret_cast<const uint4*>(sc.nodes32);
        for (uint32_t i = threadIdx.x; i < 2u * end; i += BLOCK) lds_nodes[i] = src[i];
        __syncthreads();
    }
    const uint32_t count = *count_ptr;
    const uint32_t lane = lane_id();
    // Tiles of 64 consecutive rays are dealt round-robin over the waves of the
    // grid (neighbouring waves stream neighbouring memory).  A wave's own
    // tiles form its private work list: position p of the list is ray
    // (p & 63) of its (p >> 6)-th tile.  Idle lanes take the next positions
    // of that list, so there is no shared work counter.
    constexpr uint32_t REFILL_AT = RTX_REFILL_AT;   // refill once this many lanes are idle
    constexpr uint32_t PEND_AT = RTX_PEND_AT;       // test surfaces once this many lanes stand on a leaf
    const uint32_t n_waves = gridDim.x * (BLOCK / 64);
    const uint32_t wave_global = (blockIdx.x * BLOCK + threadIdx.x) >> 6;
    const uint32_t n_tiles = (count + 63u) >> 6;
    const uint32_t my_tiles = wave_global < n_tiles ? (n_tiles - wave_global + n_waves - 1u) / n_waves : 0u;
    const uint32_t p_end = my_tiles * 64u;
    uint32_t next_p = 0;
    const bool scene_filter = sc.filter_ok != 0u;

    uint32_t idx = NONE;   // ray of this lane, NONE = idle
    V3 o = v3(0, 0, 0), d = v3(1, 1, 1);
    double tmax = tmax0, best_t = tmax0;
    uint32_t best_leaf = NONE, node = 0;
    bool use_f32 = false, use_filter = false;
    RayF32 rf;
    rayf32_init(rf, o, d, tmin, tmax0);
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0;
    bool pending = false;    // standing on a leaf whose box was hit, surface not tested yet
    uint32_t leaf = NONE;

    for (;;) {
        // ---- refill idle lanes from the wave's work list once enough of them are idle
        const unsigned long long idle = __ballot(idx == NONE);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle == 64u && next_p >= p_end) break;
        if (next_p < p_end && (n_idle >= REFILL_AT || n_idle == 64u)) {
            const uint32_t p = next_p + rank_below(idle, lane);
            next_p += n_idle;
            const uint32_t mine = ((wave_global + (p >> 6) * n_waves) << 6) | (p & 63u);
            if (idx == NONE && p < p_end && mine < count) {
                idx = mine;
                o = v3(rays.ox[mine], rays.oy[mine], rays.oz[mine]);
                d = v3(rays.dx[mine], rays.dy[mine], rays.dz[mine]);
                tmax = tmax0;
                best_t = tmax0;
                best_leaf = NONE;
                node = 0;
                use_filter = scene_filter && filter_ray_ok(o, d, tmin, tmax0);
                const V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
                use_f32 = rayf32_init(rf, o, inv, tmin, tmax0) && use_filter;
            }
        }

        // ---- phase 1: walk to the next leaf whose box is hit
        // (a lane that found its leaf waits; the loop also ends as soon as
        // PEND_AT lanes are waiting, so a few long searches do not stall them)
        uint32_t n_pending = 0;   // wave-uniform
        bool searching = idx != NONE && node < end;
        while (searching) {       // lanes leave the loop one by one; the wave stays until all have left
            uint4 n0, n1;
            if (LDS_NODES) {
                n0 = lds_nodes[2u * node];
                n1 = lds_nodes[2u * node + 1u];
            } else {
                const uint4* np = reinterpret_cast<const uint4*>(sc.nodes32 + node);
                n0 = np[0];
                n1 = np[1];
            }
            if (COUNT) ++c_nodes;
            const float planes[6] = {__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z),
                                     __uint_as_float(n0.w), __uint_as_float(n1.x), __uint_as_float(n1.y)};
            int r = box_hit_f32(planes, rf);
            if (!use_f32) r = -1;
            if (r < 0) {   // undecided in f32: the exact box, f64 filter, then the division form
                const double2* bp = reinterpret_cast<const double2*>(sc.boxes64 + node);
                const double2 q0 = bp[0], q1 = bp[1], q2 = bp[2];
                r = tier2_box(q0, q1, q2, o, d, tmin, tmax, use_filter) ? 1 : 0;
            }
            const bool hit = r > 0;
            const bool is_leaf = n1.w != NONE;
            const bool found = hit && is_leaf;
            node = (hit && !is_leaf) ? node + 1u : n1.z;   // a leaf's successor is its miss link as well
            leaf = found ? n1.w : leaf;
            pending = found;
            n_pending += (uint32_t)__popcll(__ballot(found));
            searching = !found && node < end;
            if (n_pending >= PEND_AT) break;   // wave-uniform
        }

        // ---- phase 2: surface tests of the lanes standing on a leaf
        if (pending) {
            const uint32_t kind = leaf >> 30, slot = leaf & 0x3FFFFFFFu;
            double t;
            bool ph;
            if (kind == 0u) {
                if (COUNT) ++c_sph;
                const double2* spp = reinterpret_cast<const double2*>(sc.spheres + slot);
                const double2 sa = spp[0], sb = spp[1];
                DSphere sp;
                sp.c[0] = sa.x; sp.c[1] = sa.y; sp.c[2] = sb.x; sp.r = sb.y;
                ph = sphere_t(sp, o, d, tmin, tmax, t);
            } else {
                if (COUNT) ++c_pln;
                double al, be;
                ph = planar_t(sc.planars[slot], kind, o, d, tmin, tmax, t, al, be);
            }
            // `if let Some(nearest) = acc && hit.t > nearest.t { keep } else { replace }`
            if (ph && !(best_leaf != NONE && t > best_t)) {
                best_leaf = leaf;
                best_t = t;
                tmax = t;   // shrunken_ray_t.max = hit.t
                rayf32_set_tmax(rf, t);
            }
            pending = false;
        }

        // ---- retire rays that have run out of tree
        if (idx != NONE && node >= end) {
            uint32_t prim = NONE;
            if (best_leaf != NONE) {
                const uint32_t slot = best_leaf & 0x3FFFFFFFu;
                prim = (best_leaf >> 30) == 0u ? sc.sphere_prim[slot] : sc.planar_prim[slot];
            }
            hits.prim[idx] = prim;
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

// ---------------------------------------------------------------- shade
// Schlick, src/material.rs:134-137; powi(5) = x * ((x*x)*(x*x))
__device__ __forceinline__ double reflectance(double cos_theta, double refraction_index) {
    double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
    r0 = r0 * r0;
    double x = 1.0 - cos_theta;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1.0 - r0) * x5;
}

// One step of ray_color (src/camera.rs:184-208) in its iterative form:
//   miss            -> add beta * background, path ends
//   hit             -> add beta * emitted; scatter ? (beta *= attenuation, new ray) : path ends
//   depth exhausted -> path ends contributing nothing more
// The reference nests `emitted + ray_color(..) * attenuation`; with emission
// only on surfaces that do not scatter this is the same product of
// attenuations, associated from the other end.
__global__ __launch_bounds__(BLOCK, RTX_SHADE_WAVES) void k_shade(DScene sc, DCamera cam, DState cur, DHits hits, const uint32_t* count_in,
                                                 DState nxt, uint32_t* count_out, uint64_t seed, double* accum) {
    __shared__ uint32_t wave_total[BLOCK / 64];
    __shared__ uint32_t block_base;
    const uint32_t count = *count_in;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t stride = gridDim.x * BLOCK;
    // the loop bound is block-uniform so every thread reaches the barriers
    const uint32_t rounds = (count + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = r * stride + blockIdx.x * BLOCK + threadIdx.x;
        bool alive = false;
        V3 p = v3(0, 0, 0), nd = v3(0, 0, 0), beta = v3(0, 0, 0);
        uint32_t pixel = 0, sample = 0, draw = 0, depth = 0;
        if (i < count) {
            const uint32_t prim = hits.prim[i];
            beta = v3(cur.bx[i], cur.by[i], cur.bz[i]);
            pixel = cur.pixel[i];
            V3 radiance = v3(0.0, 0.0, 0.0);
            bool add = false;
            if (prim == NONE) {
                radiance = beta * ld3(cam.background);   // camera.rs:207
                add = true;
            } else {
                const V3 o = v3(cur.ox[i], cur.oy[i], cur.oz[i]);
                const V3 d = v3(cur.dx[i], cur.dy[i], cur.dz[i]);
                const double t = hits.t[i];
                sample = cur.sample[i];
                const DPrimInfo pi = sc.prims[prim];
                const DMaterial m = sc.materials[pi.material];
                // complete the Hit for the winning surface (geometry.rs:232-249, 280-288)
                p = o + d * t;
                V3 outward;
                double alpha = 0.0, beta_uv = 0.0;
                if (pi.kind == 0u) {
                    const DSphere s = sc.spheres[pi.slot];
                    outward = (p - v3(s.c[0], s.c[1], s.c[2])) / s.r;
                    if (m.kind == 4u) {   // u,v only feed UVGradient (geometry.rs:236-240)
                        const double PI = 3.14159265358979323846264338327950288;
                        double theta = acos(-p.y);
                        double phi = atan2(-p.z, p.x) + PI;
                        alpha = phi / (2.0 * PI);
                        beta_uv = theta / PI;
                    }
                } else {
                    const DPlanar& g = sc.planars[pi.slot];
                    outward = ld3(g.norm);
                    if (m.kind == 4u) {
                        V3 qp = p - ld3(g.q);
                        V3 w = ld3(g.w);
                        alpha = dot(w, cross(qp, ld3(g.v)));
                        beta_uv = dot(w, cross(ld3(g.u), qp));
                    }
                }
                const bool front_face = dot(d, outward) < 0.0;           // geometry.rs:135-145
                const V3 n = front_face ? outward : -outward;

                // Material::emitted, material.rs:35-54
                if (m.kind == 3u) {
                    radiance = beta * ld3(m.rgb);
                    add = true;
                } else if (m.kind == 4u) {
                    double a0 = 0.0 - alpha, a1 = 1.0 - alpha, b0 = 0.0 - beta_uv, b1 = 1.0 - beta_uv;
                    V3 e = v3(1.0 - sqrt(a0 * a0 + b0 * b0), 1.0 - sqrt(a1 * a1 + b0 * b0), 1.0 - sqrt(a0 * a0 + b1 * b1)) * m.param;
                    radiance = beta * e;
                    add = true;
                }

                // Material::scatter, material.rs:20-33
                PathRng rng(seed, pixel, sample, cur.draw[i]);
                rng.begin_event();
                bool scattered = false;
                V3 att = v3(1.0, 1.0, 1.0);
                if (m.kind <= 1u) {
                    // Lambertian and Metal both start with Vector3::random_unit()
                    // (material.rs:62, :86); one shared rejection loop for the wave
                    const V3 ru = random_unit(rng);
                    att = ld3(m.rgb);
                    if (m.kind == 0u) {        // lambertian, material.rs:61-73
                        V3 direction = n + ru;
                        if (is_near_zero(direction)) direction = n;
                        nd = direction;
                        scattered = true;
                    } else {                   // metal, material.rs:84-96
                        V3 reflected = reflect(d, n);
                        V3 fuzz = ru * m.param;
                        V3 fuzzed = to_unit(reflected) + fuzz;
                        nd = fuzzed;
                        scattered = dot(fuzzed, n) > 0.0;
                    }
                } else if (m.kind == 2u) {     // dielectric, material.rs:108-130
                    V3 r_in = to_unit(d);
                    double eta = front_face ? 1.0 / m.param : m.param;
                    double cos_theta = clampd(dot(-r_in, n), -1.0, 1.0);
                    double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
                    bool refl = eta * sin_theta > 1.0;
                    if (!refl) refl = reflectance(cos_theta, eta) > rng.next();   // no draw on total internal reflection
                    nd = refl ? reflect(r_in, n) : refract(r_in, n, eta);
                    scattered = true;
                }
                depth = cur.depth[i] + 1u;
                if (scattered && depth < cam.max_depth) {
                    beta = beta * att;
                    draw = rng.draw;
                    alive = true;
                }
            }
#ifdef RTX_EXPERIMENT_NO_ATOMICS   // timing experiment only
            if (add && radiance.x == 123.456) {
#else
            if (add) {
#endif
                double* px = accum + 3 * (size_t)pixel;
                if (radiance.x != 0.0) unsafeAtomicAdd(px + 0, radiance.x);
                if (radiance.y != 0.0) unsafeAtomicAdd(px + 1, radiance.y);
                if (radiance.z != 0.0) unsafeAtomicAdd(px + 2, radiance.z);
            }
        }
        // compaction: surviving paths go densely into the other state buffer.
        // 64-wide ballot + prefix popcount inside a wave, wave totals summed
        // through LDS, and ONE atomic per block and round: a returning atomic
        // on a single L2 word sustains under 100 operations per microsecond,
        // which one atomic per wave would exceed.
        const unsigned long long amask = __ballot(alive);
        if (lane == 0) wave_total[wave] = (uint32_t)__popcll(amask);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; ++w) {
                const uint32_t c = wave_total[w];
                wave_total[w] = total;   // exclusive prefix
                total += c;
            }
            block_base = total ? atomicAdd(count_out, total) : 0u;
        }
        __syncthreads();
        {
            const uint32_t base = block_base + wave_total[wave];
            if (alive) {
                const uint32_t k = base + rank_below(amask, lane);
                nxt.ox[k] = p.x; nxt.oy[k] = p.y; nxt.oz[k] = p.z;
                nxt.dx[k] = nd.x; nxt.dy[k] = nd.y; nxt.dz[k] = nd.z;
                nxt.bx[k] = beta.x; nxt.by[k] = beta.y; nxt.bz[k] = beta.z;
                nxt.pixel[k] = pixel;
                nxt.sample[k] = sample;
                nxt.draw[k] = draw;
                nxt.depth[k] = depth;
            }
        }
        __syncthreads();   // wave_total / block_base are reused by the next round
    }
}

// ---------------------------------------------------------------- output stage
__global__ __launch_bounds__(BLOCK) void k_scale(const double* in, uint64_t n, double scale, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) out[i] = in[i] * scale;
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
__global__ __launch_bounds__(BLOCK) void k_user_ids(const DPrimInfo* prims, uint32_t* ids, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        uint32_t p = ids[i];
        if (p != NONE) ids[i] = prims[p].user_id;
    }
}

// ---------------------------------------------------------------- self tests
__global__ void k_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if (op == 0) out[i] = a[i] / b[i];
        else if (op == 1) out[i] = sqrt(a[i]);
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

void launch_generate(const DCamera& cam, const DState& st, uint32_t capacity, DRenderCtl* ctl, uint32_t parity, uint64_t seed,
                     uint32_t spp_begin, uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(capacity, max_blocks)), dim3(BLOCK), 0, s, cam, st, capacity, ctl, parity, seed, spp_begin);
}
void launch_primary_rays(const DCamera& cam, uint64_t seed, uint32_t sample, uint32_t npix, double* org, double* dir,
                         uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_primary_rays, dim3(grid_for(npix, max_blocks)), dim3(BLOCK), 0, s, cam, seed, sample, npix, org, dir);
}
void launch_traverse(const DScene& sc, const DState& rays, const DHits& hits, const uint32_t* count_ptr,
                     uint64_t max_count, double tmin, double tmax, DRenderCtl* ctl, bool count_work, bool exact,
                     uint32_t max_blocks, hipStream_t s) {
    dim3 g(grid_for(max_count, max_blocks)), b(BLOCK);
    if (count_work) {
        if (exact) hipLaunchKernelGGL((k_traverse<true, true>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<true, false>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
    } else {
        if (exact) hipLaunchKernelGGL((k_traverse<false, true>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<false, false>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
    }
}
void launch_traverse_f32(const DScene& sc, const DState& rays, const DHits& hits, const uint32_t* count_ptr, uint64_t max_count,
                         double tmin, double tmax, DRenderCtl* ctl, bool count_work, bool lds_nodes, uint32_t max_blocks,
                         hipStream_t s) {
    dim3 g(grid_for(max_count, max_blocks)), b(BLOCK);
    const size_t lds = lds_nodes ? (size_t)sc.n_nodes * sizeof(DNode32) : 0;
    if (lds_nodes) {
        if (count_work) hipLaunchKernelGGL((k_traverse_f32<true, true>), g, b, lds, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse_f32<false, true>), g, b, lds, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
    } else {
        if (count_work) hipLaunchKernelGGL((k_traverse_f32<true, false>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse_f32<false, false>), g, b, 0, s, sc, rays, hits, count_ptr, tmin, tmax, ctl);
    }
}
void launch_shade(const DScene& sc, const DCamera& cam, const DState& cur, const DHits& hits, const uint32_t* count_in,
                  const DState& nxt, uint32_t* count_out, uint64_t max_count, uint64_t seed, double* accum, uint32_t max_blocks,
                  hipStream_t s) {
    hipLaunchKernelGGL(k_shade, dim3(grid_for(max_count, max_blocks)), dim3(BLOCK), 0, s, sc, cam, cur, hits, count_in, nxt, count_out,
                       seed, accum);
}
void launch_scale(const double* in, uint64_t n, double scale, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, scale, out);
}
void launch_quantize(const double* in, uint64_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, out);
}
void launch_rays_to_state(const double* org, const double* dir, uint64_t n, const DState& st, hipStream_t s) {
    hipLaunchKernelGGL(k_rays_to_state, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, org, dir, n, st);
}
void launch_user_ids(const DPrimInfo* prims, uint32_t* ids, uint64_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_user_ids, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, prims, ids, n);
}
void launch_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_math, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, op, a, b, n, out);
}
void launch_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_rng, dim3(1), dim3(64), 0, s, seed, pixel, sample, n, out);
}

}  // namespace rtx
