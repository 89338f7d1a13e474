// rtx_kernels.hip -- gfx950 kernels of the wavefront path tracer.
//
//   k_generate      first fill of the path pool (sample_square + get_ray)
//   k_traverse      closest hit over the threaded pre-order BVH  (BVH::hit)
//   k_shade         complete the hit, emitted + scatter, accumulate finished
//                   paths, restart the slot on the next (pixel, sample), and
//                   compact the live slots into the next queue (64-wide
//                   ballot + prefix count, one atomic per wave)
//   k_scale / k_quantize   output stage
//
// Citations "file:line" are into the reference repository.
#include "rtx_device.h"
#include "rtx_kernels.h"

namespace rtx {

constexpr int BLOCK = 256;

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---------------------------------------------------------------- primary rays
// sample_square + get_ray, src/camera.rs:164-182.  `g` is the linear path
// index: pixel = g % (W*H), sample = spp_begin + g / (W*H).
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

__device__ __forceinline__ void store_new_path(const DPool& pool, uint32_t slot, const DCamera& cam, uint64_t seed,
                                               uint32_t spp_begin, unsigned long long g) {
    unsigned long long npix = (unsigned long long)cam.image_width * cam.image_height;
    uint32_t sample_off = (uint32_t)(g / npix);
    uint32_t pixel = (uint32_t)(g - (unsigned long long)sample_off * npix);
    uint32_t sample = spp_begin + sample_off;
    V3 o, d;
    uint32_t draw;
    start_path(cam, seed, pixel, sample, o, d, draw);
    pool.ox[slot] = o.x; pool.oy[slot] = o.y; pool.oz[slot] = o.z;
    pool.dx[slot] = d.x; pool.dy[slot] = d.y; pool.dz[slot] = d.z;
    pool.bx[slot] = 1.0; pool.by[slot] = 1.0; pool.bz[slot] = 1.0;
    pool.pixel[slot] = pixel;
    pool.sample[slot] = sample;
    pool.draw[slot] = draw;
    pool.depth[slot] = 0u;
}

// Starts new paths in the slots the shade kernel has freed (or, on the first
// round, in every slot): slot free_queue[i] gets the next unstarted
// (pixel, sample) pair and is appended to the live queue.  One atomic per
// wave on the path counter and one on the queue length.
__global__ __launch_bounds__(BLOCK) void k_generate(DCamera cam, DPool pool, const uint32_t* free_queue, const uint32_t* n_free_ptr,
                                                    uint32_t* queue_out, uint32_t* count_out, DRenderCtl* ctl, uint64_t seed,
                                                    uint32_t spp_begin) {
    const uint32_t n_free = *n_free_ptr;
    const uint32_t lane = lane_id();
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t rounds = (n_free + stride - 1) / stride;   // wave-uniform bound: every lane reaches the ballots
    const unsigned long long total = ctl->total_paths;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t i = r * stride + blockIdx.x * BLOCK + threadIdx.x;
        bool want = i < n_free;
        unsigned long long wmask = __ballot(want);
        if (!wmask) continue;
        uint32_t leader = __ffsll((long long)wmask) - 1;
        unsigned long long base = 0;
        if (lane == leader) base = atomicAdd(&ctl->next_path, (unsigned long long)__popcll(wmask));
        base = __shfl(base, leader);
        unsigned long long g = base + __popcll(wmask & ((1ull << lane) - 1ull));
        bool start = want && g < total;
        uint32_t slot = 0;
        if (start) {
            slot = free_queue[i];
            store_new_path(pool, slot, cam, seed, spp_begin, g);
        }
        unsigned long long smask = __ballot(start);
        if (smask) {
            uint32_t sl = __ffsll((long long)smask) - 1;
            uint32_t qbase = 0;
            if (lane == sl) qbase = atomicAdd(count_out, (uint32_t)__popcll(smask));
            qbase = __shfl(qbase, sl);
            if (start) queue_out[qbase + __popcll(smask & ((1ull << lane) - 1ull))] = slot;
        }
    }
}
__global__ __launch_bounds__(BLOCK) void k_iota(uint32_t* q, uint32_t n) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) q[i] = i;
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

// ---------------------------------------------------------------- closest hit
// impl Hittable for BVH::hit, src/bvh/mod.rs:134-174.
//
// The reference pops node indices from an explicit stack, pushing the right
// child and then the left child (i+1) of every internal node whose box is
// hit.  Because the children are always visited left first, the node popped
// after a subtree is finished or culled depends only on where that subtree
// ends in the pre-order array.  That successor is stored per node (`miss`),
// so the traversal needs no stack: box missed -> miss; internal hit -> i+1;
// leaf -> test the surface, then miss.  The visit order, the shrinking
// interval and the tie rule (`hit.t > nearest.t` keeps the old hit, :159-167)
// are exactly the reference's.
template <bool COUNT, bool EXACT>
__device__ __forceinline__ void closest_hit(const DScene& sc, V3 o, V3 d, double tmin, double tmax, uint32_t& best_prim,
                                            double& best_t, uint32_t& n_nodes, uint32_t& n_sph, uint32_t& n_pln) {
    best_prim = NONE;
    best_t = tmax;
    const bool filtered = !EXACT && sc.filter_ok && filter_ray_ok(o, d, tmin);
    V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    uint32_t node = 0;
    const uint32_t end = sc.n_nodes;
    while (node < end) {
        const DNode& nd = sc.nodes[node];
        if (COUNT) ++n_nodes;
        bool hit = filtered ? box_hit_filtered(nd, o, d, inv, tmin, tmax) : box_hit_exact(nd, o, d, tmin, tmax);
        if (!hit) {
            node = nd.miss;
            continue;
        }
        uint32_t prim = nd.prim;
        if (prim == NONE) {
            node = node + 1;
            continue;
        }
        double t;
        bool ph;
        if (nd.kind == 0u) {
            if (COUNT) ++n_sph;
            ph = sphere_t(sc.spheres[nd.slot], o, d, tmin, tmax, t);
        } else {
            if (COUNT) ++n_pln;
            double al, be;
            ph = planar_t(sc.planars[nd.slot], nd.kind, o, d, tmin, tmax, t, al, be);
        }
        // `if let Some(nearest) = acc && hit.t > nearest.t { keep } else { replace }`
        if (ph && !(best_prim != NONE && t > best_t)) {
            best_prim = prim;
            best_t = t;
            tmax = t;   // shrunken_ray_t.max = hit.t
        }
        node = nd.miss;
    }
}

template <bool COUNT, bool EXACT>
__global__ __launch_bounds__(BLOCK) void k_traverse(DScene sc, DPool pool, const uint32_t* queue, const uint32_t* count_ptr,
                                                    uint32_t* zero_a, uint32_t* zero_b, double tmin, double tmax, DRenderCtl* ctl) {
    const uint32_t count = *count_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // lengths of the queues the following shade / generate kernels append to
        if (zero_a) *zero_a = 0u;
        if (zero_b) *zero_b = 0u;
        if (ctl) atomicAdd(&ctl->rays, (unsigned long long)count);
    }
    uint32_t c_nodes = 0, c_sph = 0, c_pln = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < count; i += gridDim.x * BLOCK) {
        uint32_t slot = queue ? queue[i] : i;
        V3 o = v3(pool.ox[slot], pool.oy[slot], pool.oz[slot]);
        V3 d = v3(pool.dx[slot], pool.dy[slot], pool.dz[slot]);
        uint32_t prim;
        double t;
        closest_hit<COUNT, EXACT>(sc, o, d, tmin, tmax, prim, t, c_nodes, c_sph, c_pln);
        pool.hit_prim[slot] = prim;
        pool.hit_t[slot] = t;
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
__global__ __launch_bounds__(BLOCK) void k_shade(DScene sc, DCamera cam, DPool pool, const uint32_t* queue_in,
                                                 const uint32_t* count_in, uint32_t* queue_out, uint32_t* count_out,
                                                 uint32_t* free_queue, uint32_t* n_free, uint64_t seed, double* accum) {
    const uint32_t count = *count_in;
    const uint32_t lane = lane_id();
    const uint32_t stride = gridDim.x * BLOCK;
    // the loop bound is wave-uniform so every lane reaches the ballots
    const uint32_t rounds = (count + stride - 1) / stride;
    for (uint32_t r = 0; r < rounds; ++r) {
        uint32_t i = r * stride + blockIdx.x * BLOCK + threadIdx.x;
        bool active = i < count;
        bool alive = false;     // slot goes into the next queue
        uint32_t slot = 0;
        if (active) {
            slot = queue_in[i];
            uint32_t prim = pool.hit_prim[slot];
            V3 beta = v3(pool.bx[slot], pool.by[slot], pool.bz[slot]);
            uint32_t pixel = pool.pixel[slot];
            V3 radiance = v3(0.0, 0.0, 0.0);
            bool add = false;
            if (prim == NONE) {
                radiance = beta * ld3(cam.background);   // camera.rs:207
                add = true;
            } else {
                V3 o = v3(pool.ox[slot], pool.oy[slot], pool.oz[slot]);
                V3 d = v3(pool.dx[slot], pool.dy[slot], pool.dz[slot]);
                double t = pool.hit_t[slot];
                DPrimInfo pi = sc.prims[prim];
                DMaterial m = sc.materials[pi.material];
                // complete the Hit for the winning surface (geometry.rs:232-249, 280-288)
                V3 p = o + d * t;
                V3 outward;
                double alpha = 0.0, beta_uv = 0.0;
                if (pi.kind == 0u) {
                    DSphere s = sc.spheres[pi.slot];
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
                bool front_face = dot(d, outward) < 0.0;           // geometry.rs:135-145
                V3 n = front_face ? outward : -outward;

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
                PathRng rng(seed, pixel, pool.sample[slot], pool.draw[slot]);
                rng.begin_event();
                bool scattered = false;
                V3 nd = v3(0.0, 0.0, 0.0), att = v3(1.0, 1.0, 1.0);
                if (m.kind <= 1u) {
                    // Lambertian and Metal both start with Vector3::random_unit()
                    // (material.rs:62, :86); one shared rejection loop for the wave
                    V3 ru = random_unit(rng);
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
                uint32_t depth = pool.depth[slot] + 1u;
                if (scattered && depth < cam.max_depth) {
                    beta = beta * att;
                    pool.ox[slot] = p.x; pool.oy[slot] = p.y; pool.oz[slot] = p.z;
                    pool.dx[slot] = nd.x; pool.dy[slot] = nd.y; pool.dz[slot] = nd.z;
                    pool.bx[slot] = beta.x; pool.by[slot] = beta.y; pool.bz[slot] = beta.z;
                    pool.draw[slot] = rng.draw;
                    pool.depth[slot] = depth;
                    alive = true;
                }
            }
            if (add) {
                double* px = accum + 3 * (size_t)pixel;
                if (radiance.x != 0.0) unsafeAtomicAdd(px + 0, radiance.x);
                if (radiance.y != 0.0) unsafeAtomicAdd(px + 1, radiance.y);
                if (radiance.z != 0.0) unsafeAtomicAdd(px + 2, radiance.z);
            }
        }
        // compaction: live slots into the next queue, finished slots into the
        // free queue (64-wide ballot + prefix popcount, one atomic per wave and queue)
        const bool finished = active && !alive;
        unsigned long long amask = __ballot(alive);
        unsigned long long fmask = __ballot(finished);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (amask) {
            uint32_t leader = __ffsll((long long)amask) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(count_out, (uint32_t)__popcll(amask));
            base = __shfl(base, leader);
            if (alive) queue_out[base + __popcll(amask & below)] = slot;
        }
        if (fmask) {
            uint32_t leader = __ffsll((long long)fmask) - 1;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(n_free, (uint32_t)__popcll(fmask));
            base = __shfl(base, leader);
            if (finished) free_queue[base + __popcll(fmask & below)] = slot;
        }
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

// transposes rays [n][3] (API layout) into the pool's per-component arrays
__global__ __launch_bounds__(BLOCK) void k_rays_to_pool(const double* org, const double* dir, uint64_t n, DPool pool) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        pool.ox[i] = org[3 * i + 0]; pool.oy[i] = org[3 * i + 1]; pool.oz[i] = org[3 * i + 2];
        pool.dx[i] = dir[3 * i + 0]; pool.dy[i] = dir[3 * i + 1]; pool.dz[i] = dir[3 * i + 2];
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

void launch_generate(const DCamera& cam, const DPool& pool, const uint32_t* free_queue, const uint32_t* n_free, uint32_t* queue_out,
                     uint32_t* count_out, uint64_t max_count, DRenderCtl* ctl, uint64_t seed, uint32_t spp_begin,
                     uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_generate, dim3(grid_for(max_count, max_blocks)), dim3(BLOCK), 0, s, cam, pool, free_queue, n_free, queue_out,
                       count_out, ctl, seed, spp_begin);
}
void launch_iota(uint32_t* q, uint32_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, q, n);
}
void launch_primary_rays(const DCamera& cam, uint64_t seed, uint32_t sample, uint32_t npix, double* org, double* dir,
                         uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_primary_rays, dim3(grid_for(npix, max_blocks)), dim3(BLOCK), 0, s, cam, seed, sample, npix, org, dir);
}
void launch_traverse(const DScene& sc, const DPool& pool, const uint32_t* queue, const uint32_t* count_ptr, uint32_t* zero_a,
                     uint32_t* zero_b, uint64_t max_count, double tmin, double tmax, DRenderCtl* ctl, bool count_work, bool exact,
                     uint32_t max_blocks, hipStream_t s) {
    dim3 g(grid_for(max_count, max_blocks)), b(BLOCK);
    if (count_work) {
        if (exact) hipLaunchKernelGGL((k_traverse<true, true>), g, b, 0, s, sc, pool, queue, count_ptr, zero_a, zero_b, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<true, false>), g, b, 0, s, sc, pool, queue, count_ptr, zero_a, zero_b, tmin, tmax, ctl);
    } else {
        if (exact) hipLaunchKernelGGL((k_traverse<false, true>), g, b, 0, s, sc, pool, queue, count_ptr, zero_a, zero_b, tmin, tmax, ctl);
        else hipLaunchKernelGGL((k_traverse<false, false>), g, b, 0, s, sc, pool, queue, count_ptr, zero_a, zero_b, tmin, tmax, ctl);
    }
}
void launch_shade(const DScene& sc, const DCamera& cam, const DPool& pool, const uint32_t* queue_in, const uint32_t* count_in,
                  uint32_t* queue_out, uint32_t* count_out, uint32_t* free_queue, uint32_t* n_free, uint64_t max_count,
                  uint64_t seed, double* accum, uint32_t max_blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_shade, dim3(grid_for(max_count, max_blocks)), dim3(BLOCK), 0, s, sc, cam, pool, queue_in, count_in,
                       queue_out, count_out, free_queue, n_free, seed, accum);
}
void launch_scale(const double* in, uint64_t n, double scale, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, scale, out);
}
void launch_quantize(const double* in, uint64_t n, uint8_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_quantize, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, in, n, out);
}
void launch_rays_to_pool(const double* org, const double* dir, uint64_t n, const DPool& pool, hipStream_t s) {
    hipLaunchKernelGGL(k_rays_to_pool, dim3(grid_for(n, 2048)), dim3(BLOCK), 0, s, org, dir, n, pool);
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
