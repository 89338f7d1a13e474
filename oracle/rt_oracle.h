/*
 * rt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the path
 * tracer in seridescent/raytracing (Rust).  It exists to check the HIP path
 * and to serve as the timed CPU baseline.  Nothing in the shipped product
 * (raytracing_amd/, include/) may include, link or call it.
 *
 * Pinning: the reference cannot be compiled in this environment (no Rust
 * toolchain), so the restatement is pinned by the reference's own fixtures:
 *   - the five BVH-builder unit tests in src/bvh/mod.rs:191-479 (restated in
 *     tests/test_builder_reference_cases.py), and
 *   - images/hello_triangle_10k.png and images/cornell_box_20k.png (copied as
 *     data into tests/golden/), compared with converged renders.
 * Ray/box, ray/primitive, traversal and scatter arithmetic have no reference
 * test: for those the oracle is "parity unpinned" beyond the converged image.
 *
 * All citations "file:line" are into the reference repository.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* geometry kinds, src/geometry.rs:25-47 */
enum { ORC_SPHERE = 0, ORC_QUAD = 1, ORC_TRIANGLE = 2 };
/* material kinds, src/material.rs:9-17 */
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2, ORC_DIFFUSE_LIGHT = 3, ORC_UV_GRADIENT = 4 };
/* partition strategies, src/bvh/mod.rs:13-34 */
enum { ORC_BISECT = 0, ORC_MIDPOINT = 1, ORC_SAH_EQUAL_SIZE = 2, ORC_SAH_PER_SURFACE = 3 };

/* One Surface{geometry, material} as the caller constructs it
 * (src/surface.rs:15-18); derived fields (norm, d, w) are computed inside. */
typedef struct orc_surface {
    uint32_t geom_kind;
    uint32_t mat_kind;
    double q[3];      /* sphere: center; planar: q */
    double u[3];      /* planar only */
    double v[3];      /* planar only */
    double radius;    /* sphere only */
    double rgb[3];    /* albedo (Lambertian, Metal) | emit (DiffuseLight) */
    double param;     /* fuzz_radius | refraction_index | intensity */
} orc_surface;

/* Camera as the caller fills it, src/camera.rs:11-26 */
typedef struct orc_camera {
    double aspect_ratio;
    uint32_t image_width;
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    uint32_t _pad;
    double v_fov;
    double look_from[3];
    double look_at[3];
    double v_up[3];
    double defocus_angle;
    double focus_dist;
    double background[3];
} orc_camera;

/* InitializedCamera, src/camera.rs:28-43 */
typedef struct orc_icamera {
    uint32_t image_width;
    uint32_t image_height;
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    double defocus_angle;
    double background[3];
    double pixel_samples_scale;
    double center[3];
    double pixel00_loc[3];
    double pixel_du[3];
    double pixel_dv[3];
    double defocus_disk_u[3];
    double defocus_disk_v[3];
} orc_icamera;

typedef struct orc_counters {
    uint64_t rays;          /* world.hit() calls */
    uint64_t paths;         /* primary samples */
    uint64_t node_visits;   /* nodes popped == box tests */
    uint64_t sphere_tests;
    uint64_t planar_tests;
} orc_counters;

typedef struct orc_scene orc_scene;

const char* orc_last_error(void);

/* BVH::from_slice (src/bvh/mod.rs:76-131).  Returns NULL on failure
 * ("No valid splitting plane", negative radius, degenerate midpoint split). */
orc_scene* orc_scene_create(const orc_surface* surfaces, uint32_t n, int strategy, uint32_t buckets);
void orc_scene_destroy(orc_scene*);

uint32_t orc_scene_n_nodes(const orc_scene*);
/* Pre-order node array.  bmin/bmax: [3][n_nodes] (x row, y row, z row);
 * right[i] = right child index (0xFFFFFFFF for a leaf);
 * leaf_surface[i] = index of the leaf's surface in the caller's array
 * (0xFFFFFFFF for an internal node). */
void orc_scene_nodes(const orc_scene*, double* bmin, double* bmax, uint32_t* right, uint32_t* leaf_surface);

/* Surface::bounding_box for one surface, out = {min xyz, max xyz} */
int orc_surface_bbox(const orc_surface* s, double out[6]);
/* sah::surface_area_heuristic (src/bvh/partition.rs:92-106) on boxes given as {min xyz,max xyz} */
double orc_sah_cost(const double left[6], uint64_t n_left, const double right[6], uint64_t n_right, const double parent[6]);

/* AABB::hit (src/aabb.rs:55-76) on one box/ray */
int orc_aabb_hit(const double box[6], const double org[3], const double dir[3], double tmin, double tmax);
/* Geometry::hit for one surface; returns 1 on hit and fills
 * out = {t, p.x,p.y,p.z, alpha, beta, front_face, n.x,n.y,n.z} */
int orc_surface_hit(const orc_surface* s, const double org[3], const double dir[3], double tmin, double tmax, double out[10]);

/* BVH::hit (src/bvh/mod.rs:134-174) for n rays; org/dir are [n][3].
 * out_id = caller surface index or 0xFFFFFFFF; out_t = hit.t (unspecified on miss). */
void orc_trace_closest(const orc_scene*, const double* org, const double* dir, uint64_t n,
                       double tmin, double tmax, uint32_t* out_id, double* out_t, orc_counters* ctr);
/* impl Hittable for &[Surface] (src/surface.rs:40-59) in caller order */
void orc_trace_bruteforce(const orc_scene*, const double* org, const double* dir, uint64_t n,
                          double tmin, double tmax, uint32_t* out_id, double* out_t);

/* Camera::initialize (src/camera.rs:64-113) */
void orc_camera_initialize(const orc_camera* cam, orc_icamera* out);

/* Primary rays exactly as the renderer generates them for sample index
 * `sample` of every pixel (sample_square + get_ray, src/camera.rs:164-182)
 * with the seeded counter RNG.  org/dir are [W*H][3]. */
void orc_primary_rays(const orc_icamera* cam, uint64_t seed, uint32_t sample, double* org, double* dir);

/* The sample loop of InitializedCamera::render (src/camera.rs:133-151)
 * for sample indices [spp_begin, spp_begin+spp_count): out_sum[W*H][3] gets
 * the UNSCALED sum of ray_color over those samples (added in sample order).
 * n_threads <= 0 uses hardware_concurrency().  iterative = 0 evaluates
 * ray_color recursively as written (src/camera.rs:184-208); iterative = 1
 * evaluates the algebraically identical loop form the device uses. */
void orc_render(const orc_scene*, const orc_icamera* cam, uint64_t seed,
                uint32_t spp_begin, uint32_t spp_count, double* out_sum,
                int n_threads, int iterative, orc_counters* ctr);

/* ppm_pixel (src/camera.rs:248-266) on n linear RGB triples -> n*3 bytes */
void orc_ppm_quantize(const double* rgb, uint64_t n_pixels, uint8_t* out);

/* Philox4x32 block with the given number of rounds; the path streams (the seeded counter RNG that replaces rand's
 * ThreadRng) use orc_philox_rounds() of them */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
int orc_philox_rounds(void);
/* first n uniform doubles of the stream (seed, pixel, sample) */
void orc_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out);

/* one Material::scatter + emitted evaluation with the stream (seed,pixel,sample)
 * starting at draw index `draw`; hit10 as produced by orc_surface_hit.
 * out = {scattered?, o.xyz, d.xyz, att.rgb, emitted.rgb, draws_consumed} (14 doubles) */
void orc_scatter(const orc_surface* s, const double org[3], const double dir[3], const double hit10[10],
                 uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw, double out[14]);

#ifdef __cplusplus
}
#endif
#endif
