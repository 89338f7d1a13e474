/*
 * rtx_host.h -- C view of the host-side scene / camera / BVH-builder API
 * (raytracing_amd/host/raytracing.hpp), for hosts that cannot include C++
 * (the Python test and benchmark drivers bind it with ctypes).
 *
 * These entry points run on the CPU only.  They restate, for the drop-in
 * boundary, what the reference host does before it renders:
 *     Surface construction        src/geometry.rs:56-88, src/surface.rs:20-24
 *     BVH::from_slice             src/bvh/mod.rs:76-131, src/bvh/partition.rs
 *     Camera::initialize          src/camera.rs:64-113
 * Citations are into the reference repository.
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include "rtx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PartitionBy / SAHBucketStrategy, src/bvh/mod.rs:13-34: the values of enum rtx_partition */
enum rth_partition { RTH_BISECT = RTX_BISECT, RTH_MIDPOINT = RTX_MIDPOINT, RTH_SAH_EQUAL_SIZE = RTX_SAH_EQUAL_SIZE,
                     RTH_SAH_PER_SURFACE = RTX_SAH_PER_SURFACE };

/* One Surface { geometry, material } as a caller of the reference
 * constructs it (src/surface.rs:15-24); derived plane fields are computed
 * by the builder exactly as Geometry::quadrilateral / triangle do. */
typedef struct rtx_surface {
    uint32_t geom_kind;   /* rtx_geom_kind */
    uint32_t mat_kind;    /* rtx_mat_kind */
    double q[3];          /* sphere: center; planar: q */
    double u[3];          /* planar only */
    double v[3];          /* planar only */
    double radius;        /* sphere only */
    double rgb[3];        /* albedo | emit */
    double param;         /* fuzz_radius | refraction_index | intensity */
} rtx_surface;

/* struct Camera, src/camera.rs:11-26 */
typedef struct rtx_camera_config {
    double aspect_ratio;
    uint32_t image_width;
    uint32_t samples_per_pixel;
    uint32_t max_depth;
    uint32_t _reserved;
    double v_fov;
    double look_from[3];
    double look_at[3];
    double v_up[3];
    double defocus_angle;
    double focus_dist;
    double background[3];
} rtx_camera_config;

typedef struct rth_bvh rth_bvh;

const char* rth_last_error(void);

/* BVH::from_slice(surfaces, &partition_by).  Fails (RTX_ERR_BUILD) where the
 * reference panics or never returns: "No valid splitting plane"
 * (src/bvh/partition.rs:219), a negative sphere radius
 * (src/geometry.rs:56-62), a partition with an empty side. */
int rth_bvh_build(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, rth_bvh** out);
/* The same tree, for every strategy, with the partitioning done on the
 * current HIP device (rtx_bvh_build); the surfaces' boxes and the flattened tables are made on
 * the host.  stats may be NULL. */
int rth_bvh_build_device(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, rth_bvh** out,
                         rtx_build_stats* stats);
/* Fills `out` with pointers into the BVH's tables (valid until rth_bvh_destroy). */
void rth_bvh_desc(const rth_bvh* bvh, rtx_scene_desc* out);
void rth_bvh_destroy(rth_bvh* bvh);

/* Camera::initialize */
int rth_camera_initialize(const rtx_camera_config* cfg, rtx_camera* out);

/* The scene and camera of one of the reference's example programs
 * (examples/<name>.rs: hello_triangle, cover_spheres, cornell_box, quads,
 * simple_light, simple_scene, demo_spheres), as its main() hands them to
 * RenderRunner.  cover_spheres draws its scene at random; scene_seed makes
 * that repeatable.  surfaces may be NULL to ask for the count only; at most
 * `capacity` records are written, *n is the scene's size. */
int rth_example(const char* name, uint64_t scene_seed, rtx_surface* surfaces, uint32_t capacity, uint32_t* n,
                rtx_camera_config* cfg);

/* RenderRunner::run (src/runner.rs:23-41) on the current HIP device:
 * build, render, write a P3 PPM to `ppm_path` (NULL: no image file) and the
 * linear image to out_rgb ([H][W][3], may be NULL). */
int rth_run(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
            uint64_t seed, const char* ppm_path, double* out_rgb);
/* rth_run with the BVH built by rth_bvh_build_device */
int rth_run_device_build(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
                         uint64_t seed, const char* ppm_path, double* out_rgb);

/* rth_run on n_gpus GPUs of this node from this process (RenderRunner with n_gpus set): the samples per pixel are
 * split over the devices and summed with one RCCL reduce (rtx_multi_create / rtx_multi_render of rtx.h).  n_gpus = 1
 * is rth_run.  The reference has no multi-device path; this is what its RenderRunner::run (src/runner.rs:23-41)
 * calls when the host has several GPUs. */
int rth_run_multi(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
                  uint64_t seed, int n_gpus, const char* ppm_path, double* out_rgb);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HOST_H */
