/*
 * rtx.h -- C ABI of the MI355X render backend.
 *
 * This is the drop-in boundary for the per-pixel sample/bounce loop of
 * seridescent/raytracing.  The reference has no FFI; the path sits behind
 *     trait Hittable::hit              (src/surface.rs:9-12)
 *     InitializedCamera::render        (src/camera.rs:128-162)
 *     RenderRunner::run                (src/runner.rs:23-41)
 * A per-ray Hittable call is the wrong granularity for a GPU, so the ABI
 * replaces the two batch operations those calls add up to:
 *     trace_closest(rays)  -> (surface id, t)      == BVH::hit over a batch
 *     render(camera, bvh)  -> linear RGB image     == the sample loop
 * All types are POD, all pointers are plain, all functions return an int
 * status (0 = OK, < 0 = error, message via rtx_last_error()).  Nothing
 * throws or aborts across this boundary.  Citations are into the reference.
 */
#ifndef RTX_H
#define RTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_NONE 0xFFFFFFFFu

enum rtx_status {
    RTX_OK = 0,
    RTX_ERR_BAD_ARG = -1,
    RTX_ERR_NO_DEVICE = -2,
    RTX_ERR_HIP = -3,
    RTX_ERR_OOM = -4,
    RTX_ERR_BUILD = -5,
    RTX_ERR_RCCL = -6
};

/* enum Geometry, src/geometry.rs:25-47 */
enum rtx_geom_kind { RTX_SPHERE = 0, RTX_QUAD = 1, RTX_TRIANGLE = 2 };
/* enum Material, src/material.rs:9-17 */
enum rtx_mat_kind { RTX_LAMBERTIAN = 0, RTX_METAL = 1, RTX_DIELECTRIC = 2, RTX_DIFFUSE_LIGHT = 3, RTX_UV_GRADIENT = 4 };

/*
 * A built BVH, flattened.  Replaces `BVH { tree: Box<[Node]> }`
 * (src/bvh/mod.rs:51-73): the same pre-order array (left child of node i is
 * i+1) with each enum field moved into its own table.  Vector tables are
 * stored as three rows: x[0..n), y[0..n), z[0..n).
 *
 * Boxes must be the ones the reference computes (Node::bounding_box,
 * src/bvh/mod.rs:59-69), including the leaf boxes: the box test is part of
 * the observable result, not only an optimisation.
 */
typedef struct rtx_scene_desc {
    uint32_t n_nodes;
    uint32_t n_prims;
    uint32_t n_materials;
    uint32_t _reserved;

    const double*   node_bmin;      /* [3][n_nodes] */
    const double*   node_bmax;      /* [3][n_nodes] */
    const uint32_t* node_right;     /* Internal: right_idx; Leaf: RTX_NONE */
    const uint32_t* node_prim;      /* Leaf: index into the prim tables; Internal: RTX_NONE */

    const uint8_t*  prim_kind;      /* rtx_geom_kind */
    const double*   prim_q;         /* [3][n_prims]  sphere: center; planar: q */
    const double*   prim_u;         /* [3][n_prims]  planar only (src/geometry.rs:31-46) */
    const double*   prim_v;
    const double*   prim_norm;
    const double*   prim_w;
    const double*   prim_d;         /* [n_prims]     planar only */
    const double*   prim_radius;    /* [n_prims]     sphere only */
    const uint32_t* prim_material;  /* index into the material tables */
    const uint32_t* prim_user_id;   /* index of the surface in the caller's original array */

    const uint32_t* mat_kind;       /* rtx_mat_kind */
    const double*   mat_rgb;        /* [3][n_materials] albedo | emit */
    const double*   mat_param;      /* fuzz_radius | refraction_index | intensity */
} rtx_scene_desc;

/* The fields of InitializedCamera, src/camera.rs:28-43 (all private there). */
typedef struct rtx_camera {
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
} rtx_camera;

/* Counters and timers of one call (the reference has only the three
 * wall-clock timers of src/runner.rs:24-38 and no ray counter). */
typedef struct rtx_stats {
    uint64_t rays;           /* closest-hit queries == world.hit() calls, every bounce */
    uint64_t paths;          /* primary samples */
    uint64_t node_visits;    /* box tests (0 unless counting was requested) */
    uint64_t sphere_tests;
    uint64_t planar_tests;
    uint64_t box_fallbacks;  /* box tests the single-precision tier left to the f64 tiers (counted with node_visits) */
    uint64_t iterations;     /* single-launch render: rounds of the busiest wave; round loop: rounds launched, over all lanes */
    double   ms_total;       /* device time of the call, HIP events */
    double   ms_traverse;    /* summed duration of the traverse kernel launches (0 unless RTX_FLAG_TIME_KERNELS). */
    double   ms_shade;       /* A render call runs several independent lanes side by side on their own streams, */
    double   ms_generate;    /* so kernels of different lanes overlap and these sums can exceed ms_total.       */
    uint32_t kernel_form;    /* which kernels a render call ran: enum rtx_kernel_form (of the call's last launch) */
    uint32_t _reserved;
} rtx_stats;

/* rtx_stats.kernel_form */
enum rtx_kernel_form {
    RTX_FORM_NONE = 0,       /* nothing was launched (or not a render call) */
    RTX_FORM_ROUND_LOOP = 1, /* k_generate / k_traverse* / k_shade launches per round */
    RTX_FORM_POOLED = 2,     /* one launch, paths in a pool in HBM between bounces (k_render_persistent) */
    RTX_FORM_RESIDENT = 3    /* one launch, paths in registers from bounce to bounce (k_render_resident) */
};

typedef struct rtx_scene rtx_scene;

/* PartitionBy / SAHBucketStrategy, src/bvh/mod.rs:13-34 */
enum rtx_partition { RTX_BISECT = 0, RTX_MIDPOINT = 1, RTX_SAH_EQUAL_SIZE = 2, RTX_SAH_PER_SURFACE = 3 };

/* Timers and sizes of one rtx_bvh_build call (the reference has one
 * wall-clock timer around BVH::from_slice, src/runner.rs:26-29). */
typedef struct rtx_build_stats {
    uint32_t levels;        /* passes over the surfaces == depth of the tree */
    uint32_t n_nodes;
    double   ms_sort;       /* upload of the boxes + the three initial sorts, HIP events */
    double   ms_levels;     /* all levels + the leaf pass */
    uint64_t device_bytes;  /* HBM held while building */
} rtx_build_stats;

/* flags for rtx_render_accumulate / rtx_trace_closest */
#define RTX_FLAG_COUNT_WORK   1u  /* fill node_visits / *_tests with the box and surface tests of the reference's walk (slower) */
#define RTX_FLAG_TIME_KERNELS 2u  /* render with the round loop (generate / traverse / shade launches) and time every launch with HIP events */
#define RTX_FLAG_EXACT_DIVIDE 4u  /* box test by six true divisions per node, exactly as src/aabb.rs:55-76 */
#define RTX_FLAG_F64_FILTER   8u  /* skip the single-precision tier of the box test */
#define RTX_FLAG_NO_LDS       16u /* keep the f32 node records in global memory even when they fit in LDS */
#define RTX_FLAG_NO_WAIT      32u /* rtx_render_accumulate only: enqueue the work on `stream` and return; no stats */
#define RTX_FLAG_COUNT_ACTUAL 64u /* fill node_visits / *_tests with the tests this library's walk makes.  They can be fewer than
                                     the reference's (RTX_FLAG_COUNT_WORK): the walk culls quads and triangles with boxes smaller
                                     than the reference's padded ones (src/aabb.rs:78-87) where that provably leaves every hit as it is */

/* Message of the calling thread's last failed call. */
const char* rtx_last_error(void);

/* Number of visible HIP devices (0 when there is none). */
int rtx_device_count(void);

/* Copies the flattened BVH into HBM of the current HIP device.  Replaces
 * holding `world: BVH` across the render (src/runner.rs:27-31).  The caller
 * keeps ownership of every array in `desc`; they are not referenced after
 * the call returns. */
/* A scene handle is used by ONE host thread at a time: all calls on it share
 * its path pool, control block and timing events (the reference calls
 * render once, from the main thread, src/runner.rs:31).  Different handles
 * may be used from different threads concurrently. */
int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out);
void rtx_scene_destroy(rtx_scene* scene);

/* BVH::from_slice(surfaces, &partition_by) on the current HIP device
 * (src/bvh/mod.rs:76-131) for every strategy of enum rtx_partition
 * (src/bvh/partition.rs:58-266); the tree is the reference's, node for node.
 * RTX_BISECT rests on sort_unstable_by (partition.rs:58-68), whose order of
 * equal keys the reference leaves open: here, as in rth_bvh_build, equal keys
 * keep their order.
 * leaf_boxes: host, [n][6] = Surface::bounding_box of every surface
 * (min x,y,z, max x,y,z).  Outputs, host: leaf_order[n] = index of the
 * surface that is the i-th leaf in pre-order (the order `surfaces` is left
 * in); node tables of 2n-1 nodes in the layout of rtx_scene_desc with
 * node_prim = i for that leaf.  Fails with RTX_ERR_BUILD where the reference
 * panics or never returns: "No valid splitting plane" (partition.rs:219), a
 * partition with an empty side.  stats may be NULL. */
int rtx_bvh_build(const double* leaf_boxes, uint32_t n, int strategy, uint32_t buckets,
                  uint32_t* leaf_order, double* node_bmin, double* node_bmax,
                  uint32_t* node_right, uint32_t* node_prim, rtx_build_stats* stats);

/* Closest hit for n rays.  Replaces n calls of `impl Hittable for BVH::hit`
 * (src/bvh/mod.rs:134-174) with ray_t = [tmin, tmax].  org/dir: host, [n][3].
 * out_id[i] = prim_user_id of the hit surface or RTX_NONE; out_t[i] = hit.t
 * (+inf on a miss).  stats may be NULL. */
int rtx_trace_closest(rtx_scene* scene, const double* org, const double* dir, uint64_t n,
                      double tmin, double tmax, uint32_t flags,
                      uint32_t* out_id, double* out_t, rtx_stats* stats);

/* No counterpart in the reference.  rtx_trace_closest for callers that cannot
 * allocate device memory: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same kernels, same results, but
 *   - org/dir/out_id/out_t are DEVICE memory of the scene's device ([n][3],
 *     [n][3], [n], [n]);
 *   - the per-component ray arrays, hit records and control block the kernels
 *     work on live in a fixed scratch array inside the library's code object
 *     (15 MiB per GPU, held until the process ends; loading it costs a
 *     process about 5 ms once, 4x that for a 2^20-ray array, which is why it is not larger) instead of buffers
 *     allocated on demand, so n is limited to RTX_TRACE_DEVICE_MAX_RAYS (larger
 *     batches are rejected with RTX_ERR_BAD_ARG: split them);
 *   - the work is enqueued on `stream` (a hipStream_t, NULL = default stream)
 *     and the call returns without waiting; outputs are complete when the
 *     stream has reached that point;
 *   - flags select the tier of the box test only (RTX_FLAG_EXACT_DIVIDE,
 *     RTX_FLAG_F64_FILTER, RTX_FLAG_NO_LDS); there is no stats output.
 * The call makes no hipMalloc/hipFree.  Every call on a GPU uses that GPU's one
 * scratch array: calls that overlap on one GPU (two streams, two threads, two
 * scenes) race; order them. */
#define RTX_TRACE_DEVICE_MAX_RAYS (1u << 18)
int rtx_trace_closest_device(rtx_scene* scene, const double* d_org, const double* d_dir, uint64_t n,
                             double tmin, double tmax, uint32_t flags,
                             uint32_t* d_out_id, double* d_out_t, void* stream);

/* Primary rays of sample index `sample` for every pixel, as the render
 * kernels generate them (sample_square + get_ray, src/camera.rs:164-182).
 * org/dir: host, [W*H][3]. */
int rtx_primary_rays(const rtx_camera* cam, uint64_t seed, uint32_t sample, double* org, double* dir);

/* The sample loop of InitializedCamera::render (src/camera.rs:133-151) for
 * sample indices [spp_begin, spp_begin+spp_count) of every pixel.  ADDS the
 * unscaled radiance sums into d_accum (device memory of the scene's device,
 * [H][W][3] f64, pixel = row*W+col as src/camera.rs:139).  Work is enqueued
 * on `stream` (a hipStream_t, NULL = default stream); the call returns after
 * the work has completed -- or, with RTX_FLAG_NO_WAIT, as soon as it is
 * enqueued: d_accum is complete when `stream` has reached that point, stats
 * is left zero, and the flag cannot be combined with the flags that need the
 * round loop (RTX_FLAG_EXACT_DIVIDE, RTX_FLAG_F64_FILTER,
 * RTX_FLAG_TIME_KERNELS: RTX_ERR_BAD_ARG) or RTX_FLAG_COUNT_WORK.  Calls on
 * one scene handle must be ordered (same stream, or the caller's own events).  Sample s of pixel p always uses RNG stream
 * (seed, p, s), so any split of the sample range over calls or GPUs yields
 * the same image up to the order of floating-point addition. */
int rtx_render_accumulate(rtx_scene* scene, const rtx_camera* cam, uint64_t seed,
                          uint32_t spp_begin, uint32_t spp_count, uint32_t flags,
                          double* d_accum, void* stream, rtx_stats* stats);

/* out[i] = accum[i] * scale on the device (`* self.pixel_samples_scale`,
 * src/camera.rs:146).  May run in place. */
int rtx_scale(const double* d_accum, uint64_t n_values, double scale, double* d_out, void* stream);

/* ppm_pixel (src/camera.rs:248-266) on the device: gamma, clamp, 8-bit. */
int rtx_quantize(const double* d_rgb, uint64_t n_values, uint8_t* d_out, void* stream);

/* The text InitializedCamera::render prints (src/camera.rs:129-131,155-161),
 * byte for byte: "P3\n{W} {H}\n255\n", then "{r} {g} {b}\n" per pixel with
 * ppm_pixel's values (src/camera.rs:248-266), formatted on the device.
 * d_rgb: device, [H][W][3] f64 linear RGB.  out_text: host, `capacity` bytes;
 * rtx_ppm_text_bound(W, H) bytes are always enough.  *out_len = bytes written
 * (no terminating NUL).  At most 2^28 pixels. */
uint64_t rtx_ppm_text_bound(uint32_t width, uint32_t height);
int rtx_ppm_text(const double* d_rgb, uint32_t width, uint32_t height, char* out_text, uint64_t capacity,
                 uint64_t* out_len, void* stream);

/* Whole render on the current device with host output: out_rgb is host
 * memory, [H][W][3] f64 linear RGB already multiplied by 1/spp.  Replaces
 * `camera.initialize().render(&world)` (src/runner.rs:31) minus the PPM
 * text output. */
int rtx_render(rtx_scene* scene, const rtx_camera* cam, uint64_t seed, uint32_t flags,
               double* out_rgb, rtx_stats* stats);

/* The whole of `camera.initialize().render(&world)` (src/camera.rs:128-162,
 * src/runner.rs:31): rtx_render plus the P3 text of rtx_ppm_text.  out_rgb
 * (host, [H][W][3]) or out_text (host, `capacity` bytes) may be NULL. */
int rtx_render_ppm(rtx_scene* scene, const rtx_camera* cam, uint64_t seed, uint32_t flags, double* out_rgb,
                   char* out_text, uint64_t capacity, uint64_t* out_len, rtx_stats* stats);

/* ---- one frame on several GPUs of a node, from one process ----
 * Replaces `camera.initialize().render(&world)` of RenderRunner::run
 * (src/runner.rs:23-41) when the caller has more than one GPU: the sample loop
 * (src/camera.rs:133-151) is split by sample index, device g of n renders
 * samples [begin_g, begin_g + count_g) of every pixel (500 over 8 devices ->
 * 63,63,63,63,62,62,62,62) with the scene replicated in its HBM, and the f64
 * accumulation buffers are summed onto device 0 with one RCCL reduce over xGMI
 * (the reference's per-pixel `.reduce(|| ZERO, |a, b| a + b)`,
 * src/camera.rs:147-148), then multiplied by 1/spp there.  Sample s of pixel p
 * uses RNG stream (seed, p, s) whatever n is, so the image does not depend on
 * n up to the order of the f64 additions.  librccl is loaded on first use;
 * RTX_ERR_RCCL reports its absence or its errors. */
#define RTX_MULTI_MAX_GPUS 16
typedef struct rtx_multi rtx_multi;
typedef struct rtx_multi_stats {
    uint64_t rays;                             /* over all devices */
    uint64_t paths;
    uint32_t n_gpus;
    uint32_t collective_ranks;                 /* ranks of the communicator the reduce ran on (ncclCommCount) */
    double   ms_frame;                         /* host wall clock: first device call of the frame -> scaled image resident on device 0 */
    double   ms_reduce;                        /* host wall clock: last render finished -> reduce and 1/spp scaling complete */
    uint32_t spp_begin[RTX_MULTI_MAX_GPUS];    /* sample range of each device */
    uint32_t spp_count[RTX_MULTI_MAX_GPUS];
    double   ms_render[RTX_MULTI_MAX_GPUS];    /* host wall clock of each device's render call */
    double   ms_device[RTX_MULTI_MAX_GPUS];    /* device time of it (rtx_stats.ms_total) */
} rtx_multi_stats;

/* Set-up, outside any timed region: the scene uploaded to devices 0..n_gpus-1,
 * one stream per device, and the RCCL communicator clique (ncclCommInitAll,
 * hundreds of milliseconds).  1 <= n_gpus <= rtx_device_count() and
 * <= RTX_MULTI_MAX_GPUS.  The arrays of `desc` are not referenced after the call. */
int rtx_multi_create(const rtx_scene_desc* desc, int n_gpus, rtx_multi** out);
void rtx_multi_destroy(rtx_multi* multi);
/* One frame: one host thread per device runs rtx_render_accumulate on its
 * sample range; when all have finished, the calling thread issues the reduce
 * (one ncclGroupStart/End over the devices' ranks) and the 1/spp scaling.  out_rgb: host, [H][W][3] f64
 * linear RGB already multiplied by 1/spp, or NULL to leave the image on
 * device 0 (rtx_multi_image).  stats may be NULL.  One caller at a time per handle. */
int rtx_multi_render(rtx_multi* multi, const rtx_camera* cam, uint64_t seed, uint32_t flags,
                     double* out_rgb, rtx_multi_stats* stats);
/* Device-0 address of the last frame's image ([H][W][3] f64), valid until the next call on the handle. */
const double* rtx_multi_image(const rtx_multi* multi);
/* rtx_multi_create + rtx_multi_render + rtx_multi_destroy in one call (pays the set-up every time). */
int rtx_render_multi(const rtx_scene_desc* desc, const rtx_camera* cam, uint64_t seed, int n_gpus,
                     uint32_t flags, double* out_rgb, rtx_multi_stats* stats);

/* Device arithmetic self-test hooks used by the parity tests:
 * op 0: out[i] = a[i] / b[i]; op 1: out[i] = sqrt(a[i]);
 * op 2: out[i] = filtered-quotient of a[i], b[i] as the box test computes it;
 * op 3: out[i] = (float)a[i] / (float)b[i] in single precision. */
int rtx_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out);

/* The boxes the closest-hit walk tests first (host arithmetic only, no device needed): for every node of `desc` the
 * reference's box, or -- shrink != 0, as rtx_scene_create does unless RTX_TIGHT_BOXES=0 -- a smaller box for nodes above
 * quads and triangles, chosen so that no hit can change (the reference pads such boxes to three times the surface's extent,
 * src/aabb.rs:78-87).  out_bmin / out_bmax: host, [3][n_nodes]; *out_differ = nodes whose box is smaller than the reference's. */
int rtx_selftest_cull_boxes(const rtx_scene_desc* desc, int shrink, double* out_bmin, double* out_bmax, uint32_t* out_differ);

/* Philox4x32-7 uniform doubles of stream (seed, pixel, sample) computed on the device. */
int rtx_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif /* RTX_H */
