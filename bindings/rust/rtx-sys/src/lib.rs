//! Raw bindings to `librtx.so`: one item per declaration of `include/rtx.h`, same names, same layout.
//! Nothing here is safe; the in-crate adapter (`bindings/rust/gpu.rs`) wraps it.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

pub const RTX_NONE: u32 = 0xFFFF_FFFF;
pub const RTX_TRACE_DEVICE_MAX_RAYS: u64 = 1 << 18;

// enum rtx_status
pub const RTX_OK: c_int = 0;
pub const RTX_ERR_BAD_ARG: c_int = -1;
pub const RTX_ERR_NO_DEVICE: c_int = -2;
pub const RTX_ERR_HIP: c_int = -3;
pub const RTX_ERR_OOM: c_int = -4;
pub const RTX_ERR_BUILD: c_int = -5;
pub const RTX_ERR_RCCL: c_int = -6;
pub const RTX_MULTI_MAX_GPUS: usize = 16;

// enum rtx_geom_kind (enum Geometry, src/geometry.rs:25-47)
pub const RTX_SPHERE: u32 = 0;
pub const RTX_QUAD: u32 = 1;
pub const RTX_TRIANGLE: u32 = 2;

// enum rtx_mat_kind (enum Material, src/material.rs:9-17)
pub const RTX_LAMBERTIAN: u32 = 0;
pub const RTX_METAL: u32 = 1;
pub const RTX_DIELECTRIC: u32 = 2;
pub const RTX_DIFFUSE_LIGHT: u32 = 3;
pub const RTX_UV_GRADIENT: u32 = 4;

// enum rtx_partition (PartitionBy / SAHBucketStrategy, src/bvh/mod.rs:13-34)
pub const RTX_BISECT: c_int = 0;
pub const RTX_MIDPOINT: c_int = 1;
pub const RTX_SAH_EQUAL_SIZE: c_int = 2;
pub const RTX_SAH_PER_SURFACE: c_int = 3;

// flags of rtx_render_accumulate / rtx_trace_closest
pub const RTX_FLAG_COUNT_WORK: u32 = 1;
pub const RTX_FLAG_TIME_KERNELS: u32 = 2;
pub const RTX_FLAG_EXACT_DIVIDE: u32 = 4;
pub const RTX_FLAG_F64_FILTER: u32 = 8;
pub const RTX_FLAG_NO_LDS: u32 = 16;
pub const RTX_FLAG_NO_WAIT: u32 = 32;
pub const RTX_FLAG_COUNT_ACTUAL: u32 = 64;

/// A built BVH, flattened: the pre-order node array of `BVH { tree }` (src/bvh/mod.rs:51-73) with each enum
/// field in its own table.  Vector tables are three rows: x[0..n), y[0..n), z[0..n).
#[repr(C)]
pub struct rtx_scene_desc {
    pub n_nodes: u32,
    pub n_prims: u32,
    pub n_materials: u32,
    pub _reserved: u32,
    pub node_bmin: *const f64,
    pub node_bmax: *const f64,
    pub node_right: *const u32,
    pub node_prim: *const u32,
    pub prim_kind: *const u8,
    pub prim_q: *const f64,
    pub prim_u: *const f64,
    pub prim_v: *const f64,
    pub prim_norm: *const f64,
    pub prim_w: *const f64,
    pub prim_d: *const f64,
    pub prim_radius: *const f64,
    pub prim_material: *const u32,
    pub prim_user_id: *const u32,
    pub mat_kind: *const u32,
    pub mat_rgb: *const f64,
    pub mat_param: *const f64,
}

/// The fields of `InitializedCamera` (src/camera.rs:28-43).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_camera {
    pub image_width: u32,
    pub image_height: u32,
    pub samples_per_pixel: u32,
    pub max_depth: u32,
    pub defocus_angle: f64,
    pub background: [f64; 3],
    pub pixel_samples_scale: f64,
    pub center: [f64; 3],
    pub pixel00_loc: [f64; 3],
    pub pixel_du: [f64; 3],
    pub pixel_dv: [f64; 3],
    pub defocus_disk_u: [f64; 3],
    pub defocus_disk_v: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_stats {
    pub rays: u64,
    pub paths: u64,
    pub node_visits: u64,
    pub sphere_tests: u64,
    pub planar_tests: u64,
    pub box_fallbacks: u64,
    pub iterations: u64,
    pub ms_total: f64,
    pub ms_traverse: f64,
    pub ms_shade: f64,
    pub ms_generate: f64,
    pub kernel_form: u32,
    pub _reserved: u32,
}

/// `rtx_stats.kernel_form`
pub const RTX_FORM_NONE: u32 = 0;
pub const RTX_FORM_ROUND_LOOP: u32 = 1;
pub const RTX_FORM_POOLED: u32 = 2;
pub const RTX_FORM_RESIDENT: u32 = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_build_stats {
    pub levels: u32,
    pub n_nodes: u32,
    pub ms_sort: f64,
    pub ms_levels: f64,
    pub device_bytes: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_multi_stats {
    pub rays: u64,
    pub paths: u64,
    pub n_gpus: u32,
    pub collective_ranks: u32,
    pub ms_frame: f64,
    pub ms_reduce: f64,
    pub spp_begin: [u32; RTX_MULTI_MAX_GPUS],
    pub spp_count: [u32; RTX_MULTI_MAX_GPUS],
    pub ms_render: [f64; RTX_MULTI_MAX_GPUS],
    pub ms_device: [f64; RTX_MULTI_MAX_GPUS],
}

#[repr(C)]
pub struct rtx_scene {
    _private: [u8; 0],
}

#[repr(C)]
pub struct rtx_multi {
    _private: [u8; 0],
}

extern "C" {
    pub fn rtx_last_error() -> *const c_char;
    pub fn rtx_device_count() -> c_int;
    pub fn rtx_scene_create(desc: *const rtx_scene_desc, out: *mut *mut rtx_scene) -> c_int;
    pub fn rtx_scene_destroy(scene: *mut rtx_scene);
    pub fn rtx_bvh_build(
        leaf_boxes: *const f64,
        n: u32,
        strategy: c_int,
        buckets: u32,
        leaf_order: *mut u32,
        node_bmin: *mut f64,
        node_bmax: *mut f64,
        node_right: *mut u32,
        node_prim: *mut u32,
        stats: *mut rtx_build_stats,
    ) -> c_int;
    pub fn rtx_trace_closest(
        scene: *mut rtx_scene,
        org: *const f64,
        dir: *const f64,
        n: u64,
        tmin: f64,
        tmax: f64,
        flags: u32,
        out_id: *mut u32,
        out_t: *mut f64,
        stats: *mut rtx_stats,
    ) -> c_int;
    pub fn rtx_trace_closest_device(
        scene: *mut rtx_scene,
        d_org: *const f64,
        d_dir: *const f64,
        n: u64,
        tmin: f64,
        tmax: f64,
        flags: u32,
        d_out_id: *mut u32,
        d_out_t: *mut f64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rtx_primary_rays(cam: *const rtx_camera, seed: u64, sample: u32, org: *mut f64, dir: *mut f64) -> c_int;
    pub fn rtx_render_accumulate(
        scene: *mut rtx_scene,
        cam: *const rtx_camera,
        seed: u64,
        spp_begin: u32,
        spp_count: u32,
        flags: u32,
        d_accum: *mut f64,
        stream: *mut c_void,
        stats: *mut rtx_stats,
    ) -> c_int;
    pub fn rtx_scale(d_accum: *const f64, n_values: u64, scale: f64, d_out: *mut f64, stream: *mut c_void) -> c_int;
    pub fn rtx_quantize(d_rgb: *const f64, n_values: u64, d_out: *mut u8, stream: *mut c_void) -> c_int;
    pub fn rtx_ppm_text_bound(width: u32, height: u32) -> u64;
    pub fn rtx_ppm_text(
        d_rgb: *const f64,
        width: u32,
        height: u32,
        out_text: *mut c_char,
        capacity: u64,
        out_len: *mut u64,
        stream: *mut c_void,
    ) -> c_int;
    pub fn rtx_render(
        scene: *mut rtx_scene,
        cam: *const rtx_camera,
        seed: u64,
        flags: u32,
        out_rgb: *mut f64,
        stats: *mut rtx_stats,
    ) -> c_int;
    pub fn rtx_render_ppm(
        scene: *mut rtx_scene,
        cam: *const rtx_camera,
        seed: u64,
        flags: u32,
        out_rgb: *mut f64,
        out_text: *mut c_char,
        capacity: u64,
        out_len: *mut u64,
        stats: *mut rtx_stats,
    ) -> c_int;
    pub fn rtx_multi_create(desc: *const rtx_scene_desc, n_gpus: c_int, out: *mut *mut rtx_multi) -> c_int;
    pub fn rtx_multi_destroy(multi: *mut rtx_multi);
    pub fn rtx_multi_render(
        multi: *mut rtx_multi,
        cam: *const rtx_camera,
        seed: u64,
        flags: u32,
        out_rgb: *mut f64,
        stats: *mut rtx_multi_stats,
    ) -> c_int;
    pub fn rtx_multi_image(multi: *const rtx_multi) -> *const f64;
    pub fn rtx_render_multi(
        desc: *const rtx_scene_desc,
        cam: *const rtx_camera,
        seed: u64,
        n_gpus: c_int,
        flags: u32,
        out_rgb: *mut f64,
        stats: *mut rtx_multi_stats,
    ) -> c_int;
    pub fn rtx_selftest_math(op: c_int, a: *const f64, b: *const f64, n: u64, out: *mut f64) -> c_int;
    pub fn rtx_selftest_rng(seed: u64, pixel: u32, sample: u32, n: u32, out: *mut f64) -> c_int;
    pub fn rtx_selftest_cull_boxes(desc: *const rtx_scene_desc, shrink: c_int, out_bmin: *mut f64, out_bmax: *mut f64, out_differ: *mut u32) -> c_int;
}

#[cfg(test)]
mod layout {
    use super::*;
    use std::mem::size_of;

    #[test]
    fn struct_sizes_match_the_header() {
        assert_eq!(size_of::<rtx_scene_desc>(), 16 + 17 * 8);
        assert_eq!(size_of::<rtx_camera>(), 16 + 8 + 24 + 8 + 6 * 24);
        assert_eq!(size_of::<rtx_stats>(), 7 * 8 + 4 * 8);
        assert_eq!(size_of::<rtx_build_stats>(), 8 + 8 + 8 + 8);
        assert_eq!(size_of::<rtx_multi_stats>(), 16 + 8 + 16 + RTX_MULTI_MAX_GPUS * 24);
    }
}
