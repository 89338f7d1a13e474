//! `src/gpu.rs` of seridescent/raytracing: the adapter between the crate's own types and `librtx.so`.
//!
//! To switch the renderer to the MI355X backend a maintainer
//!   1. copies this file to `src/gpu.rs` and adds `pub mod gpu;` to `src/lib.rs`,
//!   2. adds `rtx-sys = { path = "<backend checkout>/bindings/rust/rtx-sys" }` to `[dependencies]`,
//!   3. calls `RenderRunner::run_gpu` where an example calls `RenderRunner::run`
//!      (`RTX_LIB_DIR=<backend checkout>/raytracing_amd cargo run --release --example cover_spheres > image.ppm`).
//! It lives inside the crate because `Node`, `BVH.tree` (src/bvh/mod.rs:52,72) and the fields of
//! `InitializedCamera` (src/camera.rs:28-43) are private; it needs `pub(crate)` on `Node`, on `BVH.tree`
//! and on those fields, nothing else.
//!
//! Written against the sources of the crate; it could not be compiled where the backend was built
//! (no Rust toolchain there).  The same adapter in C++ (`raytracing_amd/host/raytracing.hpp`: `BVH::desc`,
//! `BVH::from_slice_on_device`, `RenderRunner::run`) is compiled and tested.

use std::error::Error;
use std::ffi::CStr;
use std::io::Write;
use std::time::Instant;

use rtx_sys as sys;

use crate::bvh::{Node, PartitionBy, SAHBucketStrategy, BVH};
use crate::camera::InitializedCamera;
use crate::geometry::Geometry;
use crate::material::Material;
use crate::runner::RenderRunner;
use crate::surface::{Hittable, Surface};
use crate::vector::Vector3;

/// The SoA tables of `rtx_scene_desc`, owned.
#[derive(Default)]
pub(crate) struct FlatScene {
    node_bmin: Vec<f64>, // [3][n]
    node_bmax: Vec<f64>,
    node_right: Vec<u32>,
    node_prim: Vec<u32>,
    prim_kind: Vec<u8>,
    prim_q: Vec<f64>, // [3][m]
    prim_u: Vec<f64>,
    prim_v: Vec<f64>,
    prim_norm: Vec<f64>,
    prim_w: Vec<f64>,
    prim_d: Vec<f64>,
    prim_radius: Vec<f64>,
    prim_material: Vec<u32>,
    prim_user_id: Vec<u32>,
    mat_kind: Vec<u32>,
    mat_rgb: Vec<f64>, // [3][m]
    mat_param: Vec<f64>,
}

fn put_row3(table: &mut [f64], count: usize, i: usize, v: Vector3) {
    table[i] = v.x;
    table[count + i] = v.y;
    table[2 * count + i] = v.z;
}

impl FlatScene {
    /// Pre-order node array -> tables.  `user_ids[i]` is the index the i-th leaf's surface had in the caller's slice.
    fn from_tree(tree: &[Node], user_ids: &[u32]) -> Self {
        let n = tree.len();
        let m = tree.iter().filter(|node| matches!(node, Node::Leaf(_))).count();
        let mut f = FlatScene {
            node_bmin: vec![0.0; 3 * n],
            node_bmax: vec![0.0; 3 * n],
            node_right: vec![sys::RTX_NONE; n],
            node_prim: vec![sys::RTX_NONE; n],
            prim_kind: vec![0; m],
            prim_q: vec![0.0; 3 * m],
            prim_u: vec![0.0; 3 * m],
            prim_v: vec![0.0; 3 * m],
            prim_norm: vec![0.0; 3 * m],
            prim_w: vec![0.0; 3 * m],
            prim_d: vec![0.0; m],
            prim_radius: vec![0.0; m],
            prim_material: vec![0; m],
            prim_user_id: vec![0; m],
            mat_kind: vec![0; m],
            mat_rgb: vec![0.0; 3 * m],
            mat_param: vec![0.0; m],
        };
        let mut leaf = 0usize;
        for (i, node) in tree.iter().enumerate() {
            // leaf boxes included: the box test is part of the observable result (src/bvh/mod.rs:59-69)
            let bb = node.bounding_box();
            put_row3(&mut f.node_bmin, n, i, bb.min());
            put_row3(&mut f.node_bmax, n, i, bb.max());
            match node {
                Node::Internal(right, _) => f.node_right[i] = right.expect("built tree") as u32,
                Node::Leaf(surface) => {
                    f.node_prim[i] = leaf as u32;
                    f.push_surface(m, leaf, surface, user_ids.get(leaf).copied().unwrap_or(leaf as u32));
                    leaf += 1;
                }
                Node::Placeholder => unreachable!("a built tree holds no placeholder"),
            }
        }
        f
    }

    /// One row of the prim and material tables; the derived plane fields are copied, never recomputed.
    fn push_surface(&mut self, m: usize, i: usize, surface: &Surface, user_id: u32) {
        match &surface.geometry {
            Geometry::Sphere { center, radius } => {
                self.prim_kind[i] = sys::RTX_SPHERE as u8;
                put_row3(&mut self.prim_q, m, i, *center);
                self.prim_radius[i] = *radius;
            }
            Geometry::Quadrilateral { q, u, v, norm, d, w } | Geometry::Triangle { q, u, v, norm, d, w } => {
                self.prim_kind[i] = match surface.geometry {
                    Geometry::Quadrilateral { .. } => sys::RTX_QUAD as u8,
                    _ => sys::RTX_TRIANGLE as u8,
                };
                put_row3(&mut self.prim_q, m, i, *q);
                put_row3(&mut self.prim_u, m, i, *u);
                put_row3(&mut self.prim_v, m, i, *v);
                put_row3(&mut self.prim_norm, m, i, *norm);
                put_row3(&mut self.prim_w, m, i, *w);
                self.prim_d[i] = *d;
            }
        }
        let (kind, rgb, param) = match surface.material {
            Material::Lambertian { albedo } => (sys::RTX_LAMBERTIAN, albedo, 0.0),
            Material::Metal { albedo, fuzz_radius } => (sys::RTX_METAL, albedo, fuzz_radius),
            Material::Dielectric { refraction_index } => (sys::RTX_DIELECTRIC, Vector3::ZERO, refraction_index),
            Material::DiffuseLight { emit } => (sys::RTX_DIFFUSE_LIGHT, emit, 0.0),
            Material::UVGradient { intensity } => (sys::RTX_UV_GRADIENT, Vector3::ZERO, intensity),
        };
        self.prim_material[i] = i as u32; // a Surface owns its Material (src/surface.rs:15-18)
        self.prim_user_id[i] = user_id;
        self.mat_kind[i] = kind;
        put_row3(&mut self.mat_rgb, m, i, rgb);
        self.mat_param[i] = param;
    }

    /// Borrowed view for `rtx_scene_create`; valid while `self` lives.
    fn desc(&self) -> sys::rtx_scene_desc {
        sys::rtx_scene_desc {
            n_nodes: self.node_right.len() as u32,
            n_prims: self.prim_kind.len() as u32,
            n_materials: self.mat_kind.len() as u32,
            _reserved: 0,
            node_bmin: self.node_bmin.as_ptr(),
            node_bmax: self.node_bmax.as_ptr(),
            node_right: self.node_right.as_ptr(),
            node_prim: self.node_prim.as_ptr(),
            prim_kind: self.prim_kind.as_ptr(),
            prim_q: self.prim_q.as_ptr(),
            prim_u: self.prim_u.as_ptr(),
            prim_v: self.prim_v.as_ptr(),
            prim_norm: self.prim_norm.as_ptr(),
            prim_w: self.prim_w.as_ptr(),
            prim_d: self.prim_d.as_ptr(),
            prim_radius: self.prim_radius.as_ptr(),
            prim_material: self.prim_material.as_ptr(),
            prim_user_id: self.prim_user_id.as_ptr(),
            mat_kind: self.mat_kind.as_ptr(),
            mat_rgb: self.mat_rgb.as_ptr(),
            mat_param: self.mat_param.as_ptr(),
        }
    }
}

fn check(rc: i32) -> Result<(), Box<dyn Error>> {
    if rc == sys::RTX_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(sys::rtx_last_error()) }.to_string_lossy().into_owned();
    Err(format!("rtx error {rc}: {msg}").into())
}

impl BVH {
    /// Surfaces are numbered in leaf order here: the crate's `Surface` carries no identity, so the position a
    /// surface had in the caller's slice is only known to `from_slice_on_device`, which returns `leaf_order`.
    pub(crate) fn flatten(&self) -> FlatScene {
        FlatScene::from_tree(&self.tree, &[])
    }

    /// `BVH::from_slice` with the partitioning done by the device builder: the same tree, node for node.
    pub fn from_slice_on_device(surfaces: Box<[Surface]>, partition_by: &PartitionBy) -> Result<(Self, Vec<u32>), Box<dyn Error>> {
        let m = surfaces.len();
        if m == 0 {
            return Ok((BVH { tree: Box::from([]) }, vec![]));
        }
        let mut boxes = Vec::with_capacity(6 * m);
        for s in surfaces.iter() {
            let bb = s.bounding_box();
            boxes.extend_from_slice(&[bb.min().x, bb.min().y, bb.min().z, bb.max().x, bb.max().y, bb.max().z]);
        }
        let (strategy, buckets) = match partition_by {
            PartitionBy::LongestAxisBisectSlice => (sys::RTX_BISECT, 0),
            PartitionBy::LongestAxisMidpoint => (sys::RTX_MIDPOINT, 0),
            PartitionBy::SurfaceAreaHeuristic(SAHBucketStrategy::EqualSize(b)) => (sys::RTX_SAH_EQUAL_SIZE, *b),
            PartitionBy::SurfaceAreaHeuristic(SAHBucketStrategy::PerSurface) => (sys::RTX_SAH_PER_SURFACE, 0),
        };
        let n = 2 * m - 1;
        let mut leaf_order = vec![0u32; m];
        let (mut bmin, mut bmax) = (vec![0f64; 3 * n], vec![0f64; 3 * n]);
        let (mut right, mut prim) = (vec![0u32; n], vec![0u32; n]);
        check(unsafe {
            sys::rtx_bvh_build(
                boxes.as_ptr(),
                m as u32,
                strategy,
                buckets,
                leaf_order.as_mut_ptr(),
                bmin.as_mut_ptr(),
                bmax.as_mut_ptr(),
                right.as_mut_ptr(),
                prim.as_mut_ptr(),
                std::ptr::null_mut(),
            )
        })?;
        let tree: Vec<Node> = (0..n)
            .map(|i| {
                if prim[i] == sys::RTX_NONE {
                    let lo = Vector3::new(bmin[i], bmin[n + i], bmin[2 * n + i]);
                    let hi = Vector3::new(bmax[i], bmax[n + i], bmax[2 * n + i]);
                    Node::Internal(Some(right[i] as usize), crate::aabb::AABB::new(lo, hi))
                } else {
                    Node::Leaf(surfaces[leaf_order[prim[i] as usize] as usize].clone())
                }
            })
            .collect();
        Ok((BVH { tree: tree.into_boxed_slice() }, leaf_order))
    }
}

impl InitializedCamera {
    pub(crate) fn to_ffi(&self) -> sys::rtx_camera {
        let a = |v: Vector3| [v.x, v.y, v.z];
        sys::rtx_camera {
            image_width: self.image_width,
            image_height: self.image_height,
            samples_per_pixel: self.samples_per_pixel,
            max_depth: self.max_depth,
            defocus_angle: self.defocus_angle,
            background: a(self.background),
            pixel_samples_scale: self.pixel_samples_scale,
            center: a(self.center),
            pixel00_loc: a(self.pixel00_loc),
            pixel_du: a(self.pixel_du),
            pixel_dv: a(self.pixel_dv),
            defocus_disk_u: a(self.defocus_disk_u),
            defocus_disk_v: a(self.defocus_disk_v),
        }
    }

    /// `render` (src/camera.rs:128-162) on the GPU: prints the same P3 text to stdout.
    pub fn render_gpu(&self, world: &BVH, seed: u64) -> Result<(), Box<dyn Error>> {
        let flat = world.flatten();
        let cam = self.to_ffi();
        let mut scene = std::ptr::null_mut();
        check(unsafe { sys::rtx_scene_create(&flat.desc(), &mut scene) })?;
        let capacity = unsafe { sys::rtx_ppm_text_bound(cam.image_width, cam.image_height) };
        let mut text = vec![0u8; capacity as usize];
        let mut len = 0u64;
        let rc = unsafe {
            sys::rtx_render_ppm(scene, &cam, seed, 0, std::ptr::null_mut(), text.as_mut_ptr() as *mut _, capacity, &mut len, std::ptr::null_mut())
        };
        unsafe { sys::rtx_scene_destroy(scene) };
        check(rc)?;
        std::io::stdout().lock().write_all(&text[..len as usize])?;
        Ok(())
    }
}

impl InitializedCamera {
    /// The same frame on `n_gpus` GPUs of this node from this one process: samples per pixel split over the devices,
    /// one RCCL reduce onto device 0 (rtx_multi_render), then the P3 text of the summed image from device 0.
    pub fn render_multi_gpu(&self, world: &BVH, seed: u64, n_gpus: i32) -> Result<sys::rtx_multi_stats, Box<dyn Error>> {
        let flat = world.flatten();
        let cam = self.to_ffi();
        let mut multi = std::ptr::null_mut();
        check(unsafe { sys::rtx_multi_create(&flat.desc(), n_gpus, &mut multi) })?;   // scene on every device + communicator: once
        let mut stats = sys::rtx_multi_stats::default();
        let mut rc = unsafe { sys::rtx_multi_render(multi, &cam, seed, 0, std::ptr::null_mut(), &mut stats) };
        let mut text = Vec::new();
        let mut len = 0u64;
        if rc == 0 {
            let capacity = unsafe { sys::rtx_ppm_text_bound(cam.image_width, cam.image_height) };
            text.resize(capacity as usize, 0u8);
            // the image is device-0 memory; the main thread's current device is 0 unless the caller changed it
            rc = unsafe {
                sys::rtx_ppm_text(sys::rtx_multi_image(multi), cam.image_width, cam.image_height, text.as_mut_ptr() as *mut _, capacity, &mut len,
                                  std::ptr::null_mut())
            };
        }
        unsafe { sys::rtx_multi_destroy(multi) };
        check(rc)?;
        std::io::stdout().lock().write_all(&text[..len as usize])?;
        Ok(stats)
    }
}

impl RenderRunner {
    /// `run` (src/runner.rs:23-41) with the sample loop on the GPU.  The reference seeds its RNG from the OS;
    /// here the seed of the counter-based streams is drawn the same way, so two runs differ as they do there.
    pub fn run_gpu(self, surfaces: Box<[Surface]>) -> Result<(), Box<dyn Error>> {
        let start_time = Instant::now();

        let bvh_start_time = Instant::now();
        let world = BVH::from_slice(surfaces, &self.partition_strategy);
        let bvh_time = bvh_start_time.elapsed();

        let render_start_time = Instant::now();
        self.camera.initialize().render_gpu(&world, rand::random::<u64>())?;
        let render_time = render_start_time.elapsed();

        let total_time = start_time.elapsed();
        eprintln!(
            "\n\nDone!\nTotal runtime: {total_time:#?}\nBVH construction: {bvh_time:#?}\nRendering: {render_time:#?}",
        );
        Ok(())
    }
}
