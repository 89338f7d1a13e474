// rtx_kernels.h -- host-callable launchers of the kernels in rtx_kernels.hip.
#pragma once
#include <string>

#include "../../include/rtx.h"
#include "rtx_device.h"

namespace rtx {

void launch_generate(const DCamera& cam, const DState& st, DRenderCtl* ctl, const uint32_t* surv, uint32_t* live, uint32_t n_sub,
                     uint32_t sub_capacity, uint32_t parity, uint64_t seed, uint32_t spp_begin, hipStream_t s);
void launch_primary_rays(const DCamera& cam, uint64_t seed, uint32_t sample, uint32_t npix, double* org, double* dir,
                         uint32_t max_blocks, hipStream_t s);
// The pool kernels run one wave per sub-pool of `pool`.
void launch_traverse(const DScene& sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                     DRenderCtl* ctl, bool count_work, bool exact, hipStream_t s);
void launch_traverse_f32(const DScene& sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                         DRenderCtl* ctl, bool count_work, uint32_t lds_mode, hipStream_t s);
void launch_shade(const DScene& sc, const DCamera& cam, const DState& cur, const DHits& hits, const DPool& pool, const DState& nxt,
                  uint32_t* counts_out, uint64_t seed, double* accum, hipStream_t s);
// The whole sample loop as one launch of persistent waves (f32 box tier only).  `params` is device memory.
size_t persistent_lds_bytes(const DScene& sc, uint32_t lds_mode, bool recs);
uint32_t persistent_waves_per_cu(const DScene& sc, uint32_t lds_mode, bool recs);
void launch_render_persistent(const DRenderParams* params, const DRenderParams& host_params, bool count_work, uint32_t lds_mode, hipStream_t s);
// The same with every path resident in registers (scenes whose tree, surface tables and shade records fit one LDS copy per CU)
uint32_t resident_waves_per_cu(const DScene& sc, uint32_t batch);   // 0: the kernel cannot be resident with that scene
void launch_render_resident(const DRenderParams* params, const DRenderParams& host_params, hipStream_t s);
void launch_scale(const double* in, uint64_t n, double scale, double* out, hipStream_t s);
void launch_add(const double* in, uint64_t n, double* out, hipStream_t s);   // out[i] += in[i]
void launch_quantize(const double* in, uint64_t n, uint8_t* out, hipStream_t s);
void launch_rays_to_state(const double* org, const double* dir, uint64_t n, const DState& st, hipStream_t s);
void launch_user_ids(const uint32_t* rec_user_id, uint32_t n_spheres, uint32_t* ids, uint64_t n, hipStream_t s);
void launch_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out, hipStream_t s);
void launch_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out, hipStream_t s);

// BVH::from_slice on the current device (rtx_build.hip); arguments as rtx_bvh_build, n >= 1.
int build_bvh_device(const double* leaf_boxes, uint32_t n, int strategy, uint32_t buckets, uint32_t* leaf_order, double* node_bmin,
                     double* node_bmax, uint32_t* node_right, uint32_t* node_prim, rtx_build_stats* stats, std::string& msg);

// The P3 text of an image on the current device (rtx_output.hip); arguments as rtx_ppm_text.
int ppm_text_device(const double* d_rgb, uint32_t width, uint32_t height, char* out, uint64_t capacity, uint64_t* out_len, hipStream_t st,
                    std::string& msg);

}  // namespace rtx
