// rtx_host.cpp -- C entry points over raytracing.hpp (see include/rtx_host.h).
#include "../../include/rtx_host.h"
#include "raytracing.hpp"
#include "examples.hpp"

#include <string>

using namespace raytracing;

namespace {
thread_local std::string g_err;

Vector3 vec(const double a[3]) { return Vector3(a[0], a[1], a[2]); }

std::vector<Surface> to_surfaces(const rtx_surface* in, uint32_t n) {
    std::vector<Surface> out;
    out.reserve(n);
    for (uint32_t i = 0; i < n; ++i) {
        const rtx_surface& s = in[i];
        Geometry g;
        switch (s.geom_kind) {
            case RTX_SPHERE: g = Geometry::sphere(vec(s.q), s.radius); break;
            case RTX_QUAD: g = Geometry::quadrilateral(vec(s.q), vec(s.u), vec(s.v)); break;
            case RTX_TRIANGLE: g = Geometry::triangle(vec(s.q), vec(s.u), vec(s.v)); break;
            default: throw BuildError("unknown geometry kind");
        }
        if (s.mat_kind > RTX_UV_GRADIENT) throw BuildError("unknown material kind");
        Material m{(Material::Kind)s.mat_kind, vec(s.rgb), s.param};
        out.emplace_back(g, m);
    }
    return out;
}

PartitionBy to_partition(int strategy, uint32_t buckets) {
    switch (strategy) {
        case RTH_BISECT: return PartitionBy::bisect();
        case RTH_MIDPOINT: return PartitionBy::midpoint();
        case RTH_SAH_EQUAL_SIZE: return PartitionBy::surface_area_heuristic(SAHBucketStrategy::equal_size(buckets));
        case RTH_SAH_PER_SURFACE: return PartitionBy::surface_area_heuristic(SAHBucketStrategy::per_surface());
        default: throw BuildError("unknown partition strategy");
    }
}

Camera to_camera(const rtx_camera_config& c) {
    Camera cam;
    cam.aspect_ratio = c.aspect_ratio;
    cam.image_width = c.image_width;
    cam.samples_per_pixel = c.samples_per_pixel;
    cam.max_depth = c.max_depth;
    cam.v_fov = c.v_fov;
    cam.look_from = vec(c.look_from);
    cam.look_at = vec(c.look_at);
    cam.v_up = vec(c.v_up);
    cam.defocus_angle = c.defocus_angle;
    cam.focus_dist = c.focus_dist;
    cam.background = vec(c.background);
    return cam;
}
}  // namespace

struct rth_bvh {
    BVH bvh;
};

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

int rth_bvh_build(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, rth_bvh** out) {
    if (!out || (n && !surfaces)) {
        g_err = "rth_bvh_build: null argument";
        return RTX_ERR_BAD_ARG;
    }
    *out = nullptr;
    try {
        BVH b = BVH::from_slice(to_surfaces(surfaces, n), to_partition(strategy, buckets));
        *out = new rth_bvh{std::move(b)};
        return RTX_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RTX_ERR_BUILD;
    }
}

int rth_bvh_build_device(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, rth_bvh** out,
                         rtx_build_stats* stats) {
    if (!out || (n && !surfaces)) {
        g_err = "rth_bvh_build_device: null argument";
        return RTX_ERR_BAD_ARG;
    }
    *out = nullptr;
    try {
        BVH b = BVH::from_slice_on_device(to_surfaces(surfaces, n), to_partition(strategy, buckets), stats);
        *out = new rth_bvh{std::move(b)};
        return RTX_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RTX_ERR_BUILD;
    }
}

void rth_bvh_desc(const rth_bvh* bvh, rtx_scene_desc* out) { *out = bvh->bvh.desc(); }

void rth_bvh_destroy(rth_bvh* bvh) { delete bvh; }

int rth_camera_initialize(const rtx_camera_config* cfg, rtx_camera* out) {
    if (!cfg || !out) {
        g_err = "rth_camera_initialize: null argument";
        return RTX_ERR_BAD_ARG;
    }
    *out = to_camera(*cfg).initialize();
    return RTX_OK;
}

int rth_example(const char* name, uint64_t scene_seed, rtx_surface* surfaces, uint32_t capacity, uint32_t* n, rtx_camera_config* cfg) {
    if (!name || !n || !cfg) {
        g_err = "rth_example: null argument";
        return RTX_ERR_BAD_ARG;
    }
    try {
        const examples::Example e = examples::by_name(name, scene_seed);
        *n = (uint32_t)e.surfaces.size();
        for (uint32_t i = 0; surfaces && i < *n && i < capacity; ++i) {
            const Surface& s = e.surfaces[i];
            rtx_surface& o = surfaces[i];
            std::memset(&o, 0, sizeof o);
            o.geom_kind = (uint32_t)s.geometry.kind;
            o.mat_kind = (uint32_t)s.material.kind;
            const Vector3 q = s.geometry.kind == Geometry::Sphere ? s.geometry.center : s.geometry.q;
            for (int a = 0; a < 3; ++a) {
                o.q[a] = q[a];
                o.u[a] = s.geometry.u[a];
                o.v[a] = s.geometry.v[a];
                o.rgb[a] = s.material.rgb[a];
            }
            o.radius = s.geometry.radius;
            o.param = s.material.param;
        }
        const Camera& c = e.camera;
        std::memset(cfg, 0, sizeof *cfg);
        cfg->aspect_ratio = c.aspect_ratio;
        cfg->image_width = c.image_width;
        cfg->samples_per_pixel = c.samples_per_pixel;
        cfg->max_depth = c.max_depth;
        cfg->v_fov = c.v_fov;
        for (int a = 0; a < 3; ++a) {
            cfg->look_from[a] = c.look_from[a];
            cfg->look_at[a] = c.look_at[a];
            cfg->v_up[a] = c.v_up[a];
            cfg->background[a] = c.background[a];
        }
        cfg->defocus_angle = c.defocus_angle;
        cfg->focus_dist = c.focus_dist;
        return RTX_OK;
    } catch (const std::exception& e) {
        g_err = e.what();
        return RTX_ERR_BAD_ARG;
    }
}

static int run_impl(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
                    uint64_t seed, const char* ppm_path, double* out_rgb, bool build_on_device, int n_gpus = 1, bool through_multi = false) {
    if (!cfg || (n && !surfaces)) {
        g_err = "rth_run: null argument";
        return RTX_ERR_BAD_ARG;
    }
    FILE* f = nullptr;
    try {
        RenderRunner runner;
        runner.camera = to_camera(*cfg);
        runner.partition_strategy = to_partition(strategy, buckets);
        runner.seed = seed;
        runner.build_on_device = build_on_device;
        runner.n_gpus = n_gpus;
        runner.through_multi = through_multi;
        runner.keep_image = out_rgb != nullptr || ppm_path == nullptr;
        if (ppm_path) {
            f = std::fopen(ppm_path, "wb");
            if (!f) throw RenderError(std::string("cannot open ") + ppm_path);
        }
        std::vector<double> img = runner.run(to_surfaces(surfaces, n), f, nullptr);
        if (f) std::fclose(f);
        if (out_rgb) std::memcpy(out_rgb, img.data(), img.size() * sizeof(double));
        return RTX_OK;
    } catch (const BuildError& e) {
        if (f) std::fclose(f);
        g_err = e.what();
        return RTX_ERR_BUILD;
    } catch (const std::exception& e) {
        if (f) std::fclose(f);
        g_err = e.what();
        return RTX_ERR_HIP;
    }
}

int rth_run(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
            uint64_t seed, const char* ppm_path, double* out_rgb) {
    return run_impl(surfaces, n, strategy, buckets, cfg, seed, ppm_path, out_rgb, false);
}

int rth_run_device_build(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
                         uint64_t seed, const char* ppm_path, double* out_rgb) {
    return run_impl(surfaces, n, strategy, buckets, cfg, seed, ppm_path, out_rgb, true);
}

int rth_run_multi(const rtx_surface* surfaces, uint32_t n, int strategy, uint32_t buckets, const rtx_camera_config* cfg,
                  uint64_t seed, int n_gpus, const char* ppm_path, double* out_rgb) {
    if (n_gpus < 1) {
        g_err = "rth_run_multi: n_gpus must be at least 1";
        return RTX_ERR_BAD_ARG;
    }
    return run_impl(surfaces, n, strategy, buckets, cfg, seed, ppm_path, out_rgb, false, n_gpus, true);   // also with one GPU: communicator, reduce, scaling
}

}  // extern "C"
