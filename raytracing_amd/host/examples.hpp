// examples.hpp -- the scenes and cameras of the reference's seven example
// programs (examples/*.rs), built with the C++ mirror of its API.  Each
// function returns what the example's main() hands to RenderRunner: the
// surfaces, in the order of its Surface::new calls, and the Camera.
// Citations are into the reference repository.
#pragma once

#include <cmath>
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "raytracing.hpp"

namespace raytracing {
namespace examples {

struct Example {
    std::vector<Surface> surfaces;
    Camera camera;
};

// Seeded stand-in for rand's OS-seeded ThreadRng where an example draws its scene at random.
struct SplitMix64 {
    uint64_t state;
    explicit SplitMix64(uint64_t seed) : state(seed) {}
    uint64_t next_u64() {
        state += 0x9E3779B97F4A7C15ull;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double random() { return (double)(next_u64() >> 11) * 0x1.0p-53; }                  // random::<f64>()
    double random_range(double lo, double hi) { return random() * (hi - lo) + lo; }    // random_range(lo..hi)
};

inline Surface sphere(Vector3 center, double radius, Material m) { return Surface(Geometry::sphere(center, radius), m); }
inline Surface quad(Vector3 q, Vector3 u, Vector3 v, Material m) { return Surface(Geometry::quadrilateral(q, u, v), m); }
inline Surface triangle(Vector3 q, Vector3 u, Vector3 v, Material m) { return Surface(Geometry::triangle(q, u, v), m); }

inline Example hello_triangle() {   // examples/hello_triangle.rs:13-51
    const double side_length = 2.0;
    const double height = side_length * std::sqrt(3.0) / 2.0;
    const Vector3 top{0.0, height * 0.5, 0.0};
    const Vector3 bottom_left{-side_length * 0.5, -height * 0.5, 0.0};
    const Vector3 bottom_right{side_length * 0.5, -height * 0.5, 0.0};
    Example e;
    e.surfaces.push_back(triangle(bottom_left, bottom_right - bottom_left, top - bottom_left, Material::UVGradient(1.0)));
    e.camera.aspect_ratio = 16.0 / 9.0;
    e.camera.image_width = 1920;
    e.camera.samples_per_pixel = 1;
    e.camera.max_depth = 50;
    e.camera.v_fov = 45.0;
    e.camera.look_from = {0.0, 0.0, 3.0};
    e.camera.look_at = {0.0, 0.0, 0.0};
    return e;
}

// examples/cover_spheres.rs:12-133: the same procedure in the same draw order, on a seeded generator.
inline Example cover_spheres(uint64_t seed = 0xC0FFEEull) {
    SplitMix64 rng(seed);
    const double small_r = 0.2, big_r = 1.0;
    Example e;
    e.surfaces.push_back(sphere({0.0, -1000.0, 0.0}, 1000.0, Material::Lambertian({0.5, 0.5, 0.5})));
    const Surface big[3] = {
        sphere({-4.0, 1.0, 0.0}, big_r, Material::Lambertian({0.4, 0.2, 0.1})),
        sphere({0.0, 1.0, 0.0}, big_r, Material::Dielectric(1.5)),
        sphere({4.0, 1.0, 0.0}, big_r, Material::Metal({0.7, 0.6, 0.5}, 0.0)),
    };
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            const double cx = (double)a + 0.9 * rng.random();
            const double cz = (double)b + 0.9 * rng.random();
            const Vector3 center{cx, small_r, cz};
            bool too_close = false;
            for (const Surface& s : big)
                if ((s.geometry.center - center).length() < big_r + small_r) {
                    too_close = true;
                    break;
                }
            if (too_close) continue;
            const double choose = rng.random();
            Material m;
            if (choose < 0.8) {
                const Vector3 v1{rng.random(), rng.random(), rng.random()};
                const Vector3 v2{rng.random(), rng.random(), rng.random()};
                m = Material::Lambertian(v1 * v2);
            } else if (choose < 0.95) {
                const Vector3 albedo{rng.random_range(0.5, 1.0), rng.random_range(0.5, 1.0), rng.random_range(0.5, 1.0)};
                m = Material::Metal(albedo, rng.random_range(0.0, 0.5));
            } else {
                m = Material::Dielectric(1.5);
            }
            e.surfaces.push_back(sphere(center, small_r, m));
        }
    for (const Surface& s : big) e.surfaces.push_back(s);
    e.camera.aspect_ratio = 16.0 / 9.0;
    e.camera.image_width = 1200;
    e.camera.samples_per_pixel = 500;
    e.camera.max_depth = 50;
    e.camera.look_from = {13.0, 2.0, 3.0};
    e.camera.look_at = {0.0, 0.0, 0.0};
    e.camera.v_fov = 20.0;
    e.camera.defocus_angle = 0.6;
    e.camera.focus_dist = 10.0;
    e.camera.background = {0.7, 0.8, 1.0};
    return e;
}

// examples/cornell_box.rs:67-130: the six faces of the box spanned by a and b, turned by theta about the vertical through its centre
inline void push_box(std::vector<Surface>& out, Vector3 a, Vector3 b, Material m, double theta) {
    const Vector3 mn{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    const Vector3 mx{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    const Vector3 center = mn + (mx - mn) * 0.5;
    const double cos_t = std::cos(theta), sin_t = std::sin(theta);
    auto rotate_y = [&](Vector3 v) {
        const Vector3 rel = v - center;
        return Vector3{cos_t * rel.x + sin_t * rel.z, rel.y, -sin_t * rel.x + cos_t * rel.z} + center;
    };
    const Vector3 v000 = rotate_y({mn.x, mn.y, mn.z}), v001 = rotate_y({mn.x, mn.y, mx.z});
    const Vector3 v010 = rotate_y({mn.x, mx.y, mn.z}), v011 = rotate_y({mn.x, mx.y, mx.z});
    const Vector3 v100 = rotate_y({mx.x, mn.y, mn.z}), v101 = rotate_y({mx.x, mn.y, mx.z});
    const Vector3 v110 = rotate_y({mx.x, mx.y, mn.z}), v111 = rotate_y({mx.x, mx.y, mx.z});
    out.push_back(quad(v001, v101 - v001, v011 - v001, m));   // front
    out.push_back(quad(v100, v000 - v100, v110 - v100, m));   // back
    out.push_back(quad(v000, v001 - v000, v010 - v000, m));   // left
    out.push_back(quad(v101, v100 - v101, v111 - v101, m));   // right
    out.push_back(quad(v000, v100 - v000, v001 - v000, m));   // bottom
    out.push_back(quad(v010, v011 - v010, v110 - v010, m));   // top
}

inline Example cornell_box() {   // examples/cornell_box.rs:10-202
    const Material white = Material::Lambertian({0.73, 0.73, 0.73});
    const Material red = Material::Lambertian({0.65, 0.05, 0.05});
    const Material green = Material::Lambertian({0.12, 0.45, 0.15});
    const Material light = Material::DiffuseLight({50.0, 50.0, 50.0});
    const double pi = 3.141592653589793;
    Example e;
    push_box(e.surfaces, Vector3{0.0, 0.0, 0.0} + Vector3{265.0, 0.0, 295.0}, Vector3{165.0, 330.0, 165.0} + Vector3{265.0, 0.0, 295.0},
             Material::Metal({0.7, 0.6, 0.5}, 0.0), 18.0 * (pi / 180.0));
    push_box(e.surfaces, Vector3{0.0, 0.0, 0.0} + Vector3{100.0, 0.0, 65.0}, Vector3{165.0, 165.0, 165.0} + Vector3{100.0, 0.0, 65.0}, white,
             -18.0 * (pi / 180.0));
    e.surfaces.push_back(quad({555.0, 0.0, 0.0}, {0.0, 555.0, 0.0}, {0.0, 0.0, 555.0}, red));
    e.surfaces.push_back(quad({0.0, 0.0, 0.0}, {0.0, 555.0, 0.0}, {0.0, 0.0, 555.0}, green));
    e.surfaces.push_back(quad({343.0, 554.0, 332.0}, {-130.0, 0.0, 0.0}, {0.0, 0.0, -105.0}, light));
    e.surfaces.push_back(quad({0.0, 0.0, 0.0}, {555.0, 0.0, 0.0}, {0.0, 0.0, 555.0}, white));
    e.surfaces.push_back(quad({555.0, 555.0, 555.0}, {-555.0, 0.0, 0.0}, {0.0, 0.0, -555.0}, white));
    e.surfaces.push_back(quad({0.0, 0.0, 555.0}, {555.0, 0.0, 0.0}, {0.0, 555.0, 0.0}, white));
    e.camera.aspect_ratio = 1.0;
    e.camera.image_width = 600;
    e.camera.samples_per_pixel = 2000;
    e.camera.max_depth = 50;
    e.camera.v_fov = 40.0;
    e.camera.look_from = {278.0, 278.0, -800.0};
    e.camera.look_at = {278.0, 278.0, 0.0};
    return e;
}

inline Example quads() {   // examples/quads.rs:13-98
    Example e;
    e.surfaces.push_back(quad({-3.0, -2.0, 5.0}, {0.0, 0.0, -4.0}, {0.0, 4.0, 0.0}, Material::Lambertian({1.0, 0.2, 0.2})));
    e.surfaces.push_back(quad({-2.0, -2.0, 0.0}, {4.0, 0.0, 0.0}, {0.0, 4.0, 0.0}, Material::Lambertian({0.2, 1.0, 0.2})));
    e.surfaces.push_back(quad({3.0, -2.0, 1.0}, {0.0, 0.0, 4.0}, {0.0, 4.0, 0.0}, Material::Lambertian({0.2, 0.2, 1.0})));
    e.surfaces.push_back(quad({-2.0, 3.0, 1.0}, {4.0, 0.0, 0.0}, {0.0, 0.0, 4.0}, Material::Lambertian({1.0, 0.5, 0.0})));
    e.surfaces.push_back(quad({-2.0, -3.0, 5.0}, {4.0, 0.0, 0.0}, {0.0, 0.0, -4.0}, Material::Lambertian({0.2, 0.8, 0.8})));
    e.camera.aspect_ratio = 1.0;
    e.camera.image_width = 400;
    e.camera.samples_per_pixel = 100;
    e.camera.max_depth = 50;
    e.camera.v_fov = 80.0;
    e.camera.look_from = {0.0, 0.0, 9.0};
    e.camera.look_at = {0.0, 0.0, 0.0};
    e.camera.background = {0.7, 0.8, 1.0};
    return e;
}

inline Example simple_light() {   // examples/simple_light.rs:13-70
    Example e;
    e.surfaces.push_back(sphere({0.0, -1000.0, 0.0}, 1000.0, Material::Lambertian({0.6, 0.5, 0.4})));
    e.surfaces.push_back(sphere({0.0, 2.0, 0.0}, 2.0, Material::Lambertian({0.8, 0.4, 0.6})));
    e.surfaces.push_back(quad({3.0, 1.0, -2.0}, {2.0, 0.0, 0.0}, {0.0, 2.0, 0.0}, Material::DiffuseLight({10.0, 10.0, 10.0})));
    e.camera.aspect_ratio = 16.0 / 9.0;
    e.camera.image_width = 400;
    e.camera.samples_per_pixel = 1000;
    e.camera.max_depth = 50;
    e.camera.v_fov = 20.0;
    e.camera.look_from = {26.0, 3.0, 6.0};
    e.camera.look_at = {0.0, 2.0, 0.0};
    e.camera.focus_dist = 1.0;
    return e;
}

inline Example simple_scene() {   // examples/simple_scene.rs:13-70
    Example e;
    e.surfaces.push_back(sphere({0.0, 0.0, -1.0}, 0.5, Material::Lambertian({0.7, 0.3, 0.3})));
    e.surfaces.push_back(sphere({-1.0, 0.0, -1.0}, 0.5, Material::Lambertian({0.3, 0.3, 0.7})));
    e.surfaces.push_back(sphere({1.0, 0.0, -1.0}, 0.5, Material::Metal({0.8, 0.8, 0.9}, 0.0)));
    e.surfaces.push_back(sphere({0.0, -100.5, -1.0}, 100.0, Material::Lambertian({0.8, 0.8, 0.0})));
    e.camera.aspect_ratio = 16.0 / 9.0;
    e.camera.image_width = 400;
    e.camera.samples_per_pixel = 100;
    e.camera.max_depth = 50;
    e.camera.v_fov = 90.0;
    e.camera.look_from = {0.0, 0.0, 0.0};
    e.camera.look_at = {0.0, 0.0, -1.0};
    e.camera.focus_dist = 1.0;
    e.camera.background = {0.7, 0.8, 1.0};
    return e;
}

inline Example demo_spheres() {   // examples/demo_spheres.rs:10-79
    Example e;
    e.surfaces.push_back(sphere({0.0, -100.5, -1.0}, 100.0, Material::Lambertian({0.8, 0.8, 0.0})));
    e.surfaces.push_back(sphere({0.0, 0.0, -1.2}, 0.5, Material::Lambertian({0.1, 0.2, 0.5})));
    e.surfaces.push_back(sphere({-1.0, 0.0, -1.0001}, 0.5, Material::Dielectric(1.5)));
    e.surfaces.push_back(sphere({-1.0, 0.0, -1.0}, 0.4, Material::Dielectric(1.0 / 1.5)));
    e.surfaces.push_back(sphere({1.0, 0.0, -1.0}, 0.5, Material::Metal({0.8, 0.6, 0.2}, 1.0)));
    e.camera.aspect_ratio = 16.0 / 9.0;
    e.camera.image_width = 400;
    e.camera.samples_per_pixel = 100;
    e.camera.max_depth = 50;
    e.camera.look_from = {-2.0, 2.0, 1.0};
    e.camera.look_at = {0.0, 0.0, -1.0};
    e.camera.v_fov = 20.0;
    e.camera.defocus_angle = 10.0;
    e.camera.focus_dist = 3.4;
    e.camera.background = {0.7, 0.8, 1.0};
    return e;
}

inline const char* const NAMES[] = {"hello_triangle", "cover_spheres", "cornell_box", "quads", "simple_light", "simple_scene", "demo_spheres"};

// The example of that name; `seed` only matters for cover_spheres.  Throws std::invalid_argument for an unknown name.
inline Example by_name(const std::string& name, uint64_t seed = 0xC0FFEEull) {
    if (name == "hello_triangle") return hello_triangle();
    if (name == "cover_spheres") return cover_spheres(seed);
    if (name == "cornell_box") return cornell_box();
    if (name == "quads") return quads();
    if (name == "simple_light") return simple_light();
    if (name == "simple_scene") return simple_scene();
    if (name == "demo_spheres") return demo_spheres();
    throw std::invalid_argument("unknown example '" + name + "'");
}

}  // namespace examples
}  // namespace raytracing
