// rt_oracle.cpp -- CPU restatement of seridescent/raytracing's render path.
//
// TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Written to follow the Rust
// source statement by statement; every function cites the lines it restates.
// Build with -ffp-contract=off and without fast-math: Rust never fuses a*b+c.
//
// Additions that the reference does not have (needed to compare against a
// device implementation): surface indices carried through the builder,
// ray / node / primitive counters, and a seeded counter-based RNG in place
// of rand's OS-seeded ThreadRng (src/vector.rs:3, src/camera.rs:1).

#include "rt_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

const double INF = std::numeric_limits<double>::infinity();
const double QNAN = std::numeric_limits<double>::quiet_NaN();
const double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI

// ---------------------------------------------------------------- Vector3
// src/vector.rs:8-12
struct V3 {
    double x, y, z;
};
inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // vector.rs:111-121
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // vector.rs:210-220
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // vector.rs:131-141
inline V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }     // vector.rs:143-153
inline V3 operator*(double s, V3 a) { return a * s; }                              // vector.rs:155-161
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                         // vector.rs:230-240
// Vector3 / f64 multiplies by the reciprocal, vector.rs:179-190
inline V3 operator/(V3 a, double s) {
    double inv = 1.0 / s;
    return v3(a.x * inv, a.y * inv, a.z * inv);
}
// Vector3 / Vector3 divides per component, vector.rs:198-208
inline V3 div_vv(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // vector.rs:35-37
inline double length(V3 a) { return std::sqrt(length_squared(a)); }               // vector.rs:31-33
inline V3 to_unit(V3 a) { return a / length(a); }                                 // vector.rs:39-41
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // vector.rs:242-244
inline V3 cross(V3 a, V3 b) {                                                      // vector.rs:246-252
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline bool is_near_zero(V3 a) {                                                   // vector.rs:106-108
    return std::fabs(a.x) < 1e-8 && std::fabs(a.y) < 1e-8 && std::fabs(a.z) < 1e-8;
}
// f64::clamp: NaN passes through (core::f64::clamp)
inline double clampd(double x, double lo, double hi) {
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}
// f64::min / f64::max ignore a NaN operand
inline double rmin(double a, double b) { return std::fmin(a, b); }
inline double rmax(double a, double b) { return std::fmax(a, b); }
inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }                 // vector.rs:254-256
inline V3 refract(V3 r_in, V3 n, double eta) {                                     // vector.rs:258-264
    double cos_theta = clampd(dot(-r_in, n), -1.0, 1.0);
    V3 r_out_perp = eta * (r_in + cos_theta * n);
    V3 r_out_par = -std::sqrt(std::fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_par;
}
// f64::total_cmp
inline int total_cmp(double a, double b) {
    int64_t l, r;
    std::memcpy(&l, &a, 8);
    std::memcpy(&r, &b, 8);
    l ^= (int64_t)(((uint64_t)(l >> 63)) >> 1);
    r ^= (int64_t)(((uint64_t)(r >> 63)) >> 1);
    return (l < r) ? -1 : (l > r ? 1 : 0);
}

// ---------------------------------------------------------------- Ray, Interval
struct Ray {   // src/ray.rs:4-16
    V3 origin, direction;
    V3 at(double t) const { return origin + direction * t; }
};
struct Interval {   // src/interval.rs:1-47
    double min, max;
    bool contains(double x) const { return min <= x && x <= max; }
    bool surrounds(double x) const { return min < x && x < max; }
    double clamp(double x) const { return clampd(x, min, max); }
};

// ---------------------------------------------------------------- AABB
struct AABB {   // src/aabb.rs:6-9
    V3 min, max;
};
const AABB AABB_EMPTY = {{INF, INF, INF}, {-INF, -INF, -INF}};   // aabb.rs:12-15
inline AABB aabb_new(V3 a, V3 b) {                                // aabb.rs:33-38
    return AABB{v3(rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)), v3(rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z))};
}
inline AABB aabb_merge(const AABB& a, const AABB& b) {            // aabb.rs:40-53
    return AABB{v3(rmin(a.min.x, b.min.x), rmin(a.min.y, b.min.y), rmin(a.min.z, b.min.z)),
                v3(rmax(a.max.x, b.max.x), rmax(a.max.y, b.max.y), rmax(a.max.z, b.max.z))};
}
inline V3 aabb_dimensions(const AABB& b) { return b.max - b.min; }                       // aabb.rs:25-27
inline V3 aabb_centroid(const AABB& b) { return b.min + (aabb_dimensions(b) / 2.0); }   // aabb.rs:29-31
inline AABB aabb_padded(const AABB& b, double padding) {                                 // aabb.rs:78-87
    V3 dims = aabb_dimensions(b);
    V3 pad = v3(rmax(dims.x, padding), rmax(dims.y, padding), rmax(dims.z, padding));
    return aabb_new(b.min - pad, b.max + pad);
}
// aabb.rs:55-76.  The NaN guard at :62 compares with `==` semantics
// (slice::contains), which is never true for NaN, so it is omitted.
inline bool aabb_hit(const AABB& b, const Ray& ray, const Interval& ray_t) {
    V3 t_0 = div_vv(b.min - ray.origin, ray.direction);
    V3 t_1 = div_vv(b.max - ray.origin, ray.direction);
    double lowers[3] = {rmin(t_0.x, t_1.x), rmin(t_0.y, t_1.y), rmin(t_0.z, t_1.z)};
    double uppers[3] = {rmax(t_0.x, t_1.x), rmax(t_0.y, t_1.y), rmax(t_0.z, t_1.z)};
    double lowers_max = QNAN, uppers_min = QNAN;
    for (int i = 0; i < 3; ++i) lowers_max = rmax(lowers_max, ray_t.clamp(lowers[i]));
    for (int i = 0; i < 3; ++i) uppers_min = rmin(uppers_min, ray_t.clamp(uppers[i]));
    return lowers_max < uppers_min;
}

// ---------------------------------------------------------------- Geometry
struct Hit {   // src/geometry.rs:10-23
    double t;
    V3 p;
    double alpha, beta;
    bool front_face;
    V3 face_normal;
};
struct Geometry {   // src/geometry.rs:25-47
    uint32_t kind;
    V3 q;           // sphere: center
    double radius;  // sphere
    V3 u, v, norm, w;
    double d;
};
struct Material {   // src/material.rs:9-17
    uint32_t kind;
    V3 rgb;         // albedo | emit
    double param;   // fuzz_radius | refraction_index | intensity
};
struct Surface {    // src/surface.rs:15-18 (+ caller index, an oracle addition)
    Geometry geometry;
    Material material;
    uint32_t id;
};

// geometry.rs:64-88 (quadrilateral and triangle constructors are identical)
inline void planar_ctor(Geometry& g) {
    V3 n = cross(g.u, g.v);
    g.norm = to_unit(n);
    g.d = dot(g.norm, g.q);
    g.w = n / dot(n, n);
}
// geometry.rs:135-145
inline void compute_face_normal(const Ray& ray, V3 outward, bool& front_face, V3& face_normal) {
    front_face = dot(ray.direction, outward) < 0.0;
    face_normal = front_face ? outward : -outward;
}
// geometry.rs:204-250
inline bool sphere_hit(V3 center, double radius, const Ray& ray, const Interval& ray_t, Hit& out) {
    V3 oc = center - ray.origin;
    double a = length_squared(ray.direction);
    double h = dot(ray.direction, oc);
    double c = length_squared(oc) - radius * radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = std::sqrt(discriminant);
    double root = (h - sqrtd) / a;
    if (!ray_t.surrounds(root)) {
        root = (h + sqrtd) / a;
        if (!ray_t.surrounds(root)) return false;
    }
    out.t = root;
    out.p = ray.at(root);
    V3 outward = (out.p - center) / radius;
    compute_face_normal(ray, outward, out.front_face, out.face_normal);
    double theta = std::acos(-out.p.y);
    double phi = std::atan2(-out.p.z, out.p.x) + PI;
    out.alpha = phi / (2.0 * PI);
    out.beta = theta / PI;
    return true;
}
// geometry.rs:147-154 + 166-190
inline bool uv_hit(const Geometry& g, const Ray& ray, const Interval& ray_t, double& t, V3& p, double& alpha, double& beta) {
    double denominator = dot(g.norm, ray.direction);
    if (std::fabs(denominator) < 1e-10) return false;
    t = (g.d - dot(g.norm, ray.origin)) / denominator;
    if (!ray_t.contains(t)) return false;
    p = ray.at(t);
    V3 qp = p - g.q;
    alpha = dot(g.w, cross(qp, g.v));
    beta = dot(g.w, cross(g.u, qp));
    return true;
}
// geometry.rs:264-289
inline bool quad_hit(const Geometry& g, const Ray& ray, const Interval& ray_t, Hit& out) {
    if (!uv_hit(g, ray, ray_t, out.t, out.p, out.alpha, out.beta)) return false;
    const Interval unit = {0.0, 1.0};
    if (!unit.contains(out.alpha) || !unit.contains(out.beta)) return false;
    compute_face_normal(ray, g.norm, out.front_face, out.face_normal);
    return true;
}
// geometry.rs:302-327
inline bool triangle_hit(const Geometry& g, const Ray& ray, const Interval& ray_t, Hit& out) {
    if (!uv_hit(g, ray, ray_t, out.t, out.p, out.alpha, out.beta)) return false;
    if (!(out.alpha >= 0.0 && out.beta >= 0.0 && out.alpha + out.beta <= 1.0)) return false;
    compute_face_normal(ray, g.norm, out.front_face, out.face_normal);
    return true;
}
// geometry.rs:90-110
inline bool geometry_hit(const Geometry& g, const Ray& ray, const Interval& ray_t, Hit& out) {
    switch (g.kind) {
        case ORC_SPHERE: return sphere_hit(g.q, g.radius, ray, ray_t, out);
        case ORC_QUAD: return quad_hit(g, ray, ray_t, out);
        default: return triangle_hit(g, ray, ray_t, out);
    }
}
// geometry.rs:112-133, 252-255, 291-293, 329-331
inline AABB geometry_bbox(const Geometry& g) {
    switch (g.kind) {
        case ORC_SPHERE: {
            V3 radii = v3(g.radius, g.radius, g.radius);
            return aabb_new(g.q + radii, g.q - radii);
        }
        case ORC_QUAD: return aabb_padded(aabb_new(g.q, g.q + g.u + g.v), 0.0001);
        default: return aabb_padded(aabb_merge(aabb_new(g.q, g.q + g.u), aabb_new(g.q, g.q + g.v)), 0.0001);
    }
}

bool surface_from_desc(const orc_surface& s, uint32_t id, Surface& out) {
    Geometry g{};
    g.kind = s.geom_kind;
    g.q = v3(s.q[0], s.q[1], s.q[2]);
    if (s.geom_kind == ORC_SPHERE) {
        if (s.radius < 0.0) {   // geometry.rs:56-62
            g_err = "invalid radius " + std::to_string(s.radius) + " (expected non-negative radius)";
            return false;
        }
        g.radius = s.radius;
    } else if (s.geom_kind == ORC_QUAD || s.geom_kind == ORC_TRIANGLE) {
        g.u = v3(s.u[0], s.u[1], s.u[2]);
        g.v = v3(s.v[0], s.v[1], s.v[2]);
        planar_ctor(g);
    } else {
        g_err = "unknown geometry kind";
        return false;
    }
    if (s.mat_kind > ORC_UV_GRADIENT) {
        g_err = "unknown material kind";
        return false;
    }
    out.geometry = g;
    out.material = Material{s.mat_kind, v3(s.rgb[0], s.rgb[1], s.rgb[2]), s.param};
    out.id = id;
    return true;
}

// ---------------------------------------------------------------- RNG
// Philox4x32 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"); the path streams use seven rounds, the
// fewest for which the paper reports the generator Crush-resistant.
const int PHILOX_ROUNDS = 7;
inline void philox4x32(const uint32_t c_in[4], const uint32_t k_in[2], int rounds, uint32_t out[4]) {
    uint32_t c0 = c_in[0], c1 = c_in[1], c2 = c_in[2], c3 = c_in[3];
    uint32_t k0 = k_in[0], k1 = k_in[1];
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Random numbers of one path.  Stream = (seed, pixel index, sample index):
// block b is Philox4x32-7 with counter (pixel, sample, b, 0) and key
// (seed_lo, seed_hi); it yields the 64-bit words w0 = out[1]:out[0] and
// w1 = out[3]:out[2].
//   * a uniform double takes one word, (w >> 11) * 2^-53 in [0,1), the same
//     mapping rand's StandardUniform uses (draw i is word i & 1 of block i >> 1);
//   * one trial of random_unit's rejection loop takes one whole block: three
//     42-bit uniforms (top 42 bits of w0, top 42 bits of w1, low 22 bits of w0
//     followed by the low 20 bits of w1).  rand's random_range uses 52 bits;
//     42 keeps a trial to one block and is far below any visible granularity.
struct PathRng {
    uint32_t key[2];
    uint32_t pixel, sample;
    uint32_t draw;
    uint32_t cached_block;
    double cached[2];
    PathRng(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t draw_ = 0)
        : pixel(pixel_), sample(sample_), draw(draw_), cached_block(0xFFFFFFFFu) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
    }
    // every sampling event (camera sample, one scatter call) starts on a block boundary
    void begin_event() { draw = (draw + 1u) & ~1u; }
    void block(uint32_t blk, uint64_t& w0, uint64_t& w1) const {
        uint32_t c[4] = {pixel, sample, blk, 0u}, o[4];
        philox4x32(c, key, PHILOX_ROUNDS, o);
        w0 = ((uint64_t)o[1] << 32) | o[0];
        w1 = ((uint64_t)o[3] << 32) | o[2];
    }
    double next() {
        uint32_t blk = draw >> 1;
        if (blk != cached_block) {
            uint64_t w0, w1;
            block(blk, w0, w1);
            cached[0] = (double)(w0 >> 11) * 0x1.0p-53;
            cached[1] = (double)(w1 >> 11) * 0x1.0p-53;
            cached_block = blk;
        }
        return cached[draw++ & 1u];
    }
    // rand::random_range(lo..hi) for f64: u * (hi - lo) + lo
    double range(double lo, double hi) { return next() * (hi - lo) + lo; }
    void next3(double& a, double& b, double& c) {
        begin_event();
        uint64_t w0, w1;
        block(draw >> 1, w0, w1);
        draw += 2u;
        a = (double)(w0 >> 22) * 0x1.0p-42;
        b = (double)(w1 >> 22) * 0x1.0p-42;
        c = (double)(((w0 & 0x3FFFFFull) << 20) | (w1 & 0xFFFFFull)) * 0x1.0p-42;
    }
};

// vector.rs:59-81
inline V3 random_unit(PathRng& rng) {
    for (;;) {
        double a, b, c;
        rng.next3(a, b, c);
        V3 candidate = v3(a * 2.0 + -1.0, b * 2.0 + -1.0, c * 2.0 + -1.0);   // random_range(-1.0..1.0)
        double lensq = length_squared(candidate);
        if (1e-160 < lensq && lensq <= 1.0) return candidate / std::sqrt(lensq);
    }
}
// vector.rs:83-95
inline V3 random_in_unit_disk(PathRng& rng) {
    for (;;) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        V3 candidate = v3(x, y, 0.0);
        if (length_squared(candidate) < 1.0) return candidate;
    }
}

// ---------------------------------------------------------------- Material
// material.rs:35-54
inline V3 material_emitted(const Material& m, const Hit& hit) {
    switch (m.kind) {
        case ORC_DIFFUSE_LIGHT: return m.rgb;
        case ORC_UV_GRADIENT: {
            double a0 = 0.0 - hit.alpha, a1 = 1.0 - hit.alpha, b0 = 0.0 - hit.beta, b1 = 1.0 - hit.beta;
            double r = 1.0 - std::sqrt(a0 * a0 + b0 * b0);
            double g = 1.0 - std::sqrt(a1 * a1 + b0 * b0);
            double b = 1.0 - std::sqrt(a0 * a0 + b1 * b1);
            return v3(r, g, b) * m.param;
        }
        default: return v3(0.0, 0.0, 0.0);
    }
}
// material.rs:134-137; powi(2) = x*x, powi(5) = x * ((x*x)*(x*x))
inline double reflectance(double cos_theta, double refraction_index) {
    double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
    r0 = r0 * r0;
    double x = 1.0 - cos_theta;
    double x2 = x * x;
    double x5 = x * (x2 * x2);
    return r0 + (1.0 - r0) * x5;
}
// material.rs:20-33 and the three scatter modules :57-131
inline bool material_scatter(const Material& m, const Ray& ray, const Hit& hit, PathRng& rng, Ray& out_ray, V3& attenuation) {
    rng.begin_event();
    switch (m.kind) {
        case ORC_LAMBERTIAN: {   // material.rs:61-73
            V3 direction = hit.face_normal + random_unit(rng);
            if (is_near_zero(direction)) direction = hit.face_normal;
            out_ray = Ray{hit.p, direction};
            attenuation = m.rgb;
            return true;
        }
        case ORC_METAL: {        // material.rs:84-96
            V3 reflected = reflect(ray.direction, hit.face_normal);
            V3 fuzz = random_unit(rng) * m.param;
            V3 fuzzed = to_unit(reflected) + fuzz;
            if (dot(fuzzed, hit.face_normal) > 0.0) {
                out_ray = Ray{hit.p, fuzzed};
                attenuation = m.rgb;
                return true;
            }
            return false;
        }
        case ORC_DIELECTRIC: {   // material.rs:108-130
            V3 r_in = to_unit(ray.direction);
            double eta = hit.front_face ? 1.0 / m.param : m.param;
            double cos_theta = clampd(dot(-r_in, hit.face_normal), -1.0, 1.0);
            double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
            V3 r_out;
            // `||` short-circuits: no draw on total internal reflection
            if (eta * sin_theta > 1.0 || reflectance(cos_theta, eta) > rng.next()) {
                r_out = reflect(r_in, hit.face_normal);
            } else {
                r_out = refract(r_in, hit.face_normal, eta);
            }
            out_ray = Ray{hit.p, r_out};
            attenuation = v3(1.0, 1.0, 1.0);
            return true;
        }
        default: return false;   // DiffuseLight, UVGradient: material.rs:30-31
    }
}

// ---------------------------------------------------------------- BVH
const uint32_t NONE = 0xFFFFFFFFu;
struct Node {   // src/bvh/mod.rs:51-57
    bool leaf;
    uint32_t right;    // Internal: right_idx
    AABB box;          // Internal: stored box; Leaf: surface.bounding_box()
    Surface surface;   // Leaf
};

}  // namespace

struct orc_scene {
    std::vector<Surface> surfaces;   // caller order (for the brute-force list)
    std::vector<Node> tree;          // BVH.tree, src/bvh/mod.rs:71-73
};

namespace {

inline double get_component(int axis, V3 v) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }   // partition.rs:20-26

// partition.rs:28-40: max_by keeps the LAST maximum
inline int longest_axis(const AABB& b) {
    int best = 0;
    for (int axis = 1; axis < 3; ++axis) {
        double a = get_component(best, b.max) - get_component(best, b.min);
        double c = get_component(axis, b.max) - get_component(axis, b.min);
        if (total_cmp(a, c) <= 0) best = axis;
    }
    return best;
}
// impl Hittable for &[Surface]::bounding_box, surface.rs:54-58
inline AABB slice_bbox(const Surface* s, size_t n) {
    AABB acc = AABB_EMPTY;
    for (size_t i = 0; i < n; ++i) acc = aabb_merge(acc, geometry_bbox(s[i].geometry));
    return acc;
}
// partition.rs:42-56: double-ended find / rfind swap loop, then partition_point
template <class Pred>
size_t partition_in_place(Surface* s, size_t n, Pred pred) {
    size_t lo = 0, hi = n;
    for (;;) {
        size_t left = n;
        while (lo < hi) {
            size_t e = lo++;
            if (!pred(s[e])) { left = e; break; }
        }
        if (left == n) break;
        size_t right = n;
        while (lo < hi) {
            size_t e = --hi;
            if (pred(s[e])) { right = e; break; }
        }
        if (right == n) break;
        std::swap(s[left], s[right]);
    }
    // slice::partition_point: binary search for the first element failing pred
    size_t a = 0, b = n;
    while (a < b) {
        size_t mid = a + (b - a) / 2;
        if (pred(s[mid])) a = mid + 1; else b = mid;
    }
    return a;
}
// partition.rs:58-68.  sort_unstable_by leaves the order of equal keys
// unspecified; a stable sort is used here (identical whenever keys differ).
size_t longest_axis_bisect_slice(Surface* s, size_t n) {
    AABB bb = slice_bbox(s, n);
    int axis = longest_axis(bb);
    std::stable_sort(s, s + n, [axis](const Surface& a, const Surface& b) {
        return total_cmp(get_component(axis, geometry_bbox(a.geometry).min), get_component(axis, geometry_bbox(b.geometry).min)) < 0;
    });
    return n / 2;
}
// partition.rs:70-78
size_t longest_axis_midpoint(Surface* s, size_t n) {
    AABB bb = slice_bbox(s, n);
    int axis = longest_axis(bb);
    double midpoint = get_component(axis, aabb_centroid(bb));
    return partition_in_place(s, n, [axis, midpoint](const Surface& e) {
        return get_component(axis, aabb_centroid(geometry_bbox(e.geometry))) < midpoint;
    });
}
// partition.rs:87-90
inline double surface_area_factor(const AABB& b) {
    V3 dims = aabb_dimensions(b);
    return dims.x * dims.y + dims.x * dims.z + dims.y * dims.z;
}
// partition.rs:92-106
inline double surface_area_heuristic(const AABB& left, size_t n_left, const AABB& right, size_t n_right, const AABB& bb) {
    double parent_saf = surface_area_factor(bb);
    double p_left = surface_area_factor(left) / parent_saf;
    double p_right = surface_area_factor(right) / parent_saf;
    const double ROOT_TEST_COST = 1.0;
    return ROOT_TEST_COST + p_left * (double)n_left + p_right * (double)n_right;
}
struct SplitVolumes {   // partition.rs:108-116
    size_t n_left;
    AABB left;
    size_t n_right;
    AABB right;
    Interval interval;
};
struct SplitsCache {
    double min, max;
    std::vector<SplitVolumes> splits;
};
// partition.rs:119-171
SplitsCache splits_cache(const Surface* s, size_t n, int axis) {
    std::vector<AABB> sorted_boxes(n);
    for (size_t i = 0; i < n; ++i) sorted_boxes[i] = geometry_bbox(s[i].geometry);
    // ties between equal centroids never influence a selected split (a split
    // is only looked up across a strict gap between distinct centroid values)
    std::stable_sort(sorted_boxes.begin(), sorted_boxes.end(), [axis](const AABB& a, const AABB& b) {
        return total_cmp(get_component(axis, aabb_centroid(a)), get_component(axis, aabb_centroid(b))) < 0;
    });
    std::vector<AABB> prefix(n - 1), suffix(n - 1);
    AABB state = AABB_EMPTY;
    for (size_t i = 0; i + 1 < n; ++i) {
        state = aabb_merge(state, sorted_boxes[i]);
        prefix[i] = state;
    }
    state = AABB_EMPTY;
    for (size_t i = n - 1; i >= 1; --i) {
        state = aabb_merge(state, sorted_boxes[i]);
        suffix[i - 1] = state;
    }
    SplitsCache out;
    out.splits.resize(n - 1);
    for (size_t i = 0; i + 1 < n; ++i) {
        out.splits[i] = SplitVolumes{i + 1, prefix[i], n - i - 1, suffix[i],
                                     Interval{get_component(axis, aabb_centroid(sorted_boxes[i])),
                                              get_component(axis, aabb_centroid(sorted_boxes[i + 1]))}};
    }
    out.min = out.splits.front().interval.min;
    out.max = out.splits.back().interval.max;
    return out;
}
struct Plane {
    int axis;
    double split;
};
// partition.rs:173-224.  Returns false for "No valid splitting plane" (:219).
bool sah_partition_impl(Surface* s, size_t n, const std::vector<Plane>& planes, size_t& n_left_out) {
    SplitsCache caches[3] = {splits_cache(s, n, 0), splits_cache(s, n, 1), splits_cache(s, n, 2)};
    AABB bounding_box = slice_bbox(s, n);
    bool have = false;
    int best_axis = 0;
    double best_split = 0.0, best_cost = 0.0;
    for (const Plane& pl : planes) {
        const SplitsCache& c = caches[pl.axis];
        if (pl.split < c.min || pl.split >= c.max) continue;   // :189-192
        // :194 partition_point(|split| intercept >= split.interval.max)
        size_t a = 0, b = c.splits.size();
        while (a < b) {
            size_t mid = a + (b - a) / 2;
            if (pl.split >= c.splits[mid].interval.max) a = mid + 1; else b = mid;
        }
        const SplitVolumes& sv = c.splits[a];
        double cost = surface_area_heuristic(sv.left, sv.n_left, sv.right, sv.n_right, bounding_box);
        // min_by keeps the FIRST minimum (:218)
        if (!have || total_cmp(best_cost, cost) > 0) {
            have = true;
            best_axis = pl.axis;
            best_split = pl.split;
            best_cost = cost;
        }
    }
    if (!have) {
        g_err = "No valid splitting plane";
        return false;
    }
    n_left_out = partition_in_place(s, n, [best_axis, best_split](const Surface& e) {
        return get_component(best_axis, aabb_centroid(geometry_bbox(e.geometry))) <= best_split;
    });
    return true;
}
// partition.rs:229-248 (axis-major candidate order)
bool sah_equal_size(Surface* s, size_t n, uint32_t buckets, size_t& n_left) {
    AABB bb = slice_bbox(s, n);
    std::vector<Plane> planes;
    for (int axis = 0; axis < 3; ++axis) {
        double start = get_component(axis, bb.min);
        double step = get_component(axis, aabb_dimensions(bb)) / (double)buckets;
        for (uint32_t i = 1; i < buckets; ++i) planes.push_back(Plane{axis, start + ((double)i * step)});
    }
    return sah_partition_impl(s, n, planes, n_left);
}
// partition.rs:254-266 (surface-major, axis-minor candidate order)
bool sah_per_surface(Surface* s, size_t n, size_t& n_left) {
    std::vector<Plane> planes;
    planes.reserve(3 * n);
    for (size_t i = 0; i < n; ++i) {
        V3 centroid = aabb_centroid(geometry_bbox(s[i].geometry));
        for (int axis = 0; axis < 3; ++axis) planes.push_back(Plane{axis, get_component(axis, centroid)});
    }
    return sah_partition_impl(s, n, planes, n_left);
}
// PartitionBy::partition, src/bvh/mod.rs:36-49
bool partition(int strategy, uint32_t buckets, Surface* s, size_t n, size_t& n_left) {
    switch (strategy) {
        case ORC_BISECT: n_left = longest_axis_bisect_slice(s, n); return true;
        case ORC_MIDPOINT: n_left = longest_axis_midpoint(s, n); return true;
        case ORC_SAH_EQUAL_SIZE: return sah_equal_size(s, n, buckets, n_left);
        case ORC_SAH_PER_SURFACE: return sah_per_surface(s, n, n_left);
        default: g_err = "unknown partition strategy"; return false;
    }
}
inline AABB node_bbox(const Node& nd) { return nd.box; }   // src/bvh/mod.rs:59-69

// src/bvh/mod.rs:92-131
bool build_tree_rec(int strategy, uint32_t buckets, std::vector<Node>& nodes, Surface* s, size_t n) {
    if (n == 1) {
        nodes.push_back(Node{true, NONE, geometry_bbox(s[0].geometry), s[0]});
    } else if (n == 2) {
        size_t n_left;
        if (!partition(strategy, buckets, s, n, n_left)) return false;
        if (n_left != 1) {   // the reference indexes left[0] / right[0] and would panic
            g_err = "partition of two surfaces left one side empty";
            return false;
        }
        Surface left = s[0], right = s[1];
        uint32_t here = (uint32_t)nodes.size();
        nodes.push_back(Node{false, here + 2, aabb_merge(geometry_bbox(left.geometry), geometry_bbox(right.geometry)), Surface{}});
        nodes.push_back(Node{true, NONE, geometry_bbox(left.geometry), left});
        nodes.push_back(Node{true, NONE, geometry_bbox(right.geometry), right});
    } else {
        size_t n_left;
        if (!partition(strategy, buckets, s, n, n_left)) return false;
        if (n_left == 0 || n_left == n) {   // the reference recurses without bound here
            g_err = "partition left one side empty";
            return false;
        }
        size_t parent_idx = nodes.size();
        nodes.push_back(Node{false, NONE, AABB_EMPTY, Surface{}});   // Placeholder
        if (!build_tree_rec(strategy, buckets, nodes, s, n_left)) return false;
        size_t right_idx = nodes.size();
        if (!build_tree_rec(strategy, buckets, nodes, s + n_left, n - n_left)) return false;
        nodes[parent_idx].right = (uint32_t)right_idx;
        nodes[parent_idx].box = aabb_merge(node_bbox(nodes[parent_idx + 1]), node_bbox(nodes[right_idx]));
    }
    return true;
}

struct Counters {
    uint64_t rays = 0, paths = 0, node_visits = 0, sphere_tests = 0, planar_tests = 0;
};

// impl Hittable for BVH::hit, src/bvh/mod.rs:134-174
inline bool bvh_hit(const orc_scene& sc, const Ray& ray, const Interval& ray_t, Hit& out_hit, const Surface*& out_surface, Counters& ctr) {
    ctr.rays++;
    if (sc.tree.empty()) return false;   // the reference would index out of bounds; an empty world never hits
    uint32_t stack[256];
    std::vector<uint32_t> big;   // only used if the fixed stack overflows
    int sp = 0;
    stack[sp++] = 0;
    bool have = false;
    Interval shrunken = ray_t;
    const size_t len = sc.tree.size();
    for (;;) {
        uint32_t i;
        if (!big.empty()) { i = big.back(); big.pop_back(); }
        else if (sp > 0) i = stack[--sp];
        else break;
        const Node& curr = sc.tree[i];
        ctr.node_visits++;
        if (!aabb_hit(node_bbox(curr), ray, shrunken)) continue;
        if (!curr.leaf) {
            auto push = [&](uint32_t v) {
                if (big.empty() && sp < 256) stack[sp++] = v; else big.push_back(v);
            };
            push(curr.right);
            if ((size_t)i + 1 < len) push(i + 1);
        } else {
            if (curr.surface.geometry.kind == ORC_SPHERE) ctr.sphere_tests++; else ctr.planar_tests++;
            Hit h;
            if (geometry_hit(curr.surface.geometry, ray, shrunken, h)) {
                if (have && h.t > out_hit.t) {
                    // keep acc
                } else {
                    shrunken.max = h.t;
                    out_hit = h;
                    out_surface = &curr.surface;
                    have = true;
                }
            }
        }
    }
    return have;
}

// impl Hittable for &[Surface]::hit, src/surface.rs:40-53
inline bool list_hit(const orc_scene& sc, const Ray& ray, const Interval& ray_t, Hit& out_hit, const Surface*& out_surface) {
    bool have = false;
    for (const Surface& e : sc.surfaces) {
        Hit h;
        if (!geometry_hit(e.geometry, ray, ray_t, h)) continue;
        if (!have || h.t < out_hit.t) {
            out_hit = h;
            out_surface = &e;
            have = true;
        }
    }
    return have;
}

// ---------------------------------------------------------------- Camera
inline V3 arr(const double a[3]) { return v3(a[0], a[1], a[2]); }
inline void put(double a[3], V3 v) { a[0] = v.x; a[1] = v.y; a[2] = v.z; }
inline double to_radians(double deg) { return deg * (PI / 180.0); }   // f64::to_radians
inline uint32_t f64_as_u32(double x) {   // Rust `as u32`: saturating, NaN -> 0
    if (!(x > 0.0)) return 0;
    if (x >= 4294967295.0) return 4294967295u;
    return (uint32_t)x;
}
inline uint8_t f64_as_u8(double x) {
    if (!(x > 0.0)) return 0;
    if (x >= 255.0) return 255;
    return (uint8_t)x;
}

// src/camera.rs:180-182 + 164-177
inline Ray get_ray(const orc_icamera& c, uint32_t col, uint32_t row, PathRng& rng) {
    rng.begin_event();
    double ox = rng.next() - 0.5;
    double oy = rng.next() - 0.5;
    V3 pixel_sample = arr(c.pixel00_loc) + (((double)col + ox) * arr(c.pixel_du)) + (((double)row + oy) * arr(c.pixel_dv));
    V3 origin;
    if (c.defocus_angle <= 0.0) {
        origin = arr(c.center);
    } else {
        V3 p = random_in_unit_disk(rng);
        origin = arr(c.center) + (p.x * arr(c.defocus_disk_u)) + (p.y * arr(c.defocus_disk_v));
    }
    return Ray{origin, pixel_sample - origin};
}

// src/camera.rs:184-208 (recursive, as written)
V3 ray_color(const Ray& ray, const orc_scene& world, uint32_t remaining, V3 background, PathRng& rng, Counters& ctr) {
    if (remaining == 0) return v3(0.0, 0.0, 0.0);
    Hit hit;
    const Surface* surf = nullptr;
    if (bvh_hit(world, ray, Interval{0.001, INF}, hit, surf, ctr)) {
        V3 emitted = material_emitted(surf->material, hit);
        Ray scattered_ray;
        V3 attenuation;
        if (material_scatter(surf->material, ray, hit, rng, scattered_ray, attenuation)) {
            V3 scattered = ray_color(scattered_ray, world, remaining - 1, background, rng, ctr) * attenuation;
            return emitted + scattered;
        }
        return emitted;
    }
    return background;
}

// The same function with the recursion unrolled into a loop that carries the
// product of attenuations forward.  `emitted + rec * attenuation` nests the
// products from the far end of the path, this form from the near end, so the
// two differ only in the association of the f64 products (a few ulp).  The
// device kernels use this form; it lets them be compared bit for bit.
V3 ray_color_iterative(Ray ray, const orc_scene& world, uint32_t max_depth, V3 background, PathRng& rng, Counters& ctr) {
    V3 beta = v3(1.0, 1.0, 1.0);
    V3 radiance = v3(0.0, 0.0, 0.0);
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        Hit hit;
        const Surface* surf = nullptr;
        if (!bvh_hit(world, ray, Interval{0.001, INF}, hit, surf, ctr)) {
            radiance = radiance + beta * background;
            break;
        }
        if (surf->material.kind == ORC_DIFFUSE_LIGHT || surf->material.kind == ORC_UV_GRADIENT)
            radiance = radiance + beta * material_emitted(surf->material, hit);
        Ray scattered_ray;
        V3 attenuation;
        if (!material_scatter(surf->material, ray, hit, rng, scattered_ray, attenuation)) break;
        beta = beta * attenuation;
        ray = scattered_ray;
    }
    return radiance;
}

}  // namespace

// ================================================================ C API
extern "C" {

const char* orc_last_error(void) { return g_err.c_str(); }

orc_scene* orc_scene_create(const orc_surface* surfaces, uint32_t n, int strategy, uint32_t buckets) {
    orc_scene* sc = new orc_scene();
    sc->surfaces.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (!surface_from_desc(surfaces[i], i, sc->surfaces[i])) {
            delete sc;
            return nullptr;
        }
    }
    if (n > 0) {   // src/bvh/mod.rs:76-90
        std::vector<Surface> work = sc->surfaces;
        sc->tree.reserve(2 * (size_t)n);
        if (!build_tree_rec(strategy, buckets, sc->tree, work.data(), work.size())) {
            delete sc;
            return nullptr;
        }
    }
    return sc;
}
void orc_scene_destroy(orc_scene* sc) { delete sc; }

uint32_t orc_scene_n_nodes(const orc_scene* sc) { return (uint32_t)sc->tree.size(); }

void orc_scene_nodes(const orc_scene* sc, double* bmin, double* bmax, uint32_t* right, uint32_t* leaf_surface) {
    size_t n = sc->tree.size();
    for (size_t i = 0; i < n; ++i) {
        const Node& nd = sc->tree[i];
        bmin[0 * n + i] = nd.box.min.x; bmin[1 * n + i] = nd.box.min.y; bmin[2 * n + i] = nd.box.min.z;
        bmax[0 * n + i] = nd.box.max.x; bmax[1 * n + i] = nd.box.max.y; bmax[2 * n + i] = nd.box.max.z;
        right[i] = nd.leaf ? NONE : nd.right;
        leaf_surface[i] = nd.leaf ? nd.surface.id : NONE;
    }
}

int orc_surface_bbox(const orc_surface* s, double out[6]) {
    Surface sf;
    if (!surface_from_desc(*s, 0, sf)) return -1;
    AABB b = geometry_bbox(sf.geometry);
    out[0] = b.min.x; out[1] = b.min.y; out[2] = b.min.z;
    out[3] = b.max.x; out[4] = b.max.y; out[5] = b.max.z;
    return 0;
}

double orc_sah_cost(const double l[6], uint64_t n_left, const double r[6], uint64_t n_right, const double p[6]) {
    AABB L{v3(l[0], l[1], l[2]), v3(l[3], l[4], l[5])};
    AABB R{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])};
    AABB P{v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5])};
    return surface_area_heuristic(L, (size_t)n_left, R, (size_t)n_right, P);
}

int orc_aabb_hit(const double box[6], const double org[3], const double dir[3], double tmin, double tmax) {
    AABB b{v3(box[0], box[1], box[2]), v3(box[3], box[4], box[5])};
    return aabb_hit(b, Ray{arr(org), arr(dir)}, Interval{tmin, tmax}) ? 1 : 0;
}

int orc_surface_hit(const orc_surface* s, const double org[3], const double dir[3], double tmin, double tmax, double out[10]) {
    Surface sf;
    if (!surface_from_desc(*s, 0, sf)) return -1;
    Hit h;
    if (!geometry_hit(sf.geometry, Ray{arr(org), arr(dir)}, Interval{tmin, tmax}, h)) return 0;
    out[0] = h.t; out[1] = h.p.x; out[2] = h.p.y; out[3] = h.p.z; out[4] = h.alpha; out[5] = h.beta;
    out[6] = h.front_face ? 1.0 : 0.0; out[7] = h.face_normal.x; out[8] = h.face_normal.y; out[9] = h.face_normal.z;
    return 1;
}

void orc_trace_closest(const orc_scene* sc, const double* org, const double* dir, uint64_t n, double tmin, double tmax,
                       uint32_t* out_id, double* out_t, orc_counters* ctr) {
    Counters c;
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray{arr(org + 3 * i), arr(dir + 3 * i)};
        Hit h;
        const Surface* s = nullptr;
        if (bvh_hit(*sc, ray, Interval{tmin, tmax}, h, s, c)) {
            out_id[i] = s->id;
            out_t[i] = h.t;
        } else {
            out_id[i] = NONE;
            out_t[i] = INF;
        }
    }
    if (ctr) {
        ctr->rays += c.rays; ctr->node_visits += c.node_visits;
        ctr->sphere_tests += c.sphere_tests; ctr->planar_tests += c.planar_tests;
    }
}

void orc_trace_bruteforce(const orc_scene* sc, const double* org, const double* dir, uint64_t n, double tmin, double tmax,
                          uint32_t* out_id, double* out_t) {
    for (uint64_t i = 0; i < n; ++i) {
        Ray ray{arr(org + 3 * i), arr(dir + 3 * i)};
        Hit h;
        const Surface* s = nullptr;
        if (list_hit(*sc, ray, Interval{tmin, tmax}, h, s)) {
            out_id[i] = s->id;
            out_t[i] = h.t;
        } else {
            out_id[i] = NONE;
            out_t[i] = INF;
        }
    }
}

// src/camera.rs:64-113
void orc_camera_initialize(const orc_camera* cam, orc_icamera* out) {
    uint32_t image_height = f64_as_u32((double)cam->image_width / cam->aspect_ratio);
    if (image_height < 1) image_height = 1;
    double pixel_samples_scale = 1.0 / (double)cam->samples_per_pixel;
    V3 center = arr(cam->look_from);
    double theta = to_radians(cam->v_fov);
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * cam->focus_dist;
    double viewport_width = viewport_height * (double)cam->image_width / (double)image_height;
    V3 w = to_unit(arr(cam->look_from) - arr(cam->look_at));
    V3 u = to_unit(cross(arr(cam->v_up), w));
    V3 v = cross(w, u);
    V3 viewport_u = viewport_width * u;
    V3 viewport_v = viewport_height * -v;
    V3 pixel_du = viewport_u / (double)cam->image_width;
    V3 pixel_dv = viewport_v / (double)image_height;
    V3 viewport_upper_left = center - (cam->focus_dist * w) - viewport_u / 2.0 - viewport_v / 2.0;
    V3 pixel00_loc = viewport_upper_left + (pixel_du + pixel_dv) * 0.5;
    double defocus_radius = cam->focus_dist * std::tan(to_radians(cam->defocus_angle / 2.0));
    V3 defocus_disk_u = u * defocus_radius;
    V3 defocus_disk_v = v * defocus_radius;

    std::memset(out, 0, sizeof(*out));
    out->image_width = cam->image_width;
    out->image_height = image_height;
    out->samples_per_pixel = cam->samples_per_pixel;
    out->max_depth = cam->max_depth;
    out->defocus_angle = cam->defocus_angle;
    put(out->background, arr(cam->background));
    out->pixel_samples_scale = pixel_samples_scale;
    put(out->center, center);
    put(out->pixel00_loc, pixel00_loc);
    put(out->pixel_du, pixel_du);
    put(out->pixel_dv, pixel_dv);
    put(out->defocus_disk_u, defocus_disk_u);
    put(out->defocus_disk_v, defocus_disk_v);
}

void orc_primary_rays(const orc_icamera* cam, uint64_t seed, uint32_t sample, double* org, double* dir) {
    for (uint32_t row = 0; row < cam->image_height; ++row)
        for (uint32_t col = 0; col < cam->image_width; ++col) {
            uint32_t pix = row * cam->image_width + col;
            PathRng rng(seed, pix, sample);
            Ray r = get_ray(*cam, col, row, rng);
            put(org + 3 * (size_t)pix, r.origin);
            put(dir + 3 * (size_t)pix, r.direction);
        }
}

// src/camera.rs:133-151: rows x cols x samples; the reference parallelises
// with rayon, here rows are handed to std::threads from an atomic counter.
void orc_render(const orc_scene* sc, const orc_icamera* cam, uint64_t seed, uint32_t spp_begin, uint32_t spp_count,
                double* out_sum, int n_threads, int iterative, orc_counters* ctr) {
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    const uint32_t W = cam->image_width, H = cam->image_height;
    const V3 background = arr(cam->background);
    std::atomic<uint32_t> next_row(0);
    std::vector<Counters> per_thread((size_t)n_threads);
    auto work = [&](int tid) {
        Counters& c = per_thread[(size_t)tid];
        for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= H) break;
            for (uint32_t col = 0; col < W; ++col) {
                uint32_t pix = row * W + col;
                V3 acc = v3(0.0, 0.0, 0.0);
                for (uint32_t s = spp_begin; s < spp_begin + spp_count; ++s) {
                    PathRng rng(seed, pix, s);
                    Ray ray = get_ray(*cam, col, row, rng);
                    acc = acc + (iterative ? ray_color_iterative(ray, *sc, cam->max_depth, background, rng, c)
                                           : ray_color(ray, *sc, cam->max_depth, background, rng, c));
                    c.paths++;
                }
                put(out_sum + 3 * (size_t)pix, acc);
            }
        }
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (ctr) {
        for (const Counters& c : per_thread) {
            ctr->rays += c.rays; ctr->paths += c.paths; ctr->node_visits += c.node_visits;
            ctr->sphere_tests += c.sphere_tests; ctr->planar_tests += c.planar_tests;
        }
    }
}

// src/camera.rs:248-266
void orc_ppm_quantize(const double* rgb, uint64_t n_pixels, uint8_t* out) {
    for (uint64_t i = 0; i < 3 * n_pixels; ++i) {
        double lin = rgb[i];
        double g = lin > 0.0 ? std::sqrt(lin) : 0.0;
        out[i] = f64_as_u8(255.999 * clampd(g, 0.0, 0.999));
    }
}

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]) { philox4x32(ctr, key, rounds, out); }
int orc_philox_rounds(void) { return PHILOX_ROUNDS; }

void orc_rng_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out) {
    PathRng rng(seed, pixel, sample);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng.next();
}

void orc_scatter(const orc_surface* s, const double org[3], const double dir[3], const double hit10[10],
                 uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw, double out[14]) {
    Surface sf;
    std::memset(out, 0, 14 * sizeof(double));
    if (!surface_from_desc(*s, 0, sf)) return;
    Hit h;
    h.t = hit10[0]; h.p = v3(hit10[1], hit10[2], hit10[3]); h.alpha = hit10[4]; h.beta = hit10[5];
    h.front_face = hit10[6] != 0.0; h.face_normal = v3(hit10[7], hit10[8], hit10[9]);
    Ray ray{arr(org), arr(dir)};
    PathRng rng(seed, pixel, sample, draw);
    V3 emitted = material_emitted(sf.material, h);
    Ray sr{v3(0, 0, 0), v3(0, 0, 0)};
    V3 att = v3(0, 0, 0);
    bool ok = material_scatter(sf.material, ray, h, rng, sr, att);
    out[0] = ok ? 1.0 : 0.0;
    if (ok) {
        out[1] = sr.origin.x; out[2] = sr.origin.y; out[3] = sr.origin.z;
        out[4] = sr.direction.x; out[5] = sr.direction.y; out[6] = sr.direction.z;
        out[7] = att.x; out[8] = att.y; out[9] = att.z;
    }
    out[10] = emitted.x; out[11] = emitted.y; out[12] = emitted.z;
    out[13] = (double)rng.draw;
}

}  // extern "C"
