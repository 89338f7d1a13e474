// raytracing.hpp -- host side of the renderer: the scene, camera and BVH
// builder API of seridescent/raytracing, kept on the CPU, with rendering
// handed to the MI355X backend through the C ABI in include/rtx.h.
//
// The reference host is Rust; this header mirrors its public interface in
// C++ (same type names, fields, argument meaning and failure cases) so a
// program written against the reference maps over line by line:
//
//   Vector3, Interval, AABB           src/vector.rs, src/interval.rs, src/aabb.rs
//   Geometry, Material, Surface       src/geometry.rs, src/material.rs, src/surface.rs
//   PartitionBy, SAHBucketStrategy    src/bvh/mod.rs:13-34
//   BVH::from_slice                   src/bvh/mod.rs:76-131 + src/bvh/partition.rs
//   Camera, InitializedCamera         src/camera.rs:11-113
//   RenderRunner::run                 src/runner.rs:8-41
//
// What is NOT here is the per-ray code (AABB::hit, Geometry::hit,
// Material::scatter, ray_color): that is the device path.
//
// Compile with -ffp-contract=off: the SAH builder's choice between equal-cost
// planes depends on unfused f64 rounding (src/bvh/partition.rs:92-106).
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rtx.h"

namespace raytracing {

// ------------------------------------------------------------------ vector.rs
struct Vector3 {
    double x = 0.0, y = 0.0, z = 0.0;
    constexpr Vector3() = default;
    constexpr Vector3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    double length_squared() const { return x * x + y * y + z * z; }
    double length() const { return std::sqrt(length_squared()); }
    double operator[](int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
    Vector3 to_unit() const;
};
inline Vector3 operator+(Vector3 a, Vector3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vector3 operator-(Vector3 a, Vector3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vector3 operator-(Vector3 a) { return {-a.x, -a.y, -a.z}; }
inline Vector3 operator*(Vector3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vector3 operator*(double s, Vector3 a) { return a * s; }
inline Vector3 operator*(Vector3 a, Vector3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// `Vector3 / f64` multiplies by the reciprocal (src/vector.rs:179-190)
inline Vector3 operator/(Vector3 a, double s) {
    const double inv = 1.0 / s;
    return {a.x * inv, a.y * inv, a.z * inv};
}
inline Vector3 Vector3::to_unit() const { return *this / length(); }
inline double dot(Vector3 a, Vector3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3 cross(Vector3 a, Vector3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// f64::total_cmp as a strict-weak "less" plus three-way form
inline int total_cmp(double a, double b) {
    int64_t ia, ib;
    std::memcpy(&ia, &a, sizeof ia);
    std::memcpy(&ib, &b, sizeof ib);
    ia ^= (int64_t)((uint64_t)(ia >> 63) >> 1);
    ib ^= (int64_t)((uint64_t)(ib >> 63) >> 1);
    return ia < ib ? -1 : (ia > ib ? 1 : 0);
}

// ------------------------------------------------------------------ interval.rs
struct Interval {
    double min = std::numeric_limits<double>::infinity();
    double max = -std::numeric_limits<double>::infinity();
};

// ------------------------------------------------------------------ aabb.rs
class AABB {
   public:
    static AABB EMPTY() {
        const double inf = std::numeric_limits<double>::infinity();
        AABB b;
        b.min_ = {inf, inf, inf};
        b.max_ = {-inf, -inf, -inf};
        return b;
    }
    // AABB::new sorts the two corners per axis (src/aabb.rs:33-38); f64::min/max ignore NaN
    static AABB from_corners(Vector3 a, Vector3 b) {
        AABB r;
        r.min_ = {std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
        r.max_ = {std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
        return r;
    }
    static AABB merge(const AABB& a, const AABB& b) {   // src/aabb.rs:40-53
        AABB r;
        r.min_ = {std::fmin(a.min_.x, b.min_.x), std::fmin(a.min_.y, b.min_.y), std::fmin(a.min_.z, b.min_.z)};
        r.max_ = {std::fmax(a.max_.x, b.max_.x), std::fmax(a.max_.y, b.max_.y), std::fmax(a.max_.z, b.max_.z)};
        return r;
    }
    Vector3 min() const { return min_; }
    Vector3 max() const { return max_; }
    Vector3 dimensions() const { return max_ - min_; }
    Vector3 centroid() const { return min_ + (dimensions() / 2.0); }   // src/aabb.rs:29-31
    // grows each side by max(extent, padding), not by `padding` (src/aabb.rs:78-87)
    AABB padded(double padding) const {
        const Vector3 dims = dimensions();
        const Vector3 pad{std::fmax(dims.x, padding), std::fmax(dims.y, padding), std::fmax(dims.z, padding)};
        return from_corners(min_ - pad, max_ + pad);
    }
    bool operator==(const AABB& o) const {
        return min_.x == o.min_.x && min_.y == o.min_.y && min_.z == o.min_.z && max_.x == o.max_.x && max_.y == o.max_.y && max_.z == o.max_.z;
    }

   private:
    Vector3 min_, max_;
};

// ------------------------------------------------------------------ geometry.rs
struct ConstructSphereError : std::invalid_argument {   // src/geometry.rs:49-53
    explicit ConstructSphereError(double radius)
        : std::invalid_argument("invalid radius " + std::to_string(radius) + " (expected non-negative radius)") {}
};

struct Geometry {
    enum Kind : uint32_t { Sphere = RTX_SPHERE, Quadrilateral = RTX_QUAD, Triangle = RTX_TRIANGLE };
    Kind kind = Sphere;
    Vector3 center;   // Sphere
    double radius = 0.0;
    Vector3 q, u, v, norm, w;   // Quadrilateral / Triangle
    double d = 0.0;

    static Geometry sphere(Vector3 center, double radius) {   // src/geometry.rs:56-62
        if (radius < 0.0) throw ConstructSphereError(radius);
        Geometry g;
        g.kind = Sphere;
        g.center = center;
        g.radius = radius;
        return g;
    }
    static Geometry quadrilateral(Vector3 q, Vector3 u, Vector3 v) { return planar(Quadrilateral, q, u, v); }   // :64-75
    static Geometry triangle(Vector3 q, Vector3 u, Vector3 v) { return planar(Triangle, q, u, v); }             // :77-88

    AABB bounding_box() const {   // src/geometry.rs:112-133
        switch (kind) {
            case Sphere: {        // :252-255
                const Vector3 radii{radius, radius, radius};
                return AABB::from_corners(center + radii, center - radii);
            }
            case Quadrilateral:   // :291-293 (two opposite corners only)
                return AABB::from_corners(q, q + u + v).padded(0.0001);
            default:              // :329-331
                return AABB::merge(AABB::from_corners(q, q + u), AABB::from_corners(q, q + v)).padded(0.0001);
        }
    }

   private:
    static Geometry planar(Kind kind, Vector3 q, Vector3 u, Vector3 v) {
        const Vector3 n = cross(u, v);
        Geometry g;
        g.kind = kind;
        g.q = q;
        g.u = u;
        g.v = v;
        g.norm = n.to_unit();
        g.d = dot(g.norm, q);
        g.w = n / dot(n, n);
        return g;
    }
};

// ------------------------------------------------------------------ material.rs
struct Material {
    enum Kind : uint32_t {
        KLambertian = RTX_LAMBERTIAN, KMetal = RTX_METAL, KDielectric = RTX_DIELECTRIC,
        KDiffuseLight = RTX_DIFFUSE_LIGHT, KUVGradient = RTX_UV_GRADIENT
    };
    Kind kind = KLambertian;
    Vector3 rgb;          // albedo | emit
    double param = 0.0;   // fuzz_radius | refraction_index | intensity

    static Material Lambertian(Vector3 albedo) { return {KLambertian, albedo, 0.0}; }
    static Material Metal(Vector3 albedo, double fuzz_radius) { return {KMetal, albedo, fuzz_radius}; }
    static Material Dielectric(double refraction_index) { return {KDielectric, {}, refraction_index}; }
    static Material DiffuseLight(Vector3 emit) { return {KDiffuseLight, emit, 0.0}; }
    static Material UVGradient(double intensity) { return {KUVGradient, {}, intensity}; }
};

// ------------------------------------------------------------------ surface.rs
struct Surface {
    Geometry geometry;
    Material material;
    uint32_t user_id = 0;   // position in the caller's array; reported by trace_closest
    Surface() = default;
    Surface(Geometry g, Material m) : geometry(g), material(m) {}
    AABB bounding_box() const { return geometry.bounding_box(); }
};

// impl Hittable for &[Surface]::bounding_box (src/surface.rs:54-58)
inline AABB bounding_box(const Surface* s, size_t n) {
    AABB acc = AABB::EMPTY();
    for (size_t i = 0; i < n; ++i) acc = AABB::merge(acc, s[i].bounding_box());
    return acc;
}

// ------------------------------------------------------------------ bvh/mod.rs
struct SAHBucketStrategy {
    enum Kind { EqualSize, PerSurface } kind = PerSurface;
    uint32_t buckets = 0;
    static SAHBucketStrategy equal_size(uint32_t n) { return {EqualSize, n}; }
    static SAHBucketStrategy per_surface() { return {PerSurface, 0}; }
};
struct PartitionBy {
    enum Kind { LongestAxisBisectSlice, LongestAxisMidpoint, SurfaceAreaHeuristic } kind = SurfaceAreaHeuristic;
    SAHBucketStrategy sah = SAHBucketStrategy::per_surface();
    static PartitionBy bisect() { return {LongestAxisBisectSlice, {}}; }
    static PartitionBy midpoint() { return {LongestAxisMidpoint, {}}; }
    static PartitionBy surface_area_heuristic(SAHBucketStrategy s) { return {SurfaceAreaHeuristic, s}; }
};

struct BuildError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

namespace partition {   // src/bvh/partition.rs

inline int longest_axis(const AABB& bb) {   // :28-40, Iterator::max_by returns the last maximum
    const Vector3 dims[2] = {bb.min(), bb.max()};
    int pick = 0;
    for (int axis = 1; axis < 3; ++axis) {
        const double cur = dims[1][pick] - dims[0][pick];
        const double cand = dims[1][axis] - dims[0][axis];
        if (total_cmp(cur, cand) != 1) pick = axis;
    }
    return pick;
}

// :42-56.  The reference walks one double-ended iterator: `find` consumes
// from the front up to and including the first element failing `pred`,
// `rfind` consumes from the back up to and including the last element
// passing it, and the two are swapped.  The order this leaves inside each
// side decides the leaf order of the tree, so it is reproduced exactly.
template <class Pred>
size_t partition_in_place(Surface* s, size_t n, Pred pred) {
    size_t front = 0, back = n;
    while (true) {
        while (front < back && pred(s[front])) ++front;
        if (front == back) break;
        const size_t left = front++;
        while (front < back && !pred(s[back - 1])) --back;
        if (front == back) break;
        const size_t right = --back;
        std::swap(s[left], s[right]);
    }
    // partition_point(pred)
    size_t lo = 0, hi = n;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (pred(s[mid])) lo = mid + 1; else hi = mid;
    }
    return lo;
}

inline size_t longest_axis_bisect_slice(Surface* s, size_t n) {   // :58-68
    const int axis = longest_axis(bounding_box(s, n));
    // sort_unstable_by: order of equal keys is unspecified in the reference
    std::stable_sort(s, s + n, [axis](const Surface& a, const Surface& b) {
        return total_cmp(a.bounding_box().min()[axis], b.bounding_box().min()[axis]) < 0;
    });
    return n / 2;
}

inline size_t longest_axis_midpoint(Surface* s, size_t n) {   // :70-78
    const AABB bb = bounding_box(s, n);
    const int axis = longest_axis(bb);
    const double midpoint = bb.centroid()[axis];
    return partition_in_place(s, n, [=](const Surface& e) { return e.bounding_box().centroid()[axis] < midpoint; });
}

namespace sah {

inline double surface_area_factor(const AABB& b) {   // :87-90
    const Vector3 dims = b.dimensions();
    return dims.x * dims.y + dims.x * dims.z + dims.y * dims.z;
}
// :92-106; evaluated left to right, never fused
inline double surface_area_heuristic(const AABB& left, size_t n_left, const AABB& right, size_t n_right, const AABB& bb) {
    const double parent_saf = surface_area_factor(bb);
    const double p_left = surface_area_factor(left) / parent_saf;
    const double p_right = surface_area_factor(right) / parent_saf;
    return 1.0 + p_left * (double)n_left + p_right * (double)n_right;
}

// splits_cache, :119-171: for the boxes sorted by centroid along one axis,
// the box of the first k and of the remaining n-k, for k = 1..n-1.
struct AxisSplits {
    std::vector<double> key;      // sorted centroid coordinate
    std::vector<AABB> left_of;    // left_of[k-1]  = box of sorted[0..k)
    std::vector<AABB> right_of;   // right_of[k-1] = box of sorted[k..n)
    AxisSplits(const Surface* s, size_t n, int axis) {
        std::vector<std::pair<double, AABB>> items;
        items.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const AABB b = s[i].bounding_box();
            items.emplace_back(b.centroid()[axis], b);
        }
        std::stable_sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return total_cmp(a.first, b.first) < 0; });
        key.resize(n);
        for (size_t i = 0; i < n; ++i) key[i] = items[i].first;
        left_of.resize(n - 1);
        right_of.resize(n - 1);
        AABB run = AABB::EMPTY();
        for (size_t k = 1; k < n; ++k) {
            run = AABB::merge(run, items[k - 1].second);
            left_of[k - 1] = run;
        }
        run = AABB::EMPTY();
        for (size_t k = n - 1; k >= 1; --k) {
            run = AABB::merge(run, items[k].second);
            right_of[k - 1] = run;
        }
    }
    // split_at, :182-197: the number of surfaces on the left of plane
    // `intercept`, or 0 when the plane does not separate anything.
    size_t n_left_for(double intercept) const {
        const size_t n = key.size();
        if (intercept < key.front() || intercept >= key.back()) return 0;
        // first split whose interval.max (= key[k]) is above the plane
        size_t lo = 0, hi = n - 1;
        while (lo < hi) {
            const size_t mid = lo + (hi - lo) / 2;
            if (intercept >= key[mid + 1]) lo = mid + 1; else hi = mid;
        }
        return lo + 1;
    }
};

struct Plane {
    int axis;
    double at;
};

inline size_t partition_impl(Surface* s, size_t n, const std::vector<Plane>& planes) {   // :173-224
    const AxisSplits per_axis[3] = {AxisSplits(s, n, 0), AxisSplits(s, n, 1), AxisSplits(s, n, 2)};
    const AABB bb = bounding_box(s, n);
    const Plane* chosen = nullptr;
    double chosen_cost = 0.0;
    for (const Plane& p : planes) {
        const AxisSplits& ax = per_axis[p.axis];
        const size_t k = ax.n_left_for(p.at);
        if (k == 0) continue;
        const double cost = surface_area_heuristic(ax.left_of[k - 1], k, ax.right_of[k - 1], n - k, bb);
        // Iterator::min_by keeps the first of equal minima
        if (!chosen || total_cmp(cost, chosen_cost) < 0) {
            chosen = &p;
            chosen_cost = cost;
        }
    }
    if (!chosen) throw BuildError("No valid splitting plane");   // :219
    const int axis = chosen->axis;
    const double at = chosen->at;
    return partition_in_place(s, n, [=](const Surface& e) { return e.bounding_box().centroid()[axis] <= at; });
}

inline size_t equal_size(Surface* s, size_t n, uint32_t buckets) {   // :229-248, axis-major plane order
    const AABB bb = bounding_box(s, n);
    std::vector<Plane> planes;
    for (int axis = 0; axis < 3; ++axis) {
        const double start = bb.min()[axis];
        const double step = bb.dimensions()[axis] / (double)buckets;
        for (uint32_t i = 1; i < buckets; ++i) planes.push_back({axis, start + ((double)i * step)});
    }
    return partition_impl(s, n, planes);
}

inline size_t per_surface(Surface* s, size_t n) {   // :254-266, surface-major plane order
    std::vector<Plane> planes;
    planes.reserve(3 * n);
    for (size_t i = 0; i < n; ++i) {
        const Vector3 c = s[i].bounding_box().centroid();
        for (int axis = 0; axis < 3; ++axis) planes.push_back({axis, c[axis]});
    }
    return partition_impl(s, n, planes);
}

}  // namespace sah
}  // namespace partition

// The built tree, already in the layout the backend takes (rtx_scene_desc):
// the reference's pre-order `Box<[Node]>` with each enum field in its own table.
class BVH {
   public:
    // src/bvh/mod.rs:76-90.  Surfaces are numbered by their position in `surfaces`.
    static BVH from_slice(std::vector<Surface> surfaces, const PartitionBy& partition_by) {
        BVH bvh;
        for (size_t i = 0; i < surfaces.size(); ++i) surfaces[i].user_id = (uint32_t)i;
        if (!surfaces.empty()) {
            bvh.nodes_.reserve(2 * surfaces.size());
            bvh.build_tree_rec(partition_by, surfaces.data(), surfaces.size());
        }
        bvh.flatten();
        return bvh;
    }

    // The same tree with the partitioning done by the device builder
    // (rtx_bvh_build): the leaf boxes are computed here, the device returns
    // the leaf order and the node array.
    static BVH from_slice_on_device(std::vector<Surface> surfaces, const PartitionBy& partition_by, rtx_build_stats* stats = nullptr) {
        BVH bvh;
        const size_t m = surfaces.size();
        for (size_t i = 0; i < m; ++i) surfaces[i].user_id = (uint32_t)i;
        if (m > 0) {
            std::vector<double> boxes(6 * m);
            for (size_t i = 0; i < m; ++i) {
                const AABB b = surfaces[i].bounding_box();
                for (int a = 0; a < 3; ++a) {
                    boxes[6 * i + a] = b.min()[a];
                    boxes[6 * i + 3 + a] = b.max()[a];
                }
            }
            const size_t n = 2 * m - 1;
            std::vector<uint32_t> leaf_order(m), right(n), prim(n);
            std::vector<double> bmin(3 * n), bmax(3 * n);
            int strategy = RTX_BISECT;
            if (partition_by.kind == PartitionBy::LongestAxisMidpoint) strategy = RTX_MIDPOINT;
            else if (partition_by.kind == PartitionBy::SurfaceAreaHeuristic)
                strategy = partition_by.sah.kind == SAHBucketStrategy::EqualSize ? RTX_SAH_EQUAL_SIZE : RTX_SAH_PER_SURFACE;
            const int rc = rtx_bvh_build(boxes.data(), (uint32_t)m, strategy, partition_by.sah.buckets, leaf_order.data(), bmin.data(),
                                         bmax.data(), right.data(), prim.data(), stats);
            if (rc != RTX_OK) throw BuildError(rtx_last_error());
            bvh.leaves_.reserve(m);
            for (size_t i = 0; i < m; ++i) bvh.leaves_.push_back(surfaces[leaf_order[i]]);
            bvh.nodes_.resize(n);
            for (size_t i = 0; i < n; ++i) {
                Node& nd = bvh.nodes_[i];
                nd.leaf = prim[i] != RTX_NONE;
                nd.right = right[i];
                nd.leaf_slot = prim[i];
                nd.box = AABB::from_corners(Vector3{bmin[i], bmin[n + i], bmin[2 * n + i]}, Vector3{bmax[i], bmax[n + i], bmax[2 * n + i]});
            }
        }
        bvh.flatten();
        return bvh;
    }

    size_t n_nodes() const { return nodes_.size(); }
    size_t n_leaves() const { return leaves_.size(); }
    AABB bounding_box() const { return nodes_.empty() ? AABB::EMPTY() : nodes_[0].box; }   // :176-182

    // Borrowed view for rtx_scene_create; valid while this BVH lives.
    rtx_scene_desc desc() const {
        rtx_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.n_nodes = (uint32_t)nodes_.size();
        d.n_prims = (uint32_t)leaves_.size();
        d.n_materials = (uint32_t)leaves_.size();
        d.node_bmin = bmin_.data(); d.node_bmax = bmax_.data();
        d.node_right = right_.data(); d.node_prim = prim_.data();
        d.prim_kind = kind_.data();
        d.prim_q = q_.data(); d.prim_u = u_.data(); d.prim_v = v_.data(); d.prim_norm = norm_.data(); d.prim_w = w_.data();
        d.prim_d = d_.data(); d.prim_radius = radius_.data();
        d.prim_material = material_.data(); d.prim_user_id = user_id_.data();
        d.mat_kind = mat_kind_.data(); d.mat_rgb = mat_rgb_.data(); d.mat_param = mat_param_.data();
        return d;
    }

    // node table accessors (tests, tools)
    const std::vector<double>& node_bmin() const { return bmin_; }
    const std::vector<double>& node_bmax() const { return bmax_; }
    const std::vector<uint32_t>& node_right() const { return right_; }
    const std::vector<uint32_t>& node_prim() const { return prim_; }
    const std::vector<uint32_t>& prim_user_id() const { return user_id_; }

   private:
    struct Node {   // enum Node, src/bvh/mod.rs:51-57
        bool leaf;
        uint32_t right;   // Internal(Some(right_idx), _)
        AABB box;         // Internal: merged box; Leaf: surface.bounding_box() (Node::bounding_box, :59-69)
        uint32_t leaf_slot;
    };
    std::vector<Node> nodes_;
    std::vector<Surface> leaves_;   // in pre-order leaf sequence

    uint32_t push_leaf(const Surface& s) {
        leaves_.push_back(s);
        nodes_.push_back(Node{true, RTX_NONE, s.bounding_box(), (uint32_t)(leaves_.size() - 1)});
        return (uint32_t)(nodes_.size() - 1);
    }
    static size_t partition(const PartitionBy& by, Surface* s, size_t n) {   // PartitionBy::partition, :36-49
        switch (by.kind) {
            case PartitionBy::LongestAxisBisectSlice: return partition::longest_axis_bisect_slice(s, n);
            case PartitionBy::LongestAxisMidpoint: return partition::longest_axis_midpoint(s, n);
            default:
                return by.sah.kind == SAHBucketStrategy::EqualSize ? partition::sah::equal_size(s, n, by.sah.buckets)
                                                                   : partition::sah::per_surface(s, n);
        }
    }
    void build_tree_rec(const PartitionBy& by, Surface* s, size_t n) {   // :92-131
        if (n == 1) {
            push_leaf(s[0]);
            return;
        }
        const size_t n_left = partition(by, s, n);
        // The reference indexes an empty side (n == 2) or recurses on it
        // forever (n > 2); both are reported as a build failure here.
        if (n_left == 0 || n_left == n) throw BuildError("partition left one side empty");
        const size_t parent = nodes_.size();
        nodes_.push_back(Node{false, RTX_NONE, AABB::EMPTY(), RTX_NONE});   // Node::Placeholder
        build_tree_rec(by, s, n_left);
        const size_t right_idx = nodes_.size();
        build_tree_rec(by, s + n_left, n - n_left);
        nodes_[parent].right = (uint32_t)right_idx;
        nodes_[parent].box = AABB::merge(nodes_[parent + 1].box, nodes_[right_idx].box);
    }

    std::vector<double> bmin_, bmax_, q_, u_, v_, norm_, w_, d_, radius_, mat_rgb_, mat_param_;
    std::vector<uint32_t> right_, prim_, material_, user_id_, mat_kind_;
    std::vector<uint8_t> kind_;

    void flatten() {
        const size_t n = nodes_.size(), m = leaves_.size();
        bmin_.assign(3 * n, 0.0); bmax_.assign(3 * n, 0.0);
        right_.assign(n, RTX_NONE); prim_.assign(n, RTX_NONE);
        for (size_t i = 0; i < n; ++i) {
            const Node& nd = nodes_[i];
            for (int a = 0; a < 3; ++a) {
                bmin_[a * n + i] = nd.box.min()[a];
                bmax_[a * n + i] = nd.box.max()[a];
            }
            if (nd.leaf) prim_[i] = nd.leaf_slot; else right_[i] = nd.right;
        }
        kind_.assign(m, 0);
        q_.assign(3 * m, 0.0); u_.assign(3 * m, 0.0); v_.assign(3 * m, 0.0); norm_.assign(3 * m, 0.0); w_.assign(3 * m, 0.0);
        d_.assign(m, 0.0); radius_.assign(m, 0.0);
        material_.assign(m, 0); user_id_.assign(m, 0);
        mat_kind_.assign(m, 0); mat_rgb_.assign(3 * m, 0.0); mat_param_.assign(m, 0.0);
        for (size_t i = 0; i < m; ++i) {
            const Geometry& g = leaves_[i].geometry;
            const Material& mt = leaves_[i].material;
            kind_[i] = (uint8_t)g.kind;
            const Vector3 q = g.kind == Geometry::Sphere ? g.center : g.q;
            for (int a = 0; a < 3; ++a) {
                q_[a * m + i] = q[a];
                u_[a * m + i] = g.u[a]; v_[a * m + i] = g.v[a]; norm_[a * m + i] = g.norm[a]; w_[a * m + i] = g.w[a];
                mat_rgb_[a * m + i] = mt.rgb[a];
            }
            d_[i] = g.d;
            radius_[i] = g.radius;
            material_[i] = (uint32_t)i;   // a Surface owns its Material (src/surface.rs:15-18)
            user_id_[i] = leaves_[i].user_id;
            mat_kind_[i] = (uint32_t)mt.kind;
            mat_param_[i] = mt.param;
        }
    }
};

// ------------------------------------------------------------------ camera.rs
inline double to_radians(double deg) { return deg * (3.14159265358979323846264338327950288 / 180.0); }

struct InitializedCamera : rtx_camera {};   // src/camera.rs:28-43

struct Camera {   // src/camera.rs:11-26, defaults :45-61
    double aspect_ratio = 1.0;
    uint32_t image_width = 100;
    uint32_t samples_per_pixel = 10;
    uint32_t max_depth = 10;
    double v_fov = 90.0;
    Vector3 look_from{0.0, 0.0, 0.0};
    Vector3 look_at{0.0, 0.0, -1.0};
    Vector3 v_up{0.0, 1.0, 0.0};
    double defocus_angle = 0.0;
    double focus_dist = 10.0;
    Vector3 background{0.0, 0.0, 0.0};

    InitializedCamera initialize() const {   // src/camera.rs:64-113
        auto as_u32 = [](double x) -> uint32_t {   // `as u32` saturates and truncates
            if (!(x > 0.0)) return 0u;
            if (x >= 4294967295.0) return 4294967295u;
            return (uint32_t)x;
        };
        uint32_t image_height = as_u32((double)image_width / aspect_ratio);
        if (image_height < 1) image_height = 1;
        const double pixel_samples_scale = 1.0 / (double)samples_per_pixel;
        const Vector3 center = look_from;
        const double theta = to_radians(v_fov);
        const double h = std::tan(theta / 2.0);
        const double viewport_height = 2.0 * h * focus_dist;
        const double viewport_width = viewport_height * (double)image_width / (double)image_height;
        const Vector3 w = (look_from - look_at).to_unit();
        const Vector3 u = cross(v_up, w).to_unit();
        const Vector3 v = cross(w, u);
        const Vector3 viewport_u = viewport_width * u;
        const Vector3 viewport_v = viewport_height * -v;
        const Vector3 pixel_du = viewport_u / (double)image_width;
        const Vector3 pixel_dv = viewport_v / (double)image_height;
        const Vector3 viewport_upper_left = center - (focus_dist * w) - viewport_u / 2.0 - viewport_v / 2.0;
        const Vector3 pixel00_loc = viewport_upper_left + (pixel_du + pixel_dv) * 0.5;
        const double defocus_radius = focus_dist * std::tan(to_radians(defocus_angle / 2.0));
        const Vector3 defocus_disk_u = u * defocus_radius;
        const Vector3 defocus_disk_v = v * defocus_radius;

        InitializedCamera c;
        std::memset(&c, 0, sizeof c);
        c.image_width = image_width;
        c.image_height = image_height;
        c.samples_per_pixel = samples_per_pixel;
        c.max_depth = max_depth;
        c.defocus_angle = defocus_angle;
        c.pixel_samples_scale = pixel_samples_scale;
        auto put = [](double* dst, Vector3 s) { dst[0] = s.x; dst[1] = s.y; dst[2] = s.z; };
        put(c.background, background);
        put(c.center, center);
        put(c.pixel00_loc, pixel00_loc);
        put(c.pixel_du, pixel_du);
        put(c.pixel_dv, pixel_dv);
        put(c.defocus_disk_u, defocus_disk_u);
        put(c.defocus_disk_v, defocus_disk_v);
        return c;
    }
};

// ppm_pixel, src/camera.rs:248-266
inline void ppm_pixel(const double rgb[3], uint8_t out[3]) {
    for (int c = 0; c < 3; ++c) {
        const double lin = rgb[c];
        double g = lin > 0.0 ? std::sqrt(lin) : 0.0;
        if (g < 0.0) g = 0.0;
        if (g > 0.999) g = 0.999;
        const double v = 255.999 * g;
        out[c] = (uint8_t)(v > 0.0 ? (v >= 255.0 ? 255.0 : v) : 0.0);
    }
}

// ------------------------------------------------------------------ runner.rs
struct RenderError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct RenderRunner {   // src/runner.rs:8-20
    Camera camera;
    PartitionBy partition_strategy = PartitionBy::surface_area_heuristic(SAHBucketStrategy::per_surface());
    uint64_t seed = 0;   // the reference seeds from the OS; a fixed seed makes a render repeatable
    bool build_on_device = false;   // BVH::from_slice_on_device instead of the host builder
    int n_gpus = 1;                 // > 1: samples per pixel split over that many GPUs of this node, one RCCL reduce (rtx_multi_*)
    bool through_multi = false;     // take the rtx_multi_* path even with one GPU (a world of one: communicator, reduce and scaling still run)
    bool keep_image = true;         // false: run() returns an empty vector and the linear image never leaves the device (the reference returns nothing;
                                    // at 1920x1080 the copy and its 50 MB of host memory are a tenth of a cold process)

    // src/runner.rs:23-41: build the BVH, render, write the P3 image to
    // `image_out` (the reference prints to stdout) and the three timers to
    // `log_out` (stderr there).  Returns the linear image.
    std::vector<double> run(std::vector<Surface> surfaces, FILE* image_out = stdout, FILE* log_out = stderr) const {
        using clock = std::chrono::steady_clock;
        const auto start_time = clock::now();
        const BVH world = build_on_device ? BVH::from_slice_on_device(std::move(surfaces), partition_strategy)
                                          : BVH::from_slice(std::move(surfaces), partition_strategy);
        const auto bvh_done = clock::now();

        const InitializedCamera cam = camera.initialize();
        const rtx_scene_desc desc = world.desc();
        std::vector<double> image(keep_image ? 3 * (size_t)cam.image_width * cam.image_height : 0);
        double* const image_ptr = keep_image ? image.data() : nullptr;
        std::vector<char> text(image_out ? rtx_ppm_text_bound(cam.image_width, cam.image_height) : 0);   // src/camera.rs:129-131,155-161
        uint64_t text_len = 0;
        rtx_stats stats{};
        if (n_gpus > 1 || through_multi) {
            // the scene on every device and the communicator once, then one frame: one host thread per device, one reduce onto device 0
            rtx_multi* multi = nullptr;
            if (rtx_multi_create(&desc, n_gpus, &multi) != RTX_OK) throw RenderError(rtx_last_error());
            rtx_multi_stats ms{};
            int rc = rtx_multi_render(multi, &cam, seed, 0, image_ptr, &ms);
            if (rc == RTX_OK && image_out)
                rc = rtx_ppm_text(rtx_multi_image(multi), cam.image_width, cam.image_height, text.data(), text.size(), &text_len, nullptr);
            const std::string msg = rc != RTX_OK ? rtx_last_error() : "";
            rtx_multi_destroy(multi);
            if (rc != RTX_OK) throw RenderError(msg);
            stats.rays = ms.rays;
            stats.paths = ms.paths;
            stats.ms_total = ms.ms_frame;
        } else {
            rtx_scene* scene = nullptr;
            if (rtx_scene_create(&desc, &scene) != RTX_OK) throw RenderError(rtx_last_error());
            if (!image_ptr && !image_out) throw RenderError("RenderRunner::run: neither an image nor its text was asked for");
            const int rc = rtx_render_ppm(scene, &cam, seed, 0, image_ptr, image_out ? text.data() : nullptr, text.size(), &text_len, &stats);
            const std::string msg = rc != RTX_OK ? rtx_last_error() : "";
            rtx_scene_destroy(scene);
            if (rc != RTX_OK) throw RenderError(msg);
        }
        if (image_out) std::fwrite(text.data(), 1, text_len, image_out);
        const auto done = clock::now();
        if (log_out) {
            auto secs = [](clock::duration d) { return std::chrono::duration<double>(d).count(); };
            std::fprintf(log_out, "\n\nDone!\nTotal runtime: %.6fs\nBVH construction: %.6fs\nRendering: %.6fs\n", secs(done - start_time),
                         secs(bvh_done - start_time), secs(done - bvh_done));
            std::fprintf(log_out, "rays: %llu  paths: %llu  device: %.3f ms  (%.1f Mrays/s)\n", (unsigned long long)stats.rays,
                         (unsigned long long)stats.paths, stats.ms_total, stats.ms_total > 0 ? stats.rays / stats.ms_total / 1e3 : 0.0);
        }
        return image;
    }
};

}  // namespace raytracing
