// Layout study (debug aid, host only): how many 128-byte lines does the closest-hit walk of a large tree pull per ray under
// different orders of the 32-byte node records in memory?
//
// The walk is the product's: pre-order tree, left child first, successor link on a miss or after a leaf, interval shrinking
// at every accepted hit.  The cache model is what the 10^6-surface scenes see on the GPU: the top `top_depth` levels of the
// tree are always cached (every ray passes them), and beyond that a lane finds a line again only if it is among the last K
// lines it touched itself (the L2 holds about one to three lines per resident lane).
//
//   g++ -O2 -shared -fPIC -o /tmp/liblayout_sim.so tools/layout_sim.cpp ; driven by tools/layout_sim.py
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {
constexpr uint32_t NONE = 0xFFFFFFFFu;
struct V3 { double x, y, z; };
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Surface { uint32_t kind; V3 q, u, v; double radius; };

bool box_hit(const double* bmin, const double* bmax, uint32_t n, uint32_t i, V3 o, V3 d, double tmin, double tmax) {
    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    double lo = NAN, hi = NAN;
    for (int a = 0; a < 3; ++a) {
        const double t0 = (bmin[(size_t)a * n + i] - oo[a]) / dd[a], t1 = (bmax[(size_t)a * n + i] - oo[a]) / dd[a];
        double l = std::fmin(t0, t1), u = std::fmax(t0, t1);
        l = l < tmin ? tmin : (l > tmax ? tmax : l);
        u = u < tmin ? tmin : (u > tmax ? tmax : u);
        lo = std::fmax(lo, l);
        hi = std::fmin(hi, u);
    }
    return lo < hi;
}
bool surface_hit(const Surface& s, V3 o, V3 d, double tmin, double tmax, double& t) {
    if (s.kind == 0) {
        const V3 oc = s.q - o;
        const double a = dot(d, d), h = dot(d, oc), c = dot(oc, oc) - s.radius * s.radius, disc = h * h - a * c;
        if (disc < 0) return false;
        const double sq = std::sqrt(disc);
        double root = (h - sq) / a;
        if (!(tmin < root && root < tmax)) {
            root = (h + sq) / a;
            if (!(tmin < root && root < tmax)) return false;
        }
        t = root;
        return true;
    }
    const V3 n = cross(s.u, s.v);
    const V3 norm = n * (1.0 / std::sqrt(dot(n, n)));
    const double D = dot(norm, s.q);
    const V3 w = n * (1.0 / dot(n, n));
    const double den = dot(norm, d);
    if (std::fabs(den) < 1e-10) return false;
    const double tt = (D - dot(norm, o)) / den;
    if (!(tmin <= tt && tt <= tmax)) return false;
    const V3 p = o + d * tt, qp = p - s.q;
    const double al = dot(w, cross(qp, s.v)), be = dot(w, cross(s.u, qp));
    if (s.kind == 1) {
        if (!(0 <= al && al <= 1 && 0 <= be && be <= 1)) return false;
    } else if (!(al >= 0 && be >= 0 && al + be <= 1)) return false;
    t = tt;
    return true;
}
}  // namespace

extern "C" {

// Visit sequences: for each ray the nodes visited, concatenated; offsets[n_rays + 1].  Returns the total (call with seq = null to size).
uint64_t sim_visits(uint32_t n, const double* bmin, const double* bmax, const uint32_t* right, const uint32_t* leaf, const Surface* surfaces,
                    uint64_t n_rays, const double* org, const double* dir, uint32_t* seq, uint64_t* offsets, uint64_t capacity) {
    // successor links of the pre-order array
    std::vector<uint32_t> miss(n, NONE);
    if (n) miss[0] = n;
    for (uint32_t i = 0; i < n; ++i)
        if (leaf[i] == NONE) {
            miss[i + 1] = right[i];
            miss[right[i]] = miss[i];
        }
    uint64_t total = 0;
    for (uint64_t r = 0; r < n_rays; ++r) {
        const V3 o{org[3 * r], org[3 * r + 1], org[3 * r + 2]}, d{dir[3 * r], dir[3 * r + 1], dir[3 * r + 2]};
        double tmax = INFINITY;
        const double tmin = 0.001;
        bool have = false;
        if (offsets) offsets[r] = total;
        uint32_t i = 0;
        while (i < n) {
            if (seq && total < capacity) seq[total] = i;
            ++total;
            if (!box_hit(bmin, bmax, n, i, o, d, tmin, tmax)) {
                i = miss[i];
            } else if (leaf[i] == NONE) {
                i = i + 1;
            } else {
                double t;
                if (surface_hit(surfaces[leaf[i]], o, d, tmin, tmax, t) && !(have && t > tmax)) {
                    tmax = t;
                    have = true;
                }
                i = miss[i];
            }
        }
    }
    if (offsets) offsets[n_rays] = total;
    return total;
}

// Line fetches of the recorded walks when node i lives in record pos[i] (32-byte records, `recs_per_line` per line).
// depth[i] < top_depth: always cached.  Otherwise a hit only if the line is among the lane's last K distinct lines.
uint64_t sim_fetches(uint64_t n_rays, const uint32_t* seq, const uint64_t* offsets, const uint32_t* pos, const uint8_t* depth,
                     uint32_t top_depth, uint32_t K, uint32_t recs_per_line) {
    uint64_t fetches = 0;
    std::vector<uint32_t> lru(K);
    for (uint64_t r = 0; r < n_rays; ++r) {
        std::fill(lru.begin(), lru.end(), NONE);
        for (uint64_t k = offsets[r]; k < offsets[r + 1]; ++k) {
            const uint32_t node = seq[k];
            if (depth[node] < top_depth) continue;
            const uint32_t line = pos[node] / recs_per_line;
            uint32_t j = 0;
            while (j < K && lru[j] != line) ++j;
            if (j == K) {
                ++fetches;
                j = K - 1;
            }
            for (; j > 0; --j) lru[j] = lru[j - 1];
            lru[0] = line;
        }
    }
    return fetches;
}

}  // extern "C"
