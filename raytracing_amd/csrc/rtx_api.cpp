// rtx_api.cpp -- the C ABI declared in include/rtx.h, on top of the kernels.
//
// Host-side responsibilities: validate and upload the flattened BVH, own the
// path pool, drive the traverse/shade rounds, and report counters and times.
#include "rtx_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace rtx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

}  // namespace

namespace rtx {

int api_fail(int code, const std::string& msg) { return fail(code, msg); }

bool cold_log_on() {
    static const bool on = env_u32("RTX_LOG_COLD", 0, true) != 0u;
    return on;
}

uint32_t env_u32(const char* name, uint32_t dflt, bool zero_ok) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 10);
    if (end == v || (x == 0 && !zero_ok) || x > 0xFFFFFFFFull) return dflt;
    return (uint32_t)x;
}

int PathPool::ensure_ring() {
    if (!h_ctl_ring) HIP_TRY(hipHostMalloc((void**)&h_ctl_ring, RING * sizeof(DRenderCtl), hipHostMallocDefault));
    if (!h_params_ring) HIP_TRY(hipHostMalloc((void**)&h_params_ring, RING * sizeof(DRenderParams), hipHostMallocDefault));
    for (hipEvent_t& e : ring_ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return RTX_OK;
}
int PathPool::ring_acquire(uint32_t* slot) {
    int rc = ensure_ring();
    if (rc != RTX_OK) return rc;
    *slot = ring_next++ % RING;
    if (ring_used[*slot]) HIP_TRY(hipEventSynchronize(ring_ev[*slot]));
    ring_used[*slot] = true;
    return RTX_OK;
}

PathPool::~PathPool() {
    for (hipEvent_t e : ring_ev) if (e) (void)hipEventDestroy(e);
    if (h_ctl_ring) (void)hipHostFree(h_ctl_ring);
    if (h_params_ring) (void)hipHostFree(h_params_ring);
    for (hipEvent_t e : kernel_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_poll) if (e) (void)hipEventDestroy(e);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (h_poll) (void)hipHostFree(h_poll);
    if (stream) (void)hipStreamDestroy(stream);
}
int PathPool::ensure_host_side() {
    if (!stream) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (hipEvent_t& e : ev_poll) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!ev_done) HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
    if (!h_poll) HIP_TRY(hipHostMalloc((void**)&h_poll, 2 * sizeof(uint32_t), hipHostMallocDefault));
    return RTX_OK;
}
int PathPool::ensure_counters(uint32_t n_sub) {
    if (n_sub <= counters_n) return RTX_OK;
    HIP_TRY(counters.resize(3 * (size_t)n_sub));
    counters_n = n_sub;
    return RTX_OK;
}
int PathPool::ensure_ctl() {
    HIP_TRY(ctl.resize(8));   // the records behind the first are scratch for instrumented debug builds
    return RTX_OK;
}
int PathPool::ensure_capacity(size_t n) {
    int rc = ensure_ctl();
    if (rc != RTX_OK) return rc;
    if (n <= capacity) return RTX_OK;
    ColdTimer cold("path pool allocation");
    // Both new buffers are allocated before either old one is let go: if one allocation fails the pool is
    // exactly as it was (RTX_ERR_OOM is recoverable), never half replaced with `capacity` describing freed memory.
    // The old buffers are released first only when they are all that stands in the way (second attempt).
    DevBuf<double> nf;
    DevBuf<uint32_t> nu;
    hipError_t e = nf.resize(19 * n);
    if (e == hipSuccess) e = nu.resize(9 * n);
    if (e == hipErrorOutOfMemory && capacity) {
        (void)hipGetLastError();
        nf.release();
        nu.release();
        buf_f64.release();
        buf_u32.release();
        capacity = 0;
        state[0] = state[1] = DState{};
        hits = DHits{};
        e = nf.resize(19 * n);
        if (e == hipSuccess) e = nu.resize(9 * n);
    }
    HIP_TRY(e);
    buf_f64.swap(nf);
    buf_u32.swap(nu);
    double* f = buf_f64.p;
    uint32_t* u = buf_u32.p;
    for (int b = 0; b < 2; ++b) {
        DState& st = state[b];
        st.ox = f; st.oy = f + n; st.oz = f + 2 * n; st.dx = f + 3 * n; st.dy = f + 4 * n; st.dz = f + 5 * n;
        st.bx = f + 6 * n; st.by = f + 7 * n; st.bz = f + 8 * n;
        f += 9 * n;
        st.pixel = u; st.sample = u + n; st.draw = u + 2 * n; st.depth = u + 3 * n;
        u += 4 * n;
    }
    hits.t = f;
    hits.prim = u;
    capacity = n;
    return RTX_OK;
}

}  // namespace rtx

void rtx::pool_geometry(uint64_t need, uint32_t max_waves, uint32_t* n_sub, uint64_t* sub_capacity) {
    uint64_t waves = (need + 63) / 64;
    waves = (waves + 3) & ~3ull;
    if (waves < 4) waves = 4;
    if (waves > max_waves) waves = max_waves;
    *n_sub = (uint32_t)waves;
    *sub_capacity = (((need + waves - 1) / waves) + 63) & ~63ull;
}

// 0: tree in global memory; 1: f32 node records in LDS; 2: node records and surface tables in LDS
static uint32_t lds_mode_for(const rtx_scene* sc, uint32_t flags) {
    const uint32_t lds_limit = env_u32("RTX_LDS_SCENE_BYTES", 32u * 1024u);   // five blocks of the kernel per CU
    const size_t tree = node_tables_lds_bytes(sc->d.n_nodes);
    const size_t surf = (size_t)sc->d.n_spheres * sizeof(DSphere) + (size_t)sc->d.n_planars * sizeof(DPlanar);
    if ((flags & RTX_FLAG_NO_LDS) || tree > lds_limit) return 0u;
    return tree + surf <= lds_limit ? 2u : 1u;
}

// Which closest-hit kernel a call uses: the reference's division form when the
// caller asks for it, else the tiered test with the f32 records in LDS (small
// trees), from global memory (large trees), or the f64-only filter (scenes
// whose box planes are too large for f32, or on request).
void rtx::run_traverse(const rtx_scene* sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                       uint32_t flags, DRenderCtl* ctl, hipStream_t s) {
    const bool count_work = counting(flags);
    if (flags & RTX_FLAG_EXACT_DIVIDE) {   // (the f64 forms read the reference's boxes only)
        launch_traverse(sc->d, rays, hits, pool, tmin, tmax, ctl, count_work, true, s);
    } else if ((flags & RTX_FLAG_F64_FILTER) || !sc->f32_ok) {
        launch_traverse(sc->d, rays, hits, pool, tmin, tmax, ctl, count_work, false, s);
    } else {
        const uint32_t mode = lds_mode_for(sc, flags);
        launch_traverse_f32(scene_tables(sc, flags), rays, hits, pool, tmin, tmax, ctl, count_work, mode, s);
    }
}

// The cull boxes of a validated scene description (see the comment where rtx_scene_create calls this): cull[i] is the box the
// walk's first tier tests at node i.  Children follow their parents in pre-order, so one backward pass sees every child before
// its parent.  Returns the number of nodes whose cull box differs from the reference's.  Host arithmetic only.
uint32_t rtx::compute_cull_boxes(const rtx_scene_desc& desc, bool shrink, std::vector<DBox64>& cull) {
    const uint32_t n = desc.n_nodes, m = desc.n_prims;
    cull.assign(n, DBox64{});
    uint32_t n_cull = 0;
    for (uint32_t i = n; i-- > 0u;) {
        DBox64& c = cull[i];
        double rmin[3], rmax[3];   // the reference's box
        for (int a = 0; a < 3; ++a) {
            c.bmin[a] = rmin[a] = desc.node_bmin[(size_t)a * n + i];
            c.bmax[a] = rmax[a] = desc.node_bmax[(size_t)a * n + i];
        }
        if (!shrink) continue;
        const uint32_t prim = desc.node_prim[i];
        double lo[3], hi[3];
        bool ok = true;
        if (prim != RTX_NONE) {
            if (desc.prim_kind[prim] == RTX_SPHERE) continue;
            const bool quad = desc.prim_kind[prim] == RTX_QUAD;
            for (int a = 0; a < 3 && ok; ++a) {
                const double q = desc.prim_q[(size_t)a * m + prim], u = desc.prim_u[(size_t)a * m + prim], v = desc.prim_v[(size_t)a * m + prim];
                const double p0 = q, p1 = q + u, p2 = q + v, p3 = quad ? q + u + v : p0;
                const double tlo = std::fmin(std::fmin(p0, p1), std::fmin(p2, p3)), thi = std::fmax(std::fmax(p0, p1), std::fmax(p2, p3));
                const double big = std::fmax(std::fabs(rmin[a]), std::fabs(rmax[a]));
                const double need = std::fmax(1e-9 * big, 1e-6 * (std::fabs(u) + std::fabs(v))) + 1e-300;
                const double grow = std::fmax(need, 0x1.0p-13 * big);
                lo[a] = std::fmax(tlo - grow, rmin[a]);
                hi[a] = std::fmin(thi + grow, rmax[a]);
                ok = tlo - lo[a] >= need && hi[a] - thi >= need;   // (false for a NaN anywhere)
            }
        } else {
            const DBox64 &l = cull[i + 1], &r = cull[desc.node_right[i]];
            for (int a = 0; a < 3 && ok; ++a) {
                lo[a] = std::fmin(l.bmin[a], r.bmin[a]);
                hi[a] = std::fmax(l.bmax[a], r.bmax[a]);
                ok = lo[a] >= rmin[a] && hi[a] <= rmax[a];
            }
        }
        if (!ok) continue;
        bool differs = false;
        for (int a = 0; a < 3; ++a) {
            differs = differs || lo[a] != rmin[a] || hi[a] != rmax[a];
            c.bmin[a] = lo[a];
            c.bmax[a] = hi[a];
        }
        if (differs) ++n_cull;
    }
    return n_cull;
}

extern "C" {

const char* rtx_last_error(void) { return g_err.c_str(); }

int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rtx_scene_create(const rtx_scene_desc* desc, rtx_scene** out) {
    if (!desc || !out) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null argument");
    *out = nullptr;
    const uint32_t n = desc->n_nodes, m = desc->n_prims, k = desc->n_materials;
    if (n >= (1u << 27)) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: more than 2^27 - 1 nodes");
    if (n && (!desc->node_bmin || !desc->node_bmax || !desc->node_right || !desc->node_prim))
        return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null node table");
    if (m && (!desc->prim_kind || !desc->prim_q || !desc->prim_material || !desc->prim_user_id))
        return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null prim table");
    if (k && (!desc->mat_kind || !desc->mat_rgb || !desc->mat_param))
        return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null material table");

    // prims -> per-kind tables
    std::vector<DSphere> spheres;
    std::vector<DPlanar> planars;
    std::vector<DPrimInfo> prims(m);
    for (uint32_t i = 0; i < m; ++i) {
        uint32_t kind = desc->prim_kind[i];
        if (kind > RTX_TRIANGLE) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: unknown geometry kind");
        if (desc->prim_material[i] >= k) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: prim_material out of range");
        prims[i].kind = kind;
        prims[i].material = desc->prim_material[i];
        prims[i].user_id = desc->prim_user_id[i];
        if (kind == RTX_SPHERE) {
            if (!desc->prim_radius) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null prim_radius");
            DSphere s;
            for (int a = 0; a < 3; ++a) s.c[a] = desc->prim_q[(size_t)a * m + i];
            s.r = desc->prim_radius[i];
            prims[i].slot = (uint32_t)spheres.size();
            spheres.push_back(s);
        } else {
            if (!desc->prim_u || !desc->prim_v || !desc->prim_norm || !desc->prim_w || !desc->prim_d)
                return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: null planar table");
            DPlanar g;
            for (int a = 0; a < 3; ++a) {
                g.q[a] = desc->prim_q[(size_t)a * m + i];
                g.u[a] = desc->prim_u[(size_t)a * m + i];
                g.v[a] = desc->prim_v[(size_t)a * m + i];
                g.norm[a] = desc->prim_norm[(size_t)a * m + i];
                g.w[a] = desc->prim_w[(size_t)a * m + i];
            }
            g.d = desc->prim_d[i];
            prims[i].slot = (uint32_t)planars.size();
            planars.push_back(g);
        }
    }
    std::vector<DMaterial> mats(k);
    for (uint32_t i = 0; i < k; ++i) {
        if (desc->mat_kind[i] > RTX_UV_GRADIENT) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: unknown material kind");
        for (int a = 0; a < 3; ++a) mats[i].rgb[a] = desc->mat_rgb[(size_t)a * k + i];
        mats[i].param = desc->mat_param[i];
        mats[i].kind = desc->mat_kind[i];
        mats[i]._pad = 0;
    }

    // nodes: thread the pre-order array.  A parent precedes its children, so
    // one forward pass fills every successor: the left child (i+1) is
    // followed by the right child, the right child by the parent's successor.
    std::vector<DNode> nodes(n);
    std::vector<DNode32> nodes32(n);
    std::vector<uint32_t> leaf_codes(n, NONE);
    std::vector<DBox64> boxes64(n);
    if (spheres.size() > CODE_SLOT_MASK || planars.size() > CODE_SLOT_MASK)
        return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: more than 2^27 - 1 surfaces of one kind");
    const uint32_t n_spheres = (uint32_t)spheres.size();
    std::vector<uint32_t> code(m), rec_user_id(m);
    std::vector<DShadeRec> shade_recs(m);
    for (uint32_t i = 0; i < m; ++i) {
        const DMaterial& mt = mats[prims[i].material];
        const bool planar = prims[i].kind != RTX_SPHERE;
        code[i] = make_code(prims[i].kind, mt.kind, prims[i].slot);
        const uint32_t ri = planar ? n_spheres + prims[i].slot : prims[i].slot;
        DShadeRec& r = shade_recs[ri];
        if (planar) {
            for (int a = 0; a < 3; ++a) r.a[a] = planars[prims[i].slot].norm[a];
            r.s = 0.0;
        } else {
            for (int a = 0; a < 3; ++a) r.a[a] = spheres[prims[i].slot].c[a];
            r.s = 1.0 / spheres[prims[i].slot].r;
        }
        for (int a = 0; a < 3; ++a) r.rgb[a] = mt.rgb[a];
        r.param = mt.param;
        if (mt.kind == RTX_DIELECTRIC) {
            // A dielectric has no colour; its record carries what scatter would divide for on every bounce
            // (material.rs:111, 134-136): 1 / ri, and Schlick's r0^2 for eta = 1 / ri (front face) and eta = ri.
            const double eta_front = 1.0 / mt.param;
            r.rgb[0] = eta_front;
            r.rgb[1] = schlick_r0(eta_front);
            r.rgb[2] = schlick_r0(mt.param);
        }
        rec_user_id[ri] = prims[i].user_id;
    }
    bool f32_ok = true;
    std::vector<uint32_t> miss(n, NONE);
    bool filter_ok = true;
    uint32_t thin_nodes = 0;   // boxes with a side that single precision barely resolves (flat walls far from the origin)
    if (n) miss[0] = n;
    for (uint32_t i = 0; i < n; ++i) {
        DNode& nd = nodes[i];
        bool thin = false;
        for (int a = 0; a < 3; ++a) {
            nd.bmin[a] = desc->node_bmin[(size_t)a * n + i];
            nd.bmax[a] = desc->node_bmax[(size_t)a * n + i];
            if (nd.bmax[a] - nd.bmin[a] <= 0x1.0p-16 * std::fmax(std::fabs(nd.bmin[a]), std::fabs(nd.bmax[a]))) thin = true;
            if (!(std::fabs(nd.bmin[a]) <= 0x1.0p300) || !(std::fabs(nd.bmax[a]) <= 0x1.0p300)) filter_ok = false;
            if (!(std::fabs(nd.bmin[a]) <= 1e15) || !(std::fabs(nd.bmax[a]) <= 1e15)) f32_ok = false;
            boxes64[i].bmin[a] = nd.bmin[a];
            boxes64[i].bmax[a] = nd.bmax[a];
            nodes32[i].plane[2 * a] = (float)nd.bmin[a];
            nodes32[i].plane[2 * a + 1] = (float)nd.bmax[a];
        }
        if (thin) ++thin_nodes;
        if (miss[i] == NONE) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: node " + std::to_string(i) + " is not reachable in pre-order");
        if (miss[i] <= i || miss[i] > n) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: malformed tree");
        nd.miss = miss[i];
        uint32_t prim = desc->node_prim[i];
        uint32_t right = desc->node_right[i];
        if (prim != RTX_NONE) {   // Node::Leaf
            if (prim >= m) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: node_prim out of range");
            if (right != RTX_NONE) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: leaf with a right child");
            if (miss[i] != i + 1) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: leaf subtree is not a single node");
            nd.prim = code[prim];
            nd.kind = prims[prim].kind;
            nd.slot = prims[prim].slot;
            leaf_codes[i] = code[prim];
            nodes32[i].down = ((i + 1u) * 32u) | NODE32_LEAF;
        } else {                  // Node::Internal(Some(right_idx), box)
            if (right == RTX_NONE || right <= i + 1 || right >= miss[i])
                return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: right child index out of order at node " + std::to_string(i));
            nd.prim = NONE;
            nd.kind = 0;
            nd.slot = 0;
            nodes32[i].down = (i + 1u) * 32u;
            miss[i + 1] = right;
            miss[right] = miss[i];
        }
        nodes32[i].miss = nd.miss * 32u;
    }

    // ---- the boxes the walk's first tier tests ("cull boxes")
    // The reference pads the box of every quad and triangle by its own extent on each side (AABB::padded, src/aabb.rs:78-87):
    // a ray through such a box usually misses the surface, and the boxes of neighbours overlap many times over.  The walk may
    // cull with a smaller box C(node) as long as its outcome stays the reference's, which holds when, for every node,
    //   (a) C is inside the reference's box R componentwise.  The reference's f64 evaluation of its predicate is monotone in
    //       the box (RN is monotone, the quotients share the divisor, max/min/clamp are monotone), so "the predicate holds on C"
    //       implies "it holds on R": where the first tier proves a hit on C the reference passes the node too;
    //   (b) a ray for which the predicate fails on C, at the interval's current upper end, can have no accepted hit below the
    //       node at that or any smaller upper end.  Skipping the subtree then changes nothing the walk carries (best hit,
    //       interval), and the upper end only shrinks, so the rest of the walk is the reference's.
    // Leaves of two kinds satisfy (b).  "Margin" leaves: a quad or triangle whose C is the box of its true corners grown by
    // at least 1e-9 of the largest coordinate and 1e-6 of the surface's extent -- far more than the rounding of the surface
    // test, so an accepted hit point lies strictly inside C and the ray's slab intervals on C contain its t with room.
    // "Reference" leaves (spheres; planar surfaces whose reference box does not contain their corners with that room, which
    // the reference's two-corner quad boxes allow): C = R, and the reference itself tests R before the surface.  An internal
    // node takes the union of its children's C (if that lies inside its R, else R): a margin leaf below is covered by the
    // margin argument, a reference leaf below by monotonicity (its R = C lies inside the union, so the predicate failing on
    // the union implies it fails on that R, and the reference culls the leaf itself).
    // Visits the first tier cannot decide go to the f64 tiers with the reference's own box, as before.  Only the numbers of
    // box and surface tests change, so calls that count the reference's work (RTX_FLAG_COUNT_WORK) walk tables made from R.
    // C is kept thick enough for single precision to resolve (2^-13 of the largest coordinate per side, R permitting).
    std::vector<DBox64> cull;
    const uint32_t n_cull = compute_cull_boxes(*desc, env_u32("RTX_TIGHT_BOXES", 1, true) != 0u, cull);   // nodes whose cull box differs from the reference's
    // f32 records of the cull boxes; `nodes32` keeps the reference's planes for the calls that count its work
    std::vector<DNode32> nodes32_cull;
    if (n_cull) {
        nodes32_cull = nodes32;
        thin_nodes = 0;
        for (uint32_t i = 0; i < n; ++i) {
            bool thin = false;
            for (int a = 0; a < 3; ++a) {
                nodes32_cull[i].plane[2 * a] = (float)cull[i].bmin[a];
                nodes32_cull[i].plane[2 * a + 1] = (float)cull[i].bmax[a];
                if (cull[i].bmax[a] - cull[i].bmin[a] <= 0x1.0p-16 * std::fmax(std::fabs(cull[i].bmin[a]), std::fabs(cull[i].bmax[a]))) thin = true;
            }
            if (thin) ++thin_nodes;
        }
    }

    // Quantised half-size records for trees that are walked from global memory: a grid of 2^16 steps over the extent of
    // all boxes per axis, every plane rounded to the nearest step.  The walk's error bound rests on
    // |q_org + q * delta - plane| <= delta / 2 (1 + 1e-4): checked here plane by plane, and a scene that fails it
    // (extents too small against their offset for f64 to resolve the steps) keeps the f32 records.
    std::vector<DNode16> nodes16, nodes16_ref;   // of the cull boxes / of the reference's boxes (only where they differ)
    double q_org[3] = {0, 0, 0};
    float q_delta[3] = {0, 0, 0};
    bool q16_ok = f32_ok && filter_ok && n > 0;
    if (q16_ok) {
        nodes16.resize(n);
        if (n_cull) nodes16_ref.resize(n);
        for (int a = 0; a < 3 && q16_ok; ++a) {
            double lo = nodes[0].bmin[a], hi = nodes[0].bmax[a];
            for (uint32_t i = 1; i < n; ++i) {
                lo = std::fmin(lo, nodes[i].bmin[a]);
                hi = std::fmax(hi, nodes[i].bmax[a]);
            }
            q_org[a] = lo;
            q_delta[a] = (float)((hi - lo) / 65534.0);   // one step of slack for the rounding of delta to f32
            if (!(hi >= lo) || !std::isfinite(q_delta[a]) || (hi > lo && !(q_delta[a] >= 1e-30f))) q16_ok = false;
            const double dl = (double)q_delta[a];
            for (uint32_t i = 0; i < n && q16_ok; ++i) {
                const double planes[4] = {cull[i].bmin[a], cull[i].bmax[a], nodes[i].bmin[a], nodes[i].bmax[a]};
                for (int k = 0; k < (n_cull ? 4 : 2); ++k) {
                    const double q = dl > 0.0 ? std::nearbyint((planes[k] - lo) / dl) : 0.0;
                    if (!(q >= 0.0 && q <= 65535.0) || !(std::fabs(lo + q * dl - planes[k]) <= 0.50005 * dl)) q16_ok = false;
                    else if (k < 2) nodes16[i].q[2 * a + k] = (uint16_t)q;
                    else nodes16_ref[i].q[2 * a + k - 2] = (uint16_t)q;
                }
            }
        }
        for (uint32_t i = 0; i < n && q16_ok; ++i) {
            nodes16[i].link = nodes[i].prim != NONE ? nodes[i].prim : (NODE16_INTERNAL | nodes[i].miss);
            if (n_cull) nodes16_ref[i].link = nodes16[i].link;
        }
        if (!q16_ok) {
            nodes16.clear();
            nodes16_ref.clear();
        }
    }

    // Many box tests of such a scene are left to the f64 tiers: the walk then collects those lanes and resolves
    // them together instead of one by one inside its loop (RTX_BATCH_F64_BOXES=0/1 overrides the choice).
    // Where the walk is bound by the number of lines fetched (trees well beyond one L2, 4 MiB per XCD), eight records per line beat
    // four; smaller trees keep the f32 records, whose test is decided more often.  RTX_NODE16 = 0 / 1 overrides.
    const size_t node16_min = env_u32("RTX_NODE16_MIN_BYTES", 4u << 20);
    const uint32_t use16 = q16_ok ? env_u32("RTX_NODE16", (size_t)n * sizeof(DNode32) > node16_min ? 1u : 0u, true) : 0u;
    // (With the quantised records 0.9 % of the visits are undecided, and resolved one by one each makes its whole wave wait for a
    // second round trip to memory in about half of the steps: batched, 10^6 spheres render 4 % faster, 949 against 912 Mrays/s.)
    const uint32_t batch_undecided = env_u32("RTX_BATCH_F64_BOXES", ((uint64_t)thin_nodes * 20u >= n || use16) ? 1u : 0u, true);
    if (batch_undecided && !use16) {   // (the quantised records have no room for the flag; their scenes are not the thin-box kind)
        // In that form of the walk a node is flagged (bit 0 of its link, whose low five bits are otherwise zero) when every
        // surface below it lies inside the node's box with room to spare.  For such a node "a surface below is hit inside the
        // interval" implies "the node's box test passes", so where the f32 tier cannot decide the box the walk may go on as if
        // it had passed: if the reference culls the node, nothing below it can be hit, and the outcome is the same.  Room to
        // spare: every corner of every quad and triangle below at least 1e-9 of the largest box coordinate, and 1e-6 of the
        // surface's own extent, away from all six faces -- far more than the rounding of either test; the reference's padded
        // boxes (src/aabb.rs:78-87) leave 1e-4 or a whole extent.  A sphere touches its box, so nodes above spheres never qualify.
        struct Corners { double lo[3], hi[3], ext[3]; bool ok; };
        std::vector<Corners> cb(n);
        for (uint32_t i = n; i-- > 0u;) {
            Corners& c = cb[i];
            const uint32_t prim = desc->node_prim[i];
            if (prim != RTX_NONE) {
                c.ok = prims[prim].kind != RTX_SPHERE;
                if (c.ok) {
                    const DPlanar& g = planars[prims[prim].slot];
                    const bool quad = prims[prim].kind == RTX_QUAD;
                    for (int a = 0; a < 3; ++a) {
                        const double p0 = g.q[a], p1 = g.q[a] + g.u[a], p2 = g.q[a] + g.v[a], p3 = quad ? g.q[a] + g.u[a] + g.v[a] : p0;
                        c.lo[a] = std::fmin(std::fmin(p0, p1), std::fmin(p2, p3));
                        c.hi[a] = std::fmax(std::fmax(p0, p1), std::fmax(p2, p3));
                        c.ext[a] = std::fabs(g.u[a]) + std::fabs(g.v[a]);
                    }
                }
            } else {
                const Corners &l = cb[i + 1], &r = cb[desc->node_right[i]];
                c.ok = l.ok && r.ok;
                for (int a = 0; a < 3; ++a) {
                    c.lo[a] = std::fmin(l.lo[a], r.lo[a]);
                    c.hi[a] = std::fmax(l.hi[a], r.hi[a]);
                    c.ext[a] = std::fmax(l.ext[a], r.ext[a]);
                }
            }
            bool inside = c.ok;
            for (int a = 0; a < 3 && inside; ++a) {
                const double big = std::fmax(std::fabs(nodes[i].bmin[a]), std::fabs(nodes[i].bmax[a]));
                const double room = std::fmax(1e-9 * big, 1e-6 * c.ext[a]) + 1e-300;
                inside = c.lo[a] - nodes[i].bmin[a] >= room && nodes[i].bmax[a] - c.hi[a] >= room;
            }
            if (inside) {
                nodes32[i].miss |= 1u;
                if (n_cull) nodes32_cull[i].miss |= 1u;
            }
        }
    }

    ColdTimer cold_upload("scene upload (first HIP calls of a process included)");
    rtx_scene* sc = new rtx_scene();
    auto bail = [&](int code) {
        delete sc;
        return code;
    };
    hipError_t e = hipGetDevice(&sc->device);
    if (e != hipSuccess) return bail(fail(RTX_ERR_NO_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e)));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sc->device) == hipSuccess && prop.multiProcessorCount > 0) sc->n_cus = (uint32_t)prop.multiProcessorCount;
    sc->trace_max_waves = sc->n_cus * env_u32("RTX_WAVES_PER_CU", 20);

#define UP(buf, vec)                                                                                                   \
    do {                                                                                                               \
        hipError_t _e = sc->buf.resize((vec).size());                                                                  \
        if (_e == hipSuccess && !(vec).empty()) _e = hipMemcpy(sc->buf.p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice); \
        if (_e != hipSuccess) return bail(fail(_e == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP, std::string("scene upload: ") + hipGetErrorString(_e))); \
    } while (0)
    UP(nodes, nodes);
    UP(spheres, spheres);
    UP(planars, planars);
    if (n_cull) {
        UP(nodes32, nodes32_cull);
        UP(nodes32_ref, nodes32);
        UP(nodes16_ref, nodes16_ref);
    } else {
        UP(nodes32, nodes32);
    }
    UP(leaf_codes, leaf_codes);
    UP(nodes16, nodes16);
    UP(boxes64, boxes64);
    UP(rec_user_id, rec_user_id);
    UP(shade_recs, shade_recs);
#undef UP
    sc->d.nodes = sc->nodes.p;
    sc->d.spheres = sc->spheres.p;
    sc->d.planars = sc->planars.p;
    sc->d.n_nodes = n;
    sc->d.filter_ok = filter_ok ? 1u : 0u;
    sc->d.batch_undecided = batch_undecided;
    // Waiting for more lanes before the surface tests run pays where those tests are a large part of the walk: scenes of quads
    // and triangles (57 flop a test).  With the reference's tripled boxes such scenes made 4 to 55 surface tests per ray and 32
    // waiting lanes were best; culled with the smaller boxes above they make 1.6 to 7, and 24 are: Cornell box 273.6 against
    // 277.8 ms per 2000 spp, 10^6 triangles 804 against 851 ms per 20 spp (16: 808, 8: 842).
    const size_t scene_bytes = (size_t)n * sizeof(DNode32) + spheres.size() * sizeof(DSphere) + planars.size() * sizeof(DPlanar);
    const bool planar_scene = planars.size() > spheres.size(), small_scene = scene_bytes <= 32u * 1024u;
    sc->d.pend_at = env_u32("RTX_PEND_AT", planar_scene ? 24u : 16u);
    if (sc->d.pend_at > 64u) sc->d.pend_at = 64u;
    // Idle lanes take their next rays once 32 lanes are idle; once 24 are where the whole scene sits in LDS and is made of quads
    // and triangles (Cornell box: 16 / 24 idle lanes 275.8 / 273.6 ms per 2000 spp; the headline scene loses 2 % with 16, a
    // 10^6-triangle tree in global memory loses 0.8 % with 16 and 19 % with 48).
    sc->d.refill_at = env_u32("RTX_REFILL_AT", planar_scene && small_scene ? 24u : 32u);
    if (sc->d.refill_at > 64u) sc->d.refill_at = 64u;
    // The kernel that keeps the paths in registers runs a bounce (some 1500 instructions whatever the number of lanes in it) once this
    // many lanes are due: 16 / 24 / 32 / 40 / 48 / 56 / 64 give 51.8 / 45.2 / 42.5 / 40.7 / 40.3 / 40.4 / 44.9 ms per 100 spp of the headline frame
    // (measured again after the walk's last trims: 44 / 48 / 52 give 69.7 / 69.3 / 69.1 ms per 200 spp).
    sc->d.bounce_at = env_u32("RTX_BOUNCE_AT", 52u);
    if (sc->d.bounce_at > 64u) sc->d.bounce_at = 64u;
    sc->d.nodes32 = sc->nodes32.p;
    sc->d.leaf_codes = sc->leaf_codes.p;
    sc->d.nodes16 = q16_ok ? sc->nodes16.p : nullptr;
    for (int a = 0; a < 3; ++a) {
        sc->d.q_org[a] = q_org[a];
        sc->d.q_delta[a] = q_delta[a];
    }
    sc->d.use16 = use16;
    sc->d.boxes64 = sc->boxes64.p;
    sc->d.n_spheres = n_spheres;
    sc->d.n_planars = (uint32_t)planars.size();
    sc->d.rec_user_id = sc->rec_user_id.p;
    sc->d.shade_recs = sc->shade_recs.p;
    sc->f32_ok = f32_ok && filter_ok && n < (1u << 25);   // the f32 walk keeps two flags above a record's byte offset (n * 32 < 2^30)
    sc->n_cull = n_cull;
    sc->d_ref = sc->d;   // the same scene with the reference's boxes in the first tier's records: what counting calls walk
    if (n_cull) {
        sc->d_ref.nodes32 = sc->nodes32_ref.p;
        sc->d_ref.nodes16 = q16_ok ? sc->nodes16_ref.p : nullptr;
    }
    if (hipEventCreate(&sc->ev0) != hipSuccess || hipEventCreate(&sc->ev1) != hipSuccess)
        return bail(fail(RTX_ERR_HIP, "hipEventCreate failed"));
    // Set-up of the allocation-free closest-hit entry: where this device keeps its copy of the scratch array.
    // (A failure here is not the scene's: rtx_trace_closest_device looks again and reports it.)
    (void)noalloc_setup_current_device();
    *out = sc;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* sc) {
    if (!sc) return;
    if (sc->ev0) (void)hipEventDestroy(sc->ev0);
    if (sc->ev1) (void)hipEventDestroy(sc->ev1);
    delete sc;
}

static int read_counters(PathPool* pp, hipStream_t s, rtx_stats* stats) {
    DRenderCtl h;
    HIP_TRY(hipMemcpyAsync(&h, pp->ctl.p, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stats->rays += h.rays;
    stats->node_visits += h.node_visits;
    stats->sphere_tests += h.sphere_tests;
    stats->planar_tests += h.planar_tests;
    stats->box_fallbacks += h.box_fallbacks;
    return RTX_OK;
}

int rtx_trace_closest(rtx_scene* sc, const double* org, const double* dir, uint64_t n, double tmin, double tmax,
                      uint32_t flags, uint32_t* out_id, double* out_t, rtx_stats* stats) {
    if (!sc || (n && (!org || !dir || !out_id || !out_t))) return fail(RTX_ERR_BAD_ARG, "rtx_trace_closest: null argument");
    if (n > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_trace_closest: more than 2^32-1 rays in one call");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return RTX_OK;
    HIP_TRY(hipSetDevice(sc->device));
    uint32_t n_sub;
    uint64_t sub_capacity;
    pool_geometry(n, sc->trace_max_waves, &n_sub, &sub_capacity);
    if (sub_capacity * n_sub > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_trace_closest: batch too large");
    PathPool* pp = sc->lane(0);
    int rc = pp->ensure_capacity((size_t)n);
    if (rc != RTX_OK) return rc;
    hipStream_t s = nullptr;
    DevBuf<double> d_rays;
    HIP_TRY(d_rays.resize(6 * n));
    HIP_TRY(hipMemcpyAsync(d_rays.p, org, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_rays.p + 3 * n, dir, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
    DRenderCtl h{};
    h.live_total = (uint32_t)n;
    h.rays = n;
    HIP_TRY(hipMemcpyAsync(pp->ctl.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
    launch_rays_to_state(d_rays.p, d_rays.p + 3 * n, n, pp->state[0], s);
    // the batch as it lies in the state buffer, cut into one piece per wave
    const DPool pool{nullptr, n_sub, (uint32_t)sub_capacity, (uint32_t)n};
    HIP_TRY(hipEventRecord(sc->ev0, s));
    run_traverse(sc, pp->state[0], pp->hits, pool, tmin, tmax, flags, pp->ctl.p, s);
    HIP_TRY(hipEventRecord(sc->ev1, s));
    launch_user_ids(sc->d.rec_user_id, sc->d.n_spheres, pp->hits.prim, n, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_id, pp->hits.prim, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_t, pp->hits.t, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint64_t i = 0; i < n; ++i)
        if (out_id[i] == RTX_NONE) out_t[i] = INFINITY;
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev0, sc->ev1));
        stats->ms_total = ms;
        stats->ms_traverse = ms;
        stats->iterations = 1;
        rc = read_counters(pp, s, stats);
        if (rc != RTX_OK) return rc;
    }
    return RTX_OK;
}

static DCamera to_dcamera(const rtx_camera* c) {
    DCamera d{};
    d.image_width = c->image_width;
    d.image_height = c->image_height;
    d.samples_per_pixel = c->samples_per_pixel;
    d.max_depth = c->max_depth;
    d.defocus_angle = c->defocus_angle;
    for (int a = 0; a < 3; ++a) {
        d.background[a] = c->background[a];
        d.center[a] = c->center[a];
        d.pixel00_loc[a] = c->pixel00_loc[a];
        d.pixel_du[a] = c->pixel_du[a];
        d.pixel_dv[a] = c->pixel_dv[a];
        d.defocus_disk_u[a] = c->defocus_disk_u[a];
        d.defocus_disk_v[a] = c->defocus_disk_v[a];
    }
    return d;
}

static int check_camera(const rtx_camera* cam, const char* who) {
    if (!cam) return fail(RTX_ERR_BAD_ARG, std::string(who) + ": null camera");
    if (cam->image_width == 0 || cam->image_height == 0) return fail(RTX_ERR_BAD_ARG, std::string(who) + ": empty image");
    if ((uint64_t)cam->image_width * cam->image_height > 0xFFFFFFFFull)
        return fail(RTX_ERR_BAD_ARG, std::string(who) + ": more than 2^32-1 pixels");
    return RTX_OK;
}

int rtx_primary_rays(const rtx_camera* cam, uint64_t seed, uint32_t sample, double* org, double* dir) {
    int rc = check_camera(cam, "rtx_primary_rays");
    if (rc != RTX_OK) return rc;
    if (!org || !dir) return fail(RTX_ERR_BAD_ARG, "rtx_primary_rays: null output");
    const uint32_t npix = cam->image_width * cam->image_height;
    DevBuf<double> d;
    HIP_TRY(d.resize(6 * (size_t)npix));
    launch_primary_rays(to_dcamera(cam), seed, sample, npix, d.p, d.p + 3 * (size_t)npix, 2048, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(org, d.p, 3 * (size_t)npix * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(dir, d.p + 3 * (size_t)npix, 3 * (size_t)npix * sizeof(double), hipMemcpyDeviceToHost));
    return RTX_OK;
}

namespace {

// What one lane of a render call needs to enqueue a round.
struct LaneRun {
    PathPool* pp = nullptr;
    DPool pool{};
    uint32_t* live = nullptr;
    uint32_t* surv[2] = {nullptr, nullptr};
    uint32_t n_sub = 0, sub_capacity = 0;
    uint64_t iters = 0;     // rounds enqueued
    size_t n_evs = 0;       // kernel timing events used
    uint32_t polls = 0;     // polls enqueued
    bool done = false;
    DRenderCtl h_ctl{};     // source of the lane's asynchronous control-block upload: lives until the lane has drained
};

}  // namespace

int rtx_render_accumulate(rtx_scene* sc, const rtx_camera* cam, uint64_t seed, uint32_t spp_begin, uint32_t spp_count,
                          uint32_t flags, double* d_accum, void* stream, rtx_stats* stats) {
    if (!sc || !d_accum) return fail(RTX_ERR_BAD_ARG, "rtx_render_accumulate: null argument");
    int rc = check_camera(cam, "rtx_render_accumulate");
    if (rc != RTX_OK) return rc;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    const uint64_t npix = (uint64_t)cam->image_width * cam->image_height;
    const uint64_t total = npix * spp_count;
    if (total == 0 || cam->max_depth == 0) return RTX_OK;   // ray_color returns ZERO at remaining == 0 (camera.rs:190-192)
    if ((uint64_t)spp_begin + spp_count > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_render_accumulate: sample index overflow");
    const bool no_wait = (flags & RTX_FLAG_NO_WAIT) != 0;
    if (no_wait && ((flags & (RTX_FLAG_EXACT_DIVIDE | RTX_FLAG_F64_FILTER | RTX_FLAG_TIME_KERNELS)) || counting(flags) || !sc->f32_ok ||
                    env_u32("RTX_PERSISTENT", 1, true) == 0u))
        return fail(RTX_ERR_BAD_ARG, "rtx_render_accumulate: RTX_FLAG_NO_WAIT needs the single-launch render (no other box-test tier, no timing or counting flags, a scene within the f32 tier's range)");
    HIP_TRY(hipSetDevice(sc->device));
    hipStream_t s = (hipStream_t)stream;

    // Default path: the whole sample loop as ONE launch of persistent waves on the caller's stream.  It needs the
    // single-precision box tier; calls that ask for another tier, or scenes outside its range, take the round loop below.
    if (sc->f32_ok && !(flags & (RTX_FLAG_EXACT_DIVIDE | RTX_FLAG_F64_FILTER | RTX_FLAG_TIME_KERNELS)) && env_u32("RTX_PERSISTENT", 1, true) != 0u) {
        PathPool* pp = sc->lane(0);
        uint32_t mode = lds_mode_for(sc, flags);
        bool lds_recs = false;   // the shade records next to the surface tables, where there is room
        // Scenes that sit in LDS whole (tree, surface tables and shade records) can take the resident form of the kernel: paths stay
        // in registers from bounce to bounce and the pool below holds nothing.  It is the faster form where paths bounce (the headline
        // frame: 157 against 177 ms; a 62-spp call 20.2 against 23.2 ms; simple_light 4.2 against 6.7 ms, demo_spheres 1.8 against
        // 2.4) and the slower one where nearly every path ends at its first ray (a sparse soup in front of the sky: 2.6 against 2.2 ms
        // for a 16-spp call; hello_triangle 0.6 against 0.3 ms), so the scene's last call decides: 1.5 rays per path or more.  Scenes
        // whose undecided box tests are batched (thin boxes: the Cornell box, 125 against 104 ms per 500 spp) keep the pooled form, as do
        // a scene's first call (but see the probe below) and calls that count work (the counters exist in the pooled form only).
        // RTX_RESIDENT=0 / 1 forces one.
        const char* force_form = std::getenv("RTX_RESIDENT");
        const bool want_resident = !counting(flags) && !(flags & RTX_FLAG_NO_LDS) &&
                                   ((force_form && *force_form) ? std::atoi(force_form) != 0 : (sc->rays_per_path >= 1.5 && !sc->d.batch_undecided));
        const size_t wide_limit = env_u32("RTX_LDS_WIDE_BYTES", 150u * 1024u, true);
        // A scene's first call knows no rays per path yet.  Where the answer matters (the scene could take the resident form and the
        // call is long enough to gain by it) the call's first sample goes ahead as a call of its own, which measures it, and the
        // other samples follow in the form that suits: the reference renders one frame per process, so its first call is its only one.
        if (sc->rays_per_path == 0.0 && !no_wait && !(force_form && *force_form) && !counting(flags) && !(flags & RTX_FLAG_NO_LDS) &&
            !sc->d.batch_undecided && spp_count >= 8u && persistent_lds_bytes(sc->d, 3u, true) <= wide_limit &&
            sc->occupancy(1u, 16u, true, [&] { return resident_waves_per_cu(sc->d, 16u); }) != 0u && env_u32("RTX_PROBE_FIRST_CALL", 1, true) != 0u) {
            rtx_stats first{}, rest{};
            {
                ColdTimer cold("first-sample probe call (pooled form, 1 spp)");
                rc = rtx_render_accumulate(sc, cam, seed, spp_begin, 1u, flags, d_accum, stream, &first);
            }
            if (rc != RTX_OK) return rc;
            if (sc->rays_per_path == 0.0) sc->rays_per_path = 1.0;   // (cannot be: a path is at least one ray)
            rc = rtx_render_accumulate(sc, cam, seed, spp_begin + 1u, spp_count - 1u, flags, d_accum, stream, &rest);
            if (rc != RTX_OK) return rc;
            if (stats) {
                *stats = rest;
                stats->ms_total += first.ms_total;
                stats->paths += first.paths;
                stats->rays += first.rays;
                stats->node_visits += first.node_visits;
                stats->sphere_tests += first.sphere_tests;
                stats->planar_tests += first.planar_tests;
                stats->box_fallbacks += first.box_fallbacks;
                if (first.iterations > stats->iterations) stats->iterations = first.iterations;
            }
            return RTX_OK;
        }
        const bool recs_allowed = env_u32("RTX_LDS_RECS", 1, true) != 0u;
        if (mode <= 1u && !(flags & RTX_FLAG_NO_LDS)) {
            // What does not fit in LDS five times per CU may fit once: one 1024-thread block per CU then holds the tree, and the surface
            // tables if there is room, for sixteen waves (RTX_LDS_WIDE_BYTES: the most LDS such a block may take, 0 = never).
            if (persistent_lds_bytes(sc->d, 3u, false) <= wide_limit) mode = 3u;
            else if (mode == 0u && persistent_lds_bytes(sc->d, 4u, false) <= wide_limit) mode = 4u;
            if (mode == 3u) lds_recs = recs_allowed && persistent_lds_bytes(sc->d, 3u, true) <= wide_limit;
        }
        // The resident form exists as one block per CU with the tree, the surface tables and the shade records in LDS; its waves
        // stage the primary rays they make in LDS as well, 64 at a time where there is room, else 48, 32 or 16.  A small scene
        // (mode 2) moves to that block shape only once the form is known to fit; otherwise it stays in its five blocks per CU.
        uint32_t resident_waves = 0u, resident_batch = 0u;
        if (want_resident && recs_allowed && ((mode == 3u && lds_recs) || (mode == 2u && persistent_lds_bytes(sc->d, 3u, true) <= wide_limit)))
            for (uint32_t b : {64u, 48u, 32u, 16u}) {
                resident_waves = sc->occupancy(1u, b, true, [&] { return resident_waves_per_cu(sc->d, b); });
                resident_batch = b;
                if (resident_waves) break;
            }
        const bool resident = resident_waves != 0u;
        if (resident) {
            mode = 3u;
            lds_recs = true;
        } else if (mode == 2u) {
            lds_recs = recs_allowed && persistent_lds_bytes(sc->d, 2u, true) <= env_u32("RTX_LDS_SCENE_BYTES", 32u * 1024u);
        }
        const uint32_t waves = sc->n_cus * (resident ? resident_waves   // exactly what is resident at once
                                                     : sc->occupancy(0u, mode, lds_recs, [&] { return persistent_waves_per_cu(sc->d, mode, lds_recs); }));
        const uint64_t want = env_u32("RTX_POOL_SLOTS", 1u << 22);   // ~800 slots per wave: a round ends with a drain, so long rounds pay, but state that stays in the last-level cache pays too
        const uint64_t need = total < want ? total : want;
        uint32_t n_sub;
        uint64_t per_sub;
        pool_geometry(need, waves, &n_sub, &per_sub);
        if (per_sub * n_sub > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_render_accumulate: pool larger than 2^32-1 slots");
        rc = resident ? pp->ensure_ctl() : pp->ensure_capacity((size_t)per_sub * n_sub);   // (the resident form keeps no path in memory)
        if (rc != RTX_OK) return rc;
        HIP_TRY(pp->params.resize(1));
        uint32_t slot = 0;
        rc = pp->ring_acquire(&slot);
        if (rc != RTX_OK) return rc;
        DRenderCtl& h = pp->h_ctl_ring[slot];
        h = DRenderCtl{};
        h.total_paths = total;
        DRenderParams& P = pp->h_params_ring[slot];
        P = DRenderParams{};
        P.sc = scene_tables(sc, flags);
        P.cam = to_dcamera(cam);
        P.st[0] = pp->state[0];
        P.st[1] = pp->state[1];
        P.hits = pp->hits;
        P.ctl = pp->ctl.p;
        P.accum = d_accum;
        P.seed = seed;
        P.tmin = 0.001;   // Interval(0.001, inf), camera.rs:194
        P.tmax0 = INFINITY;
        P.n_sub = n_sub;
        P.cap = (uint32_t)per_sub;
        P.spp_begin = spp_begin;
        P.lds_surfaces = (mode == 2u || mode == 3u) ? 1u : 0u;
        P.lds_recs = lds_recs ? 1u : 0u;
        P.resident_batch = resident_batch;
        P.tail_in_registers = env_u32("RTX_TAIL_IN_REGISTERS", 1, true);
        // block-by-block pixel order for trees that are walked from global memory (mode 0)
        P.tile_order = (cam->image_width % 32u == 0u && cam->image_height % 16u == 0u && !resident && env_u32("RTX_TILE_ORDER", mode == 0u ? 1u : 0u, true) != 0u) ? 1u : 0u;
        // ... and in eight bands, one per XCD, each with its own path counter (only with that order, for frames of at least 2^20 pixels
        // and calls of at most 2^32 - 1 samples' worth of paths per band)
        // (10^6 spheres 967 -> 986 Mrays/s, 10^6 triangles 656 -> 665: what an L2 that sees an eighth of the primary rays buys)
        P.xcd_bands = (P.tile_order && !resident && npix >= (1u << 20)) ? env_u32("RTX_XCD_BANDS", 8u, true) : 0u;
        if (P.xcd_bands > 8u) P.xcd_bands = 8u;
        HIP_TRY(hipMemcpyAsync(pp->ctl.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
#ifdef RTX_PROFILE_PHASES
        HIP_TRY(hipMemsetAsync(pp->ctl.p + 1, 0, 7 * sizeof(DRenderCtl), s));
#endif
        HIP_TRY(hipMemcpyAsync(pp->params.p, &P, sizeof(P), hipMemcpyHostToDevice, s));
        HIP_TRY(hipEventRecord(pp->ring_ev[slot], s));   // the slot may be refilled once both uploads have executed
        HIP_TRY(hipEventRecord(sc->ev0, s));
        if (resident) launch_render_resident(pp->params.p, P, s);
        else launch_render_persistent(pp->params.p, P, counting(flags), mode, s);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(sc->ev1, s));
        if (no_wait) return RTX_OK;
        HIP_TRY(hipStreamSynchronize(s));
        DRenderCtl back;
        HIP_TRY(hipMemcpy(&back, pp->ctl.p, sizeof(back), hipMemcpyDeviceToHost));
        if (total && !counting(flags)) sc->rays_per_path = (double)back.rays / (double)total;
        if (stats) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, sc->ev0, sc->ev1);
            stats->ms_total = ms;
            stats->paths = total;
            stats->rays = back.rays;
            stats->node_visits = back.node_visits;
            stats->sphere_tests = back.sphere_tests;
            stats->planar_tests = back.planar_tests;
            stats->box_fallbacks = back.box_fallbacks;
            stats->iterations = back.rounds_max;
            stats->kernel_form = resident ? RTX_FORM_RESIDENT : RTX_FORM_POOLED;
        }
#ifdef RTX_PROFILE_PHASES
        {
            unsigned long long pr[10];
            (void)hipMemcpy(pr, reinterpret_cast<unsigned long long*>(pp->ctl.p + 1) + 16, sizeof(pr), hipMemcpyDeviceToHost);
            if (pr[7])
                std::fprintf(stderr, "PHASES per wave (cycles): generate %.0f traverse %.0f shade %.0f, life %.0f (longest %llu), %llu waves\n",
                             (double)pr[0] / pr[7], (double)pr[1] / pr[7], (double)pr[2] / pr[7], (double)pr[6] / pr[7], pr[9], pr[7]);
            unsigned long long tr[16];
            (void)hipMemcpy(tr, reinterpret_cast<unsigned long long*>(pp->ctl.p + 1), sizeof(tr), hipMemcpyDeviceToHost);
            const double all = (double)(tr[0] + tr[1] + tr[2] + tr[3]);
            if (all > 0)
                std::fprintf(stderr, "PHASES of the walk (share of its cycles): refill %.3f node loop %.3f surface tests %.3f retire %.3f; %.1f node-loop steps per pass of the outer loop\n",
                             tr[0] / all, tr[1] / all, tr[2] / all, tr[3] / all, tr[5] ? (double)tr[4] / tr[5] : 0.0);
            if (resident && tr[7])   // the resident form's waves: "refill" = new primary rays and ray set-up, "retire" = the bounce
                std::fprintf(stderr, "PHASES of a resident wave: %.0f cycles, %.1f passes of the walk per bounce, %.0f node-loop steps and %.0f bounces per wave; staging %.3f and taking %.3f of the cycles besides (%.0f stagings per wave)\n",
                             (double)tr[6] / tr[7], tr[12] ? (double)tr[5] / tr[12] : 0.0, (double)tr[4] / tr[7], (double)tr[12] / tr[7],
                             tr[13] / (all + tr[13] + tr[14]), tr[14] / (all + tr[13] + tr[14]), (double)tr[15] / tr[7]);
            if (resident && tr[7] && tr[9])   // (the instrumented build keeps the compiler's form of the node loop)
                std::fprintf(stderr, "PHASES of a resident wave's node loop: the loop proper %.3f of the cycles (the 'node loop' share above is then what surrounds it), %.0f steps of the wave at %.1f cycles each, %.1f lanes in a step, %.1f steps per pass\n",
                             tr[8] / (all + tr[8] + tr[13] + tr[14]), (double)tr[9] / tr[7], (double)tr[8] / tr[9], (double)tr[10] / tr[9], tr[5] ? (double)tr[9] / tr[5] : 0.0);
        }
#endif
        return RTX_OK;
    }

    // The paths of the call are cut into contiguous ranges, one per lane; every lane has its own
    // pool of up to `want` slots (one slot per path for small calls), one sub-pool per wave of its launches.
    const uint64_t want = env_u32("RTX_POOL_SLOTS", 1u << 22);
    const uint32_t max_waves = sc->n_cus * env_u32("RTX_WAVES_PER_CU", 20);
    const uint32_t check_every = env_u32("RTX_CHECK_EVERY", 4);
    uint32_t n_lanes = env_u32("RTX_LANES", 2);
    if (n_lanes > 8) n_lanes = 8;
    while (n_lanes > 1 && total / n_lanes < (1u << 18)) --n_lanes;   // small calls: not worth splitting
    const bool time_kernels = (flags & RTX_FLAG_TIME_KERNELS) != 0;
    const DCamera dcam = to_dcamera(cam);

    HIP_TRY(hipEventRecord(sc->ev0, s));
    std::vector<LaneRun> lanes(n_lanes);
    for (uint32_t j = 0; j < n_lanes; ++j) {
        LaneRun& L = lanes[j];
        L.pp = sc->lane(j);
        const uint64_t first = total * j / n_lanes, last = total * (j + 1) / n_lanes;
        const uint64_t need = last - first < want ? last - first : want;
        uint64_t per_sub;
        pool_geometry(need, max_waves, &L.n_sub, &per_sub);
        if (per_sub * L.n_sub > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_render_accumulate: pool larger than 2^32-1 slots");
        L.sub_capacity = (uint32_t)per_sub;
        rc = L.pp->ensure_host_side();
        if (rc == RTX_OK) rc = L.pp->ensure_capacity((size_t)L.sub_capacity * L.n_sub);
        if (rc == RTX_OK) rc = L.pp->ensure_counters(L.n_sub);
        if (rc != RTX_OK) return rc;
        L.live = L.pp->counters.p;
        L.surv[0] = L.pp->counters.p + L.n_sub;
        L.surv[1] = L.pp->counters.p + 2 * (size_t)L.n_sub;
        L.pool = DPool{L.live, L.n_sub, L.sub_capacity, 0u};
        hipStream_t ls = L.pp->stream;
        HIP_TRY(hipStreamWaitEvent(ls, sc->ev0, 0));   // after whatever the caller queued on `stream`
        L.h_ctl.next_path[0] = first;
        L.h_ctl.total_paths = last;
        HIP_TRY(hipMemcpyAsync(L.pp->ctl.p, &L.h_ctl, sizeof(L.h_ctl), hipMemcpyHostToDevice, ls));
        HIP_TRY(hipMemsetAsync(L.pp->ctl.p + 1, 0, 3 * sizeof(DRenderCtl), ls));
        HIP_TRY(hipMemsetAsync(L.pp->counters.p, 0, 3 * (size_t)L.n_sub * sizeof(uint32_t), ls));
        L.pp->h_poll[0] = L.pp->h_poll[1] = 1u;
    }

    // One round of a lane works on its state buffer (round & 1), all on the lane's stream:
    //   generate : start new paths behind the survivors until the sub-pools are full
    //   traverse : closest hit of every ray in the pool
    //   shade    : scatter / terminate; survivors are written densely into the other buffer
    int status = RTX_OK;
    auto enqueue_batch = [&](LaneRun& L) -> int {
        PathPool* pp = L.pp;
        hipStream_t ls = pp->stream;
        DRenderCtl* ctl = pp->ctl.p;
        for (uint32_t r = 0; r < check_every; ++r) {
            const uint32_t cur = (uint32_t)(L.iters & 1u), nxt = cur ^ 1u;
            hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr, e3 = nullptr;
            if (time_kernels) {
                std::vector<hipEvent_t>& evs = pp->kernel_events;
                while (evs.size() < L.n_evs + 4) {
                    hipEvent_t e = nullptr;
                    if (hipEventCreate(&e) != hipSuccess) return fail(RTX_ERR_HIP, "hipEventCreate failed");
                    evs.push_back(e);
                }
                e0 = evs[L.n_evs]; e1 = evs[L.n_evs + 1]; e2 = evs[L.n_evs + 2]; e3 = evs[L.n_evs + 3];
                L.n_evs += 4;
                (void)hipEventRecord(e0, ls);
            }
            launch_generate(dcam, pp->state[cur], ctl, L.surv[cur], L.live, L.n_sub, L.sub_capacity, cur, seed, spp_begin, ls);
            if (time_kernels) (void)hipEventRecord(e1, ls);
            run_traverse(sc, pp->state[cur], pp->hits, L.pool, 0.001, INFINITY, flags, pp->ctl.p, ls);   // Interval(0.001, inf), camera.rs:194
            if (time_kernels) (void)hipEventRecord(e2, ls);
            launch_shade(sc->d, dcam, pp->state[cur], pp->hits, L.pool, pp->state[nxt], L.surv[nxt], seed, d_accum, ls);
            if (time_kernels) (void)hipEventRecord(e3, ls);
#ifdef RTX_REPLAY
            // Kernel A/B aid (never compiled into the product): at one steady-state
            // round, re-run traverse and shade on the same input and print their times.
            if (L.iters == env_u32("RTX_REPLAY_ROUND", 15)) {
                const uint32_t reps = env_u32("RTX_REPLAY_REPS", 20);
                hipEvent_t a, b, c;
                (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventCreate(&c);
                (void)hipEventRecord(a, ls);
                for (uint32_t q = 0; q < reps; ++q) run_traverse(sc, pp->state[cur], pp->hits, L.pool, 0.001, INFINITY, flags, pp->ctl.p, ls);
                (void)hipEventRecord(b, ls);
                for (uint32_t q = 0; q < reps; ++q)
                    launch_shade(sc->d, dcam, pp->state[cur], pp->hits, L.pool, pp->state[nxt], L.surv[nxt], seed, d_accum, ls);
                (void)hipEventRecord(c, ls);
                (void)hipStreamSynchronize(ls);
                float t1 = 0.f, t2 = 0.f;
                (void)hipEventElapsedTime(&t1, a, b);
                (void)hipEventElapsedTime(&t2, b, c);
                uint32_t n_live = 0;
                (void)hipMemcpy(&n_live, &ctl->live_total, 4, hipMemcpyDeviceToHost);
                std::fprintf(stderr, "REPLAY round %llu live %u: traverse %.1f us, shade %.1f us\n", (unsigned long long)L.iters, n_live,
                             1e3 * t1 / reps, 1e3 * t2 / reps);
            }
#endif
            ++L.iters;
        }
        // a lane is finished when a round had nothing to trace: no survivors and no paths left to start
        const uint32_t slot = L.polls & 1u;
        hipError_t e = hipMemcpyAsync(pp->h_poll + slot, &ctl->live_total, sizeof(uint32_t), hipMemcpyDeviceToHost, ls);
        if (e == hipSuccess) e = hipEventRecord(pp->ev_poll[slot], ls);
        if (e != hipSuccess) return fail(RTX_ERR_HIP, std::string("render loop: ") + hipGetErrorString(e));
        ++L.polls;
        return RTX_OK;
    };

    // Two batches of rounds are kept in flight per lane, so a lane's stream does not run dry while the
    // host waits for an older batch's poll; the rounds enqueued after a lane has finished find no work.
    for (LaneRun& L : lanes)
        for (int k = 0; k < 2 && status == RTX_OK; ++k) status = enqueue_batch(L);
    uint32_t n_done = 0;
    while (status == RTX_OK && n_done < n_lanes) {
        for (LaneRun& L : lanes) {
            if (L.done || status != RTX_OK) continue;
            const uint32_t slot = L.polls & 1u;   // the older of the two polls in flight
            hipError_t e = hipEventSynchronize(L.pp->ev_poll[slot]);
            if (e != hipSuccess) {
                status = fail(RTX_ERR_HIP, std::string("render loop: ") + hipGetErrorString(e));
                break;
            }
            if (L.pp->h_poll[slot] == 0u) {
                L.done = true;
                ++n_done;
            } else {
                status = enqueue_batch(L);
            }
        }
    }
    // the caller's stream continues after every lane has drained
    for (LaneRun& L : lanes) {
        hipError_t e = hipEventRecord(L.pp->ev_done, L.pp->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, L.pp->ev_done, 0);
        if (e != hipSuccess && status == RTX_OK) status = fail(RTX_ERR_HIP, std::string("render: ") + hipGetErrorString(e));
    }
    if (status == RTX_OK) {
        hipError_t e = hipEventRecord(sc->ev1, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) status = fail(RTX_ERR_HIP, std::string("render: ") + hipGetErrorString(e));
    } else {
        for (LaneRun& L : lanes) (void)hipStreamSynchronize(L.pp->stream);
    }
    if (status == RTX_OK && stats) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, sc->ev0, sc->ev1);
        stats->ms_total = ms;
        stats->paths = total;
        stats->kernel_form = RTX_FORM_ROUND_LOOP;
        for (LaneRun& L : lanes) {
            stats->iterations += L.iters;
            const std::vector<hipEvent_t>& evs = L.pp->kernel_events;
            for (size_t i = 0; i + 3 < L.n_evs; i += 4) {
                float g = 0.f, a = 0.f, b = 0.f;
                (void)hipEventElapsedTime(&g, evs[i], evs[i + 1]);
                (void)hipEventElapsedTime(&a, evs[i + 1], evs[i + 2]);
                (void)hipEventElapsedTime(&b, evs[i + 2], evs[i + 3]);
                stats->ms_generate += g;
                stats->ms_traverse += a;
                stats->ms_shade += b;
            }
            if (status == RTX_OK) status = read_counters(L.pp, L.pp->stream, stats);
#ifdef RTX_PROFILE_PHASES
            unsigned long long pr[14];
            (void)hipMemcpy(pr, L.pp->ctl.p + 1, sizeof(pr), hipMemcpyDeviceToHost);
            if (pr[7])
                std::fprintf(stderr, "PHASES per wave: refill %.0f p1 %.0f p2 %.0f retire %.0f total %.0f cycles; p1 iterations %.1f, outer %.1f (waves %llu), wave life %.1f us; max over all waves and launches: life %.1f us, p1 iterations %llu, p1 cycles %llu; iterations with a fallback %.1f per wave costing %.0f cycles per wave\n",
                             (double)pr[0] / pr[7], (double)pr[1] / pr[7], (double)pr[2] / pr[7], (double)pr[3] / pr[7], (double)pr[6] / pr[7],
                             (double)pr[4] / pr[7], (double)pr[5] / pr[7], pr[7], (double)pr[8] / pr[7] / 100.0, (double)pr[9] / 100.0, pr[10], pr[11], (double)pr[12] / pr[7], (double)pr[13] / pr[7]);
#endif
        }
    }
    return status;
}

uint64_t rtx_ppm_text_bound(uint32_t width, uint32_t height) { return 48ull + 12ull * width * height; }

int rtx_ppm_text(const double* d_rgb, uint32_t width, uint32_t height, char* out_text, uint64_t capacity, uint64_t* out_len, void* stream) {
    if (!d_rgb || !out_text || !out_len) return fail(RTX_ERR_BAD_ARG, "rtx_ppm_text: null argument");
    if (width == 0 || height == 0) return fail(RTX_ERR_BAD_ARG, "rtx_ppm_text: empty image");
    if ((uint64_t)width * height > (1ull << 28)) return fail(RTX_ERR_BAD_ARG, "rtx_ppm_text: more than 2^28 pixels");
    if (rtx_device_count() <= 0) return fail(RTX_ERR_NO_DEVICE, "rtx_ppm_text: no HIP device");
    // The image may live on another device than the caller's current one (rtx_multi_image is device 0's memory whatever the
    // caller's device is): the work runs where the image is, and the caller's device is restored.
    int current = 0, prior = -1;
    hipPointerAttribute_t attr{};
    if (hipGetDevice(&current) == hipSuccess && hipPointerGetAttributes(&attr, d_rgb) == hipSuccess && attr.type == hipMemoryTypeDevice &&
        attr.device != current) {
        HIP_TRY(hipSetDevice(attr.device));
        prior = current;
    }
    (void)hipGetLastError();   // (a pointer the runtime does not know leaves an error behind; the kernels will report a bad one)
    std::string msg;
    const int code = ppm_text_device(d_rgb, width, height, out_text, capacity, out_len, (hipStream_t)stream, msg);
    if (prior >= 0) (void)hipSetDevice(prior);
    return code == RTX_OK ? RTX_OK : fail(code, msg);
}

int rtx_bvh_build(const double* leaf_boxes, uint32_t n, int strategy, uint32_t buckets, uint32_t* leaf_order, double* node_bmin,
                  double* node_bmax, uint32_t* node_right, uint32_t* node_prim, rtx_build_stats* stats) {
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (n == 0) return RTX_OK;   // an empty scene has an empty tree (src/bvh/mod.rs:84-86)
    if (!leaf_boxes || !leaf_order || !node_bmin || !node_bmax || !node_right || !node_prim)
        return fail(RTX_ERR_BAD_ARG, "rtx_bvh_build: null argument");
    if (n >= (1u << 26)) return fail(RTX_ERR_BAD_ARG, "rtx_bvh_build: more than 2^26 - 1 surfaces");
    if (strategy != RTX_BISECT && strategy != RTX_MIDPOINT && strategy != RTX_SAH_EQUAL_SIZE && strategy != RTX_SAH_PER_SURFACE)
        return fail(RTX_ERR_BAD_ARG, "rtx_bvh_build: unknown partition strategy");
    if (strategy == RTX_SAH_EQUAL_SIZE && buckets < 2u && n >= 2u) return fail(RTX_ERR_BUILD, "No valid splitting plane");
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const double lo = leaf_boxes[6ull * i + a], hi = leaf_boxes[6ull * i + 3 + a];
            if (!std::isfinite(lo) || !std::isfinite(hi) || lo > hi)
                return fail(RTX_ERR_BAD_ARG, "rtx_bvh_build: box of surface " + std::to_string(i) + " is not finite or has min > max");
        }
    if (rtx_device_count() <= 0) return fail(RTX_ERR_NO_DEVICE, "rtx_bvh_build: no HIP device");
    std::string msg;
    const int code = build_bvh_device(leaf_boxes, n, strategy, buckets, leaf_order, node_bmin, node_bmax, node_right, node_prim, stats, msg);
    return code == RTX_OK ? RTX_OK : fail(code, msg);
}

int rtx_scale(const double* d_accum, uint64_t n, double scale, double* d_out, void* stream) {
    if (n && (!d_accum || !d_out)) return fail(RTX_ERR_BAD_ARG, "rtx_scale: null argument");
    if (n == 0) return RTX_OK;
    launch_scale(d_accum, n, scale, d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

int rtx_quantize(const double* d_rgb, uint64_t n, uint8_t* d_out, void* stream) {
    if (n && (!d_rgb || !d_out)) return fail(RTX_ERR_BAD_ARG, "rtx_quantize: null argument");
    if (n == 0) return RTX_OK;
    launch_quantize(d_rgb, n, d_out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}

int rtx_render_ppm(rtx_scene* sc, const rtx_camera* cam, uint64_t seed, uint32_t flags, double* out_rgb, char* out_text,
                   uint64_t capacity, uint64_t* out_len, rtx_stats* stats) {
    if (!sc || (!out_rgb && !out_text) || (out_text && !out_len)) return fail(RTX_ERR_BAD_ARG, "rtx_render_ppm: null argument");
    int rc = check_camera(cam, "rtx_render_ppm");
    if (rc != RTX_OK) return rc;
    HIP_TRY(hipSetDevice(sc->device));
    const uint64_t nval = 3ull * cam->image_width * cam->image_height;
    DevBuf<double> acc;
    HIP_TRY(acc.resize(nval));
    HIP_TRY(hipMemset(acc.p, 0, nval * sizeof(double)));
    rc = rtx_render_accumulate(sc, cam, seed, 0, cam->samples_per_pixel, flags, acc.p, nullptr, stats);
    if (rc != RTX_OK) return rc;
    rc = rtx_scale(acc.p, nval, cam->pixel_samples_scale, acc.p, nullptr);
    if (rc != RTX_OK) return rc;
    if (out_text) {
        ColdTimer cold("P3 text on the device and its copy to the host");
        rc = rtx_ppm_text(acc.p, cam->image_width, cam->image_height, out_text, capacity, out_len, nullptr);
        if (rc != RTX_OK) return rc;
    }
    if (out_rgb) {
        ColdTimer cold("copy of the linear image to the host");
        HIP_TRY(hipMemcpy(out_rgb, acc.p, nval * sizeof(double), hipMemcpyDeviceToHost));
    }
    return RTX_OK;
}

int rtx_render(rtx_scene* sc, const rtx_camera* cam, uint64_t seed, uint32_t flags, double* out_rgb, rtx_stats* stats) {
    if (!out_rgb) return fail(RTX_ERR_BAD_ARG, "rtx_render: null argument");
    return rtx_render_ppm(sc, cam, seed, flags, out_rgb, nullptr, 0, nullptr, stats);
}

int rtx_selftest_cull_boxes(const rtx_scene_desc* desc, int shrink, double* out_bmin, double* out_bmax, uint32_t* out_differ) {
    if (!desc || !out_bmin || !out_bmax) return fail(RTX_ERR_BAD_ARG, "rtx_selftest_cull_boxes: null argument");
    const uint32_t n = desc->n_nodes, m = desc->n_prims;
    if (n && (!desc->node_bmin || !desc->node_bmax || !desc->node_right || !desc->node_prim))
        return fail(RTX_ERR_BAD_ARG, "rtx_selftest_cull_boxes: null node table");
    for (uint32_t i = 0; i < n; ++i) {   // what compute_cull_boxes relies on (rtx_scene_create checks more)
        const uint32_t prim = desc->node_prim[i], right = desc->node_right[i];
        if (prim != RTX_NONE) {
            if (prim >= m || !desc->prim_kind || !desc->prim_q || desc->prim_kind[prim] > RTX_TRIANGLE)
                return fail(RTX_ERR_BAD_ARG, "rtx_selftest_cull_boxes: bad leaf");
            if (desc->prim_kind[prim] != RTX_SPHERE && (!desc->prim_u || !desc->prim_v)) return fail(RTX_ERR_BAD_ARG, "rtx_selftest_cull_boxes: null planar table");
        } else if (right == RTX_NONE || right <= i + 1 || right >= n) {
            return fail(RTX_ERR_BAD_ARG, "rtx_selftest_cull_boxes: malformed tree");
        }
    }
    std::vector<DBox64> cull;
    const uint32_t differ = compute_cull_boxes(*desc, shrink != 0, cull);
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            out_bmin[(size_t)a * n + i] = cull[i].bmin[a];
            out_bmax[(size_t)a * n + i] = cull[i].bmax[a];
        }
    if (out_differ) *out_differ = differ;
    return RTX_OK;
}

int rtx_selftest_math(int op, const double* a, const double* b, uint64_t n, double* out) {
    if (n && (!a || !b || !out)) return fail(RTX_ERR_BAD_ARG, "rtx_selftest_math: null argument");
    if (n == 0) return RTX_OK;
    DevBuf<double> d;
    HIP_TRY(d.resize(3 * n));
    HIP_TRY(hipMemcpy(d.p, a, n * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d.p + n, b, n * sizeof(double), hipMemcpyHostToDevice));
    launch_selftest_math(op, d.p, d.p + n, n, d.p + 2 * n, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d.p + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost));
    return RTX_OK;
}

int rtx_selftest_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t n, double* out) {
    if (n && !out) return fail(RTX_ERR_BAD_ARG, "rtx_selftest_rng: null argument");
    if (n == 0) return RTX_OK;
    DevBuf<double> d;
    HIP_TRY(d.resize(n));
    launch_selftest_rng(seed, pixel, sample, n, d.p, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, d.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return RTX_OK;
}

}  // extern "C"
