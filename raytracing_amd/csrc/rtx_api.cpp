// rtx_api.cpp -- the C ABI declared in include/rtx.h, on top of the kernels.
//
// Host-side responsibilities: validate and upload the flattened BVH, own the
// path pool, drive the traverse/shade rounds, and report counters and times.
#include "../../include/rtx.h"
#include "rtx_kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace rtx;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            int code = (_e == hipErrorOutOfMemory) ? RTX_ERR_OOM : (_e == hipErrorNoDevice ? RTX_ERR_NO_DEVICE : RTX_ERR_HIP); \
            return fail(code, std::string(#expr) + ": " + hipGetErrorString(_e));                         \
        }                                                                                                  \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t resize(size_t count) {
        if (count <= n) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    char* end = nullptr;
    unsigned long long x = std::strtoull(v, &end, 10);
    if (end == v || x == 0 || x > 0xFFFFFFFFull) return dflt;
    return (uint32_t)x;
}

}  // namespace

struct rtx_scene {
    int device = 0;
    uint32_t n_cus = 256;
    DScene d{};
    DevBuf<DNode> nodes;
    DevBuf<DSphere> spheres;
    DevBuf<DPlanar> planars;
    DevBuf<DPrimInfo> prims;
    DevBuf<DMaterial> materials;
    DevBuf<DNode32> nodes32;
    DevBuf<DBox64> boxes64;
    DevBuf<uint32_t> sphere_prim, planar_prim;
    bool f32_ok = false;   // every box plane is small enough for the single-precision tier
    // path state (two buffers, ping-pong) and hit records, sized on first use and kept
    size_t capacity = 0;
    DevBuf<double> buf_f64;     // 2 x 9 state arrays + hit_t
    DevBuf<uint32_t> buf_u32;   // 2 x 4 state arrays + hit_prim
    DevBuf<DRenderCtl> ctl;
    DState state[2]{};
    DHits hits{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> kernel_events;   // reused by RTX_FLAG_TIME_KERNELS, 3 per round
    uint32_t* h_count = nullptr;             // pinned word the render loop polls

    int ensure_capacity(size_t n) {
        if (n <= capacity) return RTX_OK;
        HIP_TRY(buf_f64.resize(19 * n));
        HIP_TRY(buf_u32.resize(9 * n));
        HIP_TRY(ctl.resize(1));
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
};

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
    std::vector<DBox64> boxes64(n);
    std::vector<uint32_t> sphere_prim, planar_prim;
    for (uint32_t i = 0; i < m; ++i) (prims[i].kind == RTX_SPHERE ? sphere_prim : planar_prim).push_back(i);
    bool f32_ok = true;
    std::vector<uint32_t> miss(n, NONE);
    bool filter_ok = true;
    if (n) miss[0] = n;
    for (uint32_t i = 0; i < n; ++i) {
        DNode& nd = nodes[i];
        for (int a = 0; a < 3; ++a) {
            nd.bmin[a] = desc->node_bmin[(size_t)a * n + i];
            nd.bmax[a] = desc->node_bmax[(size_t)a * n + i];
            if (!(std::fabs(nd.bmin[a]) <= 0x1.0p300) || !(std::fabs(nd.bmax[a]) <= 0x1.0p300)) filter_ok = false;
            if (!(std::fabs(nd.bmin[a]) <= 1e15) || !(std::fabs(nd.bmax[a]) <= 1e15)) f32_ok = false;
            boxes64[i].bmin[a] = nd.bmin[a];
            boxes64[i].bmax[a] = nd.bmax[a];
            nodes32[i].plane[2 * a] = (float)nd.bmin[a];
            nodes32[i].plane[2 * a + 1] = (float)nd.bmax[a];
        }
        if (miss[i] == NONE) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: node " + std::to_string(i) + " is not reachable in pre-order");
        if (miss[i] <= i || miss[i] > n) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: malformed tree");
        nd.miss = miss[i];
        uint32_t prim = desc->node_prim[i];
        uint32_t right = desc->node_right[i];
        if (prim != RTX_NONE) {   // Node::Leaf
            if (prim >= m) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: node_prim out of range");
            if (right != RTX_NONE) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: leaf with a right child");
            if (miss[i] != i + 1) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: leaf subtree is not a single node");
            nd.prim = prim;
            nd.kind = prims[prim].kind;
            nd.slot = prims[prim].slot;
            if (nd.slot >= (1u << 30)) return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: more than 2^30 surfaces of one kind");
            nodes32[i].leaf = (nd.kind << 30) | nd.slot;
        } else {                  // Node::Internal(Some(right_idx), box)
            if (right == RTX_NONE || right <= i + 1 || right >= miss[i])
                return fail(RTX_ERR_BAD_ARG, "rtx_scene_create: right child index out of order at node " + std::to_string(i));
            nd.prim = NONE;
            nd.kind = 0;
            nd.slot = 0;
            nodes32[i].leaf = NONE;
            miss[i + 1] = right;
            miss[right] = miss[i];
        }
        nodes32[i].miss = nd.miss;
    }

    rtx_scene* sc = new rtx_scene();
    auto bail = [&](int code) {
        delete sc;
        return code;
    };
    hipError_t e = hipGetDevice(&sc->device);
    if (e != hipSuccess) return bail(fail(RTX_ERR_NO_DEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e)));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sc->device) == hipSuccess && prop.multiProcessorCount > 0) sc->n_cus = (uint32_t)prop.multiProcessorCount;

#define UP(buf, vec)                                                                                                   \
    do {                                                                                                               \
        hipError_t _e = sc->buf.resize((vec).size());                                                                  \
        if (_e == hipSuccess && !(vec).empty()) _e = hipMemcpy(sc->buf.p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice); \
        if (_e != hipSuccess) return bail(fail(_e == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP, std::string("scene upload: ") + hipGetErrorString(_e))); \
    } while (0)
    UP(nodes, nodes);
    UP(spheres, spheres);
    UP(planars, planars);
    UP(prims, prims);
    UP(materials, mats);
    UP(nodes32, nodes32);
    UP(boxes64, boxes64);
    UP(sphere_prim, sphere_prim);
    UP(planar_prim, planar_prim);
#undef UP
    sc->d.nodes = sc->nodes.p;
    sc->d.spheres = sc->spheres.p;
    sc->d.planars = sc->planars.p;
    sc->d.prims = sc->prims.p;
    sc->d.materials = sc->materials.p;
    sc->d.n_nodes = n;
    sc->d.n_prims = m;
    sc->d.filter_ok = filter_ok ? 1u : 0u;
    sc->d.nodes32 = sc->nodes32.p;
    sc->d.boxes64 = sc->boxes64.p;
    sc->d.sphere_prim = sc->sphere_prim.p;
    sc->d.planar_prim = sc->planar_prim.p;
    sc->f32_ok = f32_ok && filter_ok;
    if (hipEventCreate(&sc->ev0) != hipSuccess || hipEventCreate(&sc->ev1) != hipSuccess)
        return bail(fail(RTX_ERR_HIP, "hipEventCreate failed"));
    *out = sc;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* sc) {
    if (!sc) return;
    if (sc->ev0) (void)hipEventDestroy(sc->ev0);
    if (sc->ev1) (void)hipEventDestroy(sc->ev1);
    for (hipEvent_t e : sc->kernel_events) (void)hipEventDestroy(e);
    if (sc->h_count) (void)hipHostFree(sc->h_count);
    delete sc;
}

// Which closest-hit kernel a call uses: the reference's division form when the
// caller asks for it, else the tiered test with the f32 records in LDS (small
// trees), from global memory (large trees), or the f64-only filter (scenes
// whose box planes are too large for f32, or on request).
static void run_traverse(rtx_scene* sc, const DState& rays, const uint32_t* count_ptr, uint64_t max_count, double tmin, double tmax,
                         uint32_t flags, uint32_t max_blocks, hipStream_t s) {
    const bool count_work = (flags & RTX_FLAG_COUNT_WORK) != 0;
    const uint32_t lds_limit = env_u32("RTX_LDS_NODE_BYTES", 40u * 1024u);
    if (flags & RTX_FLAG_EXACT_DIVIDE) {
        launch_traverse(sc->d, rays, sc->hits, count_ptr, max_count, tmin, tmax, sc->ctl.p, count_work, true, max_blocks, s);
    } else if ((flags & RTX_FLAG_F64_FILTER) || !sc->f32_ok) {
        launch_traverse(sc->d, rays, sc->hits, count_ptr, max_count, tmin, tmax, sc->ctl.p, count_work, false, max_blocks, s);
    } else {
        const bool lds = !(flags & RTX_FLAG_NO_LDS) && (size_t)sc->d.n_nodes * sizeof(DNode32) <= lds_limit;
        launch_traverse_f32(sc->d, rays, sc->hits, count_ptr, max_count, tmin, tmax, sc->ctl.p, count_work, lds, max_blocks, s);
    }
}

static int read_counters(rtx_scene* sc, hipStream_t s, rtx_stats* stats) {
    DRenderCtl h;
    HIP_TRY(hipMemcpyAsync(&h, sc->ctl.p, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    stats->rays += h.rays;
    stats->node_visits += h.node_visits;
    stats->sphere_tests += h.sphere_tests;
    stats->planar_tests += h.planar_tests;
    return RTX_OK;
}

int rtx_trace_closest(rtx_scene* sc, const double* org, const double* dir, uint64_t n, double tmin, double tmax,
                      uint32_t flags, uint32_t* out_id, double* out_t, rtx_stats* stats) {
    if (!sc || (n && (!org || !dir || !out_id || !out_t))) return fail(RTX_ERR_BAD_ARG, "rtx_trace_closest: null argument");
    if (n > 0xFFFFFFFFull) return fail(RTX_ERR_BAD_ARG, "rtx_trace_closest: more than 2^32-1 rays in one call");
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (n == 0) return RTX_OK;
    HIP_TRY(hipSetDevice(sc->device));
    int rc = sc->ensure_capacity((size_t)n);
    if (rc != RTX_OK) return rc;
    hipStream_t s = nullptr;
    DevBuf<double> d_rays;
    HIP_TRY(d_rays.resize(6 * n));
    HIP_TRY(hipMemcpyAsync(d_rays.p, org, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_rays.p + 3 * n, dir, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
    DRenderCtl h{};
    h.count_live = (uint32_t)n;
    h.rays = n;
    HIP_TRY(hipMemcpyAsync(sc->ctl.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
    launch_rays_to_state(d_rays.p, d_rays.p + 3 * n, n, sc->state[0], s);
    const uint32_t max_blocks = sc->n_cus * env_u32("RTX_BLOCKS_PER_CU", 8);
    HIP_TRY(hipEventRecord(sc->ev0, s));
    run_traverse(sc, sc->state[0], &sc->ctl.p->count_live, n, tmin, tmax, flags, max_blocks, s);
    HIP_TRY(hipEventRecord(sc->ev1, s));
    launch_user_ids(sc->d.prims, sc->hits.prim, n, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out_id, sc->hits.prim, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(out_t, sc->hits.t, n * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (uint64_t i = 0; i < n; ++i)
        if (out_id[i] == RTX_NONE) out_t[i] = INFINITY;
    if (stats) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev0, sc->ev1));
        stats->ms_total = ms;
        stats->ms_traverse = ms;
        stats->iterations = 1;
        rc = read_counters(sc, s, stats);
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
    HIP_TRY(hipSetDevice(sc->device));
    hipStream_t s = (hipStream_t)stream;

    const uint64_t want = env_u32("RTX_POOL_SLOTS", 1u << 22);
    const uint32_t slots = (uint32_t)(total < want ? total : want);
    rc = sc->ensure_capacity(slots);
    if (rc != RTX_OK) return rc;
    const uint32_t max_blocks = sc->n_cus * env_u32("RTX_BLOCKS_PER_CU", 8);
    const uint32_t check_every = env_u32("RTX_CHECK_EVERY", 4);
    const bool time_kernels = (flags & RTX_FLAG_TIME_KERNELS) != 0;
    const DCamera dcam = to_dcamera(cam);

    // Round r works on state buffer r & 1 (all on one stream, no host round trip inside a round):
    //   generate : start new paths behind the survivors until the buffer is full
    //   traverse : closest hit of every ray in the buffer
    //   shade    : scatter / terminate; survivors are written densely into the other buffer
    DRenderCtl h{};
    h.total_paths = total;
    HIP_TRY(hipMemcpyAsync(sc->ctl.p, &h, sizeof(h), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(sc->ev0, s));
    DRenderCtl* ctl = sc->ctl.p;

    std::vector<hipEvent_t>& evs = sc->kernel_events;   // only with RTX_FLAG_TIME_KERNELS: 3 per round
    size_t n_evs = 0;
    if (!sc->h_count) HIP_TRY(hipHostMalloc((void**)&sc->h_count, sizeof(uint32_t), hipHostMallocDefault));
    uint32_t* h_count = sc->h_count;
    *h_count = slots;
    uint64_t iters = 0;
    int status = RTX_OK;
    for (;;) {
        const uint32_t cur = (uint32_t)(iters & 1u), nxt = cur ^ 1u;
        hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
        if (time_kernels) {
            while (evs.size() < n_evs + 3) {
                hipEvent_t e = nullptr;
                if (hipEventCreate(&e) != hipSuccess) break;
                evs.push_back(e);
            }
            if (evs.size() < n_evs + 3) {
                status = fail(RTX_ERR_HIP, "hipEventCreate failed");
                break;
            }
            e0 = evs[n_evs]; e1 = evs[n_evs + 1]; e2 = evs[n_evs + 2];
            n_evs += 3;
        }
        launch_generate(dcam, sc->state[cur], slots, ctl, cur, seed, spp_begin, max_blocks, s);
        if (time_kernels) (void)hipEventRecord(e0, s);
        run_traverse(sc, sc->state[cur], &ctl->count_live, slots, 0.001, INFINITY, flags, max_blocks, s);   // Interval(0.001, inf), camera.rs:194
        if (time_kernels) (void)hipEventRecord(e1, s);
        launch_shade(sc->d, dcam, sc->state[cur], sc->hits, &ctl->count_live, sc->state[nxt], &ctl->count_surv[nxt], slots, seed,
                     d_accum, max_blocks, s);
        if (time_kernels) (void)hipEventRecord(e2, s);
        ++iters;
        if (iters % check_every == 0) {
            // finished when a round had nothing to trace: no survivors and no paths left to start
            hipError_t e = hipMemcpyAsync(h_count, &ctl->count_live, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) {
                status = fail(RTX_ERR_HIP, std::string("render loop: ") + hipGetErrorString(e));
                break;
            }
            if (*h_count == 0) break;
        }
    }
    if (status == RTX_OK) {
        hipError_t e = hipEventRecord(sc->ev1, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) status = fail(RTX_ERR_HIP, std::string("render: ") + hipGetErrorString(e));
    }
    if (status == RTX_OK && stats) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, sc->ev0, sc->ev1);
        stats->ms_total = ms;
        stats->paths = total;
        stats->iterations = iters;
        for (size_t i = 0; i + 2 < n_evs; i += 3) {
            float a = 0.f, b = 0.f, g = 0.f;
            (void)hipEventElapsedTime(&a, evs[i], evs[i + 1]);
            (void)hipEventElapsedTime(&b, evs[i + 1], evs[i + 2]);
            (void)hipEventElapsedTime(&g, i == 0 ? sc->ev0 : evs[i - 1], evs[i]);   // includes the host's poll every few rounds
            stats->ms_traverse += a;
            stats->ms_shade += b;
            stats->ms_generate += g;
        }
        status = read_counters(sc, s, stats);
    }
    return status;
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

int rtx_render(rtx_scene* sc, const rtx_camera* cam, uint64_t seed, uint32_t flags, double* out_rgb, rtx_stats* stats) {
    if (!sc || !out_rgb) return fail(RTX_ERR_BAD_ARG, "rtx_render: null argument");
    int rc = check_camera(cam, "rtx_render");
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
    HIP_TRY(hipMemcpy(out_rgb, acc.p, nval * sizeof(double), hipMemcpyDeviceToHost));
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
