// rtx_internal.h -- what the source files behind the C ABI share: error
// reporting, device buffers, the path pool and the scene handle.
#pragma once
#include "../../include/rtx.h"
#include "rtx_kernels.h"

#include <chrono>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace rtx {

// Records `msg` as the calling thread's last error (rtx_last_error) and returns `code`.
int api_fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            int code = (_e == hipErrorOutOfMemory) ? RTX_ERR_OOM : (_e == hipErrorNoDevice ? RTX_ERR_NO_DEVICE : RTX_ERR_HIP); \
            return ::rtx::api_fail(code, std::string(#expr) + ": " + hipGetErrorString(_e));              \
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
    // Grows to at least `count` elements (contents are not kept).  On failure the old buffer is still there.
    hipError_t resize(size_t count) {
        if (count <= n) return hipSuccess;
        T* q = nullptr;
        hipError_t e = hipMalloc((void**)&q, (count ? count : 1) * sizeof(T));
        if (e != hipSuccess) return e;
        release();
        p = q;
        n = count;
        return hipSuccess;
    }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
    }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// Sizes and counts from the environment: 0 is not a valid value and yields the default; switches pass zero_ok.
uint32_t env_u32(const char* name, uint32_t dflt, bool zero_ok = false);

// RTX_LOG_COLD=1: the host-side cost of what a process pays once (code object load, pool allocation, the first call's
// probe launch) goes to stderr, one "rtx cold: <what> <ms> ms" line each; bench.py's cold_frame reads them.
bool cold_log_on();
struct ColdTimer {
    std::chrono::steady_clock::time_point t0;
    const char* what;
    explicit ColdTimer(const char* w) : t0(std::chrono::steady_clock::now()), what(w) {}
    ~ColdTimer() {
        if (cold_log_on())
            std::fprintf(stderr, "rtx cold: %s %.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
};

// One lane of the render loop: a path pool (two ping-pong state buffers and
// the hit records), its control block and per-sub-pool counters, and the
// stream its rounds run on.  Lanes are independent; a render call runs
// several of them side by side so that one lane's closest-hit kernel (bound
// by VALU issue) overlaps another lane's shade kernel (bound by memory latency).
struct PathPool {
    size_t capacity = 0;
    DevBuf<double> buf_f64;     // 2 x 9 state arrays + hit_t
    DevBuf<uint32_t> buf_u32;   // 2 x 4 state arrays + hit_prim
    DevBuf<DRenderCtl> ctl;
    DevBuf<uint32_t> counters;  // per sub-pool: live, survivors[0], survivors[1]
    uint32_t counters_n = 0;
    DevBuf<DRenderParams> params;   // launch record of the single-launch render
    DState state[2]{};
    DHits hits{};
    hipStream_t stream = nullptr;            // owned
    hipEvent_t ev_poll[2] = {nullptr, nullptr}, ev_done = nullptr;
    uint32_t* h_poll = nullptr;              // two pinned words the host polls for the end of the lane's work
    std::vector<hipEvent_t> kernel_events;   // reused by RTX_FLAG_TIME_KERNELS, 3 per round
    // Launch records of the single-launch render are uploaded from pinned memory, so that a call may return before its
    // work has run (RTX_FLAG_NO_WAIT): a small ring of them, each guarded by an event that says its upload has executed.
    static constexpr uint32_t RING = 8;
    DRenderCtl* h_ctl_ring = nullptr;
    DRenderParams* h_params_ring = nullptr;
    hipEvent_t ring_ev[RING] = {};
    bool ring_used[RING] = {};
    uint32_t ring_next = 0;
    int ensure_ring();
    int ring_acquire(uint32_t* slot);   // waits until the slot's previous upload has executed

    ~PathPool();
    int ensure_host_side();
    int ensure_counters(uint32_t n_sub);
    int ensure_capacity(size_t n);
    int ensure_ctl();   // the control block alone (the resident render keeps no path in memory)
};

// Splits `need` slots into one sub-pool per wave of the launch grid: at most
// `max_waves` sub-pools (a multiple of the waves per block), each a multiple of 64 slots.
void pool_geometry(uint64_t need, uint32_t max_waves, uint32_t* n_sub, uint64_t* sub_capacity);

}  // namespace rtx

// A scene handle is used by one host thread at a time: every call on it shares
// the path pool, control block and events below.
struct rtx_scene {
    int device = 0;
    uint32_t n_cus = 256;
    uint32_t trace_max_waves = 5120;   // most waves a closest-hit batch is cut into (n_cus x RTX_WAVES_PER_CU, read when the scene is created)
    rtx::DScene d{};
    rtx::DevBuf<rtx::DNode> nodes;
    rtx::DevBuf<rtx::DSphere> spheres;
    rtx::DevBuf<rtx::DPlanar> planars;
    rtx::DevBuf<rtx::DNode32> nodes32;
    rtx::DevBuf<uint32_t> leaf_codes;
    rtx::DevBuf<rtx::DNode16> nodes16;
    rtx::DevBuf<rtx::DNode32> nodes32_ref;   // first-tier records of the reference's boxes, where the walk's cull boxes differ
    rtx::DevBuf<rtx::DNode16> nodes16_ref;
    rtx::DScene d_ref{};                      // `d` with those tables: what calls that count the reference's work walk
    uint32_t n_cull = 0;                      // nodes whose cull box is smaller than the reference's
    rtx::DevBuf<rtx::DBox64> boxes64;
    rtx::DevBuf<uint32_t> rec_user_id;
    rtx::DevBuf<rtx::DShadeRec> shade_recs;
    bool f32_ok = false;   // every box plane is small enough for the single-precision tier
    double rays_per_path = 0.0;   // of the scene's last render call that was waited for (0: none yet); picks the form of the next one
    // Lanes of the render loop, created on first use and kept.  Lane 0 also serves rtx_trace_closest.
    std::vector<std::unique_ptr<rtx::PathPool>> lanes;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    // Answers of the occupancy queries behind a render call's launch shape (kernel form, LDS mode or staged batch, shade
    // records in LDS or not), asked once per scene: a query costs tens of microseconds and a call made up to six.
    std::map<uint32_t, uint32_t> occupancy_cache;
    template <class F>
    uint32_t occupancy(uint32_t form, uint32_t shape, bool recs, F query) {
        const uint32_t key = (form << 16) | (shape << 1) | (recs ? 1u : 0u);
        auto it = occupancy_cache.find(key);
        if (it == occupancy_cache.end()) it = occupancy_cache.emplace(key, query()).first;
        return it->second;
    }

    rtx::PathPool* lane(size_t j) {
        while (lanes.size() <= j) lanes.emplace_back(new rtx::PathPool());
        return lanes[j].get();
    }
};

namespace rtx {

// Calls that count work: RTX_FLAG_COUNT_WORK counts the reference's box and surface tests (the walk then uses the
// reference's boxes in its first tier), RTX_FLAG_COUNT_ACTUAL the tests the product's walk really makes.
inline bool counting(uint32_t flags) { return (flags & (RTX_FLAG_COUNT_WORK | RTX_FLAG_COUNT_ACTUAL)) != 0u; }
inline DScene scene_tables(const rtx_scene* sc, uint32_t flags) {
    DScene d = (flags & RTX_FLAG_COUNT_ACTUAL) || !(flags & RTX_FLAG_COUNT_WORK) ? sc->d : sc->d_ref;
    d.assume_in_count = (flags & RTX_FLAG_COUNT_ACTUAL) ? 1u : 0u;
    return d;
}

// The boxes the walk's first tier tests (rtx_api.cpp), from a validated scene description; returns how many differ from the reference's.
uint32_t compute_cull_boxes(const rtx_scene_desc& desc, bool shrink, std::vector<DBox64>& cull);

// Closest hit of the rays in `rays` (per-component arrays) cut into `pool`'s sub-pools, with the box-test tier
// `flags` asks for; hit records go to `hits`.  `ctl` receives the work counters of RTX_FLAG_COUNT_WORK.
void run_traverse(const rtx_scene* sc, const DState& rays, const DHits& hits, const DPool& pool, double tmin, double tmax,
                  uint32_t flags, DRenderCtl* ctl, hipStream_t s);

// The device's copy of the scratch array of rtx_trace_closest_device; looked up once per device.
int noalloc_setup_current_device();

}  // namespace rtx
