// rtx_multi.cpp -- one frame on several GPUs of a node from ONE process: rtx_multi_* and rtx_render_multi.
//
// The sample loop of InitializedCamera::render (src/camera.rs:133-151) shards by sample index: device g renders
// samples [begin_g, begin_g + count_g) of every pixel with the whole scene replicated in its own HBM, nothing is
// exchanged while rendering, and the per-device f64 accumulation buffers are summed onto device 0 with ONE RCCL
// reduce over xGMI (the reference's counterpart is the per-pixel `.reduce(|| ZERO, |a, b| a + b)` of
// src/camera.rs:147-148).  One host thread per device drives its render call; the calling thread then issues every rank of the collective as one group.
// RCCL is loaded on first use (dlopen), so the library does not depend on it for single-GPU callers.
#include "rtx_internal.h"

#include <rccl/rccl.h>   // types and prototypes only; the functions are looked up at run time

#include <dlfcn.h>

#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

using namespace rtx;

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // by soname first: a process that already holds a copy (PyTorch ships one) keeps using that copy
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            const char* e = dlerror();
            r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
#define RTX_SYM(field, name)                                                        \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));               \
    if (!r.field && r.error.empty()) r.error = std::string("librccl has no symbol ") + name
        RTX_SYM(CommInitAll, "ncclCommInitAll");
        RTX_SYM(CommDestroy, "ncclCommDestroy");
        RTX_SYM(CommCount, "ncclCommCount");
        RTX_SYM(Reduce, "ncclReduce");
        RTX_SYM(GroupStart, "ncclGroupStart");
        RTX_SYM(GroupEnd, "ncclGroupEnd");
        RTX_SYM(GetErrorString, "ncclGetErrorString");
#undef RTX_SYM
    });
    return r;
}

struct Rank {
    int device = 0;
    rtx_scene* scene = nullptr;
    hipStream_t stream = nullptr;
    DevBuf<double> accum;    // [H][W][3] radiance sums of this device's samples
    size_t accum_values = 0;
    ncclComm_t comm = nullptr;
};

}  // namespace

struct rtx_multi {
    std::vector<Rank> ranks;
    int prior_device = 0;
    bool rehearsal = false;   // RTX_MULTI_SHARE_GPU=1: more ranks than devices, no communicator (see rtx_multi_create)
    DevBuf<double> staging;   // rehearsal: a rank's buffer brought to device 0 to be added there
    explicit rtx_multi(size_t n) : ranks(n) {}
};

namespace {

// Samples [begin, begin + count) of rank `rank` of `world`: contiguous, disjoint, lengths differ by at most one
// (500 over 8 ranks -> 63, 63, 63, 63, 62, 62, 62, 62).
void spp_range(uint32_t rank, uint32_t world, uint32_t spp, uint32_t* begin, uint32_t* count) {
    const uint32_t base = spp / world, rem = spp % world;
    *begin = rank * base + (rank < rem ? rank : rem);
    *count = base + (rank < rem ? 1u : 0u);
}

void destroy(rtx_multi* m) {
    if (!m) return;
    for (Rank& r : m->ranks) {
        (void)hipSetDevice(r.device);
        if (r.comm && rccl().CommDestroy) (void)rccl().CommDestroy(r.comm);
        if (r.stream) (void)hipStreamDestroy(r.stream);
        r.accum.release();
        if (r.scene) rtx_scene_destroy(r.scene);
    }
    if (!m->ranks.empty()) (void)hipSetDevice(m->ranks[0].device);
    m->staging.release();
    (void)hipSetDevice(m->prior_device);
    delete m;
}

}  // namespace

extern "C" {

int rtx_multi_create(const rtx_scene_desc* desc, int n_gpus, rtx_multi** out) {
    if (!desc || !out) return api_fail(RTX_ERR_BAD_ARG, "rtx_multi_create: null argument");
    *out = nullptr;
    const int have = rtx_device_count();
    if (have <= 0) return api_fail(RTX_ERR_NO_DEVICE, "rtx_multi_create: no HIP device");
    // Rehearsal of the N-rank path on fewer than N devices (RTX_MULTI_SHARE_GPU=1; tests only, never a measurement): rank g runs on
    // device g modulo the devices there are, with its own scene handle, stream and host thread like any rank, and because RCCL
    // refuses two ranks on one device the sum onto rank 0 is made by plain copies and adds instead of the collective.
    const bool rehearsal = env_u32("RTX_MULTI_SHARE_GPU", 0, true) != 0u && n_gpus > have;
    if (n_gpus < 1 || (n_gpus > have && !rehearsal) || n_gpus > RTX_MULTI_MAX_GPUS)
        return api_fail(RTX_ERR_BAD_ARG, "rtx_multi_create: n_gpus = " + std::to_string(n_gpus) + ", " + std::to_string(have) +
                                             " device(s) visible, at most " + std::to_string(RTX_MULTI_MAX_GPUS) + " supported");
    Rccl& nc = rccl();
    if (!rehearsal && !nc.error.empty()) return api_fail(RTX_ERR_RCCL, "rtx_multi_create: " + nc.error);
    rtx_multi* m = new rtx_multi((size_t)n_gpus);
    m->rehearsal = rehearsal;
    (void)hipGetDevice(&m->prior_device);
    auto bail = [&](int code) {
        const std::string msg = rtx_last_error();   // the clean-up below must not replace the message
        destroy(m);
        return api_fail(code, msg);
    };
    std::vector<int> devs((size_t)n_gpus);
    for (int g = 0; g < n_gpus; ++g) {
        Rank& r = m->ranks[(size_t)g];
        r.device = devs[(size_t)g] = g % have;
        if (hipSetDevice(r.device) != hipSuccess) return bail(api_fail(RTX_ERR_HIP, "rtx_multi_create: hipSetDevice(" + std::to_string(r.device) + ") failed"));
        int rc = rtx_scene_create(desc, &r.scene);   // the whole scene in every device's HBM
        if (rc != RTX_OK) return bail(rc);
        if (hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess)
            return bail(api_fail(RTX_ERR_HIP, "rtx_multi_create: hipStreamCreate failed"));
    }
    // The communicator clique is set up here, once, and kept: it takes hundreds of milliseconds, more than a frame.
    if (!rehearsal) {
        std::vector<ncclComm_t> comms((size_t)n_gpus, nullptr);
        const ncclResult_t e = nc.CommInitAll(comms.data(), n_gpus, devs.data());
        if (e != ncclSuccess) return bail(api_fail(RTX_ERR_RCCL, std::string("ncclCommInitAll: ") + nc.GetErrorString(e)));
        for (int g = 0; g < n_gpus; ++g) m->ranks[(size_t)g].comm = comms[(size_t)g];
    }
    (void)hipSetDevice(m->prior_device);
    *out = m;
    return RTX_OK;
}

void rtx_multi_destroy(rtx_multi* m) { destroy(m); }

int rtx_multi_render(rtx_multi* m, const rtx_camera* cam, uint64_t seed, uint32_t flags, double* out_rgb, rtx_multi_stats* stats) {
    if (!m || !cam) return api_fail(RTX_ERR_BAD_ARG, "rtx_multi_render: null argument");
    if (cam->image_width == 0 || cam->image_height == 0) return api_fail(RTX_ERR_BAD_ARG, "rtx_multi_render: empty image");
    if (stats) std::memset(stats, 0, sizeof *stats);
    const uint32_t world = (uint32_t)m->ranks.size();
    const size_t nval = 3ull * cam->image_width * cam->image_height;
    Rccl& nc = rccl();
    int entry_device = 0;   // the calling thread's current device is the same after the call as before it
    (void)hipGetDevice(&entry_device);
    struct Restore {
        int device;
        ~Restore() { (void)hipSetDevice(device); }
    } restore{entry_device};

    // set-up that is not part of the frame: accumulation buffers of the right size on every device
    for (Rank& r : m->ranks) {
        HIP_TRY(hipSetDevice(r.device));
        HIP_TRY(r.accum.resize(nval));
        r.accum_values = nval;
    }

    struct Result {
        int rc = RTX_OK;
        std::string msg;
        rtx_stats st{};
        uint32_t begin = 0, count = 0;
        double ms_render = 0.0;
    };
    std::vector<Result> res(world);
    const auto t0 = std::chrono::steady_clock::now();
    // Phase 1, one host thread per device: clear the device's buffer and render its sample range.
    auto work = [&](uint32_t g) {
        Rank& r = m->ranks[g];
        Result& R = res[g];
        auto failed = [&](int rc, const std::string& msg) {
            R.rc = rc;
            R.msg = msg;
        };
        spp_range(g, world, cam->samples_per_pixel, &R.begin, &R.count);
        if (hipSetDevice(r.device) != hipSuccess) return failed(RTX_ERR_HIP, "hipSetDevice failed");
        if (hipMemsetAsync(r.accum.p, 0, nval * sizeof(double), r.stream) != hipSuccess) return failed(RTX_ERR_HIP, "hipMemsetAsync failed");
        const auto a = std::chrono::steady_clock::now();
        if (R.count) {
            const int rc = rtx_render_accumulate(r.scene, cam, seed, R.begin, R.count, flags, r.accum.p, r.stream, &R.st);
            if (rc != RTX_OK) return failed(rc, rtx_last_error());
        } else if (hipStreamSynchronize(r.stream) != hipSuccess) {
            return failed(RTX_ERR_HIP, "hipStreamSynchronize failed");
        }
        R.ms_render = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    };
    std::vector<std::thread> threads;
    for (uint32_t g = 1; g < world; ++g) threads.emplace_back(work, g);
    work(0u);
    for (std::thread& t : threads) t.join();
    // A device that failed would leave the others waiting in the collective for ever: the reduce starts only when all have rendered.
    for (uint32_t g = 0; g < world; ++g)
        if (res[g].rc != RTX_OK) {
            return api_fail(res[g].rc, "rtx_multi_render: device " + std::to_string(m->ranks[g].device) + ": " + res[g].msg);
        }
    // Phase 2, the one collective of the frame: every device's rank of the sum onto device 0, as one group.
    const auto t_reduce = std::chrono::steady_clock::now();
    int comm_ranks = 0;
    if (m->rehearsal) {
        // no communicator (see rtx_multi_create): every other rank's buffer is brought to device 0 and added there, in rank order
        Rank& r0 = m->ranks[0];
        HIP_TRY(hipSetDevice(r0.device));
        HIP_TRY(m->staging.resize(nval));
        comm_ranks = 1;
        for (uint32_t g = 1; g < world; ++g) {
            Rank& r = m->ranks[g];
            const double* src = r.accum.p;
            if (r.device != r0.device) {
                HIP_TRY(hipMemcpyPeerAsync(m->staging.p, r0.device, r.accum.p, r.device, nval * sizeof(double), r0.stream));
                src = m->staging.p;
            }
            launch_add(src, nval, r0.accum.p, r0.stream);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(r0.stream));   // (the staging buffer is reused by the next rank)
            ++comm_ranks;
        }
    } else {
        (void)nc.CommCount(m->ranks[0].comm, &comm_ranks);
        ncclResult_t e = nc.GroupStart();
        for (uint32_t g = 0; g < world && e == ncclSuccess; ++g) {
            Rank& r = m->ranks[g];
            e = nc.Reduce(r.accum.p, r.accum.p, nval, ncclDouble, ncclSum, 0, r.comm, r.stream);
        }
        const ncclResult_t e_end = nc.GroupEnd();
        if (e == ncclSuccess) e = e_end;
        if (e != ncclSuccess) {
            return api_fail(RTX_ERR_RCCL, std::string("rtx_multi_render: ncclReduce: ") + nc.GetErrorString(e));
        }
    }
    HIP_TRY(hipSetDevice(m->ranks[0].device));
    // `* self.pixel_samples_scale` (src/camera.rs:146) on the summed image
    int rc = rtx_scale(m->ranks[0].accum.p, nval, cam->pixel_samples_scale, m->ranks[0].accum.p, m->ranks[0].stream);
    if (rc != RTX_OK) return rc;
    for (uint32_t g = 0; g < world; ++g) {
        HIP_TRY(hipSetDevice(m->ranks[g].device));
        HIP_TRY(hipStreamSynchronize(m->ranks[g].stream));
    }
    const auto t1 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(m->ranks[0].device));
    if (stats) {
        stats->n_gpus = world;
        stats->collective_ranks = (uint32_t)comm_ranks;
        stats->ms_frame = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->ms_reduce = std::chrono::duration<double, std::milli>(t1 - t_reduce).count();
        for (uint32_t g = 0; g < world; ++g) {
            stats->rays += res[g].st.rays;
            stats->paths += res[g].st.paths;
            stats->spp_begin[g] = res[g].begin;
            stats->spp_count[g] = res[g].count;
            stats->ms_render[g] = res[g].ms_render;
            stats->ms_device[g] = res[g].st.ms_total;
        }
    }
    if (out_rgb) HIP_TRY(hipMemcpy(out_rgb, m->ranks[0].accum.p, nval * sizeof(double), hipMemcpyDeviceToHost));
    return RTX_OK;
}

const double* rtx_multi_image(const rtx_multi* m) { return (m && !m->ranks.empty()) ? m->ranks[0].accum.p : nullptr; }

int rtx_render_multi(const rtx_scene_desc* desc, const rtx_camera* cam, uint64_t seed, int n_gpus, uint32_t flags, double* out_rgb,
                     rtx_multi_stats* stats) {
    if (!out_rgb) return api_fail(RTX_ERR_BAD_ARG, "rtx_render_multi: null output");
    rtx_multi* m = nullptr;
    int rc = rtx_multi_create(desc, n_gpus, &m);
    if (rc != RTX_OK) return rc;
    rc = rtx_multi_render(m, cam, seed, flags, out_rgb, stats);
    const std::string msg = rc != RTX_OK ? rtx_last_error() : "";
    rtx_multi_destroy(m);
    return rc != RTX_OK ? api_fail(rc, msg) : RTX_OK;
}

}  // extern "C"
