// rtx_noalloc.hip -- rtx_trace_closest_device: the closest-hit batch for callers that cannot allocate.
//
// rtx_trace_closest stages a batch in buffers it allocates on demand (the rays as per-component arrays, the hit
// records, a control block).  Code under HIP graph capture, tight serving loops and harnesses that forbid hipMalloc
// cannot use it.  This entry point runs the same kernels on a fixed scratch array that is part of the library's
// code object: every GPU the library runs on holds one copy of it, at its own address, until the process ends.
#include "rtx_internal.h"

#include <mutex>

using namespace rtx;

namespace {

#ifdef RTX_SCRATCH_PROBE_RAYS   // measurement builds only: what does the size of the array add to the load of the code object?
constexpr uint64_t MAX_RAYS = RTX_SCRATCH_PROBE_RAYS;
#else
constexpr uint64_t MAX_RAYS = RTX_TRACE_DEVICE_MAX_RAYS;
#endif
constexpr size_t CTL_BYTES = 8 * sizeof(DRenderCtl);   // the control block and the debug records behind it
// layout: control block | ox oy oz dx dy dz t (MAX_RAYS doubles each) | surface codes (MAX_RAYS u32)
constexpr size_t OFF_F64 = (CTL_BYTES + 255) & ~(size_t)255;
constexpr size_t OFF_U32 = OFF_F64 + 7 * MAX_RAYS * sizeof(double);
constexpr size_t SCRATCH_BYTES = OFF_U32 + MAX_RAYS * sizeof(uint32_t);

// In host code this name is only a handle for hipGetSymbolAddress: the memory is reached through g_scratch[device].
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
std::mutex g_mutex;
unsigned char* g_scratch[MAX_DEVICES] = {};   // address of `scratch` on each device, once looked up

// out_id = the caller's index of the surface (or NONE), out_t = hit.t (+inf on a miss)
__global__ __launch_bounds__(256) void k_finish_hits(const uint32_t* rec_user_id, uint32_t n_spheres, const uint32_t* code, const double* t,
                                                     uint64_t n, uint32_t* out_id, double* out_t) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint32_t c = code[i];
        const bool hit = c != NONE;
        out_id[i] = hit ? rec_user_id[(code_geom(c) != 0u ? n_spheres : 0u) + code_slot(c)] : NONE;
        out_t[i] = hit ? t[i] : __builtin_inf();
    }
}

}  // namespace

int rtx::noalloc_setup_current_device() {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return api_fail(RTX_ERR_BAD_ARG, "rtx_trace_closest_device: device index beyond 63");
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_scratch[dev]) return RTX_OK;
    void* p = nullptr;
    {
        ColdTimer cold("hipGetSymbolAddress of the scratch array (loads the library's code object)");
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    }
    if (!p) return api_fail(RTX_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
    g_scratch[dev] = static_cast<unsigned char*>(p);
    return RTX_OK;
}

extern "C" int rtx_trace_closest_device(rtx_scene* sc, const double* d_org, const double* d_dir, uint64_t n, double tmin, double tmax,
                                        uint32_t flags, uint32_t* d_out_id, double* d_out_t, void* stream) {
    if (!sc || (n && (!d_org || !d_dir || !d_out_id || !d_out_t))) return api_fail(RTX_ERR_BAD_ARG, "rtx_trace_closest_device: null argument");
    if (n > MAX_RAYS)
        return api_fail(RTX_ERR_BAD_ARG, "rtx_trace_closest_device: " + std::to_string(n) + " rays, the scratch array holds at most " +
                                             std::to_string(MAX_RAYS) + " (RTX_TRACE_DEVICE_MAX_RAYS); split the batch");
    if (counting(flags) || (flags & RTX_FLAG_TIME_KERNELS))
        return api_fail(RTX_ERR_BAD_ARG, "rtx_trace_closest_device: counting and timing flags need rtx_trace_closest (there is no stats output here)");
    if (n == 0) return RTX_OK;
    // The caller's current device is left as it was: a capture or a serving loop on another device goes on undisturbed.
    struct DeviceGuard {
        int prior = -1;
        ~DeviceGuard() { if (prior >= 0) (void)hipSetDevice(prior); }
    } guard;
    int current = 0;
    HIP_TRY(hipGetDevice(&current));
    if (current != sc->device) {
        HIP_TRY(hipSetDevice(sc->device));
        guard.prior = current;
    }
    int rc = noalloc_setup_current_device();   // a lookup; the scene's creation has normally done it already
    if (rc != RTX_OK) return rc;
    unsigned char* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        base = g_scratch[sc->device];
    }
    hipStream_t s = (hipStream_t)stream;

    DRenderCtl* ctl = reinterpret_cast<DRenderCtl*>(base);
    double* f = reinterpret_cast<double*>(base + OFF_F64);
    DState rays{};   // the walk reads origin and direction only
    rays.ox = f; rays.oy = f + n; rays.oz = f + 2 * n;
    rays.dx = f + 3 * n; rays.dy = f + 4 * n; rays.dz = f + 5 * n;
    DHits hits{};
    hits.t = f + 6 * n;
    hits.prim = reinterpret_cast<uint32_t*>(base + OFF_U32);

    uint32_t n_sub;
    uint64_t sub_capacity;
    pool_geometry(n, sc->trace_max_waves, &n_sub, &sub_capacity);
    const DPool pool{nullptr, n_sub, (uint32_t)sub_capacity, (uint32_t)n};

    // The array keeps what the last call left: the control block starts at zero, the rest is overwritten before it is read.
    HIP_TRY(hipMemsetAsync(ctl, 0, CTL_BYTES, s));
    launch_rays_to_state(d_org, d_dir, n, rays, s);
    run_traverse(sc, rays, hits, pool, tmin, tmax, flags, ctl, s);
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_finish_hits, dim3((uint32_t)blocks), dim3(256), 0, s, sc->d.rec_user_id, sc->d.n_spheres, hits.prim, hits.t, n, d_out_id, d_out_t);
    HIP_TRY(hipGetLastError());
    return RTX_OK;
}
