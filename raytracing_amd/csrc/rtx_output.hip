// rtx_output.hip -- the image text of `InitializedCamera::render`
// (src/camera.rs:129-131,155-161) made on the device: "P3\n{W} {H}\n255\n",
// then one "{r} {g} {b}\n" line per pixel with ppm_pixel's 8-bit values
// (src/camera.rs:248-266), byte for byte what the reference prints.
// Lines have 6 to 12 bytes, so a pixel's place in the text is a prefix sum of
// line lengths: quantise + measure, scan, write.  Citations are into the reference.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/rtx.h"
#include "rtx_device.h"
#include "rtx_kernels.h"
#include "rtx_scan.h"

namespace rtx {
namespace output {

using scan::BLOCK;

// linear_to_gamma + ppm_pixel, src/camera.rs:248-266
__device__ __forceinline__ uint32_t ppm_value(double lin) {
    const double g = lin > 0.0 ? sqrt(lin) : 0.0;
    const double v = 255.999 * clampd(g, 0.0, 0.999);
    return (uint32_t)(v > 0.0 ? (v >= 255.0 ? 255.0 : v) : 0.0);
}
__device__ __forceinline__ uint32_t digits(uint32_t v) { return v >= 100u ? 3u : (v >= 10u ? 2u : 1u); }

struct DText {
    const double* rgb;    // [n][3] linear
    uint32_t n;           // pixels
    uint32_t* packed;     // [n] r | g << 8 | b << 16
    uint32_t* offset;     // [n + 1] byte offset of the pixel's line in the body
    char* text;           // body
};

__global__ void __launch_bounds__(BLOCK) k_quantize_pixels(DText T) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= T.n) return;
    const uint32_t r = ppm_value(T.rgb[3ull * i]), g = ppm_value(T.rgb[3ull * i + 1]), b = ppm_value(T.rgb[3ull * i + 2]);
    T.packed[i] = r | (g << 8) | (b << 16);
}

struct LengthScan {
    using T = uint32_t;
    DText D;
    __device__ T identity() const { return 0u; }
    __device__ T load(uint32_t i, uint32_t) const {
        const uint32_t p = D.packed[i];
        return digits(p & 255u) + digits((p >> 8) & 255u) + digits(p >> 16) + 3u;   // two blanks and the newline
    }
    __device__ T combine(const T& a, const T& b) const { return a + b; }
    __device__ void store(uint32_t i, uint32_t, const T& inc) const {
        D.offset[i + 1u] = inc;
        if (i == 0u) D.offset[0] = 0u;
    }
};

__device__ __forceinline__ char* put(char* at, uint32_t v, char end) {
    if (v >= 100u) *at++ = (char)('0' + v / 100u);
    if (v >= 10u) *at++ = (char)('0' + (v / 10u) % 10u);
    *at++ = (char)('0' + v % 10u);
    *at++ = end;
    return at;
}
__global__ void __launch_bounds__(BLOCK) k_write_lines(DText T) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= T.n) return;
    const uint32_t p = T.packed[i];
    char* at = T.text + T.offset[i];
    at = put(at, p & 255u, ' ');
    at = put(at, (p >> 8) & 255u, ' ');
    put(at, p >> 16, '\n');
}

}  // namespace output

// Writes the whole P3 text into host memory `out` (capacity bytes).  Returns an rtx_status; msg on failure.
int ppm_text_device(const double* d_rgb, uint32_t width, uint32_t height, char* out, uint64_t capacity, uint64_t* out_len, hipStream_t st,
                    std::string& msg) {
    using namespace output;
    const uint32_t n = width * height;
    char header[48];
    const int hl = std::snprintf(header, sizeof header, "P3\n%u %u\n255\n", width, height);
    char* base = nullptr;
    int code = RTX_OK;
    uint32_t body = 0;
    const size_t off_packed = 0, off_offset = ((size_t)n * 4u + 255u) & ~(size_t)255u;
    const size_t off_totals = (off_offset + ((size_t)n + 1u) * 4u + 255u) & ~(size_t)255u;
    const uint32_t n_tiles = (n + scan::TILE - 1u) / scan::TILE;
    const size_t off_text = (off_totals + (size_t)n_tiles * 4u + 255u) & ~(size_t)255u;
    const size_t bytes = off_text + 12ull * n;
    const uint32_t grid = (n + BLOCK - 1u) / BLOCK;
    DText T{};
    hipError_t e = hipMalloc((void**)&base, bytes);
    if (e != hipSuccess) {
        msg = std::string("rtx_ppm_text: hipMalloc: ") + hipGetErrorString(e);
        return e == hipErrorOutOfMemory ? RTX_ERR_OOM : RTX_ERR_HIP;
    }
    T.rgb = d_rgb;
    T.n = n;
    T.packed = reinterpret_cast<uint32_t*>(base + off_packed);
    T.offset = reinterpret_cast<uint32_t*>(base + off_offset);
    T.text = base + off_text;
    hipLaunchKernelGGL(k_quantize_pixels, dim3(grid), dim3(BLOCK), 0, st, T);
    scan::run_scan(LengthScan{T}, n, 1u, reinterpret_cast<uint32_t*>(base + off_totals), st);
    hipLaunchKernelGGL(k_write_lines, dim3(grid), dim3(BLOCK), 0, st, T);
    e = hipMemcpyAsync(&body, T.offset + n, sizeof body, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) {
        *out_len = (uint64_t)hl + body;
        if (*out_len > capacity) {
            msg = "rtx_ppm_text: output buffer too small";
            code = RTX_ERR_BAD_ARG;
        } else {
            std::memcpy(out, header, (size_t)hl);
            e = hipMemcpy(out + hl, T.text, body, hipMemcpyDeviceToHost);
        }
    }
    if (e != hipSuccess) {
        msg = std::string("rtx_ppm_text: ") + hipGetErrorString(e);
        code = RTX_ERR_HIP;
    }
    (void)hipFree(base);
    return code;
}

}  // namespace rtx
