// rt_capture.hip — Capture (/root/reference/src/ui.rs:82-93) on the device: the accumulation buffer becomes RGB8 where it
// lives, so a capture moves 3 bytes per pixel over PCIe instead of 16.
#include <hip/hip_runtime.h>

#include "rt_math.h"

namespace {

// one thread per pixel; 3 byte stores (the output is tiny next to the 16-byte read)
__global__ __launch_bounds__(256) void rt_capture_kernel(const float4 *__restrict__ accum, uint8_t *__restrict__ rgb, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    const float4 v = accum[i];
    rgb[3 * i + 0] = rt_capture_u8(v.x);
    rgb[3 * i + 1] = rt_capture_u8(v.y);
    rgb[3 * i + 2] = rt_capture_u8(v.z);
}

}  // namespace

hipError_t rt_launch_capture(const float *accum, uint8_t *rgb, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    rt_capture_kernel<<<dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream>>>(reinterpret_cast<const float4 *>(accum), rgb, n_pixels);
    return hipGetLastError();
}
