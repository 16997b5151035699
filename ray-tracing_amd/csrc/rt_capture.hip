// rt_capture.hip — Capture (/root/reference/src/ui.rs:82-93) on the device: the accumulation buffer becomes RGB8 where it
// lives, so a capture moves 3 bytes per pixel over PCIe instead of 16.
#include <hip/hip_runtime.h>

#include "rt_device.h"
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

// rt_group (rt_abi.h): the shards' packed bands, gathered one after the other into `gathered`, become the image.
// Row y of the image lies in band y / 8, which belongs to shard (y / 8) % n and is that shard's local band (y / 8) / n.
__global__ __launch_bounds__(256) void rt_deinterleave_kernel(const float4 *__restrict__ gathered, float4 *__restrict__ image, RtGatherLayout L) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)L.width * L.height) return;
    const uint32_t y = (uint32_t)(i / L.width), x = (uint32_t)(i - (size_t)y * L.width);
    const uint32_t band = y >> 3, shard = band % L.n_shards, local_row = (band / L.n_shards) * 8u + (y & 7u);
    image[i] = gathered[((size_t)L.first_row[shard] + local_row) * L.width + x];
}

}  // namespace

hipError_t rt_launch_deinterleave(const float *gathered, float *image, const RtGatherLayout &L, hipStream_t stream) {
    const size_t n = (size_t)L.width * L.height;
    if (n == 0) return hipSuccess;
    rt_deinterleave_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream>>>(reinterpret_cast<const float4 *>(gathered),
                                                                                       reinterpret_cast<float4 *>(image), L);
    return hipGetLastError();
}

hipError_t rt_launch_capture(const float *accum, uint8_t *rgb, size_t n_pixels, hipStream_t stream) {
    if (n_pixels == 0) return hipSuccess;
    rt_capture_kernel<<<dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream>>>(reinterpret_cast<const float4 *>(accum), rgb, n_pixels);
    return hipGetLastError();
}
