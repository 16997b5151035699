// Microbenchmark: is the L1 return path paid per lane-INSTRUCTION or per lane-DWORD?
// Every lane gathers from its own random 64-B line of a small (L1- or L2-resident) table with loads of
// 4, 8, 12 or 16 bytes.  Same number of load instructions in every variant.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_width.hip -o tools/microbench/gather_width.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W>
__global__ __launch_bounds__(256) void gather(const float *__restrict__ table, uint32_t mask_lines, int iters, float *out) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = s * 747796405u + 2891336453u;
            const float *p = table + ((s >> 8) & mask_lines) * 16u;
            if (W == 1) { acc += p[0]; }
            else if (W == 2) { float2 v = *reinterpret_cast<const float2 *>(p); acc += v.x + v.y; }
            else if (W == 3) { float v0 = p[0], v1 = p[1], v2 = p[2]; acc += v0 + v1 + v2; }   // compiler merges to dwordx3
            else { float4 v = *reinterpret_cast<const float4 *>(p); acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    float *out; (void)hipMalloc((void **)&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    for (uint32_t lines : {128u, 32768u}) {           // 8 KB (L1-resident) and 2 MB (L2-resident)
        std::vector<float> h(lines * 16, 1.0f);
        float *d; (void)hipMalloc((void **)&d, lines * 64);
        (void)hipMemcpy(d, h.data(), lines * 64, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; rep++)
            for (int w = 1; w <= 4; w++) {
                (void)hipEventRecord(e0);
                if (w == 1) hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, d, lines - 1, iters, out);
                if (w == 2) hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, d, lines - 1, iters, out);
                if (w == 3) hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, d, lines - 1, iters, out);
                if (w == 4) hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, d, lines - 1, iters, out);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double lane_loads = (double)blocks * 256 * iters * 4;
                if (rep) printf("table %5u KB, %2d-byte loads: %.3f ms, %.2f lane-loads/clk/CU (2.4 GHz), %.1f B/clk/CU\n", lines * 64 / 1024, 4 * w, ms,
                                lane_loads / (ms * 1e-3 * 2.4e9) / 256.0, lane_loads * 4 * w / (ms * 1e-3 * 2.4e9) / 256.0);
            }
        (void)hipFree(d);
    }
    return 0;
}
