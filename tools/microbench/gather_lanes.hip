// Microbenchmark: cost of divergent 16-B gathers (global_load_dwordx4, one distinct 64-B line per lane,
// L2-resident table) as a function of how many lanes of the wave are active.
// Question: is the L1 return path paid per wave-instruction or per active lane?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_lanes.hip -o gpurun_out/gather_lanes && gpurun_out/gather_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ table, uint32_t mask_lines, int active, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    if ((int)lane < active) {
        for (int i = 0; i < iters; i++) {
            // 4 dependent-address-free gathers per iteration, each lane its own 64-B line
            s = s * 747796405u + 2891336453u; uint32_t a0 = (s >> 8) & mask_lines;
            s = s * 747796405u + 2891336453u; uint32_t a1 = (s >> 8) & mask_lines;
            s = s * 747796405u + 2891336453u; uint32_t a2 = (s >> 8) & mask_lines;
            s = s * 747796405u + 2891336453u; uint32_t a3 = (s >> 8) & mask_lines;
            float4 v0 = table[a0 * 4u], v1 = table[a1 * 4u + 1u], v2 = table[a2 * 4u + 2u], v3 = table[a3 * 4u + 3u];
            acc += v0.x + v1.y + v2.z + v3.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint32_t lines = argc > 1 ? (uint32_t)atoi(argv[1]) : (1u << 15);   // default 32768 x 64 B = 2 MB: L2-resident per XCD
    std::vector<float> h(lines * 16, 1.0f);
    float4 *d; float *out;
    hipMalloc((void **)&d, lines * 64); hipMalloc((void **)&out, 4);
    hipMemcpy(d, h.data(), lines * 64, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    for (int rep = 0; rep < 2; rep++)
        for (int active : {64, 48, 32, 16, 8}) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, d, lines - 1, active, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double wave_insts = (double)blocks * 4 * iters * 4;
            if (rep) printf("active lanes %2d: %.3f ms, %.1f wave-loads/us/CU-equiv(256), %.2f cycles@2.4GHz per wave-load per CU, %.1f G lane-loads/s\n",
                            active, ms, wave_insts / (ms * 1e3) / 256.0, (ms * 1e-3 * 2.4e9) / (wave_insts / 256.0), wave_insts * active / (ms * 1e-3) / 1e9);
        }
    return 0;
}
