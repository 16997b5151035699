// Microbenchmark: cost of a 16-byte gather when groups of G adjacent lanes read the SAME address
// (G = 1: every lane its own line ... G = 64: the whole wave reads one address), table L1-resident.
// Question: are the top levels of the BVH, where many lanes of a wave sit at the same node, cheap on the L1 path?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_share.hip -o tools/microbench/gather_share.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ table, uint32_t mask_lines, int group, int iters, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    // one random stream per GROUP of lanes: lanes of a group compute identical addresses
    uint32_t s = ((blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u + (lane / (uint32_t)group)) * 2654435761u + 12345u;
    float acc = 0.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            s = s * 747796405u + 2891336453u;
            float4 v = table[((s >> 8) & mask_lines) * 4u];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    float *out; (void)hipMalloc((void **)&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 2000;
    const uint32_t lines = 128;
    std::vector<float> h(lines * 16, 1.0f);
    float4 *d; (void)hipMalloc((void **)&d, lines * 64);
    (void)hipMemcpy(d, h.data(), lines * 64, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; rep++)
        for (int g : {1, 2, 4, 8, 16, 32, 64}) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, d, lines - 1, g, iters, out);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double wave_loads = (double)blocks * 4 * iters * 4;
            if (rep) printf("groups of %2d lanes share an address: %.3f ms, %.1f clk per wave-load per CU (2.4 GHz)\n", g, ms,
                            ms * 1e-3 * 2.4e9 / (wave_loads / 256.0));
        }
    return 0;
}
