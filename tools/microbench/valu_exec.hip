// Microbenchmark: does a wave64 VALU instruction cost less when whole 16-lane quarters of EXEC are off?
// (CDNA executes a wave64 VALU op as 4 passes of 16 lanes; if empty passes were skipped, packing the lanes of
// one phase into few quarters would buy time.)  Patterns: first N lanes, every 4th lane, one lane per quarter.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_exec.hip -o tools/microbench/valu_exec.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void chain(unsigned long long mask, int iters, float *out) {
    const unsigned lane = threadIdx.x & 63u;
    float a = (float)threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; i++) {        // 8 independent-ish FMAs per iteration
            a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.999f, 0.001f);
            a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.999f, 0.001f);
        }
    }
    if (a + c + d == 123.456f) out[0] = a;
}

int main() {
    float *out; hipMalloc((void **)&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 20, iters = 20000;   // 5 waves per SIMD resident
    struct { const char *name; unsigned long long mask; } pats[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-47", (1ull << 48) - 1}, {"lanes 0-31", (1ull << 32) - 1}, {"lanes 0-15", 0xFFFFull},
        {"every 2nd lane (32)", 0x5555555555555555ull}, {"every 4th lane (16)", 0x1111111111111111ull},
        {"lanes 0-7 of each quarter (32)", 0x00FF00FF00FF00FFull}, {"one lane", 1ull}};
    for (int rep = 0; rep < 2; rep++)
        for (auto &p : pats) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, p.mask, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%-34s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", p.name, ms,
                            ms * 1e-3 * 2.4e9 / ((double)iters * 8 * (blocks * 4.0 / 1024)));
        }
    return 0;
}
