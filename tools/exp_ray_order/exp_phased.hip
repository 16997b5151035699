// exp_phased.hip — developer experiment: the phased kernel's RAYLIST instance, i.e. the production
// scheduler and 8-ary traversal run over a list of rays instead of pixels.  Built like the production phased unit (same flags).
// Not part of librt_amd.so.
#define RT_TU_DEVICE_ONLY 1
#include "../../ray-tracing_amd/csrc/rt_kernels.hip"

// resident workgroups of the instance (the persistent grid)
int exp_phased_blocks(int device, uint32_t stack_entries, bool count, bool spheres) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256 * 4;
    const size_t lds = (size_t)(stack_entries / 2u) * kBlock * 6u;
    const void *fn = count ? (spheres ? (const void *)rt_render_phased<true, false, true, true, 3, true> : (const void *)rt_render_phased<true, false, false, true, 3, true>)
                           : (spheres ? (const void *)rt_render_phased<false, false, true, true, 3, true> : (const void *)rt_render_phased<false, false, false, true, 3, true>);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 4;
    return per_cu * prop.multiProcessorCount;
}

// P.stage = rays, P.aov = order or null, P.accum = hits, P.stage_stride = number of rays; P.queue zeroed by the caller
hipError_t exp_launch_phased_raylist(const RenderParams &P, bool count, bool spheres, unsigned blocks, hipStream_t stream) {
    const size_t lds = (size_t)(P.stack_entries / 2u) * kBlock * 6u;
    const dim3 grid(blocks);
    if (count) {
        if (spheres) return launch(rt_render_phased<true, false, true, true, 3, true>, grid, lds, stream, P);
        return launch(rt_render_phased<true, false, false, true, 3, true>, grid, lds, stream, P);
    }
    if (spheres) return launch(rt_render_phased<false, false, true, true, 3, true>, grid, lds, stream, P);
    return launch(rt_render_phased<false, false, false, true, 3, true>, grid, lds, stream, P);
}
