// exp_main.hip — developer experiment: does the ORDER of the rays matter on scenes whose
// records leave the caches?  One "generation" of secondary rays of a real render is dumped (every pixel-frame's ray at a given
// sample and bounce), then traced as a plain list — in arrival order (pixel-frame order: what a wavefront tracer without a sort
// would see) and in any order the caller supplies — by (a) a plain one-ray-per-lane kernel and (b) the production phased
// scheduler run over the list (exp_phased.hip).  Hits must be the same bits whatever the order.
//
// The unit includes the library's API source so that it sees rt_context; librt_exp.so = this + the library's other objects.
// Not part of librt_amd.so.
#define RT_TU_DEVICE_ONLY 1
#include "../../ray-tracing_amd/csrc/rt_kernels.hip"
#include "../../ray-tracing_amd/csrc/rt_api.hip"

int exp_phased_blocks(int device, uint32_t stack_entries, bool count, bool spheres);
hipError_t exp_launch_phased_raylist(const RenderParams &P, bool count, bool spheres, unsigned blocks, hipStream_t stream);

namespace {

// every pixel-frame of the launch walks its path as the render would, up to (sample, bounce); the ray it would trace there is
// written to out[2q], out[2q + 1] (q = pixel-frame index: frame-major, tile-major inside a frame); direction 0 = no such ray
template <bool SPH>
__global__ __launch_bounds__(kBlock) void exp_dump_rays(const RenderParams P, uint32_t sample, uint32_t bounce, float4 *out) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    fill_next_slot_table();
    const uint32_t frame_slots = P.tiles_x * P.local_bands * 64u;
    const uint32_t q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= frame_slots * P.n_frames) return;
    out[2u * (size_t)q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    out[2u * (size_t)q + 1u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t f = q / frame_slots, s = q - f * frame_slots;
    uint32_t x, y, lr;
    if (!slot_to_pixel(P, s, x, y, lr) || P.samples == 0) return;
    const Camera C = load_camera(P);
    PixelState S;
    Tally tl;
    begin_pixel(P, C, S, x, y, lr, f);
    new_path(P, C, S);
    for (;;) {
        if (S.sample > sample) return;                    // the path of `sample` ended before `bounce`
        if (S.sample == sample && S.bounce == bounce) {
            out[2u * (size_t)q] = make_float4(S.pos.x, S.pos.y, S.pos.z, 0.0f);
            out[2u * (size_t)q + 1u] = make_float4(S.dir.x, S.dir.y, S.dir.z, 0.0f);
            return;
        }
        const Hit h = trace8<false, SPH, true>(P, stack, S.pos, S.dir, tl);
        if (shade<false, SPH>(P, C, S, h)) return;
    }
}

// one ray per lane, whole traversal per lane (the 8-ary tree, compact records)
template <bool COUNT, bool SPH>
__global__ __launch_bounds__(kBlock) void exp_trace_plain(const RenderParams P, const float4 *__restrict__ rays, const uint32_t *__restrict__ order,
                                                          uint32_t n, uint2 *__restrict__ hits) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    fill_next_slot_table();
    Tally tl;
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) {
        const uint32_t r = order ? order[i] : i;
        const float4 ro = rays[2u * (size_t)r], rd = rays[2u * (size_t)r + 1u];
        const Hit h = trace8<COUNT, SPH, true>(P, stack, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), tl);
        tl.rays++;
        // counting launches return the ray's own work instead of its hit: (node visits, primitive tests)
        hits[r] = COUNT ? make_uint2(tl.nodes, tl.tris) : make_uint2(__float_as_uint(h.t), h.slot);
    }
    if (COUNT) flush_tally<COUNT>(P, tl);       // (without counting: no atomics at all — 127 k waves adding to one address serialise)
}

float4 *g_rays = nullptr;
uint32_t *g_order = nullptr;
uint2 *g_hits = nullptr;
uint64_t g_n = 0;

}  // namespace

extern "C" {

// out: n_frames x frame slots x 8 floats (host).  *n_slots receives the number of pixel-frame slots written.
int rt_exp_dump_rays(rt_context *ctx, const RtUniform *u, uint32_t n_frames, uint32_t sample, uint32_t bounce, float *out, uint64_t cap_slots, uint64_t *n_slots) {
    int r = check_dispatch(ctx, u);
    if (r) return r;
    if (!ctx->node_compact) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp: the scene must be on 64-byte 8-ary nodes (RT_OPT_NODE_FORMAT 4)");
    RenderParams P;
    r = make_params(ctx, *u, 1, nullptr, P);
    if (r) return r;
    P.n_frames = n_frames;                         // begin_pixel takes the frame offset from the slot index; nothing is staged
    const uint64_t slots = (uint64_t)P.tiles_x * P.local_bands * 64u * n_frames;
    if (slots > cap_slots || slots >= (1ull << 32)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp_dump_rays: %llu slots", (unsigned long long)slots);
    float4 *d_out = nullptr;
    RT_HIP(dev_malloc((void **)&d_out, slots * 32u));
    const size_t lds = (size_t)P.stack_entries * kBlock * sizeof(uint32_t);
    const bool sph = !ctx->hs->spheres.empty() || ctx->n_textures > 0;
    const dim3 grid((unsigned)((slots + kBlock - 1) / kBlock));
    if (sph) hipLaunchKernelGGL(exp_dump_rays<true>, grid, dim3(kBlock), lds, ctx->stream, P, sample, bounce, d_out);
    else hipLaunchKernelGGL(exp_dump_rays<false>, grid, dim3(kBlock), lds, ctx->stream, P, sample, bounce, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, slots * 32u, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_exp_dump_rays: %s", hipGetErrorString(e));
    *n_slots = slots;
    return RT_OK;
}

// the ray list of the following rt_exp_trace calls: n x 8 floats (origin xyz _, direction xyz _)
int rt_exp_set_rays(rt_context *ctx, const float *rays, uint64_t n) {
    if (!ctx || !rays || n == 0 || n >= (1ull << 31)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp_set_rays: bad arguments");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    if (g_rays) (void)hipFree(g_rays);
    if (g_order) (void)hipFree(g_order);
    if (g_hits) (void)hipFree(g_hits);
    g_rays = nullptr; g_order = nullptr; g_hits = nullptr; g_n = 0;
    RT_HIP(dev_malloc((void **)&g_rays, n * 32u));
    RT_HIP(dev_malloc((void **)&g_order, n * 4u));
    RT_HIP(dev_malloc((void **)&g_hits, n * 8u));
    RT_HIP(hipMemcpy(g_rays, rays, n * 32u, hipMemcpyHostToDevice));
    g_n = n;
    return RT_OK;
}

// Trace the list `reps` times in the order given (order = null: as stored).  kernel 0 = plain, 1 = phased scheduler.
// ms[reps] receives each launch's device time; hits (n x 2 u32: t bits, slot) the committed hits; counters[8] the DevCounters of the last launch.
int rt_exp_trace(rt_context *ctx, const RtUniform *u, const uint32_t *order, int kernel, int count, int reps, float *ms, uint32_t *hits, uint64_t *counters) {
    int r = check_dispatch(ctx, u);
    if (r) return r;
    if (!ctx->node_compact) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp: the scene must be on 64-byte 8-ary nodes (RT_OPT_NODE_FORMAT 4)");
    if (!g_n || reps < 1 || !ms) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp_trace: no rays / bad arguments");
    if (order) {
        for (uint64_t i = 0; i < g_n; i++) if (order[i] >= g_n) return fail(RT_ERR_INVALID_ARGUMENT, "rt_exp_trace: order[%llu] out of range", (unsigned long long)i);
        RT_HIP(hipMemcpy(g_order, order, g_n * 4u, hipMemcpyHostToDevice));
    }
    RenderParams P;
    r = make_params(ctx, *u, 1, nullptr, P);
    if (r) return r;
    const bool sph = !ctx->hs->spheres.empty() || ctx->n_textures > 0;
    const uint32_t n = (uint32_t)g_n;
    for (int k = 0; k < reps; k++) {
        RT_HIP(hipMemsetAsync(ctx->d_counters, 0, sizeof(DevCounters), ctx->stream));
        RT_HIP(hipMemsetAsync(g_hits, 0xFF, g_n * 8u, ctx->stream));
        hipError_t e;
        if (kernel == 0) {
            const size_t lds = (size_t)P.stack_entries * kBlock * sizeof(uint32_t);
            const dim3 grid((n + kBlock - 1) / kBlock);
            RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
            if (count) {
                if (sph) hipLaunchKernelGGL((exp_trace_plain<true, true>), grid, dim3(kBlock), lds, ctx->stream, P, g_rays, order ? g_order : nullptr, n, g_hits);
                else hipLaunchKernelGGL((exp_trace_plain<true, false>), grid, dim3(kBlock), lds, ctx->stream, P, g_rays, order ? g_order : nullptr, n, g_hits);
            } else {
                if (sph) hipLaunchKernelGGL((exp_trace_plain<false, true>), grid, dim3(kBlock), lds, ctx->stream, P, g_rays, order ? g_order : nullptr, n, g_hits);
                else hipLaunchKernelGGL((exp_trace_plain<false, false>), grid, dim3(kBlock), lds, ctx->stream, P, g_rays, order ? g_order : nullptr, n, g_hits);
            }
            e = hipGetLastError();
        } else {
            RenderParams Q = P;
            Q.stage = g_rays;
            Q.aov = order ? g_order : nullptr;
            Q.accum = reinterpret_cast<float *>(g_hits);
            Q.stage_stride = g_n;
            Q.n_frames = 1;
            Q.queue = ctx->d_queue;
            int blocks = exp_phased_blocks(ctx->device, ctx->stack_entries, count != 0, sph);
            if (ctx->tune[3] > 0 && ctx->cu_count > 0 && blocks > ctx->tune[3] * ctx->cu_count) blocks = ctx->tune[3] * ctx->cu_count;
            const uint64_t chunks = (g_n + Q.chunk - 1) / Q.chunk;
            if ((uint64_t)blocks > (chunks + 3) / 4) blocks = (int)((chunks + 3) / 4);
            if (blocks < 1) blocks = 1;
            RT_HIP(hipMemsetAsync(Q.queue, 0, sizeof(DevQueue), ctx->stream));
            RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
            e = exp_launch_phased_raylist(Q, count != 0, sph, (unsigned)blocks, ctx->stream);
        }
        if (e != hipSuccess) return fail(RT_ERR_HIP, "rt_exp_trace: %s", hipGetErrorString(e));
        RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        RT_HIP(hipEventSynchronize(ctx->ev1));
        RT_HIP(hipEventElapsedTime(&ms[k], ctx->ev0, ctx->ev1));
    }
    if (hits) RT_HIP(hipMemcpy(hits, g_hits, g_n * 8u, hipMemcpyDeviceToHost));
    if (counters) RT_HIP(hipMemcpy(counters, ctx->d_counters, sizeof(DevCounters), hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"
