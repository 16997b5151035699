// rt_kernels.hip — the compute pass of the reference, as a gfx950 kernel.
//
// Replaces the WGSL entry point `main` and everything it calls
// (/root/reference/shaders/main.wgsl:14-118, ray.wgsl, random.wgsl, misc.wgsl)
// and the driver ray query behind rayQueryProceed (main.wgsl:79-83), for which
// CDNA4 has no hardware: traversal is a software while-while loop over the
// host-built BVH2 with a per-lane stack in LDS.
//
// Work mapping: one wavefront (64 lanes) = one 8x8 pixel tile — the reference's
// workgroup shape (main.wgsl:15) — one lane per pixel-frame.  RNG state chains
// across the samples of a pixel (main.wgsl:23-26), so a lane owns its pixel for
// the whole dispatch; inside the lane the sample and bounce loops are flattened
// into one loop of (trace, shade) steps so that a lane whose path ended starts
// its next sample at once instead of idling until the wave's longest path ends.
//
// Arithmetic follows the RTN-1 contract (rt_math.h): results are bit-identical
// to oracle/rt_oracle.c.  Only the ray/box test is free to differ (it must be
// conservative, nothing more), so it uses v_rcp_f32 and a widened far bound.
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_math.h"

namespace {

constexpr int kBlock = 256;   // 4 waves = 4 tiles per workgroup
constexpr float kTMin = 0.001f;              // main.wgsl:77
constexpr float kTMax = 3.40282347e+38f;     // main.wgsl:77
constexpr uint32_t kNoHit = 0xFFFFFFFFu;

struct Hit {
    float t, u, v;
    uint32_t slot;   // leaf-order triangle slot, kNoHit = miss
    uint32_t gid;
    bool front;
};

__device__ __forceinline__ float4 ld4(const void *p) { return *reinterpret_cast<const float4 *>(p); }

// ray/triangle test of the contract (Moller-Trumbore, explicit fmaf); mirrors
// tri_test()/consider_tri() of the oracle decision for decision.
template <bool COUNT>
__device__ __forceinline__ void intersect_leaf(const DevTri *__restrict__ tris, uint32_t first, uint32_t count,
                                               rt_v3 o, rt_v3 d, bool cull, bool cw, Hit &best,
                                               unsigned &n_tri) {
    for (uint32_t i = 0; i < count; i++) {
        const DevTri *tp = tris + first + i;
        float4 q0 = ld4(&tp->ax), q1 = ld4(&tp->by), q2 = ld4(&tp->cz);
        if (COUNT) n_tri++;
        rt_v3 a = v3(q0.x, q0.y, q0.z), b = v3(q0.w, q1.x, q1.y), c = v3(q1.z, q1.w, q2.x);
        uint32_t gid = __float_as_uint(q2.y);
        rt_v3 e1 = b - a, e2 = c - a;
        rt_v3 p = rt_cross_fma(d, e2);
        float det = rt_dot_fma(e1, p);
        if (!(det > 0.0f || det < 0.0f)) continue;
        bool front = cw ? (det < 0.0f) : (det > 0.0f);
        if (cull && !front) continue;
        float inv = 1.0f / det;
        rt_v3 tv = o - a;
        float u = rt_dot_fma(tv, p) * inv;
        if (!(u >= 0.0f && u <= 1.0f)) continue;
        rt_v3 q = rt_cross_fma(tv, e1);
        float v = rt_dot_fma(d, q) * inv;
        if (!(v >= 0.0f && u + v <= 1.0f)) continue;
        float t = rt_dot_fma(e2, q) * inv;
        if (!(t > kTMin)) continue;
        if (t < best.t || (t == best.t && gid < best.gid)) {
            best.t = t; best.u = u; best.v = v; best.slot = first + i; best.gid = gid; best.front = front;
        }
    }
}

// nearest committed hit in (kTMin, kTMax): the ray query of main.wgsl:79-83
template <bool COUNT>
__device__ __forceinline__ Hit trace(const RenderParams &P, uint32_t *__restrict__ stack, rt_v3 o, rt_v3 d,
                                     unsigned &n_node, unsigned &n_tri) {
    Hit best;
    best.t = kTMax; best.u = 0.0f; best.v = 0.0f; best.slot = kNoHit; best.gid = kNoHit; best.front = true;
    const bool cull = (P.flags & 1u) != 0;   // flags = 0x10 * (ctx.flags & 1), main.wgsl:76
    const bool cw = P.front_face_cw != 0;
    // box test only: approximate reciprocal is fine, the far bound is widened
    const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
    const DevNode *__restrict__ nodes = P.nodes;

    int sp = 0;
    int32_t cur = 0;   // root is always an inner node
    for (;;) {
        while (cur >= 0) {
            const DevNode *n = nodes + cur;
            float4 a = ld4(&n->lo0[0]), b = ld4(&n->hi0[0]), c = ld4(&n->lo1[0]), e = ld4(&n->hi1[0]);
            if (COUNT) n_node++;
            float t0x = (a.x - o.x) * ix, t1x = (b.x - o.x) * ix;
            float t0y = (a.y - o.y) * iy, t1y = (b.y - o.y) * iy;
            float t0z = (a.z - o.z) * iz, t1z = (b.z - o.z) * iz;
            float n0 = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), kTMin));
            float f0 = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), best.t));
            float s0x = (c.x - o.x) * ix, s1x = (e.x - o.x) * ix;
            float s0y = (c.y - o.y) * iy, s1y = (e.y - o.y) * iy;
            float s0z = (c.z - o.z) * iz, s1z = (e.z - o.z) * iz;
            float n1 = fmaxf(fmaxf(fminf(s0x, s1x), fminf(s0y, s1y)), fmaxf(fminf(s0z, s1z), kTMin));
            float f1 = fminf(fminf(fmaxf(s0x, s1x), fmaxf(s0y, s1y)), fminf(fmaxf(s0z, s1z), best.t));
            bool h0 = n0 <= f0 * 1.000002f;
            bool h1 = n1 <= f1 * 1.000002f;
            int32_t r0 = __float_as_int(a.w), r1 = __float_as_int(b.w);
            if (h0 && h1) {
                bool swap = n1 < n0;
                int32_t nearc = swap ? r1 : r0, farc = swap ? r0 : r1;
                stack[sp * kBlock] = (uint32_t)farc;
                sp++;
                cur = nearc;
            } else if (h0) {
                cur = r0;
            } else if (h1) {
                cur = r1;
            } else {
                if (sp == 0) return best;
                sp--;
                cur = (int32_t)stack[sp * kBlock];
            }
        }
        // leaf
        uint32_t enc = (uint32_t)~cur;
        intersect_leaf<COUNT>(P.tris, enc >> 3, enc & 7u, o, d, cull, cw, best, n_tri);
        if (sp == 0) return best;
        sp--;
        cur = (int32_t)stack[sp * kBlock];
    }
}

// shaders/misc.wgsl:10-18
__device__ __forceinline__ rt_v3 tangent_space(rt_v3 normal, rt_v3 s) {
    rt_v3 arbitrary = v3(1.0f, 0.0f, 0.0f);
    if (fabsf(normal.x) > 0.9f) arbitrary = v3(0.0f, 1.0f, 0.0f);
    else if (fabsf(normal.y) > 0.9f) arbitrary = v3(0.0f, 0.0f, 1.0f);
    rt_v3 tangent = rt_normalize(rt_cross(arbitrary, normal));
    rt_v3 bitangent = rt_cross(normal, tangent);
    return (tangent * s.x + bitangent * s.y) + normal * s.z;
}

// shaders/random.wgsl:18-29
__device__ __forceinline__ rt_v3 rand_cosine_hemisphere_vector(uint32_t &seed, rt_v3 normal) {
    const float PI = 3.141592653589793f;
    float r = sqrtf(rt_rand(seed));
    float theta = 2.0f * PI * rt_rand(seed);
    float st, ct;
    rt_sincos(theta, &st, &ct);
    rt_v3 s = v3(r * ct, r * st, sqrtf(1.0f - r * r));
    return tangent_space(normal, s);
}

// shaders/main.wgsl:104-108, pow(x, 5.0) as ((x*x)*(x*x))*x
__device__ __forceinline__ float schlick_approximation(float cos_theta, float ri) {
    float r = (1.0f - ri) / (1.0f + ri);
    float rs = r * r;
    float x = 1.0f - cos_theta;
    float x2 = x * x;
    float x5 = (x2 * x2) * x;
    return rs + (1.0f - rs) * x5;
}

// shaders/main.wgsl:111-118
__device__ __forceinline__ float tone_map1(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    float q = (x * (a * x + b)) / (x * (c * x + d) + e);
    q = (q > 0.0f) ? q : 0.0f;   // saturate, NaN -> 0
    q = (q < 1.0f) ? q : 1.0f;
    return q;
}

template <bool COUNT, bool AOV>
__global__ __launch_bounds__(kBlock) void rt_render_kernel(const RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t tile = blockIdx.x * (kBlock / 64) + (tid >> 6);
    const uint32_t tiles_total = P.tiles_x * P.local_bands;
    unsigned n_rays = 0, n_node = 0, n_tri = 0, n_pf = 0;

    // tile -> pixel.  Bands of 8 rows are dealt round-robin to shards
    // (rt_abi.h rt_set_shard); pixel_idx stays global (main.wgsl:19).
    const uint32_t band_local = tile / P.tiles_x;
    const uint32_t tx = tile - band_local * P.tiles_x;
    const uint32_t x = tx * 8u + (lane & 7u);
    const uint32_t y = (band_local * P.shard_count + P.shard_index) * 8u + (lane >> 3);
    const bool active = tile < tiles_total && x < P.width && y < P.height;   // main.wgsl:17

    if (active) {
        uint32_t *stack = lds_stack + tid;
        const uint32_t pixel_idx = y * P.width + x;
        const float Wf = (float)P.width, Hf = (float)P.height;
        const float px = (float)x / Wf - 0.5f;                    // main.wgsl:20-21
        const float py = (1.0f - (float)y / Hf) - 0.5f;
        uint32_t seed = rt_seed(pixel_idx, P.frame);              // main.wgsl:23
        const rt_v3 cam = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
        const rt_v3 forward = v3(P.forward[0], P.forward[1], P.forward[2]);
        const rt_v3 right = v3(P.right[0], P.right[1], P.right[2]);
        const rt_v3 up = v3(P.up[0], P.up[1], P.up[2]);

        rt_v3 total = v3(0.0f, 0.0f, 0.0f);     // `color` of main.wgsl:25
        uint32_t sample = 0, bounce = 0;
        rt_v3 light, color, pos, dir;
        bool first_ray = true;

        // sample(), main.wgsl:36-41: jitter (x drawn first), primary ray
        auto new_path = [&]() {
            float r0 = rt_rand(seed), r1 = rt_rand(seed);
            float offx = (r0 * 2.0f - 1.0f) / Wf, offy = (r1 * 2.0f - 1.0f) / Hf;
            float ux = (px + offx) * P.aspect * P.fov_scale;      // ray.wgsl:6-7
            float uy = (py + offy) * P.fov_scale;
            dir = rt_normalize((forward + right * ux) + up * uy); // ray.wgsl:9
            pos = cam;
            light = v3(0.0f, 0.0f, 0.0f);
            color = v3(1.0f, 1.0f, 1.0f);
            bounce = 0;
        };

        if (P.samples > 0) {
            new_path();
            for (;;) {
                Hit h = trace<COUNT>(P, stack, pos, dir, n_node, n_tri);   // main.wgsl:44
                n_rays++;
                if (AOV && first_ray) {
                    uint32_t g = kNoHit, pr = kNoHit;
                    if (h.slot != kNoHit) { g = P.shade[h.slot].geometry; pr = P.shade[h.slot].primitive; }
                    const uint32_t local_row = band_local * 8u + (lane >> 3);
                    uint2 *dst = reinterpret_cast<uint2 *>(P.aov) + (size_t)local_row * P.width + x;
                    *dst = make_uint2(g, pr);
                }
                first_ray = false;
                bool path_done;
                if (h.slot == kNoHit) {                                   // main.wgsl:46-50
                    float a = 0.5f * (dir.y + 1.0f);                      // misc.wgsl:1-4
                    float w = 1.0f - a;
                    rt_v3 bg = v3(w * 1.0f + a * 0.5f, w * 1.0f + a * 0.7f, w * 1.0f + a * 1.0f);
                    light = light + (bg * color) * P.environment;
                    path_done = true;
                } else {
                    // attribute part of trace_ray, main.wgsl:86-101
                    const DevShade *sh = P.shade + h.slot;
                    float4 s0 = ld4(&sh->p0[0]), s1 = ld4(&sh->n0y), s2 = ld4(&sh->p1[0]), s3 = ld4(&sh->n1y),
                           s4 = ld4(&sh->p2[0]), s5 = ld4(&sh->n2y);
                    const uint32_t geometry = sh->geometry;
                    const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
                    rt_v3 normal = v3(s0.w * b0 + s2.w * b1 + s4.w * b2,
                                      s1.x * b0 + s3.x * b1 + s5.x * b2,
                                      s1.y * b0 + s3.y * b1 + s5.y * b2);
                    rt_v3 position = v3(s0.x * b0 + s2.x * b1 + s4.x * b2,
                                        s0.y * b0 + s2.y * b1 + s4.y * b2,
                                        s0.z * b0 + s2.z * b1 + s4.z * b2);
                    const DevMaterial *mp = P.materials + geometry;        // models[geometry_index].material
                    float4 m0 = ld4(&mp->diffuse[0]), m1 = ld4(&mp->specular[0]), m2 = ld4(&mp->emitted[0]);
                    const uint32_t tag = mp->tag;
                    if (tag == 0u) {                                        // main.wgsl:53-61
                        // get_scattered_direction_metal, ray.wgsl:12-32 (texture ids ignored: SURVEY §8 N3)
                        float is_specular = (rt_rand(seed) < m0.w) ? 1.0f : 0.0f;
                        float smoothness = 1.0f - m1.w;
                        rt_v3 diffuse = rand_cosine_hemisphere_vector(seed, normal);
                        rt_v3 specular = rt_reflect(dir, normal);
                        rt_v3 sdir = rt_mix(diffuse, specular, smoothness * is_specular);
                        rt_v3 scol = rt_mix(v3(m0.x, m0.y, m0.z), v3(m1.x, m1.y, m1.z), is_specular);
                        light = light + v3(m2.x, m2.y, m2.z) * color;
                        color = color * scol;
                        pos = position + normal * 0.0001f;
                        dir = sdir;
                    } else if (tag == 1u) {                                 // main.wgsl:62-68
                        // get_scattered_direction_dielectric, ray.wgsl:34-49
                        rt_v3 n = rt_dot(normal, dir) < 0.0f ? normal : -normal;   // faceForward(N, N, dir)
                        float ri = m2.w;
                        if (h.front) ri = 1.0f / m2.w;
                        float c = rt_dot(-dir, n);
                        float cos_theta = (c < 1.0f) ? c : 1.0f;
                        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                        bool must_reflect = ri * sin_theta > 1.0f;
                        float reflect_prob = schlick_approximation(cos_theta, ri);
                        bool refl = must_reflect || (reflect_prob > rt_rand(seed));   // || short-circuits
                        rt_v3 next_dir = refl ? rt_reflect(dir, n) : rt_refract(dir, n, ri);
                        rt_v3 offset_dir = normal - (normal * 2.0f) * (h.front ? 1.0f : 0.0f);
                        pos = position + offset_dir * 0.0001f;
                        dir = next_dir;
                    }
                    bounce++;
                    path_done = bounce > P.max_bounces;                     // `bounce <= max_bounces`, main.wgsl:43
                }
                if (path_done) {
                    total = total + light;                                  // main.wgsl:26
                    sample++;
                    if (sample >= P.samples) break;
                    new_path();
                }
            }
        }
        // main.wgsl:27-32
        const float ns = (float)P.samples;
        total = v3(total.x / ns, total.y / ns, total.z / ns);
        rt_v3 mapped = v3(tone_map1(total.x * P.exposure), tone_map1(total.y * P.exposure), tone_map1(total.z * P.exposure));
        const float w = 1.0f / (float)(P.accumulation_frame + 1u);
        const uint32_t local_row = band_local * 8u + (lane >> 3);
        float4 *dst = reinterpret_cast<float4 *>(P.accum) + (size_t)local_row * P.width + x;
        float4 old = *dst;
        rt_v3 out = rt_mix(v3(old.x, old.y, old.z), mapped, w);
        *dst = make_float4(out.x, out.y, out.z, old.w);
        n_pf = 1;
    }

    // exact device-side counters: one atomic per wave per counter
    unsigned long long r = n_rays, pf = n_pf, nn = n_node, nt = n_tri;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        pf += __shfl_down(pf, off);
        if (COUNT) { nn += __shfl_down(nn, off); nt += __shfl_down(nt, off); }
    }
    if (lane == 0) {
        atomicAdd(&P.counters->rays, r);
        atomicAdd(&P.counters->pixel_frames, pf);
        if (COUNT) { atomicAdd(&P.counters->node_visits, nn); atomicAdd(&P.counters->triangle_tests, nt); }
    }
}

}  // namespace

// Host-side launcher.  Grid: one wave per 8x8 tile, 4 tiles per workgroup;
// dynamic LDS = stack_entries x 256 lanes x 4 B.
hipError_t rt_launch_render(const RenderParams &P, bool count, bool aov, hipStream_t stream) {
    const uint32_t tiles = P.tiles_x * P.local_bands;
    if (tiles == 0) return hipSuccess;
    const uint32_t blocks = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
    const size_t lds = (size_t)P.stack_entries * kBlock * sizeof(uint32_t);
    dim3 grid(blocks), block(kBlock);
    if (count) {
        if (aov) hipLaunchKernelGGL((rt_render_kernel<true, true>), grid, block, lds, stream, P);
        else hipLaunchKernelGGL((rt_render_kernel<true, false>), grid, block, lds, stream, P);
    } else {
        if (aov) hipLaunchKernelGGL((rt_render_kernel<false, true>), grid, block, lds, stream, P);
        else hipLaunchKernelGGL((rt_render_kernel<false, false>), grid, block, lds, stream, P);
    }
    return hipGetLastError();
}
