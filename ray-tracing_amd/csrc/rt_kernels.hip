// rt_kernels.hip — the compute pass of the reference, as gfx950 kernels.
//
// Replaces the WGSL entry point `main` and everything it calls
// (/root/reference/shaders/main.wgsl:14-118, ray.wgsl, random.wgsl, misc.wgsl)
// and the driver ray query behind rayQueryProceed (main.wgsl:79-83), for which
// CDNA4 has no hardware: traversal is a software while-while loop over the
// host-built BVH2 with a per-lane stack in LDS.
//
// Unit of work = one pixel-frame (one invocation of `main`): the RNG state
// chains across the samples of a pixel (main.wgsl:23-26), so samples of one
// pixel cannot be split over lanes.  Inside a lane the sample and bounce loops
// are flattened into one loop of (trace, shade) steps, so a lane whose path
// ended starts its next sample at once.
//
// Variant 2 (default) — persistent wavefronts, lanes scheduled by phase
//   (rt_render_phased, see the comment above it).
// Variant 0 — persistent wavefronts with a work queue, one whole ray per step:
//   each wave claims chunks of pixel slots from a global head with one atomic
//   and hands slots to its lanes as they finish.
// Variant 1 — one wave per 8x8 tile, one lane per pixel, the reference's own
//   mapping (main.wgsl:15).
// Variants 0 and 1 are kept for A/B measurements (RT_OPT_KERNEL).
//
// Arithmetic follows the RTN-1 contract (rt_math.h): results are bit-identical
// to oracle/rt_oracle.c.  Only the ray/box test is free to differ (it must be
// conservative, nothing more), so it uses v_rcp_f32, f16 box offsets fed to
// v_fma_mix_f32 and a widened far bound (see Slab / the node step below).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_device.h"
#include "rt_math.h"
#include "rt_pack.h"

// This file is compiled TWICE (Makefile): -DRT_TU_PHASED = variant 2 and the blend kernel, built with
// -mllvm -structurizecfg-skip-uniform-regions; -DRT_TU_SIMPLE = variants 0 and 1 and the launcher, built WITHOUT that flag.
// The flag keeps wave-uniform branches out of the structurizer (4 % on C3), and it MISCOMPILES a uniform branch whose arms
// hold divergent loops: step() below, the whole-ray traversals of variants 0 / 1 chosen by node format, lost hits in
// rt_render_persistent<false, false, true> on 8-ary nodes (found by tests/checkers/fuzz_parity.py).
// The phased kernel has no divergent loop at all — every loop in it exits on a wave-uniform condition — and is the only
// code that gets the flag.  With neither macro (tools/kernel_regs.sh, tools/isa_bodies.py) the file is one unit.
// -DRT_TU_DEVICE_ONLY (tools/exp_ray_order/): the device functions and kernel templates alone, no launcher — for a developer
// unit that includes this file to instantiate kernels of its own.
#if !defined(RT_TU_PHASED) && !defined(RT_TU_SIMPLE) && !defined(RT_TU_DEVICE_ONLY)
#define RT_TU_PHASED 1
#define RT_TU_SIMPLE 1
#endif

namespace {

constexpr int kBlock = 256;                  // 4 waves per workgroup
__shared__ float4 g_lds_materials[RT_LDS_MATERIALS * 4u];   // LDSM instances of the phased kernel: the scene's material records
#ifndef RT_NODE_REPS
#define RT_NODE_REPS 3                       // node steps per scheduling decision
#endif
#ifndef RT_NODE_EARLY_OUT
#define RT_NODE_EARLY_OUT 2                  // leave the unrolled node steps when under 1/2 of the entering lanes remain (-1 %)
#endif
#ifndef RT_NODE_EARLY_OUT_WIDE
#define RT_NODE_EARLY_OUT_WIDE 3             // 8-ary nodes: under 1/3 (a step is 230 instructions; C4 76.4 -> 75.3 ms per step against 1/2, 77.7 with no early return; C3 on binary nodes: 15.2 either way, 16.1 with none)
#endif
#ifndef RT_STAGE_LINES
#define RT_STAGE_LINES 0                     // per-wave LDS slots that collect the four 16-byte sums of a 64-byte line of the staging planes (0 = every sum is stored by itself)
#endif
#ifndef RT_GUIDED_GRAIN
#define RT_GUIDED_GRAIN 4u                     // chunks per claim while plenty of work is left
#endif
#ifndef RT_GUIDED_FACTOR
#define RT_GUIDED_FACTOR 8u
#endif
#ifndef RT_LEAF_REPS
#define RT_LEAF_REPS 2                       // triangle tests per scheduling decision
#endif
#ifndef RT_PHASED_WAVES
#define RT_PHASED_WAVES 5                    // waves per SIMD the phased kernel must fit (96 VGPRs); it fits 6 (80) as built, see Makefile
#endif
#ifndef RT_PHASED_WAVES_WIDE
#define RT_PHASED_WAVES_WIDE 6               // the 8-ary instance is held to 80 VGPRs (12 bytes of scratch): 6 waves per SIMD instead of 5
#endif
constexpr float kTMin = 0.001f;              // main.wgsl:77
constexpr float kTMax = 3.40282347e+38f;     // main.wgsl:77
constexpr uint32_t kNoHit = 0xFFFFFFFFu;

struct Hit {
    float t, u, v;
    uint32_t slot;   // leaf-order triangle slot, kNoHit = miss
    uint32_t gid;
    bool front;
};

struct Tally {      // per-lane counters; the wave-level ones only in counting builds
    unsigned rays = 0, pixel_frames = 0, nodes = 0, tris = 0, w_nodes = 0, w_leaf = 0, w_steps = 0, guard = 0;
};
// RT_GUARD builds (`make guard` -> librt_amd_guard.so, tests/test_gpu_guard.py): every traversal reference is range-checked
// before it is used, stack pointers are bounded and a ray's traversal steps are capped.  A failed check ends the ray and
// counts in DevCounters::guard_trips instead of faulting the GPU.  The shipping build compiles none of it.
#ifdef RT_GUARD
constexpr bool kGuard = true;
#else
constexpr bool kGuard = false;
#endif
constexpr unsigned kGuardMaxSteps = 1u << 22;    // node + leaf steps one ray may take

__device__ __forceinline__ float4 ld4(const void *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ bool first_active_lane() {
    unsigned long long m = __ballot(1);
    return (threadIdx.x & 63u) == (unsigned)(__ffsll((long long)m) - 1);
}

// ray/triangle test of the contract (Moller-Trumbore, explicit fmaf); same
// decisions as tri_test()/consider_tri() of the oracle, evaluated branch-free.
// SPH builds also accept sphere records (build-defined primitive, SURVEY §8 A5;
// oracle sphere_test()/consider_sphere()).
template <bool SPH>
__device__ __forceinline__ void intersect_tri(const DevTri *__restrict__ tris, uint32_t slot, rt_v3 o, rt_v3 d, bool cull,
                                              bool cw, Hit &best) {
    const DevTri *tp = tris + slot;
    float4 q0 = ld4(&tp->ax), q1 = ld4(&tp->e1y), q2 = ld4(&tp->e2z);
    uint32_t gid = __float_as_uint(q2.y);
    if (SPH && __float_as_uint(q2.z) != 0u) {
        const float kInf = __uint_as_float(0x7f800000u);
        rt_v3 oc = o - v3(q0.x, q0.y, q0.z);
        float r = q0.w;
        float a = rt_dot_fma(d, d);
        float hb = rt_dot_fma(oc, d);
        float c = fmaf(-r, r, rt_dot_fma(oc, oc));
        float disc = fmaf(hb, hb, -(a * c));
        float sq = sqrtf(disc);
        float t0 = (-hb - sq) / a, t1 = (-hb + sq) / a;
        float t = (t0 > kTMin && t0 < kInf) ? t0 : t1;
        bool ok = (disc >= 0.0f) & (t > kTMin) & (t < kInf) & ((t < best.t) | ((t == best.t) & (gid < best.gid)));
        if (ok) { best.t = t; best.u = 0.0f; best.v = 0.0f; best.slot = slot; best.gid = gid; best.front = true; }
        return;
    }
    rt_v3 a = v3(q0.x, q0.y, q0.z), e1 = v3(q0.w, q1.x, q1.y), e2 = v3(q1.z, q1.w, q2.x);
    rt_v3 p = rt_cross_fma(d, e2);
    float det = rt_dot_fma(e1, p);
    float inv = 1.0f / det;
    rt_v3 tv = o - a;
    float u = rt_dot_fma(tv, p) * inv;
    rt_v3 q = rt_cross_fma(tv, e1);
    float v = rt_dot_fma(d, q) * inv;
    float t = rt_dot_fma(e2, q) * inv;
    bool front = (det > 0.0f) != cw;                  // cw ? det < 0 : det > 0 wherever det != 0, and det == 0 never hits
    bool ok = (det > 0.0f || det < 0.0f) & (u >= 0.0f) & (u <= 1.0f) & (v >= 0.0f) & (u + v <= 1.0f) & (t > kTMin) &
              (!cull | front) & ((t < best.t) | ((t == best.t) & (gid < best.gid)));
    if (ok) {
        best.t = t; best.u = u; best.v = v; best.slot = slot; best.gid = gid; best.front = front;
    }
}

template <bool COUNT, bool SPH>
__device__ __forceinline__ void intersect_leaf(const DevTri *__restrict__ tris, uint32_t first, uint32_t count,
                                               rt_v3 o, rt_v3 d, bool cull, bool cw, Hit &best, Tally &tl) {
    for (uint32_t i = 0; i < count; i++) {
        if (COUNT) { tl.tris++; if (first_active_lane()) tl.w_leaf++; }
        intersect_tri<SPH>(tris, first + i, o, d, cull, cw, best);
    }
}

// Slab-test constants of one ray (outside the arithmetic contract; conservative).
// A node stores its lower corner `lo` in f32 and the child planes as f16 offsets
// from it (rt_device.h), rounded outward, so per axis
//   base = lo * inv - o * inv          (one FMA per axis per node)
//   t    = off * inv + base            (one v_fma_mix_f32 per plane)
// inv comes from v_rcp_f32 (1 ulp); a zero direction component is nudged to 1e-30
// so no inf/NaN appears.  Error of t: relative 2^-22 |t| from inv and the last
// rounding (covered by the kFarSlack factor), absolute <= 2^-24 (2|o| + |lo|) |inv|
// from o*inv and base, covered PER AXIS by pad * |inv_k|, pad = 2^-20 max_k (|o_k| + R_k),
// R_k = max |coordinate| of the root box (RenderParams::scene_abs): entry distances are
// lowered by it (folded into the origin term), exit distances raised (slab_pad).
// far bound factor of the box test: covers the relative error of t (rcp 2^-23, the phased kernel's stuffed
// inverse 2^-18, two roundings 2^-23) on both sides of the comparison
constexpr float kFarSlack = 1.000012f;
constexpr float kFarSlackStrict = 1.00001216f;   // the next f32 above kFarSlack: tn <= tf * kFarSlack implies tn < tf * kFarSlackStrict (tf > 0; tf <= 0 is a miss either way)
typedef _Float16 rt_h2 __attribute__((ext_vector_type(2)));
// min(a, b) as the bare instruction: fminf() canonicalises an operand the compiler cannot prove quiet (the committed hit's distance:
// one v_max_f32 x, x per node step, a slow-class instruction); neither operand is ever a signalling NaN here
__device__ __forceinline__ float min_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float lo16(uint32_t w) { return (float)__builtin_bit_cast(rt_h2, w).x; }
__device__ __forceinline__ float hi16(uint32_t w) { return (float)__builtin_bit_cast(rt_h2, w).y; }
__device__ __forceinline__ uint4 ld4u(const void *p) { return *reinterpret_cast<const uint4 *>(p); }

// The absolute error of one axis' slab distances as a pad of THAT axis alone.  Worst case over the roundings of the longest chain
// (64-byte 8-ary record: o * inv, grid_lo * inv - that, the pad's own subtraction, cell * inv times the corner index, the base,
// step * inv times the plane byte; grid span <= 3 R): 2^-24 (4 |o| + 9 R) |inv|, R = max |coordinate| of the root box; binary nodes
// 2^-24 (3 |o| + R) |inv|.  pad = 2^-20 max_k (|o_k| + R_k) covers it 1.8 times over.  The box tests lower the entry planes' base by
// pad * |inv_k| and raise the exit planes' by the same (one number per ray: what differs between the axes is |inv|) instead of adding
// the largest of the three products to the far bound.  A shared pad is harmless until a ray runs nearly parallel to an axis:
// |d.x| = 5.6e-9 makes the x pad 1,500 scene units, and as a shared pad that switched the y and z slabs off — such a ray visited
// every node whose x range holds its origin (C4: 10,798 node visits + 26,414 primitive tests against a mean of 27 + 4.8; 19 rays of
// 8.1 M above 1,000 visits), a dependent chain of 25 ms at the end of a launch whose other 8 M rays took 3 ms
// (tools/exp_ray_order/run.py --per-ray).  The relative error of t (the stuffed inverse's 2^-18, roundings) is kFarSlack's part.
__device__ __forceinline__ float slab_pad(const RenderParams &P, rt_v3 o) {
    return 9.5367431640625e-7f * fmaxf(fmaxf(P.scene_abs[0] + fabsf(o.x), P.scene_abs[1] + fabsf(o.y)), P.scene_abs[2] + fabsf(o.z));
}

struct Slab {
    float ix, iy, iz, ox, oy, oz, pad2;     // ox..: -o * inv lowered by the axis' pad; pad2: twice the pad per unit of |inv| (see slab_pad)
};
__device__ __forceinline__ Slab make_slab(const RenderParams &P, rt_v3 o, rt_v3 d) {
    auto nz = [](float x) { return fabsf(x) < 1e-30f ? copysignf(1e-30f, x) : x; };
    Slab s;
    s.ix = __builtin_amdgcn_rcpf(nz(d.x)); s.iy = __builtin_amdgcn_rcpf(nz(d.y)); s.iz = __builtin_amdgcn_rcpf(nz(d.z));
    const float pad = slab_pad(P, o);
    s.ox = fmaf(-pad, fabsf(s.ix), -(o.x * s.ix)); s.oy = fmaf(-pad, fabsf(s.iy), -(o.y * s.iy)); s.oz = fmaf(-pad, fabsf(s.iz), -(o.z * s.iz));
    s.pad2 = pad + pad;
    return s;
}
// both children of one node: entry distances n0/n1, hit flags, references
__device__ __forceinline__ void test_node(const float4 *n, const Slab &sl, float tbest, bool q8, bool &h0, bool &h1, float &n0,
                                          float &n1, int2 &refs) {
    const float4 a = ld4(n);
    const uint4 b = ld4u(n + 1);
    const float bx = fmaf(a.x, sl.ix, sl.ox), by = fmaf(a.y, sl.iy, sl.oy), bz = fmaf(a.z, sl.iz, sl.oz);
    const float px = sl.pad2 * fabsf(sl.ix), py = sl.pad2 * fabsf(sl.iy), pz = sl.pad2 * fabsf(sl.iz);      // what an exit distance is raised by
    float t0x, t1x, t0y, t1y, t0z, t1z, s0x, s1x, s0y, s1y, s0z, s1z;
    if (q8) {
        // two-load node (rt_pack.h): 8-bit plane offsets in units of scale = 2^(e - 20), per axis [c0.lo, c1.lo, c0.hi, c1.hi]
        const uint32_t e5 = rt_q8_exponent(a.x, a.y);
        const float scale = __uint_as_float((e5 + (127u - RT_Q8_EXP_BIAS)) << 23);
        refs.x = __float_as_int(a.w);
        refs.y = (int32_t)b.w;
        const float sx = scale * sl.ix, sy = scale * sl.iy, sz = scale * sl.iz;
        auto ub = [](uint32_t v, int k) { return (float)((v >> (8 * k)) & 255u); };
        t0x = fmaf(ub(b.x, 0), sx, bx); t1x = fmaf(ub(b.x, 2), sx, bx);
        t0y = fmaf(ub(b.y, 0), sy, by); t1y = fmaf(ub(b.y, 2), sy, by);
        t0z = fmaf(ub(b.z, 0), sz, bz); t1z = fmaf(ub(b.z, 2), sz, bz);
        s0x = fmaf(ub(b.x, 1), sx, bx); s1x = fmaf(ub(b.x, 3), sx, bx);
        s0y = fmaf(ub(b.y, 1), sy, by); s1y = fmaf(ub(b.y, 3), sy, by);
        s0z = fmaf(ub(b.z, 1), sz, bz); s1z = fmaf(ub(b.z, 3), sz, bz);
    } else {
        const uint4 c = ld4u(n + 2);
        refs.x = __float_as_int(a.w);
        refs.y = (int32_t)b.w;
        t0x = fmaf(lo16(b.x), sl.ix, bx); t1x = fmaf(hi16(b.x), sl.ix, bx);
        t0y = fmaf(lo16(b.y), sl.iy, by); t1y = fmaf(hi16(b.y), sl.iy, by);
        t0z = fmaf(lo16(b.z), sl.iz, bz); t1z = fmaf(hi16(b.z), sl.iz, bz);
        s0x = fmaf(lo16(c.x), sl.ix, bx); s1x = fmaf(hi16(c.x), sl.ix, bx);
        s0y = fmaf(lo16(c.y), sl.iy, by); s1y = fmaf(hi16(c.y), sl.iy, by);
        s0z = fmaf(lo16(c.z), sl.iz, bz); s1z = fmaf(hi16(c.z), sl.iz, bz);
    }
    n0 = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), kTMin));
    const float f0 = fminf(fminf(fmaxf(t0x, t1x) + px, fmaxf(t0y, t1y) + py), fminf(fmaxf(t0z, t1z) + pz, tbest));
    n1 = fmaxf(fmaxf(fminf(s0x, s1x), fminf(s0y, s1y)), fmaxf(fminf(s0z, s1z), kTMin));
    const float f1 = fminf(fminf(fmaxf(s0x, s1x) + px, fmaxf(s0y, s1y) + py), fminf(fmaxf(s0z, s1z) + pz, tbest));
    h0 = n0 <= f0 * kFarSlack;
    h1 = n1 <= f1 * kFarSlack;
}

// nearest committed hit in (kTMin, kTMax): the ray query of main.wgsl:79-83
template <bool COUNT, bool SPH>
__device__ __forceinline__ Hit trace(const RenderParams &P, uint32_t *__restrict__ stack, rt_v3 o, rt_v3 d, Tally &tl) {
    Hit best;
    best.t = kTMax; best.u = 0.0f; best.v = 0.0f; best.slot = kNoHit; best.gid = kNoHit; best.front = true;
    const bool cull = (P.flags & 1u) != 0;   // flags = 0x10 * (ctx.flags & 1), main.wgsl:76
    const bool cw = P.front_face_cw != 0;

    const Slab sl = make_slab(P, o, d);
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(P.nodes);

    int sp = 0;
    int32_t cur = 0;   // root is always an inner node
    for (;;) {
        while (cur >= 0) {
            if (COUNT) { tl.nodes++; if (first_active_lane()) tl.w_nodes++; }
            bool h0, h1;
            float n0, n1;
            int2 refs;
            test_node(nodes + cur, sl, best.t, P.node_q8 != 0u, h0, h1, n0, n1, refs);
            if (h0 && h1) {
                bool swap = n1 < n0;
                int32_t nearc = swap ? refs.y : refs.x, farc = swap ? refs.x : refs.y;
                stack[sp * kBlock] = (uint32_t)farc;
                sp++;
                cur = nearc;
            } else if (h0) {
                cur = refs.x;
            } else if (h1) {
                cur = refs.y;
            } else {
                if (sp == 0) return best;
                sp--;
                cur = (int32_t)stack[sp * kBlock];
            }
        }
        uint32_t enc = (uint32_t)~cur;   // leaf
        intersect_leaf<COUNT, SPH>(P.tris, enc >> 3, enc & 7u, o, d, cull, cw, best, tl);
        if (sp == 0) return best;
        sp--;
        cur = (int32_t)stack[sp * kBlock];
    }
}


// ---- 8-ary nodes (DevNode8, rt_device.h; RT_OPT_NODE_FORMAT 3) ----
// Traversal state of a ray is two GROUPS instead of one reference: the hit inner children of the last node visited
// (ng_base = chunk index of the node's first inner child, ng_mask = hit bits | imask << 8) and its hit one-primitive
// children (tg_base = first primitive slot, tg_mask = hit bits | lmask << 8).  A child's record is found by counting the
// imask / lmask bits below its slot.  Among the hit children of a group the next one taken is the one with the smallest
// (slot XOR direction signs): child slots are assigned by octant at build time, so that is near-to-far order without
// sorting — looked up in a 2 KiB table in LDS (g_lds_next_slot[signs << 8 | hits]).  The stack holds groups (two words
// per entry); a visit pushes at most one, so it is as deep as the TREE (7 levels for 11k triangles, ~13 for 200 M).
__shared__ unsigned char g_lds_next_slot[8 * 256];

__device__ __forceinline__ void fill_next_slot_table() {
    for (uint32_t i = threadIdx.x; i < 8u * 256u; i += kBlock) {
        const uint32_t flip = i >> 8, hits = i & 255u;
        uint32_t best_slot = 0, best_key = 8;
        for (uint32_t sl = 0; sl < 8; sl++)
            if (((hits >> sl) & 1u) && (sl ^ flip) < best_key) { best_key = sl ^ flip; best_slot = sl; }
        g_lds_next_slot[i] = (unsigned char)best_slot;
    }
    __syncthreads();
}

// inverse direction with the node step's rotate amount (16 for a negative component, else 0) in mantissa bits 4:0
__device__ __forceinline__ float stuffed_inverse(float dcomp) {
    const float nz = fabsf(dcomp) < 1e-30f ? copysignf(1e-30f, dcomp) : dcomp;
    const uint32_t u = __float_as_uint(__builtin_amdgcn_rcpf(nz));
    return __uint_as_float((u & 0xFFFFFFE0u) | ((u >> 27) & 16u));
}

__device__ __forceinline__ float ubyte_f(uint32_t v, int k) { return (float)((v >> (8 * k)) & 255u); }

// One visit: all eight child boxes of the node at 16-byte chunk `chunk` against the ray (ix.. = stuffed inverses; ox.. =
// -o * inv lowered by the axis' OWN pad, pad2 = twice the pad per unit of |inverse|, see slab_pad).  Returns the hit mask over the node's present children.
// Conservative like the binary nodes' test: planes are rounded outward at build time, the far bound is widened by kFarSlack,
// entry distances are computed from the lowered origin term and exit distances from the raised one.
// COMPACT: the 64-byte record (DevNode8c): four fetches; the corner is three 14-bit cell indices of the scene's grid and ox.. are
// expected in GRID space (grid_lo * inv - o * inv, see wide_origin), the plane step is cell * 2^(e - 8) per axis.
template <bool COMPACT>
__device__ __forceinline__ uint32_t test_node8(const RenderParams &P, const float4 *__restrict__ n, float ix, float iy, float iz, float ox, float oy, float oz,
                                               float pad2, float tbest, uint32_t &child_base, uint32_t &tri_base, uint32_t &imask, uint32_t &lmask) {
    float sx, sy, sz, bx, by, bz;                   // b.: base of the entry planes (the origin term comes in lowered by the axis' pad)
    uint4 qx, qy, qz;
    if (COMPACT) {
        const uint4 hd = ld4u(n);
        qx = ld4u(n + 1); qy = ld4u(n + 2); qz = ld4u(n + 3);
        // step = cell * 2^(e - 8): the exponent is added to the bit pattern of (cell * 2^-8), a power of two
        sx = __uint_as_float(P.grid_cell_bits[0] + ((hd.x >> 28) << 23)) * ix;
        sy = __uint_as_float(P.grid_cell_bits[1] + (((hd.y >> 14) & 15u) << 23)) * iy;
        sz = __uint_as_float(P.grid_cell_bits[2] + (((hd.y >> 18) & 15u) << 23)) * iz;
        // corner = grid_lo + index * cell; cell = (cell * 2^-8) * 256
        const float gx = __uint_as_float(P.grid_cell_bits[0] + (8u << 23)) * ix, gy = __uint_as_float(P.grid_cell_bits[1] + (8u << 23)) * iy,
                    gz = __uint_as_float(P.grid_cell_bits[2] + (8u << 23)) * iz;
        bx = fmaf((float)(hd.x & 0x3fffu), gx, ox);
        by = fmaf((float)((hd.x >> 14) & 0x3fffu), gy, oy);
        bz = fmaf((float)(hd.y & 0x3fffu), gz, oz);
        imask = (hd.y >> 22) & 0xffu;
        lmask = (hd.z >> 26) | (((hd.w >> 28) << 6) & 0xc0u);
        child_base = (hd.z & 0x3ffffffu) << 2;              // node index -> 16-byte chunk index
        tri_base = hd.w & 0xfffffffu;
    } else {
        const float4 a = ld4(n);
        const uint4 b = ld4u(n + 1);
        qx = ld4u(n + 2); qy = ld4u(n + 3); qz = ld4u(n + 4);
        const uint32_t w = __float_as_uint(a.w);
        // per-axis scale 2^(byte - 127) times the inverse direction
        sx = __uint_as_float((w & 0xffu) << 23) * ix; sy = __uint_as_float(((w >> 8) & 0xffu) << 23) * iy;
        sz = __uint_as_float(((w >> 16) & 0xffu) << 23) * iz;
        bx = fmaf(a.x, ix, ox); by = fmaf(a.y, iy, oy); bz = fmaf(a.z, iz, oz);
        imask = w >> 24;
        lmask = b.z & 0xffu;
        child_base = b.x;
        tri_base = b.y;
    }
    // base of the exit planes: the entry base raised by twice the axis' pad (one FMA per axis; |.| is an operand modifier)
    const float cx = fmaf(pad2, fabsf(ix), bx), cy = fmaf(pad2, fabsf(iy), by), cz = fmaf(pad2, fabsf(iz), bz);
    const uint32_t rx = __float_as_uint(ix), ry = __float_as_uint(iy), rz = __float_as_uint(iz);
    const uint32_t QX[4] = {qx.x, qx.y, qx.z, qx.w}, QY[4] = {qy.x, qy.y, qy.z, qy.w}, QZ[4] = {qz.x, qz.y, qz.z, qz.w};
    uint32_t h = 0;
    // children 7 .. 0, so that the mask can be built by shifting each hit in from below
#pragma unroll
    for (int p = 3; p >= 0; p--) {
        // [c(2p).lo, c(2p+1).lo, c(2p).hi, c(2p+1).hi] rotated by 0 or 16 bits: bytes 0, 1 = the planes the ray ENTERS through
        const uint32_t px = __builtin_amdgcn_alignbit(QX[p], QX[p], rx), py = __builtin_amdgcn_alignbit(QY[p], QY[p], ry),
                       pz = __builtin_amdgcn_alignbit(QZ[p], QZ[p], rz);
#pragma unroll
        for (int j = 1; j >= 0; j--) {
            const float tn = fmaxf(fmaxf(fmaf(ubyte_f(px, j), sx, bx), fmaf(ubyte_f(py, j), sy, by)), fmaxf(fmaf(ubyte_f(pz, j), sz, bz), kTMin));
            const float tf = fminf(fminf(fmaf(ubyte_f(px, 2 + j), sx, cx), fmaf(ubyte_f(py, 2 + j), sy, cy)), min_raw(fmaf(ubyte_f(pz, 2 + j), sz, cz), tbest));
            // hit <=> tn < tf * slack <=> the sign of tn - tf * slack (one FMA: its sign is that of the exact difference); the mask takes the
            // sign bit in from below with ONE v_alignbit: h = h << 1 | sign.  (v_mul + v_cmp + v_addc before: 11.2 clocks of a SIMD per child
            // against 7.5 by the rates of tools/microbench/calibrate.hip; strict '<' against a slack one ulp wider than kFarSlack loses nothing.)
            const float x = fmaf(tf, -kFarSlackStrict, tn);
            h = __builtin_amdgcn_alignbit(h, __float_as_uint(x), 31);
        }
    }
    return h & (imask | lmask);
}

// -o * inv, or for the compact record the same in the corner grid's space: grid_lo * inv - o * inv
__device__ __forceinline__ float wide_origin(bool compact, float grid_lo, float o, float inv) { return compact ? fmaf(grid_lo, inv, -(o * inv)) : -(o * inv); }
// nearest committed hit through the 8-ary tree, one whole ray per call (kernel variants 0 and 1)
template <bool COUNT, bool SPH, bool COMPACT>
__device__ __forceinline__ Hit trace8(const RenderParams &P, uint32_t *__restrict__ stack, rt_v3 o, rt_v3 d, Tally &tl) {
    Hit best;
    best.t = kTMax; best.u = 0.0f; best.v = 0.0f; best.slot = kNoHit; best.gid = kNoHit; best.front = true;
    const bool cull = (P.flags & 1u) != 0;
    const bool cw = P.front_face_cw != 0;
    constexpr bool compact = COMPACT;
    const uint32_t stride = compact ? (uint32_t)RT_NODE8C_CHUNKS : P.wide_stride;
    const float ix = stuffed_inverse(d.x), iy = stuffed_inverse(d.y), iz = stuffed_inverse(d.z);
    const float pad = slab_pad(P, o), pad2 = pad + pad;
    const float ox = fmaf(-pad, fabsf(ix), wide_origin(compact, P.grid_lo[0], o.x, ix)), oy = fmaf(-pad, fabsf(iy), wide_origin(compact, P.grid_lo[1], o.y, iy)),
                oz = fmaf(-pad, fabsf(iz), wide_origin(compact, P.grid_lo[2], o.z, iz));
    const uint32_t flip256 = ((__float_as_uint(ix) >> 31) | ((__float_as_uint(iy) >> 31) << 1) | ((__float_as_uint(iz) >> 31) << 2)) << 8;
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(P.nodes);
    uint32_t ng_base = 0, ng_mask = 0x0101u, tg_base = 0, tg_mask = 0;   // the root: the only child of a virtual group
    int sp = 0;
    unsigned steps = 0;
    for (;;) {
        if (kGuard && ++steps > kGuardMaxSteps) { tl.guard++; return best; }
        if (tg_mask & 0xffu) {
            const uint32_t slot = g_lds_next_slot[flip256 | (tg_mask & 0xffu)];
            tg_mask &= ~(1u << slot);
            const uint32_t tri = tg_base + (uint32_t)__popc((tg_mask >> 8) & ((1u << slot) - 1u));
            if (kGuard && tri >= P.n_slots) { tl.guard++; return best; }
            if (COUNT) { tl.tris++; if (first_active_lane()) tl.w_leaf++; }
            intersect_tri<SPH>(P.tris, tri, o, d, cull, cw, best);
            continue;
        }
        if (!(ng_mask & 0xffu)) {
            if (sp == 0) return best;
            sp -= 2;
            ng_base = stack[sp * kBlock];
            ng_mask = stack[(sp + 1) * kBlock];
        }
        const uint32_t slot = g_lds_next_slot[flip256 | (ng_mask & 0xffu)];
        ng_mask &= ~(1u << slot);
        const uint32_t node = ng_base + stride * (uint32_t)__popc((ng_mask >> 8) & ((1u << slot) - 1u));
        if (ng_mask & 0xffu) {
            if (kGuard && (uint32_t)(sp + 2) > P.stack_entries) { tl.guard++; }
            else { stack[sp * kBlock] = ng_base; stack[(sp + 1) * kBlock] = ng_mask; sp += 2; }
        }
        if (kGuard && node + stride > P.n_node_chunks) { tl.guard++; return best; }
        if (COUNT) { tl.nodes++; if (first_active_lane()) tl.w_nodes++; }
        uint32_t imask, lmask;
        const uint32_t h = test_node8<COMPACT>(P, nodes + node, ix, iy, iz, ox, oy, oz, pad2, best.t, ng_base, tg_base, imask, lmask);
        ng_mask = (h & imask) | (imask << 8);
        tg_mask = (h & lmask) | (lmask << 8);
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

// shaders/misc.wgsl:6-8 on an Rgba8Unorm texture.  The reference's sampler state is unpinned (rt_abi.h RT_OPT_SAMPLER); P.sampler:
// 0 (default) nearest / clamp-to-edge: texel = clamp(floor(uv * size), 0, size - 1);
// bit 0 linear: texel centres at +0.5, f32 weights, ((T00 (1-a) + T10 a) (1-b)) + ((T01 (1-a) + T11 a) b);
// bit 1 repeat: texel index modulo the size.  Same arithmetic as oracle/rt_oracle.c sample_rgb().
__device__ __forceinline__ int tex_index(float f) {           // floor value -> int; NaN and anything below -2^30 -> -2^30, above 2^30 -> 2^30
    f = (f > -1073741824.0f) ? f : -1073741824.0f;
    f = (f < 1073741824.0f) ? f : 1073741824.0f;
    return (int)f;
}
__device__ __forceinline__ uint32_t tex_address(int i, uint32_t n, bool repeat) {
    if (repeat) { int m = i % (int)n; return (uint32_t)(m < 0 ? m + (int)n : m); }
    return (uint32_t)(i < 0 ? 0 : (i > (int)n - 1 ? (int)n - 1 : i));
}
__device__ __forceinline__ rt_v3 texel_rgb(const RenderParams &P, const DevTexture &t, uint32_t x, uint32_t y) {
    const uint32_t px = P.texels[t.offset + y * t.width + x];
    return v3((float)(px & 255u) / 255.0f, (float)((px >> 8) & 255u) / 255.0f, (float)((px >> 16) & 255u) / 255.0f);
}
__device__ __forceinline__ rt_v3 sample_rgb(const RenderParams &P, uint32_t texture, float u, float v) {
    if (texture >= P.n_textures) return v3(0.0f, 0.0f, 0.0f);
    const DevTexture t = P.textures[texture];
    const bool repeat = (P.sampler & 2u) != 0u;
    if (P.sampler & 1u) {
        const float ux = u * (float)t.width - 0.5f, uy = v * (float)t.height - 0.5f;
        const float x0f = floorf(ux), y0f = floorf(uy);
        float a = ux - x0f, b = uy - y0f;
        a = (a >= 0.0f) ? a : 0.0f; b = (b >= 0.0f) ? b : 0.0f;              // NaN (a non-finite uv) -> 0
        const int x0 = tex_index(x0f), y0 = tex_index(y0f);
        const uint32_t xa = tex_address(x0, t.width, repeat), xb = tex_address(x0 + 1, t.width, repeat);
        const uint32_t ya = tex_address(y0, t.height, repeat), yb = tex_address(y0 + 1, t.height, repeat);
        const rt_v3 t00 = texel_rgb(P, t, xa, ya), t10 = texel_rgb(P, t, xb, ya), t01 = texel_rgb(P, t, xa, yb), t11 = texel_rgb(P, t, xb, yb);
        const float ia = 1.0f - a, ib = 1.0f - b;
        return ((t00 * ia + t10 * a) * ib) + ((t01 * ia + t11 * a) * b);
    }
    if (repeat) {
        const int x0 = tex_index(floorf(u * (float)t.width)), y0 = tex_index(floorf(v * (float)t.height));
        return texel_rgb(P, t, tex_address(x0, t.width, true), tex_address(y0, t.height, true));
    }
    float fx = floorf(u * (float)t.width), fy = floorf(v * (float)t.height);
    fx = (fx > 0.0f) ? fx : 0.0f; fy = (fy > 0.0f) ? fy : 0.0f;          // NaN -> 0
    fx = (fx < (float)(t.width - 1)) ? fx : (float)(t.width - 1);
    fy = (fy < (float)(t.height - 1)) ? fy : (float)(t.height - 1);
    return texel_rgb(P, t, (uint32_t)fx, (uint32_t)fy);
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

// One pixel-frame in flight on a lane.
struct PixelState {
    uint32_t xf, local_row;     // where the result goes: x | frame_offset << 16 (frame of a multi-frame launch)
    float px, py;               // `pos` of main.wgsl:21
    uint32_t seed;
    uint32_t sample, bounce;
    rt_v3 total;                // `color` of main.wgsl:25
    rt_v3 light, color, pos, dir;
    bool first_ray;
};

// a launch of one frame (any width) keeps x whole; multi-frame launches need width < 65536 (checked by the host)
__device__ __forceinline__ uint32_t pixel_x(const RenderParams &P, const PixelState &S) { return P.n_frames > 1u ? (S.xf & 0xffffu) : S.xf; }
__device__ __forceinline__ uint32_t pixel_frame(const RenderParams &P, const PixelState &S) { return P.n_frames > 1u ? (S.xf >> 16) : 0u; }

struct Camera {
    rt_v3 cam, forward, right, up;
    float Wf, Hf;
};

// sample(), main.wgsl:36-41: jitter (x drawn first), primary ray
__device__ __forceinline__ void new_path(const RenderParams &P, const Camera &C, PixelState &S) {
    float r0 = rt_rand(S.seed), r1 = rt_rand(S.seed);
    float offx = (r0 * 2.0f - 1.0f) / C.Wf, offy = (r1 * 2.0f - 1.0f) / C.Hf;
    float ux = (S.px + offx) * P.aspect * P.fov_scale;          // ray.wgsl:6-7
    float uy = (S.py + offy) * P.fov_scale;
    S.dir = rt_normalize((C.forward + C.right * ux) + C.up * uy);   // ray.wgsl:9
    S.pos = C.cam;
    S.light = v3(0.0f, 0.0f, 0.0f);
    S.color = v3(1.0f, 1.0f, 1.0f);
    S.bounce = 0;
}

// main.wgsl:17-25 for pixel (x, y)
__device__ __forceinline__ void begin_pixel(const RenderParams &P, const Camera &C, PixelState &S, uint32_t x, uint32_t y,
                                            uint32_t local_row, uint32_t frame_off = 0) {
    S.xf = x | (frame_off << 16);
    S.local_row = local_row;
    const uint32_t pixel_idx = y * P.width + x;
    S.px = (float)x / C.Wf - 0.5f;                     // main.wgsl:20-21
    S.py = (1.0f - (float)y / C.Hf) - 0.5f;
    S.seed = rt_seed(pixel_idx, P.frame + frame_off);  // main.wgsl:23
    S.total = v3(0.0f, 0.0f, 0.0f);
    S.sample = 0;
    S.first_ray = true;
}

// main.wgsl:27-29: mean of the samples, exposure, tone map
__device__ __forceinline__ rt_v3 resolve_pixel(const RenderParams &P, rt_v3 sum) {
    const float ns = (float)P.samples;
    rt_v3 total = v3(sum.x / ns, sum.y / ns, sum.z / ns);
    return v3(tone_map1(total.x * P.exposure), tone_map1(total.y * P.exposure), tone_map1(total.z * P.exposure));
}

// main.wgsl:27-32
__device__ __forceinline__ void finish_pixel(const RenderParams &P, const PixelState &S) {
    const size_t pix = (size_t)S.local_row * P.width + pixel_x(P, S);
    if (P.n_frames > 1u) {
        // multi-frame launch: frames of a pixel finish in any order, so the sample SUM is parked per frame and
        // rt_blend_frames applies main.wgsl:27-32 to the frames in order afterwards (with every lane busy, which
        // the few lanes that end a pixel-frame together here are not)
        P.stage[(size_t)pixel_frame(P, S) * P.stage_stride + pix] = make_float4(S.total.x, S.total.y, S.total.z, 0.0f);
        return;
    }
    const rt_v3 mapped = resolve_pixel(P, S.total);
    const float w = 1.0f / (float)(P.accumulation_frame + 1u);
    float4 *dst = reinterpret_cast<float4 *>(P.accum) + pix;
    float4 old = *dst;
    rt_v3 out = rt_mix(v3(old.x, old.y, old.z), mapped, w);
    *dst = make_float4(out.x, out.y, out.z, old.w);
}

// Everything of sample()'s loop body after the trace (main.wgsl:46-69).
// Returns true when the path ended (miss, or the bounce budget is spent).
// SPH: sphere records may be hit; TEX: materials may name textures (the phased kernel has an instance without that code: EXT below)
template <bool AOV, bool SPH, bool LDSM = false, bool TEX = SPH>
__device__ __forceinline__ bool shade_core(const RenderParams &P, PixelState &S, const Hit &h_in) {
    Hit h = h_in;
    if (AOV && S.first_ray) {
        uint32_t g = kNoHit, pr = kNoHit;
        if (h.slot != kNoHit) { g = P.shade[h.slot].geometry; pr = P.shade[h.slot].primitive; }
        uint2 *dst = reinterpret_cast<uint2 *>(P.aov) + (size_t)S.local_row * P.width + pixel_x(P, S);
        *dst = make_uint2(g, pr);
    }
    S.first_ray = false;
    bool path_done;
    if (h.slot == kNoHit) {                                         // main.wgsl:46-50
        float a = 0.5f * (S.dir.y + 1.0f);                          // misc.wgsl:1-4
        float w = 1.0f - a;
        rt_v3 bg = v3(w * 1.0f + a * 0.5f, w * 1.0f + a * 0.7f, w * 1.0f + a * 1.0f);
        S.light = S.light + (bg * S.color) * P.environment;
        path_done = true;
    } else {
        // attribute part of trace_ray, main.wgsl:86-101
        const DevShade *sh = P.shade + h.slot;
        const float4 s0 = ld4(&sh->p0[0]), s1 = ld4(&sh->p1[0]), s2 = ld4(&sh->p2[0]), s3 = ld4(&sh->n1[0]), s4 = ld4(&sh->n2[0]);
        uint32_t geometry = __float_as_uint(s3.w);
        const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
        rt_v3 normal = v3(s0.w * b0 + s3.x * b1 + s4.x * b2,
                          s1.w * b0 + s3.y * b1 + s4.y * b2,
                          s2.w * b0 + s3.z * b1 + s4.z * b2);
        rt_v3 position = v3(s0.x * b0 + s1.x * b1 + s2.x * b2,
                            s0.y * b0 + s1.y * b1 + s2.y * b2,
                            s0.z * b0 + s1.z * b1 + s2.z * b2);
        if (SPH && (geometry & 0x80000000u)) {
            // sphere: P = o + t d, N = (P - C) / r, front = dot(d, N) < 0; the record
            // holds C in p0, 1/r in n0x and the model index in pad0
            position = S.pos + S.dir * h.t;
            normal = (position - v3(s0.x, s0.y, s0.z)) * s0.w;
            h.front = rt_dot(S.dir, normal) < 0.0f;
            geometry = sh->pad0;
        }
        // models[geometry_index].material, chunk k of its record: LDSM instances read the workgroup's LDS copy
        auto mat_chunk = [&](uint32_t k) -> float4 {
            if (LDSM) return g_lds_materials[geometry * 4u + k];
            return ld4(reinterpret_cast<const float4 *>(P.materials + geometry) + k);
        };
        const float4 m0 = mat_chunk(0), m1 = mat_chunk(1), m2 = mat_chunk(2);
        const uint32_t tag = __float_as_uint(m0.x);
        if (tag == 0u) {                                            // main.wgsl:53-61
            // get_scattered_direction_metal, ray.wgsl:12-32
            float is_specular = (rt_rand(S.seed) < m1.x) ? 1.0f : 0.0f;
            float smoothness = 1.0f - m2.x;
            rt_v3 scatter_normal = normal;
            rt_v3 diffuse_color = v3(m0.y, m0.z, m0.w);
            const float4 m3 = TEX ? mat_chunk(3) : m0;               // texture ids: only instances with textures read them
            const uint32_t mat_diffuse_texture = __float_as_uint(m3.y), mat_normal_texture = __float_as_uint(m3.z);
            if (TEX && (mat_normal_texture | mat_diffuse_texture) != 0u) {
                // trace.uv, main.wgsl:96 (spheres: uv = 0)
                float tu = 0.0f, tv = 0.0f;
                if (!(__float_as_uint(s3.w) & 0x80000000u)) {
                    const float4 u01 = ld4(&sh->uv0[0]);
                    const float2 u2 = *reinterpret_cast<const float2 *>(&sh->uv2[0]);
                    tu = u01.x * b0 + u01.z * b1 + u2.x * b2;
                    tv = u01.y * b0 + u01.w * b1 + u2.y * b2;
                }
                if (mat_normal_texture > 0u) {                              // ray.wgsl:17-20
                    rt_v3 smp = sample_rgb(P, mat_normal_texture - 1u, tu, tv);
                    scatter_normal = tangent_space(normal, v3(smp.x * 2.0f - 1.0f, smp.y * 2.0f - 1.0f, smp.z * 2.0f - 1.0f));
                }
                if (mat_diffuse_texture > 0u) diffuse_color = sample_rgb(P, mat_diffuse_texture - 1u, tu, tv);   // ray.wgsl:26
            }
            rt_v3 diffuse = rand_cosine_hemisphere_vector(S.seed, scatter_normal);
            rt_v3 specular = rt_reflect(S.dir, scatter_normal);
            rt_v3 sdir = rt_mix(diffuse, specular, smoothness * is_specular);
            rt_v3 scol = rt_mix(diffuse_color, v3(m1.y, m1.z, m1.w), is_specular);
            S.light = S.light + v3(m2.y, m2.z, m2.w) * S.color;
            S.color = S.color * scol;
            S.pos = position + normal * 0.0001f;
            S.dir = sdir;
        } else if (tag == 1u) {                                     // main.wgsl:62-68
            // get_scattered_direction_dielectric, ray.wgsl:34-49
            rt_v3 n = rt_dot(normal, S.dir) < 0.0f ? normal : -normal;   // faceForward(N, N, dir)
            const float ri_in = mat_chunk(3).x;
            float ri = ri_in;
            if (h.front) ri = 1.0f / ri_in;
            float c = rt_dot(-S.dir, n);
            float cos_theta = (c < 1.0f) ? c : 1.0f;
            float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
            bool must_reflect = ri * sin_theta > 1.0f;
            float reflect_prob = schlick_approximation(cos_theta, ri);
            bool refl = must_reflect || (reflect_prob > rt_rand(S.seed));   // || short-circuits
            rt_v3 next_dir = refl ? rt_reflect(S.dir, n) : rt_refract(S.dir, n, ri);
            rt_v3 offset_dir = normal - (normal * 2.0f) * (h.front ? 1.0f : 0.0f);
            S.pos = position + offset_dir * 0.0001f;
            S.dir = next_dir;
        }
        S.bounce++;
        path_done = S.bounce > P.max_bounces;                       // `bounce <= max_bounces`, main.wgsl:43
    }
    return path_done;
}

// shade_core plus the sample bookkeeping of the flattened loops.
// Returns true when the pixel-frame is complete.
template <bool AOV, bool SPH>
__device__ __forceinline__ bool shade(const RenderParams &P, const Camera &C, PixelState &S, const Hit &h) {
    if (shade_core<AOV, SPH>(P, S, h)) {
        S.total = S.total + S.light;                                // main.wgsl:26
        S.sample++;
        if (S.sample >= P.samples) return true;
        new_path(P, C, S);
    }
    return false;
}

template <bool COUNT, bool AOV, bool SPH>
__device__ __forceinline__ bool step(const RenderParams &P, const Camera &C, uint32_t *stack, PixelState &S, Tally &tl) {
    Hit h = P.node_wide == 2u ? trace8<COUNT, SPH, true>(P, stack, S.pos, S.dir, tl)
          : P.node_wide     ? trace8<COUNT, SPH, false>(P, stack, S.pos, S.dir, tl) : trace<COUNT, SPH>(P, stack, S.pos, S.dir, tl);   // main.wgsl:44
    tl.rays++;
    return shade<AOV, SPH>(P, C, S, h);
}

__device__ __forceinline__ Camera load_camera(const RenderParams &P) {
    Camera C;
    C.cam = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    C.forward = v3(P.forward[0], P.forward[1], P.forward[2]);
    C.right = v3(P.right[0], P.right[1], P.right[2]);
    C.up = v3(P.up[0], P.up[1], P.up[2]);
    C.Wf = (float)P.width;
    C.Hf = (float)P.height;
    return C;
}

// exact device-side counters: one atomic per wave per counter
template <bool COUNT>
__device__ __forceinline__ void flush_tally(const RenderParams &P, const Tally &tl) {
    unsigned long long v[8] = {tl.rays, tl.pixel_frames, tl.nodes, tl.tris, tl.w_nodes, tl.w_leaf, tl.w_steps, tl.guard};
    const int n = kGuard ? 8 : (COUNT ? 7 : 2);
    for (int k = 0; k < n; k++)
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
    if ((threadIdx.x & 63u) == 0) {
        unsigned long long *dst = &P.counters->rays;
        for (int k = 0; k < n; k++) atomicAdd(dst + k, v[k]);
    }
}

// slot (shard-local pixel slot: tile-major, 64 slots per 8x8 tile) -> pixel.
// Bands of 8 rows are dealt round-robin to shards (rt_abi.h rt_set_shard);
// pixel_idx stays global (main.wgsl:19).
__device__ __forceinline__ bool slot_to_pixel(const RenderParams &P, uint32_t slot, uint32_t &x, uint32_t &y, uint32_t &local_row) {
    const uint32_t tile = slot >> 6, l = slot & 63u;
    // tile / tiles_x without the ~30-instruction u32 division: float quotient, corrected by +-1
    // (exact for tile < 2^24; bigger images take the integer path)
    uint32_t band_local;
    if (tile < (1u << 24)) {
        band_local = (uint32_t)((float)tile * P.inv_tiles_x);
        int32_t r = (int32_t)(tile - band_local * P.tiles_x);
        if (r < 0) band_local--;
        else if (r >= (int32_t)P.tiles_x) band_local++;
    } else {
        band_local = tile / P.tiles_x;
    }
    const uint32_t tx = tile - band_local * P.tiles_x;
    x = tx * 8u + (l & 7u);
    y = (band_local * P.shard_count + P.shard_index) * 8u + (l >> 3);
    local_row = band_local * 8u + (l >> 3);
    return x < P.width && y < P.height;                 // main.wgsl:17
}

#ifdef RT_TU_SIMPLE
// ---- variant 0: persistent wavefronts pulling pixel slots from a queue ----
template <bool COUNT, bool AOV, bool SPH>
__global__ __launch_bounds__(kBlock) void rt_render_persistent(const RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    if (P.node_wide) fill_next_slot_table();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_slots = P.tiles_x * P.local_bands * 64u;
    const Camera C = load_camera(P);
    Tally tl;
    PixelState S;
    bool busy = false;
    uint32_t w_next = 0, w_end = 0;   // the wave's claimed chunk [w_next, w_end), wave-uniform
    bool drained = false;             // queue exhausted (wave-uniform)

    for (;;) {
        unsigned long long idle = __ballot(!busy);
        while (idle != 0ull) {
            if (w_next == w_end) {
                if (drained) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&P.queue->head, P.chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base >= total_slots) { drained = true; break; }
                w_next = base;
                w_end = min(base + P.chunk, total_slots);
            }
            const uint32_t avail = w_end - w_next;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t take = min(avail, n_idle);
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            if (!busy && rank < take) {
                uint32_t x, y, lr;
                if (slot_to_pixel(P, w_next + rank, x, y, lr)) {
                    begin_pixel(P, C, S, x, y, lr);
                    new_path(P, C, S);
                    busy = P.samples > 0;
                    if (!busy) { S.total = v3(0.0f, 0.0f, 0.0f); finish_pixel(P, S); tl.pixel_frames++; }
                }
            }
            w_next += take;
            idle = __ballot(!busy);
        }
        if (__ballot(busy) == 0ull) break;
        if (busy) {
            if (COUNT && first_active_lane()) tl.w_steps++;
            if (step<COUNT, AOV, SPH>(P, C, stack, S, tl)) {
                finish_pixel(P, S);
                tl.pixel_frames++;
                busy = false;
            }
        }
    }
    flush_tally<COUNT>(P, tl);
}

#endif  // RT_TU_SIMPLE

#if defined(RT_TU_PHASED) || defined(RT_TU_DEVICE_ONLY)
// ---- variant 2: persistent wavefronts, lanes scheduled by phase ----
// Ray lengths differ wildly inside a wave (a sky ray visits 3 nodes, a ray through
// the teapot 80), so a wave that traces "one ray per lane" waits for its slowest
// lane: measured lane utilization of the node loop in variant 0 is 15 %.  Here a
// lane is a small state machine — FETCH (needs a pixel), NODE (at an inner node),
// LEAF (has triangles to test), SHADE (trace finished) — and each loop iteration
// the wave runs ONE phase body, the one most lanes are waiting for (ballot counts).
// Lanes in other phases sit that iteration out; nobody waits for a whole ray.
// Wave priority by phase (s_setprio): a level for node steps, one for leaf steps, 0 otherwise (shading, fetch).
// Binary nodes (what cache-resident scenes get): node steps 2, leaf steps 1 — the fetch-issuing phases win the issue arbitration
// against other waves' long shading steps and the L1 path stays fed: C3 15.95 -> 15.25 ms per step, C2 3.49 -> 3.23 (levels 1..3
// all within 0.5 %; raising SHADE instead: 16.65; lattice scenes: -5 % at 18 MiB of records, -1..-2 % up to 218 MiB, +2 % at 909 MiB).
// 8-ary nodes (what the library picks once the nodes leave the caches): raising the NODE steps — 208 instructions behind a round
// trip to L2 / HBM — only delays the shading that starts the next rays (C4 100.4 -> 105 ms, lattice scenes +4..+6 % from 100 MiB of
// records on); raising the LEAF steps alone (three fetches, 60 instructions) is worth 2 % (C4 100.9 -> 98.5, C5q 57.7 -> 56.3).
// The rule is by FORMAT and so costs no branch (a run-time flag cost 1.2 % on C3).
#ifndef RT_PRIO_NODE_BINARY
#define RT_PRIO_NODE_BINARY 2
#define RT_PRIO_LEAF_BINARY 1
#define RT_PRIO_NODE_WIDE 0
#define RT_PRIO_LEAF_WIDE 2
#endif
enum : int { PH_NODE = 0, PH_LEAF = 1, PH_SHADE = 2, PH_FETCH = 4, PH_DONE = 5 };
constexpr uint32_t kFreshPixel = 0xFFFFFFFFu;   // PixelState::bounce of a pixel-frame that has not started its first path

// FMT: node format — 0 = 48-byte binary nodes, 1 = 32-byte binary nodes (q8), 2 = 80-byte 8-ary nodes (DevNode8), 3 = 64-byte 8-ary nodes (DevNode8c)
// RAYLIST (developer instances, tools/exp_ray_order/): the same scheduler and traversal over a LIST of rays instead of pixels — FETCH
// takes ray P.aov[q] (or q when P.aov is null) of the P.stage_stride rays in P.stage (origin, direction: two float4 each) and SHADE
// only stores the committed hit (t bits, slot) to ((uint2 *)P.accum)[ray]: what a wavefront tracer's trace pass would run.
// EXT: 0 = triangles, untextured; 1 = + sphere records; 2 = + textures (the sampler's code costs the sphere instance registers it has no use for)
template <bool COUNT, bool AOV, int EXT, bool LDSM, int FMT, bool RAYLIST = false>
__global__ __launch_bounds__(kBlock, FMT >= 2 ? RT_PHASED_WAVES_WIDE : RT_PHASED_WAVES) void rt_render_phased(const RenderParams P) {
    constexpr bool Q8 = FMT == 1, WIDE = FMT >= 2, COMPACT = FMT == 3, SPH = EXT >= 1, TEX = EXT >= 2;
    constexpr int kPrioNode = WIDE ? RT_PRIO_NODE_WIDE : RT_PRIO_NODE_BINARY, kPrioLeaf = WIDE ? RT_PRIO_LEAF_WIDE : RT_PRIO_LEAF_BINARY;
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
#if RT_STAGE_LINES > 0
    // multi-frame launches: a finished pixel-frame's sum (16 bytes) waits here for the other three of its 64-byte line of the staging
    // planes and the line is stored whole; a slot that is needed for another line first gives up what it holds as single sums
    __shared__ float4 s_lines[kBlock / 64][RT_STAGE_LINES][4];
    __shared__ uint32_t s_line_tag[kBlock / 64][RT_STAGE_LINES];      // float4 index of the line / 4 + 1; 0 = free
    __shared__ uint32_t s_line_mask[kBlock / 64][RT_STAGE_LINES];     // which of the four sums are there
    if ((threadIdx.x & 63u) < (uint32_t)RT_STAGE_LINES) s_line_tag[threadIdx.x >> 6][threadIdx.x & 63u] = 0u;
#endif
    if (WIDE) fill_next_slot_table();
    if (LDSM) {       // the host picks this instance when P.n_materials <= RT_LDS_MATERIALS
        if (threadIdx.x < P.n_materials * 4u) g_lds_materials[threadIdx.x] = reinterpret_cast<const float4 *>(P.materials)[threadIdx.x];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t frame_slots = P.tiles_x * P.local_bands * 64u;
    const uint32_t total_slots = RAYLIST ? (uint32_t)P.stage_stride : frame_slots * P.n_frames;    // work items: frame-major (host keeps this below 2^32)
    const Camera C = load_camera(P);
    const bool cull = (P.flags & 1u) != 0;
    const bool cw = P.front_face_cw != 0;
    const float4 *__restrict__ nodes = reinterpret_cast<const float4 *>(P.nodes);
    Tally tl;
    PixelState S;
    Hit best;
    // slab constants of the current ray (see make_slab): kept as scalars, not
    // pairs, so the loop carries them in place
    float ix = 0, iy = 0, iz = 0, ox = 0, oy = 0, oz = 0, eps = 0;     // ox.. = -o * inv lowered by the axis' pad, eps = twice the pad per unit of |inv| (slab_pad)
    int phase = PH_FETCH;
    bool have_pixel = false;  // FETCH: a finished pixel-frame waits to be written out
    int32_t cur = 0;          // NODE: inner node reference (float4 index); LEAF: leaf reference (< 0)
    // 8-ary nodes: the ray's two groups (see test_node8) and the table row of its direction signs
    uint32_t ng_base = 0, ng_mask = 0, tg_base = 0, tg_mask = 0, flip256 = 0;
    uint32_t *sptr = stack;   // next free entry of the lane's LDS stack (entries are kBlock words apart)
    unsigned guard_steps = 0; // RT_GUARD: traversal steps of the current ray
    // 8-ary nodes: an entry is a group = (base, 16-bit mask), 6 bytes: the bases of all levels in one plane of words, the masks in a
    // plane of halfwords behind it, both indexed [level * kBlock + thread] (P.stack_entries counts words of the 8-byte entries the
    // other kernel variants keep: two per level).  C4: 13 levels x 6 B x 256 lanes = 19.5 KB instead of 26 KB per workgroup.
    const uint32_t stack_levels = WIDE ? P.stack_entries / 2u : P.stack_entries;
    uint32_t *const stack_end = stack + (size_t)stack_levels * kBlock;
    unsigned short *const mask_plane = reinterpret_cast<unsigned short *>(lds_stack + (size_t)stack_levels * kBlock);
    uint32_t w_next = 0, w_end = 0;   // the wave's claimed chunk, wave-uniform
    bool drained = false;
    uint32_t ng_claims = 0;           // RAYLIST: chunks this wave has taken

    auto start_trace = [&]() {        // rayQueryInitialize, main.wgsl:80
        auto nz = [](float x) { return fabsf(x) < 1e-30f ? copysignf(1e-30f, x) : x; };
        // the inverse direction carries, in mantissa bits 4:0, the rotate amount of the node step: 16 for a
        // negative component (enter through the upper plane), else 0.  Costs 2^-18 of relative accuracy,
        // which kFarSlack pays for.
        auto stuff = [](float inv) { const uint32_t u = __float_as_uint(inv); return __uint_as_float((u & 0xFFFFFFE0u) | ((u >> 27) & 16u)); };
        ix = stuff(__builtin_amdgcn_rcpf(nz(S.dir.x))); iy = stuff(__builtin_amdgcn_rcpf(nz(S.dir.y))); iz = stuff(__builtin_amdgcn_rcpf(nz(S.dir.z)));
        ox = -(S.pos.x * ix); oy = -(S.pos.y * iy); oz = -(S.pos.z * iz);
        // per-axis pads: the entry planes' origin term lowered by pad * |inv| here, the exit planes' base raised by eps * |inv| = twice that
        // in the node step, see slab_pad
        const float pad = slab_pad(P, S.pos);
        if (COMPACT) { ox = fmaf(P.grid_lo[0], ix, ox); oy = fmaf(P.grid_lo[1], iy, oy); oz = fmaf(P.grid_lo[2], iz, oz); }   // into the corner grid's space
        ox = fmaf(-pad, fabsf(ix), ox); oy = fmaf(-pad, fabsf(iy), oy); oz = fmaf(-pad, fabsf(iz), oz);
        eps = pad + pad;
        best.t = kTMax; best.u = 0.0f; best.v = 0.0f; best.slot = kNoHit; best.gid = kNoHit; best.front = true;
        cur = 0; sptr = stack;
        if (WIDE) {
            ng_base = 0; ng_mask = 0x0101u; tg_mask = 0;      // the root: the only child of a virtual group
            flip256 = ((__float_as_uint(ix) >> 31) | ((__float_as_uint(iy) >> 31) << 1) | ((__float_as_uint(iz) >> 31) << 2)) << 8;
        }
        guard_steps = 0;
        // A ray with a NaN component hits nothing under the contract (every comparison of every primitive test fails; the oracle's
        // brute force agrees) but v_max3 / v_min3 drop NaN operands, so the box test would let it into EVERY node: it goes straight
        // to SHADE as a miss.  (A zero vertex normal is enough to make one: tangent_space() normalises a zero cross product.)
        const bool nan_ray = __builtin_isunordered(S.dir.x, S.dir.y) | __builtin_isunordered(S.dir.z, S.pos.x) | __builtin_isunordered(S.pos.y, S.pos.z);
        phase = nan_ray ? PH_SHADE : PH_NODE;
    };
    // 8-ary nodes: what a lane does next once its current step is done — primitives of the node just visited first, then
    // its inner children, then the youngest group on the stack; nothing left = the trace is finished
    auto wide_advance = [&]() {
        if (tg_mask & 0xffu) phase = PH_LEAF;
        else if (ng_mask & 0xffu) phase = PH_NODE;
        else if (sptr == stack) phase = PH_SHADE;
        else { sptr -= kBlock; ng_base = sptr[0]; ng_mask = mask_plane[sptr - lds_stack]; phase = PH_NODE; }
    };
    // wave-uniform choice of the next phase body: the one most lanes wait for,
    // weighted (P.tune) so cheap bodies run at lower occupancy than expensive ones
    auto schedule = [&]() -> int {
        const int c_fetch = __popcll(__ballot(phase == PH_FETCH));
        const int c_node = __popcll(__ballot(phase == PH_NODE));
        const int c_leaf = __popcll(__ballot(phase == PH_LEAF));
        const int c_shade = __popcll(__ballot(phase == PH_SHADE));
        if ((c_fetch | c_node | c_leaf | c_shade) == 0) return PH_DONE;
        int pick = PH_NODE, best_c = c_node * P.tune[0];
        if (c_leaf * P.tune[1] > best_c) { pick = PH_LEAF; best_c = c_leaf * P.tune[1]; }
        if (c_shade * P.tune[2] > best_c) { pick = PH_SHADE; best_c = c_shade * P.tune[2]; }
        if (c_fetch * P.tune[4] > best_c) { pick = PH_FETCH; }
        // wave-uniform by construction: ballot counts (s_bcnt1 of lane masks) and kernel arguments only, all in SGPRs.  The file
        // is built with -structurizecfg-skip-uniform-regions, which keeps branches on such values scalar.
        // (An explicit readfirstlane here was tried: it moves the value through a VGPR and the kernel then needs 96 VGPRs with
        // 80 bytes of scratch instead of 78 and none.)
        return pick;
    };

#ifdef RT_TAIL_PROBE
    const unsigned long long t_start = wall_clock64();
    unsigned long long t_drain = 0;
#endif
    int pick = schedule();
    while (pick != PH_DONE) {
        if (pick == PH_NODE) {
            if constexpr (kPrioNode != 0) __builtin_amdgcn_s_setprio(kPrioNode);
            do {
                const int node_lanes = __popcll(__ballot(phase == PH_NODE));
#pragma unroll
                for (int rep = 0; rep < RT_NODE_REPS; rep++) {
                // back to the scheduler early once fewer than 1/RT_NODE_EARLY_OUT(_WIDE) of the lanes that came in are still on nodes
                if (rep > 0 && __popcll(__ballot(phase == PH_NODE)) * (WIDE ? RT_NODE_EARLY_OUT_WIDE : RT_NODE_EARLY_OUT) < node_lanes) break;
                if (kGuard && !WIDE && phase == PH_NODE &&
                    ((uint32_t)cur + (Q8 ? 2u : 3u) > P.n_node_chunks || ++guard_steps > kGuardMaxSteps)) { tl.guard++; phase = PH_SHADE; }
                if (WIDE) {
                    if (phase == PH_NODE) {
                        const uint32_t slot = g_lds_next_slot[flip256 | (ng_mask & 0xffu)];
                        ng_mask &= ~(1u << slot);
                        const uint32_t node = ng_base + (COMPACT ? (uint32_t)RT_NODE8C_CHUNKS : P.wide_stride) * (uint32_t)__popc((ng_mask >> 8) & ((1u << slot) - 1u));
                        if (kGuard && (node + (COMPACT ? 4u : 5u) > P.n_node_chunks || ++guard_steps > kGuardMaxSteps)) { tl.guard++; phase = PH_SHADE; }
                        else {
                            if (ng_mask & 0xffu) {               // siblings still to visit: the group goes on the stack
                                if (kGuard && sptr >= stack_end) { tl.guard++; }
                                else { sptr[0] = ng_base; mask_plane[sptr - lds_stack] = (unsigned short)ng_mask; sptr += kBlock; }
                            }
                            if (COUNT) { tl.nodes++; if (first_active_lane()) tl.w_nodes++; }
#ifdef RT_EXP_NODE_VALU      // sensitivity builds (tools/sensitivity.sh)
                            { float dmy = ix;
#pragma unroll
                              for (int q = 0; q < RT_EXP_NODE_VALU; q++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dmy));
                            }
#endif
#ifdef RT_EXP_NODE_LOAD
                            {
#pragma unroll
                              for (int q = 0; q < RT_EXP_NODE_LOAD; q++) { float4 dmy; asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(dmy) : "v"(nodes + node) : "memory"); }
                            }
#endif
                            uint32_t imask, lmask;
                            const uint32_t h = test_node8<COMPACT>(P, nodes + node, ix, iy, iz, ox, oy, oz, eps, best.t, ng_base, tg_base, imask, lmask);
                            ng_mask = (h & imask) | (imask << 8);
                            tg_mask = (h & lmask) | (lmask << 8);
                            wide_advance();
                        }
                    }
                } else
                if (phase == PH_NODE) {
                    if (COUNT) { tl.nodes++; if (first_active_lane()) tl.w_nodes++; }
                    const float4 *n = nodes + cur;
                    float n0, n1, f0, f1;
                    int2 refs;
#ifdef RT_EXP_NODE_VALU      // sensitivity builds (tools/sensitivity.sh): N extra dependent VALU instructions per node step
                    { float dmy = ix;
#pragma unroll
                      for (int q = 0; q < RT_EXP_NODE_VALU; q++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dmy));
                    }
#endif
#ifdef RT_EXP_NODE_LOAD      // sensitivity builds: N extra 16-byte fetches of the node's own first chunk (same cache line)
                    {
#pragma unroll
                      for (int q = 0; q < RT_EXP_NODE_LOAD; q++) { float4 dmy; 
#if defined(RT_EXP_NODE_LOAD_HOT)      /* a different, L1-resident line per lane (the top of the tree), not waited for by itself */
                          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dmy) : "v"(nodes + 3u * (lane ^ (uint32_t)q * 17u)) : "memory");
#elif defined(RT_EXP_NODE_LOAD_NOWAIT)   /* issued ahead of the node's own fetches and never waited for by itself: an older load only makes the compiler's vmcnt waits longer */
                          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dmy) : "v"(n) : "memory");
#else
                          asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(dmy) : "v"(n) : "memory");
#endif
                      }
                    }
#endif
                    // rotate amounts: bits 4:0 of the stuffed inverse (see start_trace), 16 for a negative direction
                    const uint32_t rx = __float_as_uint(ix), ry = __float_as_uint(iy), rz = __float_as_uint(iz);
                    if (Q8) {
                        // 32-byte node (rt_pack.h): two loads; per axis one dword [c0.lo, c1.lo, c0.hi, c1.hi] of 8-bit
                        // offsets in units of scale = 2^(e - 20); rotating it by 0 or 16 bits puts the planes the ray
                        // ENTERS through into bytes 0 and 1
                        const float4 a = ld4(n);
                        const uint4 b = ld4u(n + 1);
                        const uint32_t e5 = rt_q8_exponent(a.x, a.y);       // rides in the corner's lowest mantissa bits
                        const float scale = __uint_as_float((e5 + (127u - RT_Q8_EXP_BIAS)) << 23);
                        refs = make_int2(__float_as_int(a.w), (int32_t)b.w);
                        const float bx = fmaf(a.x, ix, ox), by = fmaf(a.y, iy, oy), bz = fmaf(a.z, iz, oz);
                        const float cx = fmaf(eps, fabsf(ix), bx), cy = fmaf(eps, fabsf(iy), by), cz = fmaf(eps, fabsf(iz), bz);   // exit planes' base
                        const float sx = scale * ix, sy = scale * iy, sz = scale * iz;
                        const uint32_t px = __builtin_amdgcn_alignbit(b.x, b.x, rx), py = __builtin_amdgcn_alignbit(b.y, b.y, ry),
                                       pz = __builtin_amdgcn_alignbit(b.z, b.z, rz);
                        auto ub = [](uint32_t v, int k) { return (float)((v >> (8 * k)) & 255u); };
                        n0 = fmaxf(fmaxf(fmaf(ub(px, 0), sx, bx), fmaf(ub(py, 0), sy, by)), fmaxf(fmaf(ub(pz, 0), sz, bz), kTMin));
                        n1 = fmaxf(fmaxf(fmaf(ub(px, 1), sx, bx), fmaf(ub(py, 1), sy, by)), fmaxf(fmaf(ub(pz, 1), sz, bz), kTMin));
                        f0 = fminf(fminf(fmaf(ub(px, 2), sx, cx), fmaf(ub(py, 2), sy, cy)), min_raw(fmaf(ub(pz, 2), sz, cz), best.t));
                        f1 = fminf(fminf(fmaf(ub(px, 3), sx, cx), fmaf(ub(py, 3), sy, cy)), min_raw(fmaf(ub(pz, 3), sz, cz), best.t));
                    } else {
                        // 48-byte node: three loads; rotating each (hi << 16 | lo) f16 offset pair by 0 or 16 bits makes the
                        // low half the plane the ray ENTERS through, no min/max needed
                        const float4 a = ld4(n);
                        const uint4 b = ld4u(n + 1), c = ld4u(n + 2);
                        refs = make_int2(__float_as_int(a.w), (int32_t)b.w);
                        const float bx = fmaf(a.x, ix, ox), by = fmaf(a.y, iy, oy), bz = fmaf(a.z, iz, oz);
                        const float cx = fmaf(eps, fabsf(ix), bx), cy = fmaf(eps, fabsf(iy), by), cz = fmaf(eps, fabsf(iz), bz);   // exit planes' base
                        const uint32_t p0x = __builtin_amdgcn_alignbit(b.x, b.x, rx), p0y = __builtin_amdgcn_alignbit(b.y, b.y, ry),
                                       p0z = __builtin_amdgcn_alignbit(b.z, b.z, rz);
                        const uint32_t p1x = __builtin_amdgcn_alignbit(c.x, c.x, rx), p1y = __builtin_amdgcn_alignbit(c.y, c.y, ry),
                                       p1z = __builtin_amdgcn_alignbit(c.z, c.z, rz);
                        n0 = fmaxf(fmaxf(fmaf(lo16(p0x), ix, bx), fmaf(lo16(p0y), iy, by)), fmaxf(fmaf(lo16(p0z), iz, bz), kTMin));
                        f0 = fminf(fminf(fmaf(hi16(p0x), ix, cx), fmaf(hi16(p0y), iy, cy)), min_raw(fmaf(hi16(p0z), iz, cz), best.t));
                        n1 = fmaxf(fmaxf(fmaf(lo16(p1x), ix, bx), fmaf(lo16(p1y), iy, by)), fmaxf(fmaf(lo16(p1z), iz, bz), kTMin));
                        f1 = fminf(fminf(fmaf(hi16(p1x), ix, cx), fmaf(hi16(p1y), iy, cy)), min_raw(fmaf(hi16(p1z), iz, cz), best.t));
                    }
                    const bool h0 = n0 <= f0 * kFarSlack;
                    const bool h1 = n1 <= f1 * kFarSlack;
                    // next reference: the nearer hit child; the other one (if hit) is pushed
                    const bool take1 = h1 && (!h0 || n1 < n0);
                    int32_t nxt = take1 ? refs.y : refs.x;
                    if (h0 && h1) {
                        if (kGuard && sptr >= stack_end) { tl.guard++; }       // would overflow the lane's stack: the far child is dropped
                        else { *sptr = (uint32_t)(take1 ? refs.x : refs.y); sptr += kBlock; }
                    }
                    bool finished = false;
                    if (!(h0 || h1)) {
                        if (sptr == stack) { finished = true; }
                        else { sptr -= kBlock; nxt = (int32_t)*sptr; }
                    }
                    cur = nxt;
                    phase = (int)((uint32_t)nxt >> 31);          // PH_NODE (ref >= 0) or PH_LEAF (ref < 0)
                    if (finished) phase = PH_SHADE;
                }
                }
                pick = schedule();
            } while (pick == PH_NODE);
            if constexpr (kPrioNode != 0) __builtin_amdgcn_s_setprio(0);
        } else if (pick == PH_LEAF) {
            if constexpr (kPrioLeaf != 0) __builtin_amdgcn_s_setprio(kPrioLeaf);
            do {
#pragma unroll
                for (int rep = 0; rep < RT_LEAF_REPS; rep++) {
                if (WIDE) {
                    if (phase == PH_LEAF) {
                        // one hit primitive of the node's leaf children, nearest octant first
                        const uint32_t slot = g_lds_next_slot[flip256 | (tg_mask & 0xffu)];
                        tg_mask &= ~(1u << slot);
                        const uint32_t tri = tg_base + (uint32_t)__popc((tg_mask >> 8) & ((1u << slot) - 1u));
                        if (kGuard && (tri >= P.n_slots || ++guard_steps > kGuardMaxSteps)) { tl.guard++; phase = PH_SHADE; }
                        else {
                            if (COUNT) { tl.tris++; if (first_active_lane()) tl.w_leaf++; }
                            intersect_tri<SPH>(P.tris, tri, S.pos, S.dir, cull, cw, best);
                            wide_advance();
                        }
                    }
                } else
                if (kGuard && phase == PH_LEAF &&
                    ((uint64_t)((uint32_t)~cur >> 3) + ((uint32_t)~cur & 7u) > P.n_slots || ++guard_steps > kGuardMaxSteps)) { tl.guard++; phase = PH_SHADE; }
                if (!WIDE && phase == PH_LEAF) {
                    // ~cur = first_slot << 3 | triangles left: tested last to first, the count runs down in place
                    uint32_t enc = (uint32_t)~cur;
                    if (enc & 7u) {
                        if (COUNT) { tl.tris++; if (first_active_lane()) tl.w_leaf++; }
                        enc--;
                        intersect_tri<SPH>(P.tris, (enc >> 3) + (enc & 7u), S.pos, S.dir, cull, cw, best);
                    }
                    cur = (int32_t)~enc;
                    if ((enc & 7u) == 0u) {
                        if (sptr == stack) { phase = PH_SHADE; }
                        else {
                            sptr -= kBlock;
                            cur = (int32_t)*sptr;
                            phase = (int)((uint32_t)cur >> 31);
                        }
                    }
                }
                }
                pick = schedule();
            } while (pick == PH_LEAF);
            if constexpr (kPrioLeaf != 0) __builtin_amdgcn_s_setprio(0);
        } else if (pick == PH_SHADE) {
            // hit or miss shading (main.wgsl:46-69); a path that ends starts the pixel's next sample or moves on to FETCH
            do {
            if (RAYLIST) {
                if (phase == PH_SHADE) {
                    tl.rays++;
                    reinterpret_cast<uint2 *>(P.accum)[S.xf] = make_uint2(__float_as_uint(best.t), best.slot);
                    phase = PH_FETCH;
                }
            } else
            if (phase == PH_SHADE) {
                if (COUNT && first_active_lane()) tl.w_steps++;
                // a lane comes here with a traced ray to shade, or (bounce == kFreshPixel) straight from FETCH with a
                // new pixel whose first path starts here, where more lanes share the work than at the pixel boundary
                const bool fresh = S.bounce == kFreshPixel;
                bool more = true, next_sample = fresh;
#ifdef RT_EXP_SHADE_VALU     // sensitivity builds: N extra dependent VALU instructions per shade step
                { float dmy = ix;
#pragma unroll
                  for (int q = 0; q < RT_EXP_SHADE_VALU; q++) asm volatile("v_add_f32 %0, %0, %0" : "+v"(dmy));
                }
#endif
                if (!fresh) {
                    tl.rays++;
                    if (shade_core<AOV, SPH, LDSM, TEX>(P, S, best)) {
                        S.total = S.total + S.light;                // main.wgsl:26
                        S.sample++;
                        more = S.sample < P.samples;
                        next_sample = more;
                    }
                }
                if (next_sample) new_path(P, C, S);                 // next sample, main.wgsl:36-41
                if (more) start_trace();
                else phase = PH_FETCH;
            }
            pick = schedule();
            } while (pick == PH_SHADE);
        } else {
            // pixel boundary: write out finished pixel-frames (main.wgsl:27-32), then hand
            // pixel slots of the wave's chunk to the lanes
#if RT_STAGE_LINES > 0
            if (!RAYLIST && P.n_frames > 1u) {
                const bool writer = phase == PH_FETCH && have_pixel;
                unsigned long long m = __ballot(writer);
                if (m) {
                    // everything about the line is wave-uniform and stays in SGPRs; only the sum itself and the line on its way out use VGPRs
                    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                    do {
                        uint32_t ln = lane;
                        asm volatile("" : "+v"(ln));        // keeps what is derived from the lane number inside this loop (hoisted out of the kernel's main loop it costs the hot phases registers)
                        const int l = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const uint32_t xf_l = (uint32_t)__builtin_amdgcn_readlane((int)S.xf, l), row_l = (uint32_t)__builtin_amdgcn_readlane((int)S.local_row, l);
                        const uint32_t px = xf_l & 0xffffu;
                        const uint32_t idx_l = (xf_l >> 16) * (uint32_t)P.stage_stride + row_l * P.width + px;      // < 2^32: the host keeps the work items below that
                        // the 16 lines of an 8 x 8 tile go to 16 different slots (more slots: neighbouring tiles too)
                        const uint32_t slot = (((row_l & 7u) << 1) | ((px >> 2) & 1u) | ((px >> 3) << 4)) & (uint32_t)(RT_STAGE_LINES - 1);
                        const uint32_t g1 = (idx_l >> 2) + 1u;
                        uint32_t tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_line_tag[w][slot]);
                        uint32_t msk = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_line_mask[w][slot]);
                        if (tag != g1) {
                            // (a line leaves as 16 lanes x 4 bytes: one register of data per lane instead of four)
                            if (tag != 0u) { if (ln < 16u && ((msk >> (ln >> 2)) & 1u)) (reinterpret_cast<float *>(P.stage) + (size_t)(tag - 1u) * 16u)[ln] = (&s_lines[w][slot][0].x)[ln]; }
                            tag = g1; msk = 0u;
                        }
                        __builtin_amdgcn_wave_barrier();
                        if ((int)ln == l) { float *d = &s_lines[w][slot][idx_l & 3u].x; d[0] = S.total.x; d[1] = S.total.y; d[2] = S.total.z; d[3] = 0.0f; }
                        msk |= 1u << (idx_l & 3u);
                        __builtin_amdgcn_wave_barrier();
                        if (msk == 0xFu) {
                            if (ln < 16u) (reinterpret_cast<float *>(P.stage) + (size_t)(g1 - 1u) * 16u)[ln] = (&s_lines[w][slot][0].x)[ln];
                            tag = 0u; msk = 0u;
                        }
                        if (ln == 0u) { s_line_tag[w][slot] = tag; s_line_mask[w][slot] = msk; }
                        __builtin_amdgcn_wave_barrier();
                    } while (m);
                    if (writer) { tl.pixel_frames++; have_pixel = false; }
                }
            } else
#endif
            if (phase == PH_FETCH && have_pixel) {
                finish_pixel(P, S);
                tl.pixel_frames++;
                have_pixel = false;
            }
            const unsigned long long m_free = __ballot(phase == PH_FETCH);
            const int c_free = __popcll(m_free);
            if (RAYLIST) {
                // no shared queue: a list is worked off by rays, 135 times as many claims per unit of work as by pixel-frames, and
                // that many atomics on ONE address serialise (first version of the experiment: 30 ms for 8 M rays in any order).
                // Wave w takes chunks w, w + W, w + 2 W, ... of the list: the whole grid stays inside a window of W chunks.
                if (w_next == w_end && !drained) {
                    const uint32_t n_waves = gridDim.x * (kBlock / 64u), wave = blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6);
                    const uint64_t base = ((uint64_t)ng_claims * n_waves + wave) * P.chunk;
                    ng_claims++;
                    if (base >= total_slots) drained = true;
                    else { w_next = (uint32_t)base; w_end = (uint32_t)min(base + P.chunk, (uint64_t)total_slots); }
                }
            } else
            if (w_next == w_end && !drained) {
                uint32_t base = 0;
                // guided grain: 4 chunks per claim while plenty of work is left (coherent tiles), one chunk near
                // the end so that the last claims do not turn into a long tail
                uint32_t grain = P.chunk;
                if (lane == 0) {
                    const uint32_t seen = __hip_atomic_load(&P.queue->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen < total_slots && total_slots - seen > gridDim.x * (kBlock / 64u) * P.chunk * RT_GUIDED_FACTOR) grain = P.chunk * RT_GUIDED_GRAIN;
                    base = atomicAdd(&P.queue->head, grain);
                }
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                grain = (uint32_t)__builtin_amdgcn_readfirstlane((int)grain);
                if (base >= total_slots) {
                    drained = true;
#ifdef RT_TAIL_PROBE
                    t_drain = wall_clock64();
#endif
                } else { w_next = base; w_end = min(base + grain, total_slots); }
            }
            if (drained) {
                if (phase == PH_FETCH) phase = PH_DONE;
            } else {
                const uint32_t take = min(w_end - w_next, (uint32_t)c_free);
                const uint32_t rank = (uint32_t)__popcll(m_free & ((1ull << lane) - 1ull));
                if (RAYLIST) {
                    if (phase == PH_FETCH && rank < take) {
                        const uint32_t q = w_next + rank;
                        const uint32_t r = P.aov ? P.aov[q] : q;
                        const float4 ro = P.stage[2u * (size_t)r], rd = P.stage[2u * (size_t)r + 1u];
                        S.pos = v3(ro.x, ro.y, ro.z);
                        S.dir = v3(rd.x, rd.y, rd.z);
                        S.xf = r;
                        start_trace();
                    }
                } else
                if (phase == PH_FETCH && rank < take) {
                    uint32_t q = w_next + rank, f = 0;
                    if (P.n_frames > 1u) { f = q / frame_slots; q -= f * frame_slots; }
                    uint32_t x, y, lr;
                    if (slot_to_pixel(P, q, x, y, lr)) {
                        begin_pixel(P, C, S, x, y, lr, f);
                        have_pixel = true;
                        if (P.samples == 0) { S.total = v3(0.0f, 0.0f, 0.0f); }   // stays in FETCH: written out next round
                        else { S.bounce = kFreshPixel; phase = PH_SHADE; }          // first path starts in SHADE
                    }
                }
                w_next += take;
            }
            pick = schedule();
        }
    }
#ifdef RT_TAIL_PROBE
    if (lane == 0 && P.probe) {
        unsigned long long *pr = P.probe + 3u * (blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6));
        pr[0] = t_start; pr[1] = t_drain; pr[2] = wall_clock64();
    }
#endif
#if RT_STAGE_LINES > 0
    if (!RAYLIST && P.n_frames > 1u && lane < (uint32_t)RT_STAGE_LINES) {        // what the slots still hold
        const uint32_t w = threadIdx.x >> 6, tag = s_line_tag[w][lane], msk = s_line_mask[w][lane];
        if (tag != 0u)
            for (uint32_t e = 0; e < 4u; e++)
                if ((msk >> e) & 1u) P.stage[(size_t)(tag - 1u) * 4u + e] = s_lines[w][lane][e];
    }
#endif
    if (!RAYLIST || COUNT) flush_tally<COUNT>(P, tl);
}

// Second half of a multi-frame launch: out = mix(old, tone_map(sum_f / samples * exposure), 1 / (accumulation_frame + f + 1)) for
// f = 0 .. n_frames-1 in order (main.wgsl:30-32), one thread per pixel, coalesced.
__global__ __launch_bounds__(256) void rt_blend_frames(const RenderParams P, size_t n_pixels) {
    const size_t pix = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (pix >= n_pixels) return;
    float4 *dst = reinterpret_cast<float4 *>(P.accum) + pix;
    const float4 old = *dst;
    rt_v3 acc = v3(old.x, old.y, old.z);
    for (uint32_t f = 0; f < P.n_frames; f++) {
        const float4 m = P.stage[(size_t)f * P.stage_stride + pix];
        const float w = 1.0f / (float)(P.accumulation_frame + f + 1u);
        acc = rt_mix(acc, resolve_pixel(P, v3(m.x, m.y, m.z)), w);
    }
    *dst = make_float4(acc.x, acc.y, acc.z, old.w);
}

#endif  // RT_TU_PHASED || RT_TU_DEVICE_ONLY

#ifdef RT_TU_SIMPLE
// ---- variant 1: one wave per 8x8 tile (the reference's mapping) ----
template <bool COUNT, bool AOV, bool SPH>
__global__ __launch_bounds__(kBlock) void rt_render_tiles(const RenderParams P) {
    extern __shared__ uint32_t lds_stack[];
    uint32_t *stack = lds_stack + threadIdx.x;
    if (P.node_wide) fill_next_slot_table();
    const uint32_t tile = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t tiles_total = P.tiles_x * P.local_bands;
    const Camera C = load_camera(P);
    Tally tl;
    uint32_t x, y, lr;
    if (tile < tiles_total && slot_to_pixel(P, tile * 64u + (threadIdx.x & 63u), x, y, lr)) {
        PixelState S;
        begin_pixel(P, C, S, x, y, lr);
        new_path(P, C, S);
        if (P.samples > 0) {
            for (;;) {
                if (COUNT && first_active_lane()) tl.w_steps++;
                if (step<COUNT, AOV, SPH>(P, C, stack, S, tl)) break;
            }
        } else {
            S.total = v3(0.0f, 0.0f, 0.0f);
        }
        finish_pixel(P, S);
        tl.pixel_frames = 1;
    }
    flush_tally<COUNT>(P, tl);
}

#endif  // RT_TU_SIMPLE

template <typename K>
hipError_t launch(K kernel, dim3 grid, size_t lds, hipStream_t stream, const RenderParams &P) {
    hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, stream, P);
    return hipGetLastError();
}

}  // namespace

#ifdef RT_TU_PHASED
// Persistent grid size: the number of 256-thread workgroups of the given kernel
// instance that are resident at once (register/LDS limited), so every wave
// lives for the whole launch.
// One value of a run-time flag as a compile-time constant: f(std::integral_constant<...>)
template <class F> static auto with_bool(bool b, F f) { return b ? f(std::true_type{}) : f(std::false_type{}); }
template <class F> static auto with_int3(int v, F f) {          // 0..2
    return v == 2 ? f(std::integral_constant<int, 2>{}) : (v == 1 ? f(std::integral_constant<int, 1>{}) : f(std::integral_constant<int, 0>{}));
}
template <class F> static auto with_int4(int v, F f) {          // 0..3
    return v == 3 ? f(std::integral_constant<int, 3>{}) : (v == 2 ? f(std::integral_constant<int, 2>{}) : (v == 1 ? f(std::integral_constant<int, 1>{}) : f(std::integral_constant<int, 0>{})));
}
// the phased kernel's instance for (count, aov, ext, materials in LDS, node format) handed to f as a kernel pointer
template <class F> static auto with_phased_instance(bool count, bool aov, int ext, bool ldsm, int fmt, F f) {
    return with_bool(count, [&](auto C) { return with_bool(aov, [&](auto A) { return with_int3(ext, [&](auto E) { return with_bool(ldsm, [&](auto L) {
        return with_int4(fmt, [&](auto M) { return f(rt_render_phased<decltype(C)::value, decltype(A)::value, decltype(E)::value, decltype(L)::value, decltype(M)::value>); }); }); }); }); });
}

// ext: 0 = triangles, untextured; 1 = + spheres; 2 = + textures.  fmt: 0 = 48-byte binary nodes, 1 = 32-byte binary nodes, 2 = 80-byte 8-ary nodes, 3 = 64-byte 8-ary nodes
int rt_persistent_blocks(int device, uint32_t stack_entries, bool count, bool aov, int ext, bool lds_materials, int fmt) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 256 * 4;
    int per_cu = 0;
    const size_t lds = fmt >= 2 ? (size_t)(stack_entries / 2u) * kBlock * 6u : (size_t)stack_entries * kBlock * sizeof(uint32_t);
    // the instances differ in register count (and static LDS), hence in how many workgroups fit
    const void *fn = with_phased_instance(count, aov, ext, lds_materials, fmt, [](auto kernel) { return (const void *)kernel; });
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kBlock, lds) != hipSuccess || per_cu < 1) per_cu = 4;
    return per_cu * prop.multiProcessorCount;
}

// variant 2 (and the blend of a multi-frame launch).  Lives in its own translation unit: see the Makefile on
// -structurizecfg-skip-uniform-regions.
// dynamic LDS of the phased kernel: binary nodes one word per stack entry; 8-ary nodes 6 bytes per level (see rt_render_phased)
static size_t phased_lds_bytes(uint32_t stack_entries, bool wide) {
    return wide ? (size_t)(stack_entries / 2u) * kBlock * 6u : (size_t)stack_entries * kBlock * sizeof(uint32_t);
}

hipError_t rt_launch_phased(const RenderParams &P, bool count, bool aov, int ext, unsigned blocks, size_t lds, hipStream_t stream) {
    const dim3 grid(blocks);
    lds = phased_lds_bytes(P.stack_entries, P.node_wide != 0u);
    // LDSM: the workgroup keeps the scene's material records in LDS (rt_device.h RT_LDS_MATERIALS); FMT: the node format
    const bool ldsm = P.n_materials <= RT_LDS_MATERIALS;
    const int fmt = P.node_wide ? (int)P.node_wide + 1 : (P.node_q8 ? 1 : 0);
    const hipError_t e = with_phased_instance(count, aov, ext, ldsm, fmt, [&](auto kernel) { return launch(kernel, grid, lds, stream, P); });
    if (e != hipSuccess || P.n_frames <= 1u) return e;
    const size_t n_pixels = (size_t)P.local_rows * P.width;
    if (n_pixels) rt_blend_frames<<<dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream>>>(P, n_pixels);
    return hipGetLastError();
}
#endif  // RT_TU_PHASED

#ifdef RT_TU_SIMPLE
hipError_t rt_launch_phased(const RenderParams &P, bool count, bool aov, int ext, unsigned blocks, size_t lds, hipStream_t stream);

// variants 0 and 1 (the A/B baselines)
template <bool COUNT, bool AOV, bool SPH>
static hipError_t launch_simple(const RenderParams &P, int variant, dim3 tile_grid, dim3 persistent_grid, size_t lds, hipStream_t stream) {
    if (variant == 1) return launch(rt_render_tiles<COUNT, AOV, SPH>, tile_grid, lds, stream, P);
    return launch(rt_render_persistent<COUNT, AOV, SPH>, persistent_grid, lds, stream, P);
}

// Host-side launcher.  Dynamic LDS = stack_entries x 256 lanes x 4 B.
hipError_t rt_launch_render(const RenderParams &P, int variant, bool count, bool aov, int ext, int persistent_blocks, hipStream_t stream) {
    const bool spheres = ext != 0;      // variants 0 / 1: one instance for spheres and textures
    const uint32_t tiles = P.tiles_x * P.local_bands;
    if (tiles == 0) return hipSuccess;
    const size_t lds = (size_t)P.stack_entries * kBlock * sizeof(uint32_t);
    dim3 tile_grid((tiles + (kBlock / 64) - 1) / (kBlock / 64));
    // persistent grids: no more waves than chunks of work in the WHOLE launch (all its frames), counted in 64 bits
    const uint64_t chunks = ((uint64_t)tiles * 64u * (P.n_frames ? P.n_frames : 1u) + P.chunk - 1) / P.chunk;
    uint32_t blocks = (uint32_t)persistent_blocks;
    if ((uint64_t)blocks > (chunks + 3) / 4) blocks = (uint32_t)((chunks + 3) / 4);
    if (blocks == 0) blocks = 1;
    if (variant == 2) return rt_launch_phased(P, count, aov, ext, blocks, lds, stream);
    dim3 pgrid(blocks);
    const int sel = (count ? 4 : 0) | (aov ? 2 : 0) | (spheres ? 1 : 0);
    switch (sel) {
        case 0: return launch_simple<false, false, false>(P, variant, tile_grid, pgrid, lds, stream);
        case 1: return launch_simple<false, false, true>(P, variant, tile_grid, pgrid, lds, stream);
        case 2: return launch_simple<false, true, false>(P, variant, tile_grid, pgrid, lds, stream);
        case 3: return launch_simple<false, true, true>(P, variant, tile_grid, pgrid, lds, stream);
        case 4: return launch_simple<true, false, false>(P, variant, tile_grid, pgrid, lds, stream);
        case 5: return launch_simple<true, false, true>(P, variant, tile_grid, pgrid, lds, stream);
        case 6: return launch_simple<true, true, false>(P, variant, tile_grid, pgrid, lds, stream);
        default: return launch_simple<true, true, true>(P, variant, tile_grid, pgrid, lds, stream);
    }
}
#endif  // RT_TU_SIMPLE
