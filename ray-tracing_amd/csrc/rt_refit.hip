// rt_refit.hip — BVH refit on the device: ≙ acceleration_structure.update(), /root/reference/src/app.rs:61,
// after App::upload_models moved the geometries (src/app.rs:46-62).
//
// The tree keeps its topology.  Kernel 1 re-derives every triangle record from the untransformed corner
// positions the shading records already hold (DevShade p0..p2) and the geometry's 3x4 transform, with the host's
// expression (rt_pack.h rt_transform_point: same bits as a host rebuild), and leaves each primitive's box.
// Kernel 2 starts one thread per node that has a leaf child and walks towards the root; at every node the
// second arrival (atomic counter) packs both child boxes into the node (rt_pack.h, the host's packing rules)
// and carries their union up.  RT_OPT_REFIT = 2 selects this path; 1 is the host refit it is tested against.
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_math.h"
#include "rt_pack.h"

namespace {

__device__ __forceinline__ void grow(RtRefitBox &b, const RtRefitBox &o) {
    for (int k = 0; k < 3; k++) { b.lo[k] = fminf(b.lo[k], o.lo[k]); b.hi[k] = fmaxf(b.hi[k], o.hi[k]); }
}

__global__ __launch_bounds__(256) void rt_refit_tris(DevTri *__restrict__ tris, const DevShade *__restrict__ shade,
                                                     const float *__restrict__ xform, RtRefitBox *__restrict__ tribox, uint64_t n_slots, float prim_pad) {
    const uint64_t slot = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (slot >= n_slots) return;
    DevTri t = tris[slot];
    RtRefitBox b;
    if (t.is_sphere) {
        // spheres do not move with a geometry; the box is the builder's (rt_api.hip build_and_upload)
        const float r = t.e1x * 1.0001f + 1e-6f;
        const float c[3] = {t.ax, t.ay, t.az};
        const bool finite = rt_all_finite(c, 3) && rt_all_finite(&r, 1);
        for (int k = 0; k < 3; k++) { b.lo[k] = finite ? c[k] - r : RT_EMPTY_BOX; b.hi[k] = finite ? c[k] + r : -RT_EMPTY_BOX; }
    } else {
        const DevShade &s = shade[slot];
        float c[9];
        if (xform) {
            const float *m = xform + 12u * (size_t)s.geometry;
            rt_transform_point(m, s.p0, c);
            rt_transform_point(m, s.p1, c + 3);
            rt_transform_point(m, s.p2, c + 6);
        } else {
            for (int k = 0; k < 3; k++) { c[k] = s.p0[k]; c[3 + k] = s.p1[k]; c[6 + k] = s.p2[k]; }
        }
        t.ax = c[0]; t.ay = c[1]; t.az = c[2];
        t.e1x = c[3] - c[0]; t.e1y = c[4] - c[1]; t.e1z = c[5] - c[2];
        t.e2x = c[6] - c[0]; t.e2y = c[7] - c[1]; t.e2z = c[8] - c[2];
        tris[slot] = t;
        const bool finite = rt_all_finite(c, 9);            // a non-finite corner: the empty box, as the host build gives it (rt_pack.h)
        for (int k = 0; k < 3; k++) {
            b.lo[k] = finite ? fminf(c[k], fminf(c[3 + k], c[6 + k])) : RT_EMPTY_BOX;
            b.hi[k] = finite ? fmaxf(c[k], fmaxf(c[3 + k], c[6 + k])) : -RT_EMPTY_BOX;
        }
    }
    for (int k = 0; k < 3; k++) { b.lo[k] -= prim_pad; b.hi[k] += prim_pad; }      // as the host build widens them (rt_pack.h rt_prim_pad)
    tribox[slot] = b;
}

__global__ __launch_bounds__(256) void rt_refit_nodes(float4 *__restrict__ chunks, uint32_t stride, const int32_t *__restrict__ parent,
                                                      unsigned int *__restrict__ arrived, RtRefitBox *cbox, const RtRefitBox *__restrict__ tribox,
                                                      uint32_t n_nodes, uint64_t n_slots, RtRefitBox *root_out, uint32_t q8) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_nodes) return;
    const DevNode *nd = reinterpret_cast<const DevNode *>(chunks + (size_t)i * stride);
    const int32_t refs[2] = {nd->ref0, nd->ref1};      // q8 records (stride 2) are the first 32 bytes of a DevNode
    uint32_t arrivals = 0;
    for (int child = 0; child < 2; child++) {
        if (refs[child] >= 0) continue;
        const uint32_t enc = (uint32_t)~refs[child];
        const uint64_t first = enc >> 3;
        const uint32_t count = enc & 7u;
        RtRefitBox b;
        for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
        for (uint32_t k = 0; k < count && first + k < n_slots; k++) grow(b, tribox[first + k]);
        cbox[2u * (size_t)i + child] = b;
        arrivals++;
    }
    if (arrivals == 0) return;
    uint32_t cur = i;
    for (;;) {
        if (arrivals < 2u) {
            __threadfence();                                        // my box is visible before I count as arrived
            if (atomicAdd(&arrived[cur], 1u) == 0u) return;         // the sibling's side finishes this node
            __threadfence();                                        // and its box is visible to me
        }
        const RtRefitBox b0 = cbox[2u * (size_t)cur], b1 = cbox[2u * (size_t)cur + 1u];
        DevNode *cn = reinterpret_cast<DevNode *>(chunks + (size_t)cur * stride);
        if (q8) rt_pack_children_q8(*cn, b0.lo, b0.hi, b1.lo, b1.hi);   // writes inside the first 32 bytes only
        else rt_pack_children(*cn, b0.lo, b0.hi, b1.lo, b1.hi);
        RtRefitBox u = b0;
        grow(u, b1);
        const int32_t p = parent[cur];
        if (p < 0) { *root_out = u; return; }
        cbox[(size_t)p] = u;                                        // p = parent node * 2 + child slot
        cur = (uint32_t)p >> 1;
        arrivals = 1u;
    }
}

// The same walk for the 8-ary tree (DevNode8).  One thread per node; a thread whose node has no inner children packs it from its
// primitives' boxes and carries the union to the parent; at a parent the LAST of its inner children to arrive (atomic counter
// against popcount(imask)) packs it — leaf children's boxes from the primitives again, inner children's from cbox — and goes on up.
struct RtGrid6 { float lo[3], cell[3]; };
// COMPACT: 64-byte records (DevNode8c) whose corners are cells of the grid `G`
template <bool COMPACT>
__global__ __launch_bounds__(256) void rt_refit_nodes8(float4 *__restrict__ chunks, uint32_t stride, const int32_t *__restrict__ parent,
                                                       unsigned int *__restrict__ arrived, RtRefitBox *cbox, const RtRefitBox *__restrict__ tribox,
                                                       uint32_t n_nodes, uint64_t n_slots, RtRefitBox *root_out, RtGrid6 G) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_nodes) return;
    auto imask_of = [&](uint32_t node) -> uint32_t {
        if (COMPACT) return rt_node8c_imask(*reinterpret_cast<const DevNode8c *>(chunks + (size_t)node * stride));
        return reinterpret_cast<const DevNode8 *>(chunks + (size_t)node * stride)->exps_imask >> 24;
    };
    if (imask_of(i) != 0u) return;                            // has inner children: finished by the last of them
    uint32_t cur = i;
    for (;;) {
        DevNode8 *nd = reinterpret_cast<DevNode8 *>(chunks + (size_t)cur * stride);
        DevNode8c *nc = reinterpret_cast<DevNode8c *>(chunks + (size_t)cur * stride);
        const uint32_t imask = imask_of(cur), lmask = COMPACT ? rt_node8c_lmask(*nc) : (nd->lmask & 0xffu);
        const uint64_t tri_base = COMPACT ? rt_node8c_tri_base(*nc) : nd->tri_base;
        float blo[8][3], bhi[8][3];
        RtRefitBox u;
        for (int k = 0; k < 3; k++) { u.lo[k] = INFINITY; u.hi[k] = -INFINITY; }
        uint32_t leaf_rank = 0;
        for (uint32_t sl = 0; sl < 8u; sl++) {
            RtRefitBox b;
            if ((lmask >> sl) & 1u) {
                const uint64_t slot = tri_base + leaf_rank++;
                if (slot < n_slots) b = tribox[slot];
                else { for (int k = 0; k < 3; k++) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } }
            } else if ((imask >> sl) & 1u) {
                b = cbox[8u * (size_t)cur + sl];
            } else continue;
            for (int k = 0; k < 3; k++) { blo[sl][k] = b.lo[k]; bhi[sl][k] = b.hi[k]; }
            grow(u, b);
        }
        if (COMPACT) rt_pack_node8c(*nc, blo, bhi, imask | lmask, G.lo, G.cell);
        else rt_pack_node8(*nd, blo, bhi, imask | lmask);
        const int32_t p = parent[cur];
        if (p < 0) { *root_out = u; return; }
        cbox[(size_t)p] = u;                                        // p = parent node * 8 + child slot
        const uint32_t pn = (uint32_t)p >> 3;
        const uint32_t expected = (uint32_t)__popc(imask_of(pn));
        __threadfence();                                            // my box is visible before I count as arrived
        if (atomicAdd(&arrived[pn], 1u) + 1u != expected) return;   // a sibling still to come finishes the parent
        __threadfence();                                            // and the siblings' boxes are visible to me
        cur = pn;
    }
}

}  // namespace

hipError_t rt_launch_refit8(DevTri *tris, const DevShade *shade, const float *xform, RtRefitBox *tribox, uint64_t n_slots, float4 *chunks,
                            uint32_t stride, const int32_t *parent, unsigned int *arrived, RtRefitBox *cbox, uint32_t n_nodes,
                            RtRefitBox *root_out, float prim_pad, const float *grid_lo_cell, hipStream_t stream) {
    if (n_slots == 0 || n_nodes == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(arrived, 0, (size_t)n_nodes * sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    rt_refit_tris<<<dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, stream>>>(tris, shade, xform, tribox, n_slots, prim_pad);
    RtGrid6 G = {{0, 0, 0}, {1, 1, 1}};
    if (grid_lo_cell) {
        for (int k = 0; k < 3; k++) { G.lo[k] = grid_lo_cell[k]; G.cell[k] = grid_lo_cell[3 + k]; }
        rt_refit_nodes8<true><<<dim3((n_nodes + 255u) / 256u), dim3(256), 0, stream>>>(chunks, stride, parent, arrived, cbox, tribox, n_nodes, n_slots, root_out, G);
    } else {
        rt_refit_nodes8<false><<<dim3((n_nodes + 255u) / 256u), dim3(256), 0, stream>>>(chunks, stride, parent, arrived, cbox, tribox, n_nodes, n_slots, root_out, G);
    }
    return hipGetLastError();
}

hipError_t rt_launch_refit(DevTri *tris, const DevShade *shade, const float *xform, RtRefitBox *tribox, uint64_t n_slots, float4 *chunks,
                           uint32_t stride, const int32_t *parent, unsigned int *arrived, RtRefitBox *cbox, uint32_t n_nodes,
                           RtRefitBox *root_out, bool q8, float prim_pad, hipStream_t stream) {
    if (n_slots == 0 || n_nodes == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(arrived, 0, (size_t)n_nodes * sizeof(unsigned int), stream);
    if (e != hipSuccess) return e;
    rt_refit_tris<<<dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, stream>>>(tris, shade, xform, tribox, n_slots, prim_pad);
    rt_refit_nodes<<<dim3((n_nodes + 255u) / 256u), dim3(256), 0, stream>>>(chunks, stride, parent, arrived, cbox, tribox, n_nodes, n_slots, root_out, q8 ? 1u : 0u);
    return hipGetLastError();
}
