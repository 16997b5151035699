// rt_pack.h — how child boxes become a DevNode, shared by the host repack (rt_api.hip) and the device refit
// (rt_refit.hip) so that both write nodes by the same rules.  Outside the arithmetic contract: the only
// requirement is that a packed box contains the box it was made from.
#pragma once
#include <math.h>
#include <stdint.h>

#include "rt_device.h"
#include "rt_math.h"

// Child boxes are widened by a few ulp so that a hit the triangle test accepts a hair outside its triangle
// is not culled by the box (tie policy).
RT_HD float rt_pad_lo(float v) { return v - (fabsf(v) * 4.76837158e-7f + 1e-30f); }
RT_HD float rt_pad_hi(float v) { return v + (fabsf(v) * 4.76837158e-7f + 1e-30f); }

// Absolute widening of every PRIMITIVE's box, from the scene's largest |coordinate| (see rt_api.hip host_build)
#ifndef RT_PRIM_PAD_SCALE
#define RT_PRIM_PAD_SCALE 7.62939453125e-6f   /* 2^-17 */
#endif
RT_HD float rt_prim_pad(float scene_max_abs) { return scene_max_abs <= 3.0e38f ? scene_max_abs * RT_PRIM_PAD_SCALE : 0.0f; }

// A primitive with a non-finite corner (NaN or +-inf after the geometry transform) can never be hit: every comparison of the
// contract's triangle / sphere test fails on the NaNs it produces (the oracle's brute force agrees, tests/checkers/fuzz_parity.py).
// Its box must not reach the tree — one infinite plane poisons the corner of every ancestor and culls the finite geometry
// beside it — so such a primitive gets this EMPTY box: inverted and finite, ignored by every min / max union and packed
// as an inverted child box (lo offset saturated, hi offset 0) that no ray enters.
#define RT_EMPTY_BOX 3.0e38f
RT_HD bool rt_all_finite(const float *v, int n) {
    for (int i = 0; i < n; i++)
        if (!(v[i] >= -3.0e38f && v[i] <= 3.0e38f)) return false;
    return true;
}

// x >= 0 → the f16 bit pattern nearest to x from below (up = false) or above (up = true).
// Above saturates to +inf, below to 65504; a non-zero upper bound is never an f16
// denormal, so a flush-to-zero mode cannot shrink a box.
RT_HD uint32_t rt_f16_directed(double x, bool up) {
    if (!(x > 0.0)) return 0u;
    int e;
    frexp(x, &e);                         // x = m * 2^e, m in [0.5, 1)
    const int E = e + 14;                 // biased f16 exponent of a normal number
    uint32_t bits;
    if (E >= 31) bits = up ? 0x7c00u : 0x7bffu;
    else if (E <= 0) {                    // denormal range: units of 2^-24
        const double q = ldexp(x, 24);
        bits = (uint32_t)(up ? ceil(q) : floor(q));
    } else {
        const double q = ldexp(x, 25 - E);            // [1024, 2048)
        bits = ((uint32_t)(E - 1) << 10) + (uint32_t)(up ? ceil(q) : floor(q));
    }
    if (up && bits < 0x0400u) bits = 0x0400u;
    if (bits > 0x7c00u) bits = 0x7c00u;
    return bits;
}

// Both children of one node: unpadded boxes (lo/hi per child) → lower corner + outward-rounded f16 offsets.
// References are left alone.
RT_HD void rt_pack_children(DevNode &n, const float *blo0, const float *bhi0, const float *blo1, const float *bhi1) {
    float lo0[3], hi0[3], lo1[3], hi1[3], lo[3];
    for (int k = 0; k < 3; k++) {
        lo0[k] = rt_pad_lo(blo0[k]); hi0[k] = rt_pad_hi(bhi0[k]);
        lo1[k] = rt_pad_lo(blo1[k]); hi1[k] = rt_pad_hi(bhi1[k]);
        lo[k] = lo0[k] < lo1[k] ? lo0[k] : lo1[k];
    }
    uint32_t c0[3], c1[3];
    for (int k = 0; k < 3; k++) {
        c0[k] = (rt_f16_directed((double)hi0[k] - (double)lo[k], true) << 16) | rt_f16_directed((double)lo0[k] - (double)lo[k], false);
        c1[k] = (rt_f16_directed((double)hi1[k] - (double)lo[k], true) << 16) | rt_f16_directed((double)lo1[k] - (double)lo[k], false);
    }
    n.lox = lo[0]; n.loy = lo[1]; n.loz = lo[2];
    n.c0x = c0[0]; n.c0y = c0[1]; n.c0z = c0[2];
    n.c1x = c1[0]; n.c1y = c1[1]; n.c1z = c1[2];
}

// Two-load node ("q8", RT_OPT_NODE_FORMAT): a 32-byte record = the first 32 bytes of a DevNode: the lower corner,
// per axis one dword of four 8-bit plane offsets [c0.lo, c1.lo, c0.hi, c1.hi] in units of scale = 2^(e - 20), and the
// two references.  The 5-bit e rides in the lowest mantissa bits of the corner (lox: e & 7, loy: e >> 3); the corner
// is moved DOWN to the next value with those bits, so the kernel uses the stuffed floats as they are.  Offsets are
// rounded outward.  Node extents up to 255 * 2^11: the host falls back to 48-byte nodes for scenes beyond
// RT_Q8_MAX_EXTENT.  A third less node memory: 8 % faster where node fetches miss the caches (config C4), 2 % slower
// where they do not (C3: the decode costs ~20 VALU instructions more per visit), hence chosen by size.
#define RT_Q8_EXP_BIAS 20
#define RT_Q8_MAX_EXTENT 262144.0f
// largest float <= v whose low `nbits` mantissa bits are `bits` (v finite)
RT_HD float rt_q8_stuff(float v, uint32_t bits, uint32_t nbits) {
    const uint32_t mask = (1u << nbits) - 1u;
    union { float f; uint32_t u; } x, y;
    x.f = v;
    if (!(v == v) || (x.u & 0x7f800000u) == 0x7f800000u) return v;       // NaN / inf: an empty box, never entered
    y.u = (x.u & ~mask) | bits;
    if (y.f > v) {
        // step to the previous block of 2^nbits bit patterns: towards -inf
        if ((x.u & 0x7fffffffu) < (1u << nbits) && !(x.u >> 31)) { y.u = 0x80000000u | (1u << nbits) | bits; }   // tiny positive: below zero
        else if (x.u >> 31) y.u += 1u << nbits;     // negative: larger magnitude
        else y.u -= 1u << nbits;                    // positive: smaller magnitude
    }
    return y.f;
}
RT_HD uint32_t rt_q8_exponent(float lox, float loy) {
    union { float f; uint32_t u; } x, y;
    x.f = lox; y.f = loy;
    return (x.u & 7u) | ((y.u & 3u) << 3);
}
// Can every node of a tree whose root box is [lo, hi] be written as a q8 record?  A node's offsets span its padded extent
// plus what stuffing may move its corner down (2^-19 of the corner's magnitude), in units of at most 2^11: no node of the
// tree needs more than the root's padded extent plus 2^-19 of the root's largest |coordinate|.
RT_HD bool rt_q8_fits(const float *lo, const float *hi) {
    for (int k = 0; k < 3; k++) {
        const double l = rt_pad_lo(lo[k]), h = rt_pad_hi(hi[k]);
        const double m = fabs(l) > fabs(h) ? fabs(l) : fabs(h);
        const double d = (h - l) + m * 1.9073486328125e-6 + 1e-37;
        if (!(d <= 255.0 * 2048.0) || !(d <= (double)RT_Q8_MAX_EXTENT)) return false;
    }
    return true;
}
RT_HD void rt_pack_children_q8(DevNode &n, const float *blo0, const float *bhi0, const float *blo1, const float *bhi1) {
    float lo0[3], hi0[3], lo1[3], hi1[3], lo[3];
    double ext = 0.0;
    for (int k = 0; k < 3; k++) {
        lo0[k] = rt_pad_lo(blo0[k]); hi0[k] = rt_pad_hi(bhi0[k]);
        lo1[k] = rt_pad_lo(blo1[k]); hi1[k] = rt_pad_hi(bhi1[k]);
        lo[k] = lo0[k] < lo1[k] ? lo0[k] : lo1[k];
        const float hi = hi0[k] > hi1[k] ? hi0[k] : hi1[k];
        // the corner may still move down by up to 16 bit patterns (stuffing): 2^-19 of its magnitude, plus the denormal step
        const double d = (double)hi - (double)lo[k] + fabs((double)lo[k]) * 1.9073486328125e-6 + 1e-37;
        if (d > ext) ext = d;
    }
    int e = 0;
    while (e < 31 && 255.0 * ldexp(1.0, e - RT_Q8_EXP_BIAS) < ext) e++;
    lo[0] = rt_q8_stuff(lo[0], (uint32_t)e & 7u, 3);
    lo[1] = rt_q8_stuff(lo[1], (uint32_t)e >> 3, 2);
    const double inv_scale = ldexp(1.0, RT_Q8_EXP_BIAS - e);
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const double l0 = floor(((double)lo0[k] - (double)lo[k]) * inv_scale), h0 = ceil(((double)hi0[k] - (double)lo[k]) * inv_scale);
        const double l1 = floor(((double)lo1[k] - (double)lo[k]) * inv_scale), h1 = ceil(((double)hi1[k] - (double)lo[k]) * inv_scale);
        // empty children (lo = +inf, hi = -inf) and NaNs end as lo 255 / hi 0: never entered
        const uint32_t ql0 = l0 >= 0.0 && l0 <= 255.0 ? (uint32_t)l0 : (l0 < 0.0 ? 0u : 255u), qh0 = h0 >= 0.0 && h0 <= 255.0 ? (uint32_t)h0 : (h0 > 255.0 ? 255u : 0u);
        const uint32_t ql1 = l1 >= 0.0 && l1 <= 255.0 ? (uint32_t)l1 : (l1 < 0.0 ? 0u : 255u), qh1 = h1 >= 0.0 && h1 <= 255.0 ? (uint32_t)h1 : (h1 > 255.0 ? 255u : 0u);
        q[k] = ql0 | (ql1 << 8) | (qh0 << 16) | (qh1 << 24);
    }
    n.lox = lo[0]; n.loy = lo[1]; n.loz = lo[2];
    n.c0x = q[0]; n.c0y = q[1]; n.c0z = q[2];
}

// One 8-ary node (DevNode8): child boxes (unpadded; absent children have lo > hi) -> lower corner, per-axis scale exponents and
// outward-rounded 8-bit plane offsets.  The masks and bases are left alone.  Needs rt_q8_fits(root box) for the scale range.
RT_HD void rt_pack_node8(DevNode8 &n, const float (*blo)[3], const float (*bhi)[3], uint32_t present_mask) {
    float lo[3] = {0.0f, 0.0f, 0.0f}, hi[3] = {0.0f, 0.0f, 0.0f};
    bool any = false;
    for (int c = 0; c < 8; c++) {
        if (!((present_mask >> c) & 1u)) continue;
        for (int k = 0; k < 3; k++) {
            const float l = rt_pad_lo(blo[c][k]), h = rt_pad_hi(bhi[c][k]);
            if (!any || l < lo[k]) lo[k] = l;
            if (!any || h > hi[k]) hi[k] = h;
        }
        any = true;
    }
    uint32_t eb[3];
    double inv_scale[3];
    for (int k = 0; k < 3; k++) {
        const double ext = (double)hi[k] - (double)lo[k];
        int e = -126;
        while (e < 127 && 255.0 * ldexp(1.0, e) < ext) e++;
        eb[k] = (uint32_t)(e + 127);
        inv_scale[k] = ldexp(1.0, -e);
    }
    uint32_t q[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int c = 0; c < 8; c++) {
        const bool present = (present_mask >> c) & 1u;
        for (int k = 0; k < 3; k++) {
            uint32_t ql = 255u, qh = 0u;                       // absent: an inverted box
            if (present) {
                const double l = floor(((double)rt_pad_lo(blo[c][k]) - (double)lo[k]) * inv_scale[k]);
                const double h = ceil(((double)rt_pad_hi(bhi[c][k]) - (double)lo[k]) * inv_scale[k]);
                ql = l >= 0.0 && l <= 255.0 ? (uint32_t)l : (l < 0.0 ? 0u : 255u);
                qh = h >= 0.0 && h <= 255.0 ? (uint32_t)h : (h > 255.0 ? 255u : 0u);
            }
            q[k][c >> 1] |= (ql << (8 * (c & 1))) | (qh << (16 + 8 * (c & 1)));
        }
    }
    n.lox = lo[0]; n.loy = lo[1]; n.loz = lo[2];
    n.exps_imask = (n.exps_imask & 0xFF000000u) | eb[0] | (eb[1] << 8) | (eb[2] << 16);
    for (int p = 0; p < 4; p++) { n.qx[p] = q[0][p]; n.qy[p] = q[1][p]; n.qz[p] = q[2][p]; }
}

// ---- compact 8-ary node (DevNode8c) ----
RT_HD uint32_t rt_node8c_imask(const DevNode8c &n) { return (n.w1 >> 22) & 0xffu; }
RT_HD uint32_t rt_node8c_lmask(const DevNode8c &n) { return (n.w2 >> 26) | ((n.w3 >> 28) << 6 & 0xc0u); }
RT_HD uint32_t rt_node8c_child_base(const DevNode8c &n) { return n.w2 & 0x3ffffffu; }
RT_HD uint32_t rt_node8c_tri_base(const DevNode8c &n) { return n.w3 & 0xfffffffu; }
RT_HD void rt_node8c_set_links(DevNode8c &n, uint32_t imask, uint32_t lmask, uint32_t child_base, uint32_t tri_base) {
    n.w1 = (n.w1 & 0x003fffffu) | ((imask & 0xffu) << 22);
    n.w2 = (child_base & 0x3ffffffu) | ((lmask & 0x3fu) << 26);
    n.w3 = (tri_base & 0xfffffffu) | (((lmask >> 6) & 3u) << 28);
}
// The corner grid of a scene: power-of-two cells, 2^RT_GRID_BITS of them per axis covering the (padded) scene box plus a quarter
// of its extent on either side, so that a refit may move things a little without leaving the grid.
RT_HD void rt_grid_for(const float *scene_lo, const float *scene_hi, float *grid_lo, float *grid_cell) {
    for (int k = 0; k < 3; k++) {
        const double ext = (double)scene_hi[k] - (double)scene_lo[k];
        const double want = (ext > 0.0 ? ext * 1.5 : 1.0) / (double)(1u << RT_GRID_BITS);
        int e;
        frexp(want, &e);                                     // want = m * 2^e, m in [0.5, 1): 2^e >= want
        if (e < -100) e = -100;
        grid_cell[k] = (float)ldexp(1.0, e);
        // lower corner: a multiple of the cell at or below (scene_lo - ext / 4), so that cell arithmetic in doubles is exact
        grid_lo[k] = (float)(floor(((double)scene_lo[k] - 0.25 * ext) / (double)grid_cell[k]) * (double)grid_cell[k]);
    }
}
// true when [lo, hi] lies inside the grid (with room for the padding the packer adds)
RT_HD bool rt_grid_holds(const float *lo, const float *hi, const float *grid_lo, const float *grid_cell) {
    for (int k = 0; k < 3; k++) {
        const double top = (double)grid_lo[k] + (double)grid_cell[k] * (double)((1u << RT_GRID_BITS) - 2u);
        if (!((double)rt_pad_lo(lo[k]) >= (double)grid_lo[k] && (double)rt_pad_hi(hi[k]) <= top)) return false;
    }
    return true;
}
// One compact node: child boxes -> corner cell indices, per-axis step exponents, outward-rounded plane offsets.  Links are left alone.
RT_HD void rt_pack_node8c(DevNode8c &n, const float (*blo)[3], const float (*bhi)[3], uint32_t present_mask, const float *grid_lo, const float *grid_cell) {
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool any = false;
    for (int c = 0; c < 8; c++) {
        if (!((present_mask >> c) & 1u)) continue;
        for (int k = 0; k < 3; k++) {
            const double l = rt_pad_lo(blo[c][k]), h = rt_pad_hi(bhi[c][k]);
            if (!any || l < lo[k]) lo[k] = l;
            if (!any || h > hi[k]) hi[k] = h;
        }
        any = true;
    }
    uint32_t f[3], e4[3];
    double corner[3], inv_step[3];
    for (int k = 0; k < 3; k++) {
        double cell = floor((lo[k] - (double)grid_lo[k]) / (double)grid_cell[k]);
        const double top = (double)((1u << RT_GRID_BITS) - 1u);
        if (!(cell >= 0.0)) cell = 0.0;                      // (outside the grid: the caller checks rt_grid_holds; stay defined)
        if (cell > top) cell = top;
        f[k] = (uint32_t)cell;
        corner[k] = (double)grid_lo[k] + cell * (double)grid_cell[k];
        const double ext = any ? hi[k] - corner[k] : 0.0;
        int e = 0;                                           // step = cell * 2^(e - 8)
        while (e < 15 && 255.0 * ldexp((double)grid_cell[k], e - 8) < ext) e++;
        e4[k] = (uint32_t)e;
        inv_step[k] = 1.0 / ldexp((double)grid_cell[k], e - 8);
    }
    uint32_t q[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int c = 0; c < 8; c++) {
        const bool present = (present_mask >> c) & 1u;
        for (int k = 0; k < 3; k++) {
            uint32_t ql = 255u, qh = 0u;                       // absent: an inverted box
            if (present) {
                const double l = floor(((double)rt_pad_lo(blo[c][k]) - corner[k]) * inv_step[k]);
                const double h = ceil(((double)rt_pad_hi(bhi[c][k]) - corner[k]) * inv_step[k]);
                ql = l >= 0.0 && l <= 255.0 ? (uint32_t)l : (l < 0.0 ? 0u : 255u);
                qh = h >= 0.0 && h <= 255.0 ? (uint32_t)h : (h > 255.0 ? 255u : 0u);
            }
            q[k][c >> 1] |= (ql << (8 * (c & 1))) | (qh << (16 + 8 * (c & 1)));
        }
    }
    n.w0 = f[0] | (f[1] << 14) | (e4[0] << 28);
    n.w1 = (n.w1 & 0xffc00000u) | f[2] | (e4[1] << 14) | (e4[2] << 18);
    for (int p = 0; p < 4; p++) { n.qx[p] = q[0][p]; n.qy[p] = q[1][p]; n.qz[p] = q[2][p]; }
}

// geometry transform of one corner, the host's expression (host/bvh_builder.cpp gather_triangles): every
// product and sum rounds once (this file is compiled with -ffp-contract=off on both sides)
RT_HD void rt_transform_point(const float *m, const float *v, float *dst) {
    for (int r = 0; r < 3; r++) {
        float acc = m[4 * r] * v[0] + m[4 * r + 1] * v[1];
        acc = acc + m[4 * r + 2] * v[2];
        dst[r] = acc + m[4 * r + 3];
    }
}
