// rt_math.h — the f32 arithmetic contract ("RTN-1") shared by the
// HIP kernels and the library's host code (camera basis on the host).
//
// IEEE binary32, round-to-nearest-even.  This file is compiled with
// -ffp-contract=off on both sides, so every * and + below rounds once; FMA
// appears only where fmaf() is written out.  Division and sqrt are the
// correctly rounded forms (hipcc default, -fhip-fp32-correctly-rounded-divide-sqrt).
// Expression shapes follow the WGSL they replace (cited per function).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

struct rt_v3 { float x, y, z; };

RT_HD rt_v3 v3(float x, float y, float z) { rt_v3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD rt_v3 operator+(rt_v3 a, rt_v3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD rt_v3 operator-(rt_v3 a, rt_v3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD rt_v3 operator*(rt_v3 a, rt_v3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD rt_v3 operator*(rt_v3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD rt_v3 operator-(rt_v3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD float rt_dot(rt_v3 a, rt_v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD rt_v3 rt_cross(rt_v3 a, rt_v3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// normalize(v): one reciprocal of the length, three multiplies
RT_HD rt_v3 rt_normalize(rt_v3 a) {
    float inv = 1.0f / sqrtf(rt_dot(a, a));
    return a * inv;
}
// mix(a, b, t) = a*(1-t) + b*t
RT_HD rt_v3 rt_mix(rt_v3 a, rt_v3 b, float t) { return a * (1.0f - t) + b * t; }
// reflect(e1, e2) = e1 - 2*dot(e2, e1)*e2
RT_HD rt_v3 rt_reflect(rt_v3 e1, rt_v3 e2) { return e1 - e2 * (2.0f * rt_dot(e2, e1)); }
// refract(e1, e2, e3)
RT_HD rt_v3 rt_refract(rt_v3 e1, rt_v3 e2, float e3) {
    float d = rt_dot(e2, e1);
    float k = 1.0f - e3 * e3 * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return e1 * e3 - e2 * (e3 * d + sqrtf(k));
}

// fused forms, used only by the intersectors
RT_HD float rt_dot_fma(rt_v3 a, rt_v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_HD rt_v3 rt_cross_fma(rt_v3 a, rt_v3 b) {
    return v3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}

// sin and cos of the contract: quadrant reduction by a three-part pi/2 with
// fmaf, then single-precision minimax polynomials on [-pi/4, pi/4].
RT_HD void rt_sincos(float x, float *s_out, float *c_out) {
    float kf = floorf(x * 0.63661977236758134308f + 0.5f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z
               - 0.5f * z + 1.0f;
    int q = (int)kf & 3;
    float s = (q & 1) ? cp : sp;
    float c = (q & 1) ? sp : cp;
    if (q & 2) s = -s;
    if ((q + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}

// LCG of shaders/random.wgsl:3-7 (fract() of a value in [0,1) is the value)
RT_HD float rt_rand(uint32_t &seed) {
    seed = seed * 747796405u + 2891336453u;
    return (float)(seed >> 9) / (float)(1u << 23);
}
// shaders/main.wgsl:23
RT_HD uint32_t rt_seed(uint32_t pixel_idx, uint32_t frame) {
    return (pixel_idx * 2479898233u) ^ (frame * 98379842u);
}

// Capture, src/ui.rs:84-88: `(x * 255.0) as u8` — Rust's float-to-int cast truncates toward zero, saturates, NaN -> 0
RT_HD uint8_t rt_capture_u8(float x) {
    const float v = x * 255.0f;
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
