// rt_device.h — device-side data layouts and launch parameters.
//
// The reference keeps scene data in WGSL storage buffers (vertex/index/models,
// shaders/main.wgsl:1-10) and its BVH inside the Vulkan driver.  Here the host
// repacks everything once per scene upload into 16-byte-aligned records that a
// lane fetches with a few dwordx4 loads:
//
//   DevNode   64 B  one BVH2 inner node holding BOTH children's boxes and
//                   references, so one visit = one 64-B fetch (half a 128-B
//                   L2 line) and decides both children;
//   DevTri    48 B  the three corner positions in leaf order + global id;
//   DevShade 128 B  everything trace_ray's attribute part needs
//                   (shaders/main.wgsl:86-101) for one triangle, one L2 line,
//                   instead of the reference's 12 B + 3x48 B + 128 B gathers;
//   DevMaterial 64 B the material fields sample() reads.
#pragma once
#include <stdint.h>

struct alignas(16) DevNode {
    float lo0[3]; int32_t ref0;   // child 0: box min, reference
    float hi0[3]; int32_t ref1;   // child 0: box max; child 1 reference
    float lo1[3]; uint32_t pad0;  // child 1: box min
    float hi1[3]; uint32_t pad1;  // child 1: box max
};
static_assert(sizeof(DevNode) == 64, "DevNode");

// Child reference: >= 0 → index of an inner DevNode; < 0 → leaf, with
// ~ref = (first_slot << 3) | count, count in 0..7 (0 = empty child).
static inline int32_t dev_leaf_ref(uint32_t first_slot, uint32_t count) {
    return (int32_t)~((first_slot << 3) | count);
}

struct alignas(16) DevTri {
    float ax, ay, az, bx;
    float by, bz, cx, cy;
    float cz; uint32_t gid; uint32_t pad0, pad1;   // gid = triangle's place in index-buffer order (tie-break key)
};
static_assert(sizeof(DevTri) == 48, "DevTri");

struct alignas(16) DevShade {
    float p0[3], n0x;  float n0y, n0z, uv0[2];
    float p1[3], n1x;  float n1y, n1z, uv1[2];
    float p2[3], n2x;  float n2y, n2z, uv2[2];
    uint32_t geometry, primitive, pad0, pad1;
    uint32_t pad2[4];
};
static_assert(sizeof(DevShade) == 128, "DevShade");

struct alignas(16) DevMaterial {
    float diffuse[3];  float specular_probability;
    float specular[3]; float roughness;
    float emitted[3];  float refractive_index;    // emitted = emission_color * emission_strength (main.wgsl:56)
    uint32_t tag, pad[3];
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial");

struct DevCounters {
    unsigned long long rays, pixel_frames, node_visits, triangle_tests;
};

struct RenderParams {
    const DevNode *nodes;
    const DevTri *tris;
    const DevShade *shade;
    const DevMaterial *materials;
    float *accum;            // local rows x W x 4 f32
    uint32_t *aov;           // local rows x W x 2 u32, or null
    DevCounters *counters;

    uint32_t width, height;      // uniform.window
    uint32_t tiles_x;            // ceil(W / 8)
    uint32_t local_bands;        // 8-row bands this shard owns
    uint32_t shard_index, shard_count;
    uint32_t frame, accumulation_frame, flags;
    uint32_t max_bounces, samples;
    uint32_t stack_entries;      // LDS traversal stack depth per lane
    uint32_t front_face_cw;
    float exposure, environment;
    float cam_pos[3];
    float forward[3], right[3], up[3];   // ray_direction's basis (shaders/ray.wgsl:2-4), hoisted to the host
    float aspect, fov_scale;             // ctx.camera.aspect, tan(fov/2) (shaders/ray.wgsl:6-7)
};
