// rt_device.h — device-side data layouts and launch parameters.
//
// The reference keeps scene data in WGSL storage buffers (vertex/index/models,
// shaders/main.wgsl:1-10) and its BVH inside the Vulkan driver.  Here the host
// repacks everything once per scene upload into 16-byte-aligned records that a
// lane fetches with a few dwordx4 loads:
//
//   DevNode   48 B  one BVH2 inner node holding BOTH children's boxes and
//                   references, so one visit = three 16-B fetches and decides
//                   both children;
//   DevTri    48 B  corner + two edge vectors in leaf order + global id;
//   DevShade 128 B  everything trace_ray's attribute part needs
//                   (shaders/main.wgsl:86-101) for one triangle, one L2 line,
//                   instead of the reference's 12 B + 3x48 B + 128 B gathers;
//   DevMaterial 64 B the material fields sample() reads.
#pragma once
#include <stdint.h>

// One lane reads a node with three dwordx4 loads.  Child boxes are stored as f16
// OFFSETS from the node's own lower corner (lo), rounded outward, so the slab
// test feeds them straight into v_fma_mix_f32 (f16 source, f32 result):
//   base_k = lo_k * inv_k - o_k * inv_k        3 FMAs per node
//   t      = off * inv_k + base_k              12 mixed FMAs, no unpacking
// The rounding error of an offset is <= 2^-11 of the PARENT's extent, i.e. the
// boxes grow by a fraction of a percent, never shrink.
struct alignas(16) DevNode {
    float lox, loy, loz; int32_t ref0;          // lower corner of (child 0 U child 1); child 0 reference
    uint32_t c0x, c0y, c0z; int32_t ref1;       // child 0, per axis: (hi_offset_f16 << 16) | lo_offset_f16; child 1 reference
    uint32_t c1x, c1y, c1z; uint32_t pad;       // child 1
};
static_assert(sizeof(DevNode) == 48, "DevNode");
#define RT_NODE_CHUNKS 3                        // 16-byte chunks per node

// The wide node (RT_OPT_NODE_FORMAT = 3): an 8-ary BVH node of 80 bytes = five 16-byte fetches for EIGHT child boxes
// (the 48-byte binary node: three fetches for two).  The L1 path's cost is per lane-load instruction, so
// fetches per ray are what this record minimises: ~4 visits x 5 fetches instead of ~12 x 3.
//   - child boxes: 8-bit plane offsets from the node's f32 lower corner in units of a per-axis power of two
//     (scale_k = 2^(exp_k - 127)), rounded outward; per axis and child PAIR p one dword [c(2p).lo, c(2p+1).lo, c(2p).hi,
//     c(2p+1).hi], so rotating it by 16 bits swaps the planes a ray enters and leaves through (as in the q8 binary node);
//   - children are either inner nodes (bit in imask) or ONE primitive each (bit in lmask), or absent;
//   - a node's inner children are consecutive nodes starting at child_base, its leaf children's primitives consecutive
//     slots starting at tri_base, both in child-slot order: child i is node child_base + 5 * popcount(imask & below(i))
//     (child_base is a 16-byte chunk index) or slot tri_base + popcount(lmask & below(i)) — no per-child references;
//   - child slots are assigned at build time by position: slot bit k set = the child lies towards +axis k from the
//     node's centre, so that slot XOR (ray direction signs) orders children near to far without sorting (the octant
//     order of Ylitie, Karras & Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", 2017).
struct alignas(16) DevNode8 {
    float lox, loy, loz; uint32_t exps_imask;   // bytes 0..2: biased exponents of the x, y, z scales; byte 3: imask
    uint32_t child_base, tri_base, lmask, pad;
    uint32_t qx[4], qy[4], qz[4];
};
static_assert(sizeof(DevNode8) == 80, "DevNode8");
#define RT_NODE8_CHUNKS 5

// The compact 8-ary node (RT_OPT_NODE_FORMAT = 4): the same tree and the same plane dwords in 64 bytes = FOUR fetches and ONE
// 64-byte line per visit (the 80-byte record straddles two).  The node's lower corner is not three floats but three 14-bit
// cell indices of a per-scene grid (RenderParams::grid_lo / grid_cell: a power-of-two cell, the scene's box plus a margin in
// 2^14 cells), rounded down; the plane offsets count steps of cell * 2^(e - 8) with one 4-bit e per axis.  A coarse corner
// costs next to nothing: the planes are relative to the corner as stored, so only their step grows by the fraction of a
// cell the corner moved.  Header (128 bits):
//   w0 = fx | fy << 14 | ex << 28          w1 = fz | ey << 14 | ez << 18 | imask << 22
//   w2 = child_base (node INDEX, 26 bits) | (lmask & 63) << 26          w3 = tri_base (28 bits) | (lmask >> 6) << 28
struct alignas(16) DevNode8c {
    uint32_t w0, w1, w2, w3;
    uint32_t qx[4], qy[4], qz[4];
};
static_assert(sizeof(DevNode8c) == 64, "DevNode8c");
#define RT_NODE8C_CHUNKS 4
#define RT_NODE8C_MAX_NODES (1u << 26)
#define RT_GRID_BITS 14

// Child reference: >= 0 → inner DevNode, as its index * RT_NODE_CHUNKS (a
// float4 index: the address needs a shift only); < 0 → leaf, with
// ~ref = (first_slot << 3) | count, count in 0..7 (0 = empty child).
static inline int32_t dev_leaf_ref(uint32_t first_slot, uint32_t count) {
    return (int32_t)~((first_slot << 3) | count);
}

// corner a and the two edges e1 = b - a, e2 = c - a (the f32 differences the
// contract's triangle test starts with, taken once at upload)
struct alignas(16) DevTri {
    float ax, ay, az, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z; uint32_t gid; uint32_t is_sphere, pad1;   // gid = primitive's place in (index-buffer, then sphere) order: the tie-break key
                                                     // sphere record: a = centre, e1x = radius, is_sphere = 1
};
static_assert(sizeof(DevTri) == 48, "DevTri");

// Chunk order follows use: an untextured hit reads chunks 0-4 (five 16-byte loads: positions, normals, ids);
// only textured materials read the uv chunks, only spheres read pad0.
struct alignas(16) DevShade {
    float p0[3], n0x;                       // vertex 0 position; vertex 0 normal, one component per chunk
    float p1[3], n0y;
    float p2[3], n0z;
    float n1[3]; uint32_t geometry;         // geometry_index (bit 31: sphere record)
    float n2[3]; uint32_t primitive;        // primitive_index
    float uv0[2], uv1[2];
    float uv2[2]; uint32_t pad0, pad1;      // pad0: a sphere's model index
    uint32_t pad2[4];
};
static_assert(sizeof(DevShade) == 128, "DevShade");

// Chunk order follows use: the tag comes with the first load; a metal hit reads chunks 0-2, a dielectric one 0 and 3.
struct alignas(16) DevMaterial {
    uint32_t tag; float diffuse[3];
    float specular_probability; float specular[3];
    float roughness; float emitted[3];            // emitted = emission_color * emission_strength (main.wgsl:56)
    float refractive_index; uint32_t diffuse_texture, normal_texture, pad;   // texture ids 1-based, 0 = none (src/scene.rs:170-174)
};
static_assert(sizeof(DevMaterial) == 64, "DevMaterial");

// one texture: w x h RGBA8 texels starting at `offset` (in texels) of the packed
// texel array, rows already flipped vertically (src/scene.rs:77)
struct DevTexture {
    uint32_t offset, width, height, pad;
};

// primitive / child box of the device refit (rt_refit.hip)
struct RtRefitBox {
    float lo[3], hi[3];
};

struct DevCounters {
    unsigned long long rays, pixel_frames, node_visits, triangle_tests;
    unsigned long long wave_node_iters, wave_leaf_iters, wave_steps;
    unsigned long long guard_trips;      // RT_GUARD builds: rays ended by a failed range check or the step cap (must stay 0)
};

// persistent-kernel work queue head: next unclaimed pixel slot of the launch
struct DevQueue {
    unsigned int head;
    unsigned int pad[3];
};

// rt_group gather: where each shard's packed rows start in the gathered buffer (in rows)
#define RT_GROUP_MAX 64
struct RtGatherLayout {
    uint32_t width, height, n_shards, pad;
    uint32_t first_row[RT_GROUP_MAX];
};

// Material records (64 B each) the phased kernel copies into LDS at its start when the scene has no more than this
// many: the L1 path is the busiest unit of that kernel and LDS is nearly idle.  Scenes with more read them from memory.
#define RT_TOP_NODES 64u                 // nodes the host numbers breadth-first at the start of the node array (the levels every ray walks share cache lines)
#ifndef RT_LDS_MATERIALS
#define RT_LDS_MATERIALS 16u
#endif

struct RenderParams {
    const DevNode *nodes;
    const DevTri *tris;
    const DevShade *shade;
    const DevMaterial *materials;
    const DevTexture *textures;   // EXT builds only
    const uint32_t *texels;
    uint32_t node_q8, node_wide;         // node_q8 = 1: 32-byte binary nodes with 8-bit plane offsets (rt_pack.h), stride 2 chunks;
                                         // node_wide = 1: 80-byte 8-ary nodes (DevNode8), 2: 64-byte ones (DevNode8c); `nodes` points at them
    float grid_lo[3];                    // DevNode8c: lower corner of the corner grid
    uint32_t grid_cell_bits[3];          // DevNode8c: bit pattern of (grid cell * 2^-8), a power of two, per axis
    uint32_t sampler;                    // RT_OPT_SAMPLER: bit 0 linear filter, bit 1 repeat addressing (0 = nearest, clamp to edge)
    uint32_t pad_grid;
    uint32_t n_textures, n_materials;   // n_materials: records behind `materials`
    float *accum;            // local rows x W x 4 f32
    uint32_t *aov;           // local rows x W x 2 u32, or null
    DevCounters *counters;
    DevQueue *queue;         // persistent variant only; zeroed before each launch

    uint32_t width, height;      // uniform.window
    uint32_t tiles_x;            // ceil(W / 8)
    float inv_tiles_x;           // 1 / tiles_x (slot -> tile row without an integer division)
    uint32_t local_bands;        // 8-row bands this shard owns
    uint32_t shard_index, shard_count;
    uint32_t frame, accumulation_frame, flags;
    uint32_t max_bounces, samples;
    uint32_t stack_entries;      // LDS traversal stack depth per lane
    uint32_t chunk;              // pixel slots a wave claims from the queue per atomic
    // Multi-frame launches (phased kernel only): frames frame..frame+n_frames-1 of every pixel in ONE launch.
    // Work item q = frame_offset * slots + slot; each pixel-frame parks its sample sum in
    // stage[frame_offset * stage_stride + pixel] and rt_blend_frames folds them into `accum` in frame order.
    uint32_t n_frames;           // >= 1
    uint32_t local_rows;         // rows of `accum`
    float4 *stage;               // n_frames x stage_stride, or null when n_frames == 1
    size_t stage_stride;         // >= local rows x W
    uint32_t front_face_cw;
    float exposure, environment;
    float cam_pos[3];
    float forward[3], right[3], up[3];   // ray_direction's basis (shaders/ray.wgsl:2-4), hoisted to the host
    float aspect, fov_scale;             // ctx.camera.aspect, tan(fov/2) (shaders/ray.wgsl:6-7)
    float scene_abs[3];                  // max |coordinate| of the BVH root box per axis (slab-test error bound)
    // RT_GUARD builds (make guard): what traversal references are checked against before they are used
    uint32_t n_node_chunks;              // 16-byte chunks in the node array
    uint32_t wide_stride;                // 8-ary nodes: 16-byte chunks from one node to the next (5 = packed, 8 = one node per 128-byte line)
    uint64_t n_slots;                    // triangle / sphere records
    int32_t tune[8];                     // scheduler weights: [0] NODE, [1] LEAF, [2] SHADE, [4] FETCH (others unused by the kernel)
#ifdef RT_TAIL_PROBE
    // developer build (make EXTRA=-DRT_TAIL_PROBE): per wave {start, first empty claim, exit} in 100 MHz ticks,
    // summarised on stderr after every multi-frame launch (tail of a launch)
    unsigned long long *probe;
#endif
};
