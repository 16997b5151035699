/*
 * rt_abi.h — C ABI of the MI355X path-tracing backend (librt_amd.so).
 *
 * This library is the drop-in for the reference's WGPU compute pass: the
 * `ComputePipeline` built at /root/reference/src/main.rs:44-54 and dispatched
 * at /root/reference/src/app.rs:100-101, together with the resources bound to
 * it (src/main.rs:46-53) and the driver acceleration structure created at
 * src/scene.rs:56-64.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every call returns 0 on success or a negative RtStatus; rt_last_error()
 *     returns a thread-local description of the last failure (the Rust side
 *     maps non-zero to anyhow!(), mirroring src/main.rs:26 `Result<()>`);
 *   - one context is driven from one host thread (the reference's winit loop,
 *     src/app.rs:65-103); dispatches are asynchronous on the context's HIP
 *     stream, like a wgpu queue submit; rt_sync() waits;
 *   - host arrays are copied at upload time (the reference uploads copies too,
 *     src/scene.rs:51-52); the library owns BVH memory.
 *
 * Byte layouts are WGSL/encase layouts (shaders/types.wgsl:1-57,
 * src/types.rs:17-90, src/camera.rs:18-26); every offset is static_asserted.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime failure at create */
    RT_ERR_HIP = -3,            /* a HIP call failed; see rt_last_error() */
    RT_ERR_NO_SCENE = -4,       /* dispatch before rt_scene_upload */
    RT_ERR_NO_ACCUM = -5,       /* dispatch before rt_accum_resize / rt_accum_bind */
    RT_ERR_OUT_OF_MEMORY = -6,
    RT_ERR_IO = -7,             /* host-side loaders */
    RT_ERR_PARSE = -8
} RtStatus;

/* ----------------------------------------------------------------- layouts */
/* Camera: shaders/types.wgsl:14-21, src/camera.rs:18-26. size 32, align 16. */
typedef struct RtCamera {
    float position[3]; /* @0  */
    float pitch;       /* @12 */
    float yaw;         /* @16 */
    float fov;         /* @20 */
    float aspect;      /* @24 */
    float _pad;        /* @28 */
} RtCamera;

/* Uniform: shaders/types.wgsl:1-12, src/types.rs:17-29. size 80. */
typedef struct RtUniform {
    uint32_t window[2];          /* @0  */
    uint32_t _pad0[2];           /* @8  */
    RtCamera camera;             /* @16 */
    uint32_t frame;              /* @48 */
    uint32_t accumulation_frame; /* @52 */
    uint32_t flags;              /* @56  bit0 = CULL_BACKFACES (src/types.rs:31-35) */
    float exposure;              /* @60 */
    float environment;           /* @64  (WGSL spells it `enviroment`) */
    uint32_t max_bounces;        /* @68 */
    uint32_t samples;            /* @72 */
    uint32_t _pad1;              /* @76 */
} RtUniform;

#define RT_FLAG_CULL_BACKFACES 1u

/* MetalMaterial: shaders/types.wgsl:29-41, src/types.rs:45-58. size 80. */
typedef struct RtMetalMaterial {
    float diffuse_color[3];     /* @0  */
    float _pad0;                /* @12 */
    float specular_color[3];    /* @16 */
    float specular_probability; /* @28 */
    float roughness;            /* @32 */
    float _pad1[3];             /* @36 */
    float emission_color[3];    /* @48 */
    float emission_strength;    /* @60 */
    uint32_t diffuse_texture;   /* @64  1-based, 0 = none */
    uint32_t normal_texture;    /* @68  1-based, 0 = none */
    uint32_t _pad2[2];          /* @72 */
} RtMetalMaterial;

/* Material: shaders/types.wgsl:23-27, src/types.rs:37-43. size 112. */
typedef struct RtMaterial {
    uint32_t tag;               /* @0   0 = metal, 1 = dielectric */
    uint32_t _pad0[3];          /* @4   */
    RtMetalMaterial metal;      /* @16  */
    float refractive_index;     /* @96  DielectricMaterial */
    uint32_t _pad1[3];          /* @100 */
} RtMaterial;

/* Model / GpuModel: shaders/types.wgsl:47-51, src/types.rs:65-70. size 128. */
typedef struct RtGpuModel {
    RtMaterial material;        /* @0   */
    uint32_t vertex_start;      /* @112 */
    uint32_t index_start;       /* @116 */
    uint32_t _pad[2];           /* @120 */
} RtGpuModel;

/* Vertex: shaders/types.wgsl:53-57, src/types.rs:85-90. size 48. */
typedef struct RtVertex {
    float position[3]; /* @0  */
    float _pad0;       /* @12 */
    float normal[3];   /* @16 */
    float _pad1;       /* @28 */
    float uv[2];       /* @32 */
    float _pad2[2];    /* @40 */
} RtVertex;

/* GeometryPrimitive: src/scene.rs:141-147 (one per OBJ object). size 24. */
typedef struct RtGeometryPrimitive {
    uint32_t first_vertex;
    uint32_t vertex_count;
    uint32_t first_index;
    uint32_t index_count;
    uint64_t transformation_offset; /* index into the 3x4 transform array */
} RtGeometryPrimitive;

/* Build-defined extension (no counterpart in the reference):
 * analytic sphere primitive for BASELINE config C4.  size 32. */
typedef struct RtSphere {
    float center[3];
    float radius;
    uint32_t model;   /* index into models[] for the material */
    uint32_t _pad[3];
} RtSphere;

/* One texture image: RGBA8, rows in FILE order (top row first).  The library
 * flips it vertically at upload, as src/scene.rs:77 does. */
typedef struct RtTexture {
    const uint8_t *rgba8;
    uint32_t width, height;
} RtTexture;

/* Accumulation buffer element: array<vec3f> → stride 16 B, xyz @0
 * (shaders/main.wgsl:2, src/app.rs:23). */
#define RT_ACCUM_STRIDE_BYTES 16u

/* Primary-hit AOV element (new, for parity): sample 0 of the
 * dispatched frame.  geometry_index = OBJ object order, primitive_index =
 * triangle index inside that object.  Miss = 0xFFFFFFFF in both. */
typedef struct RtHitId {
    uint32_t geometry_index;
    uint32_t primitive_index;
} RtHitId;

/* Counters since the last rt_stats_reset (device-side, exact). */
typedef struct RtStats {
    uint64_t rays;            /* trace_ray calls (shaders/main.wgsl:44) */
    uint64_t pixel_frames;    /* invocations of `main` that passed the bounds check */
    uint64_t node_visits;     /* only counted when RT_OPT_COUNT_TRAVERSAL is set */
    uint64_t triangle_tests;  /* idem */
    uint64_t wave_node_iters; /* idem: node-loop iterations summed over wavefronts (x64 = lane slots) */
    uint64_t wave_leaf_iters; /* idem: triangle-loop iterations summed over wavefronts */
    uint64_t wave_steps;      /* idem: (trace, shade) steps summed over wavefronts */
    uint64_t dispatches;      /* compute passes (frames) rendered */
    uint64_t bvh_nodes;       /* static: nodes in the device BVH */
    uint64_t bvh_bytes;       /* static: bytes of nodes + triangle records on device */
    double   bvh_build_seconds; /* last build or refit */
    uint64_t bvh_refits;      /* rt_models_update calls served by a refit */
    uint64_t bvh_rebuilds;    /* full builds (scene/sphere uploads, refit disabled) */
    uint64_t launches;        /* render-kernel launches: rt_dispatch_frames folds a run of frames into one */
    uint32_t stack_entries;   /* static: traversal stack entries per lane (BVH depth + 1), 1 KiB of LDS each per workgroup */
    uint32_t resident_blocks; /* workgroups of 256 the persistent kernel keeps resident (0 before the first launch) */
    uint64_t guard_trips;     /* librt_amd_guard.so only (make guard: every traversal reference range-checked, steps capped):
                                 rays ended by a failed check.  Always 0 in the shipping library, which compiles no checks. */
    uint32_t node_format;     /* static: the node record the uploaded scene actually got (RT_OPT_NODE_FORMAT values 1..4; 0 = no scene) */
    uint32_t pad0;
} RtStats;

/* rt_set_option keys */
#define RT_OPT_COUNT_TRAVERSAL 1   /* 0/1: count node visits + triangle tests (slower kernel) */
#define RT_OPT_AOV             2   /* 0/1: write primary hit ids on every dispatch */
#define RT_OPT_KERNEL          3   /* kernel variant: 2 (default) lanes scheduled by phase; 0 one ray per lane per step,
                                      1 one wave per 8x8 tile — both kept for A/B measurements, same bits */
#define RT_OPT_FRONT_FACE_CW   4   /* 0 (default): CCW = front; 1: CW = front */
#define RT_OPT_REFIT           5   /* what rt_models_update does to the BVH (≙ acceleration_structure.update(), src/app.rs:61):
                                      2 (default) refit on the device: the scene upload keeps the tree's topology there
                                      (set the option BEFORE the upload), triangle records and node boxes are recomputed
                                      on the GPU and only the 3x4 transforms and the materials cross PCIe;
                                      1 refit on the host and upload everything; 0 full rebuild */
#define RT_OPT_FRAME_PIPELINE  7   /* 1 (default): rt_dispatch_frames renders its frames in one launch, frames of
                                      different pixels overlapping (same bits as frame-by-frame); 0: one launch per frame */
#define RT_OPT_NODE_STRIDE     6   /* BVH node stride in 16-byte units for the NEXT scene upload: binary 48-byte nodes 3 (tightest) or 4
                                      (one 64-byte sector per node); 80-byte 8-ary nodes 8 (one 128-byte line per node); 0 = packed */

#define RT_OPT_NODE_FORMAT     8   /* BVH node record of the NEXT scene upload.  All formats are conservative: same image bits.
                                      1 = 48-byte binary nodes (f16 plane offsets, three 16-byte fetches for two boxes);
                                      2 = 32-byte binary nodes (8-bit offsets and a per-node power-of-two scale, two fetches);
                                      3 = 80-byte 8-ary nodes (8-bit offsets, per-axis scales, one primitive per leaf child, octant-ordered
                                          traversal with a two-word group stack: five fetches for eight boxes);
                                      4 = 64-byte 8-ary nodes (the same tree; the corner is three 14-bit cells of a per-scene grid:
                                          four fetches and one 64-byte line per visit);
                                      0 (default) = by size: 48-byte binary nodes while the nodes stay in the caches, 64-byte 8-ary
                                          nodes once they would exceed ~12 MiB.  Formats 2-4 need a scene extent <= 2^18 (else 1 is used). */
#define RT_OPT_SAMPLER         9   /* the texture sampler of shaders/misc.wgsl:6-8.  The reference creates it inside the un-vendored `compute` crate
                                      (src/main.rs:43 `gpu.create_sampler()`), so its state is UNPINNED; the default 0 = wgpu's SamplerDescriptor
                                      defaults.  bit 0: 1 = linear (bilinear, f32 weights, texel centres at +0.5) instead of nearest;
                                      bit 1: 1 = repeat instead of clamp-to-edge.  Takes effect at the next dispatch. */

typedef struct rt_context rt_context;

/* ------------------------------------------------------------- lifecycle */
/* ≙ Gpu::builder()…build() + compute_pipeline(…).finish(), src/main.rs:27-54 */
int rt_create(int device_id, rt_context **out);
int rt_destroy(rt_context *ctx);
const char *rt_last_error(void);
/* Use the caller's HIP stream (a hipStream_t) for all device work. NULL = the
 * library's own stream. */
int rt_set_stream(rt_context *ctx, void *hip_stream);
/* Keys >= 100 are developer knobs of tools/ (scheduler weights, queue grain, builder parameters); they are not
 * part of the interface and may change. */
int rt_set_option(rt_context *ctx, int key, int64_t value);

/* ------------------------------------------------------------------ scene */
/* ≙ Scene::finish, src/scene.rs:50-93: uploads vertex/index/model arrays and
 * builds the acceleration structure (host SAH BVH build + repack, the part
 * the Vulkan driver does for the reference at src/scene.rs:56-64).
 * `transforms` = n_primitives row-major 3x4 matrices, or NULL for identity
 * (src/scene.rs:53-54).  A triangle or sphere with a NaN / infinite corner, centre or radius (after its transform) is accepted and is
 * never hit — what the arithmetic contract's tests give for it — and does not disturb the rest of the scene.
 * Arrays are validated before anything is built (ranges inside their buffers, index counts multiples of 3, a model per geometry);
 * a refused upload leaves the context WITHOUT a scene (RT_ERR_NO_SCENE on dispatch), while a refused rt_spheres_upload or
 * rt_models_update leaves the scene that was in force untouched. */
int rt_scene_upload(rt_context *ctx,
                    const RtVertex *vertices, uint64_t n_vertices,
                    const uint32_t *indices, uint64_t n_indices,
                    const RtGeometryPrimitive *primitives, uint32_t n_primitives,
                    const float *transforms,
                    const RtGpuModel *models, uint32_t n_models);
/* Build-defined: adds analytic spheres to the scene (BASELINE config C4).
 * Must follow rt_scene_upload; rebuilds the acceleration structure. */
int rt_spheres_upload(rt_context *ctx, const RtSphere *spheres, uint64_t n_spheres);
/* ≙ the texture half of Scene::finish, src/scene.rs:69-83 (binding 7,
 * shaders/main.wgsl:10).  Material texture ids are 1-based indices into this
 * array (src/scene.rs:170-174).  Sampling = shaders/misc.wgsl:6-8 with the
 * sampler taken as nearest / clamp-to-edge (the reference's sampler is created
 * inside the un-vendored `compute` crate: unpinned).
 * Must follow rt_scene_upload (which clears textures). */
int rt_textures_upload(rt_context *ctx, const RtTexture *textures, uint32_t n_textures);
/* ≙ App::upload_models, src/app.rs:46-62: re-upload models (materials) and the
 * per-geometry transforms, then rebuild the acceleration structure
 * (acceleration_structure.update()): the tree topology is kept and its boxes
 * are refitted to the moved triangles, on the device by default (RT_OPT_REFIT).
 * transforms may be NULL (= unchanged). */
int rt_models_update(rt_context *ctx, const RtGpuModel *models, uint32_t n_models,
                     const float *transforms);

/* ----------------------------------------------------------- accumulation */
/* Row-tile sharding for multi-GPU: the image is cut into bands
 * of 8 rows (the reference's 8x8 workgroup rows, shaders/main.wgsl:15); band t
 * belongs to shard t % shard_count.  A context renders only its bands and its
 * accumulation buffer holds only those rows, packed (band t is local band
 * t / shard_count).  pixel_idx stays global, so results are bit-identical to
 * an unsharded render.  Default 0 of 1.  Changing the shard drops the accumulation buffer
 * (a bound one is un-bound): call rt_accum_resize / rt_accum_bind again afterwards. */
int rt_set_shard(rt_context *ctx, uint32_t shard_index, uint32_t shard_count);
/* Number of rows / bytes the local accumulation buffer holds for a w x h image. */
uint64_t rt_accum_local_rows(const rt_context *ctx, uint32_t w, uint32_t h);
/* ≙ src/app.rs:87-93: (re)allocate a zero-filled accumulation buffer. */
int rt_accum_resize(rt_context *ctx, uint32_t w, uint32_t h);
/* Render into caller-owned device memory instead (e.g. a torch tensor that
 * RCCL will gather).  bytes must be >= local_rows*w*16.  Not zeroed. */
int rt_accum_bind(rt_context *ctx, void *device_ptr, uint64_t bytes, uint32_t w, uint32_t h);

/* --------------------------------------------------------------- dispatch */
/* ≙ uniform_buffer.upload + compute_pipeline.dispatch(ceil(W/8), ceil(H/8), 1),
 * src/app.rs:95-101.  uniform->window must equal the accumulation size. */
int rt_dispatch(rt_context *ctx, const RtUniform *uniform);
/* One frame in pieces, for interactive hosts: the reference's open item "Execute compute shader in chunks when in
 * interactive mode" (TODO.md:9; no code in the reference yet — its one dispatch per frame, src/app.rs:100-101, holds the
 * queue for the whole frame, which on a scene of C5's size is a third of a second).  Renders bands [band_begin, band_end)
 * of the context's image — a band is 8 rows, the workgroup rows of shaders/main.wgsl:15; with rt_set_shard the context's
 * OWN bands, numbered from 0 — exactly as rt_dispatch would: pixels are independent (main.wgsl:31-32), so the pieces of a
 * frame, in any order, each band once, leave the bits of one rt_dispatch with the same uniform.  band_end beyond the last
 * band means "to the end"; an empty range (also one that starts beyond the last band) renders nothing, band_begin > band_end is
 * RT_ERR_INVALID_ARGUMENT.  The caller keeps the frame bookkeeping (as with rt_dispatch)
 * and does what it likes between pieces (the calls only queue work on the context's stream; rt_sync waits); RtStats.dispatches counts the piece that ends at the last
 * band; for rt_dispatch_frames' cold-start rule (first dispatch after a resize) any piece, even an empty one, is a dispatch.  Kernel variants 2 (default) and 0 only: variant 1 has no work queue to start part-way. */
int rt_dispatch_bands(rt_context *ctx, const RtUniform *uniform, uint32_t band_begin, uint32_t band_end);
/* n_frames consecutive dispatches; before each one `frame` is bumped and
 * `accumulation_frame` follows the cold-start sequence of src/app.rs:78-96
 * (first dispatch after a resize sees accumulation_frame = 0).  On return
 * *uniform holds the values of the last dispatch.  With the default kernel the
 * run is rendered by ONE launch (RT_OPT_FRAME_PIPELINE): pixel-frames are worked
 * off frame-major, their colours parked per frame (device staging, allocated on
 * first use, <= 4 GiB) and blended into the accumulation buffer in frame order —
 * the same bits as n_frames separate rt_dispatch calls.  Falls back to one launch
 * per frame when RT_OPT_AOV is set, for the other kernel variants and for
 * windows 65536 pixels wide or wider. */
int rt_dispatch_frames(rt_context *ctx, RtUniform *uniform, uint32_t n_frames);
/* Allocates the staging of rt_dispatch_frames for runs of up to n_frames frames now (after
 * rt_accum_resize / rt_accum_bind), so that the dispatch itself does not allocate.  No
 * counterpart in the reference (its accumulation is one dispatch per frame). */
int rt_frames_reserve(rt_context *ctx, uint32_t n_frames);
/* rt_dispatch_frames for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same frame
 * bookkeeping and the same bits in the accumulation buffer as rt_dispatch_frames, but the call makes no
 * hipMalloc / hipFree: the scratch of the launch — the work-queue head and the per-frame staging planes — lives
 * in a static device array of RT_STATIC_STAGE_BYTES (+256) bytes that every GPU the library runs on holds
 * until the process ends.  The whole run is ONE launch, so n_frames x local_rows x w x 16 bytes must fit that
 * array (n_frames = 1 stages nothing and fits at any size); a bigger request fails with
 * RT_ERR_INVALID_ARGUMENT and a message naming both sizes, nothing is rendered.  RT_OPT_AOV must be off (its
 * buffer is allocated on demand).  All contexts on one GPU share that GPU's array, which keeps the last call's
 * contents (the call clears what must start at zero).  The library serialises them: every call makes its stream wait
 * (hipStreamWaitEvent on a per-GPU event created in rt_create — no allocation) for the static call before it on
 * that GPU, whichever context and stream issued it, so two contexts cannot corrupt each other's launch; they run
 * one after the other.  The exception is a call recorded under hipStreamBeginCapture: nothing can be waited for
 * inside a capture, so the owner of the graph orders its launches against other static calls on that GPU (and the
 * timing events are not recorded: rt_last_dispatch_ms does not see a captured call).  Calls from several host
 * threads on DIFFERENT contexts are safe; one context is for one thread, as everywhere in this API. */
#define RT_STATIC_STAGE_BYTES (32ull << 20)
int rt_dispatch_frames_static(rt_context *ctx, RtUniform *uniform, uint32_t n_frames);
/* Test hook, no counterpart in the reference: hipMalloc calls this library has made in this process so far (every
 * device allocation of the library is counted).  tests/test_zz_no_alloc.py asserts that it does not move across
 * rt_dispatch_frames_static. */
uint64_t rt_debug_alloc_count(void);
int rt_sync(rt_context *ctx);

/* ---------------------------------------------------------------- readback */
/* ≙ accumulation_buffer.download_async, src/ui.rs:82.  dst receives
 * local_rows*w elements of 16 B (xyz + pad).  Synchronises. */
int rt_accum_download(rt_context *ctx, float *dst);
/* Capture ≙ src/ui.rs:82-93 without the 16-byte-per-pixel download: the `(x*255) as u8` conversion runs on the
 * device and dst receives local_rows*w*3 bytes, RGB8, row 0 = top.  Same bytes as rt_accum_download followed by
 * rt_host_capture_rgb8.  Synchronises. */
int rt_capture_rgb8(rt_context *ctx, uint8_t *dst);
/* local_rows*w RtHitId elements; requires RT_OPT_AOV=1 before the dispatch. */
int rt_aov_download(rt_context *ctx, RtHitId *dst);
int rt_stats(rt_context *ctx, RtStats *out);
int rt_stats_reset(rt_context *ctx);
/* Device time of the last rt_dispatch/rt_dispatch_frames call in milliseconds,
 * measured with HIP events on the context's stream.  Synchronises. */
int rt_last_dispatch_ms(rt_context *ctx, float *ms);

/* ------------------------------------------------------- several GPUs, one host object */
/* ≙ SURVEY §8(b) `create(device_ids[], n)`: the reference has ONE context that owns every resource of the pass
 * (Gpu::builder()…build() and the bindings, src/main.rs:27-59); an rt_group is that object for n GPUs of one node, so
 * that the Rust host reaches all of them through this FFI from its one thread.  Shard i (bands i, i+n, … of 8 rows,
 * rt_set_shard) renders on device_ids[i]; the same device may be named more than once (rehearsal on one GPU).
 *   - rt_group_scene_upload / rt_group_spheres_upload validate, build and repack the scene ONCE on the host (one host
 *     copy of the inputs, one BVH) and upload the result to every device;
 *   - rt_group_models_update ≙ App::upload_models (src/app.rs:46-62): every device refits its own copy (or, when a
 *     rebuild is needed, the host rebuilds once and every device receives the result);
 *   - rt_group_dispatch / rt_group_dispatch_frames launch every shard on its own device and stream and return at
 *     once (the devices render concurrently); *uniform advances exactly as in rt_dispatch_frames;
 *   - rt_group_accum_download / rt_group_capture_rgb8 gather the shards' bands on device_ids[0] (peer copies, xGMI),
 *     de-interleave them there in one kernel and hand back the WHOLE w x h image (16 B or 3 B per pixel): the same
 *     bytes as a one-GPU rt_accum_download / rt_capture_rgb8;
 *   - rt_group_stats sums the device-side counters over the shards (static figures are the first device's);
 *   - rt_group_context gives shard i's rt_context for per-device calls (rt_last_dispatch_ms, rt_set_stream, …). */
typedef struct rt_group rt_group;
int rt_group_create(const int *device_ids, int n_devices, rt_group **out);
int rt_group_destroy(rt_group *g);
int rt_group_size(const rt_group *g);
/* How shard i's bands reach device_ids[0] in a download: status[i] for i < rt_group_size (n = room in status[]).
 * The gather is correct in every case; without peer access HIP stages the copy through host memory. */
#define RT_PEER_SAME_DEVICE   0   /* shard i renders on device_ids[0] itself */
#define RT_PEER_ENABLED       1   /* hipDeviceEnablePeerAccess succeeded: GPU-to-GPU copy over xGMI */
#define RT_PEER_UNAVAILABLE   2   /* hipDeviceCanAccessPeer says no: host-staged copy */
#define RT_PEER_ENABLE_FAILED 3   /* peer access exists but could not be enabled: host-staged copy */
int rt_group_peer_access(const rt_group *g, int *status, int n);
int rt_group_context(rt_group *g, int shard, rt_context **out);
int rt_group_set_option(rt_group *g, int key, int64_t value);
int rt_group_scene_upload(rt_group *g,
                          const RtVertex *vertices, uint64_t n_vertices,
                          const uint32_t *indices, uint64_t n_indices,
                          const RtGeometryPrimitive *primitives, uint32_t n_primitives,
                          const float *transforms,
                          const RtGpuModel *models, uint32_t n_models);
int rt_group_spheres_upload(rt_group *g, const RtSphere *spheres, uint64_t n_spheres);
int rt_group_textures_upload(rt_group *g, const RtTexture *textures, uint32_t n_textures);
int rt_group_models_update(rt_group *g, const RtGpuModel *models, uint32_t n_models, const float *transforms);
int rt_group_accum_resize(rt_group *g, uint32_t w, uint32_t h);
int rt_group_dispatch(rt_group *g, const RtUniform *uniform);
int rt_group_dispatch_frames(rt_group *g, RtUniform *uniform, uint32_t n_frames);
/* rt_dispatch_bands on every shard: each one renders ITS bands [band_begin, band_end) (numbered from 0 per shard: shard s of n holds image
 * bands s, s + n, ...; a shard with fewer bands than band_begin renders nothing), so a frame in k pieces is k slices of every device's share. */
int rt_group_dispatch_bands(rt_group *g, const RtUniform *uniform, uint32_t band_begin, uint32_t band_end);
int rt_group_sync(rt_group *g);
/* The same gather without the copy to the host: renders finished, bands gathered and de-interleaved on device_ids[0], and
 * *device_image = the w x h x 16-byte image THERE (owned by the group, valid until the next gather / resize / destroy).
 * What a host that keeps the image on the GPU (present pass, encoder) calls; also what bench.py --mode group times. */
int rt_group_accum_gather(rt_group *g, void **device_image);
int rt_group_accum_download(rt_group *g, float *dst);
int rt_group_capture_rgb8(rt_group *g, uint8_t *dst);
int rt_group_stats(rt_group *g, RtStats *out);
int rt_group_stats_reset(rt_group *g);

/* ------------------------------------------------- host side (no GPU use) */
/* Host mirror of the reference's Rust host for this path.  These run on the
 * CPU only and are what the Rust host keeps doing in the reference. */

/* ≙ Scene::load, src/scene.rs:95-207 (+ tobj 4.0.3 triangulate/single_index
 * semantics, src/misc.rs:49-75 for Pm/Pr/Ke). */
typedef struct rt_host_scene rt_host_scene;
int rt_host_scene_load(const char *obj_path, rt_host_scene **out);
void rt_host_scene_free(rt_host_scene *s);
uint64_t rt_host_scene_vertex_count(const rt_host_scene *s);
uint64_t rt_host_scene_index_count(const rt_host_scene *s);
uint32_t rt_host_scene_model_count(const rt_host_scene *s);
const RtVertex *rt_host_scene_vertices(const rt_host_scene *s);
const uint32_t *rt_host_scene_indices(const rt_host_scene *s);
const RtGeometryPrimitive *rt_host_scene_primitives(const rt_host_scene *s);
const RtGpuModel *rt_host_scene_models(const rt_host_scene *s);
const char *rt_host_scene_model_name(const rt_host_scene *s, uint32_t i);
uint32_t rt_host_scene_texture_count(const rt_host_scene *s);
int rt_host_scene_texture(const rt_host_scene *s, uint32_t i, RtTexture *out);
/* ≙ scene.finish(&gpu), src/main.rs:39 (scene arrays + textures) */
int rt_host_scene_finish(const rt_host_scene *s, rt_context *ctx);

/* ≙ the Uniform literal at src/main.rs:73-84 + Camera::default,
 * src/camera.rs:112-123 */
void rt_host_uniform_default(RtUniform *u);

/* ≙ struct App's frame state, src/app.rs:20-38 (pipelines and buffers live in
 * rt_context): what App::render needs between frames. */
typedef struct RtAppState {
    RtUniform uniform;           /* src/app.rs:25 */
    uint32_t inner_size[2];      /* gcx.window.inner_size(), src/app.rs:83 */
    uint32_t last_window[2];     /* src/app.rs:35 */
    double   last_invalidation;  /* seconds; src/app.rs:34 (an Instant there) */
    uint8_t  accumulate;         /* src/app.rs:36 */
    uint8_t  screen_fraction;    /* src/app.rs:37 */
    uint8_t  _pad[6];
} RtAppState;
void rt_host_app_init(RtAppState *app, uint32_t inner_width, uint32_t inner_height, double now);
void rt_host_app_invalidate(RtAppState *app, double now);
int rt_host_app_set_params(RtAppState *app, const RtUniform *edited, double now);
int rt_host_app_begin_frame(RtAppState *app, double now);
/* ≙ App::render, src/app.rs:74-102, against a context: begin_frame, re-create the
 * zeroed accumulation buffer when the window changed, dispatch.  Asynchronous. */
int rt_app_render(rt_context *ctx, RtAppState *app, double now);

/* Capture ≙ src/ui.rs:82-93: (x*255) as u8 per channel (truncating,
 * saturating), RGB8.  src = n_pixels 16-B elements; dst = n_pixels*3 bytes. */
void rt_host_capture_rgb8(const float *src, uint64_t n_pixels, uint8_t *dst);
/* Writes an RGB8 PNG (stored/uncompressed deflate). */
int rt_host_write_png(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h);

/* BVH build exposed for CPU-side inspection/tests: builds the same tree
 * rt_scene_upload builds and reports its shape. */
typedef struct RtBvhInfo {
    uint64_t n_triangles;
    uint64_t n_nodes;
    uint64_t n_leaves;
    uint32_t max_depth;
    uint32_t max_leaf_size;
    double   sah_cost;
    double   build_seconds;
} RtBvhInfo;
int rt_host_bvh_info(const RtVertex *vertices, uint64_t n_vertices,
                     const uint32_t *indices, uint64_t n_indices,
                     const RtGeometryPrimitive *primitives, uint32_t n_primitives,
                     const float *transforms, RtBvhInfo *out);

#ifdef __cplusplus
} /* extern "C" */

#define RT_LAYOUT_ASSERT(T, field, off) \
    static_assert(offsetof(T, field) == (off), #T "." #field " offset")
static_assert(sizeof(RtCamera) == 32, "Camera size");
RT_LAYOUT_ASSERT(RtCamera, pitch, 12);
RT_LAYOUT_ASSERT(RtCamera, yaw, 16);
RT_LAYOUT_ASSERT(RtCamera, fov, 20);
RT_LAYOUT_ASSERT(RtCamera, aspect, 24);
static_assert(sizeof(RtUniform) == 80, "Uniform size");
RT_LAYOUT_ASSERT(RtUniform, camera, 16);
RT_LAYOUT_ASSERT(RtUniform, frame, 48);
RT_LAYOUT_ASSERT(RtUniform, accumulation_frame, 52);
RT_LAYOUT_ASSERT(RtUniform, flags, 56);
RT_LAYOUT_ASSERT(RtUniform, exposure, 60);
RT_LAYOUT_ASSERT(RtUniform, environment, 64);
RT_LAYOUT_ASSERT(RtUniform, max_bounces, 68);
RT_LAYOUT_ASSERT(RtUniform, samples, 72);
static_assert(sizeof(RtMetalMaterial) == 80, "MetalMaterial size");
RT_LAYOUT_ASSERT(RtMetalMaterial, specular_color, 16);
RT_LAYOUT_ASSERT(RtMetalMaterial, specular_probability, 28);
RT_LAYOUT_ASSERT(RtMetalMaterial, roughness, 32);
RT_LAYOUT_ASSERT(RtMetalMaterial, emission_color, 48);
RT_LAYOUT_ASSERT(RtMetalMaterial, emission_strength, 60);
RT_LAYOUT_ASSERT(RtMetalMaterial, diffuse_texture, 64);
RT_LAYOUT_ASSERT(RtMetalMaterial, normal_texture, 68);
static_assert(sizeof(RtMaterial) == 112, "Material size");
RT_LAYOUT_ASSERT(RtMaterial, metal, 16);
RT_LAYOUT_ASSERT(RtMaterial, refractive_index, 96);
static_assert(sizeof(RtGpuModel) == 128, "GpuModel size");
RT_LAYOUT_ASSERT(RtGpuModel, vertex_start, 112);
RT_LAYOUT_ASSERT(RtGpuModel, index_start, 116);
static_assert(sizeof(RtVertex) == 48, "Vertex size");
RT_LAYOUT_ASSERT(RtVertex, normal, 16);
RT_LAYOUT_ASSERT(RtVertex, uv, 32);
static_assert(sizeof(RtGeometryPrimitive) == 24, "GeometryPrimitive size");
static_assert(sizeof(RtSphere) == 32, "Sphere size");
static_assert(sizeof(RtHitId) == 8, "HitId size");
#undef RT_LAYOUT_ASSERT
#endif /* __cplusplus */

#endif /* RT_ABI_H */
