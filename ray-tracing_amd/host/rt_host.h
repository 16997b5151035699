// rt_host.h — host-side (CPU) half of the backend: what the reference's Rust
// host does for the hot path (scene.rs, camera.rs, app.rs, ui.rs Capture),
// written in C++ because this image has no Rust toolchain.  No HIP here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt_host {

// one decoded texture: RGBA8, rows in file order (top row first)
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};

// ≙ struct Scene, /root/reference/src/scene.rs:20-27
struct Scene {
    std::vector<Image> textures;
    std::vector<RtVertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<RtGeometryPrimitive> primitives;
    std::vector<RtGpuModel> models;
    std::vector<std::string> names;
};

// ≙ Scene::load, src/scene.rs:95-207
bool load_obj(const std::string &path, Scene &scene, std::string &err);
// ≙ image::load(...).into_rgba8(), src/scene.rs:157-170 (PNG and binary PPM)
bool load_image_rgba8(const std::string &path, Image &img, std::string &err);
// ≙ strip_flags, src/scene.rs:210-223
std::string strip_flags(const std::string &path);

// ---------------------------------------------------------------- BVH build
// The reference hands BVH construction to the Vulkan driver
// (create_acceleration_structure, src/scene.rs:56-64); BASELINE.json's
// north_star assigns it to the host.  Binned-SAH BVH2, leaves of <= 4
// primitives, depth-bounded so the kernel's traversal stack has a known size.
struct Aabb {
    float lo[3], hi[3];
};

struct Bvh2Node {
    Aabb box;
    uint32_t left;   // inner: children at left, left+1.  leaf: first slot in `order`
    uint32_t count;  // 0 = inner, else leaf primitive count
};

struct Bvh2 {
    std::vector<Bvh2Node> nodes;   // nodes[0] = root
    std::vector<uint32_t> order;   // slot -> primitive id
    uint32_t max_depth = 0;        // root = depth 0
    uint32_t max_leaf = 0;
    uint64_t n_leaves = 0;
    double sah_cost = 0.0;
    double build_seconds = 0.0;
};

struct BvhBuildOptions {
    uint32_t max_leaf = 4;
    uint32_t max_depth = 48;
    uint32_t bins = 16;
    uint32_t threads = 0;  // 0 = min(hardware_concurrency, 16)
    float traversal_cost = 1.0f;   // SAH: cost of one node step in units of one primitive test
};

// boxes[i] bounds primitive i.  n may be 0 (empty tree: one empty leaf).
void build_bvh2(const Aabb *boxes, uint64_t n, const BvhBuildOptions &opt, Bvh2 &out);
// ≙ acceleration_structure.update(), src/app.rs:61: keep the tree's topology, recompute every box
// bottom-up from the primitives' new bounds (children always follow their parent in `nodes`).
void refit_bvh2(const Aabb *boxes, Bvh2 &bvh);

// Triangle corner positions for the BLAS: index triples of each
// GeometryPrimitive with that geometry's 3x4 transform applied
// (src/scene.rs:51-64; transforms == nullptr → identity).
// corners: 9 floats per triangle (a, b, c); ids: (geometry, primitive) pairs.
bool gather_triangles(const RtVertex *vertices, uint64_t n_vertices, const uint32_t *indices, uint64_t n_indices,
                      const RtGeometryPrimitive *prims, uint32_t n_prims, const float *transforms,
                      std::vector<float> &corners, std::vector<uint32_t> &ids, std::string &err);

// ------------------------------------------------------------------ capture
// ≙ src/ui.rs:82-93
void capture_rgb8(const float *accum16, uint64_t n_pixels, uint8_t *rgb);
bool write_png_rgb8(const std::string &path, const uint8_t *rgb, uint32_t w, uint32_t h, std::string &err);

// ≙ src/main.rs:73-84 + src/camera.rs:112-123
void uniform_default(RtUniform &u);

}  // namespace rt_host

struct rt_host_scene {
    rt_host::Scene scene;
};
