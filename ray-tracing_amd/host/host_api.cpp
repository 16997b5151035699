// host_api.cpp — C entry points of the host half (no GPU use): the calls a
// Rust host would keep making on the CPU (include/rt_abi.h, "host side").
#include <cmath>
#include <cstring>
#include <string>

#include "rt_host.h"

void rt_set_last_error(const std::string &s);

extern "C" {

int rt_host_scene_load(const char *obj_path, rt_host_scene **out) {
    if (!obj_path || !out) { rt_set_last_error("null argument"); return RT_ERR_INVALID_ARGUMENT; }
    *out = nullptr;
    rt_host_scene *s = new rt_host_scene();
    std::string err;
    if (!rt_host::load_obj(obj_path, s->scene, err)) {
        delete s;
        rt_set_last_error(err);
        return err.rfind("cannot open", 0) == 0 ? RT_ERR_IO : RT_ERR_PARSE;
    }
    *out = s;
    return RT_OK;
}
void rt_host_scene_free(rt_host_scene *s) { delete s; }
uint64_t rt_host_scene_vertex_count(const rt_host_scene *s) { return s ? s->scene.vertices.size() : 0; }
uint64_t rt_host_scene_index_count(const rt_host_scene *s) { return s ? s->scene.indices.size() : 0; }
uint32_t rt_host_scene_model_count(const rt_host_scene *s) { return s ? (uint32_t)s->scene.models.size() : 0; }
const RtVertex *rt_host_scene_vertices(const rt_host_scene *s) { return s ? s->scene.vertices.data() : nullptr; }
const uint32_t *rt_host_scene_indices(const rt_host_scene *s) { return s ? s->scene.indices.data() : nullptr; }
const RtGeometryPrimitive *rt_host_scene_primitives(const rt_host_scene *s) { return s ? s->scene.primitives.data() : nullptr; }
const RtGpuModel *rt_host_scene_models(const rt_host_scene *s) { return s ? s->scene.models.data() : nullptr; }
const char *rt_host_scene_model_name(const rt_host_scene *s, uint32_t i) {
    return (s && i < s->scene.names.size()) ? s->scene.names[i].c_str() : nullptr;
}

uint32_t rt_host_scene_texture_count(const rt_host_scene *s) { return s ? (uint32_t)s->scene.textures.size() : 0; }
int rt_host_scene_texture(const rt_host_scene *s, uint32_t i, RtTexture *out) {
    if (!s || !out || i >= s->scene.textures.size()) { rt_set_last_error("texture index out of range"); return RT_ERR_INVALID_ARGUMENT; }
    const rt_host::Image &im = s->scene.textures[i];
    out->rgba8 = im.rgba.data();
    out->width = im.width;
    out->height = im.height;
    return RT_OK;
}

void rt_host_uniform_default(RtUniform *u) { if (u) rt_host::uniform_default(*u); }

void rt_host_capture_rgb8(const float *src, uint64_t n_pixels, uint8_t *dst) { rt_host::capture_rgb8(src, n_pixels, dst); }

int rt_host_write_png(const char *path, const uint8_t *rgb, uint32_t w, uint32_t h) {
    if (!path || !rgb) { rt_set_last_error("null argument"); return RT_ERR_INVALID_ARGUMENT; }
    std::string err;
    if (!rt_host::write_png_rgb8(path, rgb, w, h, err)) { rt_set_last_error(err); return RT_ERR_IO; }
    return RT_OK;
}

int rt_host_bvh_info(const RtVertex *vertices, uint64_t n_vertices, const uint32_t *indices, uint64_t n_indices,
                     const RtGeometryPrimitive *primitives, uint32_t n_primitives, const float *transforms,
                     RtBvhInfo *out) {
    if (!out) { rt_set_last_error("null argument"); return RT_ERR_INVALID_ARGUMENT; }
    std::string err;
    std::vector<float> corners;
    std::vector<uint32_t> ids;
    if (!rt_host::gather_triangles(vertices, n_vertices, indices, n_indices, primitives, n_primitives, transforms,
                                   corners, ids, err)) {
        rt_set_last_error(err);
        return RT_ERR_INVALID_ARGUMENT;
    }
    const uint64_t n = ids.size() / 2;
    std::vector<rt_host::Aabb> boxes(n);
    for (uint64_t t = 0; t < n; t++)
        for (int k = 0; k < 3; k++) {
            const float *c = &corners[9 * t];
            boxes[t].lo[k] = fminf(c[k], fminf(c[3 + k], c[6 + k]));
            boxes[t].hi[k] = fmaxf(c[k], fmaxf(c[3 + k], c[6 + k]));
        }
    rt_host::Bvh2 bvh;
    rt_host::build_bvh2(boxes.data(), n, rt_host::BvhBuildOptions(), bvh);
    // structural self-check: every primitive in exactly one leaf, child boxes inside parents
    std::vector<uint8_t> seen(n, 0);
    for (const auto &nd : bvh.nodes) {
        if (nd.count) {
            for (uint32_t i = 0; i < nd.count; i++) {
                uint32_t id = bvh.order[nd.left + i];
                if (id >= n || seen[id]) { rt_set_last_error("bvh: primitive missing or duplicated"); return RT_ERR_INVALID_ARGUMENT; }
                seen[id] = 1;
                for (int k = 0; k < 3; k++)
                    if (boxes[id].lo[k] < nd.box.lo[k] || boxes[id].hi[k] > nd.box.hi[k]) { rt_set_last_error("bvh: leaf box does not bound its primitive"); return RT_ERR_INVALID_ARGUMENT; }
            }
        } else if (n > 0) {
            for (int c = 0; c < 2; c++)
                for (int k = 0; k < 3; k++)
                    if (bvh.nodes[nd.left + c].box.lo[k] < nd.box.lo[k] || bvh.nodes[nd.left + c].box.hi[k] > nd.box.hi[k]) { rt_set_last_error("bvh: child box outside parent"); return RT_ERR_INVALID_ARGUMENT; }
        }
    }
    for (uint64_t i = 0; i < n; i++) if (!seen[i]) { rt_set_last_error("bvh: primitive not in any leaf"); return RT_ERR_INVALID_ARGUMENT; }
    out->n_triangles = n;
    out->n_nodes = bvh.nodes.size();
    out->n_leaves = bvh.n_leaves;
    out->max_depth = bvh.max_depth;
    out->max_leaf_size = bvh.max_leaf;
    out->sah_cost = bvh.sah_cost;
    out->build_seconds = bvh.build_seconds;
    return RT_OK;
}

}  // extern "C"
