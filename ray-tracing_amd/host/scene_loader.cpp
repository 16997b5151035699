// scene_loader.cpp — host-side scene loading, the C++ mirror of
// /root/reference/src/scene.rs:95-207 (Scene::load) and src/misc.rs:49-75.
//
// The reference delegates OBJ/MTL parsing to tobj 4.0.3 with
// `triangulate: true, single_index: true` (src/scene.rs:100-107); those
// semantics are implemented here directly on a memory-mapped-style single pass:
// one model per `o`/`g` run (or material switch), fan triangulation in file
// order, per-model (v,vt,vn) de-duplication in first-use order, model-local
// indices.  Output arrays use the WGSL layouts of include/rt_abi.h.
#include "rt_host.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace rt_host {

namespace {

struct MtlMaterial {
    std::string name;
    float diffuse[3] = {0, 0, 0};
    float specular[3] = {0, 0, 0};
    std::string diffuse_texture, normal_texture;   // tobj Material::diffuse_texture / normal_texture (empty = None)
    std::unordered_map<std::string, std::string> unknown;  // tobj Material::unknown_param
};

struct FaceVertex { int64_t v, vt, vn; };

struct KeyHash {
    size_t operator()(const FaceVertex &k) const {
        uint64_t h = (uint64_t)k.v * 0x9E3779B97F4A7C15ull;
        h ^= ((uint64_t)k.vt + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
        h ^= ((uint64_t)k.vn + 0x165667B1ull) * 0x9E3779B185EBCA87ull + (h << 6) + (h >> 2);
        return (size_t)h;
    }
};
struct KeyEq {
    bool operator()(const FaceVertex &a, const FaceVertex &b) const { return a.v == b.v && a.vt == b.vt && a.vn == b.vn; }
};

inline const char *skip_ws(const char *p) { while (*p == ' ' || *p == '\t') ++p; return p; }
inline std::string trim(const std::string &s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
// Rust `parse::<f32>()`: correctly rounded decimal → binary32, like strtof.
inline bool parse_f32(const char *&p, float &out) {
    p = skip_ws(p);
    char *end = nullptr;
    out = strtof(p, &end);
    if (end == p) return false;
    p = end;
    return true;
}

bool load_mtl(const std::string &path, std::vector<MtlMaterial> &mats,
              std::unordered_map<std::string, int> &by_name, std::string &err) {
    std::ifstream in(path);
    if (!in) { err = "cannot open material library " + path; return false; }
    std::string line;
    MtlMaterial *cur = nullptr;
    while (std::getline(in, line)) {
        line = trim(line);
        if (line.empty() || line[0] == '#') continue;
        size_t sp = line.find_first_of(" \t");
        std::string key = line.substr(0, sp);
        std::string rest = sp == std::string::npos ? std::string() : trim(line.substr(sp));
        if (key == "newmtl") {
            by_name[rest] = (int)mats.size();
            mats.emplace_back();
            cur = &mats.back();
            cur->name = rest;
        } else if (!cur) {
            continue;
        } else if (key == "Kd" || key == "Ks") {
            float *dst = key == "Kd" ? cur->diffuse : cur->specular;
            const char *p = rest.c_str();
            for (int k = 0; k < 3; k++) if (!parse_f32(p, dst[k])) { err = "bad " + key + " in " + path; return false; }
        } else if (key == "map_Kd") {
            cur->diffuse_texture = rest;
        } else if (key == "map_Bump" || key == "map_bump" || key == "bump") {
            cur->normal_texture = rest;
        } else if (key == "Ka" || key == "Ns" || key == "Ni" || key == "d" || key == "illum" ||
                   key == "map_Ka" || key == "map_Ks" || key == "map_Ns" || key == "map_d") {
            // parsed by tobj into fields the reference never reads
        } else {
            cur->unknown[key] = rest;
        }
    }
    return true;
}

// src/misc.rs:53-59
float unknown_f32(const MtlMaterial &m, const char *key) {
    auto it = m.unknown.find(key);
    if (it == m.unknown.end()) return 0.0f;
    const char *p = it->second.c_str();
    char *end = nullptr;
    float v = strtof(p, &end);
    if (end == p || *skip_ws(end) != '\0') return 0.0f;  // parse::<f32>().ok() fails on trailing text
    return v;
}
// src/misc.rs:62-75
void unknown_vec3(const MtlMaterial &m, const char *key, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    auto it = m.unknown.find(key);
    if (it == m.unknown.end()) return;
    const char *p = it->second.c_str();
    for (int k = 0; k < 3; k++) if (!parse_f32(p, out[k])) break;
}

}  // namespace

bool load_obj(const std::string &path, Scene &scene, std::string &err) {
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "cannot open " + path; return false; }
    std::string dir;
    {
        size_t slash = path.find_last_of('/');
        dir = slash == std::string::npos ? std::string(".") : path.substr(0, slash);
    }

    std::vector<float> pos, nrm, tex;
    std::vector<MtlMaterial> mats;
    std::unordered_map<std::string, int> mat_by_name;

    std::string cur_name = "unnamed_object";
    int cur_mat = -1;
    std::vector<FaceVertex> face_verts;    // corners of the faces of the open model
    std::vector<uint32_t> face_sizes;

    scene = Scene();

    auto flush = [&]() -> bool {
        if (face_sizes.empty()) return true;
        // ≙ one `model` of the loop at src/scene.rs:111-204
        const uint32_t first_vertex = (uint32_t)scene.vertices.size();
        const uint32_t first_index = (uint32_t)scene.indices.size();
        std::unordered_map<FaceVertex, uint32_t, KeyHash, KeyEq> index_map;
        index_map.reserve(face_verts.size());
        bool all_have_uv = true;
        std::vector<FaceVertex> uniq;
        auto add = [&](const FaceVertex &fv) {
            auto it = index_map.find(fv);
            uint32_t id;
            if (it == index_map.end()) {
                id = (uint32_t)uniq.size();
                index_map.emplace(fv, id);
                uniq.push_back(fv);
                if (fv.vt < 0) all_have_uv = false;
            } else id = it->second;
            scene.indices.push_back(id);
        };
        size_t base = 0;
        for (uint32_t n : face_sizes) {
            if (n < 3) { err = "points/lines are not supported"; return false; }
            for (uint32_t k = 2; k < n; k++) {  // fan: (0, k-1, k)
                add(face_verts[base]);
                add(face_verts[base + k - 1]);
                add(face_verts[base + k]);
            }
            base += n;
        }
        for (const FaceVertex &fv : uniq) {
            if (fv.vn < 0) { err = "vertex without a normal in object " + cur_name; return false; }
            RtVertex v;
            memset(&v, 0, sizeof v);
            memcpy(v.position, &pos[3 * (size_t)fv.v], 12);
            memcpy(v.normal, &nrm[3 * (size_t)fv.vn], 12);
            if (all_have_uv) memcpy(v.uv, &tex[2 * (size_t)fv.vt], 8);  // else (0,0): src/scene.rs:127-130
            scene.vertices.push_back(v);
        }
        RtGeometryPrimitive prim;                                        // src/scene.rs:141-147
        prim.first_vertex = first_vertex;
        prim.vertex_count = (uint32_t)scene.vertices.size() - first_vertex;
        prim.first_index = first_index;
        prim.index_count = (uint32_t)scene.indices.size() - first_index;
        prim.transformation_offset = scene.primitives.size();
        scene.primitives.push_back(prim);

        if (cur_mat < 0) { err = "object `" + cur_name + "` has no material"; return false; }  // unwrap(), :149
        const MtlMaterial &m = mats[(size_t)cur_mat];
        RtGpuModel model;
        memset(&model, 0, sizeof model);
        model.material.tag = 0;                                          // Material::metal, src/types.rs:102-110
        memcpy(model.material.metal.diffuse_color, m.diffuse, 12);       // :185
        memcpy(model.material.metal.specular_color, m.specular, 12);     // :186
        model.material.metal.specular_probability = unknown_f32(m, "Pm");  // :153
        model.material.metal.roughness = unknown_f32(m, "Pr");             // :154
        float ke[3];
        unknown_vec3(m, "Ke", ke);                                       // :155
        // nalgebra magnitude(): sqrt(x*x + y*y + z*z) in f32; try_normalize(0.0)
        volatile float n2 = (ke[0] * ke[0] + ke[1] * ke[1]) + ke[2] * ke[2];
        float norm = sqrtf(n2);
        if (norm > 0.0f) for (int k = 0; k < 3; k++) model.material.metal.emission_color[k] = ke[k] / norm;  // :191
        model.material.metal.emission_strength = norm;                   // :192
        // load_texture, src/scene.rs:157-178: 1-based id of the pushed image, 0 = none; not de-duplicated
        auto load_texture = [&](const std::string &rel, uint32_t &id) -> bool {
            id = 0;
            if (rel.empty()) return true;
            Image img;
            if (!load_image_rgba8(dir + "/" + strip_flags(rel), img, err)) return false;
            scene.textures.push_back(std::move(img));
            id = (uint32_t)scene.textures.size();
            return true;
        };
        if (!load_texture(m.diffuse_texture, model.material.metal.diffuse_texture)) return false;   // :177
        if (!load_texture(m.normal_texture, model.material.metal.normal_texture)) return false;     // :178
        model.material.refractive_index = 0.0f;                          // DielectricMaterial::default()
        model.vertex_start = first_vertex;                               // :197
        model.index_start = first_index;                                 // :198
        scene.models.push_back(model);
        scene.names.push_back(cur_name);
        face_verts.clear();
        face_sizes.clear();
        return true;
    };

    std::string line;
    size_t line_no = 0;
    while (std::getline(in, line)) {
        ++line_no;
        const char *p = skip_ws(line.c_str());
        if (*p == '\0' || *p == '#' || *p == '\r') continue;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            const char *q = p + 1;
            float f[3];
            for (int k = 0; k < 3; k++) if (!parse_f32(q, f[k])) { err = "bad v at line " + std::to_string(line_no); return false; }
            pos.insert(pos.end(), f, f + 3);
        } else if (p[0] == 'v' && p[1] == 'n') {
            const char *q = p + 2;
            float f[3];
            for (int k = 0; k < 3; k++) if (!parse_f32(q, f[k])) { err = "bad vn at line " + std::to_string(line_no); return false; }
            nrm.insert(nrm.end(), f, f + 3);
        } else if (p[0] == 'v' && p[1] == 't') {
            const char *q = p + 2;
            float f[2];
            for (int k = 0; k < 2; k++) if (!parse_f32(q, f[k])) { err = "bad vt at line " + std::to_string(line_no); return false; }
            tex.insert(tex.end(), f, f + 2);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            const char *q = p + 1;
            uint32_t n = 0;
            for (;;) {
                q = skip_ws(q);
                if (*q == '\0' || *q == '\r' || *q == '\n') break;
                FaceVertex fv{-1, -1, -1};
                int64_t *slots[3] = {&fv.v, &fv.vt, &fv.vn};
                const int64_t counts[3] = {(int64_t)pos.size() / 3, (int64_t)tex.size() / 2, (int64_t)nrm.size() / 3};
                for (int s = 0; s < 3; s++) {
                    if (*q != '/' && *q != ' ' && *q != '\t' && *q != '\0' && *q != '\r') {
                        char *end = nullptr;
                        long long i = strtoll(q, &end, 10);
                        if (end == q) { err = "bad face at line " + std::to_string(line_no); return false; }
                        q = end;
                        *slots[s] = i > 0 ? i - 1 : counts[s] + i;
                        if (*slots[s] < 0 || *slots[s] >= counts[s]) { err = "face index out of range at line " + std::to_string(line_no); return false; }
                    }
                    if (*q == '/') ++q; else break;
                }
                if (fv.v < 0) { err = "bad face at line " + std::to_string(line_no); return false; }
                face_verts.push_back(fv);
                ++n;
            }
            face_sizes.push_back(n);
        } else if ((p[0] == 'o' || p[0] == 'g') && (p[1] == ' ' || p[1] == '\t' || p[1] == '\0' || p[1] == '\r')) {
            if (!flush()) return false;
            cur_name = trim(std::string(p + 1));
            if (cur_name.empty()) cur_name = "unnamed_object";
        } else if (strncmp(p, "usemtl", 6) == 0) {
            std::string name = trim(std::string(p + 6));
            auto it = mat_by_name.find(name);
            int new_mat = it == mat_by_name.end() ? -1 : it->second;
            if (new_mat != cur_mat && !face_sizes.empty()) if (!flush()) return false;
            cur_mat = new_mat;
        } else if (strncmp(p, "mtllib", 6) == 0) {
            std::string name = trim(std::string(p + 6));
            std::vector<MtlMaterial> m;
            std::unordered_map<std::string, int> byn;
            if (!load_mtl(dir + "/" + name, m, byn, err)) return false;
            int base = (int)mats.size();
            for (auto &kv : byn) mat_by_name[kv.first] = base + kv.second;
            mats.insert(mats.end(), m.begin(), m.end());
        }
        // s (smoothing groups), l, p and anything else: ignored
    }
    return flush();
}

}  // namespace rt_host
