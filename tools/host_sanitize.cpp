// Host half under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on this pool).
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude \
//       tools/host_sanitize.cpp ray-tracing_amd/host/{scene_loader,bvh_builder,capture,image_decode}.cpp -lpthread -lz -o /tmp/host_sanitize
//   /tmp/host_sanitize scenes/cornell-box.obj scenes/teapot.obj scenes/teapot-circle.obj
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../ray-tracing_amd/host/rt_host.h"

int main(int argc, char **argv) {
    int bad = 0;
    for (int i = 1; i < argc; i++) {
        rt_host::Scene sc;
        std::string err;
        const size_t len = strlen(argv[i]);
        if (len > 4 && (!strcmp(argv[i] + len - 4, ".png") || !strcmp(argv[i] + len - 4, ".ppm"))) {
            // an image file, possibly damaged on purpose: decoding may fail, it must not fault or over-allocate
            rt_host::Image img;
            const bool ok = rt_host::load_image_rgba8(argv[i], img, err);
            if (ok && img.rgba.size() != (size_t)img.width * img.height * 4) { printf("%s: size mismatch\n", argv[i]); bad++; }
            continue;
        }
        if (len > 4 && !strcmp(argv[i] + len - 4, ".bad")) {       // an OBJ file damaged on purpose: may be refused, must not fault
            (void)rt_host::load_obj(argv[i], sc, err);
            continue;
        }
        if (!rt_host::load_obj(argv[i], sc, err)) { printf("%s: %s\n", argv[i], err.c_str()); bad++; continue; }
        std::vector<float> corners;
        std::vector<uint32_t> ids;
        if (!rt_host::gather_triangles(sc.vertices.data(), sc.vertices.size(), sc.indices.data(), sc.indices.size(), sc.primitives.data(),
                                       (uint32_t)sc.primitives.size(), nullptr, corners, ids, err)) { printf("%s: %s\n", argv[i], err.c_str()); bad++; continue; }
        const size_t n = ids.size() / 2;
        std::vector<rt_host::Aabb> boxes(n);
        for (size_t t = 0; t < n; t++)
            for (int k = 0; k < 3; k++) {
                const float *c = &corners[9 * t];
                boxes[t].lo[k] = std::min(c[k], std::min(c[3 + k], c[6 + k]));
                boxes[t].hi[k] = std::max(c[k], std::max(c[3 + k], c[6 + k]));
            }
        for (uint32_t max_leaf : {1u, 4u, 8u})
            for (uint32_t depth : {0u, 14u, 48u}) {
                rt_host::BvhBuildOptions opt;
                opt.max_leaf = max_leaf;
                if (depth) opt.max_depth = depth;
                rt_host::Bvh2 bvh;
                rt_host::build_bvh2(boxes.data(), n, opt, bvh);
                rt_host::refit_bvh2(boxes.data(), bvh);
                printf("%s: %zu triangles, leaf<=%u depth cap %u -> %zu nodes, depth %u\n", argv[i], n, max_leaf, depth, bvh.nodes.size(), bvh.max_depth);
            }
    }
    // degenerate inputs of the builder
    {
        std::vector<rt_host::Aabb> same(5000);
        for (auto &b : same) for (int k = 0; k < 3; k++) { b.lo[k] = 1.0f; b.hi[k] = 2.0f; }
        rt_host::Bvh2 bvh;
        rt_host::build_bvh2(same.data(), same.size(), rt_host::BvhBuildOptions(), bvh);
        rt_host::build_bvh2(same.data(), 0, rt_host::BvhBuildOptions(), bvh);
        rt_host::build_bvh2(same.data(), 1, rt_host::BvhBuildOptions(), bvh);
    }
    // capture + PNG writer + PNG decoder round trip
    {
        const uint32_t w = 37, h = 23;
        std::vector<float> acc((size_t)w * h * 4);
        std::mt19937 rng(3);
        for (auto &v : acc) v = (float)(rng() % 2000) / 1000.0f - 0.5f;
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        rt_host::capture_rgb8(acc.data(), (uint64_t)w * h, rgb.data());
        std::string err;
        if (!rt_host::write_png_rgb8("/tmp/host_sanitize.png", rgb.data(), w, h, err)) { printf("png: %s\n", err.c_str()); bad++; }
        rt_host::Image img;
        if (!rt_host::load_image_rgba8("/tmp/host_sanitize.png", img, err)) { printf("decode: %s\n", err.c_str()); bad++; }
        else if (img.width != w || img.height != h || memcmp(img.rgba.data(), rgb.data(), 3) != 0) { printf("png round trip differs\n"); bad++; }
        rt_host::Image none;
        if (rt_host::load_image_rgba8("/tmp/does-not-exist.png", none, err)) bad++;
    }
    printf(bad ? "FAILED\n" : "host sanitize run ok\n");
    return bad;
}
