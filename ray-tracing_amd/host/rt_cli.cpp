// rt_cli.cpp — headless driver: the per-frame loop of the reference's App
// (/root/reference/src/app.rs:74-102) without a window, plus Capture
// (src/ui.rs:78-94).  The reference's own TODO.md:6 asks for this mode.
// Uses only the C ABI of include/rt_abi.h, like a Rust host would.
//
//   rt_cli --scene scenes/cornell-box.obj --size 256x256 --samples 16 --frames 1 --bounces 4 \
//          --pos 0,15,14 --yaw 4.712389 --pitch 0 --out out.png
//   rt_cli ... --devices 0,1,2,3      the same image from several GPUs through ONE rt_group (bands of 8 rows dealt round-robin,
//                                     one host BVH build, gather on the first device); a device may be named more than once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"

static int die(const char *what) {
    fprintf(stderr, "rt_cli: %s: %s\n", what, rt_last_error());
    return 1;
}

int main(int argc, char **argv) {
    std::string scene, out = "out.png";
    RtUniform u;
    rt_host_uniform_default(&u);               // src/main.rs:73-84
    uint32_t w = 256, h = 256, frames = 1;
    int device = 0, node_format = 0;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { fprintf(stderr, "rt_cli: %s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--out") out = next();
        else if (a == "--size") { if (sscanf(next(), "%ux%u", &w, &h) != 2) { fprintf(stderr, "rt_cli: --size WxH\n"); return 2; } }
        else if (a == "--samples") u.samples = (uint32_t)atoi(next());
        else if (a == "--frames") frames = (uint32_t)atoi(next());
        else if (a == "--bounces") u.max_bounces = (uint32_t)atoi(next());
        else if (a == "--pos") { if (sscanf(next(), "%f,%f,%f", &u.camera.position[0], &u.camera.position[1], &u.camera.position[2]) != 3) { fprintf(stderr, "rt_cli: --pos x,y,z\n"); return 2; } }
        else if (a == "--yaw") u.camera.yaw = (float)atof(next());
        else if (a == "--pitch") u.camera.pitch = (float)atof(next());
        else if (a == "--fov") u.camera.fov = (float)atof(next());
        else if (a == "--exposure") u.exposure = (float)atof(next());
        else if (a == "--environment") u.environment = (float)atof(next());
        else if (a == "--cull-backfaces") u.flags |= RT_FLAG_CULL_BACKFACES;
        else if (a == "--device") device = atoi(next());
        else if (a == "--node-format") node_format = atoi(next());       // RT_OPT_NODE_FORMAT (0 = by scene size)
        else if (a == "--devices") { for (const char *p = next(); *p;) { devices.push_back(atoi(p)); while (*p && *p != ',') p++; if (*p == ',') p++; } }
        else { fprintf(stderr, "rt_cli: unknown argument %s\n", a.c_str()); return 2; }
    }
    if (scene.empty()) { fprintf(stderr, "usage: rt_cli --scene file.obj [--size WxH] [--samples N] [--frames N] [--bounces N] [--pos x,y,z] [--yaw r] [--pitch r] [--out file.png]\n"); return 2; }

    rt_host_scene *hs = nullptr;
    if (rt_host_scene_load(scene.c_str(), &hs)) return die("load");      // ≙ scene.load(), src/main.rs:37
    u.window[0] = w; u.window[1] = h;
    u.camera.aspect = (float)w / (float)h;                               // src/camera.rs:56
    if (!devices.empty()) {
        // several GPUs behind one host object: the rt_group twin of every call below
        const rt_host_scene *s = hs;
        rt_group *g = nullptr;
        if (rt_group_create(devices.data(), (int)devices.size(), &g)) return die("group create");
        if (node_format && rt_group_set_option(g, RT_OPT_NODE_FORMAT, node_format)) return die("node format");
        if (rt_group_scene_upload(g, rt_host_scene_vertices(s), rt_host_scene_vertex_count(s), rt_host_scene_indices(s), rt_host_scene_index_count(s),
                                  rt_host_scene_primitives(s), rt_host_scene_model_count(s), nullptr, rt_host_scene_models(s), rt_host_scene_model_count(s)))
            return die("group scene upload");
        const uint32_t n_tex = rt_host_scene_texture_count(s);
        if (n_tex) {
            std::vector<RtTexture> tex(n_tex);
            for (uint32_t t = 0; t < n_tex; t++) if (rt_host_scene_texture(s, t, &tex[t])) return die("texture");
            if (rt_group_textures_upload(g, tex.data(), n_tex)) return die("group textures");
        }
        if (rt_group_accum_resize(g, w, h)) return die("group accum");
        if (rt_group_dispatch_frames(g, &u, frames)) return die("group dispatch");
        std::vector<uint8_t> rgb((size_t)w * h * 3);
        if (rt_group_capture_rgb8(g, rgb.data())) return die("group capture");
        RtStats st;
        rt_group_stats(g, &st);
        float ms = 0, worst = 0;
        for (int k = 0; k < rt_group_size(g); k++) { rt_context *c = nullptr; if (!rt_group_context(g, k, &c) && !rt_last_dispatch_ms(c, &ms) && ms > worst) worst = ms; }
        if (rt_host_write_png(out.c_str(), rgb.data(), w, h)) return die("png");
        printf("%s: %ux%u, %u spp x %u frames on %d devices, %llu rays, slowest shard %.3f ms (%.1f Mrays/s) -> %s\n", scene.c_str(), w, h, u.samples, frames,
               rt_group_size(g), (unsigned long long)st.rays, worst, st.rays / (worst * 1e3), out.c_str());
        rt_group_destroy(g);
        rt_host_scene_free(hs);
        return 0;
    }
    rt_context *ctx = nullptr;
    if (rt_create(device, &ctx)) return die("create");
    if (node_format && rt_set_option(ctx, RT_OPT_NODE_FORMAT, node_format)) return die("node format");
    if (rt_host_scene_finish(hs, ctx)) return die("scene upload");       // ≙ scene.finish(&gpu), src/main.rs:39
    if (rt_accum_resize(ctx, w, h)) return die("accum");                 // ≙ src/app.rs:87-93
    if (rt_dispatch_frames(ctx, &u, frames)) return die("dispatch");     // ≙ src/app.rs:78-101, `frames` times
    std::vector<uint8_t> rgb((size_t)w * h * 3);
    if (rt_capture_rgb8(ctx, rgb.data())) return die("capture");         // ≙ src/ui.rs:82-88, converted on the device
    float ms = 0;
    rt_last_dispatch_ms(ctx, &ms);
    RtStats st;
    rt_stats(ctx, &st);
    if (rt_host_write_png(out.c_str(), rgb.data(), w, h)) return die("png");
    printf("%s: %ux%u, %u spp x %u frames, %llu rays in %.3f ms (%.1f Mrays/s) -> %s\n", scene.c_str(), w, h, u.samples, frames,
           (unsigned long long)st.rays, ms, st.rays / (ms * 1e3), out.c_str());
    rt_destroy(ctx);
    rt_host_scene_free(hs);
    return 0;
}
