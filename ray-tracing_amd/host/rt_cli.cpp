// rt_cli.cpp — headless driver: the per-frame loop of the reference's App
// (/root/reference/src/app.rs:74-102) without a window, plus Capture
// (src/ui.rs:78-94).  The reference's own TODO.md:6 asks for this mode.
// Uses only the C ABI of include/rt_abi.h, like a Rust host would.
//
//   rt_cli --scene scenes/cornell-box.obj --size 256x256 --samples 16 --frames 1 --bounces 4 \
//          --pos 0,15,14 --yaw 4.712389 --pitch 0 --out out.png
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
    int device = 0;
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
        else { fprintf(stderr, "rt_cli: unknown argument %s\n", a.c_str()); return 2; }
    }
    if (scene.empty()) { fprintf(stderr, "usage: rt_cli --scene file.obj [--size WxH] [--samples N] [--frames N] [--bounces N] [--pos x,y,z] [--yaw r] [--pitch r] [--out file.png]\n"); return 2; }

    rt_host_scene *hs = nullptr;
    if (rt_host_scene_load(scene.c_str(), &hs)) return die("load");      // ≙ scene.load(), src/main.rs:37
    rt_context *ctx = nullptr;
    if (rt_create(device, &ctx)) return die("create");
    if (rt_host_scene_finish(hs, ctx)) return die("scene upload");       // ≙ scene.finish(&gpu), src/main.rs:39
    if (rt_accum_resize(ctx, w, h)) return die("accum");                 // ≙ src/app.rs:87-93
    u.window[0] = w; u.window[1] = h;
    u.camera.aspect = (float)w / (float)h;                               // src/camera.rs:56
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
