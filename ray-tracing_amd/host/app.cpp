// app.cpp — the frame driver of the reference, /root/reference/src/app.rs:20-103,
// without the window: the state the winit loop keeps between frames and the
// bookkeeping App::render does before it dispatches.  Time is an argument
// (seconds, any monotonic origin) instead of Instant::now(), so the sequence
// is reproducible.
#include <cmath>
#include <cstring>

#include "rt_host.h"

extern "C" {

// ≙ the App literal at src/main.rs:63-92
void rt_host_app_init(RtAppState *app, uint32_t inner_width, uint32_t inner_height, double now) {
    if (!app) return;
    memset(app, 0, sizeof *app);
    rt_host::uniform_default(app->uniform);          // frame 0, accumulation_frame 1, samples 5, max_bounces 10, ...
    app->inner_size[0] = inner_width;
    app->inner_size[1] = inner_height;
    app->last_window[0] = app->last_window[1] = 0;   // Vector2::zeros(), src/main.rs:89
    app->last_invalidation = now;                    // Instant::now(), src/main.rs:88
    app->accumulate = 1;                             // src/main.rs:90
    app->screen_fraction = 2;                        // src/main.rs:91
}

// ≙ App::invalidate_accumulation, src/app.rs:41-44
void rt_host_app_invalidate(RtAppState *app, double now) {
    if (!app) return;
    app->last_invalidation = now;
    app->uniform.accumulation_frame = 0;
}

// The fields `impl Hash for Uniform` covers (src/types.rs:112-121): camera, flags,
// exposure, environment, max_bounces, samples — window / frame counters are not hashed.
static bool hashed_fields_equal(const RtUniform &a, const RtUniform &b) {
    return memcmp(a.camera.position, b.camera.position, 12) == 0 && memcmp(&a.camera.pitch, &b.camera.pitch, 4) == 0 &&
           memcmp(&a.camera.yaw, &b.camera.yaw, 4) == 0 && memcmp(&a.camera.fov, &b.camera.fov, 4) == 0 &&
           memcmp(&a.camera.aspect, &b.camera.aspect, 4) == 0 && a.flags == b.flags &&
           memcmp(&a.exposure, &b.exposure, 4) == 0 && memcmp(&a.environment, &b.environment, 4) == 0 &&
           a.max_bounces == b.max_bounces && a.samples == b.samples;
}

// ≙ the tail of ui(), src/ui.rs:19,97-100: take the edited render parameters and
// invalidate the accumulation when a hashed field changed.  Returns 1 if it did.
int rt_host_app_set_params(RtAppState *app, const RtUniform *edited, double now) {
    if (!app || !edited) return 0;
    const bool same = hashed_fields_equal(app->uniform, *edited);
    app->uniform.camera = edited->camera;
    app->uniform.flags = edited->flags;
    app->uniform.exposure = edited->exposure;
    app->uniform.environment = edited->environment;
    app->uniform.max_bounces = edited->max_bounces;
    app->uniform.samples = edited->samples;
    if (!same) rt_host_app_invalidate(app, now);
    return same ? 0 : 1;
}

// ≙ App::render up to the upload, src/app.rs:74-97.  Returns 1 when the window
// size changed (the caller must then re-create a zeroed accumulation buffer of
// uniform.window, src/app.rs:87-93), else 0.  A window of 0 pixels yields no dispatch.
int rt_host_app_begin_frame(RtAppState *app, double now) {
    if (!app) return 0;
    float elapsed = (float)(now - app->last_invalidation);
    if (elapsed < 0.0f) elapsed = 0.0f;
    if (elapsed > 1.0f) elapsed = 1.0f;                                     // .min(1.0)
    const float dynamic_fraction = 1.0f + (4.0f - 1.0f) * (1.0f - elapsed);   // src/app.rs:75-76

    app->uniform.accumulation_frame += 1;                                   // :78
    if (!app->accumulate) app->uniform.accumulation_frame = 0;              // :79-81

    uint32_t div = (uint32_t)app->screen_fraction;                          // :83-85
    const uint32_t dyn = (uint32_t)dynamic_fraction;                        // `screen_fraction as u32` truncates
    if (dyn > div) div = dyn;
    if (div == 0) div = 1;
    const uint32_t window[2] = {app->inner_size[0] / div, app->inner_size[1] / div};

    int resized = 0;
    if (app->last_window[0] != window[0] || app->last_window[1] != window[1]) {   // :87-93
        app->uniform.accumulation_frame = 0;
        app->last_window[0] = window[0];
        app->last_window[1] = window[1];
        resized = 1;
    }
    app->uniform.window[0] = window[0];                                     // :95
    app->uniform.window[1] = window[1];
    app->uniform.frame += 1;                                                // :96
    // Camera::handle_movement keeps aspect = inner width / height (src/camera.rs:54-56)
    if (app->inner_size[1]) app->uniform.camera.aspect = (float)app->inner_size[0] / (float)app->inner_size[1];
    return resized;
}

}  // extern "C"
