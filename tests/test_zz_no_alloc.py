"""
rt_dispatch_frames_static (include/rt_abi.h): the entry point that allocates nothing.

Checked through the C ABI against rt_dispatch_frames, bit for bit, on two calls in
a row — the second with a smaller window and fewer frames, so that whatever the
first call left in the GPU's static scratch array would show — and that a
request beyond the array's size is refused with a clear message.  Collected last
(file name) so that a failure here cannot hide the other GPU tests.
"""
import numpy as np
import pytest

from conftest import config, scene_path, assert_images_identical

pytestmark = pytest.mark.gpu


def _render(rt, ctx, cfg, w, h, frames, static, samples=4):
    ctx.accum_resize(w, h)
    u = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=samples)
    u["frame"] = 0                 # cold start, src/main.rs:76-77
    u["accumulation_frame"] = 1
    ctx.stats_reset()
    if static:
        ctx.dispatch_frames_static(u, frames)
    else:
        ctx.dispatch_frames(u, frames)
    img = ctx.accum_download()
    return img, int(u["frame"][0]), int(u["accumulation_frame"][0]), ctx.stats()


def test_static_entry_point_matches_dispatch_frames(rt):
    cfg = config("C3")
    hs = rt.HostScene(scene_path("teapot-circle"))
    with rt.Context(0) as ref, rt.Context(0) as ctx:
        ref.scene_finish(hs)
        ctx.scene_finish(hs)
        # two calls in a row on the same context; the second smaller in both window and frame count
        for (w, h, frames) in ((200, 120, 5), (72, 40, 3)):
            want, wf, wa, ws = _render(rt, ref, cfg, w, h, frames, static=False)
            got, gf, ga, gs = _render(rt, ctx, cfg, w, h, frames, static=True)
            assert (gf, ga) == (wf, wa) == (frames, frames - 1)
            assert gs.rays == ws.rays and gs.pixel_frames == ws.pixel_frames == w * h * frames
            assert gs.launches == 1                      # the whole run in one launch, staged in the static array
            assert_images_identical(got, want, f"static vs rt_dispatch_frames {w}x{h}x{frames}")
        # a continued accumulation (not a cold start) and a single frame, which stages nothing
        u1 = rt.make_uniform(72, 40, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=4)
        u1["frame"], u1["accumulation_frame"] = 3, 2
        u2 = u1.copy()
        ref.dispatch_frames(u1, 1)
        ctx.dispatch_frames_static(u2, 1)
        assert_images_identical(ctx.accum_download(), ref.accum_download(), "static, one more frame")
        assert int(u1["accumulation_frame"][0]) == int(u2["accumulation_frame"][0]) == 3


def test_static_entry_point_rejects_what_does_not_fit(rt):
    cfg = config("C3")
    hs = rt.HostScene(scene_path("teapot-circle"))
    with rt.Context(0) as ctx:
        ctx.scene_finish(hs)
        w, h = 1024, 1024                                   # 16 MiB per staged frame: 2 fit the 32 MiB array, 3 do not
        ctx.accum_resize(w, h)
        u = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=1, samples=1)
        before = u.copy()
        with pytest.raises(rt.RtError) as e:
            ctx.dispatch_frames_static(u, 3)
        assert e.value.code == -1 and "static" in str(e.value) and str(3 * w * h * 16) in str(e.value)
        assert u.tobytes() == before.tobytes()              # nothing dispatched, uniform untouched
        ctx.dispatch_frames_static(u, 2)                    # exactly the limit
        ctx.sync()
        assert ctx.stats().launches >= 1
        ctx.set_option(rt.RT_OPT_AOV, 1)
        with pytest.raises(rt.RtError):
            ctx.dispatch_frames_static(u, 1)
