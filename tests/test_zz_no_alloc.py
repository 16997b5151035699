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


# ---- the property itself: nothing is allocated, and the call can be recorded into a HIP graph ----
import ctypes as C


def _hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    h.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    h.hipStreamDestroy.argtypes = [C.c_void_p]
    h.hipStreamSynchronize.argtypes = [C.c_void_p]
    h.hipStreamBeginCapture.argtypes = [C.c_void_p, C.c_int]
    h.hipStreamEndCapture.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    h.hipGraphInstantiate.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    h.hipGraphLaunch.argtypes = [C.c_void_p, C.c_void_p]
    h.hipGraphExecDestroy.argtypes = [C.c_void_p]
    h.hipGraphDestroy.argtypes = [C.c_void_p]
    return h


def _free_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_entry_point_allocates_nothing(rt):
    """The property the entry point exists for, asserted directly: the library's own allocation counter (every hipMalloc of
    librt_amd.so goes through one counted function) and the device's free memory are the same before and after static
    dispatches — including the FIRST one on a fresh context, where rt_dispatch_frames would allocate its staging planes."""
    hip = _hip()
    cfg = config("C3")
    hs = rt.HostScene(scene_path("teapot-circle"))
    with rt.Context(0) as ctx, rt.Context(0) as ref:
        ctx.scene_finish(hs)
        ref.scene_finish(hs)
        w, h, frames = 160, 96, 4
        ctx.accum_resize(w, h)
        ref.accum_resize(w, h)
        ctx.sync()
        u = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=2)
        u["frame"], u["accumulation_frame"] = 0, 1
        allocs0, free0 = rt.lib().rt_debug_alloc_count(), _free_bytes(hip)
        ctx.dispatch_frames_static(u, frames)              # first dispatch of this context: nothing may be set up lazily by allocation
        ctx.dispatch_frames_static(u, 2)
        ctx.dispatch_frames_static(u, 1)
        ctx.sync()
        assert rt.lib().rt_debug_alloc_count() == allocs0, "rt_dispatch_frames_static allocated device memory"
        assert _free_bytes(hip) == free0, "free device memory changed across rt_dispatch_frames_static"
        # the counter does see the allocating path (so the assertion above can fail): rt_dispatch_frames allocates its staging
        u2 = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=2)
        u2["frame"], u2["accumulation_frame"] = 0, 1
        ref.dispatch_frames(u2, frames)
        ref.sync()
        assert rt.lib().rt_debug_alloc_count() > allocs0


def test_static_entry_point_under_graph_capture(rt):
    """Recorded with hipStreamBeginCapture and replayed with hipGraphLaunch: the same bits as rt_dispatch_frames.  A capture
    refuses allocation, so a call that allocated would fail here.  The graph is launched twice (the second time on a zeroed
    buffer again): a replay re-runs the recorded memset of the queue head and the recorded kernels with the recorded frame numbers."""
    hip = _hip()
    cfg = config("C3")
    hs = rt.HostScene(scene_path("teapot-circle"))
    w, h, frames = 120, 72, 3
    stream = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0          # hipStreamNonBlocking
    with rt.Context(0) as ctx, rt.Context(0) as ref:
        ref.scene_finish(hs)
        ctx.scene_finish(hs)
        want, _, _, ws = _render(rt, ref, cfg, w, h, frames, static=False)
        ctx.set_stream(stream.value)
        ctx.accum_resize(w, h)
        warm = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=4)
        ctx.dispatch_frames_static(warm, 1)                # one warm call outside the capture (occupancy query, module load)
        ctx.accum_resize(w, h)                             # fresh, zeroed buffer: the captured run is a cold start like `want`
        ctx.stats_reset()
        ctx.sync()
        u = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=4)
        u["frame"], u["accumulation_frame"] = 0, 1
        allocs0 = rt.lib().rt_debug_alloc_count()
        assert hip.hipStreamBeginCapture(stream, 0) == 0                   # hipStreamCaptureModeGlobal: the strictest
        try:
            ctx.dispatch_frames_static(u, frames)
        finally:
            graph = C.c_void_p()
            rc = hip.hipStreamEndCapture(stream, C.byref(graph))
        assert rc == 0 and graph.value
        assert rt.lib().rt_debug_alloc_count() == allocs0
        assert (int(u["frame"][0]), int(u["accumulation_frame"][0])) == (frames, frames - 1)
        gexec = C.c_void_p()
        assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
        assert hip.hipGraphLaunch(gexec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        got = ctx.accum_download()
        assert_images_identical(got, want, "graph replay of rt_dispatch_frames_static")
        assert ctx.stats().rays == ws.rays
        # a second replay on top of the finished image: the graph carries its own queue-head reset and the recorded frame
        # numbers, and the first recorded blend has weight 1/1 (it overwrites), so the same image comes out again
        assert hip.hipGraphLaunch(gexec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        assert_images_identical(ctx.accum_download(), want, "second graph replay")
        hip.hipGraphExecDestroy(gexec)
        hip.hipGraphDestroy(graph)
    hip.hipStreamDestroy(stream)


def test_static_calls_of_two_contexts_on_one_gpu_are_serialised(rt):
    """two contexts (own non-blocking streams) share the GPU's one scratch array.  Issued back to back without a
    sync, the second call used to zero the first one's live queue head.  Every static call now waits for the previous one on that
    GPU (per-device event), so both images equal the allocating path's, however the streams would have overlapped."""
    cfg = config("C3")
    hs = rt.HostScene(scene_path("teapot-circle"))
    with rt.Context(0) as a, rt.Context(0) as b, rt.Context(0) as ref:
        for c in (a, b, ref):
            c.scene_finish(hs)
        shapes = {"a": (320, 200, 6), "b": (96, 56, 3)}     # a: a long launch; b: short, issued while a's is still running
        want = {k: _render(rt, ref, cfg, *shapes[k], static=False)[0] for k in shapes}
        for rep in range(3):
            us = {}
            for k, c in (("a", a), ("b", b)):
                w, h, frames = shapes[k]
                c.accum_resize(w, h)
                us[k] = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=4)
                us[k]["frame"], us[k]["accumulation_frame"] = 0, 1
            a.sync(); b.sync()
            a.dispatch_frames_static(us["a"], shapes["a"][2])      # no sync between the two: asynchronous launches on two streams
            b.dispatch_frames_static(us["b"], shapes["b"][2])
            a.dispatch_frames_static(us["a"], 1)                   # and a third one behind them, on the first context again
            assert_images_identical(b.accum_download(), want["b"], f"context b, round {rep}")
            ref2, _, _, _ = _render(rt, ref, cfg, *shapes["a"][:2], shapes["a"][2] + 1, static=False)
            assert_images_identical(a.accum_download(), ref2, f"context a, round {rep}")
