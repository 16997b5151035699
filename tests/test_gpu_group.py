"""
rt_group (include/rt_abi.h): n GPUs behind one host object, ≙ SURVEY §8(b) `create(device_ids[], n)`.  On the one-GPU test
box the same device is named n times: everything but the physical peer link is the multi-GPU code path — one host
build shared by all contexts, per-shard streams, the event-ordered gather to the first device, the de-interleave kernel.
The gathered image must equal a single context's image bit for bit (pixel_idx stays global, main.wgsl:19,23).
"""
import numpy as np
import pytest

from conftest import config, scene_path, scenes, assert_images_identical  # noqa: F401

pytestmark = pytest.mark.gpu


def _cold(rt, cfg, w, h, samples):
    u = rt.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=samples)
    u["frame"], u["accumulation_frame"] = 0, 1          # cold start, src/main.rs:76-77
    return u


@pytest.mark.parametrize("n,w,h", [(2, 72, 52), (3, 40, 61), (8, 64, 20), (1, 33, 17)])
def test_group_equals_single_context(rt, scenes, n, w, h):
    cfg = config("C3")
    s = scenes("teapot-circle")
    frames, samples = 3, 2
    with rt.Context(0) as one:
        one.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        one.accum_resize(w, h)
        u1 = _cold(rt, cfg, w, h, samples)
        one.dispatch_frames(u1, frames)
        want, want_rgb, st1 = one.accum_download(), one.capture_rgb8(), one.stats()
    with rt.Group([0] * n) as g:
        assert len(g) == n
        g.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        g.accum_resize(w, h)
        ug = _cold(rt, cfg, w, h, samples)
        g.dispatch_frames(ug, frames)
        got, rgb, st = g.accum_download(), g.capture_rgb8(), g.stats()
        assert ug.tobytes() == u1.tobytes()                        # the uniform advanced as rt_dispatch_frames advances it
        assert st.rays == st1.rays and st.pixel_frames == st1.pixel_frames == w * h * frames
        assert st.bvh_rebuilds == 1 and st.bvh_nodes == st1.bvh_nodes     # ONE host build for the whole group
        assert_images_identical(got, want, f"group of {n}")
        assert np.array_equal(rgb, want_rgb)
        # one more interactive frame on top (rt_group_dispatch) and a second, smaller window
        w2, h2 = 24, 19
        g.accum_resize(w2, h2)
        u = rt.make_uniform(w2, h2, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=1)
        g.dispatch(u)
        small = g.accum_download()
    with rt.Context(0) as one:
        one.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        one.accum_resize(w2, h2)
        one.dispatch(u)
        assert_images_identical(small, one.accum_download(), f"group of {n}, second window")


@pytest.mark.parametrize("n,w,h,cuts", [(3, 61, 83, (0, 2, 3)), (2, 40, 24, (0, 1))])
def test_group_frame_in_pieces(rt, scenes, n, w, h, cuts):
    """rt_group_dispatch_bands: every shard renders its own bands of the range; the pieces of a frame (any order, an open end,
    a range beyond a short shard's last band) add up to one rt_dispatch of a single context, bit for bit."""
    cfg = config("C3")
    s = scenes("teapot-circle")
    with rt.Context(0) as one:
        one.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        one.accum_resize(w, h)
        for f in (0, 1):
            u = _cold(rt, cfg, w, h, 2)
            u["frame"], u["accumulation_frame"] = 21 + f, f
            one.dispatch(u)
        want, st1 = one.accum_download(), one.stats()
    with rt.Group([0] * n) as g:
        g.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        g.accum_resize(w, h)
        for f in (0, 1):
            u = _cold(rt, cfg, w, h, 2)
            u["frame"], u["accumulation_frame"] = 21 + f, f
            ends = list(cuts[1:]) + [0xFFFFFFFF]
            for b0, b1 in reversed(list(zip(cuts, ends))):
                g.dispatch_bands(u, b0, b1)
        got, st = g.accum_download(), g.stats()
        assert st.rays == st1.rays and st.pixel_frames == st1.pixel_frames == w * h * 2
        assert_images_identical(got, want, "group frame in pieces vs one context")
        with pytest.raises(rt.RtError):
            g.dispatch_bands(u, 3, 2)


@pytest.mark.parametrize("refit", [2, 0])
def test_group_models_update_and_spheres(rt, oracle_mod, obj_loader, scenes, refit):
    """rt_group_models_update ≙ App::upload_models on every device (device refit, or one host rebuild shared by all), with
    spheres in the scene; checked against the oracle for the moved geometry."""
    cfg = config("C2")
    w, h = 64, 36
    s = scenes("teapot")
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    moved = ident.copy()
    moved[0] = [0.7, 0.2, 0, 0.5, -0.2, 0.9, 0, 0.3, 0, 0, 1.2, -0.4]
    spheres = np.zeros(5, rt.SPHERE_DTYPE)
    spheres["center"] = [(0.5 * k - 1.0, 0.6, 0.3 * k) for k in range(5)]
    spheres["radius"] = 0.15
    spheres["model"] = 1
    want, _, cnt = oracle_mod.OracleScene(s, transforms=moved, spheres=spheres).render(uniforms(oracle_mod, cfg, w, h), mode=1)
    with rt.Group([0, 0, 0]) as g:
        g.set_option(rt.RT_OPT_REFIT, refit)
        g.scene_upload(s.vertices, s.indices, s.primitives, s.models, transforms=ident)
        g.spheres_upload(spheres)
        g.models_update(s.models, transforms=moved)
        g.accum_resize(w, h)
        g.dispatch(uniforms(rt, cfg, w, h))
        st = g.stats()
        assert st.rays == cnt.rays
        assert (st.bvh_refits >= 1) if refit == 2 else (st.bvh_rebuilds >= 3)
        assert_images_identical(g.accum_download(), want, f"group models_update, refit mode {refit}")


def uniforms(mod, cfg, w, h):
    return mod.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=2)


def test_group_errors(rt):
    with pytest.raises(rt.RtError):
        rt.Group([])
    with pytest.raises(rt.RtError):
        rt.Group([0, 4096])
    with rt.Group([0, 0]) as g:
        with pytest.raises(rt.RtError):
            g.spheres_upload(np.zeros(1, rt.SPHERE_DTYPE))       # no scene yet
