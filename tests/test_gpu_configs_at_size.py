"""
BASELINE configs C4 and C5 at their real scene sizes ("configs not exercised by -m gpu at size").

The scenes are the full generators of configs/baseline_configs.json — 10 M triangles + 100 k spheres (C4),
50 M triangles (C5q, quarter-scale C5) and 200 M triangles (C5) — rendered at a small window so that the CPU
oracle (its own BVH over the same primitives, `mode=1`) finishes in seconds.  C4 and C5q are compared with the
oracle bit for bit (image, primary hit ids, ray count); C5 is too big for the oracle's single-threaded builder to
be worth the minutes, so it is checked through size-independent properties and by rendering it with both node
formats, which must agree bit for bit.
"""
import contextlib
import os
import sys
import threading
import time

import numpy as np
import pytest

from conftest import ROOT, assert_images_identical

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def heartbeat(capsys, label):
    """These tests run for minutes (host BVH builds of 10^7..10^8 triangles).  The GPU harness takes a command that
    writes nothing for three minutes to be hung, so a line goes to the real stdout (and to gpurun_out/) every 30 s."""
    stop = threading.Event()
    t0 = time.time()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "at_size_progress.log"), "a")

    def beat():
        while not stop.wait(30.0):
            line = f"[{label}] still running, {time.time() - t0:.0f} s\n"
            sys.stdout.write(line)
            sys.stdout.flush()
            log.write(line)
            log.flush()
    with capsys.disabled():
        th = threading.Thread(target=beat, daemon=True)
        th.start()
        try:
            yield
        finally:
            stop.set()
            th.join()
            log.write(f"[{label}] done after {time.time() - t0:.0f} s\n")
            log.close()


def _scene(rt, name):
    import __graft_entry__ as graft
    graft.load_package()
    from ray_tracing_amd import synth
    cfg = graft.load_config(name)
    return cfg, synth.load_config_scene(cfg, ROOT)


def _uniform(mod, cfg, w, h, samples):
    # the config's own camera and aspect (W/H of the full-size window), at a small window
    return mod.make_uniform(w, h, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]),
                            max_bounces=cfg["max_bounces"], samples=samples)


def _against_oracle(rt, oracle_mod, obj_loader, name, w, h, samples, expect_wide, audit):
    """audit = (x0, x1, y0, y1, spp): a sub-rectangle of the window rendered once more, at `spp` samples, by the oracle's BRUTE FORCE
    (every primitive tested against every ray, no tree at all) and compared with the device's pixels and primary hit ids there.
    The whole-window check uses the oracle's BVH mode, which is a checker that has itself lost hits before:
    the sub-rectangle is what ties these sizes to the ground truth."""
    cfg, s = _scene(rt, name)
    n_prims = len(s.indices) // 3 + len(s.spheres)
    osc = oracle_mod.OracleScene(obj_loader.Scene(s.vertices, s.indices, s.primitives, s.models),
                                 spheres=s.spheres if len(s.spheres) else None)
    want, want_hits, cnt = osc.render(_uniform(oracle_mod, cfg, w, h, samples), mode=1, want_hits=True)
    ax0, ax1, ay0, ay1, aspp = audit
    brute, brute_hits, bcnt = osc.render(_uniform(oracle_mod, cfg, w, h, aspp), mode=0, want_hits=True, rows=(ay0, ay1), cols=(ax0, ax1))
    assert bcnt.pixel_frames == (ax1 - ax0) * (ay1 - ay0) and bcnt.triangle_tests >= bcnt.rays * (n_prims - len(s.spheres))
    del osc
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        if len(s.spheres):
            ctx.spheres_upload(s.spheres)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(_uniform(rt, cfg, w, h, samples))
        st = ctx.stats()
        got, hits = ctx.accum_download(), ctx.aov_download()
        # the frame of the brute-force audit: the same context, a fresh (zeroed) buffer, `aspp` samples
        ctx.accum_resize(w, h)
        ctx.dispatch(_uniform(rt, cfg, w, h, aspp))
        got2, hits2 = ctx.accum_download(), ctx.aov_download()
    assert st.pixel_frames == w * h
    assert st.rays == cnt.rays, (st.rays, cnt.rays)
    # the node format chosen by size (RT_OPT_NODE_FORMAT = 0): 64-byte 8-ary nodes once the binary nodes would exceed 12 MiB
    per_node = (st.bvh_bytes - n_prims * (48 + 128)) / st.bvh_nodes
    assert (per_node == 64) == expect_wide, per_node
    assert np.array_equal(hits, want_hits), f"{int((hits != want_hits).any(axis=-1).sum())} primary hit ids differ"
    assert_images_identical(got, want, name)
    # the brute-force audit of the sub-rectangle
    assert np.array_equal(hits2[ay0:ay1, ax0:ax1], brute_hits[ay0:ay1, ax0:ax1]), "primary hit ids differ from BRUTE FORCE in the audited rectangle"
    assert_images_identical(got2[ay0:ay1, ax0:ax1], brute[ay0:ay1, ax0:ax1], name + ": device vs brute force, audited rectangle")
    assert (brute_hits[ay0:ay1, ax0:ax1, 0] != 0xFFFFFFFF).mean() > 0.5, "the audited rectangle should look at geometry, not at the sky"
    return st


def test_c4_full_scene_matches_oracle(rt, oracle_mod, obj_loader, capsys):
    """C4: 1,059 baked teapots (9.99 M triangles) + 100,000 spheres, 160x90, 16 spp, 8 bounces."""
    with heartbeat(capsys, "C4 10M"):
        st = _against_oracle(rt, oracle_mod, obj_loader, "C4", 160, 90, 16, expect_wide=True, audit=(72, 88, 40, 49, 2))     # 16 x 9 pixels, 2 spp
    assert st.bvh_nodes > 10_090_000 // 8 and st.stack_entries >= 16        # at most 8 children per node


def test_c5q_50m_triangles_match_oracle(rt, oracle_mod, obj_loader, capsys):
    """C5q: 5,300 baked teapots (50 M triangles, > 4 GiB of device records: the 64-bit offset paths), 96x54, 4 spp."""
    with heartbeat(capsys, "C5q 50M"):
        st = _against_oracle(rt, oracle_mod, obj_loader, "C5q", 96, 54, 4, expect_wide=True, audit=(44, 52, 25, 29, 2))     # 8 x 4 pixels, 2 spp
    assert st.bvh_bytes > (4 << 30)


def test_c5_200m_triangles_properties(rt, capsys):
    with heartbeat(capsys, "C5 200M"):
        _c5_properties(rt)


def _c5_properties(rt):
    """C5: 21,190 baked teapots (200 M triangles).  Upload + one low-resolution frame per node format; the two formats'
    images must hold the same bits (both box tests are conservative, the triangle test is the same), and the frame
    must satisfy what holds for any render of this scene."""
    cfg, s = _scene(rt, "C5")
    n_tris = len(s.indices) // 3
    assert n_tris >= 199_000_000
    w, h, samples = 96, 54, 4
    images, stats = [], []
    for fmt in (0, 1):                                   # 0: by size (8-ary nodes at this size), 1: 48-byte binary nodes
        with rt.Context(0) as ctx:
            ctx.set_option(rt.RT_OPT_NODE_FORMAT, fmt)
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            ctx.accum_resize(w, h)
            ctx.dispatch(_uniform(rt, cfg, w, h, samples))
            stats.append(ctx.stats())
            images.append(ctx.accum_download())
    a, b = stats
    img = images[0]
    assert a.pixel_frames == b.pixel_frames == w * h
    assert a.rays == b.rays
    # every path has between 1 (primary miss) and max_bounces + 1 segments
    assert w * h * samples <= a.rays <= w * h * samples * (cfg["max_bounces"] + 1)
    assert n_tris // 8 <= a.bvh_nodes < b.bvh_nodes       # the 8-ary format was chosen by size: fewer, wider nodes
    assert a.bvh_bytes < b.bvh_bytes
    assert 2 <= a.stack_entries <= 40 and 2 <= b.stack_entries <= 48
    assert np.isfinite(img[..., :3]).all() and (img[..., :3] >= 0).all() and (img[..., :3] <= 1).all()   # tone-mapped (main.wgsl:117)
    assert (img[..., 3] == 0).all()                       # the pad lane of array<vec3f> is never written
    assert img[..., :3].max() > 0.05                      # not a black frame
    assert_images_identical(images[0], images[1], "C5: 8-ary vs 48-byte binary nodes")
