"""GPU parity tests: the HIP path through the C ABI against the CPU oracle and the golden fixtures.

Bar (north_star): primary hit ids bit-exact excluding documented tie cases; radiance within a stated
tolerance.  Because both sides implement the RTN-1 arithmetic contract the tests assert MORE: the whole
accumulation buffer is bit-identical (tolerance 0; NaN == NaN), and there are no tie cases at all at
these sizes (the assertion allows none)."""
import os

import numpy as np
import pytest

from conftest import ROOT, assert_images_identical, config, scene_path, uniforms_for

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs(rt):
    """One context per scene, reused across tests (scene upload + BVH build once)."""
    cache = {}

    def get(name):
        if name not in cache:
            c = rt.Context(0)
            c.scene_finish(rt.HostScene(scene_path(name)))
            c.set_option(rt.RT_OPT_AOV, 1)
            cache[name] = c
        return cache[name]
    yield get
    for c in cache.values():
        c.close()


def _scene_of(cfg):
    return os.path.basename(cfg["scene"])[:-4]


@pytest.mark.parametrize("cname,w,h,spp", [("C1", 64, 64, 4), ("C1", 61, 37, 2), ("C2", 128, 72, 3), ("C3", 128, 72, 3),
                                           ("C1", 8, 8, 1), ("C1", 1, 1, 5), ("C3", 200, 9, 2)])
def test_single_dispatch_bit_identical(rt, oracle_mod, oracle_scenes, ctxs, cname, w, h, spp):
    cfg = config(cname)
    ctx = ctxs(_scene_of(cfg))
    ctx.accum_resize(w, h)
    ctx.stats_reset()
    ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=spp))
    got, hits, st = ctx.accum_download(), ctx.aov_download(), ctx.stats()
    want, want_hits, cnt = oracle_scenes(_scene_of(cfg)).render(uniforms_for(oracle_mod, cfg, w, h, samples=spp), mode=1, want_hits=True)
    assert st.rays == cnt.rays and st.pixel_frames == w * h
    assert np.array_equal(hits, want_hits), f"{(hits != want_hits).any(axis=2).sum()} primary hit ids differ"
    assert_images_identical(got, want, f"{cname} {w}x{h}")


@pytest.mark.parametrize("name", ["c1_64", "c2_96x54", "c3_96x54"])
def test_golden_fixtures(rt, ctxs, name):
    """Golden vectors made by the oracle's BRUTE-FORCE mode (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    w, h, spp, frames = (int(x) for x in g["meta"])
    cfg = config(name[:2].upper())
    ctx = ctxs(_scene_of(cfg))
    ctx.accum_resize(w, h)
    ctx.stats_reset()
    u = uniforms_for(rt, cfg, w, h, samples=spp)
    u["frame"], u["accumulation_frame"] = 0, 1           # cold start, src/main.rs:76-77
    ctx.dispatch_frames(u, 1)
    hits = ctx.aov_download()
    if frames > 1:
        ctx.dispatch_frames(u, frames - 1)
    assert (int(u["frame"][0]), int(u["accumulation_frame"][0])) == (frames, frames - 1)
    assert ctx.stats().rays == int(g["rays"])
    assert np.array_equal(hits, g["hits"])
    assert_images_identical(ctx.accum_download(), g["accum"], name)


def test_closed_box_ray_count_full_c1(rt, ctxs):
    """BASELINE config C1 at full size.  The box is closed and nothing absorbs, so every path runs all
    5 segments: 256*256*16*5 = 5,242,880 rays, MINUS the segments lost when a path leaks through a wall
    seam (the walls are separate quads whose corners differ by 1e-6).  The oracle's
    brute-force mode counts 51 such escapes (46 segments never traced): 5,242,834 rays."""
    cfg = config("C1")
    ctx = ctxs("cornell-box")
    ctx.accum_resize(cfg["width"], cfg["height"])
    ctx.stats_reset()
    ctx.dispatch(uniforms_for(rt, cfg, cfg["width"], cfg["height"]))
    st = ctx.stats()
    assert st.rays == 5242834 and 256 * 256 * 16 * 5 - st.rays == 46
    img = ctx.accum_download()
    assert np.isfinite(img).all() and img[..., :3].min() >= 0 and img[..., :3].max() <= 1


def test_c1_full_size_matches_oracle(rt, oracle_mod, oracle_scenes, ctxs):
    cfg = config("C1")
    w, h = cfg["width"], cfg["height"]
    ctx = ctxs("cornell-box")
    ctx.accum_resize(w, h)
    ctx.dispatch(uniforms_for(rt, cfg, w, h))
    want, want_hits, _ = oracle_scenes("cornell-box").render(uniforms_for(oracle_mod, cfg, w, h), mode=1, want_hits=True)
    assert np.array_equal(ctx.aov_download(), want_hits)
    assert_images_identical(ctx.accum_download(), want, "C1 full")


def test_cull_backfaces_flag(rt, oracle_mod, oracle_scenes, ctxs):
    """flags bit 0 → ray flag 0x10 (main.wgsl:76): back-facing triangles are skipped."""
    cfg = config("C2")
    w, h = 96, 54
    ctx = ctxs("teapot")
    ctx.accum_resize(w, h)
    ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2, flags=1))
    want, want_hits, _ = oracle_scenes("teapot").render(uniforms_for(oracle_mod, cfg, w, h, samples=2, flags=1), mode=1, want_hits=True)
    assert np.array_equal(ctx.aov_download(), want_hits)
    assert_images_identical(ctx.accum_download(), want, "cull")
    nocull, _, _ = oracle_scenes("teapot").render(uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1)
    assert not np.array_equal(nocull, want)     # the flag does something on this scene


def test_material_override_specular_and_dielectric(rt, oracle_mod, obj_loader, scenes):
    """models_update (≙ App::upload_models, src/app.rs:46-62): metal with Pm/Pr and a dielectric teapot —
    the branches no as-loaded scene reaches (ray.wgsl:13,29-30,34-49)."""
    cfg = config("C2")
    w, h = 96, 54
    s = scenes("teapot")
    models = s.models.copy()
    models["tag"][0] = 1
    models["refractive_index"][0] = 1.5
    models["specular_probability"][1] = 0.6
    models["roughness"][1] = 0.2
    models["specular_color"][1] = (0.9, 0.8, 0.7)
    import copy
    s2 = copy.copy(s)
    s2.models = models
    osc = oracle_mod.OracleScene(s2)
    want, want_hits, cnt = osc.render(uniforms_for(oracle_mod, cfg, w, h, samples=4), mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        ctx.models_update(models)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=4))
        assert ctx.stats().rays == cnt.rays
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, "materials")


def test_geometry_transforms(rt, oracle_mod, scenes):
    """Per-geometry 3x4 transforms move the triangles the BVH sees; shading still interpolates the
    untransformed vertices (main.wgsl:95-99 multiplies by the identity instance matrix)."""
    cfg = config("C2")
    w, h = 80, 45
    s = scenes("teapot")
    t = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    t[0] = [0.8, 0, 0, 0.3, 0, 1.1, 0, 0.2, 0, 0, 0.9, -0.1]
    osc = oracle_mod.OracleScene(s, transforms=t)
    want, want_hits, _ = osc.render(uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models, transforms=t)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, "transforms")


def test_spheres(rt, oracle_mod, obj_loader, scenes):
    """Build-defined sphere primitives (BASELINE config C4): mixed with triangles in one BVH;
    metal and dielectric spheres; parity is HIP vs the oracle only (the reference has no spheres)."""
    cfg = config("C2")
    w, h = 112, 63
    s = scenes("teapot")
    models = s.models.copy()
    models["tag"][2] = 1
    models["refractive_index"][2] = 1.4
    import copy
    s2 = copy.copy(s)
    s2.models = models
    rng = np.random.default_rng(3)
    sp = np.zeros(60, obj_loader.SPHERE_DTYPE)
    sp["center"] = rng.uniform([-1.5, 0.0, -1.5], [1.5, 1.5, 1.5], (60, 3)).astype(np.float32)
    sp["radius"] = rng.uniform(0.05, 0.3, 60).astype(np.float32)
    sp["model"] = rng.integers(0, len(models), 60)
    osc = oracle_mod.OracleScene(s2, spheres=sp)
    want, want_hits, cnt = osc.render(uniforms_for(oracle_mod, cfg, w, h, samples=3), mode=0, want_hits=True)
    assert (want_hits[..., 0] >= 0x80000000).sum() > 100       # spheres are visible
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, models)
        ctx.spheres_upload(sp)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        for variant in (2, 0, 1):
            ctx.set_option(rt.RT_OPT_KERNEL, variant)
            ctx.accum_resize(w, h)
            ctx.stats_reset()
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=3))
            assert ctx.stats().rays == cnt.rays
            assert np.array_equal(ctx.aov_download(), want_hits)
            assert_images_identical(ctx.accum_download(), want, f"spheres, kernel variant {variant}")
        bad = sp.copy()
        bad["radius"][3] = 0.0
        with pytest.raises(rt.RtError):
            ctx.spheres_upload(bad)


def test_synthetic_lattice_scene(rt, oracle_mod, obj_loader):
    """Small instance of the C4 generator (baked teapot lattice + spheres): deeper BVH, incoherent bounces."""
    import __graft_entry__ as graft
    graft.load_package()
    from ray_tracing_amd import synth
    s = synth.lattice_scene(scene_path("teapot"), 26, 300, seed=1)
    w, h = 96, 54
    cam = s.camera
    ou = oracle_mod.make_uniform(w, h, cam["position"], cam["yaw"], cam["pitch"], max_bounces=8, samples=4)
    osc = oracle_mod.OracleScene(obj_loader.Scene(s.vertices, s.indices, s.primitives, s.models), spheres=s.spheres)
    want, want_hits, cnt = osc.render(ou, mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        ctx.spheres_upload(s.spheres)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(rt.make_uniform(w, h, cam["position"], cam["yaw"], cam["pitch"], max_bounces=8, samples=4))
        assert ctx.stats().rays == cnt.rays
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, "synthetic lattice")


@pytest.mark.parametrize("variant", [0, 1])
def test_kernel_variants_agree(rt, oracle_mod, oracle_scenes, ctxs, variant):
    """The A/B kernel variants (RT_OPT_KERNEL 0: persistent per-ray, 1: tile per wave) produce the same bits."""
    cfg = config("C3")
    w, h = 136, 77
    ctx = ctxs("teapot-circle")
    ctx.set_option(rt.RT_OPT_KERNEL, variant)
    try:
        ctx.accum_resize(w, h)
        ctx.stats_reset()
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=3))
        got, hits, st = ctx.accum_download(), ctx.aov_download(), ctx.stats()
    finally:
        ctx.set_option(rt.RT_OPT_KERNEL, 2)
    want, want_hits, cnt = oracle_scenes("teapot-circle").render(uniforms_for(oracle_mod, cfg, w, h, samples=3), mode=1, want_hits=True)
    assert st.rays == cnt.rays and np.array_equal(hits, want_hits)
    assert_images_identical(got, want, f"variant {variant}")


@pytest.mark.parametrize("world,w,h", [(2, 72, 52), (3, 40, 64), (8, 64, 20)])
def test_shards_tile_the_unsharded_image(rt, ctxs, world, w, h):
    """rt_set_shard: the union of the shards' packed bands is bit-identical to the 1-shard image."""
    cfg = config("C3")
    u = uniforms_for(rt, cfg, w, h, samples=2)
    full_ctx = ctxs("teapot-circle")
    full_ctx.accum_resize(w, h)
    full_ctx.dispatch(u)
    full = full_ctx.accum_download()
    hs = rt.HostScene(scene_path("teapot-circle"))
    seen = np.zeros(h, bool)
    for r in range(world):
        with rt.Context(0) as c:
            c.scene_finish(hs)
            c.set_shard(r, world)
            c.accum_resize(w, h)
            c.dispatch(u)
            part = c.accum_download()
        rows = [y for lb in range((h + 7) // 8) if lb % world == r for y in range(lb * 8, min(lb * 8 + 8, h))]
        assert part.shape[0] == len(rows)
        assert_images_identical(part, full[rows], f"shard {r}/{world}")
        seen[rows] = True
    assert seen.all()


def test_empty_and_tiny_scenes(rt, oracle_mod, obj_loader):
    """Empty scene → pure sky; a single triangle → one-leaf BVH."""
    cfg = config("C2")
    w, h = 32, 24
    empty = obj_loader.Scene(np.zeros(0, obj_loader.VERTEX_DTYPE), np.zeros(0, np.uint32),
                             np.zeros(0, obj_loader.PRIMITIVE_DTYPE), np.zeros(0, obj_loader.MODEL_DTYPE))
    v = np.zeros(3, obj_loader.VERTEX_DTYPE)
    v["position"] = [[-1, 0.5, 1], [1, 0.5, 1], [0, 2.0, 1.5]]
    v["normal"] = [0, 0, -1]
    pr = np.zeros(1, obj_loader.PRIMITIVE_DTYPE)
    pr[0] = (0, 3, 0, 3, 0)
    md = np.zeros(1, obj_loader.MODEL_DTYPE)
    md["diffuse_color"] = (0.5, 0.6, 0.7)
    one = obj_loader.Scene(v, np.arange(3, dtype=np.uint32), pr, md)
    for s in (empty, one):
        want, want_hits, cnt = oracle_mod.OracleScene(s).render(uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=0, want_hits=True)
        with rt.Context(0) as ctx:
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            ctx.set_option(rt.RT_OPT_AOV, 1)
            ctx.accum_resize(w, h)
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
            assert ctx.stats().rays == cnt.rays
            assert np.array_equal(ctx.aov_download(), want_hits)
            assert_images_identical(ctx.accum_download(), want, "tiny scene")


def test_error_paths(rt, scenes):
    with rt.Context(0) as ctx:
        u = rt.make_uniform(16, 16, (0, 0, 0), 0.0)
        with pytest.raises(rt.RtError) as e:
            ctx.dispatch(u)
        assert e.value.code == -4
        s = scenes("teapot")
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        with pytest.raises(rt.RtError) as e:
            ctx.dispatch(u)
        assert e.value.code == -5
        with pytest.raises(rt.RtError) as e:
            ctx.frames_reserve(4)                 # no accumulation buffer yet
        assert e.value.code == -5
        with pytest.raises(rt.RtError):
            ctx.set_option(rt.RT_OPT_NODE_STRIDE, 5)
        ctx.accum_resize(8, 8)
        with pytest.raises(rt.RtError) as e:
            ctx.dispatch(u)                       # window != accumulation size
        assert e.value.code == -1
        bad = s.indices.copy()
        bad[0] = 10**6
        with pytest.raises(rt.RtError):
            ctx.scene_upload(s.vertices, bad, s.primitives, s.models)


def test_traversal_counters_and_determinism(rt, ctxs):
    """Counting variant leaves the image unchanged; two runs agree bit for bit."""
    cfg = config("C3")
    w, h = 160, 90
    ctx = ctxs("teapot-circle")
    u = uniforms_for(rt, cfg, w, h, samples=4)
    ctx.accum_resize(w, h)
    ctx.dispatch(u)
    a = ctx.accum_download()
    ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1)
    ctx.accum_resize(w, h)
    ctx.stats_reset()
    ctx.dispatch(u)
    b = ctx.accum_download()
    st = ctx.stats()
    ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 0)
    assert_images_identical(a, b, "count variant")
    assert st.node_visits > st.rays and st.triangle_tests > 0


def test_full_size_c3_frame_properties(rt, ctxs):
    """BASELINE config C3 at full 4K size, one 16-spp frame: size-independent properties."""
    cfg = config("C3")
    w, h = cfg["width"], cfg["height"]
    ctx = ctxs("teapot-circle")
    ctx.set_option(rt.RT_OPT_AOV, 0)
    ctx.accum_resize(w, h)
    ctx.stats_reset()
    u = uniforms_for(rt, cfg, w, h)
    ctx.dispatch(u)
    st = ctx.stats()
    img = ctx.accum_download()
    ctx.set_option(rt.RT_OPT_AOV, 1)
    assert st.pixel_frames == w * h
    assert w * h * 16 <= st.rays <= w * h * 16 * 9          # 1..max_bounces+1 segments per path
    assert np.isfinite(img).all() and img[..., :3].min() >= 0.0 and img[..., :3].max() <= 1.0
    assert (img[..., 3] == 0).all()                          # the pad lane of array<vec3f> is never written
    # idempotence of the accumulate: blending the same frame again with weight 1/2 changes nothing
    u2 = u.copy()
    u2["accumulation_frame"] = 1
    ctx.dispatch(u2)
    again = ctx.accum_download()
    assert np.abs(again - img).max() <= 6e-8


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_headless_cli_capture_matches_oracle(rt, oracle_mod, oracle_scenes, tmp_path, devices):
    """rt_cli = the reference's frame loop (src/app.rs:74-102) + Capture (src/ui.rs:78-94), headless: its PNG
    must hold exactly the oracle's accumulation buffer pushed through `(x*255) as u8`.  With --devices the C++ host drives
    an rt_group (three shards on the one GPU here) through the same calls: same PNG."""
    import struct
    import subprocess
    import zlib
    cfg = config("C1")
    w, h, spp, frames = 40, 24, 3, 2
    out = str(tmp_path / "cli.png")
    exe = os.path.join(ROOT, "ray-tracing_amd", "rt_cli")
    pos = ",".join(str(v) for v in cfg["position"])
    r = subprocess.run([exe, "--scene", scene_path("cornell-box"), "--size", f"{w}x{h}", "--samples", str(spp), "--frames", str(frames),
                        "--bounces", str(cfg["max_bounces"]), "--pos", pos, "--yaw", repr(cfg["yaw"]), "--pitch", "0", "--out", out]
                       + (["--devices", devices] if devices else [])
                       + (["--node-format", os.environ["RT_AMD_NODE_FORMAT"]] if os.environ.get("RT_AMD_NODE_FORMAT") else []),   # the 8-ary child runs
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    accum, _ = oracle_scenes("cornell-box").render_frames(uniforms_for(oracle_mod, cfg, w, h, samples=spp), frames, mode=1)
    want = oracle_mod.capture_rgb8(accum).reshape(h, w, 3)
    data = open(out, "rb").read()
    pos_, idat = 8, b""
    while pos_ < len(data):
        n, typ = struct.unpack(">I4s", data[pos_:pos_ + 8])
        if typ == b"IDAT":
            idat += data[pos_ + 8:pos_ + 8 + n]
        pos_ += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 3 * w)
    assert np.array_equal(rows[:, 1:].reshape(h, w, 3), want)


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_contract_and_sharded_image(ranks):
    """bench.py end to end: prints ONE metric line with roofline fields, and (--verify) the image gathered from
    `ranks` band shards equals the unsharded render bit for bit.  ranks=2 is a rehearsal on one GPU: both
    ranks render on the same card and the collectives run over gloo (RT_BENCH_BACKEND), everything else —
    sharding, stream ordering, counters, the gather and the de-interleave — is the multi-GPU code path."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RT_BENCH_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--verify",
           "--cpu-sample", "240x135", "--cpu-timed", "160x90"]
    # ranks > 1: `python bench.py --gpus N` as typed — bench.py starts its own ranks (no torch.distributed.run around it)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    metric = [l for l in lines if "metric" in l]
    verify = [l for l in lines if "verify" in l]
    assert len(metric) == 1 and len(verify) == 1 and verify[0]["identical"] is True
    m = metric[0]
    assert m["n_gpus"] == ranks and m["steps"] == 2 and m["unit"] == "Mrays/s" and m["value"] > 0
    assert m["config"]["rays"] > 3840 * 2160 * 16 * 2
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic"):
        assert key in m["roofline"]
    prov = m["roofline"]["provenance"]
    assert len(prov["kernel_source_sha256"]) == 64 and isinstance(prov["stale"], bool)
    if not prov["stale"]:                            # the recorded traffic is of THIS build's device sources
        assert 0.0 < m["roofline"]["frac"] <= 1.0    # measured HBM traffic against the HBM peak: a fraction that can bind
    else:
        assert "stale" in m["roofline"]["achieved_is"]
    if "onchip" in m["roofline"]:
        oc = m["roofline"]["onchip"]
        assert isinstance(oc["stale"], bool) and oc["cus"] >= 1 and oc["clock_hz"] > 1e8
        if not oc["stale"]:                          # no clamp anywhere: a fraction above 1 would mean a peak that is not one
            assert 0.0 < oc["frac"] <= 1.0 and 0.0 < oc["valu"]["frac"] <= 1.0
    if ranks == 1:                                   # the CPU leg runs on rank 0 at N = 1 only
        assert m["cpu_baseline"]["kind"] == "port" and m["cpu_baseline"]["value"] > 0
        assert 1 <= m["cpu_baseline"]["cores"] <= (os.cpu_count() or 1) and m["cpu_baseline"]["cpu_model"]
        assert "best of 3" in m["cpu_baseline"]["sample"] and m["cpu_baseline"]["lavapipe_probe"]["usable"] is False
        assert m["parity"]["identical"] is True and m["parity"]["rays_equal"] is True and m["parity"]["brute_force_pixels_identical"] is True
    else:
        assert "cpu_baseline" not in m


def test_bench_group_mode_one_process_n_shards():
    """bench.py --mode group: ONE process, N shards behind one rt_group (here the one GPU named twice), one scene build, the
    peer-copy gather inside the timed region, the same JSON schema; --verify: the gathered image equals the unsharded render."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RT_BENCH_GROUP_DEVICES="0,0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "group", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    metric = [l for l in lines if "metric" in l]
    verify = [l for l in lines if "verify" in l]
    assert len(metric) == 1 and len(verify) == 1 and verify[0]["identical"] is True and verify[0]["mode"] == "group"
    m = metric[0]
    assert m["n_gpus"] == 2 and m["steps"] == 2 and m["value"] > 0 and m["scaling"] == "strong"
    assert m["config"]["mode"] == "group" and m["config"]["devices"] == [0, 0] and m["config"]["host_builds"] == 1
    assert m["config"]["peer_access"] == ["same device", "same device"]
    assert m["config"]["rays"] > 3840 * 2160 * 16 * 2
    assert "roofline" in m and "cpu_baseline" not in m


def test_models_update_refits_bvh(rt, oracle_mod, scenes):
    """rt_models_update ≙ App::upload_models (src/app.rs:46-62): new transforms refit the existing tree
    (acceleration_structure.update()); the image still equals the oracle's for the moved geometry."""
    cfg = config("C2")
    w, h = 80, 45
    s = scenes("teapot")
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    moved = ident.copy()
    moved[0] = [0.7, 0.2, 0, 0.5, -0.2, 0.9, 0, 0.3, 0, 0, 1.2, -0.4]
    moved[2] = [1, 0, 0, 0.2, 0, 1, 0, -0.1, 0, 0, 1, 0.3]
    want, want_hits, _ = oracle_mod.OracleScene(s, transforms=moved).render(uniforms_for(oracle_mod, cfg, w, h, samples=3), mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models, transforms=ident)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        for refit in (1, 0):
            ctx.set_option(rt.RT_OPT_REFIT, refit)
            ctx.models_update(s.models, transforms=ident)
            ctx.models_update(s.models, transforms=moved)
            ctx.accum_resize(w, h)
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=3))
            assert np.array_equal(ctx.aov_download(), want_hits)
            assert_images_identical(ctx.accum_download(), want, f"refit={refit}")
        st = ctx.stats()
        assert st.bvh_refits == 2 and st.bvh_rebuilds == 3


def test_app_render_matches_oracle_frames(rt, oracle_mod, oracle_scenes):
    """rt_app_render ≙ App::render (src/app.rs:74-102) driven headless: after the window has settled the
    accumulation buffer equals the oracle's running mean over the same (frame, accumulation_frame) pairs."""
    cfg = config("C1")
    app = rt.App(64, 48, now=0.0)
    app.state["screen_fraction"] = 1
    edited = app.uniform.copy()
    edited["position"][0] = cfg["position"]
    edited["yaw"], edited["pitch"] = np.float32(cfg["yaw"]), np.float32(cfg["pitch"])
    edited["samples"], edited["max_bounces"] = 2, 3
    app.set_params(edited, 0.0)
    with rt.Context(0) as ctx:
        ctx.scene_finish(rt.HostScene(scene_path("cornell-box")))
        seen = []
        for t in [0.0, 0.5, 1.0, 1.1, 1.2, 1.3]:      # the first frames render at 1/4, 1/2 size, then 64x48 settles at t=1.0
            app.render(ctx, t)
            u = app.uniform[0]
            seen.append((tuple(int(x) for x in u["window"]), int(u["frame"]), int(u["accumulation_frame"])))
        got = ctx.accum_download()
    assert seen[-4:] == [((64, 48), 3, 0), ((64, 48), 4, 1), ((64, 48), 5, 2), ((64, 48), 6, 3)]
    ou = oracle_mod.make_uniform(64, 48, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=3, samples=2,
                                 aspect=np.float32(64) / np.float32(48))
    acc = np.zeros((48, 64, 4), np.float32)
    for frame, af in [(3, 0), (4, 1), (5, 2), (6, 3)]:
        ou["frame"], ou["accumulation_frame"] = frame, af
        oracle_scenes("cornell-box").render(ou, acc, mode=1)
    assert_images_identical(got, acc, "app frames")


@pytest.mark.parametrize("world,w,h,spp,frames,stage_kib", [(1, 96, 54, 2, 7, 0), (3, 61, 37, 1, 5, 0), (1, 40, 24, 0, 3, 0),
                                                             (1, 64, 40, 1, 9, 100)])
def test_multi_frame_launch_equals_frame_by_frame(rt, oracle_mod, oracle_scenes, world, w, h, spp, frames, stage_kib):
    """rt_dispatch_frames renders its frames in ONE launch (frame-major queue + per-pixel ordering flags,
    RT_OPT_FRAME_PIPELINE).  Same bits as one launch per frame and as the oracle's frame loop (src/app.rs:78-96),
    twice in a row, also on a shard, and with a staging budget so small that the run splits into launches of two frames."""
    cfg = config("C3")
    rank = world - 1
    def run(pipeline):
        ctx = rt.Context(0)
        ctx.set_option(rt.RT_OPT_FRAME_PIPELINE, pipeline)
        if stage_kib:
            ctx.set_option(109, stage_kib)          # 64*40 pixels * 16 B = 40 KiB per frame -> 2 frames per launch
        ctx.scene_finish(rt.HostScene(scene_path("teapot-circle")))
        ctx.set_shard(rank, world)
        ctx.accum_resize(w, h)
        if pipeline:
            ctx.frames_reserve(frames)              # staging allocated here, not inside the dispatch
        u = uniforms_for(rt, cfg, w, h, samples=spp)
        u["frame"], u["accumulation_frame"] = 0, 1
        ctx.dispatch_frames(u, frames)
        ctx.dispatch_frames(u, 2)
        assert (int(u["frame"][0]), int(u["accumulation_frame"][0])) == (frames + 2, frames + 1)
        st = ctx.stats()
        img = ctx.accum_download()
        ctx.close()
        return img, st
    a, st_a = run(1)
    b, st_b = run(0)
    assert st_a.rays == st_b.rays and st_a.pixel_frames == st_b.pixel_frames and st_a.dispatches == st_b.dispatches == frames + 2
    assert st_b.launches == frames + 2 and st_a.launches == (2 if not stage_kib else (frames + 1) // 2 + 1)
    assert st_a.stack_entries == st_b.stack_entries >= 2 and st_a.resident_blocks > 0
    assert_images_identical(a, b, "pipelined vs frame-by-frame")
    if world == 1:
        want, cnt = oracle_scenes("teapot-circle").render_frames(uniforms_for(oracle_mod, cfg, w, h, samples=spp), frames + 2, mode=1)
        assert st_a.rays == cnt.rays
        assert_images_identical(a, want, "pipelined vs oracle")


@pytest.mark.parametrize("world,w,h,kernel,cuts", [(1, 96, 54, 2, (0, 1, 4, 6)), (1, 61, 37, 0, (0, 3, 4)), (3, 61, 83, 2, (0, 1, 2)), (1, 8, 8, 2, (0,))])
def test_frame_in_pieces_equals_one_dispatch(rt, oracle_mod, oracle_scenes, world, w, h, kernel, cuts):
    """rt_dispatch_bands (the reference's open item TODO.md:9, "Execute compute shader in chunks when in interactive mode"): a frame
    rendered as pieces of 8-row bands — in reverse order, with an empty piece, the last one open-ended, twice in a row (the second
    frame accumulates onto the first, main.wgsl:30-32) — holds the bits and the ray count of rt_dispatch; also on a shard (its own
    bands, numbered from 0) and with the queue-fed baseline kernel.  Kernel variant 1 has no queue: refused."""
    cfg = config("C3")
    rank = world - 1
    def run(pieces):
        ctx = rt.Context(0)
        ctx.set_option(rt.RT_OPT_KERNEL, kernel)
        ctx.scene_finish(rt.HostScene(scene_path("teapot-circle")))
        ctx.set_shard(rank, world)
        ctx.accum_resize(w, h)
        launches = 0
        for f in (0, 1):
            u = uniforms_for(rt, cfg, w, h, samples=2)
            u["frame"], u["accumulation_frame"] = 11 + f, f
            if not pieces:
                ctx.dispatch(u)
                launches += 1
                continue
            ends = list(cuts[1:]) + [0xFFFFFFFF]
            for b0, b1 in reversed(list(zip(cuts, ends))):
                ctx.dispatch_bands(u, b0, b1)
                launches += 1
            ctx.dispatch_bands(u, cuts[-1], cuts[-1])            # empty: nothing is rendered, nothing is launched
            ctx.dispatch_bands(u, 1000, 2000)                    # beyond the last band: empty too
        st = ctx.stats()
        img = ctx.accum_download()
        if pieces:
            with pytest.raises(rt.RtError):
                ctx.dispatch_bands(u, 2, 1)
            ctx.set_option(rt.RT_OPT_KERNEL, 1)
            with pytest.raises(rt.RtError):
                ctx.dispatch_bands(u, 0, 1)
        ctx.close()
        return img, st, launches
    a, st_a, n_a = run(True)
    b, st_b, n_b = run(False)
    assert st_a.rays == st_b.rays and st_a.pixel_frames == st_b.pixel_frames
    assert st_a.dispatches == st_b.dispatches == 2 and st_a.launches == n_a and st_b.launches == n_b == 2
    assert_images_identical(a, b, "frame in pieces vs one dispatch")
    if world == 1:
        acc = np.zeros((h, w, 4), np.float32)
        rays = 0
        for f in (0, 1):
            u = uniforms_for(oracle_mod, cfg, w, h, samples=2)
            u["frame"], u["accumulation_frame"] = 11 + f, f
            _, _, cnt = oracle_scenes("teapot-circle").render(u, acc, mode=1)
            rays += int(cnt.rays)
        assert st_a.rays == rays
        assert_images_identical(a, acc, "frame in pieces vs oracle")


def _soup(obj_loader, rng, n, centre, spread, size):
    """n random triangles around `centre`: corners within `size` of a point drawn within `spread`."""
    c = centre + (rng.random((n, 1, 3)) - 0.5) * 2 * spread
    p = (c + (rng.random((n, 3, 3)) - 0.5) * 2 * size).astype(np.float32)
    v = np.zeros(3 * n, obj_loader.VERTEX_DTYPE)
    v["position"] = p.reshape(-1, 3)
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    v["normal"] = np.repeat(nrm.astype(np.float32), 3, axis=0)
    pr = np.zeros(1, obj_loader.PRIMITIVE_DTYPE)
    pr[0] = (0, 3 * n, 0, 3 * n, 0)
    md = np.zeros(1, obj_loader.MODEL_DTYPE)
    md["diffuse_color"] = (0.7, 0.6, 0.5)
    md["specular_color"] = (0.9, 0.9, 0.9)
    md["specular_probability"] = 0.3
    return obj_loader.Scene(v, np.arange(3 * n, dtype=np.uint32), pr, md)


@pytest.mark.parametrize("case", ["far_from_origin", "huge_extent_ratio", "axis_parallel_rays", "flat_sheet"])
def test_box_test_stays_conservative(rt, oracle_mod, obj_loader, case):
    """The ray/box test is the one part outside the arithmetic contract (f16 child offsets, v_rcp, 5 stuffed
    mantissa bits): it must never cull a hit that BRUTE FORCE finds.  Scenes built to strain it: small triangles
    10^4 units from the origin (cancellation in plane*inv - o*inv), a 10^6:1 extent ratio (offsets at the bottom
    of the f16 range next to offsets at the top), a camera looking exactly along an axis through an axis-aligned
    grid (direction components of exactly 0), all triangles in one plane (zero-thickness boxes)."""
    rng = np.random.default_rng(11)
    w, h, spp, bounces = 64, 40, 2, 4
    pos, yaw, pitch = (0.0, 0.0, -6.0), np.pi / 2, 0.0          # yaw pi/2 looks along +z (ray.wgsl:52-58)
    if case == "far_from_origin":
        centre = np.array([10000.0, -7000.0, 12000.0])
        s = _soup(obj_loader, rng, 3000, centre, 2.0, 0.25)
        pos = tuple(centre + np.array([0.0, 0.0, -6.0]))
    elif case == "huge_extent_ratio":
        a = _soup(obj_loader, rng, 2500, np.zeros(3), 1.5, 0.15)
        b = _soup(obj_loader, rng, 40, np.zeros(3), 2.0e4, 3.0e3)
        v = np.zeros(len(a.vertices) + len(b.vertices), obj_loader.VERTEX_DTYPE)   # (concatenate would drop the padding)
        v[:len(a.vertices)] = a.vertices
        v[len(a.vertices):] = b.vertices
        pr = a.primitives.copy()
        pr[0] = (0, len(v), 0, len(v), 0)
        s = obj_loader.Scene(v, np.arange(len(v), dtype=np.uint32), pr, a.models)
    elif case == "axis_parallel_rays":
        # unit quads on the integer grid in the planes z = 0..3, camera on a grid line, fov tiny: primary rays run
        # (almost) exactly along +z, many with dx = dy = 0 after the jitter-free centre pixel
        quads = []
        for z in range(4):
            for x in range(-3, 3):
                for y in range(-3, 3):
                    if (x + y + z) % 2 == 0:
                        quads.append([[x, y, z], [x + 1, y, z], [x + 1, y + 1, z], [x, y, z], [x + 1, y + 1, z], [x, y + 1, z]])
        p = np.array(quads, np.float32).reshape(-1, 3)
        v = np.zeros(len(p), obj_loader.VERTEX_DTYPE)
        v["position"] = p
        v["normal"] = (0, 0, -1)
        pr = np.zeros(1, obj_loader.PRIMITIVE_DTYPE)
        pr[0] = (0, len(p), 0, len(p), 0)
        md = np.zeros(1, obj_loader.MODEL_DTYPE)
        md["diffuse_color"] = (0.8, 0.8, 0.8)
        md["specular_probability"] = 1.0          # mirror: reflected rays stay axis-parallel
        md["specular_color"] = (1, 1, 1)
        s = obj_loader.Scene(v, np.arange(len(p), dtype=np.uint32), pr, md)
    else:
        s = _soup(obj_loader, rng, 600, np.zeros(3), 2.0, 0.3)
        s.vertices["position"][:, 2] = 1.25       # everything in the plane z = 1.25
        s.vertices["normal"] = (0, 0, -1)
    kw = dict(max_bounces=bounces, samples=spp)
    if case == "axis_parallel_rays":
        kw["fov"] = 1e-4
    want, want_hits, cnt = oracle_mod.OracleScene(s, build_bvh=False).render(
        oracle_mod.make_uniform(w, h, pos, yaw, pitch, **kw), mode=0, want_hits=True)
    assert (want_hits[..., 0] != 0xFFFFFFFF).mean() > 0.05, "the camera must see the scene for the test to mean anything"
    for stride in (3, 4, 2):                     # 2: the 32-byte node format (8-bit plane offsets), RT_OPT_NODE_FORMAT
        with rt.Context(0) as ctx:
            if stride == 2:
                ctx.set_option(rt.RT_OPT_NODE_FORMAT, 2)
            else:
                ctx.set_option(rt.RT_OPT_NODE_FORMAT, 1)
                ctx.set_option(rt.RT_OPT_NODE_STRIDE, stride)
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            ctx.set_option(rt.RT_OPT_AOV, 1)
            ctx.accum_resize(w, h)
            ctx.dispatch(rt.make_uniform(w, h, pos, yaw, pitch, **kw))
            assert ctx.stats().rays == cnt.rays
            assert np.array_equal(ctx.aov_download(), want_hits), f"{case}: primary hit ids differ from brute force"
            assert_images_identical(ctx.accum_download(), want, f"{case} stride {stride}")


@pytest.mark.parametrize("kw", [dict(max_bounces=0, samples=3), dict(max_bounces=1, samples=1), dict(max_bounces=20, samples=2),
                                dict(max_bounces=4, samples=2, exposure=0.25, environment=3.0),
                                dict(max_bounces=3, samples=2, fov=0.4), dict(max_bounces=3, samples=2, frame=4000000000, accumulation_frame=7)])
def test_uniform_fields_are_honoured(rt, oracle_mod, oracle_scenes, ctxs, kw):
    """Every Uniform field the shader reads (shaders/types.wgsl:1-12): bounce budget 0 / 1 / beyond the reference UI's
    range, exposure and environment, field of view, a frame counter near u32 wrap with a non-zero accumulation
    frame (the blend weight 1/(n+1), main.wgsl:30-32, applied to a non-empty buffer)."""
    cfg = config("C1")
    w, h = 48, 40
    ctx = ctxs("cornell-box")
    ctx.accum_resize(w, h)
    ctx.stats_reset()
    first = dict(max_bounces=2, samples=1)                       # something in the buffer for the blend to act on
    ctx.dispatch(uniforms_for(rt, cfg, w, h, **first))
    ctx.dispatch(uniforms_for(rt, cfg, w, h, **kw))
    got, st = ctx.accum_download(), ctx.stats()
    osc = oracle_scenes("cornell-box")
    want, _, c1 = osc.render(uniforms_for(oracle_mod, cfg, w, h, **first), mode=1)
    want, _, c2 = osc.render(uniforms_for(oracle_mod, cfg, w, h, **kw), accum=want, mode=1)
    assert st.rays == c1.rays + c2.rays
    assert_images_identical(got, want, str(kw))
    # Capture (src/ui.rs:82-93) on the device == the host conversion of the downloaded buffer == the oracle's
    rgb = ctx.capture_rgb8()
    assert np.array_equal(rgb.reshape(-1, 3), oracle_mod.capture_rgb8(want)) and np.array_equal(rgb.reshape(-1, 3), rt.capture_rgb8(got))


@pytest.mark.parametrize("node_format", [1, 2, 3, 4])
def test_models_update_device_refit(rt, oracle_mod, obj_loader, scenes, node_format):
    """RT_OPT_REFIT = 2: rt_models_update recomputes the triangle records and the node boxes ON the device
    (csrc/rt_refit.hip).  Same hit ids and image as the oracle for the moved geometry and as the host refit,
    over several updates in a row (the arrival counters are reused), with a material edit riding along, on a
    scene whose transforms were NOT given at upload, and on a scene with spheres in the tree."""
    cfg = config("C2")
    w, h = 80, 45
    s = scenes("teapot")
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    moves = []
    for k in range(3):
        m = ident.copy()
        m[0] = [0.7 + 0.1 * k, 0.2, 0, 0.5, -0.2, 0.9, 0, 0.3 - 0.2 * k, 0, 0, 1.2, -0.4]
        m[2] = [1, 0, 0, 0.2 * k, 0, 1, 0, -0.1, 0, 0, 1, 0.3]
        moves.append(m)
    models = s.models.copy()
    models["diffuse_color"][0] = (0.9, 0.1, 0.1)                        # upload_models also re-uploads the materials
    edited = obj_loader.Scene(s.vertices, s.indices, s.primitives, models)
    u_o, u_r = uniforms_for(oracle_mod, cfg, w, h, samples=3), uniforms_for(rt, cfg, w, h, samples=3)
    with rt.Context(0) as dev, rt.Context(0) as host:
        for ctx, mode in ((dev, 2), (host, 1)):
            ctx.set_option(rt.RT_OPT_REFIT, mode)                       # before the upload: it keeps the topology on the device
            ctx.set_option(rt.RT_OPT_NODE_FORMAT, node_format)          # 48-byte / 32-byte binary nodes, 80-byte 8-ary nodes
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)   # no transforms at upload
            ctx.set_option(rt.RT_OPT_AOV, 1)
        for k, m in enumerate(moves):
            want, want_hits, cnt = oracle_mod.OracleScene(edited, transforms=m).render(u_o, mode=1, want_hits=True)
            imgs = []
            for ctx in (dev, host):
                ctx.models_update(models, transforms=m)
                ctx.accum_resize(w, h)
                ctx.stats_reset()
                ctx.dispatch(u_r)
                assert ctx.stats().rays == cnt.rays
                assert np.array_equal(ctx.aov_download(), want_hits), f"update {k}: primary hit ids differ"
                imgs.append(ctx.accum_download())
            assert_images_identical(imgs[0], want, f"device refit, update {k}")
            assert_images_identical(imgs[1], imgs[0], f"host refit vs device refit, update {k}")
        assert dev.stats().bvh_refits == 3 and dev.stats().bvh_rebuilds == 1
    # spheres share the tree and do not move; brute force is the judge here
    import __graft_entry__ as graft
    graft.load_package()
    from ray_tracing_amd import synth
    syn = synth.lattice_scene(scene_path("teapot"), 7, 120, seed=3)
    cam = syn.camera
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(syn.primitives), 1))
    moved = ident.copy()
    moved[:, 3] += 0.35
    moved[:, 0] = 1.1
    kw = dict(max_bounces=4, samples=2)
    osc = oracle_mod.OracleScene(obj_loader.Scene(syn.vertices, syn.indices, syn.primitives, syn.models), transforms=moved, spheres=syn.spheres)
    want, want_hits, cnt = osc.render(oracle_mod.make_uniform(64, 36, cam["position"], cam["yaw"], cam["pitch"], **kw), mode=0, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_REFIT, 2)
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, node_format)
        ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        ctx.spheres_upload(syn.spheres)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.models_update(syn.models, transforms=moved)
        assert ctx.stats().bvh_refits == 1
        ctx.accum_resize(64, 36)
        ctx.dispatch(rt.make_uniform(64, 36, cam["position"], cam["yaw"], cam["pitch"], **kw))
        assert ctx.stats().rays == cnt.rays
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, "device refit with spheres")


def test_refit_when_model_and_geometry_ranges_differ(rt, oracle_mod, obj_loader, scenes):
    """The BLAS reads corners through the GeometryPrimitive ranges (src/scene.rs:141-147), shading reads attributes
    through the Model ranges (main.wgsl:86-91).  The reference keeps them equal; the interface does not have to.  Here
    the models point at a second, displaced copy of the vertices: the device refit (which rebuilds triangles from
    the shading records) must step aside for the host path, and the image must still equal the oracle's."""
    cfg = config("C2")
    w, h = 64, 36
    s = scenes("teapot")
    n = len(s.vertices)
    v2 = np.zeros(2 * n, obj_loader.VERTEX_DTYPE)
    v2[:n] = s.vertices
    v2[n:] = s.vertices
    v2["position"][n:] += np.float32(0.125)
    models = s.models.copy()
    models["vertex_start"] += n
    scene = obj_loader.Scene(v2, s.indices, s.primitives, models)
    moved = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    moved[0] = [0.8, 0, 0, 0.3, 0, 1.1, 0, 0.1, 0, 0, 1, -0.2]
    want, want_hits, cnt = oracle_mod.OracleScene(scene, transforms=moved).render(uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.scene_upload(v2, s.indices, s.primitives, models)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.models_update(models, transforms=moved)
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
        assert ctx.stats().rays == cnt.rays
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, "model ranges != geometry ranges")


def test_degenerate_windows_and_frame_counts(rt):
    """Zero-width / zero-height windows, zero frames: nothing to render is not an error (the reference's bounds check,
    main.wgsl:17, simply retires every invocation)."""
    with rt.Context(0) as ctx:
        ctx.scene_finish(rt.HostScene(scene_path("cornell-box")))
        for (w, h) in [(0, 0), (0, 5), (7, 0), (1, 1)]:
            ctx.accum_resize(w, h)
            ctx.stats_reset()
            u = rt.make_uniform(w, h, (0, 15, 14), 4.712, max_bounces=2, samples=1, aspect=1.0)
            ctx.dispatch(u)
            ctx.dispatch_frames(u, 0)
            ctx.dispatch_frames(u, 3)
            ctx.sync()
            assert ctx.accum_download().shape == (h, w, 4) and ctx.capture_rgb8().shape == (h, w, 3)
            assert ctx.stats().pixel_frames == 4 * w * h


@pytest.mark.gpu
@pytest.mark.parametrize("n_geo", [16, 17, 40])
def test_material_records_in_lds_or_memory(rt, oracle_mod, obj_loader, n_geo):
    """The default kernel keeps up to 16 material records (RT_LDS_MATERIALS) in LDS and reads them from memory
    beyond that: both instances against the oracle's brute force on a scene of n_geo small soups, each with its own
    metal or dielectric material."""
    rng = np.random.default_rng(100 + n_geo)
    n = 60
    V = np.zeros(3 * n * n_geo, obj_loader.VERTEX_DTYPE)
    P = np.zeros(n_geo, obj_loader.PRIMITIVE_DTYPE)
    M = np.zeros(n_geo, obj_loader.MODEL_DTYPE)
    for g in range(n_geo):
        s = _soup(obj_loader, rng, n, np.array([(g % 8) - 3.5, (g // 8) - 2.0, 0.0]), 0.45, 0.35)
        V[3 * n * g:3 * n * (g + 1)] = s.vertices
        P[g] = (3 * n * g, 3 * n, 3 * n * g, 3 * n, g)
        M[g] = s.models[0]
        M[g]["tag"] = 1 if g % 5 == 4 else 0
        M[g]["diffuse_color"] = rng.random(3)
        M[g]["specular_color"] = rng.random(3)
        M[g]["specular_probability"] = rng.random()
        M[g]["roughness"] = rng.random()
        M[g]["emission_color"] = rng.random(3)
        M[g]["emission_strength"] = 2.0 * (g % 3 == 0)
        M[g]["refractive_index"] = 1.0 + rng.random()
        M[g]["vertex_start"] = 3 * n * g
        M[g]["index_start"] = 3 * n * g
    I = np.tile(np.arange(3 * n, dtype=np.uint32), n_geo)          # indices are object-local (main.wgsl:89-91)
    scene = obj_loader.Scene(V, I, P, M)
    w, h, kw = 72, 44, dict(max_bounces=5, samples=2)
    pos, yaw = (0.0, 0.0, -5.0), np.pi / 2
    want, want_hits, cnt = oracle_mod.OracleScene(scene, build_bvh=False).render(oracle_mod.make_uniform(w, h, pos, yaw, **kw), mode=0, want_hits=True)
    assert len(np.unique(want_hits[..., 0])) > n_geo // 2               # most materials are on screen
    with rt.Context(0) as ctx:
        ctx.scene_upload(V, I, P, M)
        ctx.set_option(rt.RT_OPT_AOV, 1)
        ctx.accum_resize(w, h)
        ctx.dispatch(rt.make_uniform(w, h, pos, yaw, **kw))
        assert ctx.stats().rays == cnt.rays
        assert np.array_equal(ctx.aov_download(), want_hits)
        assert_images_identical(ctx.accum_download(), want, f"{n_geo} materials")
        # a later upload with more / fewer materials switches instance
        ctx.set_option(rt.RT_OPT_AOV, 0)
        ctx.dispatch(rt.make_uniform(w, h, pos, yaw, **kw))
        assert_images_identical(ctx.accum_download(), want, f"{n_geo} materials, no AOV")


@pytest.mark.gpu
def test_node_format_choice_and_limits(rt, oracle_mod, oracle_scenes, scenes):
    """RT_OPT_NODE_FORMAT: 32-byte nodes give the same bits as 48-byte ones (all three kernel variants decode them);
    a scene too big for their 8-bit offsets gets 48-byte nodes all the same; a refit that moves the scene out of
    their range falls back to a host rebuild."""
    cfg = config("C3")
    w, h = 96, 54
    s = scenes("teapot-circle")
    want, _, cnt = oracle_scenes("teapot-circle").render(uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1, want_hits=True)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, 2)
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        small = ctx.stats().bvh_bytes
        for variant in (2, 0, 1):
            ctx.set_option(rt.RT_OPT_KERNEL, variant)
            ctx.accum_resize(w, h)
            ctx.stats_reset()
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
            assert ctx.stats().rays == cnt.rays
            assert_images_identical(ctx.accum_download(), want, f"32-byte nodes, variant {variant}")
        ctx.set_option(rt.RT_OPT_KERNEL, 2)
        # blow the scene up past 2^18 units: the device refit must notice and rebuild with 48-byte nodes
        big = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
        big[:, 0] = big[:, 5] = big[:, 10] = 3.0e5
        ctx.models_update(s.models, transforms=big)
        assert ctx.stats().bvh_bytes > small
        ctx.accum_resize(8, 8)
        ctx.dispatch(uniforms_for(rt, cfg, 8, 8, samples=1))
        assert np.isfinite(ctx.accum_download()).all()
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, 1)
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        assert ctx.stats().bvh_bytes > small                      # 48-byte nodes
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, 0)
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        assert ctx.stats().bvh_bytes > small                      # by size: a 0.3 MB tree keeps the 48-byte format
    v = s.vertices.copy()
    v["position"] *= np.float32(1.0e5)
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, 2)
        ctx.scene_upload(v, s.indices, s.primitives, s.models)
        assert ctx.stats().bvh_bytes > small                      # extent 2e7 > 2^18: 48-byte nodes


@pytest.mark.gpu
def test_models_update_with_more_models_than_uploaded(rt, scenes):
    """rt_models_update may bring MORE models than the upload had (only the first ones are referenced by geometries):
    the material array on the device must grow (host path instead of the device refit) and the image stays the same."""
    cfg = config("C2")
    w, h = 64, 36
    s = scenes("teapot")
    with rt.Context(0) as ctx:
        ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
        before = ctx.accum_download()
        more = np.zeros(len(s.models) + 20, s.models.dtype)
        more[:len(s.models)] = s.models
        more["diffuse_color"][len(s.models):] = (1.0, 0.0, 1.0)
        ctx.models_update(more)
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
        assert_images_identical(ctx.accum_download(), before, "more models than uploaded")


@pytest.mark.gpu
def test_models_update_changed_ranges_and_rejected_update(rt, oracle_mod, obj_loader, scenes):
    """rt_models_update with a model whose vertex_start changed AFTER the upload: the device refit would keep shading with
    the records baked at upload, so the call must take the host path — both refit modes give the oracle's image for the
    new models.  And an update the library rejects (ranges beyond the buffers) must leave the uploaded scene in force."""
    cfg = config("C2")
    w, h = 64, 36
    s = scenes("teapot")
    n = len(s.vertices)
    v2 = np.zeros(2 * n, obj_loader.VERTEX_DTYPE)
    v2[:n] = s.vertices
    v2[n:] = s.vertices
    v2["normal"][n:] *= np.float32(-0.5)               # the second copy shades differently
    new_models = s.models.copy()
    new_models["vertex_start"][0] += n
    ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(s.primitives), 1))
    want, _, cnt = oracle_mod.OracleScene(obj_loader.Scene(v2, s.indices, s.primitives, new_models), transforms=ident).render(
        uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1)
    before, _, _ = oracle_mod.OracleScene(obj_loader.Scene(v2, s.indices, s.primitives, s.models), transforms=ident).render(
        uniforms_for(oracle_mod, cfg, w, h, samples=2), mode=1)
    assert not np.array_equal(want, before)
    for refit in (2, 1):
        with rt.Context(0) as ctx:
            ctx.set_option(rt.RT_OPT_REFIT, refit)
            ctx.scene_upload(v2, s.indices, s.primitives, s.models, transforms=ident)
            # rejected: a vertex range past the end of the vertex buffer
            bad = s.models.copy()
            bad["vertex_start"][1] = 2 * n
            with pytest.raises(rt.RtError):
                ctx.models_update(bad, transforms=ident)
            ctx.accum_resize(w, h)
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
            assert_images_identical(ctx.accum_download(), before, f"refit {refit}: scene after a rejected update")
            ctx.models_update(new_models, transforms=ident)
            ctx.accum_resize(w, h)
            ctx.stats_reset()
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=2))
            assert ctx.stats().rays == cnt.rays
            assert_images_identical(ctx.accum_download(), want, f"refit {refit}: model ranges changed by rt_models_update")


@pytest.mark.gpu
def test_set_shard_after_accumulation_buffer(rt, ctxs):
    """rt_set_shard on a context that already has an accumulation buffer drops that buffer instead of failing; after a new
    rt_accum_resize the context renders the new shard."""
    cfg = config("C1")
    w, h = 40, 40
    ctx = ctxs("cornell-box")
    try:
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=1))
        full = ctx.accum_download()
        ctx.set_shard(1, 2)
        with pytest.raises(rt.RtError):                       # no buffer any more
            ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=1))
        ctx.accum_resize(w, h)
        ctx.dispatch(uniforms_for(rt, cfg, w, h, samples=1))
        part = ctx.accum_download()
        rows = [y for y in range(h) if (y // 8) % 2 == 1]
        assert part.shape[0] == len(rows)
        assert_images_identical(part, full[rows], "shard 1 of 2 after re-sharding")
    finally:
        ctx.set_shard(0, 1)


@pytest.mark.skipif(os.environ.get("RT_AMD_NODE_FORMAT") is not None, reason="once is enough: not a node-format matter")
def test_bench_rccl_path_with_one_rank():
    """The nccl (= RCCL) branch of bench.py on the one GPU there is: RT_BENCH_FORCE_DIST=1 makes a one-rank run create the process group
    (backend nccl, device_id), pass the barriers, reduce its timings and counters and issue the framebuffer gather as a collective, with the
    render kernels and the collective ordered on one stream.  More than one rank needs more than one GPU; this is what can run here."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "RT_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu", "--verify", "--config", "C2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=110, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    metric = [l for l in lines if "metric" in l]
    verify = [l for l in lines if "verify" in l]
    assert len(metric) == 1 and metric[0]["n_gpus"] == 1 and metric[0]["value"] > 0
    assert len(verify) == 1 and verify[0]["identical"] is True
