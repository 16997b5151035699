"""Host half of the library (CPU only): C-ABI exports, scene loader, BVH builder, capture/PNG."""
import ctypes
import os
import re
import struct
import zlib

import numpy as np
import pytest

from conftest import ROOT, scene_path


def test_library_exports_every_declared_symbol(rt):
    header = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    declared -= {"rt_context", "rt_host_scene"}
    assert declared == set(rt.EXPORTS), declared ^ set(rt.EXPORTS)
    L = ctypes.CDLL(rt.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name


def test_no_gpu_reports_error_not_crash(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.RtError) as e:
        rt.Context(0)
    assert e.value.code == -2


@pytest.mark.parametrize("name", ["cornell-box", "teapot", "teapot-circle"])
def test_cpp_loader_matches_python_restatement(rt, scenes, name):
    hs = rt.HostScene(scene_path(name))
    s = scenes(name)
    for k in ["vertices", "indices", "primitives", "models"]:
        assert np.array_equal(getattr(hs, k).view(np.uint8), getattr(s, k).view(np.uint8)), k
    assert hs.names == s.names


def _write(tmp_path, obj, mtl="newmtl m\nKd 0.1 0.2 0.3\nKe 3 0 4\nPm 0.25\nPr 0.5\n"):
    (tmp_path / "s.mtl").write_text(mtl)
    p = tmp_path / "s.obj"
    p.write_text("mtllib s.mtl\n" + obj)
    return str(p)


def test_loader_ragged_inputs(rt, obj_loader, tmp_path):
    # n-gon fan, negative indices, missing vt, shared corners de-duplicated in first-use order, two objects
    obj = """o A
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0.5 1.5 0
vn 0 0 1
usemtl m
f 1//1 2//1 3//1 4//1 5//1
o B
v 0 0 1
v 1 0 1
v 0 1 1
vt 0.25 0.75
usemtl m
f -3/1/1 -2/1/1 -1/1/1
"""
    path = _write(tmp_path, obj)
    hs = rt.HostScene(path)
    s = obj_loader.load(path)
    for k in ["vertices", "indices", "primitives", "models"]:
        assert np.array_equal(getattr(hs, k).view(np.uint8), getattr(s, k).view(np.uint8)), k
    assert list(hs.indices) == [0, 1, 2, 0, 2, 3, 0, 3, 4, 0, 1, 2]
    assert list(hs.primitives["first_vertex"]) == [0, 5] and list(hs.primitives["index_count"]) == [9, 3]
    assert (hs.vertices["uv"][:5] == 0).all() and (hs.vertices["uv"][5:] == [0.25, 0.75]).all()
    m = hs.models[0]
    assert m["specular_probability"] == np.float32(0.25) and m["roughness"] == np.float32(0.5)
    assert m["emission_strength"] == np.float32(5.0) and np.allclose(m["emission_color"], [0.6, 0.0, 0.8])
    assert list(hs.models["vertex_start"]) == [0, 5] and list(hs.models["index_start"]) == [0, 9]


def test_loader_errors(rt, tmp_path):
    with pytest.raises(rt.RtError) as e:
        rt.HostScene(str(tmp_path / "missing.obj"))
    assert e.value.code == -7
    # object without a material: material_id.unwrap() panics in the reference (src/scene.rs:149)
    p = _write(tmp_path, "o A\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    with pytest.raises(rt.RtError) as e:
        rt.HostScene(p)
    assert e.value.code == -8 and "material" in str(e.value)
    p = _write(tmp_path, "o A\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 9\n")
    with pytest.raises(rt.RtError):
        rt.HostScene(p)


def test_loader_empty_file(rt, tmp_path):
    p = tmp_path / "e.obj"
    p.write_text("# nothing\n")
    hs = rt.HostScene(str(p))
    assert len(hs.vertices) == 0 and len(hs.indices) == 0 and len(hs.models) == 0


@pytest.mark.parametrize("name", ["cornell-box", "teapot", "teapot-circle"])
def test_bvh_structure_on_scenes(rt, scenes, name):
    s = scenes(name)
    info = rt.bvh_info(s.vertices, s.indices, s.primitives)   # self-checks coverage + containment inside
    assert info.n_triangles == len(s.indices) // 3
    assert info.max_leaf_size <= 4 and info.max_depth <= 48
    assert info.n_nodes == 2 * info.n_leaves - 1


def _soup(rt, obj_loader, n, rng, degenerate=False):
    v = np.zeros(3 * n, obj_loader.VERTEX_DTYPE)
    if degenerate:
        v["position"] = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), (n, 1))
    else:
        c = rng.uniform(-10, 10, (n, 1, 3)).astype(np.float32)
        v["position"] = (c + rng.uniform(-0.3, 0.3, (n, 3, 3)).astype(np.float32)).reshape(-1, 3)
    v["normal"] = [0, 0, 1]
    idx = np.arange(3 * n, dtype=np.uint32)
    pr = np.zeros(1, obj_loader.PRIMITIVE_DTYPE)
    pr[0] = (0, 3 * n, 0, 3 * n, 0)
    return v, idx, pr


@pytest.mark.parametrize("n", [0, 1, 3, 5, 1000, 70000])
def test_bvh_random_soups(rt, obj_loader, n):
    v, idx, pr = _soup(rt, obj_loader, n, np.random.default_rng(n))
    info = rt.bvh_info(v, idx, pr)
    assert info.n_triangles == n and info.max_leaf_size <= 4


def test_bvh_identical_centroids_is_depth_bounded(rt, obj_loader):
    # 5000 coincident triangles: SAH cannot separate them; the median fallback must bound the depth
    v, idx, pr = _soup(rt, obj_loader, 5000, None, degenerate=True)
    info = rt.bvh_info(v, idx, pr)
    assert info.n_triangles == 5000 and info.max_depth <= 48 and info.max_leaf_size <= 4


def test_bvh_rejects_bad_ranges(rt, obj_loader):
    v, idx, pr = _soup(rt, obj_loader, 4, np.random.default_rng(1))
    pr["index_count"] = 13
    with pytest.raises(rt.RtError):
        rt.bvh_info(v, idx, pr)
    pr["index_count"] = 12
    idx[5] = 99
    with pytest.raises(rt.RtError):
        rt.bvh_info(v, idx, pr)


def test_png_roundtrip(rt, tmp_path):
    rng = np.random.default_rng(0)
    w, h = 37, 21
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    path = str(tmp_path / "o.png")
    rt.write_png(path, rgb, w, h)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + body) == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        if typ == b"IDAT":
            idat += body
        pos += 12 + n
    assert ihdr == (w, h, 8, 2, 0, 0, 0)
    raw = zlib.decompress(idat)
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 3), rgb)


def test_default_uniform(rt):
    u = rt.default_uniform()   # src/main.rs:73-84, src/camera.rs:112-123
    assert tuple(u["position"][0]) == (1.0, 1.0, 1.0) and u["fov"][0] == np.float32(np.pi / 2)
    assert u["frame"][0] == 0 and u["accumulation_frame"][0] == 1 and u["max_bounces"][0] == 10
    assert u["samples"][0] == 5 and u["exposure"][0] == 1.0 and u["environment"][0] == 1.0 and u["flags"][0] == 0


def test_cli_without_gpu_fails_cleanly(rt):
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(ROOT, "ray-tracing_amd", "rt_cli")
    r = subprocess.run([exe, "--scene", scene_path("cornell-box"), "--size", "8x8"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "no HIP device" in r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2


def test_app_state_layout(rt):
    import subprocess
    src = '#include "include/rt_abi.h"\nstatic_assert(sizeof(RtAppState) == 112, "");\nstatic_assert(offsetof(RtAppState, inner_size) == 80, "");\nstatic_assert(offsetof(RtAppState, last_invalidation) == 96, "");\nstatic_assert(offsetof(RtAppState, accumulate) == 104, "");\nint main(){}\n'
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", "-I", ROOT, "-"], input=src, text=True, check=True, cwd=ROOT)
    assert rt.APP_DTYPE.itemsize == 112


def test_app_frame_sequence(rt):
    """App::render bookkeeping (src/app.rs:74-96) from a cold start, the reference's invalidation quirk and the
    progressive screen fraction."""
    app = rt.App(1600, 900, now=0.0)
    seq = []
    # first second: dynamic fraction 4 -> 1 dominates screen_fraction 2 while it is larger
    for t in [0.0, 0.2, 0.4, 0.7, 0.8, 1.0, 1.5, 2.0]:
        resized = app.begin_frame(t)
        u = app.uniform[0]
        seq.append((tuple(int(x) for x in u["window"]), int(u["frame"]), int(u["accumulation_frame"]), resized))
    assert seq[0] == ((400, 225), 1, 0, True)            # 1 + 3*(1-0) = 4 -> window / 4; cold start sees (1, 0)
    assert seq[1] == ((533, 300), 2, 0, True)            # 3.4 -> 3
    assert seq[2] == ((800, 450), 3, 0, True)            # 2.8 -> 2 (= screen_fraction)
    assert seq[3] == ((800, 450), 4, 1, False)           # 1.9 -> 1 < screen_fraction 2: size stable, accumulates
    assert [s[2] for s in seq[3:]] == [1, 2, 3, 4, 5]
    assert [s[1] for s in seq] == list(range(1, 9))
    assert abs(float(app.uniform["aspect"][0]) - 1600 / 900) < 1e-6
    # invalidation: accumulation_frame is zeroed, then bumped to 1 by the next render unless the resize branch
    # fires — it does here, because the dynamic fraction jumps back to 4 (src/app.rs:75-76, 87-88)
    app.invalidate(3.0)
    assert app.begin_frame(3.0) is True and int(app.uniform["accumulation_frame"][0]) == 0
    assert tuple(int(x) for x in app.uniform["window"][0]) == (400, 225)
    # with screen_fraction 4 the size never changes, and the quirk shows: first frame after invalidation has weight 1/2
    app2 = rt.App(1600, 900, now=0.0)
    app2.state["screen_fraction"] = 4
    for t in [0.0, 0.5, 2.0]:
        app2.begin_frame(t)
    app2.invalidate(2.5)
    assert app2.begin_frame(2.6) is False and int(app2.uniform["accumulation_frame"][0]) == 1
    # accumulate = false pins accumulation_frame to 0
    app2.state["accumulate"] = 0
    app2.begin_frame(3.0)
    assert int(app2.uniform["accumulation_frame"][0]) == 0


def test_app_set_params_invalidates_on_hashed_fields_only(rt):
    app = rt.App(800, 600, now=0.0)
    for t in [0.0, 1.2, 1.4]:
        app.begin_frame(t)
    edited = app.uniform.copy()
    edited["frame"] = 999                                 # not hashed (src/types.rs:112-121)
    assert app.set_params(edited, 2.0) is False
    edited["exposure"] = 1.5
    assert app.set_params(edited, 2.0) is True and float(app.uniform["exposure"][0]) == 1.5
    assert int(app.uniform["accumulation_frame"][0]) == 0 and float(app.state["last_invalidation"][0]) == 2.0
    assert app.set_params(edited, 2.1) is False
    edited["position"][0] = (3.0, 2.0, 1.0)
    assert app.set_params(edited, 2.2) is True


def test_host_half_under_sanitizers(tmp_path):
    """Scene loader, BVH builder / refit, capture, PNG writer and decoder under ASan + UBSan (host code only;
    tools/host_sanitize.cpp).  Skipped where g++ cannot link the sanitizer runtimes."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    host = os.path.join(ROOT, "ray-tracing_amd", "host")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "host_sanitize.cpp")] + [os.path.join(host, f + ".cpp") for f in
           ("scene_loader", "bvh_builder", "capture", "image_decode")] + ["-lpthread", "-lz", "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if b.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + b.stderr[-300:])
    # damaged inputs: PNGs with flipped bytes, cut short or with absurd headers, OBJ text with lines cut and tokens garbled
    from test_textures import _png_bytes
    rng = np.random.default_rng(3)
    extra = []
    for i in range(40):
        ctype, depth = [(0, 1), (0, 8), (0, 16), (2, 8), (3, 2), (3, 8), (4, 8), (6, 8), (6, 16)][i % 9]
        w, h = int(rng.integers(1, 24)), int(rng.integers(1, 24))
        ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
        plte = rng.integers(0, 256, (1 << min(depth, 4), 3), dtype=np.uint8) if ctype == 3 else None
        px = rng.integers(0, len(plte) if ctype == 3 else 1 << depth, (h, w, ch))
        good = _png_bytes(rng, w, h, ctype, depth, px, plte, None, interlace=int(rng.random() < 0.5))
        for k in range(8):
            data = bytearray(good)
            kind = k % 4
            if kind == 0:
                for _ in range(int(rng.integers(1, 6))):
                    data[int(rng.integers(8, len(data)))] = int(rng.integers(0, 256))
            elif kind == 1:
                data = data[:int(rng.integers(8, len(data)))]
            elif kind == 2:
                data[16:24] = bytes(rng.integers(0, 256, 8, dtype=np.uint8))          # width / height
            else:
                data[24:26] = bytes(rng.integers(0, 17, 2, dtype=np.uint8))           # depth / colour type
            path = tmp_path / f"d{i}_{k}.png"
            path.write_bytes(bytes(data))
            extra.append(str(path))
    obj_lines = open(scene_path("cornell-box")).read().splitlines()
    for i in range(30):
        lines = list(obj_lines[:400])
        for _ in range(int(rng.integers(1, 12))):
            j = int(rng.integers(0, len(lines)))
            kind = int(rng.integers(0, 4))
            lines[j] = [lines[j][:len(lines[j]) // 2], lines[j] + " 1e999 nan -", "f 1/2/3 -99999//7 0 5//", lines[j].replace(" ", "/")][kind]
        path = tmp_path / f"o{i}.bad"
        path.write_text("\n".join(lines))
        extra.append(str(path))
    r = subprocess.run([exe] + [scene_path(n) for n in ("cornell-box", "teapot", "teapot-circle")] + extra, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:verify_asan_link_order=0"))
    if "ASan runtime does not come first" in r.stderr:
        pytest.skip("a preloaded library precedes the ASan runtime here")
    assert r.returncode == 0 and "host sanitize run ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_bench_self_launch_relays_failure():
    """`python bench.py --gpus 2` as typed starts its own two ranks; without a GPU each rank refuses to run and
    the parent must report that with a non-zero exit code (no silent success, no retry)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box this test sees no device
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode != 0
    assert "rank(s) failed" in r.stderr and "needs a GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_does_not_wait_for_a_dead_rank():
    """One rank dies while another waits (here: rank 0 sleeps, as it would inside a collective): the parent stops the
    waiting rank itself and fails within seconds instead of hanging until a collective's timeout."""
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", RT_BENCH_SELFTEST="rank0-waits")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "rank(s) failed" in r.stderr
    assert time.time() - t0 < 60


def test_app_random_event_sequences(rt):
    """The frame driver (host/app.cpp) against a line-by-line Python model of App::render / invalidate_accumulation
    (src/app.rs:41-44,74-96) on random event sequences: frames at random times, invalidations, parameter edits, window
    resizes, screen_fraction and accumulate changes."""
    f32 = np.float32
    rng = np.random.default_rng(5)
    for trial in range(200):
        W, H = int(rng.integers(1, 3000)), int(rng.integers(1, 2000))
        t = float(rng.uniform(0, 10))
        app = rt.App(W, H, now=t)
        m = dict(inner=[W, H], last_window=[0, 0], last_inv=t, accumulate=1, fraction=2, frame=0, acc=1)       # src/main.rs:63-92
        for step in range(40):
            t += float(rng.choice([0.0, rng.uniform(0, 0.3), rng.uniform(0, 3)]))
            ev = rng.random()
            if ev < 0.1:
                app.invalidate(t)
                m["last_inv"], m["acc"] = t, 0                                        # src/app.rs:41-44
            elif ev < 0.2:
                m["fraction"] = int(rng.integers(1, 9))
                app.state["screen_fraction"] = m["fraction"]
            elif ev < 0.25:
                m["accumulate"] ^= 1
                app.state["accumulate"] = m["accumulate"]
            elif ev < 0.35:
                m["inner"] = [int(rng.integers(1, 3000)), int(rng.integers(1, 2000))]
                app.state["inner_size"][0] = m["inner"]
            elif ev < 0.45:
                edited = app.uniform.copy()
                changed = rng.random() < 0.6
                if changed:
                    edited["samples"] = int(edited["samples"][0]) + 1
                edited["frame"] = 12345                                               # not a hashed field
                assert app.set_params(edited, t) is changed
                if changed:
                    m["last_inv"], m["acc"] = t, 0
            else:
                elapsed = min(max(f32(t - m["last_inv"]), f32(0.0)), f32(1.0))
                dynamic = f32(1.0) + f32(3.0) * (f32(1.0) - elapsed)                  # :75-76
                m["acc"] += 1                                                         # :78
                if not m["accumulate"]:
                    m["acc"] = 0                                                      # :79-81
                div = max(m["fraction"], int(dynamic))                                # :83-85
                window = [m["inner"][0] // div, m["inner"][1] // div]
                resized = window != m["last_window"]
                if resized:
                    m["acc"], m["last_window"] = 0, window                            # :87-93
                m["frame"] += 1                                                       # :96
                assert app.begin_frame(t) is resized, (trial, step)
                u = app.uniform[0]
                assert [int(x) for x in u["window"]] == window, (trial, step)
                assert int(u["frame"]) == m["frame"] and int(u["accumulation_frame"]) == m["acc"], (trial, step, m)
                assert float(u["aspect"]) == float(f32(m["inner"][0]) / f32(m["inner"][1]))
