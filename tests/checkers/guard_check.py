#!/usr/bin/env python3
"""Runs parity cases through the GUARDED build of the library (make guard -> librt_amd_guard.so: every traversal
reference range-checked, stack pointers bounded, steps per ray capped; a failed check counts in RtStats.guard_trips
instead of faulting the GPU) and prints one JSON line per case.  Started by tests/test_gpu_guard.py in its own
process, because the library is chosen at load time (RT_AMD_LIB)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "ray-tracing_amd", "librt_amd_guard.so")
import numpy as np  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

rt = graft.load_package()
assert rt.LIB_PATH.endswith("librt_amd_guard.so")
from oracle import obj_loader, oracle  # noqa: E402
from ray_tracing_amd import synth  # noqa: E402


def same(a, b):
    return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))


def run(name, scene, spheres, cam, w, h, samples, bounces, options=()):
    osc = oracle.OracleScene(scene, spheres=spheres)
    want, _, cnt = osc.render(oracle.make_uniform(w, h, cam["position"], cam["yaw"], cam["pitch"], max_bounces=bounces, samples=samples), mode=1)
    with rt.Context(0) as ctx:
        for k, v in options:
            ctx.set_option(k, v)
        ctx.scene_upload(scene.vertices, scene.indices, scene.primitives, scene.models)
        if spheres is not None:
            ctx.spheres_upload(spheres)
        ctx.accum_resize(w, h)
        ctx.dispatch(rt.make_uniform(w, h, cam["position"], cam["yaw"], cam["pitch"], max_bounces=bounces, samples=samples))
        st = ctx.stats()
        got = ctx.accum_download()
    print(json.dumps({"case": name, "guard_trips": int(st.guard_trips), "rays": int(st.rays), "rays_equal": int(st.rays) == int(cnt.rays),
                      "identical": same(got, want)}), flush=True)


c1 = graft.load_config("C1")
cornell = obj_loader.load(os.path.join(ROOT, c1["scene"]))
# the window of the round-1 fault report (61x37), on every node format the kernel has
for fmt in (1, 2, 3, 4):
    run(f"c1_61x37_format{fmt}", cornell, None, c1, 61, 37, 2, c1["max_bounces"], [(rt.RT_OPT_NODE_FORMAT, fmt)])
lat = synth.lattice_scene(os.path.join(ROOT, "scenes", "teapot.obj"), 26, 300, seed=1)
lscene = obj_loader.Scene(lat.vertices, lat.indices, lat.primitives, lat.models)
for fmt in (1, 2, 3, 4):
    run(f"c4_lattice_26_format{fmt}", lscene, lat.spheres, lat.camera, 96, 54, 4, 8, [(rt.RT_OPT_NODE_FORMAT, fmt)])
