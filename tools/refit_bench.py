#!/usr/bin/env python3
"""rt_models_update on a big scene: host refit (RT_OPT_REFIT=1) against device refit (=2).
usage: python tools/refit_bench.py [--config C4] [--rounds 3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C4")
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
rt = graft.load_package()
cfg = graft.load_config(args.config)
from ray_tracing_amd import synth
syn = synth.load_config_scene(cfg, ROOT) if cfg["scene"].startswith("synth:") else None
hs = None if syn else rt.HostScene(os.path.join(ROOT, cfg["scene"]))
prims = syn.primitives if syn else hs.primitives
models = syn.models if syn else hs.models
ident = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (len(prims), 1))
W, H = 640, 360
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=4, samples=2)
images = {}
for mode in (1, 2):
    ctx = rt.Context(0)
    ctx.set_option(rt.RT_OPT_REFIT, mode)
    if syn:
        ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        if len(syn.spheres): ctx.spheres_upload(syn.spheres)
    else:
        ctx.scene_finish(hs)
    ts = []
    for r in range(args.rounds):
        m = ident.copy()
        m[:, 3] += 0.05 * (r + 1)
        t0 = time.perf_counter()
        ctx.models_update(models, transforms=m)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    ctx.accum_resize(W, H)
    ctx.dispatch(u)
    images[mode] = ctx.accum_download()
    st = ctx.stats()
    print(f"RT_OPT_REFIT={mode}: rt_models_update {min(ts) * 1e3:.1f} ms (best of {args.rounds}; {st.bvh_nodes} nodes, refits {st.bvh_refits})", flush=True)
    ctx.close()
print("images identical:", bool(np.array_equal(images[1].view(np.uint32), images[2].view(np.uint32))))
