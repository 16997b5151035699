#!/usr/bin/env python3
"""Where should the scheduler weights switch from 3/4/2 to 2/4/6?  Lattice scenes of growing size at 1080p,
automatic node format, ms per step with both sets.  usage: python tools/weights_threshold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
rt = graft.load_package()
synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
cfg = graft.load_config("C4")
W, H, steps = 1920, 1080, 4
for copies in [30, 45, 60, 90, 150]:
    syn = synth.load_config_scene(dict(cfg, scene=f"synth:lattice:{copies}:{copies * 90}:1"), ROOT)
    res = {}
    for wts in ((3, 4, 2), (2, 4, 6)) * 2:
        with rt.Context(0) as ctx:
            for k, v in zip((100, 101, 102), wts):
                ctx.set_option(k, v)
            ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
            if len(syn.spheres):
                ctx.spheres_upload(syn.spheres)
            ctx.accum_resize(W, H)
            u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
            u["frame"] = 0
            u["accumulation_frame"] = 1
            ctx.dispatch_frames(u, steps)
            ctx.sync()
            res.setdefault(wts, []).append(ctx.last_dispatch_ms() / steps)
    n_prims = len(syn.indices) // 3 + len(syn.spheres)
    a, b = min(res[(3, 4, 2)]), min(res[(2, 4, 6)])
    print(f"{copies} copies: {n_prims} primitives ({n_prims * 24 / 2**20:.1f} MiB rule): 3/4/2 {a:.2f} ms, 2/4/6 {b:.2f} ms ({(b / a - 1) * 100:+.1f} %)", flush=True)
