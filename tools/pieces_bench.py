#!/usr/bin/env python3
"""What a frame costs when it is dispatched in pieces (rt_dispatch_bands; the reference's TODO.md:9): one frame as 1, 2, 4, ... pieces of
8-row bands, each piece one launch, the host waiting for every piece (what an interactive host does between two looks at its event queue).
Prints the frame time and the longest piece.   usage: python tools/pieces_bench.py [--config C3] [--pieces 1,2,4,8,16,32] [--frames 4]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--pieces", default="1,2,4,8,16,32")
ap.add_argument("--frames", type=int, default=4)
args = ap.parse_args()
rt = graft.load_package()
cfg = graft.load_config(args.config)
W, H = cfg["width"], cfg["height"]
ctx = rt.Context(0)
if cfg["scene"].startswith("synth:"):
    synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
    syn = synth.load_config_scene(cfg, ROOT)
    ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
    if len(syn.spheres):
        ctx.spheres_upload(syn.spheres)
else:
    ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
ctx.accum_resize(W, H)
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
bands = (H + 7) // 8
ctx.dispatch(u); ctx.sync()
for n in [int(x) for x in args.pieces.split(",")]:
    n = min(n, bands)
    cut = [bands * k // n for k in range(n + 1)]
    worst = 0.0
    t0 = time.perf_counter()
    for f in range(args.frames):
        u["frame"] += 1
        for k in range(n):
            t1 = time.perf_counter()
            ctx.dispatch_bands(u, cut[k], cut[k + 1]); ctx.sync()
            worst = max(worst, time.perf_counter() - t1)
    dt = (time.perf_counter() - t0) / args.frames
    print(f"{args.config}: {n:3d} pieces of {bands / n:6.1f} bands: {dt * 1e3:8.2f} ms per frame, longest piece {worst * 1e3:7.2f} ms", flush=True)
