#!/usr/bin/env python3
"""How much of a GPU's time a 1/N image shard costs on ONE GPU (tail + launch overhead of small dispatches).
usage: python tools/shard_scaling.py [--config C3] [--worlds 1,2,4,8] [--steps 16]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--pipeline", type=int, default=1)
ap.add_argument("--opt", default="", help="key=value,... passed to rt_set_option")
args = ap.parse_args()
rt = graft.load_package()
cfg = graft.load_config(args.config)
W, H = cfg["width"], cfg["height"]
scene = rt.HostScene(os.path.join(ROOT, cfg["scene"]))
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
base = None
for world in [int(x) for x in args.worlds.split(",")]:
    worst = 0.0
    for rank in sorted({0, world - 1}):
        ctx = rt.Context(0)          # rt_set_shard must precede the first rt_accum_resize
        ctx.scene_finish(scene)
        if args.chunk: ctx.set_option(108, args.chunk)
        ctx.set_option(rt.RT_OPT_FRAME_PIPELINE, args.pipeline)
        for kv in filter(None, args.opt.split(",")):
            ctx.set_option(int(kv.split("=")[0]), int(kv.split("=")[1]))
        ctx.set_shard(rank, world)
        ctx.accum_resize(W, H)
        ctx.frames_reserve(args.steps)
        ctx.dispatch_frames(u, 1); ctx.sync()
        t0 = time.perf_counter()
        ctx.dispatch_frames(u, args.steps); ctx.sync()
        dt = (time.perf_counter() - t0) * 1e3 / args.steps
        worst = max(worst, dt)
    if base is None: base = worst
    print(f"world {world}: {worst:.3f} ms/step on the slowest of ranks 0 and {world - 1}; ideal {base / world:.3f}; efficiency {base / world / worst:.3f}", flush=True)
