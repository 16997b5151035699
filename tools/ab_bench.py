#!/usr/bin/env python3
"""A/B timing of kernel variants in ONE process, interleaved rounds (cdna guide §5.4 rule 24).
usage: python tools/ab_bench.py [--config C3] [--scale 1.0] [--rounds 5] [--variants 0,1] [--count]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C3")
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--variants", default="0,1")
ap.add_argument("--count", action="store_true")
ap.add_argument("--bounces", type=int, default=-1)
ap.add_argument("--bvh", default="", help="max_leaf,bins[,traversal_cost*16] builder options")
ap.add_argument("--bvh-sweep", default="", help="semicolon-separated builder option sets: one context each, interleaved")
ap.add_argument("--stride", default="", help="comma-separated node strides (3,4): one context each, variant 2, interleaved")
ap.add_argument("--tune", default="", help="semicolon-separated weight sets for variant 2, e.g. 4,4,4,4;4,4,2,4")
args = ap.parse_args()
rt = graft.load_package()
cfg = graft.load_config(args.config)
W, H = int(cfg["width"] * args.scale), int(cfg["height"] * args.scale)
syn = None
if cfg["scene"].startswith("synth:"):
    from ray_tracing_amd import synth
    syn = synth.load_config_scene(cfg, ROOT)
def make_ctx(stride=0, bvh=None):
    ctx = rt.Context(0)
    bvh = bvh or args.bvh
    if bvh:
        o = [int(x) for x in bvh.split(",")]
        ctx.set_option(106, o[0]); ctx.set_option(107, o[1])
        if len(o) > 2: ctx.set_option(110, o[2])
    ctx.set_option(rt.RT_OPT_NODE_STRIDE, stride)
    if syn is not None:
        ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        if len(syn.spheres): ctx.spheres_upload(syn.spheres)
    else:
        ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
    ctx.accum_resize(W, H)
    return ctx
ctx = make_ctx()
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"] if args.bounces < 0 else args.bounces, samples=cfg["samples"],
                    aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]))
variants = [int(v) for v in args.variants.split(",")]
times = {v: [] for v in variants}
ref = None
for r in range(args.rounds + 1):
    for v in variants:
        ctx.set_option(rt.RT_OPT_KERNEL, v)
        ctx.stats_reset()
        ctx.dispatch(u)
        ms = ctx.last_dispatch_ms()
        st = ctx.stats()
        if r > 0:
            times[v].append(ms)
        if r == 0:
            img = ctx.accum_download()
            if ref is None: ref = img
            print(f"variant {v}: rays {st.rays} image==variant{variants[0]}: {np.array_equal(img.view(np.uint32), ref.view(np.uint32))}")
for v in variants:
    t = np.array(times[v])
    print(f"variant {v}: median {np.median(t):.3f} ms  min {t.min():.3f}  -> {st.rays / np.median(t) / 1e3:.0f} Mrays/s")
if args.stride:
    strides = [int(x) for x in args.stride.split(",")]
    ctxs = {sd: make_ctx(sd) for sd in strides}
    res = {sd: [] for sd in strides}
    for r in range(args.rounds + 1):
        for sd in strides:
            ctxs[sd].dispatch(u)
            ms = ctxs[sd].last_dispatch_ms()
            if r > 0: res[sd].append(ms)
    for sd in strides:
        print(f"node stride {sd * 16} B: median {np.median(res[sd]):.3f} ms min {min(res[sd]):.3f}")
if args.bvh_sweep:
    sets = args.bvh_sweep.split(";")
    ctxs = {b: make_ctx(0, b) for b in sets}
    res = {b: [] for b in sets}
    for r in range(args.rounds + 1):
        for b in sets:
            ctxs[b].dispatch(u)
            ms = ctxs[b].last_dispatch_ms()
            if r > 0: res[b].append(ms)
    for b in sets:
        c = ctxs[b]
        c.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1); c.stats_reset(); c.dispatch(u); st = c.stats()
        print(f"bvh {b}: median {np.median(res[b]):.3f} ms min {min(res[b]):.3f} | nodes/ray {st.node_visits / st.rays:.2f} tris/ray {st.triangle_tests / st.rays:.2f} bvh_nodes {st.bvh_nodes}")
if args.tune:
    ctx.set_option(rt.RT_OPT_KERNEL, 2)
    sets = [[int(x) for x in t.split(",")] for t in args.tune.split(";")]
    res = {i: [] for i in range(len(sets))}
    for r in range(args.rounds + 1):
        for i, ws in enumerate(sets):
            for k, w in enumerate(ws):
                ctx.set_option(100 + k, w)
            ctx.dispatch(u)
            ms = ctx.last_dispatch_ms()
            if r > 0: res[i].append(ms)
    for i, ws in enumerate(sets):
        print(f"tune {ws}: median {np.median(res[i]):.3f} ms min {min(res[i]):.3f}")
    for k, w in enumerate([3, 4, 2, 0, 64, 0]): ctx.set_option(100 + k, w)
if args.count:
    for v in variants:
        ctx.set_option(rt.RT_OPT_KERNEL, v)
        ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1)
        ctx.stats_reset()
        ctx.dispatch(u)
        st = ctx.stats()
        ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 0)
        print(f"variant {v}: nodes/ray {st.node_visits / st.rays:.2f} tris/ray {st.triangle_tests / st.rays:.2f} | lane utilization: "
              f"node loop {st.node_visits / (64 * st.wave_node_iters):.3f}  leaf loop {st.triangle_tests / (64 * st.wave_leaf_iters):.3f}  "
              f"steps {st.rays / (64 * st.wave_steps):.3f} | per wave-step: node iters {st.wave_node_iters / st.wave_steps:.1f} leaf iters {st.wave_leaf_iters / st.wave_steps:.1f}")
