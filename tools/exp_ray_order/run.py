#!/usr/bin/env python3
"""
Ray-order experiment: is a wavefront path with sorted rays worth building for scenes whose
records leave the caches?

Dumps one generation of secondary rays of a real render (every pixel-frame's ray at a given sample and bounce), then traces
the list in several orders with (a) a plain one-ray-per-lane kernel and (b) the production phased scheduler run over the list.
Hits must hold the same bits in every order.  Needs tools/exp_ray_order/librt_exp.so (make -C tools/exp_ray_order).

  python tools/exp_ray_order/run.py --config C4 --frames 4 --sample 1 --bounce 2 --out gpurun_out/exp_ray_order.json
  --pmc: a short sequence (phased kernel, arrival then sorted, one launch each) for a rocprofv3 --pmc pass
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_AMD_LIB"] = os.path.join(ROOT, "tools", "exp_ray_order", "librt_exp.so")
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402


def part1by2(v):
    """spread the low 10 bits of v (uint32 array) to every third bit"""
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton(cells, bits):
    c = (cells >> (10 - bits)).astype(np.uint32) if bits < 10 else cells.astype(np.uint32)
    return part1by2(c[:, 0]) | (part1by2(c[:, 1]) << 1) | (part1by2(c[:, 2]) << 2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C4")
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--sample", type=int, default=1)
    ap.add_argument("--bounce", type=int, default=2)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--size", default=None, help="WxH instead of the config's")
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--per-ray", action="store_true", help="only: the distribution of node visits per ray of the generation")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    rt = graft.load_package()
    L = rt.lib()
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.rt_exp_dump_rays.argtypes = [vp, vp, u32, u32, u32, vp, u64, C.POINTER(u64)]
    L.rt_exp_set_rays.argtypes = [vp, vp, u64]
    L.rt_exp_trace.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]

    cfg = graft.load_config(args.config)
    W, H = (int(x) for x in args.size.split("x")) if args.size else (cfg["width"], cfg["height"])
    ctx = rt.Context(0)
    ctx.set_option(rt.RT_OPT_NODE_FORMAT, 4)
    t0 = time.time()
    if cfg["scene"].startswith("synth:"):
        synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
        syn = synth.load_config_scene(cfg, ROOT)
        ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        if len(syn.spheres):
            ctx.spheres_upload(syn.spheres)
        del syn
    else:
        ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
    print(f"[exp] scene up in {time.time() - t0:.1f} s", flush=True)
    ctx.accum_resize(W, H)
    # the production path on the same scene in the same process, for scale: K frames in one launch
    pu = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]),
                         max_bounces=cfg["max_bounces"], samples=cfg["samples"])
    pu["frame"], pu["accumulation_frame"] = 0, 1
    ctx.dispatch_frames(pu, args.frames)
    ctx.stats_reset()
    ctx.accum_resize(W, H)
    pu["frame"], pu["accumulation_frame"] = 0, 1
    ctx.dispatch_frames(pu, args.frames)
    prod_ms, prod_rays = ctx.last_dispatch_ms(), int(ctx.stats().rays)
    print(f"[exp] production render, {args.frames} frames in one launch: {prod_ms:.2f} ms, {prod_rays} rays, {prod_ms * 1e6 / prod_rays:.4f} ns/ray (shading included)", flush=True)
    ctx.accum_resize(W, H)
    u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]),
                        max_bounces=cfg["max_bounces"], samples=cfg["samples"])
    slots = ((W + 7) // 8) * ((H + 7) // 8) * 64 * args.frames
    dump = np.zeros((slots, 8), np.float32)
    n_slots = u64()
    t0 = time.time()
    rt._check(L.rt_exp_dump_rays(ctx.handle, u.ctypes.data, args.frames, args.sample, args.bounce, dump.ctypes.data, slots, C.byref(n_slots)))
    valid = (dump[:, 4:7] != 0).any(axis=1)
    rays = np.ascontiguousarray(dump[valid])          # arrival order = pixel-frame order
    del dump
    n = len(rays)
    print(f"[exp] dumped generation (sample {args.sample}, bounce {args.bounce}) of {args.frames} frames at {W}x{H}: {n} rays of {slots} slots "
          f"in {time.time() - t0:.1f} s", flush=True)
    rt._check(L.rt_exp_set_rays(ctx.handle, rays.ctypes.data, n))
    nonfinite = ~np.isfinite(rays[:, [0, 1, 2, 4, 5, 6]]).all(axis=1)
    print(f"[exp] rays with a non-finite origin or direction component: {int(nonfinite.sum())}", flush=True)
    if args.per_ray:
        # how long is the longest ray?  (a launch of one generation is as long as its slowest ray: a dependent chain of node visits)
        work = np.zeros((n, 2), np.uint32)
        ms1 = np.zeros(1, np.float32)
        rt._check(L.rt_exp_trace(ctx.handle, u.ctypes.data, None, 0, 1, 1, ms1.ctypes.data, work.ctypes.data, None))
        v = work[:, 0].astype(np.int64)
        qs = [50, 90, 99, 99.9, 99.99, 99.999, 100]
        print("[exp] node visits per ray: mean %.1f; " % v.mean() + ", ".join(f"p{q}: {int(np.percentile(v, q))}" for q in qs), flush=True)
        top = np.argsort(v)[-8:][::-1]
        for i in top:
            print(f"[exp]   ray {i}: {v[i]} node visits, {work[i, 1]} primitive tests, origin {rays[i, 0:3]}, direction {rays[i, 4:7]}, non-finite {bool(nonfinite[i])}", flush=True)
        print(f"[exp] visits of the non-finite rays: {v[nonfinite][:16]}", flush=True)
        print(f"[exp] rays above 1000 visits: {int((v > 1000).sum())}, above 10000: {int((v > 10000).sum())}; share of all visits in rays above 1000: {v[v > 1000].sum() / v.sum():.4f}", flush=True)
        ctx.close()
        return

    o, d = rays[:, 0:3], rays[:, 4:7]
    lo, hi = o.min(axis=0), o.max(axis=0)
    cells = np.minimum(((o - lo) / np.maximum(hi - lo, 1e-20) * 1024.0).astype(np.int64), 1023).astype(np.uint32)
    octant = ((d[:, 0] < 0).astype(np.uint32) | ((d[:, 1] < 0).astype(np.uint32) << 1) | ((d[:, 2] < 0).astype(np.uint32) << 2))
    rng = np.random.default_rng(1)
    orders = {
        "arrival": None,
        "octant_then_morton30": np.argsort((octant.astype(np.uint64) << 30) | morton(cells, 10), kind="stable").astype(np.uint32),
        "morton30_then_octant": np.argsort((morton(cells, 10).astype(np.uint64) << 3) | octant, kind="stable").astype(np.uint32),
        "morton15_then_octant (18-bit key: one counting-sort pass)": np.argsort((morton(cells, 5).astype(np.uint64) << 3) | octant, kind="stable").astype(np.uint32),
        "morton30 (origin only)": np.argsort(morton(cells, 10), kind="stable").astype(np.uint32),
        "random": rng.permutation(n).astype(np.uint32),
    }
    if args.pmc:
        orders = {k: orders[k] for k in ("arrival", "morton30_then_octant")}

    results = {"config": args.config, "window": [W, H], "frames": args.frames, "sample": args.sample, "bounce": args.bounce, "rays": n,
               "production": {"ms": prod_ms, "rays": prod_rays, "ns_per_ray": prod_ms * 1e6 / prod_rays}, "runs": []}
    ref_hits = None
    for kernel, kname in ((1, "phased"),) if args.pmc else ((0, "plain"), (1, "phased")):
        for oname, order in orders.items():
            ms = np.zeros(args.reps if not args.pmc else 1, np.float32)
            hits = np.zeros((n, 2), np.uint32)
            rt._check(L.rt_exp_trace(ctx.handle, u.ctypes.data, None if order is None else order.ctypes.data, kernel, 0, len(ms),
                                     ms.ctypes.data, hits.ctypes.data, None))
            if ref_hits is None:
                ref_hits = hits
            same = bool(np.array_equal(hits, ref_hits))
            run = {"kernel": kname, "order": oname, "ms": [float(x) for x in ms], "best_ms": float(ms.min()), "ns_per_ray": float(ms.min()) * 1e6 / n,
                   "grays_per_s": n / (float(ms.min()) * 1e-3) / 1e9, "hits_identical": same, "hit_fraction": float((hits[:, 1] != 0xFFFFFFFF).mean())}
            if not args.pmc:
                cnt = np.zeros(8, np.uint64)
                ms1 = np.zeros(1, np.float32)
                rt._check(L.rt_exp_trace(ctx.handle, u.ctypes.data, None if order is None else order.ctypes.data, kernel, 1, 1, ms1.ctypes.data, None, cnt.ctypes.data))
                # DevCounters: rays, pixel_frames, node_visits, triangle_tests, wave_node_iters, wave_leaf_iters, wave_steps, guard
                run.update({"node_visits_per_ray": float(cnt[2]) / n, "prim_tests_per_ray": float(cnt[3]) / n,
                            "node_lane_utilization": float(cnt[2]) / max(1.0, float(cnt[4]) * 64.0),
                            "leaf_lane_utilization": float(cnt[3]) / max(1.0, float(cnt[5]) * 64.0)})
            results["runs"].append(run)
            print("[exp] " + json.dumps(run), flush=True)
            assert same, "hits differ between orders"
    base = {r["kernel"]: r["best_ms"] for r in results["runs"] if r["order"] == "arrival"}
    for r in results["runs"]:
        r["speedup_vs_arrival"] = base[r["kernel"]] / r["best_ms"]
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in results.items() if k != "runs"}))
    for r in results["runs"]:
        print(f"{r['kernel']:7s} {r['order'][:40]:40s} {r['best_ms']:8.3f} ms  {r['ns_per_ray']:.3f} ns/ray  x{r['speedup_vs_arrival']:.2f}"
              + (f"  node util {r['node_lane_utilization']:.2f}  visits {r['node_visits_per_ray']:.1f}" if "node_lane_utilization" in r else ""))
    ctx.close()


if __name__ == "__main__":
    main()
