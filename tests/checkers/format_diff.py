#!/usr/bin/env python3
"""Render one frame of a config with two node formats, list the pixels whose bits differ and ask the oracle's BRUTE-FORCE
mode (every primitive tested, no BVH: the ground truth of SURVEY section 8 A4) which one is right.
usage: python tests/checkers/format_diff.py [config] [WxH] [fmtA] [fmtB] [max pixels to check] [frames: each frame number 1..N is rendered on its own]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as graft
rt = graft.load_package()
from oracle import obj_loader, oracle
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "960x540").split("x"))
fa, fb = int(sys.argv[3]) if len(sys.argv) > 3 else 3, int(sys.argv[4]) if len(sys.argv) > 4 else 2
max_check = int(sys.argv[5]) if len(sys.argv) > 5 else 6
n_frames = int(sys.argv[6]) if len(sys.argv) > 6 else 1
cfg = graft.load_config(name)
synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
s = synth.load_config_scene(cfg, ROOT) if cfg["scene"].startswith("synth:") else None
hs = None if s else rt.HostScene(os.path.join(ROOT, cfg["scene"]))
asp = np.float32(cfg["width"]) / np.float32(cfg["height"])
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=asp, max_bounces=cfg["max_bounces"], samples=cfg["samples"])
imgs, rays = {}, {}
ctxs = {}
for f in (fa, fb):
    with rt.Context(0) as ctx:
        ctx.set_option(rt.RT_OPT_NODE_FORMAT, f)
        if s:
            ctx.scene_upload(s.vertices, s.indices, s.primitives, s.models)
            if len(s.spheres):
                ctx.spheres_upload(s.spheres)
        else:
            ctx.scene_finish(hs)
        imgs[f], rays[f] = [], []
        for fr in range(1, n_frames + 1):
            u["frame"] = fr
            ctx.accum_resize(W, H)
            ctx.stats_reset()
            ctx.dispatch(u)
            imgs[f].append(ctx.accum_download())
            rays[f].append(int(ctx.stats().rays))
osc = None
for fr in range(1, n_frames + 1):
  ia, ib = imgs[fa][fr - 1], imgs[fb][fr - 1]
  diff = np.argwhere((ia.view(np.uint32) != ib.view(np.uint32)).any(axis=-1))
  print(json.dumps({"config": name, "window": [W, H], "frame": fr, "rays": [rays[fa][fr - 1], rays[fb][fr - 1]], "differing_pixels": len(diff)}), flush=True)
  if len(diff) and osc is None:
    if s:
        osc = oracle.OracleScene(obj_loader.Scene(s.vertices, s.indices, s.primitives, s.models), spheres=s.spheres if len(s.spheres) else None, build_bvh=True)
    else:
        osc = oracle.OracleScene(obj_loader.load(os.path.join(ROOT, cfg["scene"])))
  if len(diff):
    ou = oracle.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=asp, max_bounces=cfg["max_bounces"], samples=cfg["samples"], frame=fr)
    for (y, x) in diff[:max_check]:
        res = {}
        for mode in (0, 1):
            t0 = time.time()
            acc = np.zeros((H, W, 4), np.float32)
            osc.render(ou, acc, mode=mode, rows=(int(y), int(y) + 1), cols=(int(x), int(x) + 1))
            res[mode] = acc[y, x].copy()
            dt = time.time() - t0
        same = lambda a, b: bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
        print(json.dumps({"pixel": [int(x), int(y)], f"format{fa}_eq_brute": same(ia[y, x], res[0]), f"format{fb}_eq_brute": same(ib[y, x], res[0]),
                          "oracle_bvh_eq_brute": same(res[1], res[0]), "brute_seconds": round(dt, 1)}), flush=True)
