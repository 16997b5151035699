#!/usr/bin/env python3
"""Per-ray traversal counts of the device BVH and per-wave phase iterations (RT_OPT_COUNT_TRAVERSAL build of the kernel).
usage: python tools/phase_counts.py [config] [WxH] [key=value,...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
rt = graft.load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = graft.load_config(name)
W, H = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1920x1080").split("x"))
ctx = rt.Context(0)
for kv in filter(None, (sys.argv[3] if len(sys.argv) > 3 else "").split(",")):
    k, v = kv.split("=")
    ctx.set_option(int(k), int(v))
if cfg["scene"].startswith("synth:"):
    synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
    syn = synth.load_config_scene(cfg, ROOT)
    ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
    if len(syn.spheres):
        ctx.spheres_upload(syn.spheres)
else:
    ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
ctx.set_option(rt.RT_OPT_COUNT_TRAVERSAL, 1)
ctx.accum_resize(W, H)
u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
ctx.dispatch(u)
s = ctx.stats()
ms = ctx.last_dispatch_ms()
r = s.rays
print(json.dumps({"config": name, "window": [W, H], "rays": r, "ms": ms, "bvh_nodes": s.bvh_nodes, "stack_entries": s.stack_entries,
                  "node_visits_per_ray": s.node_visits / r, "tri_tests_per_ray": s.triangle_tests / r,
                  "wave_node_iters_per_ray_x64": s.wave_node_iters * 64 / r, "wave_leaf_iters_per_ray_x64": s.wave_leaf_iters * 64 / r,
                  "wave_shade_iters_per_ray_x64": s.wave_steps * 64 / r,
                  "node_lane_util": s.node_visits / (s.wave_node_iters * 64), "leaf_lane_util": s.triangle_tests / (s.wave_leaf_iters * 64),
                  "shade_lane_util": r / (s.wave_steps * 64)}))
