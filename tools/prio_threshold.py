#!/usr/bin/env python3
"""Where does raising the wave priority in the node / leaf phases (knob 112) stop paying?  Lattice scenes of growing size at
1080p, the format the library picks by size, ms per step with the priority off (112 = 1) and on (112 = 2).
usage: python tools/prio_threshold.py [copies ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
rt = graft.load_package()
synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
cfg = graft.load_config("C4")
W, H, steps = 1920, 1080, 4
for copies in [int(a) for a in sys.argv[1:]] or [3, 10, 30, 60, 120, 250, 500]:
    c = dict(cfg, scene=f"synth:lattice:{copies}:{copies * 90}:1")
    syn = synth.load_config_scene(c, ROOT)
    res = {}
    for knob in (1, 2, 1, 2):
        with rt.Context(0) as ctx:
            ctx.set_option(112, knob)
            ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
            if len(syn.spheres):
                ctx.spheres_upload(syn.spheres)
            ctx.accum_resize(W, H)
            u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
            u["frame"] = 0
            u["accumulation_frame"] = 1
            ctx.dispatch_frames(u, steps)
            ctx.sync()
            st = ctx.stats()
            res.setdefault(knob, []).append(ctx.last_dispatch_ms() / steps)
    n_prims = len(syn.indices) // 3 + len(syn.spheres)
    per_node = (st.bvh_bytes - n_prims * 176) / max(st.bvh_nodes, 1)
    print(f"{copies} copies: {n_prims} primitives, {st.bvh_bytes / 2**20:.0f} MiB of records, {per_node:.0f}-byte nodes: "
          f"off {min(res[1]):.2f} ms, on {min(res[2]):.2f} ms per step ({(min(res[2]) / min(res[1]) - 1) * 100:+.1f} %)", flush=True)
