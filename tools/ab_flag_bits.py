#!/usr/bin/env python3
"""Same frames from two builds of the library — the shipping one (phased kernel built with -structurizecfg-skip-uniform-regions)
and `make noflag` (no such flag anywhere) — at FULL size, per config and node format: the accumulation buffers must hold the
same bits and the ray counts must agree.  Product code on both sides (no oracle): this is the at-scale complement of the
randomised checker, which compares small frames with brute force.
usage: python tools/ab_flag_bits.py [--quick]  (parent: runs itself once per library, compares)
       python tools/ab_flag_bits.py --child    (prints one line per case: name, sha256 of the buffer, rays, ms per step)"""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
QUICK = [("C2", 0, 2), ("C3", 0, 2), ("C3", 4, 2), ("C4", 0, 2), ("C4", 1, 2)]
CASES = QUICK if "--quick" in sys.argv else [("C2", 0, 4), ("C3", 0, 4), ("C3", 2, 4), ("C3", 3, 4), ("C3", 4, 4), ("C4", 0, 2), ("C4", 1, 2), ("C4", 2, 2), ("C4", 3, 2)]

if "--child" in sys.argv:
    import __graft_entry__ as graft
    rt = graft.load_package()
    for name, fmt, steps in CASES:
        cfg = graft.load_config(name)
        W, H = cfg["width"], cfg["height"]
        with rt.Context(0) as ctx:
            ctx.set_option(rt.RT_OPT_NODE_FORMAT, fmt)
            if cfg["scene"].startswith("synth:"):
                synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
                syn = synth.load_config_scene(cfg, ROOT)
                ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
                if len(syn.spheres):
                    ctx.spheres_upload(syn.spheres)
                del syn
            else:
                ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
            ctx.accum_resize(W, H)
            u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
            u["frame"] = 0
            u["accumulation_frame"] = 1
            ctx.dispatch_frames(u, steps)
            img = ctx.accum_download()
            print(json.dumps({"case": f"{name} format {fmt}", "sha256": hashlib.sha256(img.tobytes()).hexdigest(), "rays": int(ctx.stats().rays),
                              "ms_per_step": ctx.last_dispatch_ms() / steps}), flush=True)
    sys.exit(0)

libs = {"flag": os.path.join(ROOT, "ray-tracing_amd", "librt_amd.so"), "noflag": os.path.join(ROOT, "ray-tracing_amd", "librt_amd_noflag.so")}
res = {}
for tag, path in libs.items():
    assert os.path.exists(path), f"{path}: build it with `make -C ray-tracing_amd noflag`"
    out = subprocess.run([sys.executable, "-u", os.path.abspath(__file__), "--child"] + (["--quick"] if "--quick" in sys.argv else []), env=dict(os.environ, RT_AMD_LIB=path),
                         capture_output=True, text=True, check=True).stdout
    res[tag] = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
bad = 0
for a, b in zip(res["flag"], res["noflag"]):
    same = a["sha256"] == b["sha256"] and a["rays"] == b["rays"]
    bad += 0 if same else 1
    print(f"{a['case']}: {'identical' if same else 'DIFFERENT'}; rays {a['rays']}; ms per step {a['ms_per_step']:.2f} with the flag, {b['ms_per_step']:.2f} without")
sys.exit(1 if bad else 0)
