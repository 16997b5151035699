#!/usr/bin/env python3
"""
bench.py — headline benchmark: Mrays/s on teapot-circle 3840x2160, 8 bounces
(BASELINE.json config C3), on N GPUs of one node.

A "step" is one dispatch of the compute pass (one frame of `samples`=16 spp over
the whole 4K image, src/app.rs:95-101).  With --gpus N the image's 8-row bands
are dealt round-robin to N ranks (one process per GPU), the scene is replicated,
and ONE framebuffer gather to rank 0 (RCCL) closes the timed region.  Total work
is fixed as N grows → "scaling": "strong".

Rays are counted exactly on the device (one count per trace_ray call,
shaders/main.wgsl:44).  The roofline entry prices the render kernel against HBM
with SURVEY.md §8(d)'s algorithmic bytes per ray; the cpu_baseline entry times
the CPU oracle (a port, not lavapipe) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
S_NODE, S_TRI, S_HIT, S_FB = 32, 36, 284, 32   # SURVEY.md §8(d) bytes


def source_stamp():
    """sha256 over the device sources a profile was taken with (tools/pmc_summary.py records the same digest in profiles/*.json)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rt_kernels.hip", "rt_device.h", "rt_pack.h", "rt_math.h"):
        h.update(open(os.path.join(ROOT, "ray-tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def lavapipe_probe():
    """the intended CPU baseline is the reference's WGSL on lavapipe through wgpu.  Probe, at run time, for what that
    needs; the timed baseline falls back to the CPU restatement when any of it is missing (it always is in this image)."""
    import ctypes.util
    import glob
    import shutil
    icds = glob.glob("/usr/share/vulkan/icd.d/*lvp*") + glob.glob("/etc/vulkan/icd.d/*lvp*")
    probe = {"cargo": shutil.which("cargo") is not None, "rustc": shutil.which("rustc") is not None,
             "vulkan_loader": ctypes.util.find_library("vulkan") is not None, "vulkaninfo": shutil.which("vulkaninfo") is not None,
             "lavapipe_icd": bool(icds), "reference_sources": os.path.isdir("/root/reference")}
    probe["usable"] = all(probe[k] for k in ("cargo", "rustc", "vulkan_loader", "lavapipe_icd", "reference_sources"))
    return probe


def cpu_baseline_and_byte_model(cfg, sample_w, sample_h, timed_w, timed_h, audit_pixels):
    """Oracle (BVH mode, OpenMP) on one sample_w x sample_h frame of the same scene/camera/spp: the frame the HIP path is checked
    against bit for bit, and the per-ray node/triangle means of the byte model.  The reported CPU rate is the BEST OF 3 runs of a
    timed_w x timed_h frame.  audit_pixels random pixels of the big frame are rendered once more by BRUTE FORCE."""
    from oracle import obj_loader, oracle
    if cfg["scene"].startswith("synth:"):
        synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
        s = synth.load_config_scene(cfg, ROOT)
        sc = oracle.OracleScene(obj_loader.Scene(s.vertices, s.indices, s.primitives, s.models),
                                spheres=s.spheres if len(s.spheres) else None)
    else:
        sc = oracle.OracleScene(obj_loader.load(os.path.join(ROOT, cfg["scene"])))
    u = oracle.make_uniform(sample_w, sample_h, cfg["position"], cfg["yaw"], cfg["pitch"],
                            aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]),
                            max_bounces=cfg["max_bounces"], samples=cfg["samples"])
    import threading
    done = threading.Event()

    def heartbeat():            # the big configs' sample frame takes minutes on the CPU: say so (stderr; stdout carries the JSON line only)
        while not done.wait(30.0):
            print(f"[bench] cpu_baseline sample still rendering ({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    try:
        img, _, cnt = sc.render(u, mode=1)
    finally:
        dt = time.perf_counter() - t0
        done.set()
    threads = int(oracle.lib().oracle_threads())       # omp_get_max_threads() of the library that just rendered
    n_node = cnt.node_visits / cnt.rays
    n_tri = cnt.triangle_tests / cnt.rays
    hit = cnt.hits / cnt.rays
    # best of 3 on the timed sample (same scene, camera, spp; the config's aspect)
    ut = oracle.make_uniform(timed_w, timed_h, cfg["position"], cfg["yaw"], cfg["pitch"],
                             aspect=np.float32(cfg["width"]) / np.float32(cfg["height"]), max_bounces=cfg["max_bounces"], samples=cfg["samples"])
    runs = []
    for _ in range(3):
        t1 = time.perf_counter()
        _, _, c3 = sc.render(ut, mode=1)
        runs.append((time.perf_counter() - t1, c3.rays))
    best_s, best_rays = min(runs)
    # brute-force audit: the BVH mode that checks the big frame has lost hits before, so random pixels of
    # it are rendered again with every primitive tested against every ray
    rng = np.random.default_rng(20251112)
    audit = []
    brute = np.array(img, copy=True)
    for _ in range(audit_pixels):
        x, y = int(rng.integers(0, sample_w)), int(rng.integers(0, sample_h))
        brute[y, x] = 0.0
        sc.render(u, brute, mode=0, rows=(y, y + 1), cols=(x, x + 1))
        audit.append((x, y))
    return {"mrays": best_rays / best_s / 1e6, "timed_seconds": [r[0] for r in runs], "timed_rays": best_rays,
            "full_frame_mrays": cnt.rays / dt / 1e6, "brute_image": brute, "audit": audit,
            "seconds": dt, "rays": cnt.rays, "threads": threads, "image": img,
            "n_node": n_node, "n_tri": n_tri, "hit": hit,
            "bytes_per_ray": n_node * S_NODE + n_tri * S_TRI + hit * S_HIT}


def cpu_model():
    """Model name plus what identifies the part when the name is generic (a VM shows "AMD EPYC"): family / model / stepping, cache, CPUs
    this process may use."""
    info = {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (x.strip() for x in line.split(":", 1))
                if k in ("model name", "cpu family", "model", "stepping", "cache size") and k not in info:
                    info[k] = v
            elif info:
                break
    except OSError:
        pass
    name = info.get("model name", "unknown")
    ident = ", ".join(f"{k} {info[k]}" for k in ("cpu family", "model", "stepping", "cache size") if k in info)
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 0
    return f"{name} ({ident}; {usable} CPUs usable by this process)" if ident else name


def self_launch(n):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it): start N fresh child processes,
    one rank per GPU, BEFORE this process touches torch.cuda or HIP; wait; relay rank 0's output (it prints the JSON
    line); fail if any child fails — and if one dies while the others wait for it in a collective, stop those (the
    exact processes started here) instead of hanging until RCCL's timeout.  Nothing is exec'ed and nothing is retried."""
    import tempfile
    # rendezvous through a file store, not a TCP port: a port picked by bind-and-close can be taken by someone else before
    # rank 0 binds it again; a fresh temporary file cannot
    rdv = tempfile.mkdtemp(prefix="rt_bench_rdv_")
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), RT_BENCH_INIT_FILE=os.path.join(rdv, "store"),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else None))
    failed = False
    while True:
        rcs = [p.poll() for p in procs]          # every rank, every round (poll() is what refreshes returncode)
        if any(rc not in (None, 0) for rc in rcs):
            failed = True
            break
        if all(rc is not None for rc in rcs):
            break
        time.sleep(0.2)
    if failed:                                   # a rank died: the rest would wait for it in a collective
        time.sleep(2.0)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    rcs = [p.returncode for p in procs]
    out0.seek(0)
    sys.stdout.write(out0.read().decode("utf-8", "replace"))
    sys.stdout.flush()
    import shutil
    shutil.rmtree(rdv, ignore_errors=True)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py --gpus {n}: rank(s) failed (rank, exit code): {bad}")



def report(args, cfg, rt, ctx, world, dt, rays, kernel_ms, launches, stats, cus, clock_hz, how, extra_config=None):
    """The bench line (rank 0 / the group's process).  ctx: a context that holds the scene (the parity frame is rendered on it)."""
    W, H = cfg["width"], cfg["height"]
    out = {
        "metric": "Mrays/s (node) teapot-circle 4K 8-bounce; % of HBM roofline" if args.config == "C3" else f"Mrays/s (node) config {args.config}",
        "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": f"synthetic camera on {cfg['scene']}",
        "config": {"workload": f"{args.config}: {os.path.basename(cfg['scene'])} {W}x{H}, {cfg['samples']} spp/step x {args.steps} steps, "
                               f"{cfg['max_bounces']} bounces, {how}",
                   "bvh_nodes": int(stats.bvh_nodes), "bvh_gbytes": stats.bvh_bytes / 1e9, "bvh_build_s": stats.bvh_build_seconds,
                   "node_format": int(stats.node_format),
                   "stack_entries": int(stats.stack_entries), "resident_waves_per_simd": int(stats.resident_blocks) * 4 / (cus * 4),
                   "rays": rays, "spp": cfg["samples"] * args.steps,
                   "spp_per_s": cfg["samples"] * args.steps / dt, "msamples_per_s": W * H * cfg["samples"] * args.steps / dt / 1e6},
    }
    if extra_config:
        out["config"].update(extra_config)
    # cpu_baseline: N=1 only.  The roofline's per-ray byte model comes from the same oracle run
    # (counting mode); without that leg it is taken from configs/baseline_configs.json (same measurement, frozen).
    model, model_src = cfg.get("byte_model"), "configs/baseline_configs.json"
    if not args.no_cpu and world == 1:
        sw, sh = (int(x) for x in args.cpu_sample.split("x"))
        tw, th = (int(x) for x in args.cpu_timed.split("x"))
        cpu = cpu_baseline_and_byte_model(cfg, sw, sh, tw, th, audit_pixels=64)
        out["cpu_baseline"] = {"value": cpu["mrays"], "unit": "Mrays/s", "cores": cpu["threads"], "kind": "port", "cpu_model": cpu_model(),
                               "sample": f"oracle/rt_oracle.c (CPU restatement, not lavapipe), BVH mode, OpenMP; best of 3 runs of one {tw}x{th} frame of the "
                                         f"same scene/camera, {cfg['samples']} spp, {cfg['max_bounces']} bounces: {cpu['timed_rays']} rays in "
                                         + " / ".join(f"{x:.2f}" for x in cpu["timed_seconds"]) + f" s; the {sw}x{sh} parity frame ran at {cpu['full_frame_mrays']:.1f} Mrays/s",
                               "lavapipe_probe": lavapipe_probe()}
        model = {"nodes_per_ray": cpu["n_node"], "tris_per_ray": cpu["n_tri"], "hit_fraction": cpu["hit"]}
        model_src = "measured by this run's cpu_baseline leg"
        # the oracle frame is also the checker: the same frame from the HIP path (outside the timed region) must
        # hold the same bits (at the default --cpu-sample that is one full-size frame)
        ctx.set_shard(0, 1)
        ctx.accum_resize(sw, sh)
        ctx.stats_reset()
        ctx.dispatch(rt.make_uniform(sw, sh, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(W) / np.float32(H),
                                     max_bounces=cfg["max_bounces"], samples=cfg["samples"]))
        got = ctx.accum_download()
        ax = np.array([a[0] for a in cpu["audit"]]), np.array([a[1] for a in cpu["audit"]])
        out["parity"] = {"what": f"one {sw}x{sh} frame, HIP accumulation buffer vs oracle (BVH mode), bit for bit; {len(cpu['audit'])} random pixels of it "
                                 "also vs the oracle's BRUTE FORCE (no tree)",
                         "identical": bool(np.array_equal(got.view(np.uint32), cpu["image"].view(np.uint32))),
                         "rays_equal": int(ctx.stats().rays) == int(cpu["rays"]),
                         "brute_force_pixels_identical": bool(np.array_equal(got[ax[1], ax[0]].view(np.uint32), cpu["brute_image"][ax[1], ax[0]].view(np.uint32)))}
    if model is not None:
        bytes_per_ray = model["nodes_per_ray"] * S_NODE + model["tris_per_ray"] * S_TRI + model["hit_fraction"] * S_HIT
        rays_per_launch = rays / launches / world         # per GPU, per launch
        frames_per_launch = args.steps / launches
        algo_bytes = rays_per_launch * bytes_per_ray + (W * H / world) * frames_per_launch * S_FB
        algo_gbs = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # HBM bytes per launch from the rocprofv3 PMC passes of the same command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
        # MI355X_MICROARCH.md §HBM), recorded by tools/pmc_summary.py with the digest of the device sources they were taken
        # with.  A profile of OTHER sources is stale: it is still printed, flagged, and not used for `achieved`.
        stamp = source_stamp()
        traffic, traffic_rec = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic_rec = json.load(open(tpath)).get(args.config)
            if traffic_rec:
                per_frame = traffic_rec.get("hbm_bytes_per_frame", traffic_rec.get("hbm_bytes_per_launch"))
                traffic = None if per_frame is None else per_frame * frames_per_launch / world
        traffic_stale = bool(traffic_rec) and traffic_rec.get("kernel_source_sha256") != stamp
        use_traffic = traffic is not None and not traffic_stale
        # `achieved`/`frac`: what HBM really carries (measured traffic / kernel time) against its peak — the fraction
        # that can bind.  The SURVEY §8(d) per-ray byte model (what a cache-less machine would read) is kept under
        # `algorithmic`: on a cache-resident scene it exceeds the HBM peak because L1/L2 serve those bytes.
        hbm_gbs = traffic / (kernel_ms * 1e-3) / 1e9 if use_traffic else None
        out["roofline"] = {"bound": "hbm", "achieved": hbm_gbs if hbm_gbs is not None else algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": (hbm_gbs if hbm_gbs is not None else algo_gbs) / HBM_PEAK_GBS, "traffic": traffic,
                           "achieved_is": "measured HBM traffic / kernel time" if hbm_gbs is not None else
                                          ("algorithmic bytes / kernel time (the recorded traffic is of other sources: stale)" if traffic_stale else "algorithmic bytes / kernel time (config never profiled)"),
                           "kernel": "rt_render_phased", "kernel_ms": kernel_ms, "frames_per_launch": frames_per_launch,
                           "provenance": {"kernel_source_sha256": stamp,
                                          "traffic_profile": None if not traffic_rec else {k: traffic_rec.get(k) for k in ("profile", "kernel_source_sha256", "git_head_at_reduction", "frames_per_launch")},
                                          "stale": traffic_stale},
                           "algorithmic": {"bytes_per_launch": algo_bytes, "achieved": algo_gbs, "ratio_to_hbm_peak": algo_gbs / HBM_PEAK_GBS,
                                           "bytes_per_ray": bytes_per_ray, "nodes_per_ray": model["nodes_per_ray"],
                                           "tris_per_ray": model["tris_per_ray"], "hit_fraction": model["hit_fraction"], "byte_model": model_src,
                                           "note": "SURVEY section 8(d) model: 32 B per box test, 36 B per triangle test, 284 B per hit, 32 B per pixel-frame; "
                                                   "a ratio above 1 (it is a ratio, not a utilisation) means the records are cache-resident and these bytes never reach HBM"}}
        # The on-chip ceilings (profiles/onchip.json, reduced from the PMC passes of the same command by tools/pmc_summary.py), each in
        # its counter's own unit against a peak calibrated with known instruction / lane-load counts (profiles/r3_calibration.txt):
        #   L1 path:  CUs x clock x 1.50 TCP_TOTAL_CACHE_ACCESSES per nominal clock per CU / the kernel's accesses per ray
        #   VALU:     SIMDs x clock x 0.392 SQ_INSTS_VALU per nominal clock per SIMD (fastest instruction class; FMA 0.320) / the kernel's wave instructions per ray
        opath = os.path.join(ROOT, "profiles", "onchip.json")
        if os.path.exists(opath):
            oc = json.load(open(opath)).get(args.config)
            if oc:
                stale = oc.get("kernel_source_sha256") != stamp
                achieved_mrays = rays / launches / world / (kernel_ms * 1e-3) / 1e6
                l1_peak = oc["l1_tag_lookups_per_clk_per_cu_peak"]
                l1 = l1_peak * cus * clock_hz / oc["l1_tag_lookups_per_ray"] / 1e6
                valu_peak, fma_peak = oc.get("valu_insts_per_simd_clk_peak", 0.392), oc.get("valu_insts_per_simd_clk_fma_peak", 0.320)
                valu = valu_peak * cus * 4 * clock_hz / oc["valu_wave_insts_per_ray"] / 1e6
                out["roofline"]["onchip"] = {
                    "bound": "l1_tag_lookups", "ceiling_mrays": l1, "achieved_mrays": achieved_mrays, "frac": achieved_mrays / l1,
                    "l1_tag_lookups_per_ray": oc["l1_tag_lookups_per_ray"], "l1_tag_lookups_per_clk_per_cu_peak": l1_peak,
                    "valu": {"ceiling_mrays": valu, "frac": achieved_mrays / valu, "rate_vs_all_fma_stream": achieved_mrays / (valu * fma_peak / valu_peak),
                             "wave_insts_per_ray": oc["valu_wave_insts_per_ray"], "insts_per_simd_clk_peak": valu_peak, "insts_per_simd_clk_fma_peak": fma_peak,
                             "insts_per_simd_clk_in_profile": oc.get("measured_valu_insts_per_simd_clk"), "lane_utilization": oc.get("valu_lane_utilization")},
                    "l2_read_GBps_in_profile": oc.get("l2_read_GBps"), "l2_hit_rate_in_profile": oc.get("l2_hit_rate"), "waves_per_simd_in_profile": oc.get("waves_per_simd"),
                    "kernel_ms_in_profile": oc.get("kernel_ms_in_profile"),
                    "cus": cus, "clock_hz": clock_hz, "profile": oc.get("profile"), "kernel_source_sha256_of_profile": oc.get("kernel_source_sha256"),
                    "git_head_at_reduction": oc.get("git_head_at_reduction"), "stale": stale,
                    "note": "ceilings = units x nominal clock x calibrated peak rate of the counter (profiles/r3_calibration.txt: 1.50 TCP_TOTAL_CACHE_ACCESSES per clock per CU for "
                            "divergent 16-byte lane-loads that hit L1; 0.392 SQ_INSTS_VALU per clock per SIMD for the fastest VALU class, 0.320 for v_fma_f32; all per NOMINAL clock, "
                            "i.e. per unit of time) / the kernel's own count per ray (profiles/onchip.json).  stale = the profile was taken with other device sources than this run's"}
    return out


def run_group(args):
    """--mode group: ONE process, N devices behind one rt_group (SURVEY section 8(b) `create(device_ids[], n)`; the reference's
    single context owning everything, src/main.rs:27-59).  The scene is generated, built and repacked ONCE and uploaded to every
    device; rt_group_dispatch_frames launches every shard on its own device and stream; the timed region ends when the bands have
    been gathered (peer copies over xGMI) and de-interleaved on the first device.  RT_BENCH_GROUP_DEVICES="0,0" rehearses
    several shards on the GPUs there are."""
    import ctypes as C
    rt = graft.load_package()
    cfg = graft.load_config(args.config)
    W, H = cfg["width"], cfg["height"]
    hip = C.CDLL("libamdhip64.so")
    n_dev = C.c_int(0)
    if hip.hipGetDeviceCount(C.byref(n_dev)) != 0 or n_dev.value < 1:
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    if os.environ.get("RT_BENCH_GROUP_DEVICES"):
        devices = [int(x) for x in os.environ["RT_BENCH_GROUP_DEVICES"].split(",")]
        if len(devices) != args.gpus:
            raise SystemExit(f"RT_BENCH_GROUP_DEVICES names {len(devices)} devices, --gpus {args.gpus}")
    else:
        if n_dev.value < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs {args.gpus} GPUs, this node shows {n_dev.value} (RT_BENCH_GROUP_DEVICES=0,0,... rehearses on the GPUs there are)")
        devices = list(range(args.gpus))
    world = len(devices)
    t_setup = time.perf_counter()
    g = rt.Group(devices)
    for kv in filter(None, os.environ.get("RT_BENCH_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        g.set_option(int(k), int(v))
    if cfg["scene"].startswith("synth:"):
        synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
        syn = synth.load_config_scene(cfg, ROOT)
        g.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        if len(syn.spheres):
            g.spheres_upload(syn.spheres)
        del syn
    else:
        hs = rt.HostScene(os.path.join(ROOT, cfg["scene"]))
        g.scene_upload(hs.vertices, hs.indices, hs.primitives, hs.models)
        if hs.textures:
            g.textures_upload(hs.textures)
    setup_s = time.perf_counter() - t_setup
    g.accum_resize(W, H)
    shard = [C.c_void_p() for _ in devices]
    for i in range(world):
        rt._check(rt.lib().rt_group_context(g.handle, i, C.byref(shard[i])))
        rt._check(rt.lib().rt_frames_reserve(shard[i], args.steps))       # staging allocated here, not in the timed region

    def fresh_uniform():
        u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"], samples=cfg["samples"])
        u["frame"], u["accumulation_frame"] = 0, 1
        return u

    if args.warmup > 0:
        g.dispatch_frames(fresh_uniform(), args.warmup)
    g.accum_gather()
    # timed: exactly K steps from a cold accumulation buffer + the gather on device 0
    g.accum_resize(W, H)
    g.stats_reset()
    g.sync()
    u = fresh_uniform()
    t0 = time.perf_counter()
    g.dispatch_frames(u, args.steps)
    g.accum_gather()                       # waits for every shard, peer-copies and de-interleaves, then synchronises device 0
    dt = time.perf_counter() - t0
    stats = g.stats()
    launches = max(int(stats.launches) // world, 1)                      # per shard
    kernel_ms = max(g.shard_ms(i) for i in range(world)) / launches      # HIP events on each shard's stream; the slowest shard
    rays = int(stats.rays)
    assert int(stats.pixel_frames) == W * H * args.steps, (int(stats.pixel_frames), W * H * args.steps)
    cus, clock_khz = C.c_int(0), C.c_int(0)
    hip.hipDeviceGetAttribute(C.byref(cus), 63, devices[0])              # hipDeviceAttributeMultiprocessorCount
    hip.hipDeviceGetAttribute(C.byref(clock_khz), 5, devices[0])         # hipDeviceAttributeClockRate
    peer_names = {0: "same device", 1: "peer access enabled", 2: "peer access unavailable (host-staged copy)", 3: "peer access could not be enabled (host-staged copy)"}
    out = report(args, cfg, rt, _BorrowedContext(rt, shard[0]), world, dt, rays, kernel_ms, launches, stats,
                 cus=cus.value or 256, clock_hz=(clock_khz.value or 2400000) * 1e3,
                 how=f"8-row bands round-robin over {world} GPU(s), ONE process driving all of them through rt_group, one peer-copy gather",
                 extra_config={"mode": "group", "devices": devices, "peer_access": [peer_names.get(p, str(p)) for p in g.peer_access()],
                               "scene_setup_s": setup_s, "host_builds": int(stats.bvh_rebuilds)})
    print(json.dumps(out), flush=True)
    if args.verify:
        got = g.accum_download()
        ref = rt.Context(devices[0])
        if cfg["scene"].startswith("synth:"):
            synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
            syn = synth.load_config_scene(cfg, ROOT)
            ref.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
            if len(syn.spheres):
                ref.spheres_upload(syn.spheres)
        else:
            ref.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
        ref.set_option(rt.RT_OPT_FRAME_PIPELINE, 0)
        ref.accum_resize(W, H)
        ref.dispatch_frames(fresh_uniform(), args.steps)
        want = ref.accum_download()
        ok = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        print(json.dumps({"verify": "gathered image == unsharded render", "n_gpus": world, "mode": "group", "identical": ok}), flush=True)
        ref.close()
        if not ok:
            raise SystemExit("verification failed: the group's image differs from the unsharded render")
    g.close()


class _BorrowedContext:
    """A shard's rt_context seen through the single-context calls report() needs; owned by the group, never destroyed here."""

    def __init__(self, rt, handle):
        self._c = rt.Context.__new__(rt.Context)
        self._c.handle, self._c.width, self._c.height = handle, 0, 0

    def __getattr__(self, name):
        return getattr(self._c, name)

    def __del__(self):
        self._c.handle = None              # keep Context.__del__ from destroying what the group owns

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--cpu-sample", default="3840x2160", help="resolution of the oracle frame the HIP path is checked against bit for bit")
    ap.add_argument("--cpu-timed", default="1920x1080", help="resolution of the frame whose best of 3 oracle runs is cpu_baseline.value")
    ap.add_argument("--mode", choices=("auto", "ranks", "group"), default="auto",
                    help="ranks: one process per GPU, torch.distributed gather (what the driver launches); group: ONE process drives all N GPUs "
                         "through rt_group (one scene generation and host build, peer-copy gather); auto: group for synthetic scenes typed without a launcher")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="after timing, rank 0 re-renders unsharded and compares bit for bit")
    args = ap.parse_args()

    mode = args.mode
    if mode == "auto":
        synthetic = graft.load_config(args.config)["scene"].startswith("synth:")
        mode = "group" if (synthetic and args.gpus > 1 and "WORLD_SIZE" not in os.environ) else "ranks"
    if mode == "group":
        if int(os.environ.get("RANK", "0")) != 0:
            return                               # under a launcher: rank 0 drives every GPU, the other ranks have nothing to do
        return run_group(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)

    if os.environ.get("RT_BENCH_SELFTEST") == "rank0-waits" and os.environ.get("RANK") == "0":
        time.sleep(300)      # tests/test_host_side.py: a rank that waits (as in a collective) for one that has died
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # RT_BENCH_BACKEND=gloo is a REHEARSAL mode for a one-GPU box: all ranks share the visible GPU(s) and the
    # collectives run on CPU copies.  The real multi-GPU run uses nccl (= RCCL) and one GPU per rank.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} GPUs, this node shows {torch.cuda.device_count()} "
                         "(RT_BENCH_BACKEND=gloo rehearses the ranks on the GPUs there are)")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")     # where collective operands live
    # RT_BENCH_FORCE_DIST=1: a process group, barrier, reductions and the gather even for ONE rank — on a one-GPU box this is the only
    # way the nccl (= RCCL) branch below ever executes (tests/test_gpu_parity.py::test_bench_rccl_path_with_one_rank)
    force_dist = os.environ.get("RT_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ and not os.environ.get("RT_BENCH_INIT_FILE"):
            import tempfile
            os.environ["RT_BENCH_INIT_FILE"] = os.path.join(tempfile.mkdtemp(prefix="rt_bench_rdv_"), "store")
        kw = {}
        if os.environ.get("RT_BENCH_INIT_FILE"):          # started by self_launch: file store
            kw = dict(init_method="file://" + os.environ["RT_BENCH_INIT_FILE"], rank=rank, world_size=world)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, **kw)
        else:
            dist.init_process_group(backend, **kw)

    rt = graft.load_package()
    sharding = __import__("ray_tracing_amd.sharding", fromlist=["x"])
    cfg = graft.load_config(args.config)
    W, H = cfg["width"], cfg["height"]

    ctx = rt.Context(device_index)
    for kv in filter(None, os.environ.get("RT_BENCH_OPTIONS", "").split(",")):     # developer knob: "key=value,..." for rt_set_option
        k, v = kv.split("=")
        ctx.set_option(int(k), int(v))
    if cfg["scene"].startswith("synth:"):
        synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
        syn = synth.load_config_scene(cfg, ROOT)
        ctx.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
        if len(syn.spheres):
            ctx.spheres_upload(syn.spheres)
        del syn
    else:
        ctx.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
    # One explicit stream carries everything: torch ops, the render kernels (rt_set_stream) and — through
    # torch.distributed's stream ordering — the RCCL gather.  (torch's default stream is the null stream,
    # which rt_set_stream reads as "use the library's own stream"; that would leave the gather unordered
    # against the kernels.)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx.set_stream(stream.cuda_stream)
    ctx.set_shard(rank, world)
    local = torch.zeros(sharding.local_buffer_floats(W, H, world), dtype=torch.float32, device=dev)

    def fresh_uniform():
        u = rt.make_uniform(W, H, cfg["position"], cfg["yaw"], cfg["pitch"], max_bounces=cfg["max_bounces"],
                            samples=cfg["samples"])
        u["frame"] = 0                 # cold start, src/main.rs:76-77: first dispatch sees (1, 0)
        u["accumulation_frame"] = 1
        return u

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up: W untimed steps (+ one gather so RCCL's channels exist)
    ctx.accum_bind(local.data_ptr(), local.numel() * 4, W, H)
    ctx.frames_reserve(args.steps)       # staging of the K-frame launch: allocated here, not in the timed region
    u = fresh_uniform()
    if args.warmup > 0:
        ctx.dispatch_frames(u, args.warmup)
    def gather():
        src = local if cdev.type == "cuda" else local.cpu()
        return sharding.gather_framebuffer(src, W, H, rank, world, force_collective=force_dist)

    img = gather()
    del img
    barrier()

    # timed: exactly K steps from a cold accumulation buffer + the final gather
    local.zero_()
    ctx.accum_bind(local.data_ptr(), local.numel() * 4, W, H)
    ctx.stats_reset()
    u = fresh_uniform()
    barrier()
    t0 = time.perf_counter()
    ctx.dispatch_frames(u, args.steps)
    img = gather()
    barrier()
    dt = time.perf_counter() - t0

    stats = ctx.stats()
    launches = max(int(stats.launches), 1)                 # rt_dispatch_frames folds the K frames into one launch (or a few)
    kernel_ms = ctx.last_dispatch_ms() / launches          # HIP events on the launch stream, per launch
    t = torch.tensor([dt], dtype=torch.float64, device=cdev)
    r = torch.tensor([stats.rays, stats.pixel_frames], dtype=torch.int64, device=cdev)
    k = torch.tensor([kernel_ms], dtype=torch.float64, device=cdev)
    if world > 1 or force_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    rays, pixel_frames = int(r[0].item()), int(r[1].item())
    kernel_ms = float(k.item())

    if rank == 0:
        assert pixel_frames == W * H * args.steps, (pixel_frames, W * H * args.steps)
        assert img is not None and bool(torch.isfinite(img[..., :3]).all())
        props = torch.cuda.get_device_properties(dev)
        out = report(args, cfg, rt, ctx, world, dt, rays, kernel_ms, launches, stats,
                     cus=int(props.multi_processor_count), clock_hz=float(getattr(props, "clock_rate", 2.4e6)) * 1e3,
                     how=f"8-row bands round-robin over {world} GPU(s), one process per GPU, one framebuffer gather (torch.distributed)")
        print(json.dumps(out), flush=True)

    if args.verify and rank == 0:
        # the gathered multi-rank image must equal a 1-shard render of the same frames bit for bit
        ref = rt.Context(device_index)
        if cfg["scene"].startswith("synth:"):
            synth = __import__("ray_tracing_amd.synth", fromlist=["x"])
            syn = synth.load_config_scene(cfg, ROOT)
            ref.scene_upload(syn.vertices, syn.indices, syn.primitives, syn.models)
            if len(syn.spheres):
                ref.spheres_upload(syn.spheres)
        else:
            ref.scene_finish(rt.HostScene(os.path.join(ROOT, cfg["scene"])))
        ref.set_option(rt.RT_OPT_FRAME_PIPELINE, 0)      # one launch per frame, blended in place
        ref.accum_resize(W, H)
        ref.dispatch_frames(fresh_uniform(), args.steps)
        want = ref.accum_download()
        got = img.cpu().numpy()
        ok = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        print(json.dumps({"verify": "gathered image == unsharded render", "n_gpus": world, "identical": ok}), flush=True)
        ref.close()
        if not ok:
            raise SystemExit("verification failed: sharded image differs from the unsharded render")

    ctx.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
