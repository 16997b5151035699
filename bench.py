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
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CLOCK_HZ = 2.4e9            # MI355X_MICROARCH.md: max engine clock
N_SIMD = 256 * 4            # 256 CUs x 4 SIMD16; a wave64 VALU instruction holds its SIMD for 4 clocks
S_NODE, S_TRI, S_HIT, S_FB = 32, 36, 284, 32   # SURVEY.md §8(d) bytes


def cpu_baseline_and_byte_model(cfg, sample_w, sample_h):
    """Oracle (BVH mode, OpenMP) on a reduced-resolution frame of the same
    scene/camera/spp: gives the CPU Mrays/s and the per-ray node/triangle means."""
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
    return {"mrays": cnt.rays / dt / 1e6, "seconds": dt, "rays": cnt.rays, "threads": threads, "image": img,
            "n_node": n_node, "n_tri": n_tri, "hit": hit,
            "bytes_per_ray": n_node * S_NODE + n_tri * S_TRI + hit * S_HIT}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(n):
    """`python bench.py --gpus N` typed as is (no torch.distributed.run around it): start N fresh child processes,
    one rank per GPU, BEFORE this process touches torch.cuda or HIP; wait; relay rank 0's output (it prints the JSON
    line); fail if any child fails — and if one dies while the others wait for it in a collective, stop those (the
    exact processes started here) instead of hanging until RCCL's timeout.  Nothing is exec'ed and nothing is retried."""
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
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
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f"bench.py --gpus {n}: rank(s) failed (rank, exit code): {bad}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--cpu-sample", default="3840x2160", help="resolution of the cpu_baseline sample frame")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", action="store_true", help="after timing, rank 0 re-renders unsharded and compares bit for bit")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

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
        if world > 1:
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
        return sharding.gather_framebuffer(src, W, H, rank, world)

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
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    rays, pixel_frames = int(r[0].item()), int(r[1].item())
    kernel_ms = float(k.item())

    if rank == 0:
        assert pixel_frames == W * H * args.steps, (pixel_frames, W * H * args.steps)
        assert img is not None and bool(torch.isfinite(img[..., :3]).all())
        out = {
            "metric": "Mrays/s (node) teapot-circle 4K 8-bounce; % of HBM roofline" if args.config == "C3" else f"Mrays/s (node) config {args.config}",
            "value": rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": f"synthetic camera on {cfg['scene']}",
            "config": {"workload": f"{args.config}: {os.path.basename(cfg['scene'])} {W}x{H}, {cfg['samples']} spp/step x {args.steps} steps, "
                                   f"{cfg['max_bounces']} bounces, 8-row bands round-robin over {world} GPU(s), one framebuffer gather",
                       "bvh_nodes": int(stats.bvh_nodes), "bvh_gbytes": stats.bvh_bytes / 1e9, "bvh_build_s": stats.bvh_build_seconds,
                       "stack_entries": int(stats.stack_entries), "resident_waves_per_simd": int(stats.resident_blocks) * 4 / 1024,
                       "rays": rays, "spp": cfg["samples"] * args.steps,
                       "spp_per_s": cfg["samples"] * args.steps / dt, "msamples_per_s": W * H * cfg["samples"] * args.steps / dt / 1e6},
        }
        # cpu_baseline: rank 0 at N=1 only.  The roofline's per-ray byte model comes from the same oracle run
        # (counting mode); without that leg it is taken from configs/baseline_configs.json (same measurement, frozen).
        model, model_src = cfg.get("byte_model"), "configs/baseline_configs.json"
        if not args.no_cpu and world == 1:
            sw, sh = (int(x) for x in args.cpu_sample.split("x"))
            cpu = cpu_baseline_and_byte_model(cfg, sw, sh)
            out["cpu_baseline"] = {"value": cpu["mrays"], "unit": "Mrays/s", "cores": cpu["threads"], "kind": "port", "cpu_model": cpu_model(),
                                   "sample": f"oracle/rt_oracle.c (CPU restatement, not lavapipe), BVH mode, OpenMP; one {sw}x{sh} frame of the "
                                             f"same scene/camera, {cfg['samples']} spp, {cfg['max_bounces']} bounces: {cpu['rays']} rays in {cpu['seconds']:.2f} s"}
            model = {"nodes_per_ray": cpu["n_node"], "tris_per_ray": cpu["n_tri"], "hit_fraction": cpu["hit"]}
            model_src = "measured by this run's cpu_baseline leg"
            # the oracle frame is also the checker: the same frame from the HIP path (outside the timed region) must
            # hold the same bits (at the default --cpu-sample that is one full-size frame)
            ctx.accum_resize(sw, sh)
            ctx.stats_reset()
            ctx.dispatch(rt.make_uniform(sw, sh, cfg["position"], cfg["yaw"], cfg["pitch"], aspect=np.float32(W) / np.float32(H),
                                         max_bounces=cfg["max_bounces"], samples=cfg["samples"]))
            got = ctx.accum_download()
            out["parity"] = {"what": f"one {sw}x{sh} frame, HIP accumulation buffer vs oracle, bit for bit",
                             "identical": bool(np.array_equal(got.view(np.uint32), cpu["image"].view(np.uint32))),
                             "rays_equal": int(ctx.stats().rays) == int(cpu["rays"])}
        if model is not None:
            bytes_per_ray = model["nodes_per_ray"] * S_NODE + model["tris_per_ray"] * S_TRI + model["hit_fraction"] * S_HIT
            rays_per_launch = rays / launches / world         # per GPU, per launch
            algo_bytes = rays_per_launch * bytes_per_ray + (W * H / world) * (args.steps / launches) * S_FB
            algo_gbs = algo_bytes / (kernel_ms * 1e-3) / 1e9
            # HBM bytes per launch from the rocprofv3 PMC passes of the same command (FETCH_SIZE x2 on gfx950 +
            # WRITE_SIZE, MI355X_MICROARCH.md §HBM), recorded by tools/pmc_summary.py; null if never profiled
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                rec = json.load(open(tpath)).get(args.config, {})
                per_frame = rec.get("hbm_bytes_per_frame", rec.get("hbm_bytes_per_launch"))
                traffic = None if per_frame is None else per_frame * args.steps / launches / world
            # `achieved`/`frac`: what HBM really carries (measured traffic / kernel time) against its peak — the fraction
            # that can bind.  The SURVEY §8(d) per-ray byte model (what a cache-less machine would read) is kept under
            # `algorithmic`: on a cache-resident scene it exceeds the HBM peak because L1/L2 serve those bytes.
            hbm_gbs = None if traffic is None else traffic / (kernel_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": hbm_gbs if hbm_gbs is not None else algo_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": (hbm_gbs if hbm_gbs is not None else algo_gbs) / HBM_PEAK_GBS, "traffic": traffic,
                               "achieved_is": "measured HBM traffic / kernel time" if hbm_gbs is not None else "algorithmic bytes / kernel time (config never profiled)",
                               "kernel": "rt_render_phased", "kernel_ms": kernel_ms, "frames_per_launch": args.steps / launches,
                               "algorithmic": {"bytes_per_launch": algo_bytes, "achieved": algo_gbs, "frac": algo_gbs / HBM_PEAK_GBS,
                                               "bytes_per_ray": bytes_per_ray, "nodes_per_ray": model["nodes_per_ray"],
                                               "tris_per_ray": model["tris_per_ray"], "hit_fraction": model["hit_fraction"], "byte_model": model_src,
                                               "note": "SURVEY section 8(d) model: 32 B per box test, 36 B per triangle test, 284 B per hit, 32 B per pixel-frame; "
                                                       "above 1 means the records are cache-resident and these bytes never reach HBM"}}
            # The on-chip ceiling that binds (profiles/onchip.json, reduced from the PMC passes of the same command by
            # tools/pmc_summary.py): the L1 path = tag look-ups per clock per CU (tools/microbench/gather_width.hip) x 256 CUs x
            # clock / the kernel's tag look-ups per ray.
            opath = os.path.join(ROOT, "profiles", "onchip.json")
            if os.path.exists(opath):
                oc = json.load(open(opath)).get(args.config)
                if oc:
                    achieved_mrays = rays / launches / world / (kernel_ms * 1e-3) / 1e6
                    l1 = oc["l1_tag_lookups_per_clk_per_cu_peak"] * 256 * CLOCK_HZ / oc["l1_tag_lookups_per_ray"] / 1e6
                    out["roofline"]["onchip"] = {"bound": "l1_tag_lookups", "ceiling_mrays": l1, "achieved_mrays": achieved_mrays, "frac": achieved_mrays / l1,
                                                 "l1_tag_lookups_per_ray": oc["l1_tag_lookups_per_ray"], "l1_tag_lookups_per_clk_per_cu_peak": oc["l1_tag_lookups_per_clk_per_cu_peak"],
                                                 "valu_wave_insts_per_ray": oc["valu_wave_insts_per_ray"], "valu_insts_per_simd_clk_in_profile": oc.get("measured_valu_insts_per_simd_clk"),
                                                 "valu_lane_utilization": oc.get("valu_lane_utilization"), "clock_hz": CLOCK_HZ, "profile": oc.get("profile"),
                                                 "note": "ceiling = 256 CUs x clock x 1.5 L1 tag look-ups per clock per CU (tools/microbench/gather_width.hip: divergent lane-loads "
                                                         "that hit L1, any width up to 16 B) / the kernel's own look-ups per ray (profiles/onchip.json).  The VALU pipes issue "
                                                         "at their nominal rate too (0.25 wave64 instructions per clock per SIMD), but dummy-instruction builds show the time "
                                                         "follows fetches, not VALU work (DESIGN.md section 5)"}
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
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
