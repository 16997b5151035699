#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: kernel stats + per-launch PMC means + derived figures.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [out_dir [config-key-for-profiles/traffic.json and onchip.json [frames-per-launch]]]
The profiled command (profiles/run_profile.sh) makes every rt_render launch the same size: --steps S --warmup S."""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Calibrated peaks (profiles/r3_calibration.txt, tools/microbench/calibrate.hip: independent instruction streams, 6 waves per SIMD;
# fully divergent 16-byte lane-loads that hit L1), in the counters' OWN units:
# All three are RATES PER NOMINAL CLOCK (hipDeviceAttributeClockRate, 2.4 GHz), i.e. per unit of TIME: the microbenchmark divides its
# known counts by its own un-profiled HIP-event time, and so does this reducer with the kernel's (trace-pass time).  GRBM_GUI_ACTIVE / 8
# is NOT used as the clock: under the VALU streams it reads 0.76 of the nominal clock (the part runs them at about 1.83 GHz), under the
# gather kernels 1.00 — a fraction formed with it would compare rates at different clocks.
NOMINAL_CLOCK_HZ = 2.4e9
VALU_PEAK_INSTS_PER_SIMD_CLK = 0.392     # SQ_INSTS_VALU per nominal clock per SIMD of the fastest class (v_add_u32, v_mul_f32: 2.55 clocks); v_fma_f32 0.320, v_fma_mix / v_max3 / v_alignbit / v_cvt_f32_ubyte / v_min 0.23-0.24, v_mul_lo_u32 0.22, v_rcp / v_sqrt 0.122
VALU_FMA_INSTS_PER_SIMD_CLK = 0.320
L1_PEAK_ACCESSES_PER_CU_CLK = 1.50       # TCP_TOTAL_CACHE_ACCESSES_sum per nominal clock per CU (0.982 per divergent 16-byte lane-load x 1.502 lane-loads; one access per QUAD of lanes that shares a line)


def source_stamp():
    """What ties a profile to a build: sha256 over the device sources, and the git head at the time of the reduction."""
    h = hashlib.sha256()
    for f in ("rt_kernels.hip", "rt_device.h", "rt_pack.h", "rt_math.h"):
        h.update(open(os.path.join(ROOT, "ray-tracing_amd", "csrc", f), "rb").read())
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        head = ""
    return {"kernel_source_sha256": h.hexdigest(), "git_head_at_reduction": head}


src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
vals, n = {}, {}
for f in sorted(glob.glob(os.path.join(src, 'pmc_*/*/*counter_collection.csv'))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rt_render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v); n[k] = len(v)
stats = glob.glob(os.path.join(src, 'trace/*/*kernel_stats.csv'))
if len(stats) != 1:      # gpurun merges every call's files into gpurun_out/: two runs under one tag would be averaged together
    sys.exit(f"{src}: {len(stats)} trace passes found, expected 1 (delete gpurun_out/prof_<tag> before profiling under the same tag again)")
stats = stats[0]
kern = [r for r in csv.DictReader(open(stats)) if 'rt_render' in r['Name']][0]
avg_ms = float(kern['AverageNs']) / 1e6
d = {"kernel": kern['Name'], "calls": int(kern['Calls']), "avg_ms": avg_ms, "pmc_mean_per_launch": vals}
gui = vals.get('GRBM_GUI_ACTIVE', 0) / 8
cyc = avg_ms * 1e-3 * NOMINAL_CLOCK_HZ            # nominal clocks of one launch (time base of the calibrated peaks)
if gui:
    d["derived"] = {
        "grbm_gui_active_clock_GHz": gui / (avg_ms * 1e-3) / 1e9,
        "rate_time_base": "counter / (kernel time from the trace pass x 2.4 GHz nominal)",
        # VALU issue rate: wave64 VALU instructions per clock per SIMD (SQ_INSTS_VALU counts exactly one per wave-instruction:
        # profiles/r3_calibration.txt), against the CALIBRATED peaks of independent instruction streams at 6 waves per SIMD.
        # No clamp: the fraction of the fastest class's peak cannot exceed 1.  valu_rate_vs_all_fma_stream is a RATIO of two rates, not a
        # utilisation: the kernel's issue rate over that of a stream of nothing but v_fma_f32 (0.320), which a mix holding two-operand
        # adds / multiplies (0.392) can exceed.
        # (Rounds 1 and 2 divided by a nominal 0.25 — a dependent-chain figure — and got 1.14 / 1.21.)
        "valu_insts_per_simd_clk": vals['SQ_INSTS_VALU'] / 1024 / cyc,
        "valu_frac_of_peak": vals['SQ_INSTS_VALU'] / 1024 / cyc / VALU_PEAK_INSTS_PER_SIMD_CLK,
        "valu_rate_vs_all_fma_stream": vals['SQ_INSTS_VALU'] / 1024 / cyc / VALU_FMA_INSTS_PER_SIMD_CLK,
        # SQ_ACTIVE_INST_VALU counts 1 per 4-clock instruction and 2 per 8-clock one (v_rcp, v_sqrt, ...): the share of 8-clock instructions
        "valu_8clk_inst_fraction": vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_INSTS_VALU'] - 1.0,
        "valu_lane_utilization": vals['SQ_THREAD_CYCLES_VALU'] / (vals['SQ_ACTIVE_INST_VALU'] * 64),
        "waves_per_simd": vals['SQ_WAVE_CYCLES'] * 4 / 1024 / gui,
        "l2_hit_rate": vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']),
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) reads exactly half of a wide streaming read on gfx950 -> x2;
        # WRITE_SIZE (KiB) is exact.  Scattered record fetches, calibrated (profiles/r3_calibration_hbm.txt): one L2 miss = one fabric request =
        # one 128-byte line, tallied at 64 bytes whether 64 or 128 bytes of it are wanted -> x2 is the memory side's traffic here too (raw = the
        # bytes a 64-byte record wanted).  Both figures are given.
        "hbm_read_bytes_raw": vals['FETCH_SIZE'] * 1024, "hbm_read_bytes_x2": vals['FETCH_SIZE'] * 2048,
        "hbm_write_bytes": vals['WRITE_SIZE'] * 1024,
        "hbm_GBps_x2": (vals['FETCH_SIZE'] * 2048 + vals['WRITE_SIZE'] * 1024) / (avg_ms * 1e-3) / 1e9,
        "l2_read_GBps": vals['TCP_TCC_READ_REQ_sum'] * 64 / (avg_ms * 1e-3) / 1e9,
    }
# rays of the profiled launch: the bench line the trace pass printed
rays = None
try:
    for line in open(os.path.join(src, 'trace.log')):
        if line.startswith('{') and '"metric"' in line:
            b = json.loads(line)
            rays = b["config"]["rays"]
except OSError:
    pass
if rays and gui:
    d["per_ray"] = {"rays_per_launch": rays,
                    "valu_wave_insts_per_ray": vals['SQ_INSTS_VALU'] / rays,
                    "valu_lane_insts_per_ray": vals['SQ_THREAD_CYCLES_VALU'] / rays if 'SQ_THREAD_CYCLES_VALU' in vals else None,
                    "l1_tag_lookups_per_ray": vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / rays,
                    "vmem_wave_loads_per_ray": vals['SQ_INSTS_VMEM_RD'] / rays,
                    "l2_requests_per_ray": vals['TCP_TCC_READ_REQ_sum'] / rays,
                    "lds_wave_insts_per_ray": vals.get('SQ_INSTS_LDS', 0) / rays}
    d["derived"]["l1_tag_lookups_per_clk_per_cu"] = vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc
    d["derived"]["l1_frac_of_peak"] = vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc / L1_PEAK_ACCESSES_PER_CU_CLK
    d["derived"]["lds_bank_conflict_fraction"] = vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE'] if vals.get('SQ_LDS_IDX_ACTIVE') else None
print(json.dumps(d, indent=1))
if out and len(sys.argv) > 3:
    # record the per-launch HBM traffic for bench.py's roofline.traffic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tp = os.path.join(root, "profiles", "traffic.json")
    t = json.load(open(tp)) if os.path.exists(tp) else {}
    fpl = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    t[sys.argv[3]] = {"hbm_bytes_per_launch": d["derived"]["hbm_read_bytes_x2"] + d["derived"]["hbm_write_bytes"],
                      "frames_per_launch": fpl,
                      "hbm_bytes_per_frame": (d["derived"]["hbm_read_bytes_x2"] + d["derived"]["hbm_write_bytes"]) / fpl,
                      "read_bytes_fetch_size_x2": d["derived"]["hbm_read_bytes_x2"], "write_bytes": d["derived"]["hbm_write_bytes"],
                      "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE KiB x 2 (gfx950 reads half), WRITE_SIZE KiB x 1",
                      "profile": out, **source_stamp()}
    json.dump(t, open(tp, "w"), indent=1)
    if "per_ray" in d:
        # what bench.py's roofline.onchip prices the kernel with (ceilings at the kernel's own instruction counts)
        op = os.path.join(root, "profiles", "onchip.json")
        o = json.load(open(op)) if os.path.exists(op) else {}
        o[sys.argv[3]] = {"valu_wave_insts_per_ray": d["per_ray"]["valu_wave_insts_per_ray"],
                          "valu_insts_per_simd_clk_peak": VALU_PEAK_INSTS_PER_SIMD_CLK, "valu_insts_per_simd_clk_fma_peak": VALU_FMA_INSTS_PER_SIMD_CLK,
                          "l1_tag_lookups_per_ray": d["per_ray"]["l1_tag_lookups_per_ray"],
                          "l1_tag_lookups_per_clk_per_cu_peak": L1_PEAK_ACCESSES_PER_CU_CLK,
                          "frames_per_launch": fpl,
                          "valu_lane_utilization": d["derived"]["valu_lane_utilization"],
                          "measured_l1_tag_lookups_per_clk_per_cu": d["derived"]["l1_tag_lookups_per_clk_per_cu"],
                          "measured_valu_insts_per_simd_clk": d["derived"]["valu_insts_per_simd_clk"],
                          # what BASELINE.md asks to be reported beside the HBM fraction: L2 bandwidth, wave occupancy (VALU busy = the valu entry)
                          "l2_read_GBps": d["derived"]["l2_read_GBps"], "l2_hit_rate": d["derived"]["l2_hit_rate"], "waves_per_simd": d["derived"]["waves_per_simd"],
                          "kernel_ms_in_profile": avg_ms,
                          "method": "rocprofv3 --pmc SQ_INSTS_VALU / TCP_TOTAL_CACHE_ACCESSES_sum per launch divided by the launch's exact device-side ray count; "
                                    "peaks in the counters' own units from profiles/r3_calibration.txt (tools/microbench/calibrate.hip under the same --pmc passes): "
                                    "1.50 TCP_TOTAL_CACHE_ACCESSES per nominal clock per CU (fully divergent 16-byte lane-loads, L1 hits), "
                                    "0.392 / 0.320 SQ_INSTS_VALU per nominal clock per SIMD (independent v_add_u32 / v_fma_f32 streams, 6 waves per SIMD)",
                          "profile": out, **source_stamp()}
        json.dump(o, open(op, "w"), indent=1)
if out:
    os.makedirs(out, exist_ok=True)
    shutil.copy(stats, os.path.join(out, 'kernel_stats.csv'))
    json.dump(d, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
