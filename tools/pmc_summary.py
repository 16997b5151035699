#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: kernel stats + per-launch PMC means + derived figures.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [out_dir [config-key-for-profiles/traffic.json and onchip.json [frames-per-launch]]]
The profiled command (profiles/run_profile.sh) makes every rt_render launch the same size: --steps S --warmup S."""
import collections, csv, glob, json, os, shutil, sys
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
stats = glob.glob(os.path.join(src, 'trace/*/*kernel_stats.csv'))[0]
kern = [r for r in csv.DictReader(open(stats)) if 'rt_render' in r['Name']][0]
avg_ms = float(kern['AverageNs']) / 1e6
d = {"kernel": kern['Name'], "calls": int(kern['Calls']), "avg_ms": avg_ms, "pmc_mean_per_launch": vals}
cyc = vals.get('GRBM_GUI_ACTIVE', 0) / 8
if cyc:
    d["derived"] = {
        "shader_clock_GHz": cyc / (avg_ms * 1e-3) / 1e9,
        # VALU issue rate: wave64 VALU instructions per clock per SIMD.  A SIMD16 needs 4 clocks per wave64 instruction (0.25 per
        # clock, tools/microbench/valu_exec.hip), so valu_issue_fraction = rate / 0.25; it can read a little above 1 because
        # GRBM_GUI_ACTIVE / 8 undercounts the clocks of a kernel this long by a few percent (the shader clock it implies is
        # below the 2.4 GHz the SQ counters ran at).  (Round 1 printed this as "valu_busy" = 1.14.)
        "valu_insts_per_simd_clk": vals['SQ_INSTS_VALU'] / 1024 / cyc,
        "valu_issue_fraction": min(1.0, vals['SQ_INSTS_VALU'] * 4 / 1024 / cyc),
        "valu_lane_utilization": vals['SQ_THREAD_CYCLES_VALU'] / (vals['SQ_ACTIVE_INST_VALU'] * 64),
        "waves_per_simd": vals['SQ_WAVE_CYCLES'] * 4 / 1024 / cyc,
        "l2_hit_rate": vals['TCC_HIT_sum'] / (vals['TCC_HIT_sum'] + vals['TCC_MISS_sum']),
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) reads exactly half of a wide streaming read on gfx950 -> x2;
        # WRITE_SIZE (KiB) is exact.  Scattered 16-B loads are uncalibrated: both figures are given.
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
if rays and cyc:
    d["per_ray"] = {"rays_per_launch": rays,
                    "valu_wave_insts_per_ray": vals['SQ_INSTS_VALU'] / rays,
                    "valu_lane_insts_per_ray": vals['SQ_THREAD_CYCLES_VALU'] / 4 / rays if 'SQ_THREAD_CYCLES_VALU' in vals else None,
                    "l1_tag_lookups_per_ray": vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / rays,
                    "vmem_wave_loads_per_ray": vals['SQ_INSTS_VMEM_RD'] / rays,
                    "l2_requests_per_ray": vals['TCP_TCC_READ_REQ_sum'] / rays,
                    "lds_wave_insts_per_ray": vals.get('SQ_INSTS_LDS', 0) / rays}
    d["derived"]["l1_tag_lookups_per_clk_per_cu"] = vals['TCP_TOTAL_CACHE_ACCESSES_sum'] / 256 / cyc
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
                      "profile": out}
    json.dump(t, open(tp, "w"), indent=1)
    if "per_ray" in d:
        # what bench.py's roofline.onchip prices the kernel with (ceilings at the kernel's own instruction counts)
        op = os.path.join(root, "profiles", "onchip.json")
        o = json.load(open(op)) if os.path.exists(op) else {}
        o[sys.argv[3]] = {"valu_wave_insts_per_ray": d["per_ray"]["valu_wave_insts_per_ray"], "valu_clocks_per_wave_inst": 4.0,
                          "l1_tag_lookups_per_ray": d["per_ray"]["l1_tag_lookups_per_ray"],
                          "l1_tag_lookups_per_clk_per_cu_peak": 1.5,
                          "valu_lane_utilization": d["derived"]["valu_lane_utilization"],
                          "measured_l1_tag_lookups_per_clk_per_cu": d["derived"]["l1_tag_lookups_per_clk_per_cu"],
                          "measured_valu_insts_per_simd_clk": d["derived"]["valu_insts_per_simd_clk"],
                          "method": "rocprofv3 --pmc SQ_INSTS_VALU / TCP_TOTAL_CACHE_ACCESSES_sum per launch divided by the launch's exact device-side ray count; "
                                    "peak L1 rate 1.5 tag look-ups per clock per CU from tools/microbench/gather_width.hip (fully divergent 16-byte lane-loads, L1 hits); "
                                    "VALU: 4 clocks per wave64 instruction per SIMD16 (tools/microbench/valu_exec.hip)",
                          "profile": out}
        json.dump(o, open(op, "w"), indent=1)
if out:
    os.makedirs(out, exist_ok=True)
    shutil.copy(stats, os.path.join(out, 'kernel_stats.csv'))
    json.dump(d, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
