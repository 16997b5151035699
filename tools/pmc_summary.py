#!/usr/bin/env python3
"""Summarise a profiles/run_profile.sh output directory: kernel stats + per-launch PMC means + derived figures.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> [out_dir [config-key-for-profiles/traffic.json [frames-per-launch]]]
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
        # rocprof's VALUBusy formula (4 cycles per wave64 VALU instruction, 1024 SIMDs).  An estimate: it reads
        # slightly above 1 when the pipes are saturated, because not every VALU instruction holds a pipe 4 cycles.
        "valu_busy": vals['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc,
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
if out:
    os.makedirs(out, exist_ok=True)
    shutil.copy(stats, os.path.join(out, 'kernel_stats.csv'))
    json.dump(d, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
