#!/usr/bin/env python3
"""Join tools/microbench/calibrate.bin's own report (known instruction / lane-load counts, times) with the rocprofv3 --pmc passes
of the same binary, by dispatch order: what each counter behind bench.py's roofline counts per known unit.
usage: python tools/calibration_summary.py gpurun_out/calib > profiles/r3_calibration.txt
  gpurun_out/calib/plain.txt                      the un-profiled run
  gpurun_out/calib/pmc_*/**/*counter_collection.csv   one directory per --pmc pass"""
import collections
import csv
import glob
import os
import sys

src = sys.argv[1]
rows = []
for line in open(os.path.join(src, "plain.txt")):
    if line.startswith("#"):
        print(line.rstrip())
        continue
    p = [x.strip() for x in line.split("|")]
    if len(p) < 5:
        continue
    unit, count = p[2].split()
    rows.append({"idx": int(p[0]), "what": p[1], "unit": unit, "count": float(count), "ms": float(p[3].split()[0]), "derived": p[4]})
counters = collections.defaultdict(dict)          # dispatch order -> counter -> value
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for order, did in enumerate(sorted(per)):
        counters[order].update(per[did])
print("\n== per dispatch (second repetition of each): known count, time, counters per known unit ==")
for r in rows:
    if " rep 1" not in r["what"]:
        continue
    c = counters.get(r["idx"], {})
    print(f"\n[{r['idx']}] {r['what']}: {r['unit']} = {r['count']:.0f}, {r['ms']:.4f} ms un-profiled; {r['derived']}")
    gui = c.get("GRBM_GUI_ACTIVE")
    for k in sorted(c):
        extra = ""
        if gui and k.startswith("SQ_INSTS_VALU"):
            extra = f"   -> {c[k] / 1024 / (gui / 8):.4f} per clock per SIMD (GRBM_GUI_ACTIVE / 8 clocks, 1024 SIMDs)"
        if gui and k.startswith("TCP_TOTAL_CACHE_ACCESSES"):
            extra = f"   -> {c[k] / 256 / (gui / 8):.4f} per clock per CU"
        print(f"    {k:36s} {c[k]:16.0f}   {c[k] / r['count']:10.5f} per {r['unit'][:-1]}{extra}")
