#!/usr/bin/env python3
"""Summarise a profiles/run_pmc_extra.sh output directory (TA/TD/TCP busy and stall counters).
usage: python tools/mem_pipe_summary.py gpurun_out/prof_<tag> profiles/<dir> [frames-per-launch]"""
import collections, csv, glob, json, os, sys
src, out = sys.argv[1], sys.argv[2]
vals = {}
for f in sorted(glob.glob(os.path.join(src, 'pmc_*/*/*counter_collection.csv'))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'rt_render' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        vals[k] = sum(v) / len(v)
cyc = vals['GRBM_GUI_ACTIVE'] / 8          # summed over the 8 XCDs
vals['derived'] = {'cycles_per_launch': cyc, 'frames_per_launch': int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                   'TA_busy': vals['TA_TA_BUSY_sum'] / (256 * cyc), 'TD_busy': vals['TD_TD_BUSY_sum'] / (256 * cyc)}
os.makedirs(out, exist_ok=True)
json.dump(vals, open(os.path.join(out, 'pmc_summary.json'), 'w'), indent=1)
print(json.dumps(vals['derived']))
