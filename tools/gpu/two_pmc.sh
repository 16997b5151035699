#!/bin/bash
# VALU counters of one C4 launch for a variant library (lane utilization, instructions per launch)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
for NAME in "$@"; do
  if [ "$NAME" = ship ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/ab/librt_amd_$NAME.so; fi
  rm -rf $OUT/two_pmc_$NAME
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/two_pmc_$NAME -- python3 bench.py --config C4 --steps 4 --warmup 4 --no-cpu > $OUT/two_pmc_$NAME.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/two_pmc_$NAME.log; exit 1; }
  python3 - $OUT/two_pmc_$NAME $NAME <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_render_phased" in r["Kernel_Name"]:
            tot[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items())[-1:]:
    print(sys.argv[2], "VALU insts %.3e" % v["SQ_INSTS_VALU"], "lane util %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)),
          "SALU insts %.3e" % v["SQ_INSTS_SALU"], "wave cycles %.3e" % v["SQ_WAVE_CYCLES"], "wait_inst_any %.3e" % v["SQ_WAIT_INST_ANY"], flush=True)
PY
done
