#!/bin/bash
# per-wave LDS line buffers for the staging planes (RT_STAGE_LINES), against the shipping library.
# For every tools/ab/librt_amd_<name>.so named: the multi-frame parity tests, WRITE_SIZE of a C3 16-frame launch, then timings (two rounds, interleaved)
# usage: bash tools/gpu/stage_lines.sh "ship SL16 SL32"
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
LIBS=$1
for l in $LIBS; do
  if [ "$l" = ship ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/ab/librt_amd_$l.so; fi
  if [ "$l" != ship ]; then
    timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py tests/test_gpu_group.py -x -q -k "frames or static or group" --timeout 300 --timeout-method thread -p no:cacheprovider > $OUT/sl_tests_$l.log 2>&1 || { echo "tests failed on $l"; tail -20 $OUT/sl_tests_$l.log; exit 1; }
    tail -1 $OUT/sl_tests_$l.log
  fi
  rm -rf $OUT/sl_pmc_$l
  for cfg in "C3 16" "C4 4"; do
    set -- $cfg
    timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/sl_pmc_$l/$1 -- python3 bench.py --config $1 --steps $2 --warmup $2 --no-cpu > $OUT/sl_pmc_${l}_$1.log 2>&1 || { echo "pmc failed on $l"; tail -5 $OUT/sl_pmc_${l}_$1.log; exit 1; }
    python3 - $OUT/sl_pmc_$l/$1 $l $1 <<'PY'
import csv, glob, sys
tot = {}; n = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rt_render_phased" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            k = r["Dispatch_Id"]; tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
v = sorted(tot.values())
print(sys.argv[2], sys.argv[3], "WRITE_SIZE per launch (KiB x 1024): %s" % ", ".join("%.1f MB" % (x * 1024 / 1e6) for x in v), flush=True)
PY
  done
done
unset RT_AMD_LIB
bash tools/gpu/sweep_libs.sh "$LIBS" "--config C3 --steps 16 --warmup 16" "--config C4 --steps 4 --warmup 4"
echo stage_lines done
