#!/bin/bash
# round 3, third GPU batch: per-axis slab pads of the 8-ary node test — A/B against the previous library, per-ray visits, parity
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
PREV=$PWD/tools/ab/librt_amd_prev.so
run() { # name, lib ("" = shipping), bench args...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export RT_AMD_LIB=$lib; else unset RT_AMD_LIB; fi
  timeout -k 10 300 python -u bench.py --no-cpu "$@" > $OUT/b3_$name.json 2> $OUT/b3_$name.err || { echo "bench $name failed"; tail -5 $OUT/b3_$name.err; exit 1; }
  python - $OUT/b3_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], "ms_per_step %.2f"%d["ms_per_step"], "Mrays/s %.0f"%d["value"], "rays", d["config"]["rays"])
PY
}
run c4s4_prev $PREV --config C4 --steps 4 --warmup 4
run c4s4_new  ""    --config C4 --steps 4 --warmup 4
run c4s8_prev $PREV --config C4 --steps 8 --warmup 8
run c4s8_new  ""    --config C4 --steps 8 --warmup 8
run c5qs4_prev $PREV --config C5q --steps 4 --warmup 4
run c5qs4_new  ""    --config C5q --steps 4 --warmup 4
run c3_prev $PREV --config C3 --steps 16 --warmup 16
run c3_new  ""    --config C3 --steps 16 --warmup 16
unset RT_AMD_LIB
timeout -k 10 250 python -u tools/exp_ray_order/run.py --config C4 --frames 4 --sample 1 --bounce 2 --per-ray > $OUT/exp_perray_s1b2_new.log 2>&1 || { echo "per-ray failed"; tail $OUT/exp_perray_s1b2_new.log; exit 1; }
grep -E "node visits per ray|above 1000" $OUT/exp_perray_s1b2_new.log
timeout -k 10 600 python -u -m pytest tests/test_gpu_fuzz.py tests/test_gpu_wide_nodes.py tests/test_gpu_guard.py -x -q -m gpu --timeout 500 --timeout-method thread -p no:cacheprovider > $OUT/b3_tests.log 2>&1 || { echo "tests failed"; tail -40 $OUT/b3_tests.log; exit 1; }
tail -3 $OUT/b3_tests.log
echo batch3 done
