#!/bin/bash
# round 3, fifth GPU batch: resident workgroups per CU and scheduler weights re-swept now that the slice rays are gone
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
run() { # name, options, bench args...
  local name=$1 opts=$2; shift 2
  export RT_BENCH_OPTIONS=$opts
  timeout -k 10 400 python -u bench.py --no-cpu "$@" > $OUT/b5_$name.json 2> $OUT/b5_$name.err || { echo "bench $name failed"; tail -5 $OUT/b5_$name.err; exit 1; }
  python - $OUT/b5_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], "ms_per_step %.2f"%d["ms_per_step"], "Mrays/s %.0f"%d["value"], "waves/simd", d["config"]["resident_waves_per_simd"], flush=True)
PY
}
for wg in 4 5 6; do run c4_wg$wg "103=$wg" --config C4 --steps 4 --warmup 4; done
for wg in 3 4 5 6; do run c5q_wg$wg "103=$wg" --config C5q --steps 4 --warmup 4; done
for t in "100=3,101=4,102=2" "100=2,101=4,102=4" "100=2,101=4,102=6" "100=2,101=4,102=8" "100=3,101=4,102=6"; do
  n=$(echo $t | tr -d '=,' ); run c4_t$n "$t" --config C4 --steps 4 --warmup 4; done
for wg in 3 4 5; do run c5_wg$wg "103=$wg" --config C5 --steps 4 --warmup 0; done
echo batch5 done
