#!/bin/bash
# round 3, fourth GPU batch: per-axis slab pads on every node format — C3 / C2 A/B, shard scaling, C5, the whole GPU suite
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
PREV=$PWD/tools/ab/librt_amd_prev.so
run() { # name, lib ("" = shipping), bench args...
  local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export RT_AMD_LIB=$lib; else unset RT_AMD_LIB; fi
  timeout -k 10 400 python -u bench.py --no-cpu "$@" > $OUT/b4_$name.json 2> $OUT/b4_$name.err || { echo "bench $name failed"; tail -5 $OUT/b4_$name.err; exit 1; }
  python - $OUT/b4_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], "ms_per_step %.2f"%d["ms_per_step"], "Mrays/s %.0f"%d["value"], "rays", d["config"]["rays"], flush=True)
PY
}
run c3_prev $PREV --config C3 --steps 16 --warmup 16
run c3_new  ""    --config C3 --steps 16 --warmup 16
run c3_prev2 $PREV --config C3 --steps 16 --warmup 16
run c3_new2  ""    --config C3 --steps 16 --warmup 16
run c2_prev $PREV --config C2 --steps 4 --warmup 4
run c2_new  ""    --config C2 --steps 4 --warmup 4
export RT_AMD_LIB=$PREV
timeout -k 10 200 python -u tools/shard_scaling.py --config C3 > $OUT/b4_shard_prev.log 2>&1 || { echo "shard prev failed"; tail $OUT/b4_shard_prev.log; exit 1; }
unset RT_AMD_LIB
timeout -k 10 200 python -u tools/shard_scaling.py --config C3 > $OUT/b4_shard_new.log 2>&1 || { echo "shard new failed"; tail $OUT/b4_shard_new.log; exit 1; }
echo "shard prev:"; tail -5 $OUT/b4_shard_prev.log; echo "shard new:"; tail -5 $OUT/b4_shard_new.log
run c5s2_new "" --config C5 --steps 2 --warmup 0
run c5s6_new "" --config C5 --steps 6 --warmup 0
echo batch4 done
