#!/bin/bash
# A/B of tools/ab/librt_amd_prev.so against the shipping library: bash tools/gpu/ab.sh "<bench args>" ["<bench args>" ...]; two rounds each, interleaved
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
PREV=$PWD/tools/ab/librt_amd_prev.so
run() { local name=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export RT_AMD_LIB=$lib; else unset RT_AMD_LIB; fi
  timeout -k 10 400 python -u bench.py --no-cpu "$@" > $OUT/ab_$name.json 2> $OUT/ab_$name.err || { echo "bench $name failed"; tail -5 $OUT/ab_$name.err; exit 1; }
  python - $OUT/ab_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], "ms_per_step %.2f"%d["ms_per_step"], "Mrays/s %.0f"%d["value"], "rays", d["config"]["rays"], flush=True)
PY
}
k=0
for args in "$@"; do
  k=$((k+1))
  for round in 1 2; do
    run case${k}_prev_$round $PREV $args
    run case${k}_new_$round "" $args
  done
done
