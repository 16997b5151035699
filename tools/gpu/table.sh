#!/bin/bash
# the secondary configs on the build in the tree (one line each), then the one-GPU shard rehearsal
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
run() { local name=$1; shift
  timeout -k 10 500 python -u bench.py --no-cpu "$@" > $OUT/table_$name.json 2> $OUT/table_$name.err || { echo "bench $name failed"; tail -5 $OUT/table_$name.err; exit 1; }
  python - $OUT/table_$name.json $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).readline())
print(sys.argv[2], "ms_per_step %.2f"%d["ms_per_step"], "Mrays/s %.0f"%d["value"], "waves/simd", d["config"].get("resident_waves_per_simd"), flush=True)
PY
}
run c2 --config C2 --steps 4 --warmup 4
run c4 --config C4 --steps 4 --warmup 4
run c5q --config C5q --steps 4 --warmup 4
run c5 --config C5 --steps 4 --warmup 0
timeout -k 10 400 python -u tools/shard_scaling.py > $OUT/table_shards.log 2>&1 || { echo "shard rehearsal failed"; tail -5 $OUT/table_shards.log; exit 1; }
tail -6 $OUT/table_shards.log
echo table done
