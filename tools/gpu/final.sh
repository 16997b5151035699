#!/bin/bash
# round end: fuzz campaign, the whole GPU suite, smoke and the default bench line on the build in the tree
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
SEED0=${SEED0:-12000} ITERS=${ITERS:-3000} BIG=${BIG:-80} SESS=${SESS:-1000} bash tools/gpu/fuzz_campaign.sh final > $OUT/final_fuzz.log 2>&1 || { echo "fuzz campaign failed"; tail -20 $OUT/final_fuzz.log; exit 1; }
tail -4 $OUT/final_fuzz.log
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 600 --timeout-method thread -p no:cacheprovider > $OUT/final_suite.log 2>&1 || { echo "suite failed"; tail -30 $OUT/final_suite.log; exit 1; }
tail -2 $OUT/final_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 300 python -u bench.py > $OUT/bench_final.json 2> $OUT/bench_final.err || { echo "bench failed"; tail $OUT/bench_final.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_final.json").readline()); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], "stale:", r["provenance"]["stale"], "onchip:", r["onchip"]["frac"], r["onchip"]["stale"], "parity:", d["parity"]["identical"])
PY
echo final done
