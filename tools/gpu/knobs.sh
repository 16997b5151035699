#!/bin/bash
# developer-knob sweep of one config: bash tools/gpu/knobs.sh "<bench args>" "<k=v,...>" ["<k=v,...>" ...]   ("-" = no option)
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
ARGS=$1; shift
for round in 1 2; do
  for o in "$@"; do
    if [ "$o" = "-" ]; then unset RT_BENCH_OPTIONS; else export RT_BENCH_OPTIONS=$o; fi
    timeout -k 10 400 python -u bench.py --no-cpu $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$o', 'round $round', '%.2f ms/step' % d['ms_per_step'], 'nodes', d['config'].get('bvh_nodes'), 'build %.1f s' % d['config'].get('bvh_build_s', 0), flush=True)" || { echo "$o failed"; exit 1; }
  done
done
