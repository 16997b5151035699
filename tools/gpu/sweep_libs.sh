#!/bin/bash
# bench every tools/ab/librt_amd_<name>.so named on the command line (and the shipping library, "ship") on the given configs, two rounds, interleaved
# usage: bash tools/gpu/sweep_libs.sh "ship NR2 NR4" "--config C4 --steps 4 --warmup 4" ["<more bench args>" ...]
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
LIBS=$1; shift
for args in "$@"; do
  echo "== $args"
  for round in 1 2; do
    for l in $LIBS; do
      if [ "$l" = ship ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/tools/ab/librt_amd_$l.so; fi
      timeout -k 10 400 python -u bench.py --no-cpu $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$l', 'round $round', '%.2f ms/step' % d['ms_per_step'], flush=True)" || { echo "$l failed"; exit 1; }
    done
  done
done
