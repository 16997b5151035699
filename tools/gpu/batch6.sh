#!/bin/bash
# round 3: rocprofv3 profiles of the bench command on the current build (C3, C4), phase counts, the default bench line
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
TAG=${1:-r3}
bash profiles/run_profile.sh ${TAG}_c3 C3 16 || exit 1
bash profiles/run_pmc_extra.sh ${TAG}_c3 C3 16 || exit 1
bash profiles/run_profile.sh ${TAG}_c4 C4 4 || exit 1
bash profiles/run_pmc_extra.sh ${TAG}_c4 C4 4 || exit 1
timeout -k 10 300 python -u bench.py > $OUT/bench_${TAG}_c3.json 2> $OUT/bench_${TAG}_c3.err || { echo "bench failed"; tail $OUT/bench_${TAG}_c3.err; exit 1; }
cut -c1-600 $OUT/bench_${TAG}_c3.json
echo batch6 done
