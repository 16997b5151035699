#!/bin/bash
# fuzz campaign on the current build: random scenes against brute force on every node format, big scenes against the oracle's BVH mode, random sessions
set -u
OUT=$PWD/gpurun_out; mkdir -p $OUT
TAG=${1:-r3}
for f in 0 1 2 3 4; do
  timeout -k 10 400 python -u tests/checkers/fuzz_parity.py --iters ${ITERS:-2000} --max-tris 1500 --seed $((${SEED0:-9000} + f)) --node-format $f > $OUT/fuzz_${TAG}_fmt$f.log 2>&1 || { echo "fuzz fmt $f FAILED"; tail -5 $OUT/fuzz_${TAG}_fmt$f.log; exit 1; }
  tail -1 $OUT/fuzz_${TAG}_fmt$f.log
done
for f in 0 4; do
  timeout -k 10 400 python -u tests/checkers/fuzz_parity.py --big --iters ${BIG:-60} --max-tris 60000 --seed $((${SEED0:-9000} + 100 + f)) --node-format $f > $OUT/fuzzbig_${TAG}_fmt$f.log 2>&1 || { echo "fuzz big fmt $f FAILED"; tail -5 $OUT/fuzzbig_${TAG}_fmt$f.log; exit 1; }
  tail -1 $OUT/fuzzbig_${TAG}_fmt$f.log
done
for f in 0 3 4; do
  timeout -k 10 400 python -u tests/checkers/fuzz_session.py --sessions ${SESS:-600} --seed $((${SEED0:-9000} + 200 + f)) --node-format $f > $OUT/sess_${TAG}_fmt$f.log 2>&1 || { echo "sessions fmt $f FAILED"; tail -5 $OUT/sess_${TAG}_fmt$f.log; exit 1; }
  tail -1 $OUT/sess_${TAG}_fmt$f.log
done
echo campaign done
