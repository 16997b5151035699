#!/bin/bash
# the two-pixel-frames-per-lane variant (RT_TWO_RAYS, tools/ab/librt_amd_<name>.so) against the shipping library: parity on the 8-ary formats, then timings
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
NAME=${1:-TWO}
export RT_AMD_LIB=$PWD/tools/ab/librt_amd_$NAME.so
for f in 4 3; do
  RT_AMD_NODE_FORMAT=$f timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -p no:cacheprovider -k "not rccl and not bench" > $OUT/two_parity_fmt$f.log 2>&1 || { echo "parity failed on format $f"; tail -30 $OUT/two_parity_fmt$f.log; exit 1; }
  tail -1 $OUT/two_parity_fmt$f.log
  timeout -k 10 400 python -u tests/checkers/fuzz_parity.py --iters 1500 --max-tris 1500 --seed $((41000 + f)) --node-format $f 2>&1 | tail -2 || exit 1
  timeout -k 10 400 python -u tests/checkers/fuzz_session.py --sessions 400 --seed $((41200 + f)) --node-format $f 2>&1 | tail -2 || exit 1
done
unset RT_AMD_LIB
bash tools/gpu/sweep_libs.sh "ship $NAME" "--config C4 --steps 4 --warmup 4" "--config C5q --steps 4 --warmup 4"
echo two_rays done
