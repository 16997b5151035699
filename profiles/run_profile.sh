#!/bin/bash
# Profile recipe for the headline bench (run on the GPU box through gpurun).
# Usage: bash profiles/run_profile.sh <tag> [config] [steps]    → gpurun_out/prof_<tag>/
set -u
TAG=${1:-r1}
CONFIG=${2:-C3}
STEPS=${3:-8}
OPTS=${4:-}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# warm-up = steps: rt_dispatch_frames folds a run of frames into ONE launch, so both launches are the same size
# and rocprofv3's per-kernel average is the duration of a $STEPS-frame launch
[ -n "$OPTS" ] && export RT_BENCH_OPTIONS=$OPTS
CMD="python3 bench.py --config $CONFIG --steps $STEPS --warmup $STEPS --no-cpu"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
            "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $pass failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
echo done
