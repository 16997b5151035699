#!/bin/bash
# Extra PMC passes: texture-addresser / L1 (TCP) / data-return busy counters.
# Usage: bash profiles/run_pmc_extra.sh <tag> [config] [steps]
set -u
TAG=${1:-x}; CONFIG=${2:-C3}; STEPS=${3:-8}
OPTS=${4:-}
OUT=$PWD/gpurun_out/prof_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
[ -n "$OPTS" ] && export RT_BENCH_OPTIONS=$OPTS
CMD="python3 bench.py --config $CONFIG --steps $STEPS --warmup $STEPS --no-cpu"
for pass in "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" \
            "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
            "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 240 rocprofv3 --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- $CMD > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass $pass failed"; tail -5 "$OUT/pmc_$name.log"; exit 1; }
done
echo done
