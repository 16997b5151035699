#!/bin/bash
# round 3, second GPU batch: ray-order experiment v2 (no shared atomics) with its PMC passes, counter calibration
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/calib
step() { echo "=== $*"; }
step "ray-order experiment C4 (sample 1, bounce 2)"
timeout -k 10 400 python -u tools/exp_ray_order/run.py --config C4 --frames 4 --sample 1 --bounce 2 --out $OUT/exp_ray_order_c4_s1b2.json > $OUT/exp_c4_s1b2.log 2>&1 || { echo "experiment failed"; tail -30 $OUT/exp_c4_s1b2.log; exit 1; }
tail -15 $OUT/exp_c4_s1b2.log
step "ray-order experiment C4 (sample 2, bounce 5)"
timeout -k 10 400 python -u tools/exp_ray_order/run.py --config C4 --frames 4 --sample 2 --bounce 5 --out $OUT/exp_ray_order_c4_s2b5.json > $OUT/exp_c4_s2b5.log 2>&1 || { echo "experiment failed"; tail -30 $OUT/exp_c4_s2b5.log; exit 1; }
tail -15 $OUT/exp_c4_s2b5.log
step "ray-order experiment, pmc passes (phased kernel: arrival, then sorted)"
for pass in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/exp_pmc/pmc_$name -- python3 tools/exp_ray_order/run.py --config C4 --frames 4 --sample 1 --bounce 2 --pmc > $OUT/exp_pmc_$name.log 2>&1 || { echo "exp pmc pass failed"; tail -5 $OUT/exp_pmc_$name.log; exit 1; }
done
step "calibration, plain"
timeout -k 10 120 ./tools/microbench/calibrate.bin > $OUT/calib/plain.txt 2>&1 || { echo "calibrate failed"; tail -5 $OUT/calib/plain.txt; exit 1; }
grep "rep 1" $OUT/calib/plain.txt | cut -c1-200
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
            "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/calib/pmc_$name
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/calib/pmc_$name -- ./tools/microbench/calibrate.bin > $OUT/calib/pmc_$name.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/calib/pmc_$name.log; exit 1; }
done
python tools/calibration_summary.py $OUT/calib > $OUT/r3_calibration.txt 2>&1 || { echo "summary failed"; tail -5 $OUT/r3_calibration.txt; }
echo batch2 done
