#!/bin/bash
# round 3, first GPU batch: changed C-ABI paths, counter calibration, ray-order experiment
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT/calib
step() { echo "=== $*"; }
step "no-alloc / group tests"
timeout -k 10 420 python -u -m pytest tests/test_zz_no_alloc.py tests/test_gpu_group.py -x -q -m gpu --timeout 180 --timeout-method thread -p no:cacheprovider > $OUT/b1_tests.log 2>&1 || { echo "tests failed"; tail -40 $OUT/b1_tests.log; exit 1; }
tail -3 $OUT/b1_tests.log
step "calibration, plain"
timeout -k 10 120 ./tools/microbench/calibrate.bin > $OUT/calib/plain.txt 2>&1 || { echo "calibrate failed"; tail -5 $OUT/calib/plain.txt; exit 1; }
cat $OUT/calib/plain.txt
for pass in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" \
            "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  step "calibration pmc $name"
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $OUT/calib/pmc_$name -- ./tools/microbench/calibrate.bin > $OUT/calib/pmc_$name.log 2>&1 || { echo "pmc pass failed"; tail -5 $OUT/calib/pmc_$name.log; exit 1; }
done
python tools/calibration_summary.py $OUT/calib > $OUT/r3_calibration.txt 2>&1 || { echo "summary failed"; tail -5 $OUT/r3_calibration.txt; }
step "ray-order experiment C4"
timeout -k 10 560 python -u tools/exp_ray_order/run.py --config C4 --frames 4 --sample 1 --bounce 2 --out $OUT/exp_ray_order_c4_s1b2.json > $OUT/exp_c4_s1b2.log 2>&1 || { echo "experiment failed"; tail -30 $OUT/exp_c4_s1b2.log; exit 1; }
tail -16 $OUT/exp_c4_s1b2.log
echo batch1 done
