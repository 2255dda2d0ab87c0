#!/bin/bash
# Busy-cycle counters of the frame's kernels, with the kernel trace of the same run (durations -> engine clock under load):
# scripts/pmc_busy.sh <tag> [lib.so]
TAG=${1:-now}
REPO=$(pwd)
[ -n "$2" ] && export AR_LIB=$REPO/$2
OUT=$REPO/gpurun_out/busy_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_busy -- python3 $REPO/scripts/pmc_run.py > $OUT/run.log 2>&1 || { echo 'busy pass failed'; tail -5 $OUT/run.log; exit 1; }
cd $REPO
python3 scripts/pmc_summary.py $OUT/pmc_busy > $OUT/pmc_busy_summary.txt
grep -A7 -E "^k_(tile|setup|fill|ascii|binscan)" $OUT/pmc_busy_summary.txt
ls $OUT/pmc_busy/*/ | head
