#!/bin/bash
# Profiles the bench workload on the GPU box: kernel-trace stats of bench.py, then counter passes
# (one rocprofv3 run per counter group) over a few frames.  usage: scripts/profile_round.sh <tag>
set -e
TAG=${1:-now}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/stats.log
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_write.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_inst -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_inst.log 2>&1 || echo 'inst pass failed'
timeout -s KILL 90 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_wait -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_wait.log 2>&1 || echo 'wait pass failed'
cd $REPO
for p in fetch write inst wait; do python3 scripts/pmc_summary.py $OUT/pmc_$p > $OUT/pmc_${p}_summary.txt; done
cat $OUT/bench_profiled.json
cat $OUT/pmc_inst_summary.txt $OUT/pmc_wait_summary.txt
