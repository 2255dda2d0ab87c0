#!/bin/bash
# L2 view of the frame's kernels: hits / misses and the read / write requests that leave the L2 (two counter passes).
# usage: scripts/pmc_l2.sh <tag>
TAG=${1:-now}
REPO=$(pwd)
OUT=$REPO/gpurun_out/l2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/pmc_hit -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_hit.log 2>&1 || echo 'hit pass failed'
timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/pmc_ea -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_ea.log 2>&1 || echo 'ea pass failed'
cd $REPO
for p in hit ea; do python3 scripts/pmc_summary.py $OUT/pmc_$p > $OUT/pmc_${p}_summary.txt; done
cat $OUT/pmc_hit_summary.txt $OUT/pmc_ea_summary.txt
