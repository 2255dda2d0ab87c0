#!/bin/bash
# Instruction counters of every kernel of a few C3 frames: scripts/pmc_frame.sh <tag> [lib.so]
TAG=${1:-now}
REPO=$(pwd)
[ -n "$2" ] && export AR_LIB=$REPO/$2
OUT=$REPO/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -- python3 $REPO/scripts/pmc_run.py > $OUT/run.log 2>&1 || exit 1
cd $REPO
python3 scripts/pmc_summary.py $OUT/pmc | grep -A9 "^k_" | grep -E "^k_|VALU|SALU|LDS|WAVE_CYCLES"
