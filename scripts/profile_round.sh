#!/bin/bash
# Profiles the bench workload on the GPU box: kernel-trace stats of bench.py, then HBM traffic
# counters (one rocprofv3 pass per counter) of a few frames.  usage: scripts/profile_round.sh <tag>
set -e
TAG=${1:-now}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --no-cpu-baseline > $OUT/bench_profiled.json 2> $OUT/stats.log
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/scripts/pmc_run.py > $OUT/pmc_write.log 2>&1
cd $REPO
python3 scripts/pmc_summary.py $OUT/pmc_fetch > $OUT/pmc_fetch_summary.txt
python3 scripts/pmc_summary.py $OUT/pmc_write > $OUT/pmc_write_summary.txt
cat $OUT/bench_profiled.json
cat $OUT/pmc_fetch_summary.txt $OUT/pmc_write_summary.txt
