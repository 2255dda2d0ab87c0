#!/bin/bash
# Everything the bench line's counter figures come from, in one GPU call: kernel-trace stats of bench.py, the counter
# passes (each a rocprofv3 run of its own), the L2 and busy-cycle passes, then profiles/traffic.json (stamped with the
# fingerprint of the kernel sources) and the per-round copies under profiles/.  usage: scripts/profile_all.sh <tag>
TAG=${1:-now}
# the counter passes render chains of so many frames (what bench.py's timed region launches), so many chains each
export AR_CHAIN=${AR_CHAIN:-16} AR_FRAMES=${AR_FRAMES:-3}
REPO=$(pwd)
bash scripts/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1 || { tail -5 gpurun_out/profile_round_$TAG.log; exit 1; }
bash scripts/pmc_l2.sh $TAG > gpurun_out/pmc_l2_$TAG.log 2>&1 || { tail -5 gpurun_out/pmc_l2_$TAG.log; exit 1; }
bash scripts/pmc_busy.sh $TAG > gpurun_out/pmc_busy_$TAG.log 2>&1 || { tail -5 gpurun_out/pmc_busy_$TAG.log; exit 1; }
P=gpurun_out/prof_$TAG
python3 scripts/traffic_json.py $P gpurun_out/traffic_$TAG.json gpurun_out/l2_$TAG gpurun_out/busy_$TAG || exit 1
mkdir -p gpurun_out/profiles_$TAG
cp gpurun_out/traffic_$TAG.json gpurun_out/profiles_$TAG/traffic.json
cp $P/bench_profiled.json gpurun_out/profiles_$TAG/${TAG}_bench_profiled.json
for p in fetch write inst wait; do cp $P/pmc_${p}_summary.txt gpurun_out/profiles_$TAG/${TAG}_pmc_$p.txt; done
cat gpurun_out/l2_$TAG/pmc_hit_summary.txt gpurun_out/l2_$TAG/pmc_ea_summary.txt > gpurun_out/profiles_$TAG/${TAG}_pmc_l2.txt
cp gpurun_out/busy_$TAG/pmc_busy_summary.txt gpurun_out/profiles_$TAG/${TAG}_pmc_busy.txt
find $P/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/profiles_$TAG/${TAG}_kernel_stats.csv
find $P/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 scripts/trace_passes.py {} > gpurun_out/profiles_$TAG/${TAG}_kernel_trace_passes.txt 2>/dev/null
ls gpurun_out/profiles_$TAG
