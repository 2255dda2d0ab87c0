#!/bin/bash
# k_tile under three counter passes (issue / LDS, L2, L1) for library variants: scripts/pmc_ab.sh <lib.so>...
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export AR_LIB=$REPO/$lib
  name=$(basename $lib .so)
  echo "== $name"
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU --output-format csv -d $REPO/gpurun_out/pmc_ab_${name}_sq -- python3 $REPO/scripts/pmc_run.py > /dev/null 2>&1 || exit 1
  python3 $REPO/scripts/pmc_summary.py $REPO/gpurun_out/pmc_ab_${name}_sq | grep -A9 -E "^k_tile" | grep -v "^k_"
  timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $REPO/gpurun_out/pmc_ab_${name}_l2 -- python3 $REPO/scripts/pmc_run.py > /dev/null 2>&1 || exit 1
  python3 $REPO/scripts/pmc_summary.py $REPO/gpurun_out/pmc_ab_${name}_l2 | grep -A5 -E "^k_tile" | grep -v "^k_"
  timeout -s KILL 90 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --output-format csv -d $REPO/gpurun_out/pmc_ab_${name}_l1 -- python3 $REPO/scripts/pmc_run.py > /dev/null 2>&1 || exit 1
  python3 $REPO/scripts/pmc_summary.py $REPO/gpurun_out/pmc_ab_${name}_l1 | grep -A4 -E "^k_tile" | grep -v "^k_"
done
