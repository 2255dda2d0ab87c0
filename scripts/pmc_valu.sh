#!/bin/bash
# Vector / scalar / LDS instruction counts per kernel for library variants: scripts/pmc_valu.sh <lib.so>...
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export AR_LIB=$REPO/$lib
  name=$(basename $lib .so)
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $REPO/gpurun_out/pmc_v_$name -- python3 $REPO/scripts/pmc_run.py > /dev/null 2>&1 || exit 1
  echo "== $name"
  python3 $REPO/scripts/pmc_summary.py $REPO/gpurun_out/pmc_v_$name | grep -A6 -E "^k_tile" | grep -E "VALU|SALU|LDS|VMEM"
done
