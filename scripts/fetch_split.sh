set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/fetch_split; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-cur stamps_nocand}; do
  AR_LIB=$REPO/build_variants/$v.so timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$v -- python3 $REPO/scripts/pmc_run.py > $OUT/$v.log 2>&1
  python3 $REPO/scripts/pmc_summary.py $OUT/$v | grep -A1 "^k_tile"
done
