#!/bin/bash
# FETCH_SIZE and WRITE_SIZE (separate passes: TCC has 4 slots, FETCH_SIZE takes 3) of one python command, per kernel.
# usage (GPU box, repo root): tools/pmc_fetch_write.sh TAG script.py [args]  ->  gpurun_out/TAG_pmc_fetch_write.csv
set -e -o pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $ROOT/"$@" > $OUT/pmc_$C.out 2> $OUT/pmc_$C.err || { echo "[pmc] $C failed"; tail -3 $OUT/pmc_$C.err; }
  echo "[pmc] $C done"
done
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}_pmc_fetch_write.csv \
    FETCH_SIZE=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
    WRITE_SIZE=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
