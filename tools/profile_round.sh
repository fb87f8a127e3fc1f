#!/bin/bash
# Profile the default bench on the GPU box: rocprofv3 kernel stats, then separate PMC
# passes (FETCH_SIZE, WRITE_SIZE, MFMA busy + clock).  Raw output -> gpurun_out/<tag>/,
# summaries -> gpurun_out/<tag>_*.{csv,json,txt} (copy the ones to keep into profiles/).
# usage (from the repo root, on the GPU box): tools/profile_round.sh TAG [bench.py arguments, e.g. --precision bf16]
set -e -o pipefail
TAG=${1:-prof}
shift || true
ARGS="$@"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/${TAG}_bench.json
echo "[profile] plain bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py $ARGS --steps 60 --warmup 5 --no-settle \
    --no-cpu-baseline --no-extras > $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/${TAG}_kernel_stats.csv
echo "[profile] kernel stats done"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python3 $ROOT/bench.py $ARGS --steps 4 --warmup 0 \
      --no-cpu-baseline --no-extras --no-settle --no-kernel-timers > /dev/null 2> $OUT/pmc_$C.err
  echo "[profile] pmc $C done"
done
# (copy this summary to profiles/latest_pmc.csv -- latest_pmc_bf16.csv for --precision bf16 -- and note
#  its date and commit in profiles/latest_pmc.json: bench.py quotes both in `traffic_source`)
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/${TAG}_pmc_fetch_write.csv \
    FETCH_SIZE=$(find $OUT/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
    WRITE_SIZE=$(find $OUT/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) > /dev/null
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_busy -o pmc -- python3 $ROOT/bench.py $ARGS \
    --steps 4 --warmup 0 --no-cpu-baseline --no-extras --no-settle --no-kernel-timers > /dev/null 2> $OUT/pmc_busy.err
python3 $ROOT/tools/pmc_mfma_busy.py $(find $OUT/pmc_busy -name '*counter_collection.csv' | head -1) \
    > $ROOT/gpurun_out/${TAG}_mfma_busy_clock.txt
echo "[profile] done"
