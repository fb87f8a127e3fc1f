#!/bin/bash
# Separate `rocprofv3 --pmc` passes (8 SQ slots each) of one python command, then one table per kernel.
# The program comes directly after `--` (no env / bash -c hop: the profiler initialises the GPU first).
# usage (GPU box, repo root): tools/pmc_passes.sh TAG script.py [script arguments]
# output: gpurun_out/TAG_counters.csv (+ raw passes under gpurun_out/TAG/)
set -e -o pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
 "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VALU SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES"
)
i=0
FILES=""
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- python3 $ROOT/"$@" > $OUT/pass$i.out 2> $OUT/pass$i.err || { echo "[pmc] pass $i failed"; tail -5 $OUT/pass$i.err; }
  F=$(find $OUT/pass$i -name '*counter_collection.csv' | head -1)
  [ -n "$F" ] && FILES="$FILES $F"
  echo "[pmc] pass $i done: $P"
  i=$((i+1))
done
python3 $ROOT/tools/pmc_table.py $FILES > $ROOT/gpurun_out/${TAG}_counters.csv
cat $ROOT/gpurun_out/${TAG}_counters.csv
