#!/bin/bash
# round 6, call z: placements of the R3 period's loads (CDML_R3_SCHED=0 / 1; k-contiguous products), alternating processes
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -k "resident or ragged or five" > $O/r06z_tests0.txt 2>&1
echo "[r06z] tests sched0 rc=$? $(tail -1 $O/r06z_tests0.txt)"
CDML_R3_SCHED=1 timeout -k 10 300 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -k "resident or ragged or five" > $O/r06z_tests1.txt 2>&1
echo "[r06z] tests sched1 rc=$? $(tail -1 $O/r06z_tests1.txt)"
for rnd in 1 2; do for sc in 0 1; do echo "## CDML_R3_SCHED=$sc round $rnd"; CDML_R3_SCHED=$sc timeout -k 10 300 python tools/f16x2_rate.py | grep -E "^(FC1|FC2|dH1|sum)" | sed 's/.*|//'; done; done > $O/r06z_rate.txt 2>&1
echo "[r06z] rate rc=$?"; cat $O/r06z_rate.txt
