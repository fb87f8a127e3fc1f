#!/bin/bash
# round 6, call y: f16x2 -- all its tests on the R3 walk, smoke, then the profile set of the f16x2 step (lean job)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s > $O/r06y_tests.txt 2>&1
echo "[r06y] tests rc=$? $(tail -1 $O/r06y_tests.txt)"; grep -E "^(FAILED|ERROR)|^gradient error" $O/r06y_tests.txt | cut -c1-2500 | head
timeout -k 10 300 python __graft_entry__.py --smoke > $O/r06y_smoke.txt 2>&1
echo "[r06y] smoke rc=$?"; grep smoke $O/r06y_smoke.txt
export CDML_F16X2_LEAN=1
timeout -k 10 900 bash tools/profile_round.sh r06_f16x2 --only f16x2 --steps 100 --warmup 10 > $O/r06y_profile.txt 2>&1
echo "[r06y] profile rc=$?"; tail -3 $O/r06y_profile.txt; head -12 $O/r06_f16x2_kernel_stats.csv | cut -c1-200; cat $O/r06_f16x2_mfma_busy_clock.txt | head -20
