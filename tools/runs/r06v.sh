#!/bin/bash
# round 6, call v: precision f16x2 as a training step -- plane writers, the step against the oracle, a run across the scale checks
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s > $O/r06v_tests.txt 2>&1
echo "[r06v] tests rc=$? $(tail -1 $O/r06v_tests.txt)"; grep -E "^(FAILED|ERROR)|^\{|^f16x2 against" $O/r06v_tests.txt | head
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "train_steps_config0" > $O/r06v_parity.txt 2>&1
echo "[r06v] parity rc=$? $(tail -1 $O/r06v_parity.txt)"; grep -E "^(FAILED|ERROR)" $O/r06v_parity.txt | head
