#!/bin/bash
# round 6, call am: build_graph's switches (clip, regulariser, variance, momentum) on both plane forms
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "build_graph_switches" > $O/r06am_tests.txt 2>&1
echo "[r06am] tests rc=$? $(tail -1 $O/r06am_tests.txt)"; (grep -E "^(FAILED|ERROR)|^E  " $O/r06am_tests.txt | cut -c1-500 | head -12) || true
