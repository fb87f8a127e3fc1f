#!/bin/bash
# round 6, call u: the two-plane fp16 GEMMs -- error gate at the production shapes, ragged shapes, rate beside the six-plane kernels
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s > $O/r06u_tests.txt 2>&1
echo "[r06u] tests rc=$? $(tail -1 $O/r06u_tests.txt)"; grep -E "^(FAILED|ERROR)|^\{" $O/r06u_tests.txt | head
timeout -k 10 300 python -m pytest tests/test_gpu_f32x3.py -m gpu -q -x > $O/r06u_x3.txt 2>&1
echo "[r06u] x3 rc=$? $(tail -1 $O/r06u_x3.txt)"
timeout -k 10 300 python tools/f16x2_rate.py > $O/r06u_rate.txt 2>&1
echo "[r06u] rate rc=$?"; cat $O/r06u_rate.txt | tail -8
