#!/bin/bash
# round 5, GPU call I: the data gradient's interleaved plane output against the interleaved row-major output
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_gpu_f32x3.py -m gpu -q -k "interleaved or tnk or plane_outputs or half_tiles or bitmask" > $O/r05i_tests.txt 2>&1
echo "[r05i] tests rc=$? $(tail -1 $O/r05i_tests.txt)"; grep "^FAILED\|^E  " $O/r05i_tests.txt | head
