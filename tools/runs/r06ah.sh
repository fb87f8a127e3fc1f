#!/bin/bash
# round 6, call ah: f16x2 in the full-size tests (configs 1 / 2 on 1 M rows, well-conditioned gradients, bit-exact resume with the scales in the checkpoint)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "f16x2" > $O/r06ah_tests.txt 2>&1
echo "[r06ah] tests rc=$? $(tail -1 $O/r06ah_tests.txt)"; grep -E "^(FAILED|ERROR)|^E  |^worst" $O/r06ah_tests.txt | cut -c1-400 | head -20
