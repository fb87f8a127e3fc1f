#!/bin/bash
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q > $O/r05p_gpu_tests.txt 2>&1
echo "[r05p] gpu suite rc=$? $(tail -1 $O/r05p_gpu_tests.txt)"; grep "^FAILED" $O/r05p_gpu_tests.txt | head
bash tools/runs/r05_profile.sh $1
