#!/bin/bash
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/r05l_gpu_tests.txt 2>&1
echo "[r05l] gpu suite rc=$? $(tail -1 $O/r05l_gpu_tests.txt)"; grep "^FAILED" $O/r05l_gpu_tests.txt | head
