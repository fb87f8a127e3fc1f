#!/bin/bash
# round 6, first call: the GPU suite and the default bench line on HEAD as the round starts (this box's baseline)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -m gpu -q -x > $O/r06a_gpu_tests.txt 2>&1
echo "[r06a] gpu suite rc=$? $(tail -1 $O/r06a_gpu_tests.txt)"
timeout -k 10 300 python bench.py > $O/r06a_bench.json 2> $O/r06a_bench.err
echo "[r06a] bench rc=$?"; cut -c1-600 $O/r06a_bench.json
