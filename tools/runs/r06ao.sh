#!/bin/bash
# round 6, call ao: f16x2 inference bits independent of the chunk
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -k "chunk or catalogue" > $O/r06ao_tests.txt 2>&1
echo "[r06ao] tests rc=$? $(tail -1 $O/r06ao_tests.txt)"; (grep -E "^(FAILED|ERROR)|^E  " $O/r06ao_tests.txt | cut -c1-500 | head -14) || true
