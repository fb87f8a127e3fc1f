#!/bin/bash
# round 6, call ap: the whole GPU suite once more on another box (flakiness screen), smoke
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r06ap_gpu_tests.txt 2>&1
echo "[r06ap] gpu suite rc=$? $(tail -1 $O/r06ap_gpu_tests.txt)"; (grep -E "^(FAILED|ERROR)" $O/r06ap_gpu_tests.txt | head) || true
timeout -k 10 300 python __graft_entry__.py --smoke > $O/r06ap_smoke.txt 2>&1
echo "[r06ap] smoke rc=$?"; grep smoke $O/r06ap_smoke.txt
