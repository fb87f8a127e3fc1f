#!/bin/bash
# round 6, call b: the trainable table on the plane kernels (production width, two ranks), the capture-origin invariant over RCCL,
# the N > 1 line's per-rank facts, the f16x2 probe
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_table.py tests/test_gpu_dist.py -m gpu -q -x -k "table or trainable or hipgraph_over_rccl or two_rank_rehearsal" > $O/r06b_tests.txt 2>&1
echo "[r06b] tests rc=$? $(tail -1 $O/r06b_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06b_tests.txt | head
timeout -k 10 600 python tools/f16x2_probe.py > $O/r06b_f16x2_probe.txt 2> $O/r06b_f16x2_probe.err
echo "[r06b] probe rc=$?"; cat $O/r06b_f16x2_probe.txt; tail -5 $O/r06b_f16x2_probe.err
