#!/bin/bash
# round 5, GPU call H: the k8-interleaved weight-gradient kernel: parity with the k-strided form, timing at both batch sizes
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_f32x3.py -m gpu -q -k "tnk or tn_weight" > $O/r05h_tests.txt 2>&1
echo "[r05h] tests rc=$? $(tail -1 $O/r05h_tests.txt)"; grep "^FAILED\|^E  " $O/r05h_tests.txt | head
timeout -k 10 300 python tools/x3_tnk_probe.py 16384 2>&1 | grep "R=" | tee $O/r05h_tnk_probe.txt
timeout -k 10 300 python tools/x3_tnk_probe.py 8192 2>&1 | grep "R=" | tee -a $O/r05h_tnk_probe.txt
