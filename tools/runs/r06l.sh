#!/bin/bash
# round 6, call l: the operand-swapped LDS-free plane epilogues (FC1, data gradient): parity, then A/B by CDML_X3_SWAP on one box
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/r06l_tests.txt 2>&1
echo "[r06l] tests rc=$? $(tail -1 $O/r06l_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06l_tests.txt | head
for sw in 0 1 0 1; do
  CDML_X3_SWAP=$sw timeout -k 10 200 python tools/x3_gemm_probe.py --cases fc1m,dh1m --rows 16384 --rounds 3 2>&1 | grep -v amdgpu.ids | sed "s/^/swap=$sw rows=16384 /"
done | tee $O/r06l_swap_ab.txt
for sw in 0 1; do
  CDML_X3_SWAP=$sw timeout -k 10 200 python tools/x3_gemm_probe.py --cases fc1m,dh1m --rows 8192 --rounds 3 2>&1 | grep -v amdgpu.ids | sed "s/^/swap=$sw rows=8192 /"
done | tee -a $O/r06l_swap_ab.txt
for sw in 0 1 0 1; do
  CDML_X3_SWAP=$sw timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('swap=$sw', d['ms_per_step'], json.dumps(d['kernels']))"
done | tee -a $O/r06l_swap_ab.txt
