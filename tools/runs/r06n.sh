#!/bin/bash
# round 6, call n: config 4's FC2 as a row-streaming kernel: parity, A/B alone and in the step; kNN in chunks
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -m gpu -q -x -k "fc2_row_streaming" > $O/r06n_tests1.txt 2>&1
echo "[r06n] stream kernel tests rc=$? $(tail -1 $O/r06n_tests1.txt)"; grep -E "^(FAILED|ERROR)" $O/r06n_tests1.txt | head
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_knn.py tests/test_gpu_f32x3.py -m gpu -q > $O/r06n_tests.txt 2>&1
echo "[r06n] tests rc=$? $(tail -1 $O/r06n_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06n_tests.txt | head
for v in 0 1 0 1; do
  CDML_BF16_FC2_STREAM=$v timeout -k 10 200 python tools/x3_gemm_probe.py --cases c4fc2 --rounds 3 2>&1 | grep -v amdgpu.ids | sed "s/^/stream=$v /"
done | tee $O/r06n_fc2_stream_ab.txt
for v in 0 1 0 1; do
  CDML_BF16_FC2_STREAM=$v timeout -k 10 300 python bench.py --precision bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('stream=$v', d['ms_per_step'], json.dumps(d['kernels']))"
done | tee -a $O/r06n_fc2_stream_ab.txt
