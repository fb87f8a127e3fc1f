#!/bin/bash
# round 6, call q: the miner's prep launch normalising z itself (one launch fewer in config 2's step): parity, then A/B in the step
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_f32x3.py -m gpu -q > $O/r06q_tests.txt 2>&1
echo "[r06q] tests rc=$? $(tail -1 $O/r06q_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06q_tests.txt | head
for v in 0 1 0 1; do
  CDML_MINE_NORM=$v timeout -k 10 300 python bench.py --mode semihard --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mine_norm=$v', d['ms_per_step'], d['loss'], json.dumps(d['kernels']))"
done | tee $O/r06q_mine_norm_ab.txt
