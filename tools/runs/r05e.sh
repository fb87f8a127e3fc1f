#!/bin/bash
# round 5, GPU call E: the tests that failed in call D, persistent-tile A/B (CDML_X3_PERSIST) on the miner and the headline
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q > $O/r05e_gpu_tests.txt 2>&1
echo "[r05e] tests rc=$? $(tail -1 $O/r05e_gpu_tests.txt)"; grep "^FAILED" $O/r05e_gpu_tests.txt | head
for P in 0 1 0 1; do CDML_X3_PERSIST=$P python tools/mine_probe.py 2>&1 | sed "s/^/persist=$P /" | tee -a $O/r05e_persist.txt; done
for i in 1 2; do for P in 0 1; do
  CDML_X3_PERSIST=$P python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $O/r05e_persist_${P}_$i.json 2>> $O/r05e.err
  python -c "
import json; d=json.load(open('$O/r05e_persist_${P}_$i.json')); print('headline persist=$P run $i', d['ms_per_step'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')}, d['loss'])" | tee -a $O/r05e_persist.txt
done; done
for P in 0 1; do
  CDML_X3_PERSIST=$P python bench.py --rows 1000000 --batch 4096 --steps 200 --warmup 10 --no-extras --no-cpu-baseline > $O/r05e_persist_c1_${P}.json 2>> $O/r05e.err
  python -c "
import json; d=json.load(open('$O/r05e_persist_c1_${P}.json')); print('config1 persist=$P', d['ms_per_step'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')}, d['loss'])" | tee -a $O/r05e_persist.txt
done
CDML_X3_PERSIST=1 python -m pytest tests/test_gpu_f32x3.py -m gpu -q -k "half_tiles or production_shapes or race_screen or plane_outputs" > $O/r05e_persist_tests.txt 2>&1
echo "[r05e] persist tests rc=$? $(tail -1 $O/r05e_persist_tests.txt)"
