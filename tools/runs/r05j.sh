#!/bin/bash
# round 5, GPU call J: the k8-interleaved weight-gradient operands end to end: whole -m gpu suite, headline / config 1 A/B
# (CDML_X3_KI=0|1), gather with the second copy
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/r05j_gpu_tests.txt 2>&1
echo "[r05j] gpu suite rc=$? $(tail -1 $O/r05j_gpu_tests.txt)"; grep "^FAILED" $O/r05j_gpu_tests.txt | head
for i in 1 2; do for K in 0 1; do
  CDML_X3_KI=$K python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $O/r05j_ki_${K}_$i.json 2>> $O/r05j.err
  python -c "
import json; d=json.load(open('$O/r05j_ki_${K}_$i.json')); print('headline ki=$K run $i', d['ms_per_step'], d['value'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')}, 'gather', d['gather']['frac'], d['gather']['launch_ms'], d['gather']['steps_per_launch'], d['loss'])" | tee -a $O/r05j_ki.txt
done; done
for K in 0 1; do
  CDML_X3_KI=$K python bench.py --rows 1000000 --batch 4096 --steps 200 --warmup 10 --no-extras --no-cpu-baseline > $O/r05j_ki_c1_${K}.json 2>> $O/r05j.err
  python -c "
import json; d=json.load(open('$O/r05j_ki_c1_${K}.json')); print('config1 ki=$K', d['ms_per_step'], d['value'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')}, 'gather', d['gather']['frac'], d['gather']['launch_ms'], d['gather']['steps_per_launch'], d['loss'])" | tee -a $O/r05j_ki.txt
done
