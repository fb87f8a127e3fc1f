#!/bin/bash
# round 5, GPU call D: whole -m gpu suite; miner ablations; FC2 slab length A/B (60 vs 120 steps) on the headline and config 1
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/r05d_gpu_tests.txt 2>&1
echo "[r05d] gpu suite rc=$? $(tail -1 $O/r05d_gpu_tests.txt)"; grep "^FAILED" $O/r05d_gpu_tests.txt | head
python tools/mine_probe.py > $O/r05d_mine_probe.txt 2>&1
for T in mine_noepi mine_noloop; do CDML_LIB_PATH=build/variants/libcdml_$T.so python tools/mine_probe.py >> $O/r05d_mine_probe.txt 2>&1; done
grep "B=" $O/r05d_mine_probe.txt
for i in 1 2; do for S in 60 120; do
  CDML_X3_SLAB_STEPS=$S python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $O/r05d_slab_${S}_$i.json 2>> $O/r05d.err
  python -c "
import json; d=json.load(open('$O/r05d_slab_${S}_$i.json')); print('headline slab=$S run $i', d['ms_per_step'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')})" | tee -a $O/r05d_slab.txt
done; done
for S in 60 120; do
  CDML_X3_SLAB_STEPS=$S python bench.py --rows 1000000 --batch 4096 --steps 200 --warmup 10 --no-extras --no-cpu-baseline > $O/r05d_slab_c1_${S}.json 2>> $O/r05d.err
  python -c "
import json; d=json.load(open('$O/r05d_slab_c1_${S}.json')); print('config1 slab=$S', d['ms_per_step'], {k: d['kernels'][k] for k in ('fc1_fwd_ms','fc2_fwd_ms','dH1_ms','dW1_ms','dW2_ms')})" | tee -a $O/r05d_slab.txt
done
