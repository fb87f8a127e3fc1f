#!/bin/bash
# round 6, call o: final tree -- the whole GPU suite, the driver-style default line, the training demo on every path
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r06o2_gpu_tests.txt 2>&1
echo "[r06o2] gpu suite rc=$? $(tail -1 $O/r06o2_gpu_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06o2_gpu_tests.txt | head
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_style.json 2> $O/r06o2_bench.err
echo "[r06o2] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_driver_style.json'))
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_fc1_fwd']['frac'], d['roofline']['gather']['frac'], d['roofline']['gather']['frac_algorithmic_8d'])
for k in ('config1','config2_semihard','config4_per_gpu','reference_recipe','train_table','fusion_resnet','f32_mfma','f16x2'):
    r=d.get(k,{}); print(k, r.get('ms_per_step'), r.get('value'), r.get('error'), (r.get('roofline') or {}).get('frac'))
print('knn', d['knn'].get('value'), d['knn'].get('seconds')); print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
(echo "# tools/train_demo.py 600: end-to-end training at the production dimensions (1500 -> 5000 -> 256, B = 4096, Adam 2e-4) on a LEARNABLE catalogue"
 echo "# (co-watched videos share one of 2000 clusters), round 6 final tree: the fp32-MFMA path, the split-fp32 path (in-batch negatives; semi-hard negatives"
 echo "# mined in the epilogue of the score product, the indexed hinge with its fused tail) and config-4 precision; gpurun, one box"
 timeout -k 10 500 python tools/train_demo.py 600 2>&1 | grep -v amdgpu.ids) > $O/r06_training_demo.txt
echo "[r06o2] demo rc=$?"; tail -4 $O/r06_training_demo.txt
