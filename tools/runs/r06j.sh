#!/bin/bash
# round 6, call j: the whole GPU suite (TrainStep's default precision is "auto" now) and the driver-style default line
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r06j_gpu_tests.txt 2>&1
echo "[r06j] gpu suite rc=$? $(tail -1 $O/r06j_gpu_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06j_gpu_tests.txt | head -30
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/r06j_bench.json 2> $O/r06j_bench.err
echo "[r06j] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06j_bench.json'))
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline_fc1_fwd']['frac'], d['roofline'].get('gather',{}).get('frac'))
for k in ('config1','config2_semihard','config4_per_gpu','reference_recipe','train_table','fusion_resnet','f32_mfma','data_learnable'):
    r=d.get(k,{}); print(k, r.get('ms_per_step'), r.get('value'), r.get('error'), (r.get('roofline') or {}).get('frac'))
print('knn', d.get('knn')); print('dp_form', {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('dp_form_one_gpu',{}).items()})
print('predict', {k:v.get('value') for k,v in d.get('predict',{}).items()} if 'error' not in d.get('predict',{}) else d['predict'])
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
