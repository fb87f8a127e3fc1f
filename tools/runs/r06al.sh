#!/bin/bash
# round 6, call ac: the final tree -- the whole GPU suite, smoke, the driver-style line; then the f16x2 step's own profile
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/r06al_gpu_tests.txt 2>&1
echo "[r06al] gpu suite rc=$? $(tail -1 $O/r06al_gpu_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06al_gpu_tests.txt | head
timeout -k 10 300 python __graft_entry__.py --smoke > $O/r06al_smoke.txt 2>&1
echo "[r06al] smoke rc=$?"; grep smoke $O/r06al_smoke.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_style.json 2> $O/r06al_bench.err
echo "[r06al] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_driver_style.json'))
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_fc1_fwd']['frac'], d['roofline']['gather']['frac'])
for k in ('config1','config2_semihard','config4_per_gpu','reference_recipe','train_table','fusion_resnet','f32_mfma','f16x2'):
    r=d.get(k,{}); print(k, r.get('ms_per_step'), r.get('value'), r.get('error'), (r.get('roofline') or {}).get('frac'))
print('f16x2 configs', json.dumps(d.get('f16x2',{}).get('baseline_configs_on_f16x2'))[:600])
print('knn', d['knn'].get('value'), d['knn'].get('seconds'), (d['knn'].get('f16x2') or {}).get('value'))
print('predict', {k: v.get('value') for k, v in d.get('predict',{}).items() if isinstance(v, dict)})
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
