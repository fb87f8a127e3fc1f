#!/bin/bash
# round 5, GPU call G: half tiles for tile counts that are no multiple of 8 (the narrow layer at the reference recipe's 3 072 rows):
# f32x3 tests, the reference recipe with 60- and 120-step slabs, then the default line
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_dist.py -m gpu -q > $O/r05g_tests.txt 2>&1
echo "[r05g] tests rc=$? $(tail -1 $O/r05g_tests.txt)"; grep "^FAILED" $O/r05g_tests.txt | head
for S in 60 120 60 120; do
  CDML_X3_SLAB_STEPS=$S python bench.py --only reference_recipe --steps 200 --warmup 10 --no-cpu-baseline > $O/r05g_recipe_$S.json 2>> $O/r05g.err
  python -c "
import json; d=json.load(open('$O/r05g_recipe_$S.json')); r=d['reference_recipe']; print('recipe slab=$S', r['ms_per_step'], r['kernels'])" | tee -a $O/r05g_recipe.txt
done
python bench.py --steps 20 --warmup 5 > $O/r05g_bench.json 2> $O/r05g_bench.err
echo "[r05g] bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05g_bench.json'))
print('headline', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'fc1', d['roofline_fc1_fwd']['frac'], 'gather', d['gather']['frac'], d['gather']['steps_per_launch'])
for r in ('config1','config2_semihard','f32_mfma','config4_per_gpu','reference_recipe','fusion_resnet','data_learnable'):
    x=d.get(r,{})
    print(r, x.get('value'), x.get('ms_per_step'), x.get('error'), 'roof', (x.get('roofline') or {}).get('frac'), 'gather', (x.get('gather') or {}).get('frac'), (x.get('gather') or {}).get('steps_per_launch'))
print('predict', {k: v.get('value') for k, v in d.get('predict', {}).items()} if 'error' not in d.get('predict', {}) else d['predict'])
print('dp_form', {k: (v.get('ms_per_step') if isinstance(v, dict) else v) for k, v in d.get('dp_form_one_gpu', {}).items() if k in ('bucketed','two','single','error')})
PY
