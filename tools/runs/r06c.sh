#!/bin/bash
# round 6, call c: the indexed hinge's block scan + fused tail, the slab rule by row-tile class, config 2 / recipe / train_table records
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32x3.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/r06c_tests.txt 2>&1
echo "[r06c] tests rc=$? $(tail -1 $O/r06c_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06c_tests.txt | head
timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extras config2_semihard,reference_recipe,train_table > $O/r06c_bench.json 2> $O/r06c_bench.err
echo "[r06c] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06c_bench.json'))
print('headline', d['ms_per_step'], d['roofline'].get('gather',{}).get('frac'), d['roofline'].get('gather',{}).get('frac_algorithmic_8d'))
for k in ('config2_semihard','reference_recipe','train_table'):
    r=d.get(k,{})
    print(k, r.get('ms_per_step'), r.get('error'), json.dumps(r.get('kernels')), json.dumps(r.get('roofline_dx')), json.dumps(r.get('table_adam')))
PY
timeout -k 10 300 python tools/f16x2_probe.py --no-rate > $O/r06c_f16x2_probe_errors.txt 2> $O/r06c_f16x2_probe.err
echo "[r06c] probe rc=$?"; tail -4 $O/r06c_f16x2_probe_errors.txt
