#!/bin/bash
# round 5, GPU call C: the whole -m gpu suite on the tree so far, config 2 with the LDS-staged row constants, gather with
# balanced grids / auto steps per launch, the default bench line.
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/r05c_gpu_tests.txt 2>&1
echo "[r05c] gpu suite rc=$? $(tail -1 $O/r05c_gpu_tests.txt)"
python bench.py --mode semihard --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/r05c_config2_bench.json 2> $O/r05c_c2.err
python -c "
import json; d=json.load(open('$O/r05c_config2_bench.json')); print('config2', d['ms_per_step'], d['value'], d['kernels'])"
python tools/gather_sweep.py --kind f16 --mode 0 --steps 2,3,4 > $O/r05c_gather_sweep.txt 2>&1
python tools/gather_sweep.py --kind x3 --steps 1,2,4 >> $O/r05c_gather_sweep.txt 2>&1
python tools/gather_sweep.py --kind x3 --rows 1000000 --batch 4096 --steps 2,3,4 >> $O/r05c_gather_sweep.txt 2>&1
grep gather $O/r05c_gather_sweep.txt | cut -c1-200
python bench.py --steps 20 --warmup 5 > $O/r05c_bench.json 2> $O/r05c_bench.err
echo "[r05c] bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05c_bench.json'))
print('headline', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'fc1', d['roofline_fc1_fwd']['frac'], 'gather', d['gather']['frac'], d['gather']['steps_per_launch'])
for r in ('config1','config2_semihard','f32_mfma','config4_per_gpu','reference_recipe'):
    x=d.get(r,{})
    print(r, x.get('value'), x.get('ms_per_step'), x.get('error'), 'gather', (x.get('gather') or {}).get('frac'), (x.get('gather') or {}).get('steps_per_launch'))
PY
