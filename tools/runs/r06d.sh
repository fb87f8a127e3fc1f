#!/bin/bash
# round 6, call d: the indexed hinge with 4 rows per wave in registers, half tiles for the unsplit fp32-output product, kNN on the
# plane kernels at the reference's catalogue size, the probe's one-accumulator form in the production K-slabs
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32x3.py tests/test_gpu_knn.py tests/test_gpu_table.py -m gpu -q -x > $O/r06d_tests.txt 2>&1
echo "[r06d] tests rc=$? $(tail -1 $O/r06d_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06d_tests.txt | head
timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extras config2_semihard,train_table,knn > $O/r06d_bench.json 2> $O/r06d_bench.err
echo "[r06d] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06d_bench.json'))
print('headline', d['ms_per_step'])
for k in ('config2_semihard','train_table','knn'):
    r=d.get(k,{})
    print(k, r.get('ms_per_step'), r.get('error'), json.dumps(r.get('kernels')), json.dumps(r.get('roofline_dx')), r.get('seconds'), r.get('inner_product_tflops'), r.get('frac_of_mfma_peak'))
PY
timeout -k 10 300 python tools/f16x2_probe.py --no-rate > $O/r06d_f16x2_probe_errors.txt 2> $O/r06d_f16x2_probe.err
echo "[r06d] probe rc=$?"; grep -E "dW|GATE|subnormal" $O/r06d_f16x2_probe_errors.txt
