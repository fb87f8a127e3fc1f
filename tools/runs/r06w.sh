#!/bin/bash
# round 6, call w: precision f16x2 -- the training test against the oracle, then the record on the headline workload
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s -k "training" > $O/r06w_tests.txt 2>&1
echo "[r06w] tests rc=$? $(tail -1 $O/r06w_tests.txt)"; grep -E "^(FAILED|ERROR)|^gradient error" $O/r06w_tests.txt | cut -c1-1500 | head
timeout -k 10 600 python bench.py --only f16x2 --steps 100 --warmup 10 > $O/r06w_f16x2.json 2> $O/r06w_f16x2.err
echo "[r06w] bench rc=$?"; tail -3 $O/r06w_f16x2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06w_f16x2.json'))['f16x2']
for k in ('value','ms_per_step','loss','f16x2_against_f32x3','plane_scales','plane_scales_are','roofline','roofline_fc1_fwd','kernels','gather','learnable_catalogue_loss_every_30_steps'):
    print(k, json.dumps(d.get(k))[:700])
PY
