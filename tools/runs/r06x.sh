#!/bin/bash
# round 6, call x: the three-product resident-plane walk (R3) of the fp16 form: tests, rate against the general loop, the record
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s > $O/r06x_tests.txt 2>&1
echo "[r06x] tests rc=$? $(tail -1 $O/r06x_tests.txt)"; grep -E "^(FAILED|ERROR)|^\{" $O/r06x_tests.txt | cut -c1-900 | head
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "train_steps_config0" > $O/r06x_parity.txt 2>&1
echo "[r06x] parity rc=$? $(tail -1 $O/r06x_parity.txt)"
(echo "## CDML_X3_WALK=general"; CDML_X3_WALK=general timeout -k 10 300 python tools/f16x2_rate.py; echo "## resident-plane walk (R3)"; timeout -k 10 300 python tools/f16x2_rate.py) > $O/r06x_rate.txt 2>&1
echo "[r06x] rate rc=$?"; grep -v amdgpu.ids $O/r06x_rate.txt
timeout -k 10 600 python bench.py --only f16x2 --steps 100 --warmup 10 > $O/r06x_f16x2.json 2> $O/r06x_f16x2.err
echo "[r06x] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06x_f16x2.json'))['f16x2']
for k in ('value','ms_per_step','roofline','roofline_fc1_fwd','kernels'):
    print(k, json.dumps(d.get(k))[:500])
PY
