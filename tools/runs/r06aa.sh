#!/bin/bash
# round 6, call aa: f16x2 -- LARS / momentum plane writers, semi-hard on the fused miner, catalogue inference, the record with the BASELINE configs
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py -m gpu -q > $O/r06aa_tests.txt 2>&1
echo "[r06aa] tests rc=$? $(tail -1 $O/r06aa_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06aa_tests.txt | head
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f32x3.py -m gpu -q -k "train_steps_config0 or semihard_config2 or lars_and_momentum" > $O/r06aa_parity.txt 2>&1
echo "[r06aa] parity rc=$? $(tail -1 $O/r06aa_parity.txt)"; grep -E "^(FAILED|ERROR)" $O/r06aa_parity.txt | head
timeout -k 10 900 python bench.py --only f16x2 --steps 100 --warmup 10 > $O/r06aa_f16x2.json 2> $O/r06aa_f16x2.err
echo "[r06aa] bench rc=$?"; tail -2 $O/r06aa_f16x2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06aa_f16x2.json'))['f16x2']
for k in ('value','ms_per_step','baseline_configs_on_f16x2'):
    print(k, json.dumps(d.get(k))[:900])
PY
