#!/bin/bash
# round 6, call ai: the trainable catalogue on f16x2 (tests, the record), the saturation watch through the f16x2 tests
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_table.py tests/test_gpu_f16x2.py -m gpu -q > $O/r06ai_tests.txt 2>&1
echo "[r06ai] tests rc=$? $(tail -1 $O/r06ai_tests.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/r06ai_tests.txt | cut -c1-400 | head -12 || true
timeout -k 10 600 python - > $O/r06ai_table.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
class A: pass
args = A(); args.precision = "f32x3"; args.gather_ahead = "auto"; args.no_kernel_timers = False
dev = torch.device("cuda:0")
r = bench.rec_train_table(dev, args, 30, 5, {})
print(json.dumps({k: r[k] for k in ("ms_per_step", "value", "f16x2")}))
PY
echo "[r06ai] table record rc=$?"; grep -v amdgpu.ids $O/r06ai_table.txt | tail -3
