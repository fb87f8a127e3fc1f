#!/bin/bash
# round 6, call ab: the miner and the kNN filter on fp16 planes; LARS / momentum fp16 plane writers; the records
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_knn.py -m gpu -q > $O/r06ab_tests.txt 2>&1
echo "[r06ab] tests rc=$? $(tail -1 $O/r06ab_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06ab_tests.txt | head
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "semihard" > $O/r06ab_parity.txt 2>&1
echo "[r06ab] parity rc=$? $(tail -1 $O/r06ab_parity.txt)"; grep -E "^(FAILED|ERROR)" $O/r06ab_parity.txt | head
timeout -k 10 900 python bench.py --only f16x2 --steps 100 --warmup 10 > $O/r06ab_f16x2.json 2> $O/r06ab_f16x2.err
echo "[r06ab] bench rc=$?"; tail -2 $O/r06ab_f16x2.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06ab_f16x2.json'))['f16x2']
for k in ('value','ms_per_step','baseline_configs_on_f16x2'):
    print(k, json.dumps(d.get(k))[:900])
PY
timeout -k 10 600 python - > $O/r06ab_knn.txt 2>&1 <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
dev = torch.device("cuda:0")
for p in ("f32x3", "f16x2", "f32x3", "f16x2"):
    r = bench.rec_knn(dev, precision=p)
    print(p, r["value"], r["seconds"], r.get("frac_of_mfma_peak"))
PY
echo "[r06ab] knn rc=$?"; grep -v amdgpu.ids $O/r06ab_knn.txt | tail -6
