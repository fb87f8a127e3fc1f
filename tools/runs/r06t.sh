#!/bin/bash
# round 6, call t: the fusion towers' visual branch on the plane kernels: parity, two-rank equivalence, the record
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_dist.py -m gpu -q -k "fusion" > $O/r06t_tests.txt 2>&1
echo "[r06t] tests rc=$? $(tail -1 $O/r06t_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06t_tests.txt | head
timeout -k 10 300 python bench.py --only fusion_resnet --steps 100 --warmup 10 > $O/r06t_fusion.json 2> $O/r06t_fusion.err
echo "[r06t] bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r06t_fusion.json')); print(json.dumps(d['fusion_resnet']))"
