#!/bin/bash
# round 6, call ak: the fusion towers' visual branch on f16x2 (tests incl. two-rank), the fusion record on three forms
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_dist.py -m gpu -q -k "fusion" > $O/r06ak_tests.txt 2>&1
echo "[r06ak] tests rc=$? $(tail -1 $O/r06ak_tests.txt)"; (grep -E "^(FAILED|ERROR)|^E  " $O/r06ak_tests.txt | cut -c1-400 | head -12) || true
timeout -k 10 300 python bench.py --only fusion_resnet --steps 100 --warmup 10 > $O/r06ak_fusion.json 2> $O/r06ak_fusion.err
echo "[r06ak] bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r06ak_fusion.json'))['fusion_resnet']; print({k: (d[k] if not isinstance(d[k], dict) else (d[k]['ms_per_step'], d[k]['value'], d[k]['loss'])) for k in ('ms_per_step','value','loss','f32_mfma','f16x2')})"
