#!/bin/bash
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 120 python tools/dbg_gather_ki.py 2>&1 | grep -v amdgpu.ids | tee $O/r05k_dbg.txt
for K in 0 1; do
  CDML_X3_KI=$K python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $O/r05k_ki_${K}.json 2>> $O/r05k.err
  python -c "
import json; d=json.load(open('$O/r05k_ki_${K}.json')); print('headline ki=$K', d['ms_per_step'], d['value'], d['kernels'], 'gather', d['gather']['frac'], d['gather']['launch_ms'], d['gather']['steps_per_launch'])" | tee -a $O/r05k_ki.txt
done
