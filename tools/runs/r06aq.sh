#!/bin/bash
# round 6, call aq: A/B -- the second layer's weight gradient on a side stream beside the first layer's (CDML_DW2_SIDE=0/1), alternating processes
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for rnd in 1 2 3; do for sd in 0 1; do
  CDML_DW2_SIDE=$sd timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline > $O/r06aq_tmp.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r06aq_tmp.json')); k=d['kernels']
print('side=$sd round $rnd: ms_per_step', d['ms_per_step'], 'dW1', k.get('dW1_ms'), 'dW2', k.get('dW2_ms'), 'dH1', k.get('dH1_ms'), 'fc1', k.get('fc1_fwd_ms'))"
done; done > $O/r06aq_ab.txt 2>&1
echo "[r06aq] rc=$?"; cat $O/r06aq_ab.txt
