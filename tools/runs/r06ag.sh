#!/bin/bash
# round 6, call ag: the default bench timed by the wall clock; the f16x2 stress incl. a 12 000-step run at the demo's settings
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
T0=$(date +%s)
timeout -k 10 900 python bench.py > $O/r06ag_default_bench.json 2> $O/r06ag_default_bench.err
echo "[r06ag] default bench rc=$? wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06ag_default_bench.json'))
print('headline', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], 'f16x2', d['f16x2']['ms_per_step'], d['f16x2']['roofline']['traffic'], d['f16x2'].get('error'))
PY
timeout -k 10 900 python tools/f16x2_stress.py 1500 12000 > $O/r06_f16x2_stress.txt 2>&1
echo "[r06ag] stress rc=$?"; grep -v amdgpu.ids $O/r06_f16x2_stress.txt | grep -E "^#" | tail -8
