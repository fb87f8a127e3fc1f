#!/bin/bash
# round 6, call ad: f16x2 data-parallel -- two-rank equivalence (whole and bucketed), the bench's N > 1 path on it, the main path at N = 1
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -k "f16x2" > $O/r06ad_tests.txt 2>&1
echo "[r06ad] tests rc=$? $(tail -1 $O/r06ad_tests.txt)"; grep -E "^(FAILED|ERROR)|^E  " $O/r06ad_tests.txt | cut -c1-600 | head -20
timeout -k 10 400 python bench.py --precision f16x2 --steps 100 --warmup 10 > $O/r06ad_main.json 2> $O/r06ad_main.err
echo "[r06ad] main-path bench rc=$?"; tail -2 $O/r06ad_main.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06ad_main.json'))
print(d['value'], d['ms_per_step'], d['dtype'][:60], d['roofline']['frac'], d['roofline_fc1_fwd']['frac'], d['roofline']['gather']['frac'] if 'gather' in d['roofline'] else d.get('gather',{}).get('frac'))
PY
