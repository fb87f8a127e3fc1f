#!/bin/bash
# round 6, call f: the indexed hinge's batched hub rows (probe), the miner's operand-swapped LDS-free epilogue and the
# cross-tile run-ahead (A/B by CDML_X3_RA), config 2 in the step
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "indexed or semihard or config2" > $O/r06f_tests.txt 2>&1
echo "[r06f] tests rc=$? $(tail -1 $O/r06f_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06f_tests.txt | head
timeout -k 10 300 python tools/indexed_hinge_probe.py > $O/r06f_indexed_hinge_probe.txt 2>&1
echo "[r06f] probe rc=$?"; grep -v amdgpu.ids $O/r06f_indexed_hinge_probe.txt
for ra in 0 1 0 1; do CDML_X3_RA=$ra timeout -k 10 120 python tools/mine_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/RA=$ra /"; done | tee $O/r06f_mine_probe.txt
timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --extras config2_semihard > $O/r06f_bench.json 2> $O/r06f_bench.err
echo "[r06f] bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06f_bench.json'))
print('headline', d['ms_per_step'])
r=d.get('config2_semihard',{}); print('config2', r.get('ms_per_step'), r.get('error'), json.dumps(r.get('kernels')))
PY
