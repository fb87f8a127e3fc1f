#!/bin/bash
# round 6, call m: the whole GPU suite, the f16x2 probe (both tables, one file), then the evidence passes
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r06m_gpu_tests.txt 2>&1
echo "[r06m] gpu suite rc=$? $(tail -1 $O/r06m_gpu_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06m_gpu_tests.txt | head
timeout -k 10 600 python tools/f16x2_probe.py > $O/r06_f16x2_probe.txt 2> $O/r06m_f16x2_probe.err
echo "[r06m] probe rc=$?"; grep -E "GATE|subnormal|sum of the five" $O/r06_f16x2_probe.txt
bash tools/runs/r06_profile.sh $1
