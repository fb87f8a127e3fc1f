#!/bin/bash
# round 6, call af: f16x2 under runs that move its tensors (two learning rates)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python tools/f16x2_stress.py 1500 > $O/r06_f16x2_stress.txt 2>&1
echo "[r06af] stress rc=$?"; grep -v amdgpu.ids $O/r06_f16x2_stress.txt | grep -E "^#|^ 1[05]00|^  [159]00|^    1 "
