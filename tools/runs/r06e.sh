#!/bin/bash
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "indexed or semihard" > $O/r06e_tests.txt 2>&1
echo "[r06e] tests rc=$? $(tail -1 $O/r06e_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06e_tests.txt | head
timeout -k 10 300 python tools/indexed_hinge_probe.py > $O/r06e_indexed_hinge_probe.txt 2>&1
echo "[r06e] probe rc=$?"; cat $O/r06e_indexed_hinge_probe.txt | grep -v amdgpu.ids
