#!/bin/bash
# round 6, call g: indexed hinge v3 (wave-independent scan), the miner against round 5's kernel on one box (variant library)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "indexed or semihard or config2" > $O/r06g_tests.txt 2>&1
echo "[r06g] tests rc=$? $(tail -1 $O/r06g_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06g_tests.txt | head
timeout -k 10 300 python tools/indexed_hinge_probe.py > $O/r06g_indexed_hinge_probe.txt 2>&1
echo "[r06g] probe rc=$?"; grep -v amdgpu.ids $O/r06g_indexed_hinge_probe.txt
for v in r5 ra0 ra1 r5 ra0 ra1; do
  if [ $v = r5 ]; then export CDML_LIB_PATH=$ROOT/build/variants/libcdml_r5mine.so; unset CDML_X3_RA; else unset CDML_LIB_PATH; export CDML_X3_RA=${v#ra}; fi
  timeout -k 10 120 python tools/mine_probe.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done | tee $O/r06g_mine_probe.txt
