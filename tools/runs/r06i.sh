#!/bin/bash
# round 6, call i: the miner's lazy "farthest" scan against round 5's kernel (one box, alternating), indexed hinge with 16 hits in flight
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_f32x3.py -m gpu -q -k "indexed or semihard or config2 or mine" > $O/r06i_tests.txt 2>&1
echo "[r06i] tests rc=$? $(tail -1 $O/r06i_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06i_tests.txt | head
timeout -k 10 300 python tools/indexed_hinge_probe.py > $O/r06i_indexed_hinge_probe.txt 2>&1
echo "[r06i] probe rc=$?"; grep -v amdgpu.ids $O/r06i_indexed_hinge_probe.txt
for rnd in 1 2 3; do
for v in r5mine product; do
  unset CDML_LIB_PATH
  [ $v = r5mine ] && export CDML_LIB_PATH=$ROOT/build/variants/libcdml_r5mine.so
  timeout -k 10 120 python tools/mine_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$v /"
done; done | tee $O/r06i_mine_probe.txt
