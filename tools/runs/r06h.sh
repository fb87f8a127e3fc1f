#!/bin/bash
# round 6, call h: indexed hinge v4 (keys scanned from LDS); the miner: round 5's kernel / swapped epilogue / + run-ahead, and the
# timing ablations without epilogue / without loop, one box, alternating
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -k "indexed or semihard or config2" > $O/r06h_tests.txt 2>&1
echo "[r06h] tests rc=$? $(tail -1 $O/r06h_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06h_tests.txt | head
timeout -k 10 300 python tools/indexed_hinge_probe.py > $O/r06h_indexed_hinge_probe.txt 2>&1
echo "[r06h] probe rc=$?"; grep -v amdgpu.ids $O/r06h_indexed_hinge_probe.txt
for rnd in 1 2; do
for v in r5mine ra0 ra1 noepi6_ra0 noepi6_ra1 noloop6_ra1; do
  unset CDML_LIB_PATH CDML_X3_RA
  case $v in
    r5mine) export CDML_LIB_PATH=$ROOT/build/variants/libcdml_r5mine.so;;
    ra0) export CDML_X3_RA=0;; ra1) export CDML_X3_RA=1;;
    noepi6_ra0) export CDML_LIB_PATH=$ROOT/build/variants/libcdml_noepi6.so CDML_X3_RA=0;;
    noepi6_ra1) export CDML_LIB_PATH=$ROOT/build/variants/libcdml_noepi6.so CDML_X3_RA=1;;
    noloop6_ra1) export CDML_LIB_PATH=$ROOT/build/variants/libcdml_noloop6.so CDML_X3_RA=1;;
  esac
  timeout -k 10 120 python tools/mine_probe.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$v /"
done; done | tee $O/r06h_mine_probe.txt
