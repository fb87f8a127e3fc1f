#!/bin/bash
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fullsize.py -m gpu -q > $O/r05o_tests.txt 2>&1
echo "[r05o] tests rc=$? $(tail -1 $O/r05o_tests.txt)"; grep "^FAILED" $O/r05o_tests.txt | head
for i in 1 2; do for L in "" build/variants/libcdml_gather_f16_oldnorm.so build/variants/libcdml_gather_f16_nonorm.so; do
  CDML_LIB_PATH=$L python tools/gather_sweep.py --kind f16 --mode 0 --steps 3,4 2>&1 | grep gather | sed 's/^\[\]/[product: dot2 norm]/' | tee -a $O/r05o_f16_norm.txt
done; done
