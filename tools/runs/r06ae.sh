#!/bin/bash
# round 6, call ae: f16x2 under a run that moves its tensors (stress), the training demo on every path
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python tools/f16x2_stress.py 1500 > $O/r06_f16x2_stress.txt 2>&1
echo "[r06ae] stress rc=$?"; grep -v amdgpu.ids $O/r06_f16x2_stress.txt | tail -24
(echo "# tools/train_demo.py 600: end-to-end training at the production dimensions (1500 -> 5000 -> 256, B = 4096, Adam 2e-4) on a LEARNABLE catalogue"
 echo "# (co-watched videos share one of 2000 clusters), round 6 final tree: the fp32-MFMA path, the six-plane path and the two-plane fp16 path (in-batch"
 echo "# negatives; semi-hard negatives mined in the epilogue of the score product, the indexed hinge with its fused tail) and config-4 precision; gpurun, one box"
 timeout -k 10 700 python tools/train_demo.py 600 2>&1 | grep -v amdgpu.ids) > $O/r06_training_demo.txt
echo "[r06ae] demo rc=$?"; grep -E "^==|step  600" $O/r06_training_demo.txt
