#!/bin/bash
# Build A/B variants of libcdml_hip.so (GEMM tuning switches) into lib/variants/.
# usage: tools/gemm_variants.sh name "-DCDML_GEMM_FRAG_PREFETCH=0 ..." [name flags ...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
OUT=$ROOT/collaborative-deep-metric-learning_amd/lib/variants
mkdir -p $OUT
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  (cd $CSRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags \
     -o $OUT/libcdml_$name.so *.hip) &
done
wait
ls -la $OUT
