#!/bin/bash
# Build A/B variants of the library for the sampler+gather kernel (rows in flight per wave,
# non-temporal loads / stores) into build/variants/ (travels to the GPU box with the snapshot);
# run each with:  CDML_LIB_PATH=build/variants/libcdml_<tag>.so python tools/gather_bench.py
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
mkdir -p $ROOT/build/variants
build() { # tag, flags...
  local tag=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o $ROOT/build/variants/libcdml_$tag.so $CSRC/*.hip 2>/dev/null
  echo built $tag
}
build rpw1 -DCDML_GATHER_ROWS_PER_WAVE=1 &
build rpw4 -DCDML_GATHER_ROWS_PER_WAVE=4 &
wait
build rpw2_ntld -DCDML_GATHER_NT=1 &
build rpw2_ntst -DCDML_GATHER_NT_STORE=1 &
wait
build rpw1_ntst -DCDML_GATHER_ROWS_PER_WAVE=1 -DCDML_GATHER_NT_STORE=1 &
build rpw2_ntboth -DCDML_GATHER_NT=1 -DCDML_GATHER_NT_STORE=1 &
wait
