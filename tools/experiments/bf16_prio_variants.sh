#!/bin/bash
# s_setprio in the bf16 ping-pong GEMM: per MFMA cluster (shipped) vs none vs static priority for the second row group
# (MI355X_MICROARCH.md, two waves per SIMD, item 4).  Variant libraries from a scratch copy of csrc/.
#   CDML_LIB_PATH=build/variants/libcdml_prio<N>.so python tools/gemm_bf16_bench.py 24576 20
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
mkdir -p $ROOT/build/variants
build() { # tag, sed script
  local tag=$1 d=$(mktemp -d)
  cp $CSRC/*.hip $CSRC/*.h $d/
  sed -i -e "$2" $d/gemm_bf16_256.hip
  (cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$CSRC -o $ROOT/build/variants/libcdml_$tag.so *.hip 2>/dev/null)
  rm -rf $d
  echo built $tag
}
build prio0 's|__builtin_amdgcn_s_setprio([01]);||' &
build prio2 's|__builtin_amdgcn_s_setprio([01]);||; s|  if (grp == 1) CDML_BARRIER();                          // group 1 runs one barrier behind|  if (grp == 1) { CDML_BARRIER(); __builtin_amdgcn_s_setprio(1); }|' &
wait
