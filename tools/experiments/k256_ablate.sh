#!/bin/bash
# Timing ablations of the K = 256 streaming data-gradient kernel (gemm_bf16_k256.hip): what bounds it?
# Variant libraries are built from a scratch COPY of csrc/ with the edits below applied (the shipped
# source carries no ablation switches): nostore = the dz1 stores dropped, nomask = the mask loads dropped,
# neither = both.  WRONG RESULTS by construction; run each with
#   CDML_LIB_PATH=build/variants/libcdml_k256_<tag>.so python tools/dh1_bench.py 24576 50 0
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
mkdir -p $ROOT/build/variants
build() { # tag, sed script
  local tag=$1 d=$(mktemp -d)
  cp $CSRC/*.hip $CSRC/*.h $d/
  mkdir -p $d/../../include 2>/dev/null || true
  sed -i -e "$2" $d/gemm_bf16_k256.hip
  (cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$ROOT/include -I$CSRC -o $ROOT/build/variants/libcdml_k256_$tag.so *.hip 2>/dev/null)
  rm -rf $d
  echo built $tag
}
NOSTORE='s|if (row < g.M) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8 \*>(g.C + (int64_t)row \* g.ldc + ncol0 + ec));|asm volatile("" :: "v"(o));|'
NOMASK='s|const bool has_aux = g.aux != nullptr;|const bool has_aux = false;|'
build nostore "$NOSTORE" &
build nomask "$NOMASK" &
wait
build neither "$NOSTORE;$NOMASK" &
wait
