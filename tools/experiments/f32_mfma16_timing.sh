#!/bin/bash
# Timing-only probe: would the fp32 GEMMs run faster on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32?
# (bf16: the 16x16x32 shape holds a higher clock, +7 % on this tower.)  Variant library in which every 32x32x2 MFMA
# (64 cycles, 4096 flop) is replaced by TWO 16x16x4 MFMAs (32 cycles, 2048 flop each) on quarters of the same
# accumulator registers, same LDS reads, same DMA, same operands -- WRONG RESULTS, same cycles and register
# footprint, live data.  If the wall time drops, the real 16x16x4 fragment layout is worth building.
#   CDML_LIB_PATH=build/variants/libcdml_f32_mfma16.so python tools/gemm_bench.py 4096 20
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/collaborative-deep-metric-learning_amd/csrc
d=$(mktemp -d)
cp $CSRC/*.hip $CSRC/*.h $d/
python3 - $d/gemm_f32.hip <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
n0 = s.count("f32x16 acc[TM][TN];")
s = s.replace("f32x16 acc[TM][TN];", "f32x4 acc[TM][TN][4];")
s = s.replace("for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;", "for (int r = 0; r < 16; ++r) acc[mi][ni][r >> 2][r & 3] = 0.f;")
old = "acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32((F).a[mi][u], (F).b[ni][u], acc[mi][ni], 0, 0, 0)"
new = ("{ acc[mi][ni][(2 * u) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((F).a[mi][u], (F).b[ni][u], acc[mi][ni][(2 * u) & 3], 0, 0, 0); "
       "acc[mi][ni][(2 * u + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32((F).a[mi][u], (F).b[ni][u], acc[mi][ni][(2 * u + 1) & 3], 0, 0, 0); }")
n1 = s.count(old)
s = s.replace(old, new)
n2 = s.count("acc[mi][ni][r];")
s = s.replace("acc[mi][ni][r];", "acc[mi][ni][r >> 2][r & 3];")
print("replaced", n0, n1, n2)
open(p, "w").write(s)
PY
mkdir -p $ROOT/build/variants
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I$CSRC -o $ROOT/build/variants/libcdml_f32_mfma16.so *.hip 2>&1 | grep -E "error" || true)
rm -rf $d
ls -la $ROOT/build/variants/libcdml_f32_mfma16.so
