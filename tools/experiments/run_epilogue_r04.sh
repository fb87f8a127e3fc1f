# Round 4: what bounds the plane-output epilogue of the 256 x 256 kernel (FC1: h1 planes + sign bits; the data gradient: dz1
# planes)?  Variants (tools/experiments/recipes/epi_*.py): non-temporal plane stores; every tile storing into tile (0, 0)'s
# region (no fabric writes); no stores at all.  One box, alternating processes.
for v in base epi_nt_stores epi_alias_stores epi_no_stores base epi_nt_stores epi_alias_stores epi_no_stores; do
  echo "== $v"
  if [ $v = base ]; then unset CDML_LIB_PATH; else export CDML_LIB_PATH=$PWD/build/variants/libcdml_$v.so; fi
  python tools/x3_gemm_probe.py --cases fc1m,dh1m,fc2 --rounds 3 2>&1 | grep -v amdgpu.ids
done
