# Round 4: the rewritten 16-B epilogue of the 256 x 256 kernel (`new` = the tree) against the committed one (`head` =
# build/variants/libcdml_head.so from tools/experiments/variant_from_rev.sh head <rev> gemm_bf16_256.hip); one box, alternating.
for v in head new head new; do
  echo "== $v"
  if [ $v = new ]; then unset CDML_LIB_PATH; else export CDML_LIB_PATH=$PWD/build/variants/libcdml_$v.so; fi
  python tools/x3_gemm_probe.py --cases fc1m,dh1m,c4fc1,c4dh1 --rounds 3 2>&1 | grep -v amdgpu.ids
done
