# STATUS: `head` = the shipped kernels, `base` = a tree with the SGPR-offset DMA in the plain bf16 loops (measured, not shipped:
# profiles/r04_stagger_and_fc1_rounds.txt item 5)
for v in base head base head; do
  echo "== $v"
  if [ $v = base ]; then unset CDML_LIB_PATH; else export CDML_LIB_PATH=$PWD/build/variants/libcdml_$v.so; fi
  python tools/x3_gemm_probe.py --cases c4fc1,c4fc2,c4dw1,c4dw2 --rounds 3 2>&1 | grep -v amdgpu.ids
done
