# STATUS: the CDML_X3_STAGGER_US knob was an experiment of round 4 (profiles/r04_stagger_and_fc1_rounds.txt item 1) and is no longer in the tree
for st in 0 1 2 3 0 2; do
  echo "== stagger $st us"
  if [ $st = 0 ]; then unset CDML_X3_STAGGER_US; else export CDML_X3_STAGGER_US=$st; fi
  python tools/x3_gemm_probe.py --cases fc1,dh1 --rounds 3 2>&1 | grep -v amdgpu.ids
done
