# Round 4: non-temporal plane stores in the step (the consumer of the planes runs right after: does it pay for them?).
# base = shipped; epi_nt_mask = the data gradient's stores only; epi_nt = FC1's as well.  One box, alternating processes.
for v in base epi_nt_mask epi_nt base epi_nt_mask epi_nt; do
  echo "== $v"
  if [ $v = base ]; then unset CDML_LIB_PATH; else export CDML_LIB_PATH=$PWD/build/variants/libcdml_$v.so; fi
  python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms_per_step %.4f  fc1 %.1f fc2 %.1f dH1 %.1f dW1 %.1f dW2 %.1f' % (d['ms_per_step'], k['fc1_fwd_ms']*1e3, k['fc2_fwd_ms']*1e3, k['dH1_ms']*1e3, k['dW1_ms']*1e3, k['dW2_ms']*1e3))"
done
