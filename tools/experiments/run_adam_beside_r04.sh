# Round 4: the second layer's Adam launch beside dW1 on a side stream (CDML_X3_ADAM_BESIDE=1) against the serial step (=0);
# one box, alternating processes, bench.py --steps 100 (hipGraph replay).
for v in 0 1 0 1 0 1; do
  echo "== CDML_X3_ADAM_BESIDE=$v"
  CDML_X3_ADAM_BESIDE=$v python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms_per_step %.4f  loss %s  dW1 %.1f adam_w1 %.1f adam_w2 %.1f' % (d['ms_per_step'], d.get('loss'), k['dW1_ms']*1e3, k.get('adam_w1_ms',0)*1e3, k.get('adam_w2_ms',0)*1e3))"
done
