# Round 4 (late): config 4 with leaky-relu' as a sign bitmask (CDML_BF16_MASKBITS=1) against the value mask (unset), after the
# rewrite of the 16-B epilogue made the bit-writing FC1 tail cheap; bench.py --precision bf16, one box, alternating processes.
for v in 0 1 0 1; do
  echo "== CDML_BF16_MASKBITS=$v"
  if [ $v = 1 ]; then export CDML_BF16_MASKBITS=1; else unset CDML_BF16_MASKBITS; fi
  python bench.py --precision bf16 --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('ms_per_step %.4f  loss %s ' % (d['ms_per_step'], d.get('loss')), {a: round(b*1e3,1) for a,b in k.items() if a.endswith('_ms')})"
done
