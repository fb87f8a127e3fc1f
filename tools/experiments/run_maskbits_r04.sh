for v in 1 0 1 0; do
  echo "== CDML_X3_MASKBITS=$v"
  CDML_X3_MASKBITS=$v python bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'],d['kernels'])"
done
