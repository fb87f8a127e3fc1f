for v in clone f32 f32x3 clone f32 f32x3; do
echo "== CDML_SETTLE=$v --steps 20 --warmup 5"; CDML_SETTLE=$v python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'],d['value'])"; done
echo "== steady"; python bench.py --steps 200 --warmup 50 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'],d['value'])"
