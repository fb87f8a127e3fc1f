for a in "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-settle" "--steps 20 --warmup 50" "--steps 200 --warmup 10" "--steps 20 --warmup 5" "--steps 20 --warmup 5 --no-settle"; do
echo "== $a"; python bench.py $a --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'],d['value'])"; done
