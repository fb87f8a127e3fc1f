for a in "--only reference_recipe --steps 60 --warmup 5" "--only reference_recipe --steps 60 --warmup 5 --no-kernel-timers" "--only reference_recipe --steps 200 --warmup 50"; do
echo "== $a"; python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);r=d['reference_recipe'];print(r['ms_per_step'],r.get('kernels'))"; done
