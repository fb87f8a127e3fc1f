set -o pipefail
run() { echo "== $*"; python bench.py "$@" --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/tmp/err.txt | python -c "
import json,sys;d=json.load(sys.stdin);print(d['ms_per_step'],d['value'],d['dtype'][:40],d['config']['workload'][:60],'|',d.get('roofline',{}).get('frac'))" || tail -5 /tmp/err.txt; }
run --precision f32
run --graph
run --mode uniform
run --mode semihard
run --precision f32x3-3
run --mode predict
run --only reference_recipe
run --precision bf16 --graph
