cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_f32x3.py tests/test_gpu_parity.py -x -q -m gpu -k "f32x3 or train_steps_config0" 2>&1 | tail -2
for i in 1 2; do
for v in prev new; do
L=build/variants/libcdml_x3prev.so; [ $v = new ] && L=collaborative-deep-metric-learning_amd/lib/libcdml_hip.so
echo "== $v"; CDML_LIB_PATH=$L timeout -k 10 300 python bench.py --precision f32x3 --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], json.dumps(d.get('kernels')))"
done; done
