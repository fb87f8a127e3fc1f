#!/bin/bash
# round 6, call k: the kNN export through the filter epilogue (tests incl. the reference's catalogue size), its record, the shard test fix
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_fullsize.py -m gpu -q -k "knn or config3_per_gpu" > $O/r06k_tests.txt 2>&1
echo "[r06k] tests rc=$? $(tail -1 $O/r06k_tests.txt)"; grep -E "^(FAILED|ERROR)" $O/r06k_tests.txt | head
timeout -k 10 300 python - > $O/r06k_knn.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
import bench
from cdml_amd import knn
dev = torch.device('cuda:0')
for fused in (False, True, False, True):
    orig = knn.knn_search
    r = None
    def ks(*a, **k):
        k.setdefault('fused', fused)
        return orig(*a, **k)
    knn.knn_search = ks
    try:
        r = bench.rec_knn(dev)
    finally:
        knn.knn_search = orig
    print('fused=%s' % fused, r['seconds'], r['value'], r['inner_product_tflops'], r['frac_of_mfma_peak'], r['mean_second_neighbour_d2'])
PY
echo "[r06k] knn rc=$?"; grep -v amdgpu.ids $O/r06k_knn.txt | tail -6
