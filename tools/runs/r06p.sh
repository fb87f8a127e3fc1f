#!/bin/bash
# round 6, call p: the kNN export's first-block length (how much goes through score blocks before the filter launch)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python - > $O/r06p_knn_first_block.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from cdml_amd import knn
dev = torch.device('cuda:0')
n, D, k = 343455, 256, 51
g = torch.Generator(device=dev); g.manual_seed(0)
e = torch.randn(n, D, device=dev, generator=g)
knn.knn_search(e[:8192], e[:8192], k)
torch.cuda.synchronize()
ref = None
for rnd in range(2):
    for fb, qc in ((32768, 262144), (16384, 131072), (8192, 65536), (65536, 262144)):
        t0 = time.perf_counter()
        Dk, Ik = knn.knn_search(e, e, k, first_block=fb, q_chunk=qc)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if ref is None: ref = (Dk.clone(), Ik.clone())
        print("first_block %6d q_chunk %6d: %.4f s  %.0f queries/s  same result: %s" % (fb, qc, el, n / el, bool(torch.equal(Ik, ref[1]) and torch.equal(Dk, ref[0]))), flush=True)
PY
echo "[r06p] rc=$?"; grep -v amdgpu.ids $O/r06p_knn_first_block.txt
