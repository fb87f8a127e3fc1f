#!/usr/bin/env python3
"""Where does the fp32 data-gradient GEMM (dx = (dy @ W^T) * lrelu'(x), k-contiguous operands) spend its time?
The product at M = 8192 rows, 5120 output columns, for contraction lengths 256 (the step's) ... 2048, with and
without the mask operand: the slope over the contraction is the MFMA loop, the intercept the per-tile prologue
and epilogue.  usage: python tools/f32_nt_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
M, KO = 8192, 5120
torch.manual_seed(0)
x_post = torch.rand(M, KO, device=dev) - 0.3
dx = torch.empty(M, KO, device=dev)


def timed_pair(fa, fb, rounds=5):
    """median ms of fa and fb, measured in alternating blocks (clock state and box drift hit both alike)"""
    res = ([], [])
    for _ in range(rounds):
        for i, fn in enumerate((fa, fb)):
            for _ in range(3):
                fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / iters)
    return sorted(res[0])[rounds // 2], sorted(res[1])[rounds // 2]


# the first ~100 ms of MFMA work after idle run at a lower clock: spend them here
w = torch.randn(4096, 4096, device=dev)
t_end = torch.cuda.Event(enable_timing=True)
for _ in range(60):
    torch.mm(w, w)
torch.cuda.synchronize()

for N in (256, 512, 1024, 2048):
    dy = torch.randn(M, N, device=dev) * 0.01
    W = torch.randn(KO, N, device=dev) * 0.02
    fl = 2.0 * M * KO * N
    a, b = timed_pair(lambda: ops.fc_bwd_data(dy, W, x_post, dx, M, KO, N),
                      lambda: ops.fc_bwd_data(dy, W, None, dx, M, KO, N))
    print("contraction %5d: with mask %.4f ms (%.3f of 157.3 TF)   without %.4f ms (%.3f)"
          % (N, a, fl / a / 1e9 / 157.3, b, fl / b / 1e9 / 157.3))
