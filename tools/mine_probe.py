#!/usr/bin/env python3
"""The fused semi-hard miner alone at BASELINE config 2's shape (B = 8192 anchors x 16 384 embedded rows, D = 256):
event-timed launches of cdml_semihard_mine_x3 (prep + score product with the selection epilogue + finish), and of the
round-2 form (fp32-MFMA score GEMM + k_semihard_select) beside it.  Variant libraries: CDML_LIB_PATH (recipes mine_*.py).
usage: python tools/mine_probe.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = 256
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
E = torch.nn.functional.normalize(torch.randn(2 * B, D, device=dev, generator=g) + 1.0, dim=1).contiguous()
rows = torch.randint(0, 1000000, (2 * B,), device=dev, generator=g, dtype=torch.int32)
e3 = torch.zeros((2 * B, 3 * D), dtype=torch.bfloat16, device=dev)
sqn, dp = torch.zeros(2 * B, device=dev), torch.zeros(B, device=dev)
ws = torch.zeros(ops.semihard_mine_x3_workspace(B) // 4, device=dev)
neg = torch.zeros(B, dtype=torch.int32, device=dev)
S = torch.empty((B, 2 * B), device=dev)
neg2 = torch.zeros(B, dtype=torch.int32, device=dev)
tag = os.path.basename(os.environ.get("CDML_LIB_PATH", "product"))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


t_f = timed(lambda: ops.semihard_mine_x3(E, rows, B, D, e3, D, sqn, dp, ws, neg))
def old():
    ops.fc_bwd_data(E[0::2], E, None, S, B, 2 * B, D)
    ops.semihard_select(S, E, rows, B, D, sqn, neg2)
t_o = timed(old)
agree = float((neg == neg2).float().mean().item())
print("[%s] B=%d: fused miner %.4f ms (%.1f GFLOP -> %.1f TF fp32-equivalent) | score matrix form %.4f ms | same negative for %.4f of the anchors"
      % (tag, B, t_f, 2.0 * B * 2 * B * D / 1e9, 2.0 * B * 2 * B * D / t_f / 1e9, t_o, agree), flush=True)
