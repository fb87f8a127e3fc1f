#!/usr/bin/env python3
"""Micro-benchmark of the tower GEMMs at the config-1 shapes on REALISTIC data:
the buffers of one actual training step (normalised U[0,1) rows, Xavier weights,
real activations and gradients).  MFMA clocks depend on operand values (DVFS), so
dense random [-1,1] operands under-report what the step sees by up to 1.6x.
usage: python tools/gemm_bench.py [B] [iters] [inbatch|uniform] [lib]
"lib" adds the vendor library (torch.mm -> hipBLASLt/rocBLAS sgemm, fp32, no fused
epilogue) on the same operands as a yardstick."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, ops, train  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "inbatch"
dev = torch.device("cuda:0")
N = 200000
table = engine.FeatureTable.synthetic(N, 1500, 0, dev)
rng = np.random.RandomState(0)
pairs = rng.randint(0, N, size=(500000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
ts = train.TrainStep(table, pairs, B, mode=mode, device=dev)
for _ in range(2):
    ts.step()
ts.fetch(); ts.forward_loss(); ts.backward()
torch.cuda.synchronize()
p, ws, L, R = ts.params, ts.ws, ts.layout, ts.R
F, H, D = L.F, L.H, L.D

cases = [
    ("fc1_fwd  NN", 2.0 * R * F * H, lambda: ops.fc_lrelu_fwd(ws.x_hat, p.W1, p.b1, ws.h1, R, L.Fp, L.Hp)),
    ("fc2_fwd  NN", 2.0 * R * H * D, lambda: ops.fc_lrelu_fwd(ws.h1, p.W2, p.b2, ws.z, R, L.Hp, L.Dp)),
    ("dW2      TN", 2.0 * R * H * D, lambda: ops.fc_bwd_weight(ws.h1, ws.dz2, p.gW2, p.gb2, ws.bw, R, L.Hp, L.Dp)),
    ("dH1      NT", 2.0 * R * H * D, lambda: ops.fc_bwd_data(ws.dz2, p.W2, ws.h1, ws.dz1, R, L.Hp, L.Dp)),
    ("dW1      TN", 2.0 * R * F * H, lambda: ops.fc_bwd_weight(ws.x_hat, ws.dz1, p.gW1, p.gb1, ws.bw, R, L.Fp, L.Hp)),
]
if ws.sk_bytes:            # the single-GPU step's form: both weight gradients in one stream-K launch + fix-up
    cases.append(("dW1+dW2  SK", 2.0 * R * F * H + 2.0 * R * H * D,
                  lambda: ops.fc_bwd_weight2(ws.x_hat, ws.dz1, p.gW1, p.gb1, L.Fp, L.Hp, ws.h1, ws.dz2, p.gW2, p.gb2,
                                             L.Hp, L.Dp, R, ws.bw)))
if len(sys.argv) > 4 and sys.argv[4] == "lib":
    torch.backends.cuda.matmul.allow_tf32 = False
    xh, h1, dz1, dz2 = ws.x_hat[:R], ws.h1[:R], ws.dz1[:R], ws.dz2[:R]
    o1, o2, o3, o4 = torch.empty_like(h1), torch.empty_like(ws.z[:R]), torch.empty_like(p.W2), torch.empty_like(p.W1)
    cases += [
        ("lib fc1  NN", 2.0 * R * F * H, lambda: torch.mm(xh, p.W1, out=o1)),
        ("lib fc2  NN", 2.0 * R * H * D, lambda: torch.mm(h1, p.W2, out=o2)),
        ("lib dW2  TN", 2.0 * R * H * D, lambda: torch.mm(h1.t(), dz2, out=o3)),
        ("lib dH1  NT", 2.0 * R * H * D, lambda: torch.mm(dz2, p.W2.t(), out=o1)),
        ("lib dW1  TN", 2.0 * R * F * H, lambda: torch.mm(xh.t(), dz1, out=o4)),
    ]
total_ms, total_fl = 0.0, 0.0
for name, flops, fn in cases:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    total_ms += ms
    total_fl += flops
    print("%-12s %8.4f ms  %7.2f TF/s  (%.3f of 157.3)" % (name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3))
print("%-12s %8.4f ms  %7.2f TF/s  (%.3f of 157.3)" % ("sum", total_ms, total_fl / total_ms / 1e9,
                                                       total_fl / total_ms / 1e9 / 157.3))
