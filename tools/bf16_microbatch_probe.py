#!/usr/bin/env python3
"""Would the config-4 tower run faster in two micro-batches?  h1 and dz1 are 252 MB each at R = 24576 rows -- the size of the
Infinity Cache (256 MB) -- and are each read again two or three times (FC2, dW2, the data gradient's mask; dW1); at half the
rows they would stay cache-resident between producer and consumer.  Times the GEMM chain of one step (FC1, FC2, dW2, dH1, dW1;
no tail: dz2 is random data) at R rows once against R/2 rows twice, same weights.  usage: python tools/bf16_microbatch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, engine_bf16, ops  # noqa: E402

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
L = engine_bf16.layout_bf16(1500, 5000, 256)
p = engine.VNetParams(L, dev, 42)


def make(rows):
    ws = engine_bf16.TowerWorkspaceBF16(L, rows, dev)
    engine_bf16.refresh_weights(p, ws)
    x = torch.rand(rows, L.Fp, device=dev)
    x[:, L.F:] = 0
    ws.x_hat.copy_((x / x.norm(dim=1, keepdim=True)).bfloat16())
    ws.dz2.copy_(torch.randn(rows, L.Dp, device=dev) * 1e-3)
    ws.dz2_bf.copy_(ws.dz2.bfloat16())
    ws.tail_done = True
    return ws


def chain(ws):
    engine_bf16.tower_forward(p, ws, normalize=False)
    ws.tail_done = True
    engine_bf16.tower_backward(p, ws)


full, h0, h1 = make(R), make(R // 2), make(R // 2)
variants = {"one batch of %d rows" % R: lambda: chain(full),
            "two micro-batches of %d rows" % (R // 2): lambda: (chain(h0), chain(h1))}
res = {k: [] for k in variants}
for rnd in range(5):
    for name, fn in variants.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        res[name].append(s.elapsed_time(e) / 20)
for name, v in res.items():
    print("%-34s median %.4f ms  (min %.4f)" % (name, sorted(v)[len(v) // 2], min(v)))
