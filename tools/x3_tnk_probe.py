#!/usr/bin/env python3
"""The weight-gradient products on k8-interleaved operands (cdml_gemm_bf16x3_tnk) against the row-major k-strided form
(cdml_gemm_bf16x3_tn), alone, at the step's shapes: dW1 = x_hat^T dz1 (1536 x 5120 over R rows), dW2 = h1^T dz2 (5120 x 256).
Event-timed back-to-back launches (slab combine included), results compared bit for bit; also the cost of the interleave
pass itself.  usage: python tools/x3_tnk_probe.py [R]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)


def planes(x, plane):
    hi = x.to(torch.bfloat16); r = x - hi.float(); mid = r.to(torch.bfloat16); lo = (r - mid.float()).to(torch.bfloat16)
    out = torch.zeros(x.shape[0], 3 * plane, dtype=torch.bfloat16, device=dev)
    for p, t in enumerate((hi, mid, lo)):
        out[:, p * plane:p * plane + x.shape[1]] = t
    return out


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


for name, M, N in (("dW1", 1536, 5120), ("dW2", 5120, 256)):
    A = torch.randn(R, M, device=dev, generator=g) * 0.05
    B = torch.randn(R, N, device=dev, generator=g) * 0.02
    A3, B3 = planes(A, M), planes(B, N)
    Ai = torch.empty(3 * R * M, dtype=torch.bfloat16, device=dev)
    Bi = torch.empty(3 * R * N, dtype=torch.bfloat16, device=dev)
    t_ia = timed(lambda: ops.interleave8_bf16x3(A3, M, R, M, Ai))
    t_ib = timed(lambda: ops.interleave8_bf16x3(B3, N, R, N, Bi))
    ws = torch.empty(max(ops.gemm_bf16x3_workspace(True, M, N, R, 6), 16) // 4, device=dev)
    C1, C2 = torch.empty((M, N), device=dev), torch.empty((M, N), device=dev)
    cs1, cs2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
    t_tn = timed(lambda: ops.gemm_bf16x3_tn(A3, M, B3, N, C1, M, N, R, workspace=ws, colsum=cs1))
    t_tk = timed(lambda: ops.gemm_bf16x3_tnk(Ai, M, 0, Bi, N, 0, C2, M, N, R, workspace=ws, colsum=cs2))
    same = bool(torch.equal(C1, C2) and torch.equal(cs1, cs2))
    fl = 2.0 * R * M * N
    print("%s R=%d: row-major k-strided %.1f us (%.1f TF = %.3f) | k8-interleaved %.1f us (%.1f TF = %.3f) | bit-identical %s | "
          "interleave pass: A %.1f us, B %.1f us" % (name, R, t_tn * 1e3, fl / t_tn / 1e9, fl / t_tn / 1e9 / 416.7, t_tk * 1e3,
                                                     fl / t_tk / 1e9, fl / t_tk / 1e9 / 416.7, same, t_ia * 1e3, t_ib * 1e3), flush=True)
