#!/usr/bin/env python3
"""Feasibility probe: fp32 products of the tower computed on the bf16 MFMA with every fp32 operand split EXACTLY
into three bf16 planes (x = hi + mid + lo: three roundings to nearest capture 24 significant bits) and the six
plane products hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid summed in the fp32 accumulator (what is dropped --
mid*lo, lo*mid, lo*lo -- is below 2^-26 of a product; bf16 x bf16 products are exact in fp32).  Here the six
products are one long bf16 GEMM over materialised concatenations [hi hi mid hi lo mid] . [hi mid hi lo hi mid]^T,
so the existing kernels time what a plane-walking K loop would cost, and the error against fp64 is the method's.
usage: python tools/f32x3_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
PA, PB = (0, 0, 1, 0, 2, 1), (0, 1, 0, 2, 0, 1)


def split3(x):
    hi = x.to(torch.bfloat16)
    r = x - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters)
    return sorted(ts)[1]


w = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    torch.mm(w, w)
torch.cuda.synchronize()

R, Fp, Hp = 8192, 1536, 5120
x = torch.rand(R, Fp, device=dev)
x = x / x.norm(dim=1, keepdim=True)                       # unit rows, as the gather hands them to FC1
W = (torch.rand(Fp, Hp, device=dev) * 2 - 1) * (6.0 / (1500 + 5000)) ** 0.5
b = torch.zeros(Hp, device=dev)
xs, ws = split3(x), split3(W.t().contiguous())
assert torch.equal(xs[0].float() + xs[1].float() + xs[2].float(), x), "three planes hold an fp32 value exactly"
for nprod in (6, 3):
    A = torch.cat([xs[p] for p in PA[:nprod]], dim=1).contiguous()
    B = torch.cat([ws[p] for p in PB[:nprod]], dim=1).contiguous()
    K = A.shape[1]
    C = torch.empty(R, Hp, device=dev)
    wsz = ops.gemm_bf16_workspace(R, Hp, K)
    wk = torch.empty(max(wsz, 16) // 4, device=dev)
    f = lambda: ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_F32, A, B, C, R, Hp, K, bias=b, alpha=1.0, workspace=wk)
    f()
    ref = x.double() @ W.double()
    c32 = torch.empty(R, Hp, device=dev)
    ops.fc_lrelu_fwd(x, W, b, c32, R, Fp, Hp, alpha=1.0)
    e3 = (C.double() - ref).abs().max().item() / ref.abs().max().item()
    e32 = (c32.double() - ref).abs().max().item() / ref.abs().max().item()
    t3 = timed(f)
    t32 = timed(lambda: ops.fc_lrelu_fwd(x, W, b, c32, R, Fp, Hp, alpha=1.0))
    fl = 2.0 * R * Fp * Hp
    print("FC1 shape, %d plane products: %.4f ms (%.0f TF fp32-equivalent, %.3f of the bf16 peak on %d x the flops)   "
          "native fp32 kernel %.4f ms   max error / max|C| vs fp64: split %.2e, native %.2e"
          % (nprod, t3, fl / t3 / 1e9, nprod * fl / t3 / 1e9 / 2500e3 * 1e3, nprod, t32, e3, e32))
