#!/usr/bin/env python3
"""The plane GEMMs of the split-fp32 step (precision "f32x3") and the bf16 GEMMs of config 4 at their production
shapes, one by one: timed with event pairs in interleaved rounds (one process, one device), or run a fixed number
of times under `rocprofv3 --pmc ...` (tools/pmc_passes.sh) so that every counter row belongs to a known product.

usage: python tools/x3_gemm_probe.py [--cases fc1,dw1,...] [--iters N] [--rounds M] [--rows R] [--rows4 R4]
cases: x3 step (R rows, default 8192 = config 1): fc1 fc1m fc2 dh1 dh1m dw1 dw2;  config 4 (R4 rows, default 24576): c4fc1 c4fc2 c4dh1 c4dw1 c4dw2
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="fc1,fc2,dh1,dw1,dw2,c4fc1,c4fc2,c4dh1,c4dw1,c4dw2")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--rows", type=int, default=8192)
ap.add_argument("--rows4", type=int, default=24576)
ap.add_argument("--no-settle", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
F, H, D = 1536, 5120, 256
R, R4 = args.rows, args.rows4


def planes(x):
    """[rows][3 * cols] bf16: hi | mid | lo of the fp32 x"""
    out = torch.empty((x.shape[0], 3 * x.shape[1]), dtype=torch.bfloat16, device=dev)
    ops.split_f32_bf16x3(x.contiguous(), out, x.shape[1])
    return out


def unit_rows(r, c):
    x = torch.rand(r, c, device=dev)
    return x / x.norm(dim=1, keepdim=True)


cases = {}
want = args.cases.split(",")
if any(c in want for c in ("fc1", "fc1a", "fc1b", "fc1m", "fc2", "dh1", "dh1m", "dw1", "dw1n", "dw2", "dw2n")):
    x3 = planes(unit_rows(R, F))
    W1T = planes((torch.rand(H, F, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5)
    W2T = planes((torch.rand(D, H, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5)
    W2 = planes((torch.rand(H, D, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5)
    h1 = planes(torch.randn(R, H, device=dev).abs() * 0.05)
    dz1 = planes(torch.randn(R, H, device=dev) * 1e-3)
    dz2 = planes(torch.randn(R, D, device=dev) * 1e-3)
    b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)
    nb = max(ops.gemm_bf16x3_workspace(False, R, D, H, 6), ops.gemm_bf16x3_workspace(True, F, H, R, 6),
             ops.gemm_bf16x3_workspace(True, H, D, R, 6), 16)
    ws = torch.empty(nb // 4, device=dev)
    h1o = torch.empty_like(h1)
    dz1o = torch.empty_like(dz1)
    z = torch.empty(R, D, device=dev)
    gW1, gb1 = torch.empty(F, H, device=dev), torch.empty(H, device=dev)
    gW2, gb2 = torch.empty(H, D, device=dev), torch.empty(D, device=dev)
    cases["fc1"] = (2.0 * R * F * H, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, x3, F, W1T, F, h1o, R, H, F,
                                                                    plane_c=H, bias=b1))
    # FC1 by tile columns: 16 of the 20 (512 tiles: exactly two rounds of 256 CUs) and the last 4 (128 tiles: half a round)
    h1a = torch.empty(R, 3 * H, dtype=torch.bfloat16, device=dev)
    cases["fc1a"] = (2.0 * R * F * 4096, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, x3, F, W1T[:4096], F, h1a, R, 4096, F,
                                                                       plane_c=H, bias=b1[:4096]))
    cases["fc1b"] = (2.0 * R * F * 1024, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, x3, F, W1T[4096:], F, h1a[:, 4096:], R, 1024, F,
                                                                       plane_c=H, bias=b1[4096:]))
    cases["fc2"] = (2.0 * R * H * D, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, h1, H, W2T, H, z, R, D, H,
                                                                    bias=b2, workspace=ws))
    cases["dh1"] = (2.0 * R * H * D, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_MASK_X3, dz2, D, W2, D, dz1o, R, H, D,
                                                                    plane_c=H, aux=h1))
    # the forms the step runs: FC1 also writes the sign bitmask of h1, the data gradient reads it as its leaky-relu' mask
    bits = torch.zeros(R, H // 8, dtype=torch.uint8, device=dev)
    cases["fc1m"] = (2.0 * R * F * H, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, x3, F, W1T, F, h1o, R, H, F,
                                                                     plane_c=H, bias=b1, aux=bits))
    cases["dh1m"] = (2.0 * R * H * D, 6, lambda: ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, dz2, D, W2, D, dz1o, R, H, D,
                                                                     plane_c=H, aux=bits))
    cases["dw1"] = (2.0 * R * F * H, 6, lambda: ops.gemm_bf16x3_tn(x3, F, dz1, H, gW1, F, H, R, workspace=ws, colsum=gb1))
    # (without the bias-gradient column sums riding along)
    cases["dw1n"] = (2.0 * R * F * H, 6, lambda: ops.gemm_bf16x3_tn(x3, F, dz1, H, gW1, F, H, R, workspace=ws))
    cases["dw2n"] = (2.0 * R * H * D, 6, lambda: ops.gemm_bf16x3_tn(h1, H, dz2, D, gW2, H, D, R, workspace=ws))
    cases["dw2"] = (2.0 * R * H * D, 6, lambda: ops.gemm_bf16x3_tn(h1, H, dz2, D, gW2, H, D, R, workspace=ws, colsum=gb2))
if any(c.startswith("c4") for c in want):
    xb = unit_rows(R4, F).bfloat16()
    W1Tb = ((torch.rand(H, F, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5).bfloat16()
    h1b = (torch.randn(R4, H, device=dev).abs() * 0.05).bfloat16()
    dz1b = (torch.randn(R4, H, device=dev) * 1e-3).bfloat16()
    dz2b = (torch.randn(R4, D, device=dev) * 1e-3).bfloat16()
    b1b = torch.zeros(H, device=dev)
    h1ob = torch.empty_like(h1b)
    gW1b, gb1b = torch.empty(F, H, device=dev), torch.empty(H, device=dev)
    gW2b, gb2b = torch.empty(H, D, device=dev), torch.empty(D, device=dev)
    nb4 = max(ops.gemm_bf16_tn_workspace(F, H, R4), ops.gemm_bf16_tn_workspace(H, D, R4), 16)
    ws4 = torch.empty(nb4 // 4, device=dev)
    cases["c4fc1"] = (2.0 * R4 * F * H, 1, lambda: ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16, xb, W1Tb, h1ob, R4, H, F, bias=b1b))
    W2Tb = ((torch.rand(D, H, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5).bfloat16()
    W2b = W2Tb.t().contiguous()
    zb = torch.empty(R4, D, device=dev)
    b2b = torch.zeros(D, device=dev)
    dz1ob = torch.empty_like(dz1b)
    wsf = torch.empty(max(ops.gemm_bf16_workspace(R4, D, H), 16) // 4, device=dev)
    cases["c4fc2"] = (2.0 * R4 * H * D, 1, lambda: ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_F32, h1b, W2Tb, zb, R4, D, H, bias=b2b, workspace=wsf))
    cases["c4dh1"] = (2.0 * R4 * H * D, 1, lambda: ops.gemm_bf16_nt(ops.BE_MASK_BF16, dz2b, W2b, dz1ob, R4, H, D, aux=h1b))
    cases["c4dw1"] = (2.0 * R4 * F * H, 1, lambda: ops.gemm_bf16_tn(xb, dz1b, gW1b, F, H, R4, workspace=ws4, colsum=gb1b))
    cases["c4dw2"] = (2.0 * R4 * H * D, 1, lambda: ops.gemm_bf16_tn(h1b, dz2b, gW2b, H, D, R4, workspace=ws4, colsum=gb2b))

sel = [c for c in want if c in cases]
if not args.no_settle:                       # the first ~100 ms of MFMA work after idle run slower: settle the clock
    w = torch.randn(4096, 4096, device=dev)
    for _ in range(60):
        torch.mm(w, w)
    torch.cuda.synchronize()
times = {c: [] for c in sel}
for rnd in range(args.rounds):
    for c in sel:
        fn = cases[c][2]
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        times[c].append(s.elapsed_time(e) / args.iters)
for c in sel:
    fl, q, _ = cases[c]
    t = sorted(times[c])
    med = t[len(t) // 2]
    print("%-6s median %8.1f us  min %8.1f us   %7.1f TF %s  = %.3f of 2.5 PF%s"
          % (c, med * 1e3, t[0] * 1e3, fl / med / 1e9, "fp32-equivalent" if q == 6 else "bf16",
             q * fl / (med * 1e-3) / 2.5e15, " (six plane products per fp32 product)" if q == 6 else ""))
