#!/usr/bin/env python3
"""bf16 projection GEMMs at the config-4 shapes (R rows per GPU), 128x128 kernel vs
the 256x256 ping-pong kernel (CDML_BF16_TILE).  Operands: N(0,1)/sqrt(K) and N(0,1)
(dense random -- the pessimistic case for MFMA clocks).
usage: python tools/gemm_bf16_bench.py [R] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
F, H, D = 1536, 5120, 256
cases = [("fc1  R x H x F", ops.BE_BIAS_LRELU_BF16, R, H, F), ("fc2  R x D x H", ops.BE_BIAS_LRELU_F32, R, D, H),
         ("dH1  R x H x D", ops.BE_MASK_BF16, R, H, D), ("dW1  F x H x R", ops.BE_F32, F, H, R),
         ("dW2  H x D x R", ops.BE_F32, H, D, R)]
for name, epi, M, N, K in cases:
    A = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.zeros(N, device=dev)
    aux = torch.randn(M, N, device=dev).bfloat16() if epi == ops.BE_MASK_BF16 else None   # the lrelu mask (h1)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if epi in (ops.BE_BIAS_LRELU_F32, ops.BE_F32) else torch.bfloat16)
    ws = torch.empty(max(ops.gemm_bf16_workspace(M, N, K), 16) // 4, device=dev)
    res = {}
    for tile in ("128", "256"):
        os.environ["CDML_BF16_TILE"] = tile
        fn = lambda: ops.gemm_bf16_nt(epi, A, B, out, M, N, K, bias=bias, aux=aux, workspace=ws)
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
        res[tile] = (ms, out.float().abs().sum().item())
    fl = 2.0 * M * N * K
    print("%-16s 128: %7.4f ms %7.1f TF (%.3f) | 256: %7.4f ms %7.1f TF (%.3f) | checksum rel diff %.2e"
          % (name, res["128"][0], fl / res["128"][0] / 1e9, fl / res["128"][0] / 1e9 / 2500,
             res["256"][0], fl / res["256"][0] / 1e9, fl / res["256"][0] / 1e9 / 2500,
             abs(res["128"][1] - res["256"][1]) / max(res["128"][1], 1e-9)))

# MFMA shape of the 256x256 kernel, k-contiguous form: 32x32x16 vs 16x16x32, interleaved rounds in this process
os.environ["CDML_BF16_TILE"] = "256"
for name, epi, M, N, K in cases[:2] + cases[3:]:
    A = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    bias = torch.zeros(N, device=dev)
    out = torch.empty((M, N), device=dev, dtype=torch.float32 if epi in (ops.BE_BIAS_LRELU_F32, ops.BE_F32) else torch.bfloat16)
    ws = torch.empty(max(ops.gemm_bf16_workspace(M, N, K), 16) // 4, device=dev)
    fn = lambda: ops.gemm_bf16_nt(epi, A, B, out, M, N, K, bias=bias, workspace=ws)
    t = {"32": [], "16": []}
    outs = {}
    for rnd in range(6):
        for shape in ("32", "16"):
            os.environ["CDML_BF16_MFMA"] = shape
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            t[shape].append(s.elapsed_time(e) / iters)
            outs[shape] = out.float().clone()
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print("%-16s mfma 32x32x16: median %7.4f ms (min %7.4f) %7.1f TF (%.3f) | 16x16x32: median %7.4f ms (min %7.4f) %7.1f TF (%.3f) | "
          "max abs diff %.2e" % (name, med["32"], min(t["32"]), fl / med["32"] / 1e9, fl / med["32"] / 1e9 / 2500,
                                 med["16"], min(t["16"]), fl / med["16"] / 1e9, fl / med["16"] / 1e9 / 2500,
                                 (outs["32"] - outs["16"]).abs().max().item()))
os.environ.pop("CDML_BF16_MFMA", None)

# weight gradients from k-strided operands (no transposed copies) vs transposes + NT
os.environ["CDML_BF16_TILE"] = "0"
for name, M, N, K in (("dW1 tn F x H x R", F, H, R), ("dW2 tn H x D x R", H, D, R)):
    if not ops.gemm_bf16_tn_supported(M, N, K, M, N):
        continue
    A = (torch.randn(K, M, device=dev) / K ** 0.5).bfloat16()
    B = torch.randn(K, N, device=dev).bfloat16()
    AT, BT = torch.empty((M, K), device=dev, dtype=torch.bfloat16), torch.empty((N, K), device=dev, dtype=torch.bfloat16)
    out, out2 = torch.empty((M, N), device=dev), torch.empty((M, N), device=dev)
    ws = torch.empty(max(ops.gemm_bf16_tn_workspace(M, N, K), ops.gemm_bf16_workspace(M, N, K), 16) // 4, device=dev)

    def via_transposes():
        ops.transpose_to_bf16(A, AT, K, M)
        ops.transpose_to_bf16(B, BT, K, N)
        ops.gemm_bf16_nt(ops.BE_F32, AT, BT, out2, M, N, K, workspace=ws)
    t = {"32": [], "16": []}
    for rnd in range(6):                                  # MFMA shape A/B on the k-strided form, interleaved rounds
        for shape in ("32", "16"):
            os.environ["CDML_BF16_MFMA"] = shape
            for _ in range(2):
                ops.gemm_bf16_tn(A, B, out, M, N, K, workspace=ws)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                ops.gemm_bf16_tn(A, B, out, M, N, K, workspace=ws)
            e.record()
            torch.cuda.synchronize()
            t[shape].append(s.elapsed_time(e) / iters)
    os.environ.pop("CDML_BF16_MFMA", None)
    fl = 2.0 * M * N * K
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print("%-16s mfma 32x32x16: median %7.4f ms %7.1f TF (%.3f) | 16x16x32: median %7.4f ms %7.1f TF (%.3f)   [+ slab combine]"
          % (name, med["32"], fl / med["32"] / 1e9, fl / med["32"] / 1e9 / 2500, med["16"], fl / med["16"] / 1e9,
             fl / med["16"] / 1e9 / 2500))
    res = []
    for fn in (lambda: ops.gemm_bf16_tn(A, B, out, M, N, K, workspace=ws), via_transposes):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / iters)
    fl = 2.0 * M * N * K
    print("%-16s tn: %7.4f ms %7.1f TF (%.3f) | transposes+nt: %7.4f ms | max abs diff %.2e"
          % (name, res[0], fl / res[0] / 1e9, fl / res[0] / 1e9 / 2500, res[1], (out - out2).abs().max().item()))

# both weight gradients in one stream-K launch + fix-up against the two split-K launches + their combines
M1, N1, M2, N2, K = F, H, H, D, R
nb = ops.gemm_bf16_tn2_workspace(M1, N1, M2, N2, K)
if nb:
    A1 = (torch.randn(K, M1, device=dev) / K ** 0.5).bfloat16(); B1 = torch.randn(K, N1, device=dev).bfloat16()
    A2 = (torch.randn(K, M2, device=dev) / K ** 0.5).bfloat16(); B2 = torch.randn(K, N2, device=dev).bfloat16()
    C1, C2 = torch.empty((M1, N1), device=dev), torch.empty((M2, N2), device=dev)
    d1, d2 = torch.empty(N1, device=dev), torch.empty(N2, device=dev)
    ws = torch.empty(max(nb, ops.gemm_bf16_tn_workspace(M1, N1, K), ops.gemm_bf16_tn_workspace(M2, N2, K)) // 4, device=dev)

    def two():
        ops.gemm_bf16_tn(A2, B2, C2, M2, N2, K, workspace=ws, colsum=d2)
        ops.gemm_bf16_tn(A1, B1, C1, M1, N1, K, workspace=ws, colsum=d1)
    joint = lambda: ops.gemm_bf16_tn2(A1, B1, C1, M1, N1, A2, B2, C2, M2, N2, K, ws, colsum1=d1, colsum2=d2)
    t = {"two": [], "joint": []}
    for rnd in range(6):
        for name, fn in (("two", two), ("joint", joint)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            t[name].append(s.elapsed_time(e) / iters)
    fl = 2.0 * K * (M1 * N1 + M2 * N2)
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print("dW1 + dW2 (+ db1, db2): two split-K launches + combines: median %7.4f ms %7.1f TF (%.3f) | one stream-K launch + fix-up: "
          "median %7.4f ms %7.1f TF (%.3f)" % (med["two"], fl / med["two"] / 1e9, fl / med["two"] / 1e9 / 2500,
                                               med["joint"], fl / med["joint"] / 1e9, fl / med["joint"] / 1e9 / 2500))

# the vendor library (torch.mm -> hipBLASLt, bf16 in / bf16 out, no fused epilogue) on the same shapes
if len(sys.argv) > 3 and sys.argv[3] == "lib":
    for name, M, N, K, tb in (("lib fc1  NT", R, H, F, True), ("lib dW1  TN", F, H, R, False), ("lib fc2  NT", R, D, H, True)):
        if tb:
            A = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
            fn = lambda: torch.mm(A, B.t())
        else:
            A = (torch.randn(K, M, device=dev) / K ** 0.5).bfloat16(); B = torch.randn(K, N, device=dev).bfloat16()
            fn = lambda: torch.mm(A.t(), B)
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
        print("%-16s %7.4f ms %7.1f TF (%.3f of 2500)" % (name, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 2500))
