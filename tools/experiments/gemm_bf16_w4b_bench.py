#!/usr/bin/env python3
"""A/B of the two big-tile bf16 kernels on the config-4 shapes (R rows per GPU): the 256x256x64
ping-pong kernel (two waves per SIMD) vs the one-wave-per-SIMD 256x256x64 kernel (CDML_BF16_W4B=1).
Operands: step-like (|x| small, as the l2-normalised rows are) and N(0,1) (the pessimistic case for
the clocks).  usage: python tools/gemm_bf16_w4b_bench.py [R] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
F, H, D = 1536, 5120, 256
os.environ["CDML_BF16_TILE"] = "256"
cases = [("fc1  R x H x F", ops.BE_BIAS_LRELU_BF16, R, H, F), ("fc2  R x D x H", ops.BE_BIAS_LRELU_F32, R, D, H),
         ("dH1  R x H x D", ops.BE_MASK_BF16, R, H, D), ("dW1  F x H x R (k-contig copies)", ops.BE_F32, F, H, R)]
for data in ("step-like", "N(0,1)"):
    print("== operands:", data)
    for name, epi, M, N, K in cases:
        if data == "step-like":
            A = (torch.rand(M, K, device=dev) * 0.05).bfloat16()
            B = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.03).bfloat16()
        else:
            A = (torch.randn(M, K, device=dev) / K ** 0.5).bfloat16()
            B = torch.randn(N, K, device=dev).bfloat16()
        bias = torch.zeros(N, device=dev)
        aux = torch.randn(M, N, device=dev).bfloat16() if epi == ops.BE_MASK_BF16 else None
        out = torch.empty((M, N), device=dev, dtype=torch.float32 if epi in (ops.BE_BIAS_LRELU_F32, ops.BE_F32) else torch.bfloat16)
        ws = torch.empty(max(ops.gemm_bf16_workspace(M, N, K), 16) // 4, device=dev)
        res = {}
        for w4 in ("0", "1"):
            os.environ["CDML_BF16_W4B"] = w4
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
            res[w4] = (s.elapsed_time(e) / iters, out.float().double().abs().sum().item())
        fl = 2.0 * M * N * K
        print("%-34s ping-pong %7.4f ms %7.1f TF (%.3f) | 1 wave/SIMD %7.4f ms %7.1f TF (%.3f) | checksum rel diff %.1e"
              % (name, res["0"][0], fl / res["0"][0] / 1e9, fl / res["0"][0] / 1e9 / 2500, res["1"][0],
                 fl / res["1"][0] / 1e9, fl / res["1"][0] / 1e9 / 2500,
                 abs(res["0"][1] - res["1"][1]) / max(res["0"][1], 1e-9)))
