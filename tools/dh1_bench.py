#!/usr/bin/env python3
"""The K=256 data gradient of config 4 (dz1 = (dz2 . W2^T) * lrelu'(h1), bf16) is output-bound: per
launch it reads the mask (R x H bf16) and writes dz1 (R x H bf16) around a 4-K-tile product.  Times the
kernels that can run it against the bytes it has to move.  usage: python tools/dh1_bench.py [R] [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
H, D = 5120, 256
A = (torch.randn(R, D, device=dev) * 0.01).bfloat16()
B = (torch.randn(H, D, device=dev) * 0.03).bfloat16()
aux = torch.randn(R, H, device=dev).bfloat16()
out = torch.empty((R, H), device=dev, dtype=torch.bfloat16)
bytes_ = R * H * 2 * 2 + R * D * 2 + H * D * 2
for tile in sys.argv[3:] or ("128", "256", "0"):
    os.environ["CDML_BF16_TILE"] = tile
    fn = lambda: ops.gemm_bf16_nt(ops.BE_MASK_BF16, A, B, out, R, H, D, aux=aux)
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        times.append(s.elapsed_time(e) / iters)
    ms = sorted(times)[2]
    print("tile %s: %.4f ms  %.2f TB/s of %.0f MB  (%.0f TFLOP/s)  checksum %.6e"
          % (tile, ms, bytes_ / ms / 1e9, bytes_ / 1e6, 2.0 * R * H * D / ms / 1e9, out.float().abs().sum().item()))
