#!/usr/bin/env python3
"""Micro-benchmark of the fused sampler+gather kernel: GB/s of algorithmic bytes (rows read +
normalised rows written) on a 1M x 1500 fp32 table for several batch sizes and steps-per-launch.
Timing: event pairs around 10 back-to-back launches (a pair around one 20-us launch adds ~25 %);
run it under `rocprofv3 --kernel-trace --stats` for the kernel's own timestamps.
usage: python tools/gather_bench.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
N, F = 1000000, 1500
table = engine.FeatureTable.synthetic(N, F, 0, dev)
rng = np.random.RandomState(0)
pairs = rng.randint(0, N, size=(4000000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
for B in (4096, 8192):
    for mode, rpt in ((1, 2), (0, 3)):
        for K in (1, 2, 4):
            R = B * rpt
            x = torch.empty((K, R, 1536), device=dev)
            idx = torch.empty((K, R), dtype=torch.int32, device=dev)
            shift = torch.zeros(K, dtype=torch.int32, device=dev)
            if K == 1:
                fn = lambda s: ops.sample_gather(mode, pairs, 1234, s, B, table.data, F, idx[0], x[0], shift_out=shift)
            else:
                fn = lambda s: ops.sample_gather(mode, pairs, 1234, s, B, table.data, F, idx, x, shift_out=shift,
                                                 n_steps=K)
            for s in range(3):
                fn(s)
            torch.cuda.synchronize()
            ts = []
            for it in range(iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for j in range(10):
                    fn(3 + (it * 10 + j) * K)
                b.record()
                ts.append((a, b))
            torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in ts]) / 10
            gb = 2.0 * K * R * F * 4 / 1e9
            print("B=%5d mode=%d steps/launch=%d rows=%6d  %.4f ms/launch (10 per event pair)  -> %.0f GB/s = %.3f of 8 TB/s"
                  % (B, mode, K, K * R, np.median(ms), gb / np.median(ms) * 1e3, gb / np.median(ms) / 8.0))
