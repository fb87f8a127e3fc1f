#!/usr/bin/env python3
"""Micro-benchmark of the gather kernels: GB/s of algorithmic bytes (rows read +
normalised rows written) for several batch sizes on a 1M x 1500 table.
usage: python tools/gather_bench.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
N, F = 1000000, 1500
table = engine.FeatureTable.synthetic(N, F, 0, dev)
rng = np.random.RandomState(0)
pairs = rng.randint(0, N, size=(4000000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
for B in (4096, 8192, 65536):
    for mode, rpt in ((1, 2), (0, 3)):
        R = B * rpt
        x = torch.empty((R, 1536), device=dev)
        idx = torch.empty(R, dtype=torch.int32, device=dev)
        shift = torch.zeros(1, dtype=torch.int32, device=dev)
        fn = lambda s: ops.sample_gather(mode, pairs, 1234, s, B, table.data, F, idx, x, shift_out=shift)
        for s in range(3):
            fn(s)
        torch.cuda.synchronize()
        ts = []
        for s in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(s + 3); b.record()
            ts.append((a, b))
        torch.cuda.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in ts])
        gb = 2.0 * R * F * 4 / 1e9
        print("B=%6d mode=%d rows=%7d  median %.4f ms  min %.4f ms  -> %.0f GB/s median, %.0f GB/s best"
              % (B, mode, R, np.median(ms), ms.min(), gb / np.median(ms) * 1e3, gb / ms.min() * 1e3))
