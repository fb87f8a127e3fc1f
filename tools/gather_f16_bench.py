#!/usr/bin/env python3
"""The fp16 fused sampler + gather (config 4: fp16 catalogue -> l2-normalised bf16 rows) alone, at config 4's shape:
10 M x 1500 fp16 table (rows of 3 072 B), B = 8192 uniform triplets (24 576 rows a step), 1 / 2 / 4 / 8 steps per launch.
GB/s of algorithmic bytes (3 000 B read + 3 000 B written per row); event pairs around 10 back-to-back launches, fresh
steps every launch (re-fetching rows would be served by the Infinity Cache).  Run a variant library with
CDML_LIB_PATH=build/variants/libcdml_<tag>.so (tools/experiments/mk_variant.py + recipes/gather_f16_*.py).
usage: python tools/gather_f16_bench.py [iters] [rows]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine_bf16, ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
F, B = 1500, 8192
dev = torch.device("cuda:0")
table = engine_bf16.FeatureTableF16.synthetic(N, F, 0, dev)
pairs = torch.from_numpy(bench.synth_pairs(N, 600000, seed=0)).to(dev)
tag = os.path.basename(os.environ.get("CDML_LIB_PATH", "product"))
for mode, rpt in ((0, 3), (1, 2)):
    for K in (1, 4, 8):
        R = B * rpt
        x = torch.empty((K, R, 1536), dtype=torch.bfloat16, device=dev)
        idx = torch.empty((K, R), dtype=torch.int32, device=dev)
        shift = torch.zeros(K, dtype=torch.int32, device=dev)
        if K == 1:
            fn = lambda s: ops.sample_gather(mode, pairs, 1234, s, B, table.data, F, idx[0], x[0], shift_out=shift)
        else:
            fn = lambda s: ops.sample_gather(mode, pairs, 1234, s, B, table.data, F, idx, x, shift_out=shift, n_steps=K)
        for s in range(3):
            fn(s * K)
        torch.cuda.synchronize()
        ts = []
        for it in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for j in range(10):
                fn(100 + (it * 10 + j) * K)
            b.record()
            ts.append((a, b))
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ts])) / 10
        gb = K * R * F * 4.0 / 1e9
        print("[%s] fp16 gather mode=%d steps/launch=%d rows=%6d  %.4f ms/launch -> %.0f GB/s = %.3f of 8 TB/s"
              % (tag, mode, K, K * R, ms, gb / ms * 1e3, gb / ms / 8.0), flush=True)
