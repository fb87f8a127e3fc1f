#!/usr/bin/env python3
"""The fused sampler + gather alone, swept over steps per launch (the bytes one launch writes against the 256 MB Infinity
Cache) and over the SPAN of the catalogue the ids are drawn from (the footprint the address translation has to cover).
GB/s of algorithmic bytes; event pairs around 10 back-to-back launches, fresh steps every launch.
  --kind x3   fp32 table -> three bf16 planes (the headline path; 6000 B read + 9000 B written per row)
  --kind f16  fp16 table -> bf16 rows (config 4; 3000 + 3000)
  --kind f32  fp32 table -> fp32 rows (6000 + 6000)
usage: python tools/gather_sweep.py --kind x3 --rows 10000000 --batch 8192 --mode 1 [--steps 1,2,4,8] [--spans 0.01,0.1,1]
Run a variant library with CDML_LIB_PATH=build/variants/libcdml_<tag>.so."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine, engine_bf16, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kind", default="x3", choices=["x3", "f16", "f32"])
ap.add_argument("--rows", type=int, default=10000000)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--mode", type=int, default=1)
ap.add_argument("--steps", default="1,2,4,8")
ap.add_argument("--spans", default="1")
ap.add_argument("--iters", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
F, B, N = 1500, a.batch, a.rows
rpt = 3 if a.mode == 0 else 2
table = (engine_bf16.FeatureTableF16 if a.kind == "f16" else engine.FeatureTable).synthetic(N, F, 0, dev)
tag = os.path.basename(os.environ.get("CDML_LIB_PATH", "product"))
per_row = {"x3": 6000 + 9000, "f16": 3000 + 3000, "f32": 6000 + 6000}[a.kind]
for span in [float(v) for v in a.spans.split(",")]:
    n_span = max(1000, int(N * span))
    pairs = torch.from_numpy(bench.synth_pairs(n_span, max(1000, min(n_span // 3, 600000)), seed=0)).to(dev)
    for K in [int(v) for v in a.steps.split(",")]:
        R = B * rpt
        if a.kind == "x3":
            x = torch.empty((K, R, 3 * 1536), dtype=torch.bfloat16, device=dev)
        elif a.kind == "f16":
            x = torch.empty((K, R, 1536), dtype=torch.bfloat16, device=dev)
        else:
            x = torch.empty((K, R, 1536), dtype=torch.float32, device=dev)
        idx = torch.empty((K, R), dtype=torch.int32, device=dev)
        shift = torch.zeros(K, dtype=torch.int32, device=dev)
        if K == 1:
            fn = lambda s: ops.sample_gather(a.mode, pairs, 1234, s, B, table.data, F, idx[0], x[0], shift_out=shift)
        else:
            fn = lambda s: ops.sample_gather(a.mode, pairs, 1234, s, B, table.data, F, idx, x, shift_out=shift, n_steps=K)
        for s in range(3):
            fn(s * K)
        torch.cuda.synchronize()
        ts = []
        for it in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for j in range(10):
                fn(100 + (it * 10 + j) * K)
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        ms = float(np.median([p.elapsed_time(q) for p, q in ts])) / 10
        gb = K * R * per_row / 1e9
        print("[%s] %s gather rows=%d B=%d mode=%d span=%.3f steps/launch=%d (%d MB written) %.4f ms/launch -> %.0f GB/s = %.3f of 8 TB/s"
              % (tag, a.kind, N, B, a.mode, span, K, K * R * (per_row - (3000 if a.kind == 'f16' else 6000)) // 1000000, ms,
                 gb / ms * 1e3, gb / ms / 8.0), flush=True)
        del x
