#!/usr/bin/env python3
"""Throughput of the device co-watch graph / selection (parse_data.py:221-289 equivalents).
usage: python tools/etl_bench.py [n_pairs] [n_videos]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import parse_data  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
dev = torch.device("cuda:0")
a = torch.randint(0, N, (P,), device=dev, dtype=torch.int32)
# popular videos co-occur more often: skew the partner towards small ids
b = ((a.long() + 1 + (torch.rand(P, device=dev) ** 3 * (N - 1)).long()) % N).to(torch.int32)
pairs = torch.stack([a, b], 1).contiguous()
parse_data.select_cowatch(pairs[:1000], 2, device=dev)
torch.cuda.synchronize()
for name, fn in (("graph (distinct edges + counts)", lambda: parse_data.cowatch_graph(pairs, device=dev)),
                 ("select threshold 2", lambda: parse_data.select_cowatch(pairs, 2, device=dev)),
                 ("select threshold 2, unique", lambda: parse_data.select_cowatch(pairs, 2, unique=True, device=dev))):
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = out[0].shape[0] if isinstance(out, tuple) else out.shape[0]
    print("%-34s %d pairs -> %d rows in %.3f s  (%.0f M pairs/s)" % (name, P, n, dt, P / dt / 1e6))
