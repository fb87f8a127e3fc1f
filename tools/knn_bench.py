#!/usr/bin/env python3
"""Throughput of the exact kNN export: self-kNN of n unit vectors (256-d), k=81
(faiss_knn.py flag nearest_num).  usage: python tools/knn_bench.py [n] [k]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import knn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 81
dev = torch.device("cuda:0")
e = torch.randn(n, 256, device=dev)
knn.knn_search(e[:8192], e[:8192], k)
torch.cuda.synchronize()
t0 = time.time()
D, I = knn.knn_search(e, e, k)
torch.cuda.synchronize()
dt = time.time() - t0
print("n=%d k=%d: %.3f s  -> %.0f queries/s, %.1f TFLOP/s on the inner products, scores streamed %.2f TB/s"
      % (n, k, dt, n / dt, 2.0 * n * n * 256 / dt / 1e12, 2.0 * n * n * 4 / dt / 1e12))
print("self is first:", bool((I[:, 0] == torch.arange(n, device=dev)).all()), " mean 2nd-neighbour d^2 %.4f" % D[:, 1].mean().item())
