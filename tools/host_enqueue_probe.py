#!/usr/bin/env python3
"""How long does the HOST take to enqueue one training step (no device sync inside the loop), step by step from a cold
process?  A step whose kernels take less than that is host-bound.  usage: python tools/host_enqueue_probe.py [B] [mode] [optimizer]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine, train  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
opt = sys.argv[3] if len(sys.argv) > 3 else "lars"
dev = torch.device("cuda:0")
table = engine.FeatureTable.synthetic(1000000, 1500, seed=0, device=dev)
pairs = torch.from_numpy(bench.synth_pairs(1000000, 300000, seed=0)).to(dev)
ts = train.TrainStep(table, pairs, B, mode=mode, optimizer=opt, base_learning_rate=1.0 if opt == "lars" else 0.01,
                     device=dev, precision="f32x3")
torch.cuda.synchronize()
host, wall = [], []
for blk in range(8):
    t0 = time.perf_counter()
    hs = []
    for _ in range(25):
        a = time.perf_counter()
        ts.step()
        hs.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("steps %3d-%3d: host enqueue median %.3f ms max %.3f ms per step; 25 steps enqueued in %.1f ms, done after %.1f ms -> %.3f ms/step"
          % (blk * 25, blk * 25 + 24, np.median(hs) * 1e3, max(hs) * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) / 25 * 1e3))
