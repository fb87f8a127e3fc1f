#!/usr/bin/env python3
"""Is the benchmarked step doing live work?  Prints per-step loss / distances /
active fraction / gradient and activation magnitudes of the config-1 workload.
usage: python tools/step_health.py [steps] [lr]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
dev = torch.device("cuda:0")
N = 1000000
table = engine.FeatureTable.synthetic(N, 1500, 0, dev)
pairs = torch.from_numpy(bench.synth_pairs(N, N // 3, 0)).to(dev)
ts = train.TrainStep(table, pairs, 4096, mode="inbatch", optimizer="adam", base_learning_rate=lr,
                     seed=1234, weight_seed=42, device=dev)
for s in range(steps):
    ts.step()
    torch.cuda.synchronize()
    st = ts.stats.tolist()
    e = ts.ws.e[:ts.R, :256]
    print("step %2d loss %.9f pos %.3e neg %.3e active %.3f | e row-std %.3e |dz1| mean %.3e zero-frac %.3f "
          "|h1| mean %.3e |gW1| max %.3e |W1| max %.3e"
          % (s, st[0], st[1], st[2], st[3], e.std(dim=0).mean().item(), ts.ws.dz1[:ts.R].abs().mean().item(),
             (ts.ws.dz1[:ts.R, :5000] == 0).float().mean().item(), ts.ws.h1[:ts.R].abs().mean().item(),
             ts.params.gW1.abs().max().item(), ts.params.W1.abs().max().item()))
