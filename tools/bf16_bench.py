#!/usr/bin/env python3
"""Step time of the config-4 precision path (fp16 table + bf16 MFMA) vs fp32 on one GPU.
usage: python tools/bf16_bench.py [B] [steps] [inbatch|uniform]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, engine_bf16, train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mode = sys.argv[3] if len(sys.argv) > 3 else "inbatch"
dev = torch.device("cuda:0"); N = 1000000
rng = np.random.RandomState(0)
pairs = rng.randint(0, N, size=(2000000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
for prec, tab in (("f32", engine.FeatureTable), ("bf16", engine_bf16.FeatureTableF16)):
    table = tab.synthetic(N, 1500, 0, dev)
    ts = train.TrainStep(table, pairs, B, mode=mode, precision=prec, device=dev)
    for _ in range(3):
        ts.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    rpt = ts.rows_per_triplet
    fl = B * rpt * (2.0 * 1500 * 5000 + 2.0 * 5000 * 256) * 2 + B * rpt * 2.0 * 5000 * 256
    print("%-5s B=%d %s: %.3f ms/step  %.0f triplets/s  %.1f TFLOP/s  loss %.4f" % (prec, B, mode, dt * 1e3, B / dt, fl / dt / 1e12, ts.loss()))
    del ts, table; torch.cuda.empty_cache()
