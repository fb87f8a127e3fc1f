#!/usr/bin/env python3
"""Where the indexed hinge's time goes in BASELINE config 2's step (1 M rows, B = 8192, semi-hard mining, precision f32x3):
how the mined negatives are distributed over the embedded rows (hub rows are a chain of dependent adds on one wave) and the
call's time with and without the fused tail, on the step's own tensors.  usage: python tools/indexed_hinge_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine, ops, train  # noqa: E402

dev = torch.device("cuda:0")
table = engine.FeatureTable.synthetic(1000000, bench.F, seed=0, device=dev)
pairs = torch.from_numpy(bench.synth_pairs(1000000, 333333, seed=0)).to(dev)
ts = train.TrainStep(table, pairs, 8192, output_size=bench.D, hidden_size=bench.H, margin=bench.MARGIN, mode="semihard",
                     optimizer="adam", base_learning_rate=0.01, device=dev, precision="f32x3")
for steps in (1, 30):
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize()
    nr = ts.neg_row.cpu().numpy()
    cnt = np.bincount(nr[nr >= 0], minlength=2 * ts.B)
    print("after %2d steps: mined negatives per embedded row: max %d, 99.9th pct %d, rows mined at all %d of %d, masked triplets %d"
          % (ts.global_step, cnt.max(), int(np.percentile(cnt, 99.9)), int((cnt > 0).sum()), 2 * ts.B, int((nr < 0).sum())))
L = ts.layout


def timed(fn, reps=30):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


e_ = ts.ws.e
base = lambda **kw: ops.triplet_hinge_indexed(e_, ts.neg_row, ts.B, L.Dp, ts.margin, ts.pos, ts.neg, ts.hinge, ts.scale, ts.stats, **kw)
print("forward + statistics only (no gradient)         %7.1f us" % timed(lambda: base()))
print("forward + backward (de)                         %7.1f us" % timed(lambda: base(de=ts.ws.de)))
print("forward + backward + fused tail (dz2 + planes)  %7.1f us" % timed(lambda: base(de=ts.ws.de, z=ts.ws.z, dz2=ts.ws.dz2, dz2_bf16=ts.ws.dz2_3, plane_bf=L.Dp)))
sep = lambda: (base(de=ts.ws.de), ops.l2norm_bwd(ts.ws.z, ts.ws.de, L.Dp, ts.ws.dz2, lrelu_alpha=ops.LRELU_ALPHA),
               ops.split_f32_bf16x3(ts.ws.dz2, ts.ws.dz2_3, L.Dp))
print("the same as separate launches                   %7.1f us" % timed(sep))
uni = torch.randint(0, 2 * ts.B, (ts.B,), dtype=torch.int32, device=dev)
print("fused, uniformly random negatives (no hubs)     %7.1f us" % timed(lambda: ops.triplet_hinge_indexed(
    e_, uni, ts.B, L.Dp, ts.margin, ts.pos, ts.neg, ts.hinge, ts.scale, ts.stats, de=ts.ws.de, z=ts.ws.z, dz2=ts.ws.dz2,
    dz2_bf16=ts.ws.dz2_3, plane_bf=L.Dp)))
