#!/usr/bin/env python3
"""Precision "f16x2" under a run that MOVES its tensors: production widths (1500 -> 5000 -> 256), B = 1024 in-batch negatives,
Adam at the reference's learning rate 0.01 (fifty times the learnable-catalogue demo's) for 1 500 steps on the learnable
catalogue -- the weights grow by an order of magnitude, the loss collapses and the gradients with it.  Every 100 steps: the
loss beside the same run on "f32x3", the plane scales (log2), how often they moved, and the largest magnitude in the hi plane
of every plane tensor (65504 = something saturated).  usage: python tools/f16x2_stress.py [steps]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
table, pairs = bench.learnable_catalogue(200000, dev)
mk = lambda prec: train.TrainStep(table, pairs, 1024, mode="inbatch", optimizer="adam", base_learning_rate=0.01, device=dev,
                                  precision=prec, gather_ahead=1)
a, b = mk("f16x2"), mk("f32x3")
L = a.layout
worst = 0.0
print("# step  loss f16x2  loss f32x3 | log2 scales w1 w2 h1 dz2 dz1 | moves | max |hi plane| of W1T W2 h1 dz2 dz1 | max |W1|")
for t in range(1, steps + 1):
    a.step()
    b.step()
    if t % 100 == 0 or t in (1, 10, 30):
        s = a.ws.scales
        hi = [float(x[:, :w].float().abs().max()) for x, w in ((a.ws.W1T, L.Fp), (a.ws.W2, L.Dp), (a.ws.h1, L.Hp), (a.ws.dz2_2, L.Dp),
                                                                (a.ws.dz1, L.Hp))]
        worst = max(worst, max(hi))
        print("%5d  %.5f  %.5f | %s | %d | %s | %.3f" % (t, a.loss(), b.loss(), " ".join("%d" % round(math.log2(getattr(s, k)))
              for k in ("w1", "w2", "h1", "dz2", "dz1")), s.changes, " ".join("%.0f" % v for v in hi), float(a.params.W1.abs().max())))
print("# largest hi-plane magnitude seen: %.0f (fp16 max 65504); scale moves after calibration: %d in %d steps"
      % (worst, a.ws.scales.changes, steps))
assert worst < 65504.0 and math.isfinite(a.loss())
