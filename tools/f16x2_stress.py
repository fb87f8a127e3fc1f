#!/usr/bin/env python3
"""Precision "f16x2" under runs that MOVE its tensors: production widths (1500 -> 5000 -> 256), B = 1024 in-batch negatives, Adam,
1 500 steps on the learnable catalogue, at two learning rates: 1e-3 (five times the demo's: it learns, the weights grow) and
the reference's 0.01 (fifty times: on this catalogue the embeddings collapse to loss = margin for a thousand steps while the
weights grow seventy-fold, then escape -- a chaotic regime in which ANY two fp32-accurate paths part ways; what is under
test is the range management: the scales must follow, nothing may saturate).  Every 100 steps: the loss beside the same run
on "f32x3", the plane scales (log2), how often they moved, the largest magnitude in the hi plane of every plane tensor
(65504 = something saturated).  After each run the two paths take ONE step from the f16x2 run's final weights on the same
batch: the gradients' relative L2 difference (two fp32-accurate paths: 1e-4 .. 1e-3, as f32 against f32x3).
usage: python tools/f16x2_stress.py [steps [long_steps]]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cdml_amd import engine_x3, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
table, pairs = bench.learnable_catalogue(200000, dev)
long_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # a third, LONG run at the demo's settings (B = 4096, lr 2e-4) when asked for
for lr, batch, steps in ((1e-3, 1024, steps), (1e-2, 1024, steps)) + (((2e-4, 4096, long_steps),) if long_steps else ()):
    mk = lambda prec: train.TrainStep(table, pairs, batch, mode="inbatch", optimizer="adam", base_learning_rate=lr, device=dev,
                                      precision=prec, gather_ahead=1)
    a, b = mk("f16x2"), mk("f32x3")
    L = a.layout
    worst = 0.0
    print("## Adam, learning rate %g, batch %d, %d steps" % (lr, batch, steps))
    print("# step  loss f16x2  loss f32x3 | log2 scales w1 w2 h1 dz2 dz1 | moves | max |hi plane| of W1T W2 h1 dz2 dz1 | max |W1|")
    for t in range(1, steps + 1):
        a.step()
        b.step()
        if t % max(100, steps // 15) == 0 or t in (1, 10, 30):
            s = a.ws.scales
            hi = [float(x[:, :w].float().abs().max()) for x, w in ((a.ws.W1T, L.Fp), (a.ws.W2, L.Dp), (a.ws.h1, L.Hp), (a.ws.dz2_2, L.Dp),
                                                                    (a.ws.dz1, L.Hp))]
            worst = max(worst, max(hi))
            print("%5d  %.5f  %.5f | %s | %d | %s | %.3f" % (t, a.loss(), b.loss(), " ".join("%d" % round(math.log2(getattr(s, k)))
                  for k in ("w1", "w2", "h1", "dz2", "dz1")), s.changes, " ".join("%.0f" % v for v in hi), float(a.params.W1.abs().max())))
    # one step each from the f16x2 run's weights, slots and batch
    b.params.flat.copy_(a.params.flat)
    b.m.copy_(a.m)
    b.v.copy_(a.v)
    engine_x3.refresh_weights(b.params, b.ws)
    b.global_step = a.global_step
    b.step_dev.fill_(a.global_step)
    a.step()
    b.step()
    ga, gb = a.params.grad.double(), b.params.grad.double()
    print("# largest hi-plane magnitude seen: %.0f (fp16 max 65504); scale moves after calibration: %d in %d steps; from the run's final "
          "weights, one step on each path: loss %.6f / %.6f, gradient relative L2 difference %.2e"
          % (worst, a.ws.scales.changes, steps, a.loss(), b.loss(), float((ga - gb).norm() / gb.norm().clamp_min(1e-300))))
    assert worst < 65504.0 and a.ws.scales.saturated == 0 and math.isfinite(a.loss()) and torch.equal(a.idx, b.idx)
    del a, b
    torch.cuda.empty_cache()
