#!/usr/bin/env python3
"""Micro-benchmark of the fused tower tail (l2norm -> hinge -> dE -> l2norm-bwd -> lrelu'), 10 launches
per event pair: with and without the step statistics (a second small launch), both negative modes, and the four
separate kernels it replaces.  usage: python tools/tail_bench.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = 256
dev = torch.device("cuda:0")


def timeit(fn, reps=20, per=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(per):
            fn()
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev])) / per * 1e3


for mode, rpt in ((1, 2), (0, 3)):
    R = B * rpt
    z = torch.randn(R, D, device=dev) * 0.3
    rows = torch.randint(0, 1000000, (R,), dtype=torch.int32, device=dev)
    shift = torch.tensor([17], dtype=torch.int32, device=dev)
    e, dz, de = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
    pos, neg, hinge = (torch.empty(B, device=dev) for _ in range(3))
    valid = torch.empty(B, dtype=torch.uint8, device=dev)
    stats = torch.zeros(8, device=dev)
    t_stats = timeit(lambda: ops.vnet_tail(mode, z, rows, shift, B, D, 0.8, e, pos, neg, hinge, dz, valid=valid,
                                           stats=stats))
    t_plain = timeit(lambda: ops.vnet_tail(mode, z, rows, shift, B, D, 0.8, e, pos, neg, hinge, dz, valid=valid))

    def separate():
        ops.l2norm_fwd(z, D, e)
        if mode == 0:
            ops.triplet_hinge(e, B, D, 0.8, pos, neg, hinge, stats[:4], de)
        else:
            ops.triplet_hinge_inbatch(e, rows, shift, B, D, 0.8, pos, neg, hinge, valid, stats[:4], de)
        ops.l2norm_bwd(z, de, D, dz, lrelu_alpha=0.2)
    t_sep = timeit(separate)
    print("mode %d B=%d rows=%d: fused+stats %.1f us | fused, no stats %.1f us | four separate kernels %.1f us"
          % (mode, B, R, t_stats, t_plain, t_sep))
