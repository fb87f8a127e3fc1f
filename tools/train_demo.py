#!/usr/bin/env python3
"""End-to-end training at the production dimensions (1500 -> 5000 -> 256, B = 4096) on a
LEARNABLE synthetic catalogue: co-watched videos share a cluster (imitation_data's iid features
carry nothing to learn, so the benchmark's loss stays at the margin).  Prints the loss and the
held-out mean positive distance (evaluate.py:57-73) as training goes, for the fp32 paths (fp32 MFMA,
"f32x3", "f16x2") and for the config-4 precision.  usage: python tools/train_demo.py [steps] [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import engine, engine_bf16, evaluate, predict, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
dev = torch.device("cuda:0")
F, C, B = 1500, 2000, 4096
g = torch.Generator(device=dev)
g.manual_seed(0)
centers = torch.rand(C, F, device=dev, generator=g)
cluster = torch.randint(0, C, (N,), device=dev, generator=g)
feats = (centers[cluster] + 0.35 * torch.randn(N, F, device=dev, generator=g)).clamp_(0.0, 1.0)
# co-watch pairs: two different videos of one cluster
order = torch.argsort(cluster)
a, b = order[:-1], order[1:]
same = cluster[a] == cluster[b]
pairs = torch.stack([a[same], b[same]], 1).to(torch.int32)
pairs = pairs[torch.randperm(pairs.shape[0], device=dev, generator=g)]
eval_pairs, train_pairs = pairs[:2000], pairs[2000:].contiguous()
print("catalogue %d x %d, %d clusters, %d training pairs, %d held-out pairs" % (N, F, C, len(train_pairs), len(eval_pairs)))
combos = [tuple(c.split(":")) for c in (sys.argv[3].split(",") if len(sys.argv) > 3 else
                                        ["f32:inbatch", "f32x3:inbatch", "f32x3:semihard", "f16x2:inbatch", "f16x2:semihard",
                                         "bf16:inbatch"])]
for prec, mode in combos:
    if prec != "bf16":
        table = engine.FeatureTable(torch.zeros((N, 1536), device=dev), F)
        table.data[:, :F] = feats
    else:
        table = engine_bf16.FeatureTableF16.from_numpy(feats.cpu().numpy(), dev)
    ts = train.TrainStep(table, train_pairs, B, mode=mode, optimizer="adam", base_learning_rate=2e-4,
                         precision=prec, device=dev)
    pred = predict.Prediction(params=ts.params, device=dev, precision=prec if prec in ("f32x3", "f16x2") else "f32")
    ev_rows = torch.unique(eval_pairs.reshape(-1).to(torch.int64))
    remap = torch.full((N,), -1, dtype=torch.int64, device=dev)
    remap[ev_rows] = torch.arange(len(ev_rows), device=dev)
    ev_local = remap[eval_pairs.to(torch.int64)].cpu().numpy().tolist()
    ev_feats = feats[ev_rows]

    ev = evaluate.Evaluation(None, [], device=dev)

    def held_out():
        return ev.mean_dist(pred.predict(ev_feats), ev_local)

    print("== %s, %s negatives%s" % (prec, mode, " (mined in the epilogue of the score product)" if getattr(ts, "mine_fused", False) else ""))
    print("step %4d  held-out mean positive distance %.4f" % (0, held_out()))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for s in range(1, steps + 1):
        ts.step()
        if s % (steps // 6) == 0:
            print("step %4d  loss %.4f  held-out mean positive distance %.4f  (%.0f triplets/s so far)%s"
                  % (s, ts.loss(), held_out(), s * B / (time.perf_counter() - t0),
                     "  [plane scales moved %d times]" % ts.ws.scales.changes if prec == "f16x2" else ""))
    del ts, table
    torch.cuda.empty_cache()
