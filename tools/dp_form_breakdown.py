#!/usr/bin/env python3
"""Where does the N>1 step form lose time against the plain single-GPU step?  config 3's per-GPU workload (10 M rows,
B = 8192 in-batch) over RCCL at world size 1, hooks switched on one at a time.  usage: python tools/dp_form_breakdown.py [steps]"""
import datetime
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cdml_amd import dist as cdist, engine, train  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
precision = sys.argv[2] if len(sys.argv) > 2 else "f32x3"
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % bench.free_port(), rank=0, world_size=1, device_id=dev,
                        timeout=datetime.timedelta(seconds=120))
table = engine.FeatureTable.synthetic(10000000, 1500, seed=0, device=dev)
pairs = torch.from_numpy(bench.synth_pairs(10000000, 600000, seed=0)).to(dev)
bench.settle_gpu(dev, precision=precision[:5])


def run(name, **kw):
    ts = train.TrainStep(table, pairs, 8192, mode="inbatch", device=dev, batch_global=8192, precision=precision, **kw)
    el = bench.timed_steps(ts, steps, 5, dev)
    print("%-58s %.4f ms/step" % (name, el / steps * 1e3), flush=True)


for rnd in range(2):
    run("plain single-GPU step")
    run("+ row exchange (prefetch stream, own communicator)", exchange=cdist.RowExchange(10000000, group=dist.new_group(), skip_self=False))
    run("+ row exchange, not prefetched (in-stream)", exchange=cdist.RowExchange(10000000, group=dist.new_group(), skip_self=False),
        prefetch=False)
    for form in train.TrainStep.GRAD_SYNC_MODES:
        run("+ gradient sync '%s' (no exchange)" % form, grad_sync=cdist.GradSync(device=dev, skip_self=False), grad_sync_mode=form)
    run("+ both, 'single'", exchange=cdist.RowExchange(10000000, group=dist.new_group(), skip_self=False),
        grad_sync=cdist.GradSync(device=dev, skip_self=False), grad_sync_mode="single")
dist.destroy_process_group()
