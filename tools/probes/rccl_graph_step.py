"""The data-parallel TrainStep captured into hipGraphs WITH its RCCL collectives inside (world size 1 on
cuda:0, skip_self=False: the two all-to-alls of the row exchange on the forked prefetch stream / own
communicator and the bucketed asynchronous all-reduces).  Prints progress; run under a timeout.
usage: python tools/probes/rccl_graph_step.py [exchange] [allreduce]"""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cdml_amd import dist as cdist, engine, train  # noqa: E402
from oracle import synth as osynth  # noqa: E402  (probe = test infrastructure)


faulthandler.enable()
EX_SELF = "exchange" not in sys.argv[1:]      # argv: which collectives stay inside the capture
GS_SELF = "allreduce" not in sys.argv[1:]


def say(m):
    print("[rccl-graph-step] " + m, flush=True)


os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n_rows, F, H, D, B = 3000, 200, 300, 64, 64
pairs = torch.from_numpy(osynth.cowatch_pairs(n_rows, 400, 0)).to(dev)
table = engine.FeatureTable.synthetic(n_rows, F, 0, dev)
mk = lambda g_: train.TrainStep(table, pairs, B, hidden_size=H, output_size=D, mode="uniform", device=dev,
                                exchange=cdist.RowExchange(n_rows, group=dist.new_group(), skip_self=EX_SELF),
                                grad_sync=cdist.GradSync(device=dev, skip_self=GS_SELF), batch_global=B, use_graph=g_)
e1, g1 = mk(False), mk(True)
for i in range(6):
    say("step %d eager" % i)
    e1.step()
    torch.cuda.synchronize()
    say("step %d graph path" % i)
    g1.step()
    torch.cuda.synchronize()
say("graphs captured: %d" % len(g1._graphs))
say("equal to eager: %s (max diff %g)" % (torch.equal(e1.params.flat, g1.params.flat),
                                           float((e1.params.flat - g1.params.flat).abs().max())))
g1._graphs.clear()
torch.cuda.synchronize()
say("graphs released")
dist.destroy_process_group()
say("done")
