"""Can two RCCL ranks share ONE GPU on this stack?  (A yes would let the N > 1 path run over RCCL on the
1-GPU box.)  usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/probes/rccl_same_gpu.py"""
import datetime
import os

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
try:
    t = torch.full((1024,), float(rank + 1), device=dev)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", rank, "all_reduce ok:", float(t[0]))
    a = torch.arange(world * 4, device=dev, dtype=torch.float32) + 100 * rank
    o = torch.empty_like(a)
    dist.all_to_all_single(o, a)
    torch.cuda.synchronize()
    print("rank", rank, "all_to_all ok:", o.tolist())
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", repr(e)[:500])
dist.destroy_process_group()
