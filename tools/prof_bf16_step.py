import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from cdml_amd import engine_bf16, train
dev = torch.device("cuda:0"); N = 200000; B = 4096
rng = np.random.RandomState(0)
pairs = rng.randint(0, N, size=(500000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
table = engine_bf16.FeatureTableF16.synthetic(N, 1500, 0, dev)
ts = train.TrainStep(table, pairs, B, mode="inbatch", precision="bf16", device=dev)
for _ in range(12):
    ts.step()
torch.cuda.synchronize()
