import os, sys, torch
sys.path.insert(0, "/root/repo")
from cdml_amd import engine, ops
dev = torch.device("cuda:0")
L = engine.TowerLayout(1500, 5000, 256)
R = 8192
ws = engine.TowerWorkspace(L, R, dev)
p = engine.VNetParams(L, dev, 42)
ws.x_hat.copy_(torch.rand(R, L.Fp, device=dev)); ws.dz1.copy_(torch.randn(R, L.Hp, device=dev) * 0.01)
ws.h1.copy_(torch.rand(R, L.Hp, device=dev)); ws.dz2.copy_(torch.randn(R, L.Dp, device=dev) * 0.01)
def run(with_db):
    ops.fc_bwd_weight2(ws.x_hat, ws.dz1, p.gW1, p.gb1 if with_db else None, L.Fp, L.Hp, ws.h1, ws.dz2, p.gW2,
                       p.gb2 if with_db else None, L.Hp, L.Dp, R, ws.bw)
res = {True: [], False: []}
for rnd in range(6):
    for w in (True, False):
        for _ in range(3): run(w)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run(w)
        e.record(); torch.cuda.synchronize()
        res[w].append(s.elapsed_time(e) / 20)
for w, v in res.items():
    print("bias gradients %s: median %.4f ms (min %.4f)" % ("ON " if w else "OFF", sorted(v)[len(v)//2], min(v)))
