import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cdml_amd import engine, ops
dev = torch.device("cuda:0")
N, F, B = 5000, 500, 64
Fp = 512
table = engine.FeatureTable.synthetic(N, F, 0, dev)
rng = np.random.RandomState(2)
pairs = rng.randint(0, N, size=(2000, 2)).astype(np.int32)
pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
mode, steps = int(sys.argv[1]), int(sys.argv[2])
R = B * (3 if mode == 0 else 2)
mk = lambda: (torch.zeros((steps, R, 3 * Fp), dtype=torch.bfloat16, device=dev), torch.zeros((steps, R), dtype=torch.int32, device=dev), torch.zeros(steps, dtype=torch.int32, device=dev))
x0, i0, s0 = mk(); x1, i1, s1 = mk()
xk = torch.full((steps, 3 * R * Fp), float("nan"), dtype=torch.bfloat16, device=dev)
args = lambda x, i, s: dict(idx_out=i if steps > 1 else i[0], x_out=x if steps > 1 else x[0], shift_out=s, n_steps=steps)
ops.sample_gather(mode, pairs, 77, 5, B, table.data, F, **args(x0, i0, s0))
ops.sample_gather(mode, pairs, 77, 5, B, table.data, F, x_ki=xk if steps > 1 else xk[0], **args(x1, i1, s1))
torch.cuda.synchronize()
x0 = x0.reshape(steps * R, 3 * Fp); x1 = x1.reshape(steps * R, 3 * Fp); xk = xk.reshape(-1)
d = (x0.view(torch.int16) != x1.view(torch.int16))
print("rowmajor differ:", int(d.sum()), "ids equal", bool(torch.equal(i0, i1)), bool(torch.equal(s0, s1)))
if d.any():
    nz = d.nonzero()
    print(nz[:10].tolist(), "rows", sorted(set(nz[:, 0].tolist()))[:20], "cols min/max", int(nz[:, 1].min()), int(nz[:, 1].max()))
    r, c = nz[0].tolist(); print(x0[r, c].item(), x1[r, c].item())
want = torch.empty(steps * 3 * R * Fp, dtype=torch.bfloat16, device=dev)
for st in range(steps):
    ops.interleave8_bf16x3(x1[st * R:(st + 1) * R], Fp, R, Fp, want[st * 3 * R * Fp:(st + 1) * 3 * R * Fp])
dk = (xk.view(torch.int16) != want.view(torch.int16))
print("ki differ:", int(dk.sum()), "of", dk.numel(), "nan in xk:", int(torch.isnan(xk.float()).sum()))
if dk.any():
    nz = dk.nonzero().flatten()
    print(nz[:16].tolist())
