"""Rate of the five projection products at the headline step's shapes (16 384 rows, F = 1536, H = 5120, D = 256) on the
two-plane fp16 kernels (csrc/gemm_f16x2_256.hip: three products, general K loop) beside the six-plane bf16 kernels
(resident-plane walk): the same launches the step makes, event pairs, 5 rounds x 10 launches, median."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=16384)
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
R, F, H, D = args.rows, 1536, 5120, 256
f16 = lambda r, c: (torch.randn(r, 2 * c, device=dev) * 8).half()
bf = lambda r, c: (torch.randn(r, 3 * c, device=dev)).bfloat16()
b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)
bits = torch.zeros(R, H // 8, dtype=torch.uint8, device=dev)
gW1, gW2, z = torch.empty(F, H, device=dev), torch.empty(H, D, device=dev), torch.empty(R, D, device=dev)
db1, db2 = torch.empty(H, device=dev), torch.empty(D, device=dev)
wsb = max(ops.gemm_bf16x3_workspace(True, F, H, R, 6), ops.gemm_bf16x3_workspace(False, R, D, H, 6), ops.gemm_bf16x3_workspace(True, H, D, R, 6),
          ops.gemm_f16x2_workspace(True, F, H, R), ops.gemm_f16x2_workspace(False, R, D, H), ops.gemm_f16x2_workspace(True, H, D, R), 16)
ws = torch.empty(wsb // 4, device=dev)
x2, W1T2, h12, W2T2, W22, g22, g12 = f16(R, F), f16(H, F), f16(R, H), f16(D, H), f16(H, D), f16(R, D), f16(R, H)
x3, W1T3, h13, W2T3, W23, g23, g13 = bf(R, F), bf(H, F), bf(R, H), bf(D, H), bf(H, D), bf(R, D), bf(R, H)
o2, o3 = torch.empty(R, 2 * H, dtype=torch.float16, device=dev), torch.empty(R, 3 * H, dtype=torch.bfloat16, device=dev)
s = 2.0 ** -20
cases = [
    ("FC1", 6, 2.0 * R * F * H, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, x3, F, W1T3, F, o3, R, H, F, plane_c=H, bias=b1, aux=bits)),
    ("FC2", 6, 2.0 * R * H * D, lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, h13, H, W2T3, H, z, R, D, H, bias=b2, workspace=ws)),
    ("dH1", 6, 2.0 * R * H * D, lambda: ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, g23, D, W23, D, o3, R, H, D, plane_c=H, aux=bits)),
    ("dW1", 6, 2.0 * R * F * H, lambda: ops.gemm_bf16x3_tn(x3, F, g13, H, gW1, F, H, R, workspace=ws, colsum=db1)),
    ("dW2", 6, 2.0 * R * H * D, lambda: ops.gemm_bf16x3_tn(h13, H, g23, D, gW2, H, D, R, workspace=ws, colsum=db2)),
    ("FC1", 3, 2.0 * R * F * H, lambda: ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_X3_BITS, x2, F, W1T2, F, o2, R, H, F, s, c_scale=1.0, plane_c=H, bias=b1, aux=bits)),
    ("FC2", 3, 2.0 * R * H * D, lambda: ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_F32, h12, H, W2T2, H, z, R, D, H, s, bias=b2, workspace=ws)),
    ("dH1", 3, 2.0 * R * H * D, lambda: ops.gemm_f16x2_nt(ops.BE_MASKBITS_X3, g22, D, W22, D, o2, R, H, D, s, c_scale=1.0, plane_c=H, aux=bits)),
    ("dW1", 3, 2.0 * R * F * H, lambda: ops.gemm_f16x2_tn(x2, F, g12, H, gW1, F, H, R, s, workspace=ws, colsum=db1)),
    ("dW2", 3, 2.0 * R * H * D, lambda: ops.gemm_f16x2_tn(h12, H, g22, D, gW2, H, D, R, s, workspace=ws, colsum=db2)),
]
w = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    torch.mm(w, w)
torch.cuda.synchronize()
times = {}
for rnd in range(5):
    for name, q, fl, fn in cases:
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        times.setdefault((name, q), []).append(a.elapsed_time(b) / 10)
med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
print("# %d rows; bf16 x 3 planes, six products (resident-plane walk)  |  fp16 x 2 planes, three products" % R)
t6s = t3s = 0.0
for name, _, fl, _ in cases[:5]:
    t6, t3 = med[(name, 6)], med[(name, 3)]
    t6s += t6
    t3s += t3
    print("%-4s %8.1f us = %6.1f TF fp32-equiv (%.3f of 2.5 PF / 6)   |   %8.1f us = %6.1f TF (%.3f of 2.5 PF / 3)   x %.2f"
          % (name, t6 * 1e3, fl / t6 / 1e9, 6 * fl / (t6 * 1e-3) / 2.5e15, t3 * 1e3, fl / t3 / 1e9, 3 * fl / (t3 * 1e-3) / 2.5e15, t6 / t3))
print("sum of the five products: %.1f us -> %.1f us (x %.2f)" % (t6s * 1e3, t3s * 1e3, t6s / t3s))
