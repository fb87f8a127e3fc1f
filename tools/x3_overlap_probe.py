#!/usr/bin/env python3
"""Experiment: can the narrow second layer (FC2: 256 blocks of 10 K-tiles) run INSIDE the half-empty third round of the
first layer (FC1: 640 tiles = 2.5 rounds of 256 CUs)?  FC2 of a row block needs FC1 of that row block only, so the forward
pass is cut in two row halves on two streams -- S1 (high priority): FC1a, FC2a;  S2: FC1b, FC2b -- and FC2a's blocks take
the CUs FC1b leaves idle.  Packed perfectly the pair costs 553 us of CU time per CU against 625 back to back.
Also: the second layer's weight gradient beside the data gradient (both need only dz2 and h1).
usage: python tools/x3_overlap_probe.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdml_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
R, F, H, D = 8192, 1536, 5120, 256


def planes(x):
    out = torch.empty((x.shape[0], 3 * x.shape[1]), dtype=torch.bfloat16, device=dev)
    ops.split_f32_bf16x3(x.contiguous(), out, x.shape[1])
    return out


x = torch.rand(R, F, device=dev)
x3 = planes(x / x.norm(dim=1, keepdim=True))
W1T = planes((torch.rand(H, F, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5)
W2T = planes((torch.rand(D, H, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5)
W2 = planes((torch.rand(H, D, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5)
b1, b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)
h1 = torch.empty(R, 3 * H, dtype=torch.bfloat16, device=dev)
bits = torch.zeros(R, H // 8, dtype=torch.uint8, device=dev)
z = torch.empty(R, D, device=dev)
wsz = max(ops.gemm_bf16x3_workspace(False, R, D, H, 6), 16)
ws = [torch.empty(wsz // 4, device=dev) for _ in range(2)]
dz2 = planes(torch.randn(R, D, device=dev) * 1e-3)
dz1 = torch.empty(R, 3 * H, dtype=torch.bfloat16, device=dev)
gW2, gb2 = torch.empty(H, D, device=dev), torch.empty(D, device=dev)
wsd = torch.empty(max(ops.gemm_bf16x3_workspace(True, H, D, R, 6), 16) // 4, device=dev)


def fc1(lo, hi):
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, x3[lo:hi], F, W1T, F, h1[lo:hi], hi - lo, H, F, plane_c=H, bias=b1, aux=bits[lo:hi])


def fc2(lo, hi, w):
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, h1[lo:hi], H, W2T, H, z[lo:hi], hi - lo, D, H, bias=b2, workspace=w)


def dh1():
    ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, dz2, D, W2, D, dz1, R, H, D, plane_c=H, aux=bits)


def dw2():
    ops.gemm_bf16x3_tn(h1, H, dz2, D, gW2, H, D, R, workspace=wsd, colsum=gb2)


hi_prio = torch.cuda.Stream(dev, priority=-1)
lo_prio = torch.cuda.Stream(dev, priority=0)


def forward_serial():
    fc1(0, R)
    fc2(0, R, ws[0])


def forward_overlap():
    cur = torch.cuda.current_stream(dev)
    e0 = torch.cuda.Event()
    e0.record(cur)
    hi_prio.wait_event(e0)
    lo_prio.wait_event(e0)
    with torch.cuda.stream(hi_prio):
        fc1(0, R // 2)
        fc2(0, R // 2, ws[0])
    with torch.cuda.stream(lo_prio):
        fc1(R // 2, R)
        fc2(R // 2, R, ws[1])
    cur.wait_stream(hi_prio)
    cur.wait_stream(lo_prio)


def backward_serial():
    dw2()
    dh1()


def backward_overlap():
    cur = torch.cuda.current_stream(dev)
    e0 = torch.cuda.Event()
    e0.record(cur)
    hi_prio.wait_event(e0)
    with torch.cuda.stream(hi_prio):
        dw2()
    dh1()
    cur.wait_stream(hi_prio)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


w = torch.randn(4096, 4096, device=dev)
for _ in range(60):
    torch.mm(w, w)
torch.cuda.synchronize()
forward_serial()
z0 = z.clone()
forward_overlap()
torch.cuda.synchronize()
print("forward results equal:", bool(torch.equal(z, z0)))
for rnd in range(3):
    print("round %d: forward serial %.1f us   two row halves on two streams %.1f us   |   dW2 then dH1 %.1f us   dW2 beside dH1 %.1f us"
          % (rnd, timed(forward_serial), timed(forward_overlap), timed(backward_serial), timed(backward_overlap)))
