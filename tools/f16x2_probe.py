#!/usr/bin/env python3
"""Probe: could the fp32 products of the step run as THREE fp16 plane products instead of six bf16 ones?

    x 2^s = hi + lo 2^-L,  hi = fp16(x 2^s),  lo = fp16((x 2^s - hi) 2^L),   products hi*hi + 2^-L (hi*lo + lo*hi)
    on v_mfma_f32_16x16x32_f16, s = a per-tensor power of two (exact) that puts max |x| just under 2^15.

Two tables, at the five production shapes of config 1 on the operands of
tests/test_gpu_f32x3.py::test_error_at_the_five_production_shapes_against_the_fp32_mfma_kernels:

  1. ERROR against fp64 (max |err| / max |result|, and relative L2) of: the fp32-MFMA kernels, the six-plane kernels (the
     headline path), and the two-plane fp16 form in two variants (tools/micro/f16x2_gemm.hip: cross terms in their own
     accumulator with lo scaled by 2^11; lo unscaled -- subnormal-prone -- with one accumulator) plus hi*hi alone.  The
     gradient operands also at HEAVY-TAILED magnitudes (|dz2| from 1e-8 to 1e-3, what fp16's five exponent bits may not
     hold under one per-tensor scale).  GATE: <= 1.5 x the fp32-MFMA kernel's error (+ 2e-8) at all five shapes.
  2. RATE: the production plane kernels walking THREE products instead of six (`products=3`, the general K loop; bf16 and
     fp16 MFMAs of this shape issue at the same rate) against the six-product resident-plane walk -- the matrix cycles a
     two-plane path would spend, with the DMA : MFMA ratio such a path has (4 images per 3 products against 6 per 6).

usage (GPU box): python tools/f16x2_probe.py [--rows 8192] > profiles/r06_f16x2_probe.txt
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cdml_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=8192)
ap.add_argument("--no-rate", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")

SO = os.path.join(ROOT, "build", "probes", "libf16x2_probe.so")
SRC = os.path.join(ROOT, "tools", "micro", "f16x2_gemm.hip")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC], check=True)
lib = C.CDLL(SO)
lib.f16x2_gemm_nt.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                                          C.c_float, C.c_float, C.c_void_p]


def split16(x, L):
    """(hi, lo, s): x 2^s = hi + lo 2^-L with hi, lo fp16 (the residual x 2^s - hi is exact in fp32)"""
    amax = float(x.abs().max())
    s = 14 - int(torch.floor(torch.log2(torch.tensor(amax))).item())       # max |x| 2^s in [2^14, 2^15)
    xs = x * (2.0 ** s)
    hi = xs.half()
    lo = ((xs - hi.float()) * (2.0 ** L)).half()
    return hi.contiguous(), lo.contiguous(), s


def f16x2_nt(A, B, mode, slabs=1):
    """C = A . B^T (fp32 [M][K], [N][K]) through the two-plane fp16 form; ``slabs`` > 1: the contraction in that many
    K-slabs, each in its own accumulator, summed in fp32 in slab order -- the partition the production weight-gradient
    kernels use (split-K slabs + k_x3_sum_slabs): what bounds a long contraction's error is the number of MFMA
    accumulation steps into ONE accumulator."""
    L = 11 if mode == 0 else 0
    ah, al, sa = split16(A, L)
    bh, bl, sb = split16(B, L)
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty(M, N, device=dev)
    per = (K // slabs + 31) // 32 * 32
    total = None
    for k0 in range(0, K, per):
        kk = min(per, K - k0)
        off = k0 * 2                                           # bytes into a row of an fp16 plane
        rc = lib.f16x2_gemm_nt(mode, ah.data_ptr() + off, al.data_ptr() + off, bh.data_ptr() + off, bl.data_ptr() + off, K, K,
                               out.data_ptr(), N, M, N, kk, 2.0 ** -11, 2.0 ** -(sa + sb), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        total = out.clone() if total is None else total + out
    return total


def planes(x, w=None):
    w = w or x.shape[1]
    out = torch.zeros((x.shape[0], 3 * w), dtype=torch.bfloat16, device=dev)
    ops.split_f32_bf16x3(x.contiguous(), out, w)
    return out


def ws_x3(tn, M, N, K):
    return torch.empty(max(ops.gemm_bf16x3_workspace(tn, M, N, K, 6), 16) // 4, device=dev)


def errs(got, ref):
    d = got.double() - ref
    return (d.abs().max() / ref.abs().max()).item(), (d.norm() / ref.norm()).item()


torch.manual_seed(7)
R, F, H, D = args.rows, 1536, 5120, 256
x = torch.rand(R, F, device=dev)
x[:, 1500:] = 0
x = x / x.norm(dim=1, keepdim=True)
W1 = (torch.rand(F, H, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5
W2 = (torch.rand(H, D, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5
lrelu = lambda t: torch.maximum(t, 0.2 * t)
b1 = torch.randn(H, device=dev) * 0.01
h1 = lrelu(x.double() @ W1.double() + b1.double()).float()
zero_b1, zero_b2 = torch.zeros(H, device=dev), torch.zeros(D, device=dev)

print("# f16x2 probe (tools/f16x2_probe.py): %d rows, F = 1536 (1500), H = 5120, D = 256; device %s" % (R, torch.cuda.get_device_name(0)))
print("# error against fp64: max |err| / max |result|   (relative L2 in brackets)")
print("%-26s %-22s %-22s %-22s %-22s %-12s %s" % ("product", "fp32 MFMA", "six bf16 planes", "f16x2 own-acc (L=11)", "f16x2 one-acc (L=0)",
                                                   "hi*hi alone", "gate (<= 1.5 x fp32 MFMA + 2e-8)"))
fmt = lambda e: "%.2e (%.2e)" % e
gate_ok = True


gate_one_slabs = True


def row(name, ref, c32, c6, A, B, slabs=1):
    """A [M][K], B [N][K]: the NT operands of the product (k-strided products pass their transposes).  ``slabs``: the K
    partition of the production kernel for this product (the weight gradients are split-K launches)."""
    global gate_ok, gate_one_slabs
    e32, e6 = errs(c32, ref), errs(c6, ref)
    e0, e1, e2 = errs(f16x2_nt(A, B, 0), ref), errs(f16x2_nt(A, B, 1), ref), errs(f16x2_nt(A, B, 2), ref)
    ok0, ok1 = e0[0] <= 1.5 * e32[0] + 2e-8, e1[0] <= 1.5 * e32[0] + 2e-8
    gate_ok = gate_ok and ok0
    extra = ""
    if slabs > 1:
        e1s = errs(f16x2_nt(A, B, 1, slabs), ref)
        ok1s = e1s[0] <= 1.5 * e32[0] + 2e-8
        gate_one_slabs = gate_one_slabs and ok1s
        extra = "; one-acc in the production's %d K-slabs: %s %s (%.2f x)" % (slabs, fmt(e1s), "PASS" if ok1s else "FAIL", e1s[0] / e32[0])
    else:
        gate_one_slabs = gate_one_slabs and ok1
    print("%-26s %-22s %-22s %-22s %-22s %-12.2e own-acc %s (%.2f x), one-acc %s (%.2f x)%s"
          % (name, fmt(e32), fmt(e6), fmt(e0), fmt(e1), e2[0], "PASS" if ok0 else "FAIL", e0[0] / e32[0], "PASS" if ok1 else "FAIL",
             e1[0] / e32[0], extra))


for tag, dz2 in (("|dz2| ~ 1e-3 (the test's)", torch.randn(R, D, device=dev) * 1e-3),
                 ("|dz2| 1e-8 .. 1e-3 heavy tail", torch.randn(R, D, device=dev) * 10.0 ** (-8 + 5 * torch.rand(R, D, device=dev)))):
    dz1 = ((dz2.double() @ W2.double().t()) * torch.where(h1 > 0, 1.0, 0.2).double()).float()
    print("# gradient operands: %s; max |dz1| %.2e, median |dz1| %.2e" % (tag, dz1.abs().max().item(), dz1.abs().median().item()))
    x3, h13, dz13, dz23 = planes(x), planes(h1), planes(dz1), planes(dz2)
    W1T3, W2T3, W23 = planes(W1.t().contiguous()), planes(W2.t().contiguous()), planes(W2)
    if tag.startswith("|dz2| ~"):
        # FC1 (no bias / activation here: the products are what is compared)
        ref = x.double() @ W1.double()
        c32 = torch.empty(R, H, device=dev)
        ops.fc_bwd_data(x, W1.t().contiguous(), None, c32, R, H, F)               # plain fp32-MFMA x . (W1^T)^T
        c6 = torch.empty(R, H, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_F32, x3, F, W1T3, F, c6, R, H, F)
        row("FC1  R x H over F", ref, c32, c6, x, W1.t().contiguous())
        ref = h1.double() @ W2.double()
        c32 = torch.empty(R, D, device=dev)
        ops.fc_bwd_data(h1, W2.t().contiguous(), None, c32, R, D, H)
        c6 = torch.empty(R, D, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_F32, h13, H, W2T3, H, c6, R, D, H, workspace=ws_x3(False, R, D, H))
        row("FC2  R x D over H", ref, c32, c6, h1, W2.t().contiguous())
    ref = dz2.double() @ W2.double().t()
    c32 = torch.empty(R, H, device=dev)
    ops.fc_bwd_data(dz2, W2, None, c32, R, H, D)
    c6 = torch.empty(R, H, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_F32, dz23, D, W23, D, c6, R, H, D)
    row("dH1  R x H over D", ref, c32, c6, dz2, W2)
    for name, a, a3, pa, g, g3, pg, M, N in (("dW1  F x H over R", x, x3, F, dz1, dz13, H, F, H), ("dW2  H x D over R", h1, h13, H, dz2, dz23, D, H, D)):
        ref = a.double().t() @ g.double()
        c32, db32 = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        ws32 = torch.empty(max(ops.fc_bwd_weight_workspace(R, M, N), 16) // 4, device=dev)
        ops.fc_bwd_weight(a, g, c32, db32, ws32, R, M, N)
        c6 = torch.empty(M, N, device=dev)
        ops.gemm_bf16x3_tn(a3, pa, g3, pg, c6, M, N, R, workspace=ws_x3(True, M, N, R))
        # (split-K of the production launch at these shapes: dW1 120 tiles x 2 slabs, dW2 20 tiles x 12 slabs = 240 blocks)
        row(name, ref, c32, c6, a.t().contiguous(), g.t().contiguous(), slabs=2 if N == H else 12)
    del x3, h13, dz13, dz23, W1T3, W2T3, W23
print("# GATE (own-accumulator form, every row above): %s" % ("PASS" if gate_ok else "FAIL"))
print("# GATE (ONE accumulator, lo unscaled, the weight gradients in the production kernels' K-slabs): %s" % ("PASS" if gate_one_slabs else "FAIL"))

# subnormal handling of the fp16 MFMA (the one-accumulator form leans on it): a product of two subnormal-scaled planes
t = torch.full((64, 32), 2.0 ** -20, device=dev).half()     # fp16 subnormal (min normal 2^-14); held: the kernel reads raw pointers
u = torch.full((64, 32), 1.0, device=dev).half()
o = torch.empty(64, 64, device=dev)
lib.f16x2_gemm_nt(2, t.data_ptr(), t.data_ptr(), u.data_ptr(), u.data_ptr(), 32, 32, o.data_ptr(), 64, 64, 64, 32,
                  0.0, 1.0, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("# fp16 subnormal inputs to v_mfma_f32_16x16x32_f16: sum of 32 x (2^-20 x 1) = %.6e (exact: %.6e) -> subnormals are %s"
      % (o[0, 0].item(), 32 * 2.0 ** -20, "honoured" if abs(o[0, 0].item() - 32 * 2.0 ** -20) < 1e-12 else "FLUSHED"))

if not args.no_rate:
    print("#\n# rate: the production plane kernels on three products (general K loop) against six (resident-plane walk); same "
          "operands, event pairs, 5 rounds x 10 launches, median")
    x3, W1T3, dz13 = planes(x), planes(W1.t().contiguous()), planes(torch.randn(R, H, device=dev) * 1e-3)
    h13, W2T3, dz23, W23 = planes(h1), planes(W2.t().contiguous()), planes(torch.randn(R, D, device=dev) * 1e-3), planes(W2)
    h1o = torch.empty(R, 3 * H, dtype=torch.bfloat16, device=dev)
    gW1, gW2, z = torch.empty(F, H, device=dev), torch.empty(H, D, device=dev), torch.empty(R, D, device=dev)
    ws = torch.empty(max(ops.gemm_bf16x3_workspace(True, F, H, R, 6), ops.gemm_bf16x3_workspace(True, F, H, R, 3),
                         ops.gemm_bf16x3_workspace(False, R, D, H, 6), ops.gemm_bf16x3_workspace(True, H, D, R, 6), 16) // 4, device=dev)
    cases = []
    for q in (6, 3):
        cases += [("FC1", q, 2.0 * R * F * H, lambda q=q: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, x3, F, W1T3, F, h1o, R, H, F, plane_c=H, bias=zero_b1, products=q)),
                  ("FC2", q, 2.0 * R * H * D, lambda q=q: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, h13, H, W2T3, H, z, R, D, H, bias=zero_b2, workspace=ws, products=q)),
                  ("dH1", q, 2.0 * R * H * D, lambda q=q: ops.gemm_bf16x3_nt(ops.BE_MASK_X3, dz23, D, W23, D, h1o, R, H, D, plane_c=H, products=q)),
                  ("dW1", q, 2.0 * R * F * H, lambda q=q: ops.gemm_bf16x3_tn(x3, F, dz13, H, gW1, F, H, R, workspace=ws, products=q)),
                  ("dW2", q, 2.0 * R * H * D, lambda q=q: ops.gemm_bf16x3_tn(h13, H, dz23, D, gW2, H, D, R, workspace=ws, products=q))]
    w = torch.randn(4096, 4096, device=dev)
    for _ in range(60):
        torch.mm(w, w)
    torch.cuda.synchronize()
    times = {}
    for rnd in range(5):
        for name, q, fl, fn in cases:
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            times.setdefault((name, q), []).append(s.elapsed_time(e) / 10)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    tot6 = tot3 = 0.0
    for name, _, fl, _ in cases[:5]:
        t6, t3 = med[(name, 6)], med[(name, 3)]
        tot6 += t6
        tot3 += t3
        print("%-4s six products %8.1f us = %6.1f TF fp32-equiv (%.3f of 2.5 PF / 6)   three products %8.1f us = %6.1f TF (%.3f of 2.5 PF / 3)   x %.2f"
              % (name, t6 * 1e3, fl / t6 / 1e9, 6 * fl / (t6 * 1e-3) / 2.5e15, t3 * 1e3, fl / t3 / 1e9, 3 * fl / (t3 * 1e-3) / 2.5e15, t6 / t3))
    print("sum of the five products: %.1f us -> %.1f us (x %.2f)" % (tot6 * 1e3, tot3 * 1e3, tot6 / tot3))
