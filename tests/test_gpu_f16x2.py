"""Precision "f16x2": every fp32 operand of the five projection products as TWO fp16 planes hi | lo of (value * 2^s),
three plane products on v_mfma_f32_16x16x32_f16 (csrc/gemm_f16x2_256.hip, engine_f16x2.py).  The bound is the one
precision "f32x3" was admitted under (tests/test_gpu_f32x3.py): error against fp64 at most 1.5 x the fp32-MFMA
kernel's on the same operands.  Reference lines: models.py:59-61, train.py:141."""
import math

import pytest
import torch

from cdml_amd import ops

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _scale(t, top=2.0 ** 10):
    """the power of two that puts max |t| in (top / 2, top]"""
    m = float(t.abs().max())
    return 2.0 ** math.floor(math.log2(top / m)) if m > 0 else 1.0


def _planes(t, width, scale, transpose=False):
    rows = t.shape[1] if transpose else t.shape[0]
    out = torch.zeros(rows, 2 * width, dtype=torch.float16, device=t.device)
    ops.split_f32_f16x2(t, out, width, scale, transpose=transpose)
    return out


def _value(p, width, scale):
    return (p[:, :width].double() + p[:, width:2 * width].double()) / scale


def _ws(tn, M, N, K):
    return torch.empty(max(ops.gemm_f16x2_workspace(tn, M, N, K), 16) // 4, device=_dev())


def test_split_planes_hold_22_bits_and_saturate():
    torch.manual_seed(0)
    dev = _dev()
    x = torch.randn(300, 512, device=dev) * torch.logspace(-6, 0, 512, device=dev)     # seven decades under one scale
    s = _scale(x)
    p = _planes(x, 512, s)
    back = _value(p, 512, s)
    # hi + lo = the scaled value to 2^-22 of ITS magnitude while lo is a normal fp16, to 2^-25 absolute (lo's subnormal grid) below
    tol = torch.maximum(x.double().abs() * 2.0 ** -21, torch.full_like(back, 2.0 ** -24 / s))
    assert ((back - x.double()).abs() <= tol).all()
    pt = _planes(x, 300, s, transpose=True)
    assert torch.equal(pt[:, :300], p[:, :512].t()) and torch.equal(pt[:, 300:], p[:, 512:].t())
    big = torch.tensor([[1e6, -1e6, 1.0, 0.0]], device=dev)
    pb = _planes(big, 4, 1.0)
    assert pb[0, 0].item() == 65504.0 and pb[0, 1].item() == -65504.0 and pb[0, 2].item() == 1.0


def test_error_at_the_five_production_shapes_against_the_fp32_mfma_kernels():
    """Every product of the step at its production shape (config 1: 8 192 rows, F = 1500 -> 1536, H = 5000 -> 5120,
    D = 256) on operands shaped like the step's -- the operands of the six-plane form's own gate test, each under ONE
    per-tensor scale -- incl. the plane-output epilogues (h1 and dz1 as fp16 planes, the sign bitmask of h1): max error
    against fp64 relative to max |result| <= 1.5 x the fp32-MFMA kernel's on the same operands + 2e-8."""
    torch.manual_seed(7)
    dev = _dev()
    R, F, H, D = 8192, 1536, 5120, 256
    x = torch.rand(R, F, device=dev)
    x[:, 1500:] = 0
    x = x / x.norm(dim=1, keepdim=True)
    W1 = (torch.rand(F, H, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5
    W2 = (torch.rand(H, D, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5
    b1, b2 = torch.randn(H, device=dev) * 0.01, torch.randn(D, device=dev) * 0.01
    lrelu = lambda t: torch.maximum(t, 0.2 * t)
    h1 = lrelu(x.double() @ W1.double() + b1.double()).float()
    dz2 = torch.randn(R, D, device=dev) * 1e-3 * torch.logspace(-5, 0, D, device=dev)       # 1e-8 .. 1e-3 under one scale
    dz1 = ((dz2.double() @ W2.double().t()) * torch.where(h1 > 0, 1.0, 0.2).double()).float()
    err = lambda got, ref: (got.double() - ref).abs().max().item() / ref.abs().max().item()
    res = {}
    sx, sw1, sw2, sh, sg2, sg1 = 2.0 ** 14, _scale(W1), _scale(W2), _scale(h1), _scale(dz2), _scale(dz1)
    x2, h12, dz12, dz22 = _planes(x, F, sx), _planes(h1, H, sh), _planes(dz1, H, sg1), _planes(dz2, D, sg2)
    W1T2, W2T2, W22 = _planes(W1, F, sw1, transpose=True), _planes(W2, H, sw2, transpose=True), _planes(W2, D, sw2)
    # FC1: fp32 output, and the planes + sign bitmask the step uses
    ref = lrelu(x.double() @ W1.double() + b1.double())
    c32 = torch.empty(R, H, device=dev)
    ops.fc_lrelu_fwd(x, W1, b1, c32, R, F, H, alpha=0.2)
    c2 = torch.empty(R, H, device=dev)
    ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_F32, x2, F, W1T2, F, c2, R, H, F, 1.0 / (sx * sw1), bias=b1, alpha=0.2)
    res["FC1"] = (err(c2, ref), err(c32, ref))
    hp = torch.zeros(R, 2 * H, dtype=torch.float16, device=dev)
    bits = torch.zeros(R, H // 8, dtype=torch.uint8, device=dev)
    ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_X3_BITS, x2, F, W1T2, F, hp, R, H, F, 1.0 / (sx * sw1), c_scale=sh, plane_c=H, bias=b1,
                      aux=bits, alpha=0.2)
    res["FC1 planes"] = (err(_value(hp, H, sh), ref), err(c32, ref))
    got_bits = torch.stack([(bits >> j) & 1 for j in range(8)], dim=2).reshape(R, H).bool()
    flips = (got_bits != (ref > 0))
    assert flips.sum().item() <= 4 and (ref.abs()[flips] < 1e-6).all()                  # a sign only where the value is rounding noise
    # FC2 (K-slabs)
    ref = lrelu(h1.double() @ W2.double() + b2.double())
    c32 = torch.empty(R, D, device=dev)
    ops.fc_lrelu_fwd(h1, W2, b2, c32, R, H, D, alpha=0.2)
    c2 = torch.empty(R, D, device=dev)
    ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_F32, h12, H, W2T2, H, c2, R, D, H, 1.0 / (sh * sw2), bias=b2, alpha=0.2, workspace=_ws(False, R, D, H))
    res["FC2"] = (err(c2, ref), err(c32, ref))
    # dH1: the product itself (fp32 output, no mask) under the gate as it stands ...
    ref = dz2.double() @ W2.double().t()
    c32 = torch.empty(R, H, device=dev)
    ops.fc_bwd_data(dz2, W2, None, c32, R, H, D)
    c2 = torch.empty(R, H, device=dev)
    ops.gemm_f16x2_nt(ops.BE_F32, dz22, D, W22, D, c2, R, H, D, 1.0 / (sg2 * sw2))
    res["dH1"] = (err(c2, ref), err(c32, ref))
    # ... and as the step runs it: times leaky-relu' of h1 from the bitmask, written as planes.  A pair of fp16 planes holds 22
    # significant bits (2^-23 of a value as its representation error) where fp32 holds 24, and this product's own error
    # is at fp32's rounding floor (K = 256): the bound for the PLANES is the gate + 2^-23
    ref = (dz2.double() @ W2.double().t()) * torch.where(got_bits, 1.0, 0.2).double()
    c32 = torch.empty(R, H, device=dev)
    ops.fc_bwd_data(dz2, W2, torch.where(got_bits, 1.0, -1.0).float(), c32, R, H, D, alpha=0.2)
    o2 = torch.zeros(R, 2 * H, dtype=torch.float16, device=dev)
    ops.gemm_f16x2_nt(ops.BE_MASKBITS_X3, dz22, D, W22, D, o2, R, H, D, 1.0 / (sg2 * sw2), c_scale=sg1, plane_c=H, aux=bits, alpha=0.2)
    res["dH1 planes"] = (err(_value(o2, H, sg1), ref), err(c32, ref))
    # dW1, dW2 (+ the bias gradients riding along)
    for name, a, a2, pa, sa, g, g2, pg, sg, M, N in (("dW1", x, x2, F, sx, dz1, dz12, H, sg1, F, H),
                                                     ("dW2", h1, h12, H, sh, dz2, dz22, D, sg2, H, D)):
        ref = a.double().t() @ g.double()
        refb = g.double().sum(0)
        c32, db32 = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        ws32 = torch.empty(max(ops.fc_bwd_weight_workspace(R, M, N), 16) // 4, device=dev)
        ops.fc_bwd_weight(a, g, c32, db32, ws32, R, M, N)
        c2, db2 = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        ops.gemm_f16x2_tn(a2, pa, g2, pg, c2, M, N, R, 1.0 / (sa * sg), workspace=_ws(True, M, N, R), colsum=db2, colsum_scale=1.0 / sg)
        res[name] = (err(c2, ref), err(c32, ref))
        res["db" + name[2]] = (err(db2, refb), err(db32, refb))
    print({k: ("%.2e" % v[0], "%.2e" % v[1]) for k, v in res.items()})
    bad = {k: v for k, v in res.items() if not v[0] <= 1.5 * v[1] + 2e-8 + (2.0 ** -23 if k.endswith("planes") else 0.0)}
    assert not bad, "two-plane fp16 error above 1.5 x the fp32-MFMA kernel's: %s (all: %s)" % (bad, res)


def test_ragged_rows_and_small_shapes():
    """M that is no multiple of 256 (the last tile's rows are masked), the smallest K, every epilogue: against fp64.  (K a multiple of 128: the walk's three steps per
    K-tile come in pairs.)"""
    torch.manual_seed(3)
    dev = _dev()
    for (M, N, K) in ((384, 256, 128), (1000, 512, 384), (128, 256, 1536)):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) * 0.1
        bias = torch.randn(N, device=dev)
        sa, sb = _scale(A), _scale(B)
        A2, B2 = _planes(A, K, sa), _planes(B, K, sb)
        ref = A.double() @ B.double().t()
        c = torch.full((M, N), 7.0, device=dev)
        ops.gemm_f16x2_nt(ops.BE_F32, A2, K, B2, K, c, M, N, K, 1.0 / (sa * sb))
        assert (c.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item(), (M, N, K)
        refl = torch.maximum(ref + bias.double(), 0.2 * (ref + bias.double()))
        sc = _scale(refl.float())
        cp = torch.zeros(M, 2 * N, dtype=torch.float16, device=dev)
        ops.gemm_f16x2_nt(ops.BE_BIAS_LRELU_X3, A2, K, B2, K, cp, M, N, K, 1.0 / (sa * sb), c_scale=sc, plane_c=N, bias=bias, alpha=0.2)
        assert (_value(cp, N, sc) - refl).abs().max().item() <= 2e-6 * refl.abs().max().item(), (M, N, K)
