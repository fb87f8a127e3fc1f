"""Precision "f16x2": every fp32 operand of the five projection products as TWO fp16 planes hi | lo of (value * 2^s),
three plane products on v_mfma_f32_16x16x32_f16 (csrc/gemm_f16x2_256.hip, engine_f16x2.py).  The bound is the one
precision "f32x3" was admitted under (tests/test_gpu_f32x3.py): error against fp64 at most 1.5 x the fp32-MFMA
kernel's on the same operands.  Reference lines: models.py:59-61, train.py:141."""
import math

import numpy as np
import pytest
import torch

from cdml_amd import engine, engine_f16x2, engine_x3, ops, predict, train
from oracle import sampler as osampler, synth as osynth, tower as otower

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


def test_plane_writers_of_the_step_agree_with_the_split_kernel():
    """The fused sampler + gather, the loss tail and the Adam launch write their fp16 planes themselves: bit for bit what
    cdml_split_f32_f16x2 makes of the fp32 tensor the fp32 form of the same kernel writes (inputs.py:125-158, train.py:141,
    train.py:146)."""
    torch.manual_seed(5)
    dev = _dev()
    N, F, B = 3000, 1500, 128
    table = engine.FeatureTable.synthetic(N, F, 0, dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 700, 0), dtype=torch.int32, device=dev)
    Fp = 1536
    for mode, rows in ((0, 3 * B), (1, 2 * B)):
        idx_a, idx_b = torch.zeros(rows, dtype=torch.int32, device=dev), torch.zeros(rows, dtype=torch.int32, device=dev)
        sh_a, sh_b = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
        xf = torch.zeros(rows, Fp, device=dev)
        xh = torch.full((rows, 2 * Fp), 7.0, dtype=torch.float16, device=dev)
        ops.sample_gather(mode, pairs, 99, 5, B, table.data, F, idx_a, xf, shift_out=sh_a)
        ops.sample_gather(mode, pairs, 99, 5, B, table.data, F, idx_b, xh, shift_out=sh_b)
        assert torch.equal(idx_a, idx_b) and torch.equal(sh_a, sh_b)
        want = _planes(xf, Fp, engine_f16x2.X_SCALE)
        assert torch.equal(xh.view(torch.int16), want.view(torch.int16)), mode
    # the loss tail
    D = 256
    z = torch.randn(3 * B, D, device=dev)
    outs = []
    for h2 in (0.0, 2.0 ** 20):
        e, dz2 = torch.zeros(3 * B, D, device=dev), torch.zeros(3 * B, D, device=dev)
        pos, neg, hinge = (torch.zeros(B, device=dev) for _ in range(3))
        pl = torch.zeros(3 * B, 2 * D, dtype=torch.float16, device=dev) if h2 else None
        ops.vnet_tail(0, z, None, None, B, D, 0.8, e, pos, neg, hinge, dz2, dz2_bf16=pl, plane_bf=D if h2 else 0, h2_scale=h2)
        outs.append((e, dz2, hinge, pl))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[1][3].view(torch.int16), _planes(outs[1][1], D, 2.0 ** 20).view(torch.int16))
    # Adam
    K, N2 = 256, 512
    W0, g = torch.randn(K, N2, device=dev) * 0.05, torch.randn(K, N2, device=dev) * 1e-3
    res = []
    for h2 in (0.0, 2.0 ** 15):
        W, m, v = W0.clone(), torch.zeros(K, N2, device=dev), torch.zeros(K, N2, device=dev)
        wt = torch.zeros(N2, 2 * K, dtype=torch.float16 if h2 else torch.bfloat16, device=dev)
        wc = torch.zeros(K, 2 * N2, dtype=torch.float16 if h2 else torch.bfloat16, device=dev)
        if h2:
            ops.adam_matrix_bf16(W, g, m, v, 0.01, 1, wt=wt, wc=wc, plane_t=K, plane_c=N2, h2_scale=h2)
        else:
            ops.adam_matrix_bf16(W, g, m, v, 0.01, 1, wt=wt[:, :K], wc=wc[:, :N2])
        res.append((W, m, v, wt, wc))
    assert all(torch.equal(a, b) for a, b in zip(res[0][:3], res[1][:3]))
    assert torch.equal(res[1][3].view(torch.int16), _planes(res[1][0], K, 2.0 ** 15, transpose=True).view(torch.int16))
    assert torch.equal(res[1][4].view(torch.int16), _planes(res[1][0], N2, 2.0 ** 15).view(torch.int16))


def test_training_holds_the_six_plane_paths_bounds_while_the_scales_move():
    """A training run at production widths (F = 1500, H = 5000, D = 256; B = 1024, in-batch negatives, Adam at the reference's
    learning rate) on precision "f16x2".  At steps on both sides of the scale checks the step is ALSO taken on precision "f32x3"
    from the run's own weights, optimizer slots and batch, and both are held against the fp64 oracle's step from those weights:
    loss and embeddings to 1e-5, and the f16x2 gradients' error (relative L2) at most 1.5 x the six-plane path's + 1e-6 -- the
    largest single element's at most 3 x: off the first steps the error of EITHER path is dominated by the few leaky-relu'
    signs it takes differently from fp64 where a pre-activation is rounding noise, and its maximum over a tensor moves by a
    factor of two from one summation order to another.  Nothing saturates; the scales move a few times, not every step
    (models.py:59-61, train.py:141-146)."""
    dev = _dev()
    N, F, B, D = 20000, 1500, 1024, 256
    feats = osynth.features_numpy(N, F, seed=0).astype(np.float32)
    f64 = feats.astype(np.float64)
    pairs_np = osynth.cowatch_pairs(N, 6000, 0)
    table = engine.FeatureTable.from_numpy(feats, dev)
    pairs = torch.as_tensor(pairs_np, dtype=torch.int32, device=dev)
    kw = dict(margin=0.8, mode="inbatch", optimizer="adam", base_learning_rate=0.01, device=dev, gather_ahead=1)
    a = train.TrainStep(table, pairs, B, precision="f16x2", **kw)
    b = train.TrainStep(table, pairs, B, precision="f32x3", **kw)
    checks = (0, 1, 3, 33, 65)
    names = ("dW1", "db1", "dW2", "db2")
    host = lambda ts_: [t.detach().cpu().numpy().copy() for t in ts_]
    worst = {}
    for t in range(max(checks) + 1):
        if t in checks:
            W = host(a.params.unpadded())
            b.params.flat.copy_(a.params.flat)
            b.m.copy_(a.m)
            b.v.copy_(a.v)
            engine_x3.refresh_weights(b.params, b.ws)
            b.global_step = t
            b.step_dev.fill_(t)
            b.step()
            Gb, lb, eb = host(b.params.unpadded(grads=True)), b.loss(), b.ws.e[:, :D].cpu().numpy()
        a.step()
        if t in checks:
            assert torch.equal(a.idx, b.idx)
            rows, tri, valid, _ = osampler.device_inbatch(pairs_np, 1234, t, B)
            W64 = [w.astype(np.float64) for w in W]
            fwd = otower.vnet_forward(f64[rows], *W64, dtype=np.float64)
            loss = float(otower.hinge_loss_indexed(fwd["l2_norm"], tri, valid.astype(bool), 0.8, np.float64)["hinge_loss"])
            dE = otower.hinge_loss_indexed_backward(fwd["l2_norm"], tri, valid.astype(bool), 0.8, np.float64)
            wg = otower.vnet_backward(fwd, W64[2], dE, np.float64)
            Ga, la, ea = host(a.params.unpadded(grads=True)), a.loss(), a.ws.e[:, :D].cpu().numpy()
            assert abs(la - loss) < 1e-5 and abs(lb - loss) < 1e-5, (t, la, lb, loss)
            assert np.abs(ea - fwd["l2_norm"]).max() < 1e-5 and np.abs(eb - fwd["l2_norm"]).max() < 1e-5, t
            for k, ga, gb in zip(names, Ga, Gb):
                size, nrm = np.abs(wg[k]).max(), np.linalg.norm(wg[k])
                ea_, eb_ = np.linalg.norm(ga - wg[k]) / nrm, np.linalg.norm(gb - wg[k]) / nrm
                ma_, mb_ = np.abs(ga - wg[k]).max() / size, np.abs(gb - wg[k]).max() / size
                worst[(t, k)] = (float(ea_), float(eb_), float(ma_), float(mb_))
                assert ea_ <= 1.5 * eb_ + 1e-6 and ma_ <= 3.0 * mb_ + 1e-6, (t, k, worst[(t, k)], a.ws.scales.state())
            hi = a.ws.h1[:, :a.layout.Hp].float().abs().max().item()
            hg = a.ws.dz1[:, :a.layout.Hp].float().abs().max().item()
            assert 16.0 <= hi < 65504.0 and 16.0 <= hg < 65504.0, (t, hi, hg, a.ws.scales.state())
    print("gradient error against the fp64 oracle (relative L2 f16x2, f32x3; max / max |g| f16x2, f32x3):",
          {k: tuple("%.1e" % x for x in v) for k, v in worst.items()}, "scale moves:", a.ws.scales.changes, a.ws.scales.state())
    assert a.ws.scales.changes <= 12 and a.ws.scales.saturated == 0
    assert int(a.step_dev.item()) == max(checks) + 1 and math.isfinite(a.loss())


def test_resident_plane_walk_equals_the_general_loop(monkeypatch):
    """The three-product resident-plane walk (R3: every plane image of a K-tile staged once, a hand-counted DMA schedule --
    replayed on the CPU by tests/test_r6_schedule.py) against the general K loop of the same kernel (CDML_X3_WALK=general, read
    per call) on the same operands: the same products in another order inside one fp32 accumulator -- equal to accumulation
    noise -- in the k-contiguous form (fp32 and plane outputs, ragged rows), the K-slab form and the k-strided form with its
    column sums."""
    torch.manual_seed(11)
    dev = _dev()

    def both(fn):
        monkeypatch.setenv("CDML_X3_WALK", "general")
        a = fn()
        monkeypatch.delenv("CDML_X3_WALK")
        b = fn()
        return a, b
    for (M, N, K) in ((1000, 512, 384), (4096, 1024, 1536), (2048, 256, 5120)):
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1
        sa, sb = _scale(A), _scale(B)
        A2, B2 = _planes(A, K, sa), _planes(B, K, sb)
        ws = _ws(False, M, N, K)

        def nt():
            c = torch.zeros(M, N, device=dev)
            ops.gemm_f16x2_nt(ops.BE_F32, A2, K, B2, K, c, M, N, K, 1.0 / (sa * sb), workspace=ws)
            return c
        g, r = both(nt)
        assert (g - r).abs().max().item() <= 4e-7 * g.abs().max().item(), (M, N, K)
        assert not torch.equal(g, r) or K <= 384                   # (another order: it IS another kernel path)
    R, F, H = 2048, 512, 768
    x, dz = torch.randn(R, F, device=dev), torch.randn(R, H, device=dev) * 1e-3
    sx, sd = _scale(x), _scale(dz)
    x2, d2 = _planes(x, F, sx), _planes(dz, H, sd)
    ws = _ws(True, F, H, R)

    def tn():
        c, db = torch.zeros(F, H, device=dev), torch.zeros(H, device=dev)
        ops.gemm_f16x2_tn(x2, F, d2, H, c, F, H, R, 1.0 / (sx * sd), workspace=ws, colsum=db, colsum_scale=1.0 / sd)
        return torch.cat([c.flatten(), db])
    g, r = both(tn)
    assert (g[:F * H] - r[:F * H]).abs().max().item() <= 4e-7 * g[:F * H].abs().max().item()
    assert (g[F * H:] - r[F * H:]).abs().max().item() <= 4e-7 * g[F * H:].abs().max().item()
    ref = x.double().t() @ dz.double()
    assert (r[:F * H].view(F, H).double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    assert (r[F * H:].double() - dz.double().sum(0)).abs().max().item() <= 2e-6 * dz.double().sum(0).abs().max().item()


def test_catalogue_inference_against_fp64():
    """predict.Prediction(precision="f16x2") (predict.py:45-96): the forward-only tower at production widths on 3 000 catalogue
    rows in 1 000-row chunks -- scales from the weights, once per pass -- against the fp64 oracle: the fp32 paths' 1e-5."""
    dev = _dev()
    N, F = 3000, 1500
    feats = osynth.features_numpy(N, F, seed=3).astype(np.float32)
    table = engine.FeatureTable.from_numpy(feats, dev)
    L = engine_x3.layout_x3(F, 5000, 256)
    params = engine.VNetParams(L, dev, 42)
    params.b1[:5000] = torch.randn(5000, device=dev) * 0.05
    pr = predict.Prediction(params=params, precision="f16x2")
    out = pr.embed_table(table, 1000).cpu().numpy()
    W = [t.detach().cpu().numpy().astype(np.float64) for t in params.unpadded()]
    want = otower.vnet_forward(feats.astype(np.float64), *W, dtype=np.float64)["l2_norm"]
    assert np.abs(out - want).max() < 1e-5
    assert pr._ws.scales.h1 >= 2.0 ** 8 and float(pr._ws.h1[:, :L.Hp].float().abs().max()) < 65504.0


@pytest.mark.parametrize("rule", ["lars", "momentum"])
def test_lars_and_momentum_write_the_fp16_planes_with_the_update(rule):
    """The reference's own optimizer (LARS, train.py:354) and Nesterov momentum (train.py:115-116) with the GEMMs' fp16 plane
    copies written by the update itself (cdml_lars_matrix_h2 / cdml_momentum_matrix_h2): weights and slots bit-equal to the flat
    kernels, the planes bit-equal to cdml_split_f32_f16x2 of the new weights at the same scale, the step counter advanced once."""
    torch.manual_seed(11)
    dev = _dev()
    Fp, Hp, Dp = 256, 512, 256
    sizes = [Fp * Hp, Hp, Hp * Dp, Dp]
    segs, o = [], 0
    for n in sizes:
        segs.append((o, n))
        o += n
    w0 = torch.randn(o, device=dev) * 0.05
    g = torch.randn(o, device=dev) * 1e-3
    acc0 = torch.randn(o, device=dev) * 1e-4
    lr_dev = torch.full((1,), 0.5, device=dev)
    wa, acca = w0.clone(), acc0.clone()
    step_a = torch.zeros(1, dtype=torch.int64, device=dev)
    scratch = torch.zeros(max(ops.lars_scratch_floats(), ops.lars_multi_scratch_floats()), device=dev)
    if rule == "lars":
        ops.lars_multi(wa, g, acca, segs, 0.0, scratch, lr_dev=lr_dev, step_dev=step_a, tickets=ops.new_tickets(dev))
    else:
        ops.momentum_step(wa, g, acca, 0.0, 0.9, True, lr_dev=lr_dev)
        ops.step_advance(step_a)
    wb, accb = w0.clone(), acc0.clone()
    step_b = torch.zeros(1, dtype=torch.int64, device=dev)
    tick = ops.new_tickets(dev)
    s1, s2 = 2.0 ** 13, 2.0 ** 12
    W1T = torch.zeros(Hp, 2 * Fp, dtype=torch.float16, device=dev)
    W2T = torch.zeros(Dp, 2 * Hp, dtype=torch.float16, device=dev)
    W2 = torch.zeros(Hp, 2 * Dp, dtype=torch.float16, device=dev)
    if rule == "lars":
        scratch2 = torch.zeros_like(scratch)
        ops.lars_multi_norms(wb, g, segs, scratch2)
        ops.lars_matrix(wb, g, accb, segs, 0, 1, Fp, Hp, 0.0, scratch2, wt=W1T, plane_t=Fp, lr_dev=lr_dev, h2_scale=s1)
        ops.lars_matrix(wb, g, accb, segs, 2, 3, Hp, Dp, 0.0, scratch2, wt=W2T, wc=W2, plane_t=Hp, plane_c=Dp, lr_dev=lr_dev,
                        step_dev=step_b, tickets=tick, h2_scale=s2)
    else:
        v = lambda t, i, r, c: t[segs[i][0]:segs[i][0] + r * c].view(r, c)
        b = lambda i: tuple(t[segs[i][0]:segs[i][0] + segs[i][1]] for t in (wb, g, accb))
        ops.momentum_matrix(v(wb, 0, Fp, Hp), v(g, 0, Fp, Hp), v(accb, 0, Fp, Hp), 0.0, wt=W1T, plane_t=Fp, lr_dev=lr_dev,
                            bias=b(1), h2_scale=s1)
        ops.momentum_matrix(v(wb, 2, Hp, Dp), v(g, 2, Hp, Dp), v(accb, 2, Hp, Dp), 0.0, wt=W2T, wc=W2, plane_t=Hp, plane_c=Dp,
                            lr_dev=lr_dev, bias=b(3), step_dev=step_b, tickets=tick, h2_scale=s2)
    torch.cuda.synchronize()
    assert torch.equal(wa, wb) and torch.equal(acca, accb), "matrix form differs from the flat kernel"
    assert int(step_a.item()) == 1 and int(step_b.item()) == 1
    W1 = wb[:Fp * Hp].view(Fp, Hp)
    W2m = wb[segs[2][0]:segs[2][0] + Hp * Dp].view(Hp, Dp)
    i16 = lambda t: t.view(torch.int16)
    assert torch.equal(i16(W1T), i16(_planes(W1, Fp, s1, transpose=True)))
    assert torch.equal(i16(W2T), i16(_planes(W2m, Hp, s2, transpose=True))) and torch.equal(i16(W2), i16(_planes(W2m, Dp, s2)))


def test_embedding_bits_do_not_depend_on_the_chunk():
    """Catalogue inference on the fp16 planes keeps the six-plane path's property: an embedding has the same bits in a
    49 152-row chunk and in 4 096-row chunks -- the narrow layer's K-slabs partition K alone (pinned for forward-only
    workspaces), and the plane scales come from the WEIGHTS (once per pass), never from a chunk's rows."""
    dev = _dev()
    F, H, D, N = 64, 2560, 32, 49152
    L = engine_x3.layout_x3(F, H, D)
    params = engine.VNetParams(L, dev, 42)
    table = engine.FeatureTable.synthetic(N, F, seed=0, device=dev)
    big = predict.Prediction(params=params, precision="f16x2").embed_table(table, N).clone()
    small = predict.Prediction(params=params, precision="f16x2").embed_table(table, 4096)
    torch.cuda.synchronize()
    assert torch.equal(big, small)
    ref = predict.Prediction(params=params, precision="f32x3").embed_table(table, 8192)
    assert (big - ref).abs().max().item() < 2e-6
