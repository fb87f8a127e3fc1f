"""Split-fp32 GEMMs (precision "f32x3", csrc/gemm_bf16x3.hip): fp32 operands as three bf16 planes, six plane
products on the bf16 MFMA.  Checked against fp64 at the fp32 kernels' own error level (the reference computes
these products in fp32: models.py:59-60, train.py:141) and against the fp32 HIP kernels."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cdml_amd import ops  # noqa: E402


def _dev():
    return torch.device("cuda:0")


def _planes(x, plane):
    """reference split in torch: [rows][hi | mid | lo], `plane` columns apart"""
    hi = x.to(torch.bfloat16)
    r = x - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    out = torch.zeros(x.shape[0], 3 * plane, dtype=torch.bfloat16, device=x.device)
    for p, t in enumerate((hi, mid, lo)):
        out[:, p * plane:p * plane + x.shape[1]] = t
    return out


def _ws(tn, M, N, K, products=6):
    n = ops.gemm_bf16x3_workspace(tn, M, N, K, products)
    return torch.empty(max(n, 16) // 4, dtype=torch.float32, device=_dev())


@pytest.mark.parametrize("rows,cols,transpose", [(64, 128, False), (300, 132, False), (200, 96, True), (1500, 260, True)])
def test_split_planes_are_exact(rows, cols, transpose):
    torch.manual_seed(0)
    x = torch.randn(rows, cols, device=_dev()) * torch.logspace(-6, 3, cols, device=_dev())
    oc = rows if transpose else cols
    plane = (oc + 7) // 8 * 8 + 8
    dst = torch.zeros(cols if transpose else rows, 3 * plane, dtype=torch.bfloat16, device=_dev())
    ops.split_f32_bf16x3(x, dst, plane, transpose=transpose)
    want = _planes(x.t().contiguous() if transpose else x, plane)
    assert torch.equal(dst, want)
    s = dst[:, :oc].float() + dst[:, plane:plane + oc].float() + dst[:, 2 * plane:2 * plane + oc].float()
    assert torch.equal(s, x.t() if transpose else x), "hi + mid + lo is the fp32 value, bit for bit"


# (K = 192 / 320: three / five K-tiles -- an ODD number of the resident-plane walk's periods, whose slots alternate by parity)
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 256, 192), (300, 512, 320), (1000, 512, 384), (2048, 256, 1024),
                                   (8192, 5120, 1536)])
@pytest.mark.parametrize("products", [6, 3])
def test_gemm_x3_nt_matches_fp64_like_fp32(M, N, K, products):
    if (products * K // 64) % 2:
        K += 64      # the general loop (three products) walks K-tiles in pairs: the nearest legal contraction instead of a skip
    torch.manual_seed(1)
    dev = _dev()
    A = torch.rand(M, K, device=dev)
    A = A / A.norm(dim=1, keepdim=True)
    B = (torch.rand(N, K, device=dev) * 2 - 1) * 0.03
    bias = torch.randn(N, device=dev) * 0.01
    ref = A.double() @ B.double().t() + bias.double()
    ref = torch.maximum(ref, 0.2 * ref)
    A3, B3 = _planes(A, K), _planes(B, K)
    C = torch.empty(M, N, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, A3, K, B3, K, C, M, N, K, products=products, bias=bias, alpha=0.2,
                       workspace=_ws(False, M, N, K, products))
    scale = ref.abs().max().item()
    err = (C.double() - ref).abs().max().item() / scale
    # the fp32 MFMA kernel on the same operands: err ~ 1e-7 .. 3e-6 at these K; six products stay at that level
    c32 = torch.empty(M, N, device=dev)
    if K % 32 == 0 and N % 64 == 0:
        ops.fc_lrelu_fwd(A, B.t().contiguous(), bias, c32, M, K, N, alpha=0.2)
        err32 = (c32.double() - ref).abs().max().item() / scale
        assert err <= (3 * err32 + 1e-7 if products == 6 else 2e-5), (err, err32)
    assert err <= (5e-6 if products == 6 else 2e-5), err


def test_gemm_x3_nt_plane_outputs_and_mask():
    torch.manual_seed(2)
    dev = _dev()
    M, N, K = 1024, 512, 256
    A = torch.randn(M, K, device=dev) * 0.1
    B = torch.randn(N, K, device=dev) * 0.1
    bias = torch.randn(N, device=dev) * 0.05
    A3, B3 = _planes(A, K), _planes(B, K)
    pc = N + 64
    out = torch.zeros(M, 3 * pc, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, A3, K, B3, K, out, M, N, K, plane_c=pc, bias=bias, alpha=0.2)
    got = out[:, :N].float() + out[:, pc:pc + N].float() + out[:, 2 * pc:2 * pc + N].float()
    ref = A.double() @ B.double().t() + bias.double()
    ref = torch.maximum(ref, 0.2 * ref)
    assert (got.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # planes are a valid split: hi is the bf16 rounding of the sum
    assert torch.equal(out[:, :N], got.to(torch.bfloat16))
    assert torch.all(out[:, N:pc] == 0) and torch.all(out[:, pc + N:2 * pc] == 0), "gaps between planes untouched"
    # epilogue 7: times the leaky-relu derivative taken from the hi plane of another activation
    h = torch.randn(M, N, device=dev)
    h3 = _planes(h, pc)
    out2 = torch.zeros(M, 3 * pc, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_MASK_X3, A3, K, B3, K, out2, M, N, K, plane_c=pc, aux=h3, alpha=0.2)
    got2 = out2[:, :N].float() + out2[:, pc:pc + N].float() + out2[:, 2 * pc:2 * pc + N].float()
    ref2 = (A.double() @ B.double().t()) * torch.where(h > 0, 1.0, 0.2).double()
    assert (got2.double() - ref2).abs().max().item() <= 2e-6 * ref2.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (512, 256, 1024), (1536, 5120, 8192), (5120, 256, 8192)])
@pytest.mark.parametrize("products", [6, 3])
def test_gemm_x3_tn_weight_gradient(M, N, K, products):
    torch.manual_seed(3)
    dev = _dev()
    X = torch.rand(K, M, device=dev)
    X = X / X.norm(dim=1, keepdim=True)
    dY = torch.randn(K, N, device=dev) * 1e-3
    X3, dY3 = _planes(X, M), _planes(dY, N)
    C = torch.empty(M, N, device=dev)
    db = torch.empty(N, device=dev)
    ops.gemm_bf16x3_tn(X3, M, dY3, N, C, M, N, K, products=products, workspace=_ws(True, M, N, K, products), colsum=db)
    ref = X.double().t() @ dY.double()
    err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= (5e-6 if products == 6 else 3e-5), err
    refb = dY.double().sum(0)
    errb = (db.double() - refb).abs().max().item() / refb.abs().max().item()
    assert errb <= (5e-6 if products == 6 else 1e-4), errb


@pytest.mark.parametrize("mode", [0, 1])
def test_sample_gather_planes_equal_split_of_the_fp32_gather(mode):
    """the fused sampler + gather writing planes == its fp32 rows, split (ids bit-equal, planes bit-equal)"""
    dev = _dev()
    from cdml_amd import engine
    N, F, B, K = 5000, 1500, 96, 3
    table = engine.FeatureTable.synthetic(N, F, 0, dev)
    rng = np.random.RandomState(0)
    pairs = rng.randint(0, N, size=(4000, 2)).astype(np.int32)
    pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
    rows, Fp = B * (3 if mode == 0 else 2), 1536
    x32 = torch.zeros(K, rows, Fp, device=dev)
    x3 = torch.zeros(K, rows, 3 * Fp, dtype=torch.bfloat16, device=dev)
    i32, i3 = (torch.zeros(K, rows, dtype=torch.int32, device=dev) for _ in range(2))
    s32, s3 = (torch.zeros(K, dtype=torch.int32, device=dev) for _ in range(2))
    ops.sample_gather(mode, pairs, 1234, 7, B, table.data, F, i32, x32, shift_out=s32, n_steps=K)
    ops.sample_gather(mode, pairs, 1234, 7, B, table.data, F, i3, x3, shift_out=s3, n_steps=K)
    assert torch.equal(i32, i3) and torch.equal(s32, s3)
    for k in range(K):
        assert torch.equal(x3[k], _planes(x32[k], Fp))


@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("mode,optimizer", [("uniform", "adam"), ("inbatch", "adam"), ("semihard", "adam"),
                                            ("uniform", "lars"), ("inbatch", "momentum")])
def test_train_step_matches_the_fp32_mfma_step(mode, optimizer, transposed, monkeypatch):
    """TrainStep(precision="f32x3") against TrainStep(precision="f32") from the same seeds: same triplets, embeddings
    and loss of the first step within 1e-5 (both are within 1e-5 of fp64: test_train_steps_config0), every sampler mode
    and optimizer (semi-hard mining picks its negatives from the embeddings: the same ones)."""
    dev = _dev()
    from cdml_amd import engine, train
    # (transposed: the hidden activations held as h1^T / dz1^T -- engine_x3.TowerWorkspaceX3; rows % 256 == 0)
    monkeypatch.setenv("CDML_X3_TRANSPOSED", "1" if transposed else "0")
    N, F, B = 6000, 500, 256 if transposed else 128
    table = engine.FeatureTable.synthetic(N, F, 0, dev)
    rng = np.random.RandomState(1)
    pairs = rng.randint(0, N, size=(3000, 2)).astype(np.int32)
    pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
    lr = {"adam": 0.01, "lars": 1.0, "momentum": 0.01}[optimizer]
    mk = lambda prec: train.TrainStep(table, pairs, B, hidden_size=700, output_size=256, mode=mode, optimizer=optimizer,
                                      base_learning_rate=lr, device=dev, precision=prec)
    a, b = mk("f32"), mk("f32x3")
    assert b.ws.transposed == transposed
    a.step(); b.step()
    torch.cuda.synchronize()
    D = 256
    assert torch.equal(a.idx, b.idx)
    assert (a.ws.e[:, :D] - b.ws.e[:, :D]).abs().max().item() < 1e-5
    if mode == "semihard":
        # round 5: the f32x3 step mines in the epilogue of its own score product (plane kernels), the fp32 step scans the
        # score matrix of the fp32-MFMA GEMM: the two sets of distances differ by rounding, and on this iid catalogue the
        # untrained embeddings lie within ~1e-3 of each other, so near-ties may break differently.  Each path's choice is
        # held to the rule under fp64 distances of ITS OWN embeddings (2e-6, as tests/test_gpu_parity.py), and its loss to
        # the oracle's on its own triplets.
        from oracle import tower as otower
        assert b.mine_fused and not a.mine_fused
        for t in (a, b):
            E = t.ws.e[:, :D].double().cpu().numpy()
            rows = t.idx.cpu().numpy()
            got = t.neg_row.cpu().numpy()
            want, dist = otower.semihard_select(E, rows)
            d_p = dist[np.arange(B), 2 * np.arange(B) + 1]
            for i in range(B):
                if want[i] < 0:
                    assert got[i] == -1
                    continue
                elig = (rows != rows[2 * i]) & (rows != rows[2 * i + 1])
                assert got[i] >= 0 and elig[got[i]]
                strict = elig & (dist[i] > d_p[i] + 2e-6)
                if dist[i, got[i]] > d_p[i] - 2e-6 and (strict.any() or dist[i, want[i]] > d_p[i]):
                    if strict.any():
                        assert dist[i, got[i]] <= dist[i][strict].min() + 2e-6
                else:
                    assert dist[i, got[i]] >= dist[i][elig].max() - 2e-6
            tri, valid = otower.semihard_triplets(got)
            assert abs(t.loss() - float(otower.hinge_loss_indexed(E, tri, valid, 0.8, np.float64)["hinge_loss"])) < 1e-5
    else:
        assert abs(a.loss() - b.loss()) < 1e-5
    for _ in range(3):                                   # keeps stepping: the plane copies follow the optimizer
        a.step(); b.step()
    assert abs(a.loss() - b.loss()) < (5e-3 if mode == "semihard" else 2e-3)
    # the planes the GEMMs read ARE the master weights
    L = b.layout
    w1t = b.ws.W1T[:, :L.Fp].float() + b.ws.W1T[:, L.Fp:2 * L.Fp].float() + b.ws.W1T[:, 2 * L.Fp:].float()
    assert torch.equal(w1t, b.params.W1.t())
    w2 = b.ws.W2[:, :L.Dp].float() + b.ws.W2[:, L.Dp:2 * L.Dp].float() + b.ws.W2[:, 2 * L.Dp:].float()
    assert torch.equal(w2, b.params.W2)


@pytest.mark.parametrize("mode", [0, 1])
def test_vnet_tail_writes_the_planes_of_dz2(mode):
    """the fused tail's plane output == the split of the fp32 dz2 it writes beside it (bit for bit)"""
    dev = _dev()
    torch.manual_seed(5)
    B, D = 96, 256
    rows = B * (3 if mode == 0 else 2)
    z = torch.randn(rows, D, device=dev)
    e, dz2 = torch.zeros(rows, D, device=dev), torch.zeros(rows, D, device=dev)
    pos, neg, hinge = (torch.zeros(B, device=dev) for _ in range(3))
    idx = torch.arange(rows, dtype=torch.int32, device=dev)
    shift = torch.tensor([7], dtype=torch.int32, device=dev)
    planes = torch.zeros(rows, 3 * D, dtype=torch.bfloat16, device=dev)
    ops.vnet_tail(mode, z, idx, shift, B, D, 0.8, e, pos, neg, hinge, dz2, dz2_bf16=planes, plane_bf=D)
    assert float(dz2.abs().max()) > 0
    assert torch.equal(planes, _planes(dz2, D))


def test_gemm_x3_nt_row_bias_planes_and_colsum():
    """the two pieces the transposed activation layout adds to the k-contiguous form: epilogue 8 (bias indexed by
    the output row) and colsum[n] = sum_k B[n][k] (the bias gradient when B is a transposed activation gradient)"""
    torch.manual_seed(7)
    dev = _dev()
    M, N, K = 512, 768, 1024
    A = torch.randn(M, K, device=dev) * 0.05
    B = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(M, device=dev) * 0.1
    A3, B3 = _planes(A, K), _planes(B, K)
    pc = N + 8
    out = torch.zeros(M, 3 * pc, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_ROWBIAS_LRELU_X3, A3, K, B3, K, out, M, N, K, plane_c=pc, bias=bias, alpha=0.2)
    got = out[:, :N].float() + out[:, pc:pc + N].float() + out[:, 2 * pc:2 * pc + N].float()
    ref = A.double() @ B.double().t() + bias.double()[:, None]
    ref = torch.maximum(ref, 0.2 * ref)
    assert (got.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()
    C = torch.empty(M, N, device=dev)
    cs = torch.empty(N, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_F32, A3, K, B3, K, C, M, N, K, workspace=_ws(False, M, N, K), colsum=cs)
    ref2 = A.double() @ B.double().t()
    assert (C.double() - ref2).abs().max().item() <= 3e-6 * ref2.abs().max().item()
    refc = B.double().sum(1)
    assert (cs.double() - refc).abs().max().item() <= 5e-6 * refc.abs().max().item()


def test_gemm_x3_tn_with_bias():
    torch.manual_seed(8)
    dev = _dev()
    M, N, K = 512, 256, 1024
    X = torch.randn(K, M, device=dev) * 0.05
    W = torch.randn(K, N, device=dev) * 0.05
    bias = torch.randn(N, device=dev) * 0.1
    C = torch.empty(M, N, device=dev)
    ops.gemm_bf16x3_tn(_planes(X, M), M, _planes(W, N), N, C, M, N, K, workspace=_ws(True, M, N, K), bias=bias, alpha=0.2)
    ref = X.double().t() @ W.double() + bias.double()
    ref = torch.maximum(ref, 0.2 * ref)
    assert (C.double() - ref).abs().max().item() <= 3e-6 * ref.abs().max().item()


def test_error_at_the_five_production_shapes_against_the_fp32_mfma_kernels():
    """Every product of the step at its production shape (config 1: 8 192 rows, F = 1500 -> 1536, H = 5000 -> 5120,
    D = 256) on operands shaped like the step's: max error against fp64, relative to max |result|, of the six-plane
    form <= 1.5 x the fp32-MFMA kernel's on the SAME operands (plus 2e-8 for products whose fp32 error is at the rounding
    floor).  The ruling under which precision "f32x3" is the headline path (models.py:59-60, train.py:141)."""
    torch.manual_seed(7)
    dev = _dev()
    R, F, H, D = 8192, 1536, 5120, 256
    x = torch.rand(R, F, device=dev)
    x[:, 1500:] = 0
    x = x / x.norm(dim=1, keepdim=True)                              # unit rows, as the gather hands them over
    W1 = (torch.rand(F, H, device=dev) * 2 - 1) * (6.0 / 6500) ** 0.5  # Xavier-uniform (models.py:26-30)
    W2 = (torch.rand(H, D, device=dev) * 2 - 1) * (6.0 / 5256) ** 0.5
    b1, b2 = torch.randn(H, device=dev) * 0.01, torch.randn(D, device=dev) * 0.01
    lrelu = lambda t: torch.maximum(t, 0.2 * t)
    h1 = lrelu(x.double() @ W1.double() + b1.double()).float()
    dz2 = torch.randn(R, D, device=dev) * 1e-3
    dz1 = ((dz2.double() @ W2.double().t()) * torch.where(h1 > 0, 1.0, 0.2).double()).float()
    err = lambda got, ref: (got.double() - ref).abs().max().item() / ref.abs().max().item()
    res = {}
    x3, h13, dz13, dz23 = _planes(x, F), _planes(h1, H), _planes(dz1, H), _planes(dz2, D)
    W1T3, W2T3, W23 = _planes(W1.t().contiguous(), F), _planes(W2.t().contiguous(), H), _planes(W2, D)
    # FC1
    ref = lrelu(x.double() @ W1.double() + b1.double())
    c32 = torch.empty(R, H, device=dev)
    ops.fc_lrelu_fwd(x, W1, b1, c32, R, F, H, alpha=0.2)
    c3 = torch.empty(R, H, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, x3, F, W1T3, F, c3, R, H, F, bias=b1, alpha=0.2)
    res["FC1"] = (err(c3, ref), err(c32, ref))
    # FC2
    ref = lrelu(h1.double() @ W2.double() + b2.double())
    c32 = torch.empty(R, D, device=dev)
    ops.fc_lrelu_fwd(h1, W2, b2, c32, R, H, D, alpha=0.2)
    c3 = torch.empty(R, D, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, h13, H, W2T3, H, c3, R, D, H, bias=b2, alpha=0.2, workspace=_ws(False, R, D, H))
    res["FC2"] = (err(c3, ref), err(c32, ref))
    # dH1 (times leaky-relu' of h1)
    ref = (dz2.double() @ W2.double().t()) * torch.where(h1 > 0, 1.0, 0.2).double()
    c32 = torch.empty(R, H, device=dev)
    ops.fc_bwd_data(dz2, W2, h1, c32, R, H, D, alpha=0.2)
    o3 = torch.zeros(R, 3 * H, dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_MASK_X3, dz23, D, W23, D, o3, R, H, D, plane_c=H, aux=h13, alpha=0.2)
    c3 = o3[:, :H].float() + o3[:, H:2 * H].float() + o3[:, 2 * H:].float()
    res["dH1"] = (err(c3, ref), err(c32, ref))
    # dW1, dW2 (+ the bias gradients riding along)
    for name, a, a3, pa, g, g3, pg, M, N in (("dW1", x, x3, F, dz1, dz13, H, F, H), ("dW2", h1, h13, H, dz2, dz23, D, H, D)):
        ref = a.double().t() @ g.double()
        refb = g.double().sum(0)
        c32, db32 = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        ws32 = torch.empty(max(ops.fc_bwd_weight_workspace(R, M, N), 16) // 4, device=dev)
        ops.fc_bwd_weight(a, g, c32, db32, ws32, R, M, N)
        c3, db3 = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
        ops.gemm_bf16x3_tn(a3, pa, g3, pg, c3, M, N, R, workspace=_ws(True, M, N, R), colsum=db3)
        res[name] = (err(c3, ref), err(c32, ref))
        res["db" + name[2]] = (err(db3, refb), err(db32, refb))
    bad = {k: v for k, v in res.items() if not v[0] <= 1.5 * v[1] + 2e-8}
    assert not bad, "six-plane error above 1.5 x the fp32-MFMA kernel's: %s (all: %s)" % (bad, res)


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("rule", ["lars", "momentum"])
def test_lars_and_momentum_write_the_operand_copies_with_the_update(rule, planes):
    """The reference's own optimizer (LARS, train.py:354) and Nesterov momentum (train.py:115-116) on the weight
    matrices with the GEMMs' operand copies written by the update itself (cdml_lars_matrix / cdml_momentum_matrix):
    weights and slots bit-equal to the flat kernels (cdml_lars_multi / cdml_momentum_step), the three planes summing to
    the new fp32 weights exactly (planes = 3), the bf16 copies their rounding (planes = 1), the step counter advanced
    once by the last launch."""
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
    # flat kernels
    wa, acca = w0.clone(), acc0.clone()
    step_a = torch.zeros(1, dtype=torch.int64, device=dev)
    scratch = torch.zeros(max(ops.lars_scratch_floats(), ops.lars_multi_scratch_floats()), device=dev)
    if rule == "lars":
        ops.lars_multi(wa, g, acca, segs, 0.0, scratch, lr_dev=lr_dev, step_dev=step_a, tickets=ops.new_tickets(dev))
    else:
        ops.momentum_step(wa, g, acca, 0.0, 0.9, True, lr_dev=lr_dev)
        ops.step_advance(step_a)
    # matrix kernels
    wb, accb = w0.clone(), acc0.clone()
    step_b = torch.zeros(1, dtype=torch.int64, device=dev)
    tick = ops.new_tickets(dev)
    mult = 3 if planes == 3 else 1
    W1T = torch.zeros(Hp, mult * Fp, dtype=torch.bfloat16, device=dev)
    W2T = torch.zeros(Dp, mult * Hp, dtype=torch.bfloat16, device=dev)
    W2 = torch.zeros(Hp, mult * Dp, dtype=torch.bfloat16, device=dev)
    pt1, pt2, pc2 = (Fp, Hp, Dp) if planes == 3 else (0, 0, 0)
    if rule == "lars":
        scratch2 = torch.zeros_like(scratch)
        ops.lars_multi_norms(wb, g, segs, scratch2)
        ops.lars_matrix(wb, g, accb, segs, 0, 1, Fp, Hp, 0.0, scratch2, wt=W1T, plane_t=pt1, lr_dev=lr_dev)
        ops.lars_matrix(wb, g, accb, segs, 2, 3, Hp, Dp, 0.0, scratch2, wt=W2T, wc=W2, plane_t=pt2, plane_c=pc2, lr_dev=lr_dev,
                        step_dev=step_b, tickets=tick)
    else:
        v = lambda t, i, r, c: t[segs[i][0]:segs[i][0] + r * c].view(r, c)
        b = lambda i: tuple(t[segs[i][0]:segs[i][0] + segs[i][1]] for t in (wb, g, accb))
        ops.momentum_matrix(v(wb, 0, Fp, Hp), v(g, 0, Fp, Hp), v(accb, 0, Fp, Hp), 0.0, wt=W1T, plane_t=pt1, lr_dev=lr_dev,
                            bias=b(1))
        ops.momentum_matrix(v(wb, 2, Hp, Dp), v(g, 2, Hp, Dp), v(accb, 2, Hp, Dp), 0.0, wt=W2T, wc=W2, plane_t=pt2, plane_c=pc2,
                            lr_dev=lr_dev, bias=b(3), step_dev=step_b, tickets=tick)
    torch.cuda.synchronize()
    assert torch.equal(wa, wb) and torch.equal(acca, accb), "matrix form differs from the flat kernel"
    assert not torch.equal(wb, w0)
    assert int(step_a.item()) == 1 and int(step_b.item()) == 1
    W1 = wb[:Fp * Hp].view(Fp, Hp)
    W2m = wb[segs[2][0]:segs[2][0] + Hp * Dp].view(Hp, Dp)
    if planes == 3:
        s3 = lambda t, p: t[:, :p].float() + t[:, p:2 * p].float() + t[:, 2 * p:3 * p].float()
        assert torch.equal(s3(W1T, Fp), W1.t()) and torch.equal(s3(W2T, Hp), W2m.t()) and torch.equal(s3(W2, Dp), W2m)
        assert torch.equal(W1T[:, :Fp], W1.t().to(torch.bfloat16))
    else:
        assert torch.equal(W1T, W1.t().to(torch.bfloat16)) and torch.equal(W2T, W2m.t().to(torch.bfloat16))
        assert torch.equal(W2, W2m.to(torch.bfloat16))


def test_sign_bitmask_epilogues_equal_the_value_mask():
    """Epilogue 9 (= 6 + the sign bitmask of its result) and epilogue 10 (= 7 reading that bitmask): the planes out of
    9 are those of 6 bit for bit, bit j of byte b of row r is C[r][8 b + j] > 0, and 10 gives what 7 gives from the hi
    plane -- leaky-relu' of the hidden layer carried as one bit per element instead of two bytes (models.py:59, train.py:141)."""
    torch.manual_seed(5)
    dev = _dev()
    M, N, K = 1000, 512, 256                                  # a ragged last row tile
    A = torch.randn(M, K, device=dev) * 0.1
    B = torch.randn(N, K, device=dev) * 0.1
    bias = torch.randn(N, device=dev) * 0.05
    A3, B3 = _planes(A, K), _planes(B, K)
    o6 = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    o9 = torch.zeros_like(o6)
    bits = torch.full((M, N // 8), 0xAA, dtype=torch.uint8, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3, A3, K, B3, K, o6, M, N, K, plane_c=N, bias=bias, alpha=0.2)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, A3, K, B3, K, o9, M, N, K, plane_c=N, bias=bias, alpha=0.2, aux=bits)
    assert torch.equal(o6, o9)
    h = o9[:, :N].float() + o9[:, N:2 * N].float() + o9[:, 2 * N:].float()
    want = (h > 0).view(M, N // 8, 8).to(torch.int32)
    want = (want << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want)
    # the data gradient with the mask as values (hi plane) and as bits
    G = torch.randn(M, 256, device=dev) * 1e-3
    W = torch.randn(N, 256, device=dev) * 0.05
    G3, W3 = _planes(G, 256), _planes(W, 256)
    d7 = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
    d10 = torch.zeros_like(d7)
    ops.gemm_bf16x3_nt(ops.BE_MASK_X3, G3, 256, W3, 256, d7, M, N, 256, plane_c=N, aux=o9, alpha=0.2)
    ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, G3, 256, W3, 256, d10, M, N, 256, plane_c=N, aux=bits, alpha=0.2)
    # (the hi plane of a tiny positive fp32 value can round to +0: the bit form takes the sign of the fp32 value itself)
    same = (h > 0) == (o9[:, :N].float() > 0)
    assert float(same.float().mean()) > 0.9999
    v7 = d7[:, :N].float() + d7[:, N:2 * N].float() + d7[:, 2 * N:].float()
    v10 = d10[:, :N].float() + d10[:, N:2 * N].float() + d10[:, 2 * N:].float()
    assert torch.equal(v7[same], v10[same])


def test_resident_plane_walk_race_screen():
    """The resident-plane walk (R6) orders its LDS-DMA against the fragment reads by counted waits and barriers only (the
    schedule itself is replayed on the CPU in tests/test_r6_schedule.py); a misplaced wait would show up as rare wrong
    tiles.  Both forms, the full chip, many launches back to back with other launches' traffic in between, every result
    compared bit for bit with the first (which is checked against fp64)."""
    torch.manual_seed(9)
    dev = _dev()
    # k-contiguous form: 256 tiles (one per CU), 16 K-tiles = 16 periods each; odd period count in the second shape
    for (M, N, K) in ((4096, 4096, 1024), (2048, 2048, 448)):
        A = torch.randn(M, K, device=dev) / 32
        B = torch.randn(N, K, device=dev)
        A3, B3 = _planes(A, K), _planes(B, K)
        out = torch.empty(M, N, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_F32, A3, K, B3, K, out, M, N, K)
        ref = A.double() @ B.double().t()
        assert (out.double() - ref).abs().max().item() <= 5e-6 * ref.abs().max().item()
        first = out.clone()
        for _ in range(40):
            ops.gemm_bf16x3_nt(ops.BE_F32, A3, K, B3, K, out, M, N, K)
            assert torch.equal(out, first)
    # k-strided form with the bias gradient riding along: 6 x 20 tiles x 2 K-halves (the weight gradient's launch shape)
    M, N, K = 1536, 5120, 2048
    X = torch.randn(K, M, device=dev) / 16
    dY = torch.randn(K, N, device=dev) * 1e-2
    X3, dY3 = _planes(X, M), _planes(dY, N)
    C, db = torch.empty(M, N, device=dev), torch.empty(N, device=dev)
    ws = _ws(True, M, N, K)
    ops.gemm_bf16x3_tn(X3, M, dY3, N, C, M, N, K, workspace=ws, colsum=db)
    ref = X.double().t() @ dY.double()
    assert (C.double() - ref).abs().max().item() <= 5e-6 * ref.abs().max().item()
    c0, d0 = C.clone(), db.clone()
    for _ in range(40):
        ops.gemm_bf16x3_tn(X3, M, dY3, N, C, M, N, K, workspace=ws, colsum=db)
        assert torch.equal(C, c0) and torch.equal(db, d0)


@pytest.mark.parametrize("M,N,K", [(1000, 512, 192),       # 8 tiles: half tiles only, a ragged last row tile, three K-tiles (odd)
                                   (2816, 6144, 256),      # 264 tiles: one full round + 8 tiles as 16 halves
                                   (2700, 6144, 128),      # the same grid with rows ending inside the second half of a tile
                                   (8192, 5120, 1536)])    # FC1 at config 1: 640 tiles = 2 rounds + 128 tiles as 256 halves
def test_half_tiles_of_the_last_round_are_bit_identical(M, N, K, monkeypatch):
    """The last, partly filled round of a plane-output product runs as 128 x 256 half tiles on twice the CUs
    (csrc/gemm_bf16_256.hip, NARROW): same operands and the same K order per output element, so the planes, the sign bits and
    the masked gradient are those of the full-tile launch bit for bit; launched back to back many times (a misplaced wait of
    the half tile's own DMA schedule would show as rare wrong tiles; the schedule itself: tests/test_r6_schedule.py)."""
    torch.manual_seed(11)
    dev = _dev()
    A = torch.randn(M, K, device=dev) * 0.1
    B = torch.randn(N, K, device=dev) * 0.1
    bias = torch.randn(N, device=dev) * 0.05
    A3, B3 = _planes(A, K), _planes(B, K)

    def run(half):
        monkeypatch.setenv("CDML_X3_HALFTILES", "1" if half else "0")
        o = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
        bits = torch.zeros(M, N // 8, dtype=torch.uint8, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_X3_BITS, A3, K, B3, K, o, M, N, K, plane_c=N, bias=bias, alpha=0.2, aux=bits)
        d = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3, A3, K, B3, K, d, M, N, K, plane_c=N, aux=bits, alpha=0.2)
        dv = torch.zeros(M, 3 * N, dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_MASK_X3, A3, K, B3, K, dv, M, N, K, plane_c=N, aux=o, alpha=0.2)
        # round 6: the unsplit fp32-output product too (the trainable table's row gradient dz1 . W1^T)
        c = torch.zeros(M, N, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_F32, A3, K, B3, K, c, M, N, K)
        return o, bits, d, dv, c

    full = run(False)
    half = run(True)
    for f, h in zip(full, half):
        assert torch.equal(f, h)
    got = half[0][:, :N].float() + half[0][:, N:2 * N].float() + half[0][:, 2 * N:].float()
    ref = A.double() @ B.double().t() + bias.double()
    ref = torch.maximum(ref, 0.2 * ref)
    assert (got.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    for _ in range(10 if M * N > 2 ** 24 else 30):
        again = run(True)
        assert torch.equal(again[0], half[0]) and torch.equal(again[2], half[2])


def test_embedding_bits_do_not_depend_on_the_chunk():
    """the narrow second layer's K partition is a function of K alone, so an embedding has the same bits in a
    49 152-row inference chunk (192 row tiles: round 4 skipped the slabs there) and in 4 096-row chunks; the single pass
    over K is an explicit opt-out (fc2_single_pass) whose results agree to rounding, and a workspace that is given but too
    small is an error at the C ABI, not a silent change of arithmetic."""
    from cdml_amd import _lib, engine, engine_x3, predict
    dev = _dev()
    F, H, D, N = 64, 2560, 32, 49152                       # Hp = 2560: 240 K-tile steps = two slabs of 120
    L = engine_x3.layout_x3(F, H, D)
    params = engine.VNetParams(L, dev, 42)
    table = engine.FeatureTable.synthetic(N, F, seed=0, device=dev)
    pr = predict.Prediction(params=params, precision="f32x3")
    big = pr.embed_table(table, N).clone()
    assert pr._ws.R // 256 >= 192 and not pr._ws.fc2_single_pass
    small = predict.Prediction(params=params, precision="f32x3").embed_table(table, 4096)
    torch.cuda.synchronize()
    assert torch.equal(big, small)
    one = predict.Prediction(params=params, precision="f32x3", fc2_single_pass=True)
    single = one.embed_table(table, N)
    assert one._ws.fc2_single_pass
    assert 0 < (single - big).abs().max().item() < 5e-6     # (another association of the same sums: close, not equal)
    # the C ABI: NULL workspace = the single pass; a workspace that is too small is refused
    M, K = 512, 2560
    A3 = _planes(torch.randn(M, K, device=dev) * 0.1, K)
    B3 = _planes(torch.randn(256, K, device=dev) * 0.1, K)
    C = torch.empty((M, 256), device=dev)
    bias = torch.zeros(256, device=dev)
    ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, A3, K, B3, K, C, M, 256, K, bias=bias, alpha=0.2)           # single pass
    with pytest.raises(_lib.CdmlError, match="slab form needs"):
        ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, A3, K, B3, K, C, M, 256, K, bias=bias, alpha=0.2,
                           workspace=torch.empty(1024, device=dev))
    # round 6: the slab LENGTH follows the row-tile class -- within a class a batch gives the same bits whole or in row
    # blocks; a pin (what the inference workspaces above use) holds ONE length across classes
    ws = torch.empty(max(ops.gemm_bf16x3_workspace(False, 16640, 256, K, 6), 16) // 4, device=dev)
    run = lambda A_, M_, **kw: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, A_, K, B3, K, torch.empty((M_, 256), device=dev), M_, 256,
                                                  K, bias=bias, alpha=0.2, workspace=ws, **kw)
    whole = run(A3, M)                                       # 2 row tiles x 2 slabs of 120 <= 64 tiles: the 60-step class
    halves = torch.cat([run(A3[:256], 256), run(A3[256:], 256)])
    assert torch.equal(whole, halves)
    pinned = run(A3, M, slab_steps=120)
    assert 0 < (pinned - whole).abs().max().item() < 5e-6    # two slabs against four: another association of the same sums
    assert torch.equal(run(A3, M, slab_steps=60), whole)     # the class's own length, pinned
    Abig = _planes(torch.randn(16640, K, device=dev) * 0.1, K)      # 65 row tiles x 2 slabs > 64: the 120-step class
    assert torch.equal(run(Abig, 16640), run(Abig, 16640, slab_steps=120))
    assert _lib.load_library().cdml_x3_slab_steps(0) == 0    # no pin is left behind


def test_split_k_geometry_has_no_empty_split():
    """rounding the K-tiles per split up to whole walk periods can leave the last requested split empty; the
    launch, the slab sizes and the workspace query now use the splits that carry K-tiles (x3_split_geometry).  A one-tile
    product over K = 3 072 rows (288 K-tile steps, split for occupancy) against fp64, with the bias-gradient sums."""
    dev = _dev()
    M, N, K = 256, 256, 3072
    g = torch.Generator(device=dev); g.manual_seed(5)
    A = torch.randn(K, M, device=dev, generator=g) * 0.1
    B = torch.randn(K, N, device=dev, generator=g) * 0.1
    A3, B3 = _planes(A, M), _planes(B, N)
    nbytes = ops.gemm_bf16x3_workspace(True, M, N, K, 6)
    C = torch.empty((M, N), device=dev)
    cs = torch.empty(N, device=dev)
    ops.gemm_bf16x3_tn(A3, M, B3, N, C, M, N, K, workspace=torch.empty(max(nbytes, 16) // 4, device=dev), colsum=cs)
    torch.cuda.synchronize()
    want = (A.double().t() @ B.double())
    assert ((C.double() - want).abs().max() / want.abs().max()).item() < 5e-6
    assert (cs.double() - B.double().sum(0)).abs().max().item() < 1e-4


@pytest.mark.parametrize("M,N,K,a0,b0", [(256, 512, 256, 0, 0), (512, 256, 1024, 256, 0), (256, 256, 3072, 0, 256)])
def test_gemm_x3_tnk_equals_tn(M, N, K, a0, b0):
    """The weight-gradient product on k8-interleaved operands ([plane][k / 8][column][8 k]: one aligned 16-B LDS read per
    fragment instead of two transposed reads) is the k-strided product -- same images, slots, DMA schedule and accumulation
    order: bit-identical results and column sums, also for a column window of wider operands (the data-parallel step's row
    blocks of W1)."""
    dev = _dev()
    g = torch.Generator(device=dev); g.manual_seed(M + N + K)
    ma, nb = M + a0 + 256, N + b0
    A = torch.randn(K, ma, device=dev, generator=g) * 0.05
    B = torch.randn(K, nb, device=dev, generator=g) * 0.02
    A3, B3 = _planes(A, ma), _planes(B, nb)
    Ai = torch.empty(3 * K * ma, dtype=torch.bfloat16, device=dev)
    Bi = torch.empty(3 * K * nb, dtype=torch.bfloat16, device=dev)
    ops.interleave8_bf16x3(A3, ma, K, ma, Ai)
    ops.interleave8_bf16x3(B3, nb, K, nb, Bi)
    # the interleave itself: element (plane p, row r, column c) at ((p * K/8 + r/8) * cols + c) * 8 + r % 8
    v = Ai.view(3, K // 8, ma, 8)
    assert torch.equal(v.permute(0, 1, 3, 2).reshape(3, K, ma)[1], A3[:, ma:2 * ma])
    ws = _ws(True, M, N, K)
    C1, C2 = torch.empty((M, N), device=dev), torch.full((M, N), float("nan"), device=dev)
    cs1, cs2 = torch.empty(N, device=dev), torch.full((N,), float("nan"), device=dev)
    ops.gemm_bf16x3_tn(A3[:, a0:], ma, B3[:, b0:], nb, C1, M, N, K, workspace=ws, colsum=cs1)
    ops.gemm_bf16x3_tnk(Ai, ma, a0, Bi, nb, b0, C2, M, N, K, workspace=ws, colsum=cs2)
    torch.cuda.synchronize()
    assert torch.equal(C1, C2) and torch.equal(cs1, cs2)
    want = A[:, a0:a0 + M].double().t() @ B[:, b0:b0 + N].double()
    assert ((C2.double() - want).abs().max() / want.abs().max()).item() < 5e-6


@pytest.mark.parametrize("M,N", [(640, 512), (6144, 4096)])
def test_data_gradient_epilogue_writes_interleaved_planes(M, N):
    """Epilogue 12 of cdml_gemm_bf16x3_nt = epilogue 10 (times leaky-relu' from the sign bitmask) -- or 7 without a mask --
    with the result's planes written k8-interleaved ([plane][row / 8][column][8 rows]) for the weight gradient that
    contracts over the rows: the same values, bit for bit, as interleaving the row-major result.  6144 x 4096: 384 tiles =
    one round of full tiles + 128 tiles as 256 half tiles (the NARROW form of the same epilogue); 640 rows: a last row
    tile of 128."""
    dev = _dev()
    K = 256
    g = torch.Generator(device=dev); g.manual_seed(M + N)
    A3 = _planes(torch.randn(M, K, device=dev, generator=g) * 0.1, K)
    B3 = _planes(torch.randn(N, K, device=dev, generator=g) * 0.1, K)
    bits = torch.randint(0, 256, (M, N // 8), device=dev, generator=g, dtype=torch.uint8)
    for aux in (bits, None):
        rowmajor = torch.zeros((M, 3 * N), dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3 if aux is not None else ops.BE_MASK_X3, A3, K, B3, K, rowmajor, M, N, K, plane_c=N,
                           aux=aux, alpha=0.2)
        want = torch.empty(3 * M * N, dtype=torch.bfloat16, device=dev)
        ops.interleave8_bf16x3(rowmajor, N, M, N, want)
        got = torch.full((3 * M * N,), float("nan"), dtype=torch.bfloat16, device=dev)
        ops.gemm_bf16x3_nt(ops.BE_MASKBITS_X3_KI, A3, K, B3, K, got, M, N, K, plane_c=M * N, aux=aux, alpha=0.2, ldc=N)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))


@pytest.mark.parametrize("mode,steps", [(0, 1), (1, 2), (1, 4)])
def test_sample_gather_writes_the_interleaved_copy(mode, steps):
    """cdml_sample_gather_x3k: beside every step's rows as row-major planes the launch writes them k8-interleaved
    ([plane][row / 8][column][8 rows]: the operand layout of cdml_gemm_bf16x3_tnk) -- the same bits as interleaving the
    row-major planes, every step of a multi-step launch, both sampler modes; the row-major output and the ids are those of
    the launch without the copy."""
    from cdml_amd import engine
    dev = _dev()
    N, F, B = 5000, 500, 64
    Fp = 512
    table = engine.FeatureTable.synthetic(N, F, 0, dev)
    rng = np.random.RandomState(2)
    pairs = rng.randint(0, N, size=(2000, 2)).astype(np.int32)
    pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
    R = B * (3 if mode == 0 else 2)
    mk = lambda: (torch.zeros((steps, R, 3 * Fp), dtype=torch.bfloat16, device=dev), torch.zeros((steps, R), dtype=torch.int32, device=dev),
                  torch.zeros(steps, dtype=torch.int32, device=dev))
    x0, i0, s0 = mk()
    x1, i1, s1 = mk()
    xk = torch.full((steps, 3 * R * Fp), float("nan"), dtype=torch.bfloat16, device=dev)
    args = lambda x, i, s: dict(idx_out=i if steps > 1 else i[0], x_out=x if steps > 1 else x[0], shift_out=s, n_steps=steps)
    ops.sample_gather(mode, pairs, 77, 5, B, table.data, F, **args(x0, i0, s0))
    ops.sample_gather(mode, pairs, 77, 5, B, table.data, F, x_ki=xk if steps > 1 else xk[0], **args(x1, i1, s1))
    torch.cuda.synchronize()
    assert torch.equal(x0.view(torch.int16), x1.view(torch.int16)) and torch.equal(i0, i1) and torch.equal(s0, s1)
    for st in range(steps):
        want = torch.empty(3 * R * Fp, dtype=torch.bfloat16, device=dev)
        ops.interleave8_bf16x3(x1[st], Fp, R, Fp, want)
        assert torch.equal(xk[st].view(torch.int16), want.view(torch.int16)), st


@pytest.mark.parametrize("mode", ["inbatch", "uniform"])
def test_train_step_on_interleaved_operands_is_the_same_step(mode, monkeypatch):
    """CDML_X3_KI=1: the fused gather also writes x_hat k8-interleaved, the data gradient writes dz1 ONLY that way and the
    first layer's weight gradient runs on cdml_gemm_bf16x3_tnk -- the same images, accumulation order and bits: after
    three steps the weights are those of the default step, bit for bit (gather_ahead 1 and 2: a launch per step and a
    launch for two steps)."""
    dev = _dev()
    from cdml_amd import engine, train
    N, F, B = 6000, 500, 128
    table = engine.FeatureTable.synthetic(N, F, 0, dev)
    rng = np.random.RandomState(1)
    pairs = rng.randint(0, N, size=(3000, 2)).astype(np.int32)
    pairs = torch.from_numpy(pairs[pairs[:, 0] != pairs[:, 1]]).to(dev)
    for ahead in (1, 2):
        def mk(ki):
            monkeypatch.setenv("CDML_X3_KI", "1" if ki else "0")
            return train.TrainStep(table, pairs, B, hidden_size=700, output_size=256, mode=mode, optimizer="adam",
                                   base_learning_rate=0.01, device=dev, precision="f32x3", gather_ahead=ahead)
        a, b = mk(False), mk(True)
        assert b.ws.kint and b.ws.dz1 is None and not a.ws.kint
        for _ in range(3):
            a.step(); b.step()
        torch.cuda.synchronize()
        assert torch.equal(a.idx, b.idx) and torch.equal(a.ws.x_hat.view(torch.int16), b.ws.x_hat.view(torch.int16))
        assert torch.equal(a.ws.dz1_f32(), b.ws.dz1_f32())
        assert torch.equal(a.params.grad, b.params.grad) and torch.equal(a.params.flat, b.params.flat)


def test_gather_rows_x3_is_the_split_of_the_gathered_rows():
    """cdml_gather_rows_x3 (the row exchange's last step on the split-fp32 path): request order and the GEMMs' operand form in
    one pass -- the planes of src[idx[r]] exactly as cdml_split_f32_bf16x3 gives them for the gathered fp32 rows; padding
    columns zero; an unanswered request (-1) a NaN row, or untouched without the flag."""
    dev = _dev()
    g = torch.Generator(device=dev); g.manual_seed(3)
    n, F, Fp, R = 700, 500, 512, 300
    src = torch.zeros((n, Fp), device=dev)
    src[:, :F] = torch.randn(n, F, device=dev, generator=g)
    idx = torch.randint(0, n, (R,), device=dev, generator=g, dtype=torch.int32)
    idx[7] = -1; idx[200] = -1
    out = torch.full((R, 3 * Fp), 5.0, dtype=torch.bfloat16, device=dev)
    ops.gather_rows_x3(src, idx, F, out, nan_missing=True)
    rows = src[idx.clamp_min(0).long()]
    want = torch.zeros((R, 3 * Fp), dtype=torch.bfloat16, device=dev)
    ops.split_f32_bf16x3(rows, want, Fp)
    ok = idx >= 0
    assert torch.equal(out[ok].view(torch.int16), want[ok].view(torch.int16))
    assert torch.isnan(out[~ok].float()).all()
    out2 = torch.full((R, 3 * Fp), 5.0, dtype=torch.bfloat16, device=dev)
    ops.gather_rows_x3(src, idx, F, out2)
    assert (out2[~ok].float() == 5.0).all() and torch.equal(out2[ok].view(torch.int16), want[ok].view(torch.int16))
