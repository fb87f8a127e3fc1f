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


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 512, 384), (2048, 256, 1024), (8192, 5120, 1536)])
@pytest.mark.parametrize("products", [6, 3])
def test_gemm_x3_nt_matches_fp64_like_fp32(M, N, K, products):
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
    assert abs(a.loss() - b.loss()) < 1e-5
    if mode == "semihard":
        assert torch.equal(a.neg_row, b.neg_row)
    for _ in range(3):                                   # keeps stepping: the plane copies follow the optimizer
        a.step(); b.step()
    assert abs(a.loss() - b.loss()) < 2e-3
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
