"""BASELINE config 4 path: fp16 catalogue + bf16 MFMA projection (build-defined
precision; the reference computes in fp32).  Kernel-level checks are exact-ish
(bf16 products are exact in fp32; only the summation order differs); the
end-to-end tolerance this path claims is 5e-3 absolute on unit-norm embeddings
and 2e-2 on the loss against the fp64 oracle run on the same fp16-rounded
features and fp32 master weights."""
import numpy as np
import pytest
import torch

from oracle import sampler as osampler, synth as osynth, tower as otower

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cd(gpu):
    import cdml_amd
    from cdml_amd import engine, engine_bf16, ops, train

    class NS:
        pass
    ns = NS()
    ns.dev, ns.engine, ns.ebf, ns.ops, ns.train, ns.pkg = gpu, engine, engine_bf16, ops, train, cdml_amd
    return ns


def bf(x):
    """Round to bf16 and come back (what the kernels see), as float64."""
    return torch.as_tensor(np.asarray(x, np.float32)).bfloat16().float().numpy().astype(np.float64)


def dbf(x, dev):
    return torch.as_tensor(np.asarray(x, np.float32)).to(dev).bfloat16()


@pytest.mark.parametrize("M,N,K,tile", [(128, 128, 64, 0), (200, 256, 192, 0), (384, 5120, 1536, 128),
                                        (77, 128, 5120, 0),
                                        # the 256x256 ping-pong kernel: minimal, ragged M, long K, many tiles
                                        (256, 256, 128, 256), (700, 512, 384, 256), (1000, 1024, 1536, 256),
                                        (77, 256, 5120, 256), (3100, 4096, 512, 0), (2000, 256, 5120, 0)])
@pytest.mark.parametrize("mfma", ["32", "16"])
def test_gemm_bf16_epilogues(cd, M, N, K, tile, mfma, monkeypatch):
    if tile:
        monkeypatch.setenv("CDML_BF16_TILE", str(tile))
    monkeypatch.setenv("CDML_BF16_MFMA", mfma)       # v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16
    rng = np.random.RandomState(M + N)
    A, B = rng.randn(M, K) / np.sqrt(K), rng.randn(N, K)
    bias, aux = rng.randn(N) * 0.1, rng.randn(M, N)
    ref = bf(A) @ bf(B).T
    dA, dB = dbf(A, cd.dev), dbf(B, cd.dev)
    dbias = torch.as_tensor(bias, dtype=torch.float32).to(cd.dev)
    lrelu = lambda v: np.maximum(v, 0.2 * v)
    tol = dict(atol=2e-5 * np.sqrt(K / 64), rtol=0)
    out = torch.empty((M, N), dtype=torch.float32, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_BIAS_LRELU_F32, dA, dB, out, M, N, K, bias=dbias)
    want = lrelu(ref + bias.astype(np.float32))
    np.testing.assert_allclose(out.cpu().numpy(), want, **tol)
    outb = torch.empty((M, N), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_BIAS_LRELU_BF16, dA, dB, outb, M, N, K, bias=dbias)
    np.testing.assert_allclose(outb.float().cpu().numpy(), want, rtol=2 ** -8, atol=1e-4)
    daux = dbf(aux, cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, dA, dB, outb, M, N, K, aux=daux)
    want_m = ref * np.where(bf(aux) > 0, 1.0, 0.2)
    np.testing.assert_allclose(outb.float().cpu().numpy(), want_m, rtol=2 ** -8, atol=1e-4)
    ws = torch.empty(max(cd.ops.gemm_bf16_workspace(M, N, K), 16) // 4, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_F32, dA, dB, out, M, N, K, workspace=ws)
    np.testing.assert_allclose(out.cpu().numpy(), ref, **tol)
    # epilogue 1 with a workspace: split-K + combine(bias, lrelu) where the layer is narrow
    out1 = torch.empty_like(out)
    cd.ops.gemm_bf16_nt(cd.ops.BE_BIAS_LRELU_F32, dA, dB, out1, M, N, K, bias=dbias, workspace=ws)
    np.testing.assert_allclose(out1.cpu().numpy(), want, **tol)
    out2 = torch.empty_like(out)
    cd.ops.gemm_bf16_nt(cd.ops.BE_F32, dA, dB, out2, M, N, K, workspace=ws)
    assert torch.equal(out, out2)                                  # deterministic split-K


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 1280), (1536, 5120, 2048), (5120, 256, 3072)])
@pytest.mark.parametrize("mfma", ["16", "32"])
def test_gemm_bf16_tn_weight_gradient_form(cd, M, N, K, mfma, monkeypatch):
    """C = A^T . B from k-strided operands (transposed LDS reads, no transposed copies), on both
    MFMA shapes (16x16x32 is what ships; 32x32x16 stays selectable for the A/B)."""
    monkeypatch.setenv("CDML_BF16_MFMA", mfma)
    rng = np.random.RandomState(M + K)
    A, B = rng.randn(K, M) / np.sqrt(K), rng.randn(K, N)
    # make a wrong k-permutation or a swapped fragment half visible: scale rows of k
    A *= (1.0 + (np.arange(K) % 7)[:, None] * 0.25)
    ref = bf(A).T @ bf(B)
    dA, dB = dbf(A, cd.dev), dbf(B, cd.dev)
    assert cd.ops.gemm_bf16_tn_supported(M, N, K, M, N)
    ws = torch.empty(max(cd.ops.gemm_bf16_tn_workspace(M, N, K), 16) // 4, device=cd.dev)
    out = torch.empty((M, N), dtype=torch.float32, device=cd.dev)
    cd.ops.gemm_bf16_tn(dA, dB, out, M, N, K, workspace=ws)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=1e-4 * np.sqrt(K / 64), rtol=0)
    out2 = torch.empty_like(out)
    cs = torch.full((N,), 7.0, dtype=torch.float32, device=cd.dev)
    cd.ops.gemm_bf16_tn(dA, dB, out2, M, N, K, workspace=ws, colsum=cs)      # + bias gradient
    assert torch.equal(out, out2)
    np.testing.assert_allclose(cs.cpu().numpy(), bf(B).sum(0), atol=2e-4 * np.sqrt(K), rtol=0)
    # strided operands (views into wider buffers), as the tower uses them
    wideA = torch.zeros((K, M + 64), dtype=torch.bfloat16, device=cd.dev)
    wideA[:, :M] = dA
    cd.ops.gemm_bf16_tn(wideA[:, :M], dB, out2, M, N, K, workspace=ws)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("mfma", ["16", "32"])
@pytest.mark.parametrize("M1,N1,M2,N2,K", [(256, 256, 256, 256, 128), (512, 768, 1280, 256, 1280),
                                            (1536, 5120, 5120, 256, 6144), (1536, 5120, 5120, 256, 24576)])
def test_gemm_bf16_tn2_joint_weight_gradients(cd, M1, N1, M2, N2, K, mfma, monkeypatch):
    """Both weight gradients (and both bias gradients) in ONE stream-K launch + fix-up pass: values against
    fp64 on the bf16-rounded operands, column sums, equality with the one-product entry point up to the
    summation order over k, bit-identical repetition, strided views; shapes from one tile per product (more
    CUs than units) to config 4's (140 tiles x 192 units over 256 CUs: tile tails and heads everywhere)."""
    monkeypatch.setenv("CDML_BF16_MFMA", mfma)
    ops = cd.ops
    g_ = torch.Generator(device=cd.dev)
    g_.manual_seed(M1 + N2 + K)
    rnd = lambda r, c, sc: (torch.randn(r, c, device=cd.dev, generator=g_) * sc)
    kscale = (1.0 + (torch.arange(K, device=cd.dev) % 7) * 0.25)[:, None]       # a wrong k-permutation shows
    A1 = (rnd(K, M1 + 64, K ** -0.5) * kscale).bfloat16()                       # wider buffers: strides honoured
    B1 = rnd(K, N1, 1.0).bfloat16()
    A2 = (rnd(K, M2, K ** -0.5) * kscale).bfloat16()
    B2 = rnd(K, N2 + 8, 1.0).bfloat16()
    nbytes = ops.gemm_bf16_tn2_workspace(M1, N1, M2, N2, K)
    assert nbytes > 0 and ops.gemm_bf16_tn2_workspace(M1 + 8, N1, M2, N2, K) == 0
    ws = torch.empty(nbytes // 4, device=cd.dev)
    C1 = torch.full((M1, N1 + 4), 5.0, device=cd.dev)
    C2 = torch.full((M2, N2), 5.0, device=cd.dev)
    d1, d2 = torch.full((N1,), 5.0, device=cd.dev), torch.full((N2,), 5.0, device=cd.dev)
    run = lambda c1, c2, e1, e2: ops.gemm_bf16_tn2(A1[:, :M1], B1, c1[:, :N1], M1, N1, A2, B2[:, :N2], c2, M2, N2, K, ws,
                                                    colsum1=e1, colsum2=e2)
    run(C1, C2, d1, d2)
    tol = 1e-4 * (K / 64) ** 0.5
    ref1 = A1[:, :M1].double().T @ B1.double()
    ref2 = A2.double().T @ B2[:, :N2].double()
    assert float((C1[:, :N1].double() - ref1).abs().max()) <= tol and float((C2.double() - ref2).abs().max()) <= tol
    assert bool((C1[:, N1:] == 5.0).all())
    assert float((d1.double() - B1.double().sum(0)).abs().max()) <= 2e-4 * K ** 0.5
    assert float((d2.double() - B2[:, :N2].double().sum(0)).abs().max()) <= 2e-4 * K ** 0.5
    # the one-product entry point: same products, other split of K
    ws1 = torch.empty(max(ops.gemm_bf16_tn_workspace(M1, N1, K), ops.gemm_bf16_tn_workspace(M2, N2, K), 16) // 4, device=cd.dev)
    S1, S2 = torch.empty((M1, N1), device=cd.dev), torch.empty((M2, N2), device=cd.dev)
    ops.gemm_bf16_tn(A1[:, :M1], B1, S1, M1, N1, K, workspace=ws1)
    ops.gemm_bf16_tn(A2, B2[:, :N2], S2, M2, N2, K, workspace=ws1)
    assert float((C1[:, :N1] - S1).abs().max()) <= tol and float((C2 - S2).abs().max()) <= tol
    for _ in range(3):                                                   # bit-identical repetition, no bias gradients asked
        E1, E2 = torch.empty_like(C1), torch.empty_like(C2)
        run(E1, E2, None, None)
        assert torch.equal(E1[:, :N1], C1[:, :N1]) and torch.equal(E2, C2)


@pytest.mark.parametrize("M,N", [(16384, 256), (2075, 5120), (24576 + 17, 2048), (4096, 1024)])
def test_gemm_bf16_k256_streaming_data_gradient(cd, M, N, monkeypatch):
    """The K = 256 mask-epilogue product on the streaming kernel (gemm_bf16_k256.hip; taken when
    M * N >= 2^22): values against fp64 on the bf16-rounded operands, bit-equal to the tiled kernel,
    ragged M, no mask, strided views, and a repeated-launch screen for its DMA / barrier ordering."""
    K = 256
    g = torch.Generator(device=cd.dev)
    g.manual_seed(M + N)
    A = (torch.randn(M, K, device=cd.dev, generator=g) / 16).bfloat16()
    B = torch.randn(N, K, device=cd.dev, generator=g).bfloat16()
    aux = torch.randn(M, N, device=cd.dev, generator=g).bfloat16()
    out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, A, B, out, M, N, K, aux=aux)
    ref = (A.double() @ B.double().T) * torch.where(aux.double() > 0, 1.0, 0.2)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2 ** -8 * ref.abs().max().item() + 1e-4, err
    monkeypatch.setenv("CDML_BF16_TILE", "256" if N % 256 == 0 and M >= 256 else "128")
    tiled = torch.empty_like(out)
    cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, A, B, tiled, M, N, K, aux=aux)
    monkeypatch.delenv("CDML_BF16_TILE")
    assert torch.equal(out, tiled)
    for _ in range(20):                                              # same buffers, every launch bit-equal
        again = torch.empty_like(out)
        cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, A, B, again, M, N, K, aux=aux)
        assert torch.equal(out, again)
    plain = torch.empty_like(out)                                    # no mask: plain bf16 product
    cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, A, B, plain, M, N, K)
    ref_p = A.double() @ B.double().T
    assert (plain.double() - ref_p).abs().max().item() <= 2 ** -8 * ref_p.abs().max().item() + 1e-4
    # operands and result as views into wider buffers; rows past M of the result stay untouched
    wa = torch.zeros((M, K + 64), dtype=torch.bfloat16, device=cd.dev)
    wa[:, :K] = A
    wo = torch.full((M + 40, N + 128), 3.0, dtype=torch.bfloat16, device=cd.dev)
    wm = torch.zeros((M, N + 8), dtype=torch.bfloat16, device=cd.dev)
    wm[:, :N] = aux
    cd.ops.gemm_bf16_nt(cd.ops.BE_MASK_BF16, wa[:, :K], B, wo[:M, :N], M, N, K, aux=wm[:, :N])
    assert torch.equal(wo[:M, :N], out)
    assert bool((wo[M:, :] == 3.0).all()) and bool((wo[:, N:] == 3.0).all())


@pytest.mark.parametrize("K,N", [(64, 64), (1536, 5120), (5120, 256)])
def test_adam_matrix_bf16_equals_adam_then_copies(cd, K, N):
    """The optimizer step that also writes the GEMMs' bf16 operand copies: bit-equal to cdml_adam_step
    followed by cdml_transpose_to_bf16 / cdml_cast_f32_bf16, and within fp32 rounding of the oracle."""
    g_ = torch.Generator(device=cd.dev)
    g_.manual_seed(K + N)
    mk = lambda s: torch.randn(K, N, device=cd.dev, generator=g_) * s
    W, G, M, V = mk(0.05), mk(1e-3), mk(1e-4), mk(1e-4).abs()
    lr = torch.tensor([0.01], device=cd.dev)
    step = torch.tensor([6], dtype=torch.int64, device=cd.dev)              # t = 1 + 6
    w0, m0, v0 = W.clone(), M.clone(), V.clone()
    cd.ops.adam_step(w0, G, m0, v0, 0.0, 1, lr_dev=lr, t_dev=step)
    wt0 = torch.empty((N, K), dtype=torch.bfloat16, device=cd.dev)
    wc0 = torch.empty((K, N), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.transpose_to_bf16(w0, wt0, K, N)
    cd.ops.cast_f32_bf16(w0, wc0, K, N)
    w1, m1, v1 = W.clone(), M.clone(), V.clone()
    wt1 = torch.full((N, K + 64), 9.0, dtype=torch.bfloat16, device=cd.dev)  # wider buffers: strides honoured
    wc1 = torch.full((K, N + 8), 9.0, dtype=torch.bfloat16, device=cd.dev)
    cd.ops.adam_matrix_bf16(w1, G, m1, v1, 0.0, 1, wt=wt1, wc=wc1, lr_dev=lr, t_dev=step)
    assert torch.equal(w0, w1) and torch.equal(m0, m1) and torch.equal(v0, v1)
    assert torch.equal(wt0, wt1[:, :K]) and torch.equal(wc0, wc1[:, :N])
    assert bool((wt1[:, K:] == 9.0).all()) and bool((wc1[:, N:] == 9.0).all())
    assert int(step.item()) == 6                                             # the counter is not advanced here
    w2 = W.clone()
    cd.ops.adam_matrix_bf16(w2, G, M.clone(), V.clone(), 0.0, 1, lr_dev=lr, t_dev=step)   # no copies asked for
    assert torch.equal(w2, w0)
    ow, om, ov = otower.adam_step(W.cpu().numpy(), G.cpu().numpy(), M.cpu().numpy(), V.cpu().numpy(), 7, 0.01)[:3]
    np.testing.assert_allclose(w1.cpu().numpy(), ow, rtol=2e-5, atol=3e-7)
    # the layer's bias vector in the same launch (bit-equal to its own cdml_adam_step), and the step
    # counter advanced by the last block
    nb = N - 3 if N > 64 else 61                                            # not a multiple of the block size
    vb = lambda sc: torch.randn(nb, device=cd.dev, generator=g_) * sc
    B_, GB, MB, VB = vb(0.1), vb(1e-3), vb(1e-4), vb(1e-4).abs()
    b0, mb0, vb0 = B_.clone(), MB.clone(), VB.clone()
    cd.ops.adam_step(b0, GB, mb0, vb0, 0.0, 1, lr_dev=lr, t_dev=step)
    w3, m3, v3, b3, mb3, vb3 = W.clone(), M.clone(), V.clone(), B_.clone(), MB.clone(), VB.clone()
    tickets = cd.ops.new_tickets(cd.dev)
    cd.ops.adam_matrix_bf16(w3, G, m3, v3, 0.0, 1, wt=wt1, lr_dev=lr, t_dev=step, bias=(b3, GB, mb3, vb3),
                            advance_tickets=tickets)
    assert torch.equal(w3, w0) and torch.equal(b3, b0) and torch.equal(mb3, mb0) and torch.equal(vb3, vb0)
    assert int(step.item()) == 7 and int(tickets.abs().sum().item()) == 0   # advanced once, tickets back at zero
    with pytest.raises(cd.pkg.CdmlError):
        cd.ops.adam_matrix_bf16(torch.zeros(96, 64, device=cd.dev), torch.zeros(96, 64, device=cd.dev),
                                torch.zeros(96, 64, device=cd.dev), torch.zeros(96, 64, device=cd.dev), 0.01, 1)


@pytest.mark.parametrize("M", [24576, 1000])
def test_bitmask_epilogues_equal_value_mask(cd, M):
    """leaky-relu' as ONE BIT per element: FC1's epilogue 4 writes the layer's bf16 output (bit-equal to
    epilogue 0) plus the sign bitmask; the K = 256 data gradient's epilogue 5 reads the bitmask and gives
    the bits of epilogue 2 reading the bf16 values.  Ragged M, strided buffers, untouched padding."""
    N, K1, K2 = 5120, 1536, 256
    ops = cd.ops
    assert ops.gemm_bf16_epilogue_supported(ops.BE_BIAS_LRELU_BF16_BITS, M, N, K1, K1, K1, N, N // 8)
    assert ops.gemm_bf16_epilogue_supported(ops.BE_MASKBITS_BF16, M, N, K2, K2, K2, N, N // 8)
    assert not ops.gemm_bf16_epilogue_supported(ops.BE_MASKBITS_BF16, M, N, 320, 320, 320, N, N // 8)
    g_ = torch.Generator(device=cd.dev)
    g_.manual_seed(M)
    A = (torch.randn(M, K1, device=cd.dev, generator=g_) / K1 ** 0.5).bfloat16()
    B = torch.randn(N, K1, device=cd.dev, generator=g_).bfloat16()
    bias = torch.randn(N, device=cd.dev, generator=g_) * 0.1
    h0 = torch.empty((M, N), dtype=torch.bfloat16, device=cd.dev)
    h1 = torch.empty_like(h0)
    bits = torch.full((M + 2, N // 8 + 16), 0xAA, dtype=torch.uint8, device=cd.dev)
    ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16, A, B, h0, M, N, K1, bias=bias)
    ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16_BITS, A, B, h1, M, N, K1, bias=bias, aux=bits[:M, :N // 8])
    assert torch.equal(h0, h1)
    want = torch.from_numpy(np.packbits((h0.float() > 0).cpu().numpy(), axis=1, bitorder="little")).to(cd.dev)
    assert torch.equal(bits[:M, :N // 8], want)
    assert bool((bits[M:] == 0xAA).all()) and bool((bits[:, N // 8:] == 0xAA).all())
    frac = float((h0.float() > 0).float().mean())
    assert 0.3 < frac < 0.7                                                  # both signs are exercised
    dz2 = (torch.randn(M, K2, device=cd.dev, generator=g_) / 16).bfloat16()
    W2 = torch.randn(N, K2, device=cd.dev, generator=g_).bfloat16()
    d0 = torch.empty((M, N), dtype=torch.bfloat16, device=cd.dev)
    d1 = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=cd.dev)
    ops.gemm_bf16_nt(ops.BE_MASK_BF16, dz2, W2, d0, M, N, K2, aux=h0)
    ops.gemm_bf16_nt(ops.BE_MASKBITS_BF16, dz2, W2, d1, M, N, K2, aux=bits[:M, :N // 8])
    assert torch.equal(d0, d1)
    with pytest.raises(ValueError):
        ops.gemm_bf16_nt(ops.BE_MASK_BF16, dz2, W2, d1, M, N, K2, aux=bits[:M, :N // 8])      # bits where values are expected
    with pytest.raises(cd.pkg.CdmlError):
        ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16_BITS, A[:, :64], B[:, :64], h1, M, N, 64, bias=bias, aux=bits[:M, :N // 8])


def test_gemm_bf16_256_race_screen(cd, monkeypatch):
    """The ping-pong kernel orders its LDS-DMA against the fragment reads by counted waits and
    barriers only; a misplaced wait shows up as rare wrong tiles.  Many launches, full chip,
    operands changing under the same buffers, every result compared bit for bit."""
    monkeypatch.setenv("CDML_BF16_TILE", "256")
    M, N, K = 4096, 4096, 1024                      # 256 tiles: one per CU, 16 K-tiles each
    g = torch.Generator(device=cd.dev)
    g.manual_seed(5)
    A = (torch.randn(M, K, device=cd.dev, generator=g) / 32).bfloat16()
    B = torch.randn(N, K, device=cd.dev, generator=g).bfloat16()
    bias = torch.zeros(N, device=cd.dev)
    out = torch.empty((M, N), dtype=torch.float32, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_BIAS_LRELU_F32, A, B, out, M, N, K, bias=bias)
    ref = (A.float() @ B.float().T)
    ref = torch.maximum(ref, 0.2 * ref)
    assert (out - ref).abs().max().item() < 2e-3
    first = out.clone()
    for _ in range(60):
        cd.ops.gemm_bf16_nt(cd.ops.BE_BIAS_LRELU_F32, A, B, out, M, N, K, bias=bias)
        assert torch.equal(out, first)
    At, Bt = A.t().contiguous(), B.t().contiguous()  # [K][M], [K][N]
    ws = torch.empty(max(cd.ops.gemm_bf16_tn_workspace(M, N, K), 16) // 4, device=cd.dev)
    cd.ops.gemm_bf16_tn(At, Bt, out, M, N, K, workspace=ws)
    assert (out - A.float() @ B.float().T).abs().max().item() < 2e-3
    first = out.clone()
    for _ in range(60):
        cd.ops.gemm_bf16_tn(At, Bt, out, M, N, K, workspace=ws)
        assert torch.equal(out, first)


def test_gemm_bf16_tn_refuses_other_shapes(cd):
    assert not cd.ops.gemm_bf16_tn_supported(192, 256, 128, 192, 256)
    a = torch.zeros((128, 192), dtype=torch.bfloat16, device=cd.dev)
    b = torch.zeros((128, 256), dtype=torch.bfloat16, device=cd.dev)
    with pytest.raises(cd.pkg.CdmlError):
        cd.ops.gemm_bf16_tn(a, b, torch.empty((192, 256), device=cd.dev), 192, 256, 128)


def test_gemm_bf16_identity_asymmetric_and_errors(cd):
    K = N = 128
    Bm = (np.arange(N * K).reshape(N, K) % 251 - 100).astype(np.float64)     # exact in bf16
    out = torch.empty((K, N), dtype=torch.float32, device=cd.dev)
    cd.ops.gemm_bf16_nt(cd.ops.BE_F32, dbf(np.eye(K), cd.dev), dbf(Bm, cd.dev), out, K, N, K)
    np.testing.assert_array_equal(out.cpu().numpy(), Bm.T)                   # C = I . B^T
    with pytest.raises(cd.pkg.CdmlError):
        cd.ops.gemm_bf16_nt(cd.ops.BE_F32, dbf(np.eye(64), cd.dev), dbf(np.eye(64), cd.dev),
                            torch.empty((64, 64), device=cd.dev), 64, 64, 64)      # N % 128


def test_transpose_cast_colsum(cd):
    rng = np.random.RandomState(0)
    x = rng.randn(300, 200).astype(np.float32)
    dx = torch.as_tensor(x).to(cd.dev)
    t = torch.zeros((200, 304), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.transpose_to_bf16(dx, t, 300, 200)
    np.testing.assert_array_equal(t[:, :300].float().cpu().numpy(), bf(x).T)
    t2 = torch.zeros((300, 200), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.transpose_to_bf16(t[:, :300], t2, 200, 300)                     # bf16 source
    np.testing.assert_array_equal(t2.float().cpu().numpy(), bf(x))
    c = torch.zeros((300, 200), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.cast_f32_bf16(dx, c, 300, 200)
    np.testing.assert_array_equal(c.float().cpu().numpy(), bf(x))
    for src, ref in ((dx, x.astype(np.float64)), (c, bf(x))):
        out = torch.empty(200, device=cd.dev)
        ws = torch.empty(cd.ops.colsum_workspace_floats(300, 200), device=cd.dev)
        cd.ops.colsum(src, 300, 200, out, ws)
        np.testing.assert_allclose(out.cpu().numpy(), ref.sum(0), atol=1e-4)


def test_fp16_table_and_gather(cd):
    N, F = 500, 1500
    table = cd.ebf.FeatureTableF16.synthetic(N, F, 3, cd.dev)
    want_tab = osynth.features_philox(0, N, F, 3).astype(np.float16)
    np.testing.assert_array_equal(table.data[:, :F].cpu().numpy(), want_tab)
    assert float(table.data[:, F:].abs().max()) == 0
    idx = np.random.RandomState(0).randint(0, N, size=130).astype(np.int32)
    x = torch.full((130, 1536), 7.0, dtype=torch.bfloat16, device=cd.dev)
    cd.ops.gather_rows_f16(table.data, 0, torch.as_tensor(idx).to(cd.dev), F, x)
    want = otower.l2_normalize(want_tab[idx].astype(np.float64), np.float64)[0]
    np.testing.assert_allclose(x[:, :F].float().cpu().numpy(), want, rtol=2 ** -8, atol=1e-6)
    assert float(x[:, F:].float().abs().max()) == 0


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_sample_gather_f16_equals_separate(cd, mode):
    """The fused sampler + gather on the fp16 catalogue (several steps per launch) == the sampler
    kernel + k_gather_rows_f16, ids and rows bit for bit."""
    N, F, B, K = 3000, 1500, 133, 3
    table = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 300, 0)).to(cd.dev)
    rpt = 3 if mode == 0 else 2
    R = B * rpt
    x = torch.full((K, R, 1536), 7.0, dtype=torch.bfloat16, device=cd.dev)
    idx = torch.full((K, R), -1, dtype=torch.int32, device=cd.dev)
    shift = torch.full((K,), -1, dtype=torch.int32, device=cd.dev)
    cd.ops.sample_gather(mode, pairs, 1234, 5, B, table.data, F, idx, x, shift_out=shift, n_steps=K)
    for s in range(K):
        i1 = torch.full((R,), -1, dtype=torch.int32, device=cd.dev)
        s1 = torch.full((1,), -1, dtype=torch.int32, device=cd.dev)
        if mode == 0:
            cd.ops.sample_uniform(pairs, N, 1234, 5 + s, B, i1)
        else:
            cd.ops.sample_inbatch(pairs, 1234, 5 + s, B, i1, s1)
            assert int(s1.item()) == int(shift[s].item())
        x1 = torch.full((R, 1536), 7.0, dtype=torch.bfloat16, device=cd.dev)
        cd.ops.gather_rows_f16(table.data, 0, i1, F, x1)
        assert torch.equal(i1, idx[s]) and torch.equal(x1, x[s])


@pytest.mark.parametrize("mode", ["uniform", "inbatch"])
def test_train_step_bf16_config4_precision(cd, mode):
    """fp16 table + bf16 MFMA step vs the fp64 oracle on the same fp16 features."""
    N, F, B, D = 8000, 1500, 128, 256
    feats16 = osynth.features_numpy(N, F, seed=0).astype(np.float16)
    pairs = osynth.cowatch_pairs(N, 2500, 0)
    table = cd.ebf.FeatureTableF16.from_numpy(feats16, cd.dev)
    ts = cd.train.TrainStep(table, torch.as_tensor(pairs).to(cd.dev), B, mode=mode, precision="bf16",
                            device=cd.dev)
    W = [t.detach().cpu().numpy().astype(np.float64) for t in ts.params.unpadded()]
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    f64 = feats16.astype(np.float64)
    if mode == "uniform":
        idx = osampler.device_triplets_vec(pairs, N, 1234, 0, B)
        np.testing.assert_array_equal(ts.idx.view(B, 3).cpu().numpy(), idx)
        fwd, loss, grads = otower.train_step_grads(f64[idx.reshape(-1)], W, 0.8, np.float64)
    else:
        rows, tri, valid, _ = osampler.device_inbatch(pairs, 1234, 0, B)
        fwd = otower.vnet_forward(f64[rows], *W, dtype=np.float64)
        loss = otower.hinge_loss_indexed(fwd["l2_norm"], tri, valid.astype(bool), 0.8, np.float64)
        dE = otower.hinge_loss_indexed_backward(fwd["l2_norm"], tri, valid.astype(bool), 0.8, np.float64)
        grads = otower.vnet_backward(fwd, W[2], dE, np.float64)
    e = ts.ws.e[:, :D].cpu().numpy()
    assert np.abs(e - fwd["l2_norm"]).max() < 5e-3                       # stated tolerance of this path
    assert abs(ts.loss() - float(loss["hinge_loss"])) < 2e-2
    np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)  # normalisation itself is fp32
    # (gradients: on this iid catalogue |g| ~ 1e-8 is a sum of cancelling terms, so no relative bar is
    # meaningful here; the gradient bar of this path -- relative L2 <= 1e-2 per tensor against fp64 --
    # is test_gpu_fullsize.py::test_gradients_well_conditioned_production_shape[bf16])
    assert all(torch.isfinite(g).all() for g in ts.params.unpadded(grads=True))
    for _ in range(2):
        ts.step()
    assert np.isfinite(ts.loss())
    with pytest.raises(ValueError):
        cd.train.TrainStep(table, torch.as_tensor(pairs).to(cd.dev), B, precision="f32", device=cd.dev)


@pytest.mark.parametrize("precision", ["f32", "bf16"])
def test_backward_in_row_blocks_equals_one_piece(cd, precision):
    """Data-parallel runs produce dW1 in two row blocks of W1 so that each block's all-reduce can
    start early: same gradient (up to the split-K summation order), and the callback ranges tile
    [W1|b1] of the flat gradient exactly."""
    N, F, B = 6000, 1500, 128
    if precision == "bf16":
        table, eng = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev), cd.ebf
    else:
        table, eng = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev), cd.engine
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 800, 0)).to(cd.dev)
    ts = cd.train.TrainStep(table, pairs, B, mode="uniform", precision=precision, device=cd.dev)
    ts.fetch(); ts.forward_loss()
    ts.backward()
    g_one = ts.params.grad.clone()
    ts.params.grad.zero_()
    ranges = []
    eng.tower_backward(ts.params, ts.ws, w1_chunks=2, after_w1_chunk=lambda lo, hi: ranges.append((lo, hi)))
    torch.cuda.synchronize()
    L = ts.layout
    assert ranges == [(0, L.Fp // 2 * L.Hp), (L.Fp // 2 * L.Hp, L.Fp * L.Hp + L.Hp)]
    scale = g_one.abs().max().item()
    assert (ts.params.grad - g_one).abs().max().item() <= 2e-6 * max(scale, 1.0) + 1e-9


def test_full_size_properties_config4(cd):
    """BASELINE config 4's per-GPU shape (B = 8192 uniform triplets -> 24 576 rows; scaled
    catalogue) through size-independent properties: bit-exact ids, unit-norm rows in and out at the
    path's tolerance, the weight gradient = x_hat^T dz1 recomputed by torch from the SAME bf16
    operands, the bias gradient = column sums, bit-identical repetition."""
    N, F, B = 200000, 1500, 8192
    table = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev)
    pairs_np = osynth.cowatch_pairs(N, 40000, 0)
    ts = cd.train.TrainStep(table, torch.as_tensor(pairs_np).to(cd.dev), B, mode="uniform", precision="bf16",
                            device=cd.dev)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    idx = osampler.device_triplets_vec(pairs_np, N, 1234, 0, B)
    np.testing.assert_array_equal(ts.idx.view(B, 3).cpu().numpy(), idx)
    xn = ts.ws.x_hat[:, :F].float().norm(dim=1)
    assert float((xn - 1).abs().max()) < 4e-3                         # bf16 rows: 2^-9 relative per element
    en = ts.ws.e.norm(dim=1)
    assert float((en - 1).abs().max()) < 1e-5                         # the output normalisation is fp32
    assert torch.isfinite(ts.params.grad).all() and abs(ts.loss() - 0.8) < 0.2
    L = ts.layout
    ref = ts.ws.x_hat.double().T @ ts.ws.dz1[:, :256].double()        # same bf16 operands, fp64 sum
    scale = float(ref.abs().max())
    assert float((ts.params.gW1[:, :256].double() - ref).abs().max()) < 1e-5 * max(scale, 1e-3) + 1e-7
    db1 = ts.ws.dz1.double().sum(0)
    assert float((ts.params.gb1.double() - db1).abs().max()) < 1e-5 * max(float(db1.abs().max()), 1e-3) + 1e-7
    g0 = ts.params.grad.clone()
    ts.fetch(); ts.forward_loss(); ts.backward()
    assert torch.equal(g0, ts.params.grad)


def test_bf16_graph_replay_equals_eager(cd):
    """Config 4 asks for a hipGraph-captured step: replay must give the eager step's bits
    (device-side step counter and learning rate, no allocation on the step path)."""
    N, F, B = 6000, 1500, 128
    table = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 800, 0)).to(cd.dev)
    kw = dict(mode="uniform", precision="bf16", device=cd.dev)
    a = cd.train.TrainStep(table, pairs, B, use_graph=False, **kw)
    b = cd.train.TrainStep(table, pairs, B, use_graph=True, **kw)
    for _ in range(4):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.params.flat, b.params.flat)
    assert torch.equal(a.idx, b.idx) and int(b.step_dev.item()) == 4


def test_bf16_training_learns(cd):
    rng = np.random.RandomState(0)
    centers = rng.random_sample((10, 64))
    cid = rng.randint(0, 10, size=2000)
    feats = (centers[cid] + 0.05 * rng.randn(2000, 64)).clip(0, None).astype(np.float16)
    a = rng.randint(0, 2000, size=6000)
    p = np.array([rng.choice(np.flatnonzero(cid == cid[i])) for i in a])
    pairs = np.stack([a, p], 1)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]].astype(np.int32)
    table = cd.ebf.FeatureTableF16.from_numpy(feats, cd.dev)
    ts = cd.train.TrainStep(table, torch.as_tensor(pairs).to(cd.dev), 64, hidden_size=128, output_size=32,
                            mode="uniform", base_learning_rate=0.002, precision="bf16", device=cd.dev)
    losses = []
    for _ in range(150):
        ts.step()
        if ts.global_step == 1 or ts.global_step % 25 == 0:
            losses.append(ts.loss())
    assert max(losses[2:]) < 0.5 * losses[0], losses        # from ~margin down, and it stays down
