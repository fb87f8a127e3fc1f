"""Parity of the HIP path (called through the C ABI) against the CPU oracle.

Bars: bit-exact for sampled indices and the synthetic table; <= 1e-5 absolute
(fp32) on embeddings and loss -- the tolerance BASELINE.json's north_star
states; gradients / optimizer updates within 1e-5 of the fp64 oracle scaled to
their magnitude.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sampler as osampler, synth as osynth, tower as otower

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope="module")
def cd(gpu):
    import cdml_amd
    from cdml_amd import engine, inputs, losses, models, ops, train, utils
    cdml_amd.load_library()

    class NS:
        pass
    ns = NS()
    ns.dev = gpu
    ns.engine, ns.inputs, ns.losses, ns.models = engine, inputs, losses, models
    ns.ops, ns.train, ns.utils, ns.pkg = ops, train, utils, cdml_amd
    return ns


def dt(a, dev, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(dev)


def padded(a, rows, cols, dev):
    out = torch.zeros((rows, cols), dtype=torch.float32, device=dev)
    out[:a.shape[0], :a.shape[1]] = dt(a, dev)
    return out


def ru(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------ table + sampler --
def test_fill_table_bit_exact(cd):
    for n, F, row0, seed in ((64, 1500, 0, 0), (33, 100, 1000, 7), (5, 7, 2 ** 33, 3)):
        stride = ru(F, 64)
        t = torch.full((n, stride), -1.0, device=cd.dev)
        cd.ops.fill_uniform_table(t, row0, F, seed)
        got = t.cpu().numpy()
        np.testing.assert_array_equal(got[:, :F], osynth.features_philox(row0, n, F, seed))
        assert np.all(got[:, F:] == 0)


@pytest.mark.parametrize("n_rows,batch", [(10000, 128), (3, 64), (5, 1), (1000000, 4096)])
def test_sampler_uniform_bit_exact(cd, n_rows, batch):
    rng = np.random.RandomState(0)
    pairs = rng.randint(0, n_rows, size=(1000, 2))
    pairs = pairs[pairs[:, 0] != pairs[:, 1]].astype(np.int32)
    dp = dt(pairs, cd.dev, torch.int32)
    out = torch.empty((batch, 3), dtype=torch.int32, device=cd.dev)
    for step in (0, 1, 5, 2 ** 33 + 9):
        cd.ops.sample_uniform(dp, n_rows, 1234, step, batch, out)
        want = osampler.device_triplets_vec(pairs, n_rows, 1234, step, batch)
        np.testing.assert_array_equal(out.cpu().numpy(), want)
    # step read from device memory, rank slice of a global batch
    step_dev = torch.tensor([3], dtype=torch.int64, device=cd.dev)
    half = max(batch // 2, 1)
    cd.ops.sample_uniform(dp, n_rows, 99, None, half, out[:half], slot0=half, batch_global=2 * half,
                          step_dev=step_dev)
    want = osampler.device_triplets_vec(pairs, n_rows, 99, 3, half, slot0=half, batch_global=2 * half)
    np.testing.assert_array_equal(out[:half].cpu().numpy(), want)
    cd.ops.step_advance(step_dev)
    assert int(step_dev.item()) == 4


def test_sampler_scalar_spec_small(cd):
    pairs = np.array([[0, 1], [1, 2], [2, 0]], dtype=np.int32)
    out = torch.empty((32, 3), dtype=torch.int32, device=cd.dev)
    cd.ops.sample_uniform(dt(pairs, cd.dev, torch.int32), 3, 5, 2, 32, out)
    np.testing.assert_array_equal(out.cpu().numpy(), osampler.device_triplets(pairs, 3, 5, 2, 32))


def test_sampler_inbatch_bit_exact(cd):
    pairs = osynth.cowatch_pairs(500, 100, 1)
    dp = dt(pairs, cd.dev, torch.int32)
    for B in (2, 16, 4096):
        rows = torch.empty(2 * B, dtype=torch.int32, device=cd.dev)
        shift = torch.empty(1, dtype=torch.int32, device=cd.dev)
        for step in (0, 3, 2 ** 32 + 1):
            cd.ops.sample_inbatch(dp, 77, step, B, rows, shift)
            wrows, _, _, ws = osampler.device_inbatch(pairs, 77, step, B)
            np.testing.assert_array_equal(rows.cpu().numpy(), wrows)
            assert int(shift.item()) == ws


# -------------------------------------------------------------------- gather --
@pytest.mark.parametrize("F", [1500, 100, 6, 2048])
def test_gather_rows(cd, F):
    n = 300
    feats = np.random.RandomState(1).random_sample((n, F)).astype(np.float32)
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    idx = np.random.RandomState(2).randint(0, n, size=77).astype(np.int32)
    didx = dt(idx, cd.dev, torch.int32)
    stride = ru(F, 64)
    out = torch.full((77, stride), 9.0, device=cd.dev)
    inv = torch.empty(77, device=cd.dev)
    cd.ops.gather_rows(table.data, 0, didx, F, out, normalize=True, inv_norm_out=inv)
    want, winv = otower.l2_normalize(osampler.gather(feats, idx), np.float32)
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, :F], want, atol=1e-6)
    assert np.all(got[:, F:] == 0)
    np.testing.assert_allclose(inv.cpu().numpy(), winv[:, 0], rtol=1e-6)
    # raw gather is a bit-exact copy (inputs.py:158)
    raw = torch.empty((77, ru(F, 4)), device=cd.dev)
    cd.ops.gather_rows(table.data, 0, didx, F, raw, normalize=False)
    np.testing.assert_array_equal(raw.cpu().numpy()[:, :F], feats[idx])


def test_gather_shard_and_oob_flag(cd):
    feats = np.random.RandomState(1).random_sample((50, 8)).astype(np.float32)
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev, row0=100, n_rows_global=200)
    flag = torch.zeros(1, dtype=torch.int32, device=cd.dev)
    out = torch.empty((3, 8), device=cd.dev)
    cd.ops.gather_rows(table.data, 100, dt([100, 149, 120], cd.dev, torch.int32), 8, out,
                       normalize=False, oob_flag=flag)
    np.testing.assert_array_equal(out.cpu().numpy(), feats[[0, 49, 20]])
    assert int(flag.item()) == 0
    cd.ops.gather_rows(table.data, 100, dt([99, 150, 120], cd.dev, torch.int32), 8, out,
                       normalize=False, oob_flag=flag)
    assert int(flag.item()) == 1


def test_route_and_scatter_rows(cd):
    """Fixed-capacity routing of the row exchange: per-owner segments filled in ascending request
    order, -1 padding, overflow / bad-id flags; the scatter is the inverse of the gather by slot;
    gather_rows leaves -1 slots untouched."""
    rng = np.random.RandomState(0)
    for world, n, per, cap in ((8, 24576, 1250000, 4184), (2, 96, 1500, 96), (3, 5000, 700, 1800), (1, 300, 10 ** 6, 300)):
        ids = rng.randint(0, per * world, size=n).astype(np.int32)
        send = torch.zeros(world * cap, dtype=torch.int32, device=cd.dev)
        slot = torch.zeros(n, dtype=torch.int32, device=cd.dev)
        flag = torch.zeros(1, dtype=torch.int32, device=cd.dev)
        cd.ops.route_rows(dt(ids, cd.dev, torch.int32), per, world, cap, send, slot, flag)
        owner = ids // per
        want_send = np.full(world * cap, -1, np.int32)
        want_slot = np.zeros(n, np.int32)
        cnt = np.zeros(world, np.int64)
        for r in range(n):
            want_slot[r] = owner[r] * cap + cnt[owner[r]]
            want_send[want_slot[r]] = ids[r]
            cnt[owner[r]] += 1
        assert cnt.max() <= cap and int(flag.item()) == 0
        np.testing.assert_array_equal(send.cpu().numpy(), want_send)
        np.testing.assert_array_equal(slot.cpu().numpy(), want_slot)
        # rows travel by slot and come back by slot
        rows = torch.as_tensor(rng.randn(n, 64).astype(np.float32)).to(cd.dev)
        buf = torch.full((world * cap, 64), -5.0, device=cd.dev)
        cd.ops.scatter_rows(rows, slot, buf, 64)
        back = torch.full((n, 64), 9.0, device=cd.dev)
        cd.ops.gather_rows(buf, 0, slot, 64, back, normalize=False)
        assert torch.equal(back, rows)
        pad = torch.as_tensor(want_send == -1).to(cd.dev)
        assert bool((buf[pad] == -5.0).all())                       # padding slots untouched
        out = torch.full((world * cap, 64), 3.0, device=cd.dev)
        tbl = torch.as_tensor(rng.randn(per * world if per * world < 10000 else 10000, 64).astype(np.float32)).to(cd.dev)
        cd.ops.gather_rows(tbl, 0, torch.clamp(send, max=tbl.shape[0] - 1), 64, out, normalize=False)
        assert bool((out[pad] == 3.0).all())                        # gather skips id -1
    # a full segment and an id outside the catalogue raise the flag bits
    ids = dt(np.array([0, 1, 2, 3, 4, 2500], np.int32), cd.dev, torch.int32)
    send = torch.zeros(8, dtype=torch.int32, device=cd.dev)
    slot = torch.zeros(6, dtype=torch.int32, device=cd.dev)
    flag = torch.zeros(1, dtype=torch.int32, device=cd.dev)
    cd.ops.route_rows(ids, 1000, 2, 4, send, slot, flag)
    assert int(flag.item()) == 3 and slot.cpu().tolist() == [0, 1, 2, 3, -1, -1]


@pytest.mark.parametrize("n_steps", [2, 3])
@pytest.mark.parametrize("mode", [0, 1])
def test_sample_gather_several_steps_per_launch(cd, mode, n_steps):
    """One launch for steps t..t+n-1 == n single-step launches (ids, shift and rows, bit for bit)."""
    N, F, B = 3000, 1500, 133
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = dt(osynth.cowatch_pairs(N, 300, 0), cd.dev, torch.int32)
    rpt = 3 if mode == 0 else 2
    R = B * rpt
    x = torch.full((n_steps, R, 1536), -1.0, device=cd.dev)
    idx = torch.full((n_steps, R), -1, dtype=torch.int32, device=cd.dev)
    shift = torch.full((n_steps,), -1, dtype=torch.int32, device=cd.dev)
    step_dev = torch.tensor([7], dtype=torch.int64, device=cd.dev)
    cd.ops.sample_gather(mode, pairs, 1234, None, B, table.data, F, idx, x, shift_out=shift, slot0=5,
                         batch_global=B + 9, step_dev=step_dev, n_steps=n_steps)
    for s in range(n_steps):
        x1 = torch.full((R, 1536), -1.0, device=cd.dev)
        i1 = torch.full((R,), -1, dtype=torch.int32, device=cd.dev)
        s1 = torch.full((1,), -1, dtype=torch.int32, device=cd.dev)
        cd.ops.sample_gather(mode, pairs, 1234, 7 + s, B, table.data, F, i1, x1, shift_out=s1, slot0=5,
                             batch_global=B + 9)
        assert torch.equal(i1, idx[s]) and torch.equal(x1, x[s])
        if mode == 1:
            assert int(s1.item()) == int(shift[s].item())


@pytest.mark.parametrize("mode", [0, 1])
def test_fused_sample_gather_equals_separate(cd, mode):
    N, F, B = 2000, 1500, 130
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = dt(osynth.cowatch_pairs(N, 200, 0), cd.dev, torch.int32)
    rpt = 3 if mode == 0 else 2
    Fp = table.data.shape[1]
    x1 = torch.empty((rpt * B, Fp), device=cd.dev)
    x2 = torch.empty_like(x1)
    i1 = torch.empty(rpt * B, dtype=torch.int32, device=cd.dev)
    i2 = torch.empty_like(i1)
    s1 = torch.zeros(1, dtype=torch.int32, device=cd.dev)
    s2 = torch.zeros_like(s1)
    cd.ops.sample_gather(mode, pairs, 5, 11, B, table.data, F, i1, x1, shift_out=s1)
    if mode == 0:
        cd.ops.sample_uniform(pairs, N, 5, 11, B, i2)
    else:
        cd.ops.sample_inbatch(pairs, 5, 11, B, i2, s2)
    cd.ops.gather_rows(table.data, 0, i2, F, x2, normalize=True)
    assert torch.equal(i1, i2) and torch.equal(s1, s2)
    assert float((x1 - x2).abs().max()) < 1e-7          # same rows; fma contraction may differ per kernel
    np.testing.assert_allclose(x1[:, :F].norm(dim=1).cpu().numpy(), 1.0, atol=1e-6)


# --------------------------------------------------------------------- norms --
def test_l2norm_fwd_bwd(cd):
    rng = np.random.RandomState(0)
    z = rng.randn(70, 256).astype(np.float32)
    z[3] = 0                                   # clamped row: eps branch
    z[4] = 1e-8
    g = rng.randn(70, 256).astype(np.float32)
    dz_, dg = dt(z, cd.dev), dt(g, cd.dev)
    y = torch.empty_like(dz_)
    inv = torch.empty(70, device=cd.dev)
    cd.ops.l2norm_fwd(dz_, 256, y, inv)
    wy, winv = otower.l2_normalize(z, np.float64)
    np.testing.assert_allclose(y.cpu().numpy(), wy, atol=1e-6)
    np.testing.assert_allclose(inv.cpu().numpy(), winv[:, 0], rtol=1e-6)
    out = torch.empty_like(dz_)
    cd.ops.l2norm_bwd(dz_, dg, 256, out, lrelu_alpha=-1.0)
    want = otower.l2_normalize_backward(z.astype(np.float64), winv, g, np.float64)
    scale = np.abs(want).max(axis=1, keepdims=True) + 1e-30
    assert np.max(np.abs(out.cpu().numpy() - want) / scale) < 1e-5
    cd.ops.l2norm_bwd(dz_, dg, 256, out, lrelu_alpha=0.2)
    want2 = otower.leaky_relu_backward(z.astype(np.float64), want)
    assert np.max(np.abs(out.cpu().numpy() - want2) / scale) < 1e-5


# ---------------------------------------------------------------------- GEMM --
FC_SHAPES = [(15, 64, 64), (130, 96, 192), (384, 1536, 5120), (257, 5120, 256), (64, 32, 128)]


@pytest.mark.parametrize("M,K1,N1,K2,N2", [(777, 1024, 4096, 4096, 128), (8192, 1536, 5120, 5120, 256),
                                            (4100, 1536, 5120, 5120, 256), (256, 2048, 2048, 128, 128),
                                            (300, 4096, 4096, 128, 128),      # 1025 tiles: two whole rounds + one tile in 10 parts
                                            (1000, 4096, 2048, 2048, 128)])   # 528 tiles: one whole round + 16 split
def test_fc_bwd_weight2_stream_k(cd, M, K1, N1, K2, N2):
    """Both weight gradients in one stream-K launch == the per-layer launches (fp32 summation
    order aside) == x^T dy in fp64; bias gradients ride along; repeatable bit for bit."""
    g = torch.Generator(device=cd.dev)
    g.manual_seed(M)
    x1 = torch.randn(M, K1, device=cd.dev, generator=g) / 8
    dy1 = torch.randn(M, N1, device=cd.dev, generator=g) / 8
    x2 = torch.randn(M, K2, device=cd.dev, generator=g) / 8
    dy2 = torch.randn(M, N2, device=cd.dev, generator=g) / 8
    nb = cd.ops.fc_bwd_weight2_workspace(M, K1, N1, K2, N2)
    assert nb > 0
    ws = torch.empty(nb // 4, device=cd.dev)
    f = lambda *s: torch.full(s, 7.0, device=cd.dev)
    dW1, db1, dW2, db2 = f(K1, N1), f(N1), f(K2, N2), f(N2)
    cd.ops.fc_bwd_weight2(x1, dy1, dW1, db1, K1, N1, x2, dy2, dW2, db2, K2, N2, M, ws)
    for x, dy, dW, db in ((x1, dy1, dW1, db1), (x2, dy2, dW2, db2)):
        ref = (x.double().T @ dy.double())
        scale = float(ref.abs().max())
        assert float((dW.double() - ref).abs().max()) < 2e-6 * scale * max(1.0, np.sqrt(M / 256))
        rb = dy.double().sum(0)
        assert float((db.double() - rb).abs().max()) < 2e-6 * float(rb.abs().max()) * max(1.0, np.sqrt(M / 256)) + 1e-6
    first = [t.clone() for t in (dW1, db1, dW2, db2)]
    for _ in range(3):
        cd.ops.fc_bwd_weight2(x1, dy1, dW1, db1, K1, N1, x2, dy2, dW2, db2, K2, N2, M, ws)
    assert all(torch.equal(a, b) for a, b in zip(first, (dW1, db1, dW2, db2)))
    # no bias gradients wanted
    cd.ops.fc_bwd_weight2(x1, dy1, dW1, None, K1, N1, x2, dy2, dW2, None, K2, N2, M, ws)
    assert torch.equal(dW1, first[0]) and torch.equal(dW2, first[2])
    assert cd.ops.fc_bwd_weight2_workspace(M, 192, 256, 256, 128) == 0          # not multiples of 128 / too few tiles


def test_fc_race_screen(cd):
    """DMA-wait / barrier ordering of the LDS-DMA staged kernels: many launches on the whole
    chip, every result compared bit for bit with the first."""
    M, K, N = 4096, 512, 2048
    g = torch.Generator(device=cd.dev)
    g.manual_seed(3)
    x = torch.randn(M, K, device=cd.dev, generator=g) / 16
    W = torch.randn(K, N, device=cd.dev, generator=g)
    dy = torch.randn(M, N, device=cd.dev, generator=g)
    b = torch.zeros(N, device=cd.dev)
    y = torch.empty((M, N), device=cd.dev)
    cd.ops.fc_lrelu_fwd(x, W, b, y, M, K, N)
    first = y.clone()
    ws = torch.empty(cd.ops.fc_bwd_weight_workspace(M, K, N) // 4, device=cd.dev)
    dW, db = torch.empty((K, N), device=cd.dev), torch.empty(N, device=cd.dev)
    cd.ops.fc_bwd_weight(x, dy, dW, db, ws, M, K, N)
    dW0 = dW.clone()
    for _ in range(40):
        cd.ops.fc_lrelu_fwd(x, W, b, y, M, K, N)
        assert torch.equal(y, first)
        cd.ops.fc_bwd_weight(x, dy, dW, db, ws, M, K, N)
        assert torch.equal(dW, dW0)


@pytest.mark.parametrize("M,K,N", FC_SHAPES)
def test_fc_lrelu_fwd(cd, M, K, N):
    rng = np.random.RandomState(M + K + N)
    x = rng.randn(M, K) / np.sqrt(K)
    W = rng.randn(K, N) * 0.5
    b = rng.randn(N) * 0.1
    y = torch.full((M, N), 7.0, device=cd.dev)
    cd.ops.fc_lrelu_fwd(dt(x, cd.dev), dt(W, cd.dev), dt(b, cd.dev), y, M, K, N)
    want = otower.fully_connected(x.astype(np.float32).astype(np.float64),
                                  W.astype(np.float32).astype(np.float64),
                                  b.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(y.cpu().numpy(), want, atol=TOL)


def test_fc_fwd_identity_asymmetric(cd):
    """A = I with an asymmetric B catches a transposed C/D register map."""
    K = N = 128
    Bm = np.arange(K * N, dtype=np.float64).reshape(K, N) % 251 - 100
    y = torch.empty((K, N), device=cd.dev)
    cd.ops.fc_lrelu_fwd(dt(np.eye(K), cd.dev), dt(Bm, cd.dev), dt(np.zeros(N), cd.dev), y, K, K, N,
                        alpha=1.0)
    np.testing.assert_array_equal(y.cpu().numpy(), Bm.astype(np.float32))


@pytest.mark.parametrize("M,K,N", [(15, 64, 64), (130, 192, 96), (384, 5120, 256), (200, 128, 32)])
def test_fc_bwd_data(cd, M, K, N):
    rng = np.random.RandomState(M + K)
    dy = rng.randn(M, N) / np.sqrt(N)
    W = rng.randn(K, N) * 0.5
    xp = rng.randn(M, K)
    dx = torch.empty((M, K), device=cd.dev)
    cd.ops.fc_bwd_data(dt(dy, cd.dev), dt(W, cd.dev), dt(xp, cd.dev), dx, M, K, N)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    want = otower.leaky_relu_backward(f(xp), f(dy) @ f(W).T)
    np.testing.assert_allclose(dx.cpu().numpy(), want, atol=TOL)
    cd.ops.fc_bwd_data(dt(dy, cd.dev), dt(W, cd.dev), None, dx, M, K, N)
    np.testing.assert_allclose(dx.cpu().numpy(), f(dy) @ f(W).T, atol=TOL)


@pytest.mark.parametrize("M,K", [(8192, 5120), (4096, 2048), (12288, 384)])
def test_fc_bwd_data_output_layer_shapes(cd, M, K):
    """The output layer's data gradient (256-deep contraction) at step-sized shapes: against fp64,
    without a mask, strided views, repeated launches bit-equal, a ragged row count."""
    N = 256
    g_ = torch.Generator(device=cd.dev)
    g_.manual_seed(M + K)
    dy = torch.randn(M, N, device=cd.dev, generator=g_) / 64
    W = torch.randn(K, N, device=cd.dev, generator=g_) * (1.0 + (torch.arange(N, device=cd.dev) % 5) * 0.5)
    xp = torch.randn(M, K, device=cd.dev, generator=g_)
    dx = torch.full((M, K), 7.0, device=cd.dev)
    cd.ops.fc_bwd_data(dy, W, xp, dx, M, K, N)
    ref = (dy.double() @ W.double().T) * torch.where(xp > 0, 1.0, 0.2).double()
    assert (dx.double() - ref).abs().max().item() <= TOL
    for _ in range(5):
        again = torch.empty_like(dx)
        cd.ops.fc_bwd_data(dy, W, xp, again, M, K, N)
        assert torch.equal(dx, again)
    cd.ops.fc_bwd_data(dy, W, None, dx, M, K, N)
    assert (dx.double() - dy.double() @ W.double().T).abs().max().item() <= TOL
    wide_dy = torch.zeros((M, N + 64), device=cd.dev)
    wide_dy[:, :N] = dy
    wide_dx = torch.full((M + 8, K + 32), 3.0, device=cd.dev)
    wide_xp = torch.zeros((M, K + 4), device=cd.dev)
    wide_xp[:, :K] = xp
    cd.ops.fc_bwd_data(wide_dy[:, :N], W, wide_xp[:, :K], wide_dx[:M, :K], M, K, N)
    assert torch.equal(wide_dx[:M, :K], again)
    assert bool((wide_dx[M:, :] == 3.0).all()) and bool((wide_dx[:, K:] == 3.0).all())
    cd.ops.fc_bwd_data(dy[:M - 8], W, xp[:M - 8], dx[:M - 8], M - 8, K, N)
    assert (dx[:M - 8].double() - ref[:M - 8]).abs().max().item() <= TOL


@pytest.mark.parametrize("M,K,N", [(15, 64, 64), (130, 192, 128), (384, 1536, 5120),
                                   (3000, 5120, 256), (1000, 128, 64)])
def test_fc_bwd_weight(cd, M, K, N):
    rng = np.random.RandomState(M + N)
    x = rng.randn(M, K) / np.sqrt(M)
    dy = rng.randn(M, N)
    nb = cd.ops.fc_bwd_weight_workspace(M, K, N)
    assert nb > 0
    ws = torch.empty(nb // 4, device=cd.dev)
    dW = torch.empty((K, N), device=cd.dev)
    db = torch.empty(N, device=cd.dev)
    cd.ops.fc_bwd_weight(dt(x, cd.dev), dt(dy, cd.dev), dW, db, ws, M, K, N)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(dW.cpu().numpy(), f(x).T @ f(dy), atol=2e-5)
    np.testing.assert_allclose(db.cpu().numpy(), f(dy).sum(0), atol=2e-5 * np.sqrt(M))
    # deterministic: a second run is bit-identical
    dW2 = torch.empty_like(dW)
    cd.ops.fc_bwd_weight(dt(x, cd.dev), dt(dy, cd.dev), dW2, db, ws, M, K, N)
    assert torch.equal(dW, dW2)


# ---------------------------------------------------------------------- loss --
@pytest.mark.parametrize("margin", [0.1, 0.8])
def test_hinge_known_answer_facade(cd, golden_dir, margin):
    k = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    x = np.zeros((5, 3, 4), np.float32)
    x[:, :, :2] = np.array(k["loss_input"], np.float32)       # embed dim padded 2 -> 4 with zeros
    loss_fn = cd.utils.find_class_by_name("HingeLoss", [cd.losses])()
    out = loss_fn.calculate_loss(dt(x, cd.dev), margin=margin)
    want = k[f"loss_margin_{margin}"]
    assert tuple(out["pos_dist"].shape) == (5, 1) and tuple(out["anchors"].shape) == (5, 1, 4)
    np.testing.assert_allclose(out["pos_dist"].cpu().numpy()[:, 0], k["loss_pos_dist"])
    np.testing.assert_allclose(out["neg_dist"].cpu().numpy()[:, 0], k["loss_neg_dist"])
    np.testing.assert_allclose(out["hinge_dist"].cpu().numpy()[:, 0], want["hinge_dist"], rtol=1e-6)
    np.testing.assert_allclose(out["hinge_loss"].item(), want["hinge_loss"], rtol=1e-6)


@pytest.mark.parametrize("B,D", [(1, 4), (7, 256), (300, 256), (64, 32)])
def test_hinge_fwd_bwd_vs_oracle(cd, B, D):
    rng = np.random.RandomState(B)
    e = otower.l2_normalize(rng.randn(3 * B, D), np.float64)[0]
    de = torch.empty((3 * B, D), device=cd.dev)
    pos, neg, hinge = (torch.empty(B, device=cd.dev) for _ in range(3))
    stats = torch.empty(4, device=cd.dev)
    cd.ops.triplet_hinge(dt(e, cd.dev), B, D, 0.8, pos, neg, hinge, stats, de)
    e32 = e.astype(np.float32).astype(np.float64).reshape(B, 3, D)
    w = otower.hinge_loss(e32, 0.8, np.float64)
    np.testing.assert_allclose(pos.cpu().numpy(), w["pos_dist"][:, 0], atol=TOL)
    np.testing.assert_allclose(hinge.cpu().numpy(), w["hinge_dist"][:, 0], atol=TOL)
    np.testing.assert_allclose(stats[0].item(), w["hinge_loss"], atol=TOL)
    np.testing.assert_allclose(stats[1].item(), w["pos_dist"].mean(), atol=TOL)
    np.testing.assert_allclose(stats[2].item(), w["neg_dist"].mean(), atol=TOL)
    wd = otower.hinge_loss_backward(e32, 0.8, np.float64).reshape(3 * B, D)
    np.testing.assert_allclose(de.cpu().numpy(), wd, atol=TOL)


def test_hinge_inbatch_vs_oracle(cd):
    pairs = osynth.cowatch_pairs(40, 30, 2)            # tiny catalogue -> some invalid negatives
    B, D = 64, 32
    rng = np.random.RandomState(1)
    e = otower.l2_normalize(rng.randn(2 * B, D), np.float64)[0].astype(np.float32)
    n_invalid = 0
    for step in range(4):
        rows, tri, valid, s = osampler.device_inbatch(pairs, 3, step, B)
        n_invalid += int((valid == 0).sum())
        pos, neg, hinge = (torch.empty(B, device=cd.dev) for _ in range(3))
        v = torch.empty(B, dtype=torch.uint8, device=cd.dev)
        stats = torch.empty(4, device=cd.dev)
        de = torch.empty((2 * B, D), device=cd.dev)
        cd.ops.triplet_hinge_inbatch(dt(e, cd.dev), dt(rows, cd.dev, torch.int32),
                                     dt([s], cd.dev, torch.int32), B, D, 0.8, pos, neg, hinge, v,
                                     stats, de)
        np.testing.assert_array_equal(v.cpu().numpy(), valid)
        w = otower.hinge_loss_indexed(e.astype(np.float64), tri, valid.astype(bool), 0.8, np.float64)
        np.testing.assert_allclose(hinge.cpu().numpy(), w["hinge_dist"], atol=TOL)
        np.testing.assert_allclose(stats[0].item(), w["hinge_loss"], atol=TOL)
        wd = otower.hinge_loss_indexed_backward(e.astype(np.float64), tri, valid.astype(bool), 0.8,
                                                np.float64)
        np.testing.assert_allclose(de.cpu().numpy(), wd, atol=TOL)
    assert n_invalid > 0


@pytest.mark.parametrize("mode,B,D", [("uniform", 70, 256), ("inbatch", 64, 256), ("uniform", 33, 64),
                                      ("inbatch", 4096, 256), ("uniform", 4096, 256), ("inbatch", 50, 512),
                                      ("uniform", 9, 1024)])
def test_vnet_tail_fused_equals_separate_kernels_and_oracle(cd, mode, B, D):
    """The fused tail (l2norm -> hinge -> dE -> l2norm backward -> lrelu') gives the bits of the
    separate kernels, the oracle's values (fp64, 1e-5), the statistics and -- on request --
    build_graph's variance summary (train.py:67-71,151)."""
    rng = np.random.RandomState(B + D)
    rpt = 3 if mode == "uniform" else 2
    R = rpt * B
    z = (rng.randn(R, D) * 0.3 + 0.05).astype(np.float32)
    z[1] *= 1e-8                                            # a row below the 1e-12 clamp
    dz = dt(z, cd.dev)
    n_vid = 40 if B < 100 else 3000
    pairs = osynth.cowatch_pairs(n_vid, max(30, B // 8), 2)
    step = 1
    f = lambda *s: torch.full(s, -7.0, device=cd.dev)
    if mode == "uniform":
        rows = shift = None
        tri = np.arange(R).reshape(B, 3)
        valid = np.ones(B, bool)
    else:
        rows_np, tri, valid, sh = osampler.device_inbatch(pairs, 3, step, B)
        valid = valid.astype(bool)
        rows, shift = dt(rows_np, cd.dev, torch.int32), dt([sh], cd.dev, torch.int32)
    # separate kernels
    e0, de0, dz0 = f(R, D), f(R, D), f(R, D)
    p0, n0, h0, st0 = f(B), f(B), f(B), f(4)
    cd.ops.l2norm_fwd(dz, D, e0)
    if mode == "uniform":
        cd.ops.triplet_hinge(e0, B, D, 0.8, p0, n0, h0, st0, de0)
    else:
        cd.ops.triplet_hinge_inbatch(e0, rows, shift, B, D, 0.8, p0, n0, h0, None, st0, de0)
    cd.ops.l2norm_bwd(dz, de0, D, dz0, lrelu_alpha=0.2)
    # fused, twice (second time with the variance summary)
    var_ws = torch.zeros(cd.ops.vnet_tail_workspace_floats(B, D), device=cd.dev)
    for rep in range(2):
        e1, dz1 = f(R, D), f(R, D)
        p1, n1, h1, st1 = f(B), f(B), f(B), f(8)
        v1 = torch.full((B,), 9, dtype=torch.uint8, device=cd.dev)
        bf = torch.zeros((R, D), dtype=torch.bfloat16, device=cd.dev)
        cd.ops.vnet_tail(0 if mode == "uniform" else 1, dz, rows, shift, B, D, 0.8, e1, p1, n1, h1, dz1, valid=v1,
                         stats=st1, dz2_bf16=bf, var_ws=var_ws if rep else None)
        torch.cuda.synchronize()
        # same formulas; the compiler contracts multiply-adds differently in the two kernels, so
        # equal to a few ulp, not bit for bit
        close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
        assert close(e1, e0, 3e-7) and close(dz1, dz0, 2e-6), (float((e1 - e0).abs().max()), float((dz1 - dz0).abs().max()))
        assert close(p1, p0, 1e-6) and close(n1, n0, 1e-6) and close(h1, h0, 1e-6)
        assert torch.equal(bf, dz1.bfloat16())                               # round-to-nearest-even copy
        np.testing.assert_allclose(st1[:4].cpu().numpy(), st0.cpu().numpy(), rtol=2e-6, atol=1e-7)
        if mode == "inbatch":
            np.testing.assert_array_equal(v1.cpu().numpy().astype(bool), valid)
    # oracle (fp64 on the fp32 inputs)
    e64, inv = otower.l2_normalize(z.astype(np.float64), np.float64)
    np.testing.assert_allclose(e1.cpu().numpy(), e64, atol=TOL)
    w = otower.hinge_loss_indexed(e64, tri, valid, 0.8, np.float64)
    np.testing.assert_allclose(st1[0].item(), w["hinge_loss"], atol=TOL)
    np.testing.assert_allclose(h1.cpu().numpy(), w["hinge_dist"], atol=TOL)
    t = w["pos_dist"] - w["neg_dist"] + 0.8
    if np.abs(t).min() > 1e-5:                                               # nobody sits on the hinge
        dE = otower.hinge_loss_indexed_backward(e64, tri, valid, 0.8, np.float64)
        want = otower.leaky_relu_backward(z.astype(np.float64),
                                          otower.l2_normalize_backward(z.astype(np.float64), inv, dE, np.float64))
        scale = max(np.abs(want).max(), 1e-30)
        assert np.abs(dz1.cpu().numpy() - want).max() < 1e-5 * max(scale, 1.0)
    var = otower.calc_var(e64[tri], np.float64)
    np.testing.assert_allclose(st1[4].item(), var, rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------- optimizers --
def test_adam_advances_step_counter_in_the_same_launch(cd):
    """apply_gradients(global_step=...) (train.py:146): with advance_tickets the last block of
    the Adam launch stores step + 1; every block still used the OLD step for the bias correction."""
    rng = np.random.RandomState(5)
    n = 9 * 1000 * 1000                                    # > 2048 blocks x 256 x 4: all grid blocks busy
    w0 = torch.as_tensor(rng.randn(n).astype(np.float32)).to(cd.dev)
    g = torch.as_tensor((rng.randn(n) * 1e-3).astype(np.float32)).to(cd.dev)
    tick = cd.ops.new_tickets(cd.dev)
    res = []
    for adv in (False, True):
        w, m, v = w0.clone(), torch.zeros_like(w0), torch.zeros_like(w0)
        t_dev = torch.full((1,), 6, dtype=torch.int64, device=cd.dev)
        for _ in range(3):
            cd.ops.adam_step(w, g, m, v, 0.01, 1, t_dev=t_dev, advance_tickets=tick if adv else None)
            if not adv:
                cd.ops.step_advance(t_dev)
        torch.cuda.synchronize()
        assert int(t_dev.item()) == 9 and int(tick.abs().sum().item()) == 0
        res.append((w, m, v))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_grad_prepare_and_momentum_vs_oracle(cd):
    """build_graph's switches (train.py:115-116,133-145): l2 regulariser term + per-variable
    tf.clip_by_norm in place, and the Nesterov momentum update."""
    rng = np.random.RandomState(11)
    n = 70001
    w = rng.randn(n).astype(np.float32)
    g = (rng.randn(n) * 1e-2).astype(np.float32)
    scratch = torch.zeros(cd.ops.lars_scratch_floats(), device=cd.dev)
    for l2, clip in ((0.0, 1.0), (0.5, 0.0), (1e-8, 1.0), (0.3, 1e6), (0.0, 0.0)):
        dg, norms = dt(g, cd.dev), torch.zeros(2, device=cd.dev)
        cd.ops.grad_prepare(dg, dt(w, cd.dev), l2, clip, scratch, norms)
        want = g.astype(np.float64) + l2 * w.astype(np.float64)
        gn = np.sqrt((want ** 2).sum())
        np.testing.assert_allclose(norms.cpu().numpy(), [gn, (w.astype(np.float64) ** 2).sum()], rtol=1e-5)
        if clip > 0:
            want = otower.clip_by_norm(want, clip, np.float64)
            assert np.sqrt((want ** 2).sum()) <= clip * (1 + 1e-9)
        np.testing.assert_allclose(dg.cpu().numpy(), want, rtol=2e-6, atol=1e-9)
    acc = (rng.randn(n) * 1e-2).astype(np.float32)
    for nesterov in (True, False):
        dw, da = dt(w, cd.dev), dt(acc, cd.dev)
        lr_dev = torch.tensor([0.05], device=cd.dev)
        cd.ops.momentum_step(dw, dt(g, cd.dev), da, 123.0, 0.9, nesterov, lr_dev=lr_dev)   # lr_dev wins
        ww, wa = otower.momentum_step(w, g, acc, 0.05, 0.9, nesterov, np.float64)
        np.testing.assert_allclose(da.cpu().numpy(), wa, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(dw.cpu().numpy(), ww, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("precision,B", [("auto", 64), ("f32x3", 128), ("f16x2", 128)])
@pytest.mark.parametrize("optimizer", ["adam", "momentum"])
def test_train_step_build_graph_switches(cd, optimizer, precision, B):
    """clip_gradient_norm, regularization_penalty, the variance summary and the Nesterov
    momentum branch of build_graph (train.py:67-71,108-151) through TrainStep, against the oracle
    applied to the device's own raw gradients -- on the fp32 MFMA (B = 64: "auto" picks it) and on both plane forms."""
    N, F, H, D = 3000, 200, 300, 64
    feats, pairs = _clustered(N, F, 12, 3)
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    ts = cd.train.TrainStep(table, dt(pairs, cd.dev, torch.int32), B, hidden_size=H, output_size=D, mode="uniform",
                            optimizer=optimizer, base_learning_rate=0.01, clip_gradient_norm=0.05,
                            regularization_penalty=1e5, device=cd.dev, precision=precision)      # penalty*1e-8 = 1e-3: visible
    assert ts.precision == ("f32" if precision == "auto" else precision)
    ts.enable_variance()
    host = lambda ts_: [t.detach().cpu().numpy().astype(np.float64) for t in ts_]
    clipped = 0
    for step in range(3):
        W = host(ts.params.unpadded())
        slots = host(ts.params._views(ts.m)) + host(ts.params._views(ts.v)) if optimizer == "adam" \
            else host(ts.params._views(ts.acc))
        ts.fetch(); ts.forward_loss(); ts.backward()
        raw = dict(zip(("dW1", "db1", "dW2", "db2"), host(ts.params.unpadded(grads=True))))
        idx = osampler.device_triplets_vec(pairs, N, 1234, step, B)
        fwd, loss, og = otower.train_step_grads(feats[idx.reshape(-1)].astype(np.float64), W, 0.8, np.float64)
        for k in raw:
            assert np.abs(raw[k] - og[k]).max() < 1e-5
        trip = fwd["l2_norm"].reshape(B, 3, D)
        np.testing.assert_allclose(ts.variance(), otower.calc_var(trip, np.float64), rtol=1e-5)   # train.py:67-71
        ts.apply_gradients()
        ts.global_step += 1
        torch.cuda.synchronize()
        want_g, reg = otower.regularized_grads(raw, W, 1e5, dtype=np.float64)
        np.testing.assert_allclose(ts.reg_loss(), reg, rtol=1e-5)
        L = ts.layout
        sl = ((slice(0, L.F), slice(0, L.H)), (slice(0, L.H),), (slice(0, L.H), slice(0, L.D)), (slice(0, L.D),))
        for i, k in enumerate(("dW1", "db1", "dW2", "db2")):
            clipped += int(np.sqrt((want_g[k] ** 2).sum()) > 0.05)
            gk = otower.clip_by_norm(want_g[k], 0.05, np.float64)
            got_g = ts.params.unpadded(grads=True)[i].cpu().numpy()
            np.testing.assert_allclose(got_g, gk, rtol=1e-5, atol=1e-8)
            if optimizer == "adam":
                w, _, _ = otower.adam_step(W[i], gk, slots[i][sl[i]], slots[4 + i][sl[i]], step + 1, 0.01, dtype=np.float64)
            else:
                w, _ = otower.momentum_step(W[i], gk, slots[i][sl[i]], 0.01, 0.9, True, np.float64)
            assert np.abs(ts.params.unpadded()[i].cpu().numpy() - w).max() < 2e-6, (k, step)
    assert clipped >= 3 and int(ts.step_dev.item()) == 3
    assert float(ts.params.W1[ts.layout.F:].abs().max()) == 0                 # padding stays zero


def test_adam_vs_oracle(cd):
    rng = np.random.RandomState(0)
    n = 1003
    w, g = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32) * 1e-3
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    dw, dm, dv = dt(w, cd.dev), dt(m, cd.dev), dt(v, cd.dev)
    t_dev = torch.zeros(1, dtype=torch.int64, device=cd.dev)
    lr_dev = torch.full((1,), 0.01, device=cd.dev)
    for t in range(1, 4):
        g = (g * 0.9 + 1e-4).astype(np.float32)
        if t < 3:
            cd.ops.adam_step(dw, dt(g, cd.dev), dm, dv, 0.01, t)
        else:                                          # step and lr from device memory
            t_dev.fill_(t - 1)
            cd.ops.adam_step(dw, dt(g, cd.dev), dm, dv, 0.0, 1, lr_dev=lr_dev, t_dev=t_dev)
        w, m, v = otower.adam_step(w, g, m, v, t, 0.01, dtype=np.float32)
        np.testing.assert_allclose(dw.cpu().numpy(), w, atol=1e-6)
        np.testing.assert_allclose(dm.cpu().numpy(), m, rtol=2e-6, atol=1e-12)
        np.testing.assert_allclose(dv.cpu().numpy(), v, rtol=2e-6, atol=1e-15)


def test_lars_vs_oracle(cd):
    rng = np.random.RandomState(0)
    n = 5000
    w, g = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32) * 0.1
    acc = np.zeros(n, np.float32)
    dw, da = dt(w, cd.dev), dt(acc, cd.dev)
    scratch = torch.zeros(cd.ops.lars_scratch_floats(), device=cd.dev)
    for _ in range(3):
        cd.ops.lars_step(dw, dt(g, cd.dev), da, 1.0, scratch)
        w, acc = otower.lars_step(w, g, acc, 1.0, dtype=np.float64)
        np.testing.assert_allclose(dw.cpu().numpy(), w, atol=1e-6)
    z = torch.zeros(16, device=cd.dev)                 # |w| == 0 -> trust 1
    cd.ops.lars_step(z, dt(np.ones(16), cd.dev), torch.zeros(16, device=cd.dev), 1.0, scratch)
    np.testing.assert_allclose(z.cpu().numpy(), -1.0)


def test_lars_multi_vs_oracle_and_single_variable_kernel(cd):
    """The training step's LARS: every variable of the flat buffer in two launches (one trust ratio per
    segment, the reference's tf.contrib LARSOptimizer on weights and biases alike, train.py:354) ==
    the oracle per variable, == cdml_lars_step per variable, and the last block advances the counter."""
    rng = np.random.RandomState(1)
    sizes = [64 * 130, 128, 128 * 64, 64]                       # W1, b1, W2, b2-like; multiples of 4
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]])
    n = int(sum(sizes))
    w = rng.randn(n).astype(np.float32) * np.repeat([1.0, 0.1, 0.5, 0.0], sizes).astype(np.float32)   # last: |w| = 0
    g = (rng.randn(n) * 0.05).astype(np.float32)
    acc = np.zeros(n, np.float32)
    segs = list(zip(offs.tolist(), sizes))
    dw, da = dt(w, cd.dev), dt(acc, cd.dev)
    dw1, da1 = dw.clone(), da.clone()
    scratch = torch.zeros(max(cd.ops.lars_multi_scratch_floats(), cd.ops.lars_scratch_floats()), device=cd.dev)
    norms = torch.zeros(8, device=cd.dev)
    step = torch.zeros(1, dtype=torch.int64, device=cd.dev)
    tickets = cd.ops.new_tickets(cd.dev)
    for it in range(3):
        dg = dt(g, cd.dev)
        cd.ops.lars_multi(dw, dg, da, segs, 1.0, scratch, norms_out=norms, step_dev=step, tickets=tickets)
        for o, m in segs:
            cd.ops.lars_step(dw1[o:o + m], dg[o:o + m], da1[o:o + m], 1.0, scratch)
            w[o:o + m], acc[o:o + m] = otower.lars_step(w[o:o + m], g[o:o + m], acc[o:o + m], 1.0, dtype=np.float64)
        np.testing.assert_allclose(dw.cpu().numpy(), w, atol=1e-6)
        np.testing.assert_allclose(da.cpu().numpy(), acc, atol=1e-6)
        np.testing.assert_allclose(dw.cpu().numpy(), dw1.cpu().numpy(), atol=2e-7)
        g = (rng.randn(n) * 0.05).astype(np.float32)
    assert int(step.item()) == 3
    got = norms.cpu().numpy().reshape(4, 2)
    assert got[3, 1] > 0 and np.all(got[:, 0] > 0)              # (|w| of the last segment is no longer 0 after a step)
    with pytest.raises(Exception):
        cd.ops.lars_multi(dw, dt(g, cd.dev), da, [(0, 6), (6, n - 6)], 1.0, scratch)     # not multiples of 4


# ------------------------------------------------------------ end-to-end step --
def _oracle_step(feats, pairs, W, step, B, mode, seed, margin=0.8):
    """Embeddings, loss and gradients of one reference step from weights W (fp64)."""
    N = len(feats)
    if mode == "uniform":
        idx = osampler.device_triplets_vec(pairs, N, seed, step, B)
        fwd, loss, grads = otower.train_step_grads(feats[idx.reshape(-1)], W, margin, np.float64)
    else:
        rows, tri, valid, _ = osampler.device_inbatch(pairs, seed, step, B)
        fwd = otower.vnet_forward(feats[rows], *W, dtype=np.float64)
        loss = otower.hinge_loss_indexed(fwd["l2_norm"], tri, valid.astype(bool), margin, np.float64)
        dE = otower.hinge_loss_indexed_backward(fwd["l2_norm"], tri, valid.astype(bool), margin,
                                                np.float64)
        grads = otower.vnet_backward(fwd, W[2], dE, np.float64)
    return fwd["l2_norm"], float(loss["hinge_loss"]), grads


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f16x2"])
@pytest.mark.parametrize("mode,optimizer", [("uniform", "adam"), ("inbatch", "adam"),
                                            ("uniform", "lars")])
def test_train_steps_config0(cd, mode, optimizer, precision):
    """BASELINE config 0 shape: 10k x 1500 catalogue (imitation_data-shaped), 5000
    hidden, 256-d, batch 128.  Every step is checked from the device's own
    weights (Adam amplifies 1e-9 gradient noise near g = 0 into lr-sized update
    differences, so free-running fp32 and fp64 trajectories are not comparable):
    embeddings / loss / gradients against the fp64 oracle, and the optimizer
    against the oracle's update applied to the device's gradients and slots."""
    N, F, B, D = 10000, 1500, 128, 256
    feats = osynth.features_numpy(N, F, seed=0).astype(np.float32)
    pairs = osynth.cowatch_pairs(N, 3000, 0)
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    lr = 0.01 if optimizer == "adam" else 1.0
    # (precision "f32x3": the same products from three bf16 planes per operand on the bf16 MFMA -- same bounds; "f16x2": from
    # two fp16 planes under per-tensor scales on the fp16 MFMA -- same bounds)
    ts = cd.train.TrainStep(table, dt(pairs, cd.dev, torch.int32), B, margin=0.8, mode=mode,
                            optimizer=optimizer, base_learning_rate=lr, device=cd.dev, precision=precision)
    f64 = feats.astype(np.float64)
    host = lambda ts_: [t.detach().cpu().numpy().copy() for t in ts_]
    names = ("dW1", "db1", "dW2", "db2")
    for step in range(3):
        W = host(ts.params.unpadded())
        if optimizer == "adam":
            slots = [host(ts.params._views(ts.m)), host(ts.params._views(ts.v))]
        else:
            slots = [host(ts.params._views(ts.acc))]
        ts.step()
        we, wl, wg = _oracle_step(f64, pairs, [w.astype(np.float64) for w in W], step, B, mode, 1234)
        e = ts.ws.e[:, :D].cpu().numpy()
        assert np.abs(e - we).max() < TOL, f"embeddings step {step}"
        assert abs(ts.loss() - wl) < TOL, f"loss step {step}"
        np.testing.assert_allclose(np.linalg.norm(e, axis=1), 1.0, atol=1e-5)
        G = host(ts.params.unpadded(grads=True))
        for got, k in zip(G, names):
            scale = max(np.abs(wg[k]).max(), 1e-30)
            # embeddings of iid-uniform features nearly coincide, so the gradient is a sum of
            # cancelling terms: fp32 accumulation noise is a few % of its (tiny, <1e-3) magnitude
            assert np.abs(got - wg[k]).max() < max(TOL, 5e-2 * scale), f"{k} step {step}"
        L = ts.layout
        sl = ((slice(0, L.F), slice(0, L.H)), (slice(0, L.H),), (slice(0, L.H), slice(0, L.D)),
              (slice(0, L.D),))
        for i, got in enumerate(host(ts.params.unpadded())):
            if optimizer == "adam":
                w, _, _ = otower.adam_step(W[i], G[i], slots[0][i][sl[i]], slots[1][i][sl[i]], step + 1,
                                           lr, dtype=np.float32)
            else:
                w, _ = otower.lars_step(W[i], G[i], slots[0][i][sl[i]], lr, dtype=np.float32)
            assert np.abs(got - w).max() < 1e-6, f"optimizer var {i} step {step}"
    # padded storage stays exactly zero
    L = ts.layout
    assert float(ts.params.W1[L.F:].abs().max()) == 0 and float(ts.params.W1[:, L.H:].abs().max()) == 0
    assert float(ts.params.b1[L.H:].abs().max()) == 0 and float(ts.params.W2[L.H:].abs().max()) == 0
    assert int(ts.step_dev.item()) == 3


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f16x2"])
def test_graph_replay_equals_eager(cd, precision):
    """(f16x2: the plane scales are kernel arguments baked into a capture -- its check steps 0, 1, 2, 4, 8 run eagerly inside the
    graph run, the others replay, and a scale that moved on a check step drops the graphs recorded before it.)"""
    N, F, B = 4000, 200, 64 if precision == "f32" else 128
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = dt(osynth.cowatch_pairs(N, 500, 0), cd.dev, torch.int32)
    kw = dict(hidden_size=300, output_size=64, mode="uniform", device=cd.dev, precision=precision)
    a = cd.train.TrainStep(table, pairs, B, use_graph=False, **kw)
    b = cd.train.TrainStep(table, pairs, B, use_graph=True, **kw)
    n = 12 if precision == "f16x2" else 4                # (f16x2: past the dense early checks, so that steps DO replay)
    for _ in range(n):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.params.flat, b.params.flat)
    assert torch.equal(a.idx, b.idx) and int(b.step_dev.item()) == n
    if precision == "f16x2":
        assert len(b._graphs) >= 1 and a.ws.scales.state() == b.ws.scales.state()


def test_train_config_builds_the_same_step(cd):
    """config.TrainConfig (the reference's scattered hyper-parameters, train.py:358-373 / 212-222, as one object)
    builds the step and the loop the explicit arguments build."""
    from cdml_amd.config import TrainConfig
    N, F, B = 4000, 200, 128
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = dt(osynth.cowatch_pairs(N, 500, 0), cd.dev, torch.int32)
    cfg = TrainConfig(batch_size=B, hidden_size=300, output_size=64, num_epochs=2)
    a = cfg.train_step(table, pairs, device=cd.dev)
    b = cd.train.TrainStep(table, pairs, B, hidden_size=300, output_size=64, mode="uniform", optimizer="lars",
                           base_learning_rate=1.0, margin=0.8, precision="f32x3", device=cd.dev)
    for _ in range(3):
        a.step()
        b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.params.flat, b.params.flat) and a.loss() == b.loss()
    tr = cfg.trainer(a, 500)
    assert tr.num_batches == (500 * 2) // B and tr.require_improve_num == 40 and tr.best_eval_dist == 1.0
    with pytest.raises(ValueError):
        TrainConfig(model="VENet").train_step(table, pairs, device=cd.dev)


# --------------------------------------------------------------------- facade --
def test_facade_matches_train_step(cd):
    """models.VNet / losses.HingeLoss / inputs.MPTripletPipe used the way
    train.py:105-146 uses them give the fused step's numbers."""
    N, F, B = 3000, 1500, 32
    feats = osynth.features_numpy(N, F, seed=1).astype(np.float32)
    pairs = osynth.cowatch_pairs(N, 400, 1)
    pipe = cd.inputs.MPTripletPipe(pairs=pairs, table=feats, device=cd.dev, seed=1234)
    assert pipe.cowatch_num == len(pairs)
    pipe.create_pipe(num_epochs=1, batch_size=B)
    batch = pipe.get_batch()
    assert tuple(batch.shape) == (B, 3, F) and batch.dtype == torch.float32
    idx = osampler.device_triplets_vec(pairs, N, 1234, 0, B)
    np.testing.assert_array_equal(batch.cpu().numpy(), feats[idx])       # inputs.py:158

    model = cd.utils.find_class_by_name("VNet", [cd.models])(device=cd.dev, seed=42)
    loss_fn = cd.utils.find_class_by_name("HingeLoss", [cd.losses])()
    result = model.create_model(batch.reshape(-1, F), 256)               # train.py:105,313
    assert set(result) == {"layer_1", "layer_2", "l2_norm"}
    assert tuple(result["layer_1"].shape) == (3 * B, 5000)
    trip = result["l2_norm"].reshape(-1, 3, 256)                         # train.py:128
    out = loss_fn.calculate_loss(trip, margin=0.8)
    out["hinge_loss"].backward()

    ts = cd.train.TrainStep(pipe.table, pipe.pairs, B, margin=0.8, mode="uniform", device=cd.dev)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    # two kernel routes to the same numbers (fused gather+norm vs norm of a raw batch)
    assert float((result["l2_norm"] - ts.ws.e[:, :256]).abs().max()) < 1e-6
    assert abs(out["hinge_loss"].item() - ts.loss()) < 1e-6
    assert float((model.params.flat.grad - ts.params.grad).abs().max()) < 1e-6
    with pytest.raises(StopIteration):
        cd.utils.find_class_by_name("VedeNet", [cd.models])              # train.py:43 default


def test_pipe_exhaustion_and_replay(cd):
    feats = np.random.RandomState(0).random_sample((20, 8)).astype(np.float32)
    pairs = np.array([[i, (i + 1) % 20] for i in range(10)], dtype=np.int32)
    pipe = cd.inputs.MPTripletPipe(pairs=pairs, table=feats, device=cd.dev)
    pipe.create_pipe(num_epochs=3, batch_size=4)             # 30 pairs -> 7 whole batches
    n = 0
    while True:
        b = pipe.get_batch()
        if b is None:
            break
        n += 1
        assert tuple(b.shape) == (4, 3, 8)
    assert n == 7 and pipe.get_batch() is None
    # replay mode: reference-identical triplets supplied by the caller
    ref = osampler.reference_triplets(pairs[:4], 20, seed=1234)
    b = pipe.get_batch(indices=ref)
    np.testing.assert_array_equal(b.cpu().numpy(), feats[ref])
    pipe.check_indices()
    pipe.get_batch(indices=[[0, 1, 25]])
    with pytest.raises(IndexError):
        pipe.check_indices()


def test_memory_triplet_pipe(cd):
    """inputs.TripletPipe: repeat -> batch -> shuffle(batches) like the reference's tf.data pipe."""
    from cdml_amd import inputs
    trip = np.arange(7 * 3 * 4, dtype=np.float32).reshape(7, 3, 4)
    it = inputs.TripletPipe(trip, device=cd.dev).create_pipe(batch_size=3, num_epochs=2, buffer_size=1)
    got = [b.cpu().numpy() for b in it]                       # buffer 1: stream order
    assert [len(b) for b in got] == [3, 3, 3, 3, 2]           # 14 elements, a batch straddles the passes
    np.testing.assert_array_equal(np.concatenate(got), np.concatenate([trip, trip]))
    it = inputs.TripletPipe(trip, device=cd.dev, seed=1).create_pipe(batch_size=2, num_epochs=3, buffer_size=4)
    shuf = [b.cpu().numpy() for b in it]
    assert sorted(len(b) for b in shuf) == [1] + [2] * 10
    key = lambda bs: sorted(tuple(x.ravel()) for b in bs for x in b)
    assert key(shuf) == key([trip, trip, trip])               # same multiset of elements
    assert any(not np.array_equal(a, b) for a, b in zip(shuf, [np.concatenate([trip] * 3)[i:i + 2] for i in range(0, 21, 2)]))
    with pytest.raises(StopIteration):
        it.get_next()
    forever = inputs.TripletPipe(trip, device=cd.dev).create_pipe(batch_size=5, num_epochs=None, buffer_size=2)
    assert all(forever.get_next().shape == (5, 3, 4) for _ in range(10))
    # the reference's own use (tests/test_inputs.py): row-id triplets through the pipe, then lookup
    from cdml_amd import parse_data
    table = cd.engine.FeatureTable.synthetic(50, 96, 0, cd.dev)
    ids = np.random.RandomState(0).randint(0, 50, size=(9, 3))
    batch = inputs.TripletPipe(ids, device=cd.dev).create_pipe(batch_size=2, num_epochs=None, buffer_size=1).get_next()
    assert batch.shape == (2, 3) and np.array_equal(batch.cpu().numpy(), ids[:2])
    feats = parse_data.lookup(batch, table)
    assert feats.shape == (2, 3, 96)
    np.testing.assert_array_equal(feats.cpu().numpy(), table.data[ids[:2].reshape(-1), :96].cpu().numpy().reshape(2, 3, 96))
    with pytest.raises(IndexError):
        parse_data.lookup(torch.tensor([[0, 1, 50]]), table)


def test_errors_are_raised_not_swallowed(cd):
    y = torch.empty((8, 64), device=cd.dev)
    x = torch.empty((8, 48), device=cd.dev)
    W = torch.empty((48, 64), device=cd.dev)
    b = torch.empty(64, device=cd.dev)
    with pytest.raises(cd.pkg.CdmlError) as ei:
        cd.ops.fc_lrelu_fwd(x, W, b, y, 8, 48, 64)           # K not a multiple of 32
    assert ei.value.code == -4 and "multiple of 32" in str(ei.value)
    with pytest.raises(cd.pkg.CdmlError):
        cd.ops.fc_lrelu_fwd(x[:, 1:33], W[:32], b, y, 8, 32, 64)    # misaligned base
    with pytest.raises(ValueError):
        cd.ops.l2norm_fwd(torch.empty((4, 8)), 8, torch.empty((4, 8)))   # CPU tensors: no fallback
    with pytest.raises(cd.pkg.CdmlError):
        cd.ops.sample_inbatch(torch.zeros((4, 2), dtype=torch.int32, device=cd.dev), 1, 0, 1,
                              torch.zeros(2, dtype=torch.int32, device=cd.dev),
                              torch.zeros(1, dtype=torch.int32, device=cd.dev))


# ------------------------------------------------ full-size properties (c1) ----
@pytest.mark.parametrize("precision", ["f32x3", "f32"])
def test_full_size_properties_config1(cd, precision):
    """BASELINE config 1 sizes (scaled catalogue: the properties do not depend on
    row count): B=4096 in-batch step on a 200k x 1500 table."""
    N, F, B = 200000, 1500, 4096
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs_np = osynth.cowatch_pairs(N, 40000, 0)
    ts = cd.train.TrainStep(table, dt(pairs_np, cd.dev, torch.int32), B, mode="inbatch", device=cd.dev,
                            precision=precision)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    x_hat, dz1 = (ts.ws.x_hat_f32(), ts.ws.dz1_f32()) if ts.x3 else (ts.ws.x_hat, ts.ws.dz1)
    rows = ts.idx.cpu().numpy()
    q = np.arange(B) % len(pairs_np)
    np.testing.assert_array_equal(rows.reshape(B, 2), pairs_np[q])           # bit-exact ids
    xn = x_hat[:, :F].norm(dim=1)
    assert float((xn - 1).abs().max()) < 1e-5                                 # unit rows in
    en = ts.ws.e.norm(dim=1)
    assert float((en - 1).abs().max()) < 1e-5                                 # unit rows out
    # gathered rows are the table rows, normalised (spot check vs the oracle)
    sel = np.array([0, 1, 4095, 8191])
    raw = np.concatenate([osynth.features_philox(int(r), 1, F, 0) for r in rows[sel]])
    want = otower.l2_normalize(raw, np.float32)[0]
    np.testing.assert_allclose(x_hat[sel, :F].cpu().numpy(), want, atol=1e-6)
    # loss/gradient consistency: dE rows sum to ~0 per triplet group and the
    # weight gradient equals x_hat^T dz1 recomputed by torch on a column slab
    assert torch.isfinite(ts.params.grad).all()
    ref = x_hat.double().T @ dz1[:, :256].double()
    assert float((ts.params.gW1[:, :256].double() - ref).abs().max()) < 1e-5
    sub = otower.vnet_forward(x_hat[:64, :F].cpu().numpy().astype(np.float64),
                              *[t.cpu().numpy().astype(np.float64) for t in ts.params.unpadded()],
                              dtype=np.float64)
    # x_hat is already unit norm, so re-normalising it is the identity to 1e-7
    assert np.abs(ts.ws.e[:64, :256].cpu().numpy() - sub["l2_norm"]).max() < TOL
    # determinism: a second pass over the same step is bit-identical
    g0 = ts.params.grad.clone()
    ts.fetch(); ts.forward_loss(); ts.backward()
    assert torch.equal(g0, ts.params.grad)


# ------------------------------------------------ semi-hard mining (config 2) ----
def test_semihard_select_and_indexed_loss(cd):
    """Build-defined (no reference counterpart): spec = oracle.tower.semihard_select.
    Selection is checked tolerance-aware: the device may break a < 2e-6 near-tie
    differently, so its choice must be as good as the oracle's within that band."""
    rng = np.random.RandomState(3)
    B, D = 96, 64
    for trial, n_videos in enumerate((500, 12)):           # 12 videos: many ineligible rows, some -1
        E = otower.l2_normalize(rng.randn(2 * B, D) + (0.0 if trial else 2.0), np.float64)[0].astype(np.float32)
        rows = rng.randint(0, n_videos, size=2 * B).astype(np.int32)
        if trial:
            rows[:8] = 0; rows[8:] = rng.randint(0, 2, size=2 * B - 8)   # anchors with nothing eligible
        de_, dr = dt(E, cd.dev), dt(rows, cd.dev, torch.int32)
        S = torch.empty((B, 2 * B), device=cd.dev)
        cd.ops.fc_bwd_data(de_[0::2], de_, None, S, B, 2 * B, D)
        np.testing.assert_allclose(S.cpu().numpy(), E[0::2].astype(np.float64) @ E.T.astype(np.float64), atol=1e-5)
        neg_row = torch.empty(B, dtype=torch.int32, device=cd.dev)
        cd.ops.semihard_select(S, de_, dr, B, D, torch.empty(2 * B, device=cd.dev), neg_row)
        got = neg_row.cpu().numpy()
        want, dist = otower.semihard_select(E.astype(np.float64), rows)
        d_p = dist[np.arange(B), 2 * np.arange(B) + 1]
        tol = 2e-6
        for i in range(B):
            if want[i] < 0:
                assert got[i] == -1
                continue
            assert got[i] >= 0 and rows[got[i]] not in (rows[2 * i], rows[2 * i + 1])
            elig = (rows != rows[2 * i]) & (rows != rows[2 * i + 1])
            strict = elig & (dist[i] > d_p[i] + tol)  # clearly outside, whatever the rounding
            if dist[i, got[i]] > d_p[i] - tol and (strict.any() or dist[i, want[i]] > d_p[i]):
                # semi-hard branch: at least as close as the closest clearly-outside candidate
                if strict.any():
                    assert dist[i, got[i]] <= dist[i][strict].min() + tol
            else:                                     # none outside: farthest eligible
                assert dist[i, got[i]] >= dist[i][elig].max() - tol
        assert (got == want).mean() > 0.95
        if trial:
            assert (want < 0).any()
        # loss + gradient on the DEVICE's selection vs the oracle's indexed loss
        tri, valid = otower.semihard_triplets(got)
        pos, neg, hinge, scale = (torch.empty(B, device=cd.dev) for _ in range(4))
        stats = torch.empty(4, device=cd.dev)
        dE = torch.empty((2 * B, D), device=cd.dev)
        cd.ops.triplet_hinge_indexed(de_, neg_row, B, D, 0.8, pos, neg, hinge, scale, stats, dE)
        w = otower.hinge_loss_indexed(E.astype(np.float64), tri, valid, 0.8, np.float64)
        np.testing.assert_allclose(hinge.cpu().numpy(), w["hinge_dist"], atol=TOL)
        np.testing.assert_allclose(stats[0].item(), w["hinge_loss"], atol=TOL)
        wd = otower.hinge_loss_indexed_backward(E.astype(np.float64), tri, valid, 0.8, np.float64)
        np.testing.assert_allclose(dE.cpu().numpy(), wd, atol=TOL)
        dE2 = torch.empty_like(dE)
        cd.ops.triplet_hinge_indexed(de_, neg_row, B, D, 0.8, pos, neg, hinge, scale, stats, dE2)
        assert torch.equal(dE, dE2)                   # deterministic accumulation


@pytest.mark.parametrize("B,D,skew", [(1500, 256, "uniform"), (1500, 256, "one-row"), (2051, 64, "few-rows"), (96, 128, "masked")])
def test_indexed_hinge_backward_block_scan_and_fused_tail(cd, B, D, skew):
    """Round 6: the indexed hinge's backward as a block-cooperative scan (a block of 64 rows scans neg_row once and walks
    its hits in ascending triplet order) with the rest of config 2's tail folded in.  Against the fp64 oracle on any
    distribution of mined rows -- uniform, EVERY anchor mining one row (one wave adds B terms in order), a handful of
    rows, masked triplets; batch sizes that are no multiple of 4 or of the 1 024-triplet pass -- and the fused tail
    (l2norm backward + leaky-relu' + bf16 copy / three planes) equal to the separate launches -- the row gradients bit for
    bit, dz2 to a few ulp (fma contraction differs between kernels), its planes exactly those of the dz2 it wrote
    (losses.py:32-38, models.py:61, train.py:141)."""
    rng = np.random.RandomState(B + D)
    Z = (rng.randn(2 * B, D) * 0.3).astype(np.float32)
    E = otower.l2_normalize(Z.astype(np.float64), np.float64)[0].astype(np.float32)
    if skew == "uniform":
        nr = rng.randint(0, 2 * B, size=B)
    elif skew == "one-row":
        nr = np.full(B, 77)
    elif skew == "few-rows":
        nr = rng.choice([3, 64, 65, 1000, 2 * B - 1], size=B)
    else:
        nr = rng.randint(0, 2 * B, size=B)
        nr[rng.rand(B) < 0.4] = -1
    nr = nr.astype(np.int32)
    dz_, de_, dn = dt(Z, cd.dev), dt(E, cd.dev), dt(nr, cd.dev, torch.int32)
    pos, neg, hinge, scale = (torch.empty(B, device=cd.dev) for _ in range(4))
    stats = torch.empty(4, device=cd.dev)
    dE = torch.empty((2 * B, D), device=cd.dev)
    cd.ops.triplet_hinge_indexed(de_, dn, B, D, 0.8, pos, neg, hinge, scale, stats, dE)
    tri, valid = otower.semihard_triplets(nr)
    w = otower.hinge_loss_indexed(E.astype(np.float64), tri, valid, 0.8, np.float64)
    np.testing.assert_allclose(hinge.cpu().numpy(), w["hinge_dist"], atol=TOL)
    wd = otower.hinge_loss_indexed_backward(E.astype(np.float64), tri, valid, 0.8, np.float64)
    # (a triplet within rounding of the hinge's corner may be active on one side only: compare where the oracle is clear)
    t = w["pos_dist"] - w["neg_dist"] + 0.8 if "pos_dist" in w else None
    got = dE.cpu().numpy()
    if t is not None and (np.abs(np.ravel(t)) < 1e-5).any():
        pytest.skip("a triplet sits on the hinge's corner in this draw")
    np.testing.assert_allclose(got, wd, atol=5e-5 if skew == "one-row" else TOL)     # (one row sums B terms: fp32 accumulation)
    # the separate launches of the tail ...
    dz2_a = torch.empty((2 * B, D), device=cd.dev)
    cd.ops.l2norm_bwd(dz_, dE, D, dz2_a, lrelu_alpha=cd.ops.LRELU_ALPHA)
    pl_a = torch.zeros((2 * B, 3 * D), dtype=torch.bfloat16, device=cd.dev)
    cd.ops.split_f32_bf16x3(dz2_a, pl_a, D)
    # ... against the fused form, three planes and plain bf16
    for planes in (True, False):
        dE_b, dz2_b = torch.empty_like(dE), torch.empty_like(dz2_a)
        bf = torch.zeros((2 * B, 3 * D if planes else D), dtype=torch.bfloat16, device=cd.dev)
        cd.ops.triplet_hinge_indexed(de_, dn, B, D, 0.8, pos, neg, hinge, scale, stats, dE_b, z=dz_, dz2=dz2_b, dz2_bf16=bf,
                                     plane_bf=D if planes else 0)
        assert torch.equal(dE_b, dE)
        # the l2norm backward inside another kernel: the same operations, but which multiply-adds the compiler contracts
        # into fmas differs from kernel to kernel -- a few ulp, as between k_vnet_tail and its separate kernels
        # (per row, against the size of what is subtracted there: dz2 = (g - z <z,g> / |z|^2) / |z| cancels on hub rows)
        row_scale = dE.abs().max(dim=1).values / dz_.norm(dim=1)
        assert bool(((dz2_b - dz2_a).abs().max(dim=1).values <= 1e-6 * row_scale + 1e-12).all())
        if planes:                                               # the planes are those of THIS launch's dz2, exactly
            assert torch.equal(bf[:, :D].float() + bf[:, D:2 * D].float() + bf[:, 2 * D:].float(), dz2_b)
            chk = torch.zeros_like(bf)
            cd.ops.split_f32_bf16x3(dz2_b, chk, D)
            assert torch.equal(bf, chk)
        else:
            assert torch.equal(bf, dz2_b.to(torch.bfloat16))


def _check_semihard_choice(got, want, dist, rows, B, tol=2e-6):
    """The device's negatives against the oracle's, tolerance-aware (a candidate within tol of d_p may fall on either side)."""
    d_p = dist[np.arange(B), 2 * np.arange(B) + 1]
    for i in range(B):
        if want[i] < 0:
            assert got[i] == -1
            continue
        assert got[i] >= 0 and rows[got[i]] not in (rows[2 * i], rows[2 * i + 1])
        elig = (rows != rows[2 * i]) & (rows != rows[2 * i + 1])
        strict = elig & (dist[i] > d_p[i] + tol)
        if dist[i, got[i]] > d_p[i] - tol and (strict.any() or dist[i, want[i]] > d_p[i]):
            if strict.any():
                assert dist[i, got[i]] <= dist[i][strict].min() + tol
        else:
            assert dist[i, got[i]] >= dist[i][elig].max() - tol


@pytest.mark.parametrize("h2", [False, True])
@pytest.mark.parametrize("B,D", [(128, 64), (384, 256), (640, 128)])
def test_semihard_mine_fused_into_the_score_product(cd, B, D, h2):
    """BASELINE config 2 on the plane kernels: cdml_semihard_mine_x3 computes the B x 2B score product as six bf16 plane
    products per fp32 product and selects in its epilogue -- no score matrix.  Spec = oracle.tower.semihard_select, checked
    with the tolerance of the unfused kernel's test; B = 384 / 640: a last row tile of 128 anchors; few videos: rows with
    nothing eligible (-1) and the farthest-eligible fallback.  h2: the same on the fp16 build of the kernel -- two fp16 planes of
    the unit rows times 2^14, three plane products (cdml_semihard_mine_h2; precision "f16x2")."""
    hs = 2.0 ** 14 if h2 else 0.0
    rng = np.random.RandomState(B + D)
    for trial, n_videos in enumerate((5000, 12)):
        E = otower.l2_normalize(rng.randn(2 * B, D) + (0.0 if trial else 2.0), np.float64)[0].astype(np.float32)
        rows = rng.randint(0, n_videos, size=2 * B).astype(np.int32)
        if trial:
            rows[:8] = 0; rows[8:] = rng.randint(0, 2, size=2 * B - 8)
        de_, dr = dt(E, cd.dev), dt(rows, cd.dev, torch.int32)
        e3 = (torch.zeros((2 * B, 2 * D), dtype=torch.float16, device=cd.dev) if h2
              else torch.zeros((2 * B, 3 * D), dtype=torch.bfloat16, device=cd.dev))
        sqn, dpd = torch.zeros(2 * B, device=cd.dev), torch.zeros(B, device=cd.dev)
        wsb = cd.ops.semihard_mine_x3_workspace(B)
        assert wsb == (2 * B // 256) * 4 * B * 16
        wsp = torch.full((wsb // 4,), float("nan"), device=cd.dev)
        neg_row = torch.full((B,), -7, dtype=torch.int32, device=cd.dev)
        cd.ops.semihard_mine_x3(de_, dr, B, D, e3, D, sqn, dpd, wsp, neg_row, h2_scale=hs)
        got = neg_row.cpu().numpy()
        want, dist = otower.semihard_select(E.astype(np.float64), rows)
        # the prep pass: exact planes (fp16 pair: 22 bits of them), squared norms and positive distances
        if h2:
            assert ((e3[:, :D].double() + e3[:, D:].double()) / hs - de_.double()).abs().max().item() <= 2.0 ** -22
        else:
            assert torch.equal(e3[:, :D].float() + e3[:, D:2 * D].float() + e3[:, 2 * D:].float(), de_)
        np.testing.assert_allclose(sqn.cpu().numpy(), (E.astype(np.float64) ** 2).sum(1), atol=1e-6)
        np.testing.assert_allclose(dpd.cpu().numpy(), dist[np.arange(B), 2 * np.arange(B) + 1], atol=2e-6)
        _check_semihard_choice(got, want, dist, rows, B)
        assert (got == want).mean() > 0.95
        if trial:
            assert (want < 0).any()
        # and the unfused kernels on the same input agree (both within rounding of the oracle)
        S = torch.empty((B, 2 * B), device=cd.dev)
        if D % 32 == 0:
            cd.ops.fc_bwd_data(de_[0::2], de_, None, S, B, 2 * B, D)
            old = torch.empty(B, dtype=torch.int32, device=cd.dev)
            cd.ops.semihard_select(S, de_, dr, B, D, torch.empty(2 * B, device=cd.dev), old)
            assert (old.cpu().numpy() == got).mean() > 0.95
        # deterministic
        again = torch.empty_like(neg_row)
        cd.ops.semihard_mine_x3(de_, dr, B, D, e3, D, sqn, dpd, wsp, again, h2_scale=hs)
        assert torch.equal(again, neg_row)


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f16x2"])
def test_train_step_semihard_config2_shape(cd, precision):
    """BASELINE config 2 shape (batch 8192, all-pairs mining) on a 200k-row table:
    the mined negatives satisfy the semi-hard rule under the fp64 oracle distances
    of the device's own embeddings, and loss / gradients match the oracle.  f32: the score matrix S written by
    the fp32-MFMA GEMM and scanned; f32x3: the score product on the plane kernels, the selection in its epilogue."""
    N, F, B = 200000, 1500, 8192
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs_np = osynth.cowatch_pairs(N, 40000, 0)
    ts = cd.train.TrainStep(table, dt(pairs_np, cd.dev, torch.int32), B, mode="semihard", device=cd.dev, precision=precision)
    assert ts.mine_fused == (precision != "f32")        # (f16x2 mines on the six-plane kernel too: it splits the fp32 embeddings itself)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    E = ts.ws.e.cpu().numpy().astype(np.float64)
    rows = ts.idx.cpu().numpy()
    got = ts.neg_row.cpu().numpy()
    sel = np.random.RandomState(0).choice(B, 64, replace=False)
    sq = np.sum(E * E, axis=1)
    for i in sel:
        dist = sq[2 * i] + sq - 2.0 * (E @ E[2 * i])
        elig = (rows != rows[2 * i]) & (rows != rows[2 * i + 1])
        d_p = dist[2 * i + 1]
        tol = 2e-6                                # a candidate within tol of d_p may fall either side
        strict = elig & (dist > d_p + tol)
        assert elig[got[i]]
        if dist[got[i]] > d_p - tol and (strict.any() or (elig & (dist > d_p)).any()):
            if strict.any():
                assert dist[got[i]] <= dist[strict].min() + tol
        else:
            assert dist[got[i]] >= dist[elig].max() - tol
    tri, valid = otower.semihard_triplets(got)
    w = otower.hinge_loss_indexed(E, tri, valid, 0.8, np.float64)
    assert abs(ts.loss() - float(w["hinge_loss"])) < TOL
    wd = otower.hinge_loss_indexed_backward(E, tri, valid, 0.8, np.float64)
    assert np.abs(ts.ws.de.cpu().numpy() - wd).max() < TOL
    assert torch.isfinite(ts.params.grad).all()
    for _ in range(2):
        ts.step()
    assert np.isfinite(ts.loss())


# ------------------------------------------- next rows N1 / N2: predict, eval, trainer ----
def test_evaluation_matches_reference_golden(cd, golden_dir):
    """G3: inputs and outputs produced by the reference's own evaluate.Evaluation."""
    from cdml_amd import evaluate
    g = np.load(os.path.join(golden_dir, "evaluate_mean_dist.npz"))
    ev = evaluate.Evaluation(g["features"], g["cowatches"].tolist(), device=cd.dev)
    np.testing.assert_array_equal(ev.features, g["eval_features"])          # evaluate.py:34-55
    np.testing.assert_array_equal(np.asarray(ev.cowatches), g["eval_cowatches"])
    got = ev.mean_dist(g["embeddings"], ev.cowatches)
    np.testing.assert_allclose(got, float(g["mean_dist"]), rtol=1e-6)       # evaluate.py:57-73
    emb, cw = g["embeddings"].astype(np.float64), g["eval_cowatches"]
    np.testing.assert_allclose(ev.mean_cos_dist(g["embeddings"], ev.cowatches),
                               np.mean(np.sum(emb[cw[:, 0]] * emb[cw[:, 1]], axis=-1)), rtol=1e-5)
    with pytest.raises(IndexError):
        ev.mean_dist(g["embeddings"][:3], ev.cowatches)


def test_prediction_run_features(cd, tmp_path):
    from cdml_amd import predict
    F, H, D, N = 200, 300, 64, 333
    feats = np.random.RandomState(0).random_sample((N, F)).astype(np.float32)
    params = cd.engine.VNetParams(cd.engine.TowerLayout(F, H, D), cd.dev, seed=1)
    pred = predict.Prediction(params=params)
    out = pred.run_features(feats, batch_size=128, output_dir=str(tmp_path), suffix="_t")   # 2 chunks + tail
    W = [t.cpu().numpy().astype(np.float64) for t in params.unpadded()]
    want = otower.vnet_forward(feats.astype(np.float64), *W, dtype=np.float64)["l2_norm"]
    assert out.dtype == np.float32 and out.shape == (N, D)
    np.testing.assert_allclose(out, want, atol=TOL)
    np.testing.assert_array_equal(np.load(tmp_path / "output_t.npy"), out)                  # predict.py:87-91
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)                                # padded rows
    np.testing.assert_allclose(pred.run_features(table, batch_size=1000), want, atol=TOL)
    with pytest.raises(IOError):
        predict.Prediction(ckpt=str(tmp_path / "missing.pt"))


@pytest.mark.parametrize("precision", ["f32", "f32x3", "bf16"])
def test_prediction_embed_table_production_shape(cd, precision):
    """Catalogue inference (predict.py:71-96) with the result left on the device, at the production
    width: every row of an HBM-resident catalogue in ragged chunks (last one short), fp32 against the
    fp64 oracle at 1e-5 and config-4 precision (fp16 catalogue, bf16 MFMA) at its stated 5e-3."""
    from cdml_amd import engine_bf16, predict
    F, H, D, N, chunk = 1500, 5000, 256, 2500, 1024
    feats = osynth.features_numpy(N, F, seed=3)
    if precision == "bf16":
        feats = feats.astype(np.float16)
        table = engine_bf16.FeatureTableF16.from_numpy(feats, cd.dev)
        layout = engine_bf16.layout_bf16(F, H, D)
    else:
        from cdml_amd import engine_x3
        feats = feats.astype(np.float32)
        table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
        layout = (engine_x3.layout_x3 if precision == "f32x3" else cd.engine.TowerLayout)(F, H, D)
    params = cd.engine.VNetParams(layout, cd.dev, seed=2)
    pred = predict.Prediction(params=params, precision=precision)
    out = pred.embed_table(table, chunk)
    assert out.is_cuda and out.shape == (N, D) and out.dtype == torch.float32
    W = [t.cpu().numpy().astype(np.float64) for t in params.unpadded()]
    want = otower.vnet_forward(feats.astype(np.float64), *W, dtype=np.float64)["l2_norm"]
    np.testing.assert_allclose(out.cpu().numpy(), want, atol=TOL if precision != "bf16" else 5e-3)
    np.testing.assert_allclose(out.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
    again = torch.empty_like(out)
    pred.embed_table(table, 777, out=again)                                 # other chunking, caller's buffer
    np.testing.assert_allclose(again.cpu().numpy(), out.cpu().numpy(), atol=2e-6 if precision != "bf16" else 1e-3)
    np.testing.assert_array_equal(pred.run_features(table, chunk), out.cpu().numpy())        # the ndarray API on top
    if precision == "bf16":
        with pytest.raises(ValueError):
            pred.embed_table(cd.engine.FeatureTable.from_numpy(feats.astype(np.float32), cd.dev), chunk)


def _clustered(n_videos, F, n_clusters, seed):
    rng = np.random.RandomState(seed)
    centers = rng.random_sample((n_clusters, F))
    cid = rng.randint(0, n_clusters, size=n_videos)
    feats = (centers[cid] + 0.05 * rng.randn(n_videos, F)).clip(0, None).astype(np.float32)
    a = rng.randint(0, n_videos, size=6000)
    order = np.argsort(cid, kind="stable")
    starts = np.searchsorted(cid[order], np.arange(n_clusters))
    counts = np.bincount(cid, minlength=n_clusters)
    p = order[starts[cid[a]] + rng.randint(0, 1 << 30, size=len(a)) % counts[cid[a]]]
    pairs = np.stack([a, p], axis=1)
    return feats, pairs[pairs[:, 0] != pairs[:, 1]].astype(np.int32)


@pytest.mark.parametrize("precision,B", [("auto", 64), ("f32x3", 128), ("f16x2", 128)])
def test_trainer_learns_evaluates_checkpoints_and_resumes(cd, tmp_path, precision, B):
    """train.py:224-309 policy on a learnable synthetic set (co-watched videos share
    a cluster): the loss and the held-out mean_dist go down, the best model is
    checkpointed (one file kept), and a resumed run continues bit-exactly -- on the fp32 MFMA (B = 64: what "auto" picks
    there) and on both plane forms (B = 128, twice the epochs: the same number of steps)."""
    feats, pairs = _clustered(2000, 64, 10, 0)
    eval_pairs, train_pairs = pairs[:200], pairs[200:]
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    kw = dict(hidden_size=128, output_size=32, margin=0.8, mode="uniform", optimizer="adam",
              base_learning_rate=0.002, device=cd.dev, precision=precision)
    mk = lambda: cd.train.TrainStep(table, dt(train_pairs, cd.dev, torch.int32), B, **kw)
    ts = mk()
    assert ts.precision == ("f32" if precision == "auto" else precision)
    tr = cd.train.Trainer(ts, num_epochs=2 * (B // 64), n_pairs=len(train_pairs), checkpoint_dir=str(tmp_path),
                          eval_features=feats, eval_cowatches=eval_pairs.tolist(), check_stop_epoch=0.2,
                          best_eval_dist=10.0, eval_per_epoch=8, require_improve_num=100)
    assert tr.num_batches == (len(train_pairs) * 2 * (B // 64)) // B
    hist = tr.run()
    assert tr.stopped == "end of data" and ts.global_step == tr.num_batches
    assert hist[-1][1] < 0.6 * hist[0][1], hist                 # loss went down
    ev = tr.eval_history
    assert len(ev) >= 8 and ev[-1][1] < 0.8 * ev[0][1]          # held-out positive distance too
    assert tr.best_eval_dist == min(e[1] for e in ev if e[0] > tr.check_stop_step)
    files = sorted(f for f in os.listdir(tmp_path) if f.startswith("model.ckpt-"))
    assert len(files) == 1                                                  # max_to_keep=1
    # the reference's TensorBoard scalars (train.py:154-160, 246-249; losses.py:40-41), one JSON line per evaluation
    import json
    recs = [json.loads(ln) for ln in open(tmp_path / "summaries.jsonl")]
    assert len(recs) == len(ev) and [r["global_step"] for r in recs] == [e[0] for e in ev]
    names = {"loss", "reg_loss", "variance", "final_learning_rate", "mean_pos_dist", "mean_neg_dist", "eval/eval_dist",
             "eval/best_eval_dist"}
    assert names <= set(recs[-1])
    last = recs[-1]
    assert abs(last["eval/eval_dist"] - ev[-1][1]) < 1e-12 and abs(last["eval/best_eval_dist"] - ev[-1][2]) < 1e-12
    assert last["final_learning_rate"] == 0.002 and all(np.isfinite(last[k]) for k in names)
    W1, _, W2, _ = [t.cpu().numpy().astype(np.float64) for t in ts.params.unpadded()]
    reg = 1e-8 * ((W1 ** 2).sum() + (W2 ** 2).sum()) / 2                    # models.py:28, train.py:133-136
    # (the record is the LAST EVALUATION's; a few more steps may have moved the weights since)
    assert abs(last["reg_loss"] - reg) < (1e-6 if last["global_step"] == ts.global_step else 1e-2) * reg
    # loss / distances / variance of the last step, when the run ended on an evaluation step
    if last["global_step"] == ts.global_step:
        assert abs(last["loss"] - ts.loss()) < 1e-7
        e = ts.ws.e[:, :32].double().view(B, 3, 32)
        assert abs(last["mean_pos_dist"] - float(((e[:, 0] - e[:, 1]) ** 2).sum(1).mean())) < 1e-6
        assert abs(last["mean_neg_dist"] - float(((e[:, 0] - e[:, 2]) ** 2).sum(1).mean())) < 1e-6
        var = float(((e - e.mean(dim=(0, 1), keepdim=True)) ** 2).mean())            # calc_var, train.py:67-71
        assert abs(last["variance"] - var) < 1e-6
    # resume: 5 steps + save + 5 steps  ==  10 steps straight
    a, b = mk(), mk()
    for _ in range(10):
        a.step()
    for _ in range(5):
        b.step()
    ck = tmp_path / "resume.pt"
    torch.save(b.state_dict(), ck)
    c = mk()
    c.load_state_dict(torch.load(ck, map_location="cpu"))
    for _ in range(5):
        c.step()
    torch.cuda.synchronize()
    assert c.global_step == 10 and torch.equal(a.params.flat, c.params.flat)
    assert torch.equal(a.m, c.m) and torch.equal(a.idx, c.idx)
    # early stop: patience 0 after the check-stop step
    ts2 = mk()
    tr2 = cd.train.Trainer(ts2, num_epochs=50, n_pairs=len(train_pairs), eval_features=feats,
                           eval_cowatches=eval_pairs.tolist(), check_stop_epoch=0.05,
                           best_eval_dist=1e-9, eval_per_epoch=20, require_improve_num=1)
    tr2.run()
    assert tr2.stopped == "early stop" and ts2.global_step < tr2.num_batches
    # a checkpoint feeds Prediction (predict.py:46-59)
    from cdml_amd import predict
    p = predict.Prediction(ckpt=os.path.join(tmp_path, files[0]), device=cd.dev)
    assert p.run_features(feats[:10], 4).shape == (10, 32)


def test_pipe_from_reference_format_directory(cd, tmp_path):
    """N3: a dataset directory in the reference's on-disk formats drives the pipe the
    way train.py:345 constructs it (file pattern + features.npy)."""
    from cdml_amd import online_data
    rng = np.random.RandomState(1)
    feats = rng.random_sample((300, 16)).astype(np.float32)
    pairs = rng.randint(0, 300, size=(500, 2))
    pairs = pairs[pairs[:, 0] != pairs[:, 1]].tolist()
    d = str(tmp_path)
    online_data.write_features(feats, save_dir=d)
    online_data.write_cowatches(pairs, d, split_num=3, eval_num=40, test_num=40)
    pipe = cd.inputs.MPTripletPipe(cowatch_file_patten=d + "/*.train", feature_file=d + "/features.npy",
                                   wait_times=20, device=cd.dev)
    train_pairs = np.asarray(pairs[80:], dtype=np.int32)
    assert pipe.cowatch_num == len(train_pairs) and len(pipe.cowatch_files) == 3
    pipe.create_pipe(num_epochs=1, batch_size=32)
    b = pipe.get_batch()
    idx = osampler.device_triplets_vec(train_pairs, 300, 1234, 0, 32)
    np.testing.assert_array_equal(b.cpu().numpy(), feats[idx])
