"""BASELINE configs at their TRUE sizes (VERDICT r1 "what's weak" 2-3).

  * 1 M-row fp32 catalogue (6.1 GB) and a 1.5 M-row fp16 one (4.6 GB): gathered rows whose
    byte offset lies beyond 2^32 are compared, content and all, with the oracle's generator
    (`k_gather_rows`, `k_sample_gather`, `k_gather_rows_f16`) -- inputs.py:158.
  * config 1 / config 2 steps on the 1 M-row table (size-independent properties + content).
  * config 3's per-GPU shape: a 1.25 M-row shard with row0 != 0, B = 8192 in-batch fp32,
    through RowExchange over RCCL (world size 1: the box has one GPU).
  * a WELL-CONDITIONED end-to-end gradient check at F=1500 / H=5000 / D=256 (clustered
    catalogue: co-watched videos share a cluster, so embeddings do not collapse and the
    gradient is not a sum of cancelling terms): per-tensor relative L2 error <= 1e-4 in fp32
    and <= 1e-2 on the bf16 path against the fp64 oracle -- train.py:141.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import sampler as osampler, synth as osynth, tower as otower

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F, H, D = 1500, 5000, 256
GIB4 = 1 << 32


@pytest.fixture(scope="module")
def cd(gpu):
    import cdml_amd
    from cdml_amd import engine, engine_bf16, ops, train
    cdml_amd.load_library()

    class NS:
        pass
    ns = NS()
    ns.dev, ns.engine, ns.ebf, ns.ops, ns.train = gpu, engine, engine_bf16, ops, train
    return ns


@pytest.fixture(scope="module")
def table_1m(cd):
    """config 1/2 catalogue: 1 M x 1500 fp32, 6144-B rows -> rows >= 699 051 start past 4 GiB."""
    t = cd.engine.FeatureTable.synthetic(1000000, F, 0, cd.dev)
    assert t.data.numel() * 4 > GIB4
    yield t
    del t
    torch.cuda.empty_cache()


def _rows_across_4gib(n_rows, row_bytes, rng, n_extra=40):
    """Row ids on both sides of every 2^31 / 2^32-byte boundary inside the table, its two
    ends, and random rows from the part beyond 4 GiB."""
    ids = [0, 1, n_rows - 2, n_rows - 1]
    for k in range(1, int(n_rows * row_bytes >> 31) + 1):
        r = (k << 31) // row_bytes
        ids += [r - 1, r, r + 1]
    first_hi = GIB4 // row_bytes + 1
    ids += list(rng.randint(first_hi, n_rows, size=n_extra))
    ids = np.array([i for i in ids if 0 <= i < n_rows], dtype=np.int32)
    assert (ids.astype(np.int64) * row_bytes > GIB4).sum() >= n_extra
    return ids


def _oracle_rows(ids, seed=0):
    return np.concatenate([osynth.features_philox(int(r), 1, F, seed) for r in ids])


def test_gather_rows_content_past_4gib_fp32(cd, table_1m):
    ids = _rows_across_4gib(table_1m.n_rows, 6144, np.random.RandomState(1))
    d_ids = torch.as_tensor(ids).to(cd.dev)
    raw = torch.full((len(ids), 1536), -1.0, device=cd.dev)
    cd.ops.gather_rows(table_1m.data, 0, d_ids, F, raw, normalize=False)
    want = _oracle_rows(ids)
    np.testing.assert_array_equal(raw[:, :F].cpu().numpy(), want)               # bit-exact content
    assert float(raw[:, F:].abs().max()) == 0
    x = torch.empty((len(ids), 1536), device=cd.dev)
    cd.ops.gather_rows(table_1m.data, 0, d_ids, F, x, normalize=True)
    np.testing.assert_allclose(x[:, :F].cpu().numpy(), otower.l2_normalize(want.astype(np.float64), np.float64)[0],
                               atol=1e-6)


@pytest.mark.parametrize("mode", ["inbatch", "uniform"])
def test_sample_gather_content_past_4gib(cd, table_1m, mode):
    """The fused sampler+gather on pairs that live in the last third of the table."""
    N = table_1m.n_rows
    rng = np.random.RandomState(2)
    B = 512
    pairs_np = rng.randint(GIB4 // 6144 + 1, N, size=(700, 2)).astype(np.int32)
    pairs_np[:3] = [[N - 1, N - 2], [GIB4 // 6144, GIB4 // 6144 + 1], [N - 1, 0]]
    pairs = torch.as_tensor(pairs_np).to(cd.dev)
    rpt = 2 if mode == "inbatch" else 3
    idx = torch.zeros(B * rpt, dtype=torch.int32, device=cd.dev)
    shift = torch.zeros(1, dtype=torch.int32, device=cd.dev)
    x = torch.empty((B * rpt, 1536), device=cd.dev)
    step = 3
    cd.ops.sample_gather(1 if mode == "inbatch" else 0, pairs, 1234, step, B, table_1m.data, F, idx, x,
                         shift_out=shift)
    if mode == "inbatch":
        want_idx = osampler.device_inbatch(pairs_np, 1234, step, B)[0]
    else:
        want_idx = osampler.device_triplets_vec(pairs_np, N, 1234, step, B).reshape(-1)
    got_idx = idx.cpu().numpy()
    np.testing.assert_array_equal(got_idx, want_idx)
    sel = np.unique(np.concatenate([np.arange(0, B * rpt, 37), np.arange(12), [B * rpt - 1]]))
    want = otower.l2_normalize(_oracle_rows(got_idx[sel]).astype(np.float64), np.float64)[0]
    np.testing.assert_allclose(x[sel, :F].cpu().numpy(), want, atol=1e-6)
    assert (got_idx[sel].astype(np.int64) * 6144 > GIB4).sum() > len(sel) // 2


@pytest.fixture(scope="module")
def table_f16_15m(cd):
    """fp16 catalogue (config 4): 3072-B rows, so 1.5 M rows put a third of the table past 4 GiB."""
    t = cd.ebf.FeatureTableF16.synthetic(1500000, F, 0, cd.dev)
    assert t.data.numel() * 2 > GIB4
    yield t
    del t
    torch.cuda.empty_cache()


def test_gather_rows_f16_content_past_4gib(cd, table_f16_15m):
    table, N = table_f16_15m, table_f16_15m.n_rows
    ids = _rows_across_4gib(N, 3072, np.random.RandomState(3))
    np.testing.assert_array_equal(table.data[torch.as_tensor(ids.astype(np.int64)).to(cd.dev), :F].cpu().numpy(),
                                  _oracle_rows(ids).astype(np.float16))          # the fill kernel itself
    x = torch.full((len(ids), 1536), 7.0, dtype=torch.bfloat16, device=cd.dev)
    cd.ops.gather_rows_f16(table.data, 0, torch.as_tensor(ids).to(cd.dev), F, x)
    want = otower.l2_normalize(_oracle_rows(ids).astype(np.float16).astype(np.float64), np.float64)[0]
    np.testing.assert_allclose(x[:, :F].float().cpu().numpy(), want, rtol=2 ** -8, atol=1e-6)
    assert float(x[:, F:].float().abs().max()) == 0


@pytest.mark.parametrize("mode", ["inbatch", "uniform"])
def test_sample_gather_f16_content_past_4gib(cd, table_f16_15m, mode):
    """The kernel the config-4 step runs -- k_sample_gather<mode, RowF16>: sampler fused into the fp16
    gather, four steps per launch -- on pairs that live past the 4 GiB mark of a 1.5 M-row table:
    ids bit-exact against oracle/sampler.py for every step of the launch, rows against the oracle's
    generator rounded to fp16 and normalised (bf16 output: rtol 2^-8).  inputs.py:125-127,158."""
    table, N = table_f16_15m, table_f16_15m.n_rows
    first_hi = GIB4 // 3072 + 1
    rng = np.random.RandomState(4)
    B, K, step0 = 384, 4, 7
    pairs_np = rng.randint(first_hi, N, size=(900, 2)).astype(np.int32)
    pairs_np[:3] = [[N - 1, N - 2], [first_hi - 1, first_hi], [N - 1, 0]]
    pairs = torch.as_tensor(pairs_np).to(cd.dev)
    rpt = 2 if mode == "inbatch" else 3
    R = B * rpt
    idx = torch.full((K, R), -1, dtype=torch.int32, device=cd.dev)
    shift = torch.full((K,), -1, dtype=torch.int32, device=cd.dev)
    x = torch.full((K, R, 1536), 7.0, dtype=torch.bfloat16, device=cd.dev)
    oob = torch.zeros(1, dtype=torch.int32, device=cd.dev)
    cd.ops.sample_gather(1 if mode == "inbatch" else 0, pairs, 1234, step0, B, table.data, F, idx, x,
                         shift_out=shift, n_steps=K, oob_flag=oob)
    got = idx.cpu().numpy()
    far = 0
    for s in range(K):
        if mode == "inbatch":
            want_idx, _, _, want_shift = osampler.device_inbatch(pairs_np, 1234, step0 + s, B)
            assert int(shift[s].item()) == int(want_shift)
        else:
            want_idx = osampler.device_triplets_vec(pairs_np, N, 1234, step0 + s, B).reshape(-1)
        np.testing.assert_array_equal(got[s], want_idx)
        sel = np.unique(np.concatenate([np.arange(0, R, 41), np.arange(9), [R - 1]]))
        raw = _oracle_rows(got[s][sel]).astype(np.float16)
        want = otower.l2_normalize(raw.astype(np.float64), np.float64)[0]
        np.testing.assert_allclose(x[s][sel, :F].float().cpu().numpy(), want, rtol=2 ** -8, atol=1e-6)
        far += int((got[s][sel].astype(np.int64) * 3072 > GIB4).sum())
    assert far > K * len(sel) // 2
    assert float(x[:, :, F:].float().abs().max()) == 0 and int(oob.item()) == 0


def _f32_views(ts):
    """(x_hat, dz1) as fp32 tensors: the fp32 path holds them so, precision "f32x3" as three bf16 planes"""
    if ts.x3 or getattr(ts, "h2", False):
        return ts.ws.x_hat_f32(), ts.ws.dz1_f32()
    return ts.ws.x_hat, ts.ws.dz1


def _check_step_properties(cd, ts, pairs_np, B, rows_per_triplet):
    """Size-independent properties of one step at full size (+ gathered content)."""
    rows = ts.idx.cpu().numpy()
    x_hat, dz1 = _f32_views(ts)
    xn = x_hat[:, :F].norm(dim=1)
    assert float((xn - 1).abs().max()) < 1e-5
    en = ts.ws.e.norm(dim=1)
    assert float((en - 1).abs().max()) < 1e-5
    sel = np.unique(np.concatenate([np.arange(0, len(rows), 997), [len(rows) - 1]]))
    want = otower.l2_normalize(_oracle_rows(rows[sel]).astype(np.float64), np.float64)[0]
    np.testing.assert_allclose(x_hat[sel, :F].cpu().numpy(), want, atol=1e-6)
    assert (rows.astype(np.int64) * 6144 > GIB4).sum() > len(rows) // 5       # the far part IS visited
    assert torch.isfinite(ts.params.grad).all()
    ref = x_hat.double().T @ dz1[:, :256].double()
    assert float((ts.params.gW1[:, :256].double() - ref).abs().max()) < 1e-5
    sub = otower.vnet_forward(x_hat[:64, :F].cpu().numpy().astype(np.float64),
                              *[t.cpu().numpy().astype(np.float64) for t in ts.params.unpadded()],
                              dtype=np.float64)
    assert np.abs(ts.ws.e[:64, :D].cpu().numpy() - sub["l2_norm"]).max() < 1e-5


@pytest.mark.parametrize("precision", ["f32x3", "f32", "f16x2"])
def test_config1_step_on_1m_rows(cd, table_1m, precision):
    """BASELINE config 1 as stated: 1 M x 1500 fp32 in HBM, B = 4096, in-batch negatives -- on the headline's path
    (precision "f32x3": fp32 values as three exact bf16 planes, six plane products per fp32 product) and on the
    fp32-MFMA path, the same bounds for both."""
    B = 4096
    pairs_np = osynth.cowatch_pairs(table_1m.n_rows, 60000, 0)
    ts = cd.train.TrainStep(table_1m, torch.as_tensor(pairs_np).to(cd.dev), B, mode="inbatch", device=cd.dev,
                            precision=precision)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ts.idx.cpu().numpy().reshape(B, 2), pairs_np[np.arange(B) % len(pairs_np)])
    _check_step_properties(cd, ts, pairs_np, B, 2)
    g0 = ts.params.grad.clone()
    ts.fetch(); ts.forward_loss(); ts.backward()
    assert torch.equal(g0, ts.params.grad)                                      # deterministic
    for _ in range(2):
        ts.step()
    assert np.isfinite(ts.loss()) and int(ts.step_dev.item()) == 2


@pytest.mark.parametrize("precision", ["f32x3", "f32", "f16x2"])
def test_config2_step_on_1m_rows(cd, table_1m, precision):
    """BASELINE config 2 as stated: same catalogue, semi-hard mining over the batch, B = 8192 (both fp32 paths)."""
    B = 8192
    pairs_np = osynth.cowatch_pairs(table_1m.n_rows, 60000, 0)
    ts = cd.train.TrainStep(table_1m, torch.as_tensor(pairs_np).to(cd.dev), B, mode="semihard", device=cd.dev,
                            precision=precision)
    ts.fetch(); ts.forward_loss(); ts.backward()
    torch.cuda.synchronize()
    _check_step_properties(cd, ts, pairs_np, B, 2)
    nr = ts.neg_row.cpu().numpy()
    rows = ts.idx.cpu().numpy()
    ok = nr >= 0
    assert ok.mean() > 0.99
    i = np.flatnonzero(ok)
    assert np.all(rows[nr[i]] != rows[2 * i]) and np.all(rows[nr[i]] != rows[2 * i + 1])   # eligible negatives
    # the mined negative is semi-hard (farther than the positive) whenever such a row exists
    e = ts.ws.e[:, :D].double()
    sub = i[:256]
    d = torch.cdist(e[2 * torch.as_tensor(sub)], e) ** 2
    dp = d[torch.arange(len(sub)), torch.as_tensor(2 * sub + 1)]
    dn = d[torch.arange(len(sub)), torch.as_tensor(nr[sub]).long()]
    elig = torch.as_tensor((rows[None, :] != rows[2 * sub][:, None]) & (rows[None, :] != rows[2 * sub + 1][:, None]))
    has = ((d.cpu() > dp.cpu()[:, None] + 2e-6) & elig).any(1)
    assert bool((dn.cpu()[has] > dp.cpu()[has] - 2e-6).all())
    ts.step()
    assert np.isfinite(ts.loss())


def test_config4_step_on_10m_rows_fp16(cd):
    """BASELINE config 4's per-GPU workload at its TRUE size: the 10 M x 1500 fp16 catalogue (30.7 GB in
    HBM), B = 8192 uniform (global) negatives -> 24 576 rows per step, bf16 MFMA tower, eager and
    replayed from a hipGraph.  Size-independent properties + gathered content: ids bit-exact against
    oracle/sampler.py (N = 10 M in the rejection rule), rows against the oracle's generator (fp16,
    normalised; bf16 output), unit norms, dW1 = x_hat^T dz1 and db1 = column sums recomputed from the
    SAME bf16 operands, bit-identical repetition, and graph replay == eager bit for bit over a
    gather block boundary.  inputs.py:158, models.py:46-62, train.py:141."""
    N, B = 10000000, 8192
    table = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev)
    assert table.data.numel() * 2 > 7 * GIB4
    pairs_np = osynth.cowatch_pairs(N, 60000, 0)
    pairs = torch.as_tensor(pairs_np).to(cd.dev)
    mk = lambda g: cd.train.TrainStep(table, pairs, B, mode="uniform", precision="bf16", device=cd.dev, use_graph=g)
    a = mk(False)
    a.fetch(); a.forward_loss(); a.backward()
    torch.cuda.synchronize()
    idx = osampler.device_triplets_vec(pairs_np, N, 1234, 0, B)
    rows = a.idx.cpu().numpy()
    np.testing.assert_array_equal(rows.reshape(B, 3), idx)
    assert (rows.astype(np.int64) * 3072 > GIB4).mean() > 0.5 and (rows.astype(np.int64) * 3072 > 6 * GIB4).any()
    sel = np.unique(np.concatenate([np.arange(0, len(rows), 769), np.argsort(rows)[-4:], [len(rows) - 1]]))
    raw = _oracle_rows(rows[sel]).astype(np.float16)
    np.testing.assert_array_equal(table.data[torch.as_tensor(rows[sel].astype(np.int64)).to(cd.dev), :F].cpu().numpy(), raw)
    want = otower.l2_normalize(raw.astype(np.float64), np.float64)[0]
    np.testing.assert_allclose(a.ws.x_hat[sel, :F].float().cpu().numpy(), want, rtol=2 ** -8, atol=1e-6)
    assert float((a.ws.x_hat[:, :F].float().norm(dim=1) - 1).abs().max()) < 4e-3     # bf16 rows
    assert float((a.ws.e.norm(dim=1) - 1).abs().max()) < 1e-5                        # output l2norm is fp32
    assert torch.isfinite(a.params.grad).all() and abs(a.loss() - 0.8) < 0.2 and int(a.oob.item()) == 0
    ref = a.ws.x_hat.double().T @ a.ws.dz1[:, :256].double()                         # same bf16 operands
    scale = float(ref.abs().max())
    assert float((a.params.gW1[:, :256].double() - ref).abs().max()) < 1e-5 * max(scale, 1e-3) + 1e-7
    db1 = a.ws.dz1.double().sum(0)
    assert float((a.params.gb1.double() - db1).abs().max()) < 1e-5 * max(float(db1.abs().max()), 1e-3) + 1e-7
    ref2 = a.ws.h1[:, :512].double().T @ a.ws.dz2_bf.double()
    assert float((a.params.gW2[:512].double() - ref2).abs().max()) < 1e-5 * max(float(ref2.abs().max()), 1e-3) + 1e-7
    g0 = a.params.grad.clone()
    a.fetch(); a.forward_loss(); a.backward()
    assert torch.equal(g0, a.params.grad)
    del g0, ref, ref2
    b = mk(True)
    assert 2 <= b.gather_ahead <= 4          # (steps per gather launch by the bytes a launch writes: 3 at this shape)
    for _ in range(6):                       # replays on both sides of a gather-block boundary
        a.step(); b.step()
    torch.cuda.synchronize()
    assert b.use_graph and len(b._graphs) == b.gather_ahead
    assert torch.equal(a.params.flat, b.params.flat) and torch.equal(a.m, b.m)
    assert torch.equal(a.idx, b.idx) and int(b.step_dev.item()) == 6
    np.testing.assert_array_equal(b.idx.cpu().numpy().reshape(B, 3), osampler.device_triplets_vec(pairs_np, N, 1234, 5, B))
    assert np.isfinite(b.loss())
    del a, b, table
    torch.cuda.empty_cache()


# --------------------------------------------------------- config 3, per-GPU shape --
def _config3_worker(port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from cdml_amd import dist as cdist, engine, train
        r, per, n_global, B = 5, 1250000, 10000000, 8192
        row0 = per * r
        table = engine.FeatureTable.synthetic(per, F, 0, dev, row0=row0, n_rows_global=n_global)
        rng = np.random.RandomState(5)
        pairs_np = (row0 + rng.randint(0, per, size=(40000, 2))).astype(np.int32)
        pairs_np[:2] = [[row0, row0 + per - 1], [row0 + per - 1, row0 + GIB4 // 6144 + 1]]
        pairs = torch.as_tensor(pairs_np).to(dev)
        ex = cdist.RowExchange(n_global, group=dist.new_group(), skip_self=False)   # keep the RCCL all-to-alls
        ts = train.TrainStep(table, pairs, B, mode="inbatch", device=dev, exchange=ex, grad_sync=cdist.GradSync(device=dev),
                             slot0=0, batch_global=B)
        # the same shard addressed directly (no exchange): bit-identical step
        ts2 = train.TrainStep(table, pairs, B, mode="inbatch", device=dev)
        ts2.table = engine.FeatureTable(table.data, F, row0=0, n_rows_global=per)   # local ids
        ts2.pairs = pairs - row0
        msgs = []
        for it in range(2):
            ts.step(); ts2.step()
            if it == 0:
                # the data-parallel step produces dW1 in two row blocks (other split-K order): the
                # gradient agrees to fp32 summation noise, not bit for bit
                gd = float((ts.params.grad - ts2.params.grad).abs().max())
                if gd > 1e-6:
                    msgs.append("gradients differ: %g" % gd)
                if not torch.equal(ts.ws.x_hat, ts2.ws.x_hat):
                    msgs.append("exchanged rows differ from directly gathered rows")
        torch.cuda.synchronize()
        if not torch.equal(ts.idx - row0, ts2.idx):
            msgs.append("ids differ")
        # Adam turns gradient noise near g = 0 into lr-sized update noise (iid catalogue): loose bar
        wd = (ts.params.flat - ts2.params.flat).abs()
        if float(wd.max()) > 2.5e-2 or float((wd > 1e-4).float().mean()) > 0.02:
            msgs.append("weights differ: max %g" % float(wd.max()))
        rows = ts.idx.cpu().numpy()
        want_rows = osampler.device_inbatch(pairs_np, 1234, 1, B)[0]
        if not np.array_equal(rows, want_rows):
            msgs.append("ids differ from the oracle")
        sel = np.unique(np.concatenate([np.arange(0, 2 * B, 1201), [2 * B - 1]]))
        raw = np.concatenate([osynth.features_philox(int(g), 1, F, 0) for g in rows[sel]])
        want = otower.l2_normalize(raw.astype(np.float64), np.float64)[0]
        # (TrainStep's default precision is the headline's since round 6: the gathered rows are three bf16 planes there)
        xh = ts.ws.x_hat_f32() if hasattr(ts.ws, "x_hat_f32") else ts.ws.x_hat
        err = float(np.abs(xh[sel, :F].cpu().numpy() - want).max())
        if err > 1e-6:
            msgs.append("gathered content off by %g" % err)
        if ((rows.astype(np.int64) - row0) * 6144 > GIB4).sum() < len(rows) // 5:
            msgs.append("far part of the shard not visited")
        if not np.isfinite(ts.loss()):
            msgs.append("non-finite loss")
        q.put("ok" if not msgs else "; ".join(msgs))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(traceback.format_exc())


def test_config3_per_gpu_shard_shape(gpu):
    """BASELINE config 3, one rank's share: rows [6.25 M, 7.5 M) of the 10 M catalogue (7.7 GB,
    row0 != 0), B = 8192 in-batch fp32, fetched through RowExchange over RCCL and stepped with
    GradSync in place; equals the same shard addressed directly, ids equal the oracle's, gathered
    content checked past the 4 GiB mark."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_config3_worker, args=(port, q))
    p.start()
    msg = q.get(timeout=600)
    p.join(timeout=60)
    assert msg == "ok", msg


# ------------------------------------------- well-conditioned end-to-end gradients --
def _clustered_catalogue(n_videos, n_clusters, seed):
    """Learnable catalogue at production width: co-watched videos share a cluster."""
    rng = np.random.RandomState(seed)
    centers = rng.random_sample((n_clusters, F))
    cid = rng.randint(0, n_clusters, size=n_videos)
    feats = (centers[cid] + 0.25 * rng.randn(n_videos, F)).clip(0, None).astype(np.float32)
    a = rng.randint(0, n_videos, size=8000)
    order = np.argsort(cid, kind="stable")
    starts = np.searchsorted(cid[order], np.arange(n_clusters))
    counts = np.bincount(cid, minlength=n_clusters)
    p = order[starts[cid[a]] + rng.randint(0, 1 << 30, size=len(a)) % counts[cid[a]]]
    pairs = np.stack([a, p], axis=1)
    return feats, pairs[pairs[:, 0] != pairs[:, 1]].astype(np.int32)


def _oracle_grads(f64, pairs, W, step, B, mode, N, dev_h1, dev_z, dev_t, band, band_t):
    """fp64 oracle step.  leaky-relu's derivative jumps at 0 (one entry of h1 at 3e-9 on the
    other side moves |dz1| by 4e-4 of its norm) and so does the hinge's, so entries whose
    activation lies within `band` of 0 -- and triplets whose pos-neg+margin lies within `band_t`
    -- inside the forward tolerance of the path, take the DEVICE's side of the jump; everything
    else is the oracle's own.  Returns also how many entries / triplets that was."""
    if mode == "uniform":
        idx = osampler.device_triplets_vec(pairs, N, 1234, step, B)
        rows = idx.reshape(-1)
        tri = np.arange(3 * B).reshape(B, 3)
        valid = np.ones(B, bool)
    else:
        rows, tri, valid, _ = osampler.device_inbatch(pairs, 1234, step, B)
        valid = valid.astype(bool)
    fwd = otower.vnet_forward(f64[rows], *W, dtype=np.float64)
    loss = otower.hinge_loss_indexed(fwd["l2_norm"], tri, valid, 0.8, np.float64)
    t = loss["pos_dist"] - loss["neg_dist"] + 0.8
    amb_t = np.abs(t) < band_t
    act = np.where(amb_t, dev_t >= 0, t >= 0) & valid
    n_amb = 0
    for key, dev in (("layer_1", dev_h1), ("layer_2", dev_z)):
        amb = np.abs(fwd[key]) < band
        n_amb += int(amb.sum())
        fwd[key] = np.where(amb, dev.astype(np.float64), fwd[key])
    E = fwd["l2_norm"]
    a, p, n = E[tri[:, 0]], E[tri[:, 1]], E[tri[:, 2]]
    s = (act.astype(np.float64) * 2.0 / B)[:, None]           # hinge_loss_indexed_backward with `act` given
    dE = np.zeros_like(E)
    np.add.at(dE, tri[:, 0], s * (n - p))
    np.add.at(dE, tri[:, 1], -s * (a - p))
    np.add.at(dE, tri[:, 2], s * (a - n))
    grads = otower.vnet_backward(fwd, W[2], dE, np.float64)
    return fwd, float(loss["hinge_loss"]), grads, n_amb, int(amb_t.sum())


def _rel_l2(got, want):
    return float(np.linalg.norm(got.astype(np.float64) - want) / max(np.linalg.norm(want), 1e-300))


@pytest.mark.parametrize("mode", ["uniform", "inbatch"])
@pytest.mark.parametrize("precision,bar", [("f32", 1e-4), ("f32x3", 1e-4), ("f16x2", 1e-4), ("bf16", 1e-2)])
def test_gradients_well_conditioned_production_shape(cd, mode, precision, bar):
    """Per-tensor relative L2 error of dW1, db1, dW2, db2 against the fp64 oracle at F=1500 /
    H=5000 / D=256, B=256, over several Adam steps (each step checked from the device's own
    weights).  On this catalogue the embeddings spread (mean pos << mean neg after a few steps)
    and the gradient norm stays ~1e-2..1, so a relative bar is meaningful -- unlike the iid
    catalogue of test_train_steps_config0, where |g| ~ 1e-8."""
    N, B = 6000, 256
    feats, pairs = _clustered_catalogue(N, 24, 0)
    if precision == "bf16":
        feats = feats.astype(np.float16)
        table = cd.ebf.FeatureTableF16.from_numpy(feats, cd.dev)
    else:
        table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    ts = cd.train.TrainStep(table, torch.as_tensor(pairs).to(cd.dev), B, mode=mode, base_learning_rate=2e-4,
                            precision=precision, device=cd.dev)
    f64 = feats.astype(np.float64)
    names = ("dW1", "db1", "dW2", "db2")
    worst, checked = {}, 0
    f32like = precision in ("f32", "f32x3", "f16x2")      # the plane paths are held to the fp32 path's bounds
    band = 1e-6 if f32like else 2e-4
    for step in range(4):
        W = [t.detach().cpu().numpy().astype(np.float64) for t in ts.params.unpadded()]
        ts.step()
        torch.cuda.synchronize()
        dev_t = (ts.pos - ts.neg + 0.8).cpu().numpy().astype(np.float64)
        fwd, loss, grads, n_amb, n_amb_t = _oracle_grads(f64, pairs, W, step, B, mode, N,
                                                         (ts.ws.h1_f32() if precision in ("f32x3", "f16x2") else ts.ws.h1)[:, :H].float().cpu().numpy(),
                                                         ts.ws.z[:, :D].float().cpu().numpy(), dev_t, band,
                                                         1e-5 if f32like else 1e-2)
        e = ts.ws.e[:, :D].cpu().numpy()
        assert np.abs(e - fwd["l2_norm"]).max() < (1e-5 if f32like else 5e-3)
        assert abs(ts.loss() - loss) < (1e-5 if f32like else 2e-2)
        assert n_amb_t <= (1 if f32like else 40), n_amb_t
        checked += 1
        # (leaky-relu halves the distance to 0 five-fold on the negative side: |h| < band is a
        # few 1e-4 of the entries in fp32, a few % at the bf16 band)
        assert n_amb < (3e-4 if f32like else 0.06) * fwd["layer_1"].size, n_amb
        for got, k in zip(ts.params.unpadded(grads=True), names):
            assert np.linalg.norm(grads[k]) > 1e-4, (k, step, "gradient vanished: test is ill-conditioned")
            r = _rel_l2(got.cpu().numpy(), grads[k])
            worst[k] = max(worst.get(k, 0.0), r)
            assert r <= bar, (k, step, r)
    assert checked == 4
    print("worst relative L2 error per tensor (%s, %s):" % (precision, mode), worst)


@pytest.mark.parametrize("precision", ["bf16", "f32x3", "f16x2"])
def test_resume_bf16_is_bit_exact(cd, tmp_path, precision):
    """load_state_dict must refresh the bf16 operand copies of the weights (f32x3: their planes; f16x2: the plane
    SCALES are state too) --
    5 steps + save/load into a fresh TrainStep + 5 steps == 10 straight steps."""
    N = 4000
    Table = cd.ebf.FeatureTableF16 if precision == "bf16" else cd.engine.FeatureTable
    table = Table.synthetic(N, 200, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 500, 0)).to(cd.dev)
    mk = lambda: cd.train.TrainStep(table, pairs, 128, hidden_size=256, output_size=128, mode="uniform",
                                    precision=precision, device=cd.dev)
    a, b = mk(), mk()
    for _ in range(10):
        a.step()
    for _ in range(5):
        b.step()
    ck = tmp_path / "resume_bf16.pt"
    torch.save(b.state_dict(), ck)
    c = mk()
    c.load_state_dict(torch.load(ck, map_location="cpu"))
    for _ in range(5):
        c.step()
    torch.cuda.synchronize()
    assert c.global_step == 10 and torch.equal(a.params.flat, c.params.flat)
    assert torch.equal(a.m, c.m) and torch.equal(a.idx, c.idx)


def test_gather_steps_per_launch_follow_the_bytes_a_launch_writes(cd):
    """TrainStep(gather_ahead="auto"), round 5: as many steps per fused sampler + gather launch (1 .. 4) as keep the bytes the
    launch WRITES near the 256 MB Infinity Cache (profiles/r05_gather_sweep.txt): three-plane rows at 16 384 rows a step -> 2,
    at 8 192 -> 4; fp32 rows -> 4; config 4's bf16 rows at 24 576 rows a step -> 3; an explicit value is taken as given, and
    the row-sharded path fetches per step."""
    from oracle import synth as osynth
    N, F = 20000, 1500
    t32 = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 6000, 0)).to(cd.dev)
    mk = lambda table, B, mode, prec, **kw: cd.train.TrainStep(table, pairs, B, mode=mode, device=cd.dev, precision=prec, **kw)
    assert mk(t32, 8192, "inbatch", "f32x3").gather_ahead == 2
    assert mk(t32, 4096, "inbatch", "f32x3").gather_ahead == 4
    assert mk(t32, 4096, "inbatch", "f32").gather_ahead == 4
    assert mk(t32, 4096, "inbatch", "f32x3", gather_ahead=3).gather_ahead == 3
    t16 = cd.ebf.FeatureTableF16.synthetic(N, F, 0, cd.dev) if hasattr(cd, "ebf") else None
    if t16 is not None:
        assert mk(t16, 8192, "uniform", "bf16").gather_ahead == 3
