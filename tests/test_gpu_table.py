"""Trainable catalogue rows (north_star: "the catalogue feature table and its Adam states shard
row-wise"; build-defined, the reference's features are frozen): the lazy-Adam row update through
the C ABI against oracle/table.py, and the step that uses it."""
import numpy as np
import pytest
import torch

from oracle import synth as osynth, table as otable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cd(gpu):
    from cdml_amd import engine, ops, train

    class NS:
        pass
    ns = NS()
    ns.dev, ns.engine, ns.ops, ns.train = gpu, engine, ops, train
    return ns


@pytest.mark.parametrize("F,stride", [(50, 64), (96, 96), (1500, 1536)])
def test_table_adam_rows_vs_oracle(cd, F, stride):
    rng = np.random.RandomState(F)
    n_rows, row0, R = 40, 10, 64
    tab = rng.rand(n_rows, stride).astype(np.float32)
    tab[:, F:] = 0.0
    m0 = (rng.randn(n_rows, stride) * 1e-3).astype(np.float32)
    v0 = (rng.rand(n_rows, stride) * 1e-6).astype(np.float32)
    m0[:, F:] = 0.0
    v0[:, F:] = 0.0
    # global ids: duplicates (up to 7x), ids of other shards on both sides
    idx = rng.randint(0, 70, size=R).astype(np.int32)
    idx[:7] = 23
    G = (rng.randn(R, stride) * 1e-2).astype(np.float32)
    G[:, F:] = 0.0
    t = lambda a: torch.as_tensor(a).to(cd.dev)
    d_tab, d_m, d_v = t(tab.copy()), t(m0.copy()), t(v0.copy())
    head = torch.full((n_rows,), -1, dtype=torch.int32, device=cd.dev)
    nxt = torch.zeros(R, dtype=torch.int32, device=cd.dev)
    step = torch.tensor([4], dtype=torch.int64, device=cd.dev)           # 0-based counter -> t = 5
    cd.ops.table_adam_rows(d_tab, row0, F, t(idx), t(G), d_m, d_v, head, nxt, 0.01, 1, t_dev=step)
    want_t, want_m, want_v = otable.table_adam_rows(tab[:, :F], m0[:, :F], v0[:, :F], idx, G[:, :F], 5, 0.01, row0=row0)
    got_t, got_m, got_v = (a.cpu().numpy() for a in (d_tab, d_m, d_v))
    np.testing.assert_allclose(got_m[:, :F], want_m, atol=1e-7)
    np.testing.assert_allclose(got_v[:, :F], want_v, atol=1e-9)
    np.testing.assert_allclose(got_t[:, :F], want_t, atol=2e-6)
    assert (got_t[:, F:] == 0).all() and (got_m[:, F:] == 0).all()          # pad columns stay zero
    touched = np.unique(idx[(idx >= row0) & (idx < row0 + n_rows)]) - row0
    untouched = np.setdiff1d(np.arange(n_rows), touched)
    assert np.array_equal(got_t[untouched], tab[untouched]) and np.array_equal(got_m[untouched], m0[untouched])
    assert (got_t[touched, :F] != tab[touched, :F]).any()
    assert int((head != -1).sum().item()) == 0                              # scratch is clean again
    # scheduling-independent: a second run from the same state gives the same bits
    d2, m2, v2 = t(tab.copy()), t(m0.copy()), t(v0.copy())
    cd.ops.table_adam_rows(d2, row0, F, t(idx), t(G), m2, v2, head, nxt, 0.01, 1, t_dev=step)
    assert torch.equal(d2, d_tab) and torch.equal(m2, d_m) and torch.equal(v2, d_v)
    # grad_scale (1/world in a data-parallel run): the m slot is linear in the gradient, so it
    # shows the scale that Adam's m/sqrt(v) ratio would hide in the rows themselves
    d3, m3, v3 = t(tab.copy()), t(m0.copy()), t(v0.copy())
    cd.ops.table_adam_rows(d3, row0, F, t(idx), t(G), m3, v3, head, nxt, 0.01, 1, t_dev=step, grad_scale=0.25)
    w3, wm3, wv3 = otable.table_adam_rows(tab[:, :F], m0[:, :F], v0[:, :F], idx, 0.25 * G[:, :F].astype(np.float64),
                                          5, 0.01, row0=row0)
    np.testing.assert_allclose(m3.cpu().numpy()[:, :F], wm3, atol=1e-7)
    np.testing.assert_allclose(v3.cpu().numpy()[:, :F], wv3, atol=1e-9)
    np.testing.assert_allclose(d3.cpu().numpy()[:, :F], w3, atol=5e-6)
    assert not np.allclose(m3.cpu().numpy()[touched, :F], got_m[touched, :F], atol=1e-7)


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f16x2"])
def test_train_step_with_trainable_table(cd, precision):
    """The step's table update = oracle update fed with the device's own dz1 and W1 -- on both fp32 paths (round 6: on the
    split-fp32 path the row gradient dz1 . W1^T is a sixth product on the plane kernels, W1's planes in their natural
    orientation written by the Adam launch)."""
    N, F, H, D, B = 400, 96, 160, 32, 128
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 60, 0)).to(cd.dev)
    ts = cd.train.TrainStep(table, pairs, B, hidden_size=H, output_size=D, mode="uniform", device=cd.dev,
                            train_table=True, precision=precision)
    before = table.data.clone()
    ts.fetch(); ts.forward_loss(); ts.backward()
    W1 = ts.params.unpadded()[0].cpu().numpy().astype(np.float64)
    dz1 = (ts.ws.dz1_f32() if precision in ("f32x3", "f16x2") else ts.ws.dz1)[:, :H].cpu().numpy().astype(np.float64)
    idx = ts.idx.cpu().numpy()
    ts.update_table()
    torch.cuda.synchronize()
    G = otable.grad_xhat(dz1, W1)
    got_G = ts.dxh[:, :F].cpu().numpy().astype(np.float64)
    assert np.linalg.norm(got_G - G) <= 1e-5 * np.linalg.norm(G)          # the row gradient itself, before Adam's sign amplifier
    want, _, _ = otable.table_adam_rows(before[:, :F].cpu().numpy(), np.zeros((N, F)), np.zeros((N, F)), idx, G, 1, 0.01)
    got = table.data[:, :F].cpu().numpy()
    # the first Adam step moves every touched entry by lr * sign(g): entries whose gradient is
    # within fp32 noise of zero may flip, everything else must agree
    diff = np.abs(got - want)
    assert np.mean(diff > 1e-5) < 0.01 and diff.max() <= 0.0201
    touched = np.unique(idx)
    assert np.abs(got[touched] - before[touched, :F].cpu().numpy()).max() > 0.005
    untouched = np.setdiff1d(np.arange(N), touched)
    assert np.array_equal(got[untouched], before[untouched, :F].cpu().numpy())
    if precision == "f16x2":                               # (an eager path: its scales are kernel arguments)
        for _ in range(3):
            ts.step()
        assert np.isfinite(ts.loss())
        L, sw = ts.layout, ts.ws.scales.w1                  # the natural-orientation planes hold the updated W1 (22 bits of it)
        w1n = (ts.ws.W1n[:, :L.Hp].double() + ts.ws.W1n[:, L.Hp:].double()) / sw
        assert (w1n - ts.params.W1.double()).abs().max().item() <= 2.0 ** -22 * ts.params.W1.abs().max().item()
        return
    # and the whole step runs, eagerly and from a captured graph, to the same bits
    mk = lambda g: cd.train.TrainStep(cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev), pairs, B, hidden_size=H,
                                      output_size=D, mode="uniform", device=cd.dev, train_table=True, use_graph=g,
                                      precision=precision)
    a, b = mk(False), mk(True)
    for _ in range(4):
        a.step(); b.step()
    torch.cuda.synchronize()
    assert torch.equal(a.table.data, b.table.data) and torch.equal(a.params.flat, b.params.flat)
    assert not torch.equal(a.table.data, before)
    if precision == "f32x3":                               # the natural-orientation planes ARE the updated W1
        L = a.layout
        w1n = a.ws.W1n[:, :L.Hp].float() + a.ws.W1n[:, L.Hp:2 * L.Hp].float() + a.ws.W1n[:, 2 * L.Hp:].float()
        assert torch.equal(w1n, a.params.W1)


@pytest.mark.parametrize("precision", ["f32", "f32x3", "f16x2"])
def test_trainable_table_production_width(cd, precision):
    """the trainable catalogue at PRODUCTION width (F = 1500, H = 5000, D = 256), 120 000 rows, batch 512
    triplets (1 536 gathered rows, duplicates included), on both fp32 paths: the row gradient dz1 . W1^T (K = 5 000) against
    fp64 on the device's own operands, then the lazy-Adam row update against oracle/table.py; rows nobody gathered keep
    their bits; two steps run and move the rows again."""
    N, F, H, D, B = 120000, 1500, 5000, 256, 512
    table = cd.engine.FeatureTable.synthetic(N, F, 0, cd.dev)
    pairs = torch.as_tensor(osynth.cowatch_pairs(N, 4000, 0)).to(cd.dev)
    ts = cd.train.TrainStep(table, pairs, B, hidden_size=H, output_size=D, mode="uniform", device=cd.dev,
                            train_table=True, precision=precision)
    ts.fetch(); ts.forward_loss(); ts.backward()
    idx = ts.idx.cpu().numpy()
    touched = np.unique(idx)
    before = table.data[torch.as_tensor(touched).to(cd.dev).long()][:, :F].cpu().numpy()
    probe = torch.arange(0, N, 997, device=cd.dev)                        # a sample of the whole table for the "untouched" check
    before_probe = table.data[probe].clone()
    W1 = ts.params.unpadded()[0].cpu().numpy().astype(np.float64)
    dz1 = (ts.ws.dz1_f32() if precision in ("f32x3", "f16x2") else ts.ws.dz1)[:, :H].cpu().numpy().astype(np.float64)
    ts.update_table()
    torch.cuda.synchronize()
    G = otable.grad_xhat(dz1, W1)
    got_G = ts.dxh[:, :F].cpu().numpy().astype(np.float64)
    rel = np.linalg.norm(got_G - G) / np.linalg.norm(G)
    assert rel <= 1e-5, rel
    assert (ts.dxh[:, F:] == 0).all()                                     # pad columns of the row gradient
    # oracle update on the touched rows only (ids remapped to their position among the touched rows)
    remap = np.searchsorted(touched, idx)
    want, _, _ = otable.table_adam_rows(before, np.zeros_like(before), np.zeros_like(before), remap, G, 1, 0.01)
    got = table.data[torch.as_tensor(touched).to(cd.dev).long()][:, :F].cpu().numpy()
    diff = np.abs(got - want)
    assert np.mean(diff > 1e-5) < 0.01 and diff.max() <= 0.0201, (float(np.mean(diff > 1e-5)), float(diff.max()))
    assert np.abs(got - before).max() > 0.005
    mask = ~torch.isin(probe, torch.as_tensor(touched).to(cd.dev))
    assert torch.equal(table.data[probe][mask], before_probe[mask])
    ts.apply_gradients()
    ts.global_step += 1
    for _ in range(2):
        ts.step()
    torch.cuda.synchronize()
    assert np.isfinite(ts.loss()) and bool(torch.isfinite(table.data[probe]).all())
