"""Pin the oracle: golden fixtures produced by the reference's own modules
(tests/golden/make_golden.py), the hand-derived loss known answers, Random123
known-answer vectors for Philox, and torch-CPU autograd for the gradients."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import sampler, synth, tower


def _kat(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------- sampler ----
@pytest.mark.parametrize("seed", [0, 1234])
@pytest.mark.parametrize("n_rows", [3, 10000])
def test_reference_sampler_matches_golden(golden_dir, seed, n_rows):
    g = np.load(os.path.join(golden_dir, f"sampler_ref_seed{seed}_n{n_rows}.npz"))
    got = sampler.reference_triplets(g["pairs"], int(g["n_rows"]), int(g["seed"]))
    np.testing.assert_array_equal(got, g["triplets"])
    # the rule itself: negative never equals anchor or positive
    assert not np.any(got[:, 2] == got[:, 0])
    assert not np.any(got[:, 2] == got[:, 1])


def test_reference_negative_stream_kat(golden_dir):
    want = _kat(golden_dir)["neg_stream_seed1234_n10000"]
    assert want == [8915, 1318, 7221, 7540, 664, 6137, 6833, 8471]
    rs = np.random.RandomState(1234)
    assert [int(rs.randint(0, 10000)) for _ in range(8)] == want


def test_philox_random123_known_answers():
    def run(ctr, key):
        return [int(v) for v in sampler.philox4x32_10(ctr, key)]
    assert run((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == \
        [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344),
               (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


@pytest.mark.parametrize("n_rows,batch", [(10000, 128), (3, 32), (5, 17)])
def test_device_sampler_scalar_vs_vector(n_rows, batch):
    rng = np.random.RandomState(1)
    pairs = rng.randint(0, n_rows, size=(100, 2))
    pairs = pairs[pairs[:, 0] != pairs[:, 1]]
    for step in (0, 1, 7, 2 ** 33 + 5):
        a = sampler.device_triplets(pairs, n_rows, 1234, step, batch)
        b = sampler.device_triplets_vec(pairs, n_rows, 1234, step, batch)
        np.testing.assert_array_equal(a, b)
        assert not np.any(a[:, 2] == a[:, 0]) and not np.any(a[:, 2] == a[:, 1])
        assert a[:, 2].min() >= 0 and a[:, 2].max() < n_rows
        # sequential pair stream with wrap-around (inputs.py:110-122)
        q = (step * batch + np.arange(batch)) % len(pairs)
        np.testing.assert_array_equal(a[:, :2], pairs[q])


def test_device_sampler_rank_slices_agree():
    pairs = synth.cowatch_pairs(1000, 300, 0)
    full = sampler.device_triplets_vec(pairs, 1000, 7, 3, 64)
    halves = [sampler.device_triplets_vec(pairs, 1000, 7, 3, 32, slot0=r * 32,
                                          batch_global=64) for r in (0, 1)]
    np.testing.assert_array_equal(full, np.concatenate(halves))


def test_device_sampler_is_uniform():
    pairs = np.array([[0, 1]])
    n = np.concatenate([sampler.device_triplets_vec(pairs, 10, 5, s, 4096)[:, 2]
                        for s in range(4)])
    counts = np.bincount(n, minlength=10)
    assert counts[0] == 0 and counts[1] == 0
    assert np.all(np.abs(counts[2:] / counts[2:].sum() - 1 / 8) < 0.01)


def test_inbatch_spec():
    pairs = synth.cowatch_pairs(50, 40, 1)
    for step in range(5):
        rows, tri, valid, s = sampler.device_inbatch(pairs, 9, step, 16)
        assert 1 <= s <= 15
        ap = rows.reshape(-1, 2)
        for i in range(16):
            j = (i + s) % 16
            assert tuple(tri[i]) == (2 * i, 2 * i + 1, 2 * j + 1)
            assert valid[i] == (ap[j, 1] not in (ap[i, 0], ap[i, 1]))
    assert sampler.num_batches(10, 3, 4) == 7


# ------------------------------------------------------------------ synth ----
def test_synth_features_match_imitation_data(golden_dir):
    g = np.load(os.path.join(golden_dir, "imitation_features_seed0.npz"))
    got = synth.features_numpy(8, 1500, seed=0)
    np.testing.assert_array_equal(got, g["features"])
    assert got.dtype == np.float64 and got.min() >= 0 and got.max() < 1
    assert _kat(golden_dir)["gen_triplets_shape"] == [4, 3, 16]


def test_synth_pairs_shape():
    p = synth.cowatch_pairs(10000, 3000, 0)
    assert p.dtype == np.int32 and p.shape[1] == 2
    assert not np.any(p[:, 0] == p[:, 1])
    assert 3000 * 5 < len(p) < 3000 * 29
    kat_pairs = [[1, 2], [3, 4], [4, 5], [5, 6], [7, 8], [8, 9]]
    assert len(kat_pairs) == 6


def test_get_all_cowatch_kat(golden_dir):
    k = _kat(golden_dir)
    assert k["get_all_cowatch_len"] == 6
    assert k["get_all_cowatch_sorted"] == [[1, 2], [3, 4], [4, 5], [5, 6], [7, 8], [8, 9]]


# ------------------------------------------------------------------- loss ----
@pytest.mark.parametrize("margin", [0.1, 0.8])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hinge_loss_known_answer(golden_dir, margin, dtype):
    k = _kat(golden_dir)
    out = tower.hinge_loss(np.array(k["loss_input"], dtype), margin, dtype)
    want = k[f"loss_margin_{margin}"]
    assert out["pos_dist"].shape == (5, 1) and out["hinge_dist"].shape == (5, 1)
    np.testing.assert_allclose(out["pos_dist"][:, 0], k["loss_pos_dist"])
    np.testing.assert_allclose(out["neg_dist"][:, 0], k["loss_neg_dist"])
    np.testing.assert_allclose(out["hinge_dist"][:, 0], want["hinge_dist"], rtol=1e-6)
    np.testing.assert_allclose(out["hinge_loss"], want["hinge_loss"], rtol=1e-6)
    assert out["anchors"].shape == (5, 1, 2)


def test_eval_mean_dist_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "evaluate_mean_dist.npz"))
    emb, cw = g["embeddings"], g["eval_cowatches"]
    d = np.mean(np.sum(np.square(emb[cw[:, 0]] - emb[cw[:, 1]]), axis=-1))
    np.testing.assert_allclose(d, float(g["mean_dist"]), rtol=1e-6)


# ---------------------------------------------------------- tower + grads ----
def _torch_reference(x, params, margin):
    """The same graph written with torch ops (NOT F.normalize / nn.Linear
    defaults -- TF semantics spelled out) so autograd checks the hand backward."""
    W1, b1, W2, b2 = [torch.tensor(p, dtype=torch.float64, requires_grad=True)
                      for p in params]
    x = torch.tensor(x, dtype=torch.float64)

    def l2n(t):
        return t * torch.rsqrt(torch.clamp((t * t).sum(-1, keepdim=True), min=1e-12))

    def lrelu(t):
        return torch.maximum(0.2 * t, t)

    h1 = lrelu(l2n(x) @ W1 + b1)
    z = lrelu(h1 @ W2 + b2)
    e = l2n(z).reshape(-1, 3, z.shape[-1])
    a, p, n = e[:, 0], e[:, 1], e[:, 2]
    hinge = torch.clamp(((a - p) ** 2).sum(-1) - ((a - n) ** 2).sum(-1) + margin, min=0)
    loss = hinge.mean()
    loss.backward()
    return loss.item(), e.detach().numpy(), [t.grad.numpy() for t in (W1, b1, W2, b2)]


def _small_problem(B=6, F=40, H=24, D=8, seed=0):
    rng = np.random.RandomState(seed)
    x = rng.random_sample((3 * B, F))
    params = [tower.xavier_uniform(rng, F, H, np.float64), 0.1 * rng.randn(H),
              tower.xavier_uniform(rng, H, D, np.float64), 0.1 * rng.randn(D)]
    return x, params


def test_tower_and_grads_vs_torch_autograd_fp64():
    x, params = _small_problem()
    fwd, loss, grads = tower.train_step_grads(x, params, 0.8, np.float64)
    tl, te, tg = _torch_reference(x, params, 0.8)
    np.testing.assert_allclose(loss["hinge_loss"], tl, rtol=1e-12)
    np.testing.assert_allclose(fwd["l2_norm"].reshape(te.shape), te, atol=1e-13)
    for name, g in zip(("dW1", "db1", "dW2", "db2"), tg):
        np.testing.assert_allclose(grads[name], g, atol=1e-13, err_msg=name)
    assert np.any(loss["hinge_dist"] > 0)


def test_tower_fp32_close_to_fp64():
    x, params = _small_problem(B=16, F=300, H=200, D=32, seed=3)
    f32, l32, g32 = tower.train_step_grads(x, params, 0.8, np.float32)
    f64, l64, g64 = tower.train_step_grads(x, params, 0.8, np.float64)
    assert f32["l2_norm"].dtype == np.float32
    np.testing.assert_allclose(f32["l2_norm"], f64["l2_norm"], atol=1e-5)
    np.testing.assert_allclose(l32["hinge_loss"], l64["hinge_loss"], atol=1e-5)
    np.testing.assert_allclose(np.linalg.norm(f32["l2_norm"], axis=-1), 1, atol=1e-5)


def test_l2_normalize_is_tf_form_not_torch_form():
    x = np.array([[3e-7, 4e-7]], np.float64)      # |x|^2 = 2.5e-13 < eps
    y, inv = tower.l2_normalize(x, np.float64)
    np.testing.assert_allclose(inv, 1e6)          # rsqrt(1e-12), not 1/|x|
    np.testing.assert_allclose(y, x * 1e6)
    g = np.array([[1.0, 2.0]])
    np.testing.assert_allclose(tower.l2_normalize_backward(x, inv, g, np.float64), g * 1e6)


def test_indexed_loss_matches_plain_loss_on_reference_layout():
    rng = np.random.RandomState(2)
    E = rng.randn(30, 8)
    tri = np.arange(30).reshape(10, 3)
    valid = np.ones(10, bool)
    a = tower.hinge_loss_indexed(E, tri, valid, 0.8, np.float64)
    b = tower.hinge_loss(E.reshape(10, 3, 8), 0.8, np.float64)
    np.testing.assert_allclose(a["hinge_loss"], b["hinge_loss"])
    da = tower.hinge_loss_indexed_backward(E, tri, valid, 0.8, np.float64)
    db = tower.hinge_loss_backward(E.reshape(10, 3, 8), 0.8, np.float64)
    np.testing.assert_allclose(da, db.reshape(30, 8))


def test_indexed_loss_backward_vs_autograd():
    rng = np.random.RandomState(4)
    E = rng.randn(16, 8)
    _, tri, valid, _ = sampler.device_inbatch(np.array([[1, 2], [3, 4], [2, 5], [6, 1]]),
                                              3, 0, 8)
    valid = valid.astype(bool)
    Et = torch.tensor(E, requires_grad=True)
    a, p, n = Et[tri[:, 0]], Et[tri[:, 1]], Et[tri[:, 2]]
    h = torch.clamp(((a - p) ** 2).sum(-1) - ((a - n) ** 2).sum(-1) + 0.8, min=0)
    (h * torch.tensor(valid)).sum().div(8).backward()
    got = tower.hinge_loss_indexed_backward(E, tri, valid, 0.8, np.float64)
    np.testing.assert_allclose(got, Et.grad.numpy(), atol=1e-13)


# -------------------------------------------------------------- optimizers ---
def test_adam_tf_form():
    w = np.array([1.0, -2.0]); g = np.array([0.5, 0.25])
    m = np.zeros(2); v = np.zeros(2)
    w1, m1, v1 = tower.adam_step(w, g, m, v, 1, 0.01, dtype=np.float64)
    lr_t = 0.01 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(m1, 0.1 * g, rtol=1e-12)
    np.testing.assert_allclose(v1, 0.001 * g * g, rtol=1e-12)
    _, _, v32 = tower.adam_step(w, g, m, v, 1, 0.01, dtype=np.float32)
    np.testing.assert_allclose(v32, np.float32(1 - np.float32(0.999)) * np.float32(g) ** 2, rtol=1e-6)
    np.testing.assert_allclose(w1, w - lr_t * m1 / (np.sqrt(v1) + 1e-8))
    # differs from torch.optim.Adam's epsilon placement when g is tiny
    g2 = np.array([1e-9, 1e-9])
    w2, _, _ = tower.adam_step(w, g2, m, v, 1, 0.01, dtype=np.float64)
    tw = torch.tensor(w, requires_grad=True)
    opt = torch.optim.Adam([tw], lr=0.01)
    tw.grad = torch.tensor(g2); opt.step()
    assert np.abs(w2 - tw.detach().numpy()).max() > 1e-4


def test_lars_and_lr_schedule():
    w = np.array([3.0, 4.0]); g = np.array([0.6, 0.8]); acc = np.zeros(2)
    w1, acc1 = tower.lars_step(w, g, acc, 1.0, dtype=np.float64)
    trust = 1e-3 * 5.0 / (1.0 + 1e-4 * 5.0)
    np.testing.assert_allclose(acc1, trust * (g + 1e-4 * w))
    np.testing.assert_allclose(w1, w - acc1)
    w0, _ = tower.lars_step(np.zeros(2), g, acc, 1.0, dtype=np.float64)
    np.testing.assert_allclose(w0, -g)           # trust 1 when |w| == 0
    assert tower.exponential_decay(1.0, 999999, 1000000, 0.96) == 1.0
    np.testing.assert_allclose(tower.exponential_decay(1.0, 2500000, 1000000, 0.96), 0.96 ** 2)


def test_calc_var():
    t = np.random.RandomState(0).randn(5, 3, 4)
    want = np.mean((t - t.mean(axis=(0, 1))) ** 2)
    np.testing.assert_allclose(tower.calc_var(t, np.float64), want)


# ----------------------------------------------------------- fusion towers (N4) ---
@pytest.mark.parametrize("net", ["MultiplyNet", "MlpNet", "ResNet", "ResNetV2"])
def test_fusion_towers_vs_torch_autograd(net):
    rng = np.random.RandomState(0)
    shapes = tower.fusion_layer_shapes(net, visual=20, doc=6, hidden_v=16, hidden_d=10, out=8, mlp_hidden=12)
    P = {k: (tower.xavier_uniform(rng, fi, fo, np.float64), np.full(fo, 0.1)) for k, (fi, fo) in shapes.items()}
    x = rng.random_sample((9, 26))
    t = tower.fusion_forward(net, x, P, visual=20)
    dE = rng.randn(9, 8)
    g = tower.fusion_backward(net, t, P, dE)

    tp = {k: (torch.tensor(w, requires_grad=True), torch.tensor(b, requires_grad=True)) for k, (w, b) in P.items()}
    l2n = lambda a: a * torch.rsqrt(torch.clamp((a * a).sum(-1, keepdim=True), min=1e-12))
    fc = lambda a, n: torch.maximum(0.2 * (a @ tp[n][0] + tp[n][1]), a @ tp[n][0] + tp[n][1])
    xt = torch.tensor(x)
    if net == "ResNetV2":                                  # written out as models.py:219-238 writes it
        xv, xd = l2n(xt[:, :20]), l2n(xt[:, 20:])
        v12, v21 = fc(fc(xv, "layer_visual_1_1"), "layer_visual_1_2"), fc(xv, "layer_visual_2_1")
        d12, d21 = fc(fc(xd, "layer_doc_1_1"), "layer_doc_1_2"), fc(xd, "layer_doc_2_1")
        r1 = v12 * d12 + v12 * d21 + v21 * d12 + v21 * d21 + v12 + v21 + d12 + d21
        r2 = r1 + fc(r1, "layer_fusion_1")
        pre = r2 + fc(r2, "layer_fusion_2")
        v2 = d2 = None
    else:
        v2 = fc(fc(l2n(xt[:, :20]), "layer_visual_1"), "layer_visual_2")
        d2 = fc(fc(l2n(xt[:, 20:]), "layer_doc_1"), "layer_doc_2")
    if net == "ResNetV2":
        pass
    elif net == "MultiplyNet":
        pre = v2 * d2
    elif net == "MlpNet":
        pre = fc(fc(v2 * d2, "layer_fusion_1"), "layer_fusion_2")
    else:
        r1 = v2 * d2 + v2 + d2
        r2 = r1 + fc(r1, "layer_fusion_1")
        pre = r2 + fc(r2, "layer_fusion_2")
    out = l2n(pre)
    (out * torch.tensor(dE)).sum().backward()
    np.testing.assert_allclose(t["l2_norm"], out.detach().numpy(), atol=1e-13)
    for k in P:
        np.testing.assert_allclose(g[k][0], tp[k][0].grad.numpy(), atol=1e-12, err_msg=k)
        np.testing.assert_allclose(g[k][1], tp[k][1].grad.numpy(), atol=1e-12, err_msg=k)


def test_knn_oracle_known_geometry():
    """oracle/knn.py on a configuration with a known answer: points on a circle -- the
    neighbours of a point are itself, then its ring neighbours in order of angular distance."""
    from oracle import knn as oknn
    n = 24
    ang = 2 * np.pi * np.arange(n) / n
    pts = np.stack([np.cos(ang), np.sin(ang)], 1).astype(np.float32) * 3.0    # not unit length: l2_norm fixes it
    D, I, _ = oknn.calc_knn_exact(pts, nearest_num=5)
    assert (I[:, 0] == np.arange(n)).all() and np.allclose(D[:, 0], 0.0, atol=1e-12)
    for r in range(n):
        assert set(I[r, 1:3]) == {(r - 1) % n, (r + 1) % n}
        assert set(I[r, 3:5]) == {(r - 2) % n, (r + 2) % n}
    chord = 2 - 2 * np.cos(2 * np.pi / n)                                       # squared chord of unit circle
    assert np.allclose(D[:, 1:3], chord, atol=1e-6)
    dup = np.concatenate([pts, pts[5:6], pts[5:6]])                             # exact ties: ordered by id
    _, Id, _ = oknn.calc_knn_exact(dup, nearest_num=3)
    assert list(Id[5]) == [5, n, n + 1] and list(Id[n + 1]) == [5, n, n + 1]
    D2, I2, _ = oknn.calc_knn_exact(pts[:3], nearest_num=5)                     # fewer rows than k
    assert (I2[:, 3:] == -1).all() and np.isinf(D2[:, 3:]).all()


def test_table_oracle_l2norm_backward_and_duplicates():
    """oracle/table.py: the row gradient is the l2-normalisation backward of the summed
    duplicates (finite differences), and only touched rows move."""
    from oracle import table as otable
    rng = np.random.RandomState(0)
    tab = rng.rand(6, 9)
    G = rng.randn(5, 9)
    idx = np.array([2, 4, 2, 9, 2])                       # row 2 three times, row 9 belongs to another shard
    # one Adam step from zero slots moves w by lr * dx / (|dx| + eps'): recover the sign pattern, and the
    # magnitude through m = (1 - beta1) * dx
    new, m, v = otable.table_adam_rows(tab, np.zeros_like(tab), np.zeros_like(tab), idx, G, 1, 0.01)
    g2 = G[0] + G[2] + G[4]
    f = lambda x: float((x / np.sqrt(x @ x)) @ g2)        # loss whose d/dx_hat is g2
    num = np.array([(f(tab[2] + 1e-6 * e) - f(tab[2] - 1e-6 * e)) / 2e-6 for e in np.eye(9)])
    np.testing.assert_allclose(m[2] / (1 - 0.9), num, atol=1e-6)
    np.testing.assert_allclose(new[2], tab[2] - 0.01 * np.sign(num), atol=1e-6)   # first Adam step = lr * sign
    assert np.array_equal(new[[0, 1, 3, 5]], tab[[0, 1, 3, 5]]) and not np.array_equal(new[4], tab[4])
    shard, _, _ = otable.table_adam_rows(tab[3:], np.zeros((3, 9)), np.zeros((3, 9)), idx, G, 1, 0.01, row0=3)
    np.testing.assert_array_equal(shard, new[3:])         # a shard sees only its own rows


def test_etl_oracle_matches_reference_cowatch_graph(golden_dir):
    """oracle/etl.py against the outputs of the reference's get_cowatch_graph / select_cowatch
    (fixture G6, tests/golden/make_golden.py)."""
    from oracle import etl
    g = np.load(os.path.join(golden_dir, "cowatch_graph_seed7.npz"))
    edges, counts = etl.cowatch_graph(g["cowatches"])
    np.testing.assert_array_equal(edges, g["edges"])
    np.testing.assert_array_equal(counts, g["counts"])
    for t in (1, 2, 3, 5):
        np.testing.assert_array_equal(etl.select_cowatch(g["cowatches"], t), g["select_t%d" % t])
    np.testing.assert_array_equal(etl.select_cowatch(g["cowatches"], 3, unique=True), g["unique_t3_sorted"])
    assert len(g["select_t1"]) == len(g["cowatches"]) > len(g["select_t2"]) > len(g["select_t5"]) > 0
    with pytest.raises(RuntimeError):
        etl.cowatch_graph([[1, 2], [3, 3]])


def test_torch_cpu_step_equals_numpy_oracle():
    """oracle/tower_torch.py (the CPU baseline that bench.py times) is the same step as
    oracle/tower.py: embeddings, loss, gradients and the TF-form Adam update."""
    from oracle import tower_torch
    rng = np.random.RandomState(0)
    N, F, H, D, B = 300, 40, 50, 16, 8
    table = rng.random_sample((N, F)).astype(np.float32)
    pairs = synth.cowatch_pairs(N, 40, 0)
    st = tower_torch.CpuStep(table, pairs, B, hidden=H, out=D, margin=0.8, lr=0.01)
    W = [w.detach().numpy().copy() for w in st.W]
    idx = st.sample()
    assert idx.shape == (B, 3) and np.array_equal(idx[:, :2], pairs[:B])
    assert np.all(idx[:, 2] != idx[:, 0]) and np.all(idx[:, 2] != idx[:, 1])      # the reference's rule
    x = st.fetch(idx)
    np.testing.assert_array_equal(x.numpy(), sampler.gather(table, idx).reshape(-1, F))
    e, loss, grads = st.train(x)
    fwd, wl, wg = tower.train_step_grads(x.numpy().astype(np.float64), [w.astype(np.float64) for w in W], 0.8,
                                          np.float64)
    np.testing.assert_allclose(e.detach().numpy(), fwd["l2_norm"], atol=1e-6)
    assert abs(loss - float(wl["hinge_loss"])) < 1e-6
    for g, k, w0, w1 in zip(grads, ("dW1", "db1", "dW2", "db2"), W, st.W):
        np.testing.assert_allclose(g.numpy(), wg[k], atol=1e-6)
        want, _, _ = tower.adam_step(w0, g.numpy(), 0 * w0, 0 * w0, 1, 0.01, dtype=np.float32)
        np.testing.assert_allclose(w1.detach().numpy(), want, atol=1e-6)
    tf, tt, n = tower_torch.time_steps(st, 4, 0.5, 1)
    assert n >= 3 and tf > 0 and tt > 0


def test_build_graph_switches_of_the_oracle():
    """regularization_penalty and clip_gradient_norm (train.py:133-145) and calc_var
    (train.py:67-71) against torch autograd / their definitions."""
    rng = np.random.RandomState(2)
    x, params = _rand_problem(rng, B=6, F=20, H=30, D=8) if "_rand_problem" in globals() else (None, None)
    if x is None:
        x = rng.random_sample((18, 20))
        params = [tower.xavier_uniform(rng, 20, 30, np.float64), rng.randn(30) * 0.1,
                  tower.xavier_uniform(rng, 30, 8, np.float64), rng.randn(8) * 0.1]
    fwd, loss, grads = tower.train_step_grads(x, params, 0.8, np.float64)
    reg_g, reg = tower.regularized_grads(grads, params, 7.0, l2_penalty=0.01, dtype=np.float64)
    W = [torch.tensor(p, dtype=torch.float64, requires_grad=True) for p in params]
    l2n = lambda t: t * torch.rsqrt(torch.clamp((t * t).sum(-1, keepdim=True), min=1e-12))
    lre = lambda t: torch.maximum(0.2 * t, t)
    e = l2n(lre(lre(l2n(torch.tensor(x)) @ W[0] + W[1]) @ W[2] + W[3])).view(-1, 3, 8)
    hinge = torch.clamp(((e[:, 0] - e[:, 1]) ** 2).sum(-1) - ((e[:, 0] - e[:, 2]) ** 2).sum(-1) + 0.8, min=0).mean()
    reg_t = 0.01 * ((W[0] ** 2).sum() + (W[2] ** 2).sum()) / 2          # slim.l2_regularizer = scale * l2_loss
    (7.0 * reg_t + hinge).backward()                                     # train.py:139
    assert abs(float(reg_t) - float(reg)) < 1e-12
    for w, k in zip(W, ("dW1", "db1", "dW2", "db2")):
        np.testing.assert_allclose(w.grad.numpy(), reg_g[k], atol=1e-10)
    g = rng.randn(50)
    for clip in (0.1, 100.0):
        c = tower.clip_by_norm(g, clip, np.float64)
        np.testing.assert_allclose(c, torch.nn.functional.normalize(torch.tensor(g), dim=0).numpy() * min(clip, np.linalg.norm(g)),
                                   atol=1e-12)
    t = rng.randn(5, 3, 4)
    mean = t.reshape(15, 4).mean(0)                                       # reduce_mean over axis [0,1]
    assert abs(float(tower.calc_var(t, np.float64)) - float(((t - mean) ** 2).mean())) < 1e-12
    w, a = tower.momentum_step(np.ones(3), np.full(3, 0.5), np.full(3, 2.0), 0.1, 0.9, True, np.float64)
    np.testing.assert_allclose(a, 2.3)                                    # 2*0.9 + 0.5
    np.testing.assert_allclose(w, 1 - (0.05 + 2.3 * 0.9 * 0.1))
