"""Next-row N4: the reference's fusion towers (models.py:65-157) on the HIP
kernels, against the fp64 oracle (gradient-checked against torch autograd in
tests/test_oracle.py).  Tolerance as for VNet: 1e-5 on embeddings."""
import numpy as np
import pytest
import torch

from oracle import sampler as osampler, synth as osynth, tower as otower

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def cd(gpu):
    import cdml_amd
    from cdml_amd import engine, fusion, losses, models, ops, utils

    class NS:
        pass
    ns = NS()
    ns.dev, ns.engine, ns.fusion, ns.losses, ns.models, ns.ops, ns.utils = gpu, engine, fusion, losses, models, ops, utils
    return ns


def test_elementwise_kernels(cd):
    rng = np.random.RandomState(0)
    a, b, g = (rng.randn(70, 64).astype(np.float32) for _ in range(3))
    da, db, dg = (torch.as_tensor(t).to(cd.dev) for t in (a, b, g))
    out = torch.empty_like(da)
    for mode, want in ((cd.ops.EW_MUL, a * b), (cd.ops.EW_MUL_RES, a * b + a + b), (cd.ops.EW_ADD, a + b)):
        cd.ops.ew_combine(mode, da, db, out, 70, 64)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    oa, ob = torch.empty_like(da), torch.empty_like(da)
    for res in (0, 1):
        cd.ops.ew_fusion_bwd(res, dg, da, db, oa, ob, 70, 64)
        np.testing.assert_allclose(oa.cpu().numpy(), g * (b + res) * np.where(a > 0, 1, 0.2), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ob.cpu().numpy(), g * (a + res) * np.where(b > 0, 1, 0.2), rtol=1e-6, atol=1e-6)
    cd.ops.lrelu_bwd(dg, da, out, 70, 64)
    np.testing.assert_allclose(out.cpu().numpy(), g * np.where(a > 0, 1, 0.2), rtol=1e-6)


@pytest.mark.parametrize("net", ["MultiplyNet", "MlpNet", "ResNet", "ResNetV2"])
def test_fusion_tower_vs_oracle_production_shape(cd, net):
    """feature_size 1628 = 1500 visual ++ 128 doc (online_data.py:38), hidden 5000 / 400."""
    R, F = 96, 1628
    x = np.random.RandomState(1).random_sample((R, F)).astype(np.float32)
    model = cd.utils.find_class_by_name(net, [cd.models])(device=cd.dev, seed=3)
    xt = torch.as_tensor(x).to(cd.dev)
    out = model.create_model(xt, 256)["l2_norm"]
    P = {k: (w.detach().cpu().numpy().astype(np.float64), b.detach().cpu().numpy().astype(np.float64))
         for k, (w, b) in model.params.unpadded().items()}
    assert set(P) == set(otower.FUSION_LAYERS[net])
    first_doc = "layer_doc_1_1" if net == "ResNetV2" else "layer_doc_1"
    assert float(model.params.b(first_doc)[:400].detach().min()) == pytest.approx(0.1)   # bias_init=0.1
    t = otower.fusion_forward(net, x.astype(np.float64), P)
    assert tuple(out.shape) == (R, 256)
    assert np.abs(out.detach().cpu().numpy() - t["l2_norm"]).max() < TOL
    dE = np.random.RandomState(2).randn(R, 256).astype(np.float32) * 0.01
    (out * torch.as_tensor(dE).to(cd.dev)).sum().backward()
    g = otower.fusion_backward(net, t, P, dE.astype(np.float64))
    got = model.params.unpadded(grads=True)
    for k in P:
        for j, nm in ((0, "dW"), (1, "db")):
            w = g[k][j]
            d = np.abs(got[k][j].cpu().numpy() - w).max()
            assert d < max(TOL, 1e-3 * np.abs(w).max()), (k, nm, d, np.abs(w).max())


def test_fusion_train_step_learns(cd):
    """ResNet end to end (sampler -> raw gather -> tower -> hinge -> backward -> Adam) on
    clustered features whose doc part also carries the cluster."""
    rng = np.random.RandomState(0)
    n, vis, doc = 2000, 64, 16
    centers = rng.random_sample((10, vis + doc))
    cid = rng.randint(0, 10, size=n)
    feats = (centers[cid] + 0.05 * rng.randn(n, vis + doc)).clip(0, None).astype(np.float32)
    a = rng.randint(0, n, size=5000)
    p = np.array([rng.choice(np.flatnonzero(cid == cid[i])) for i in a])
    pairs = np.stack([a, p], 1)
    pairs = pairs[pairs[:, 0] != pairs[:, 1]].astype(np.int32)
    table = cd.engine.FeatureTable.from_numpy(feats, cd.dev)
    ts = cd.fusion.FusionTrainStep("ResNet", table, torch.as_tensor(pairs).to(cd.dev), 64, margin=0.8,
                                   base_learning_rate=0.002, device=cd.dev, visual_size=vis, hidden_v=128,
                                   hidden_d=32, output_size=32)
    losses = []
    for _ in range(120):
        ts.step()
        if ts.global_step == 1 or ts.global_step % 30 == 0:
            losses.append(ts.loss())
    assert max(losses[2:]) < 0.5 * losses[0], losses
    # first step against the oracle: same triplets, same forward
    ts2 = cd.fusion.FusionTrainStep("ResNet", table, torch.as_tensor(pairs).to(cd.dev), 64, device=cd.dev,
                                    visual_size=vis, hidden_v=128, hidden_d=32, output_size=32)
    P = {k: (w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64))
         for k, (w, b) in ts2.params.unpadded().items()}
    ts2.step()
    idx = osampler.device_triplets_vec(pairs, n, 1234, 0, 64)
    t = otower.fusion_forward("ResNet", feats[idx.reshape(-1)].astype(np.float64), P, visual=vis)
    loss = otower.hinge_loss(t["l2_norm"].reshape(-1, 3, 32), 0.8, np.float64)
    assert abs(ts2.loss() - float(loss["hinge_loss"])) < TOL


@pytest.mark.parametrize("precision", ["f32x3", "f16x2"])
@pytest.mark.parametrize("net", ["MultiplyNet", "MlpNet", "ResNet"])
def test_fusion_visual_branch_on_the_plane_kernels(cd, net, precision):
    """Round 6: the visual branch of a fusion tower (1500 -> 5000 -> 256: VNet's two layers, 97 % of the tower's flop) on
    the plane kernels -- fp32 operands as three exact bf16 planes, six plane products per fp32 product -- at production
    shape, against the fp64 oracle with the fp32 tower's bounds, forward and every gradient; and against the fp32-MFMA
    tower on the same weights (models.py:65-157)."""
    R, F = 384, 1628
    x = np.random.RandomState(1).random_sample((R, F)).astype(np.float32)
    xt = torch.as_tensor(x).to(cd.dev)
    params = cd.fusion.FusionParams(net, cd.dev, doc_size=F - 1500, seed=3)
    tower = cd.fusion.FusionTower(params, R, precision=precision)      # (f16x2: two fp16 planes, three products, its scales calibrated by this first pass)
    assert tower.vx3 is not None and tower.vh2 == (precision == "f16x2")
    out = tower.forward(xt).clone()
    P = {k: (w.detach().cpu().numpy().astype(np.float64), b.detach().cpu().numpy().astype(np.float64))
         for k, (w, b) in params.unpadded().items()}
    t = otower.fusion_forward(net, x.astype(np.float64), P)
    assert np.abs(out[:, :256].cpu().numpy() - t["l2_norm"]).max() < TOL
    dE = np.random.RandomState(2).randn(R, 256).astype(np.float32) * 0.01
    tower.de.zero_()
    tower.de[:, :256] = torch.as_tensor(dE).to(cd.dev)
    params.grad.zero_()
    tower.backward()
    g = otower.fusion_backward(net, t, P, dE.astype(np.float64))
    got = params.unpadded(grads=True)
    for k in P:
        for j, nm in ((0, "dW"), (1, "db")):
            w = g[k][j]
            d = np.abs(got[k][j].cpu().numpy() - w).max()
            assert d < max(TOL, 1e-3 * np.abs(w).max()), (k, nm, d, np.abs(w).max())
    ref = cd.fusion.FusionTower(params, R, precision="f32")
    assert ref.vx3 is None
    assert float((ref.forward(xt) - out).abs().max()) < 2e-6
