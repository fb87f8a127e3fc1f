"""Fusion towers of the reference (models.py:65-157) on the HIP kernels:
MultiplyNet, MlpNet and ResNet -- visual (first 1500 columns) and doc features
go through two FC branches, are fused by an elementwise product and, for
MlpNet / ResNet, pass two more FC layers (with residual sums in ResNet).

Every fully_connected is the fp32 MFMA GEMM of ``ops.fc_lrelu_fwd`` (bias_init
0.1, leaky-relu 0.2 as in models.py:19-30); gradients use ``fc_bwd_weight`` /
``fc_bwd_data``; the pieces between layers are the ``ew_*`` kernels.  Parameters
live in one flat padded fp32 buffer like VNet's, so the same Adam/LARS launch and
the same all-reduce apply.  ResNetV2 (models.py:205-243) adds a shallow one-layer branch
beside each deep branch; its four cross products and residual sum collapse to
(v12+v21)*(d12+d21) + (v12+v21) + (d12+d21), i.e. ResNet's first residual sum on the branch
SUMS, so it runs on the same kernels.  DenseNet (broken in the reference: tf.shape used as
a static dim, models.py:197-198) is not built.
"""
import math

import numpy as np
import torch

from . import engine_f16x2, engine_x3, ops
from .engine import round_up

NETS = ("MultiplyNet", "MlpNet", "ResNet", "ResNetV2")
VISUAL = 1500                       # models.py:80,107,138: model_input[:, :1500]


class _Layer:
    def __init__(self, name, fan_in, fan_out):
        self.name, self.K, self.N = name, fan_in, fan_out
        self.Kp = round_up(fan_in, 64)
        self.Np = round_up(fan_out, 128 if fan_out >= 1024 else 64)


class FusionParams:
    def __init__(self, net, device, doc_size=128, visual_size=VISUAL, hidden_v=5000, hidden_d=400,
                 output_size=256, mlp_hidden=600, seed=42, bias_init=0.1):
        if net not in NETS:
            raise ValueError("net must be one of %s" % (NETS,))
        self.net, self.device = net, torch.device(device)
        self.visual, self.doc, self.D = visual_size, doc_size, output_size
        shapes = [("layer_visual_1", visual_size, hidden_v), ("layer_visual_2", hidden_v, output_size),
                  ("layer_doc_1", doc_size, hidden_d), ("layer_doc_2", hidden_d, output_size)]
        if net == "ResNetV2":
            shapes = [("layer_visual_1_1", visual_size, hidden_v), ("layer_visual_1_2", hidden_v, output_size),
                      ("layer_visual_2_1", visual_size, output_size), ("layer_doc_1_1", doc_size, hidden_d),
                      ("layer_doc_1_2", hidden_d, output_size), ("layer_doc_2_1", doc_size, output_size),
                      ("layer_fusion_1", output_size, output_size), ("layer_fusion_2", output_size, output_size)]
        elif net == "MlpNet":
            shapes += [("layer_fusion_1", output_size, mlp_hidden), ("layer_fusion_2", mlp_hidden, output_size)]
        elif net == "ResNet":
            shapes += [("layer_fusion_1", output_size, output_size), ("layer_fusion_2", output_size, output_size)]
        self.layers = {n: _Layer(n, fi, fo) for n, fi, fo in shapes}
        # chain consistency: a layer's padded output width is the padded input width of its consumer
        for a, b in (("layer_visual_1", "layer_visual_2"), ("layer_doc_1", "layer_doc_2"),
                     ("layer_fusion_1", "layer_fusion_2"), ("layer_visual_1_1", "layer_visual_1_2"),
                     ("layer_doc_1_1", "layer_doc_1_2")):
            if a in self.layers:
                self.layers[b].Kp = self.layers[a].Np
        v2 = "layer_visual_1_2" if net == "ResNetV2" else "layer_visual_2"
        self.Dp = self.layers[v2].Np
        for n in ("layer_doc_2", "layer_fusion_2", "layer_visual_2_1", "layer_doc_1_2", "layer_doc_2_1"):
            if n in self.layers:
                self.layers[n].Np = self.Dp
        if "layer_fusion_1" in self.layers:
            self.layers["layer_fusion_1"].Kp = self.Dp
            if net in ("ResNet", "ResNetV2"):
                self.layers["layer_fusion_1"].Np = self.Dp
                self.layers["layer_fusion_2"].Kp = self.Dp
        off = 0
        self._seg = {}
        for n, L in self.layers.items():
            self._seg[n] = (off, off + L.Kp * L.Np)
            off += L.Kp * L.Np + L.Np
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        for n, L in self.layers.items():
            W, b = self.W(n), self.b(n)
            lim = math.sqrt(6.0 / (L.K + L.N))                      # slim's Xavier-uniform default
            W[:L.K, :L.N] = (torch.rand((L.K, L.N), device=device, generator=gen) * 2 - 1) * lim
            b[:L.N] = bias_init

    def _view(self, buf, n):
        L = self.layers[n]
        lo, mid = self._seg[n]
        return buf[lo:mid].view(L.Kp, L.Np), buf[mid:mid + L.Np]

    def W(self, n): return self._view(self.flat, n)[0]
    def b(self, n): return self._view(self.flat, n)[1]
    def gW(self, n): return self._view(self.grad, n)[0]
    def gb(self, n): return self._view(self.grad, n)[1]

    def load(self, P):
        self.flat.zero_()
        for n, (W, b) in P.items():
            L = self.layers[n]
            self.W(n)[:L.K, :L.N] = torch.as_tensor(np.asarray(W, np.float32)).to(self.device)
            self.b(n)[:L.N] = torch.as_tensor(np.asarray(b, np.float32)).to(self.device)

    def unpadded(self, grads=False):
        out = {}
        for n, L in self.layers.items():
            W, b = self._view(self.grad if grads else self.flat, n)
            out[n] = (W[:L.K, :L.N], b[:L.N])
        return out


class _VisualAsVNet:
    """The visual branch of a fusion tower IS VNet's two layers (models.py:82-83 against :59-60): this view hands its
    weights, biases and gradients -- slices of the fusion tower's flat buffers -- to engine_x3's forward / backward under
    the names they use."""

    def __init__(self, fp, n1, n2):
        L1, L2 = fp.layers[n1], fp.layers[n2]
        self.layout = engine_x3.layout_x3(L1.K, L1.N, L2.N)
        self.W1, self.b1, self.W2, self.b2 = fp.W(n1), fp.b(n1), fp.W(n2), fp.b(n2)
        self.gW1, self.gb1, self.gW2, self.gb2 = fp.gW(n1), fp.gb(n1), fp.gW(n2), fp.gb(n2)
        self.grad = fp.grad

    @staticmethod
    def fits(fp, n1, n2, n_rows):
        L1, L2 = fp.layers[n1], fp.layers[n2]
        L = engine_x3.layout_x3(L1.K, L1.N, L2.N)
        return (L.Fp, L.Hp, L.Dp) == (L1.Kp, L1.Np, L2.Np) and L2.Kp == L1.Np and n_rows % 128 == 0


class FusionTower:
    """Activations + the forward / backward kernel sequences for R rows.

    ``precision`` "f32x3" / "auto" (round 6): the VISUAL branch -- 1500 -> 5000 -> 256, VNet's two layers and 97 % of the
    tower's flop -- runs on the plane kernels (fp32 operands as three exact bf16 planes, six plane products per fp32
    product on the bf16 MFMA: engine_x3, the headline path's arithmetic and bounds) wherever its padded widths are the
    plane kernels' (multiples of 256) and the rows a multiple of 128; the doc branch and the fusion layers (a few per
    cent of the flop) stay on the fp32 MFMA.  "f32": everything on the fp32 MFMA (rounds 1-5).  "f16x2" (never chosen by
    "auto"): the visual branch on the fp16 build of the plane kernels -- two fp16 planes per operand under delayed per-tensor
    scales, three plane products (engine_f16x2: what it watches and when)."""

    def __init__(self, params, n_rows, precision="auto"):
        if precision not in ("auto", "f32x3", "f32", "f16x2"):
            raise ValueError("precision must be 'auto', 'f32x3', 'f16x2' or 'f32'")
        self.p, self.R = params, int(n_rows)
        p, dev, R = params, params.device, self.R
        z = lambda n: torch.zeros((R, n), dtype=torch.float32, device=dev)
        Ls = p.layers
        first_v = "layer_visual_1_1" if p.net == "ResNetV2" else "layer_visual_1"
        first_d = "layer_doc_1_1" if p.net == "ResNetV2" else "layer_doc_1"
        self.xv, self.xd = z(Ls[first_v].Kp), z(Ls[first_d].Kp)
        self.act = {n: z(L.Np) for n, L in Ls.items()}             # post-activations of every FC
        self.dpre = {n: z(L.Np) for n, L in Ls.items()}            # gradients wrt pre-activations
        D = p.Dp
        self.fu, self.r2, self.pre, self.e, self.de = z(D), z(D), z(D), z(D), z(D)
        self.g0, self.g1, self.g2 = z(D), z(D), z(D)
        if p.net == "ResNetV2":
            self.vs, self.ds, self.dvs, self.dds = z(D), z(D), z(D), z(D)   # branch sums and their gradients
        self.din = {n: z(L.Kp) for n, L in Ls.items() if n in ("layer_fusion_1", "layer_fusion_2")}
        nb = max(ops.fc_bwd_weight_workspace(R, L.Kp, L.Np) for L in Ls.values())
        # the visual branch is VNet's two layers: both of its weight gradients in one stream-K launch when the
        # shapes allow (0 = they do not), as train.TrainStep does on one GPU
        self.sk_bytes = 0
        if "layer_visual_1" in Ls and "layer_visual_2" in Ls:
            v1, v2 = Ls["layer_visual_1"], Ls["layer_visual_2"]
            self.sk_bytes = ops.fc_bwd_weight2_workspace(R, v1.Kp, v1.Np, v2.Kp, v2.Np)
        self.bw = torch.empty(max(nb, self.sk_bytes, 16) // 4, dtype=torch.float32, device=dev)
        # the visual branch on the plane kernels
        self.vx3 = self.vp3 = None
        self.vh2, self._h2_check = False, False
        if precision != "f32" and "layer_visual_1" in Ls:
            if _VisualAsVNet.fits(p, "layer_visual_1", "layer_visual_2", R):
                self.vp3 = _VisualAsVNet(p, "layer_visual_1", "layer_visual_2")
                self.vh2 = precision == "f16x2"
                ws = (engine_f16x2.TowerWorkspaceH2(self.vp3.layout, R, dev, planes_in=False) if self.vh2 else
                      engine_x3.TowerWorkspaceX3(self.vp3.layout, R, dev, planes_in=False, kint=False))
                ws.x_hat = self.xv                                       # the l2-normalised visual rows (fp32: split in the forward pass)
                self.act["layer_visual_2"] = ws.z                        # the branch's output, where the fusion reads it
                self.dpre["layer_visual_2"] = ws.dz2                     # ... and where its gradient arrives
                self.vx3 = ws
                if not self.vh2:
                    engine_x3.refresh_weights(self.vp3, ws)
            elif precision in ("f32x3", "f16x2"):
                raise ValueError("precision '%s' needs the visual branch's padded widths to be multiples of 256 and the "
                                 "rows a multiple of 128" % precision)

    def refresh_planes(self, step=None):
        """After an optimizer step on the flat parameter buffer: the visual branch's weight planes follow the new weights.
        Precision "f16x2": on a check step of its plane scales (``step`` None: always) the weights' and the hidden layer's
        scales are re-derived first, and backward() re-derives the gradients' once the fusion layers have produced them."""
        if self.vx3 is None:
            return
        if self.vh2:
            sc = self.vx3.scales
            self._h2_check = step is None or sc.due(step)
            if self._h2_check:
                engine_f16x2.observe_weights(self.vp3, self.vx3)
            engine_f16x2.refresh_weights(self.vp3, self.vx3)
        else:
            engine_x3.refresh_weights(self.vp3, self.vx3)

    def _fc(self, n, x):
        L = self.p.layers[n]
        return ops.fc_lrelu_fwd(x, self.p.W(n), self.p.b(n), self.act[n], self.R, L.Kp, L.Np)

    def forward(self, x):
        """x: raw features [R, >= visual+doc] (row-major, ld multiple of 4).  Returns e [R, Dp]."""
        p, R, D = self.p, self.R, self.p.Dp
        ops.l2norm_fwd(x[:, :p.visual], p.visual, self.xv)                       # models.py:81
        ops.l2norm_fwd(x[:, p.visual:p.visual + p.doc], p.doc, self.xd)          # models.py:86
        if p.net == "ResNetV2":
            v12 = self._fc("layer_visual_1_2", self._fc("layer_visual_1_1", self.xv))   # models.py:222-223
            v21 = self._fc("layer_visual_2_1", self.xv)                                 # models.py:224
            d12 = self._fc("layer_doc_1_2", self._fc("layer_doc_1_1", self.xd))         # models.py:228-229
            d21 = self._fc("layer_doc_2_1", self.xd)                                    # models.py:230
            ops.ew_combine(ops.EW_ADD, v12, v21, self.vs, R, D)
            ops.ew_combine(ops.EW_ADD, d12, d21, self.ds, R, D)
            ops.ew_combine(ops.EW_MUL_RES, self.vs, self.ds, self.fu, R, D)             # layer_res_1, :232-237
            f1 = self._fc("layer_fusion_1", self.fu)
            ops.ew_combine(ops.EW_ADD, self.fu, f1, self.r2, R, D)                      # :239
            f2 = self._fc("layer_fusion_2", self.r2)
            ops.ew_combine(ops.EW_ADD, self.r2, f2, self.pre, R, D)                     # :241
            ops.l2norm_fwd(self.pre, D, self.e)
            return self.e
        if self.vx3 is not None:
            if self.vh2:
                if not (self.vx3.scales.calibrated or self._h2_check):
                    self.refresh_planes()                                        # a first pass nobody prepared: calibrate
                engine_f16x2.tower_forward(self.vp3, self.vx3, normalize=False)
            else:
                engine_x3.tower_forward(self.vp3, self.vx3, normalize=False)     # h1 as planes + sign bits, v2 = vx3.z (fp32)
            v2 = self.vx3.z
        else:
            v2 = self._fc("layer_visual_2", self._fc("layer_visual_1", self.xv))
        d2 = self._fc("layer_doc_2", self._fc("layer_doc_1", self.xd))
        if p.net == "MultiplyNet":
            ops.ew_combine(ops.EW_MUL, v2, d2, self.pre, R, D)                   # models.py:90
        elif p.net == "MlpNet":
            ops.ew_combine(ops.EW_MUL, v2, d2, self.fu, R, D)                    # models.py:117
            self.pre = self._fc("layer_fusion_2", self._fc("layer_fusion_1", self.fu))   # alias, no copy
        else:
            ops.ew_combine(ops.EW_MUL_RES, v2, d2, self.fu, R, D)                # r1, models.py:148-150
            f1 = self._fc("layer_fusion_1", self.fu)
            ops.ew_combine(ops.EW_ADD, self.fu, f1, self.r2, R, D)               # models.py:152
            f2 = self._fc("layer_fusion_2", self.r2)
            ops.ew_combine(ops.EW_ADD, self.r2, f2, self.pre, R, D)              # models.py:154
        ops.l2norm_fwd(self.pre, D, self.e)
        return self.e

    def _bwd_fc(self, n, x_in, d_pre, d_in=None, mask=None):
        """dW,db of layer n from d_pre; optionally the gradient wrt its input
        (times lrelu' of ``mask``, the post-activation of the producing layer)."""
        p, L = self.p, self.p.layers[n]
        ops.fc_bwd_weight(x_in, d_pre, p.gW(n), p.gb(n), self.bw, self.R, L.Kp, L.Np)
        if d_in is not None:
            ops.fc_bwd_data(d_pre, p.W(n), mask, d_in, self.R, L.Kp, L.Np)
        return d_in

    def backward(self, de=None, joint_visual=True):
        """de (default self.de): gradient wrt e -> p.grad.  ``joint_visual``: the visual branch's two weight
        gradients in one stream-K launch where the shapes allow (False: one split-K launch per layer; a
        data-parallel step that overlaps per-layer all-reduces would ask for that)."""
        p, R, D = self.p, self.R, self.p.Dp
        de = self.de if de is None else de
        A, dp = self.act, self.dpre
        ops.l2norm_bwd(self.pre, de, D, self.g0, lrelu_alpha=-1.0)               # d wrt pre_norm
        if p.net == "MultiplyNet":
            dfu, res = self.g0, False
        elif p.net == "MlpNet":
            ops.lrelu_bwd(self.g0, A["layer_fusion_2"], dp["layer_fusion_2"], R, D)
            d_f1 = self._bwd_fc("layer_fusion_2", A["layer_fusion_1"], dp["layer_fusion_2"],
                                dp["layer_fusion_1"], mask=A["layer_fusion_1"])   # already d_pre of fusion_1
            dfu = self._bwd_fc("layer_fusion_1", self.fu, d_f1, self.din["layer_fusion_1"])
            res = False
        else:                                                                    # ResNet and ResNetV2
            ops.lrelu_bwd(self.g0, A["layer_fusion_2"], dp["layer_fusion_2"], R, D)
            self._bwd_fc("layer_fusion_2", self.r2, dp["layer_fusion_2"], self.din["layer_fusion_2"])
            ops.ew_combine(ops.EW_ADD, self.g0, self.din["layer_fusion_2"], self.g1, R, D)      # d_r2
            ops.lrelu_bwd(self.g1, A["layer_fusion_1"], dp["layer_fusion_1"], R, D)
            self._bwd_fc("layer_fusion_1", self.fu, dp["layer_fusion_1"], self.din["layer_fusion_1"])
            ops.ew_combine(ops.EW_ADD, self.g1, self.din["layer_fusion_1"], self.g2, R, D)      # d_r1
            dfu, res = self.g2, True
        if p.net == "ResNetV2":
            # d wrt the branch sums (no activation derivative: alpha 1), then each branch's own lrelu'
            ops.ew_fusion_bwd(True, dfu, self.vs, self.ds, self.dvs, self.dds, R, D, alpha=1.0)
            for deep, first, shallow, x_in, dsum in (
                    ("layer_visual_1_2", "layer_visual_1_1", "layer_visual_2_1", self.xv, self.dvs),
                    ("layer_doc_1_2", "layer_doc_1_1", "layer_doc_2_1", self.xd, self.dds)):
                ops.lrelu_bwd(dsum, A[deep], dp[deep], R, D)
                self._bwd_fc(deep, A[first], dp[deep], dp[first], mask=A[first])
                self._bwd_fc(first, x_in, dp[first])
                ops.lrelu_bwd(dsum, A[shallow], dp[shallow], R, D)
                self._bwd_fc(shallow, x_in, dp[shallow])
            return p.grad
        ops.ew_fusion_bwd(res, dfu, A["layer_visual_2"], A["layer_doc_2"], dp["layer_visual_2"],
                          dp["layer_doc_2"], R, D)
        if self.vx3 is not None:
            # dp["layer_visual_2"] IS vx3.dz2 (the gradient of the branch's output pre-activation): planes, data gradient,
            # both weight gradients + bias gradients on the plane kernels
            self.vx3.tail_done, self.vx3.dz2_planes_done = True, False
            if self.vh2:
                if self._h2_check:                                               # the gradients' scales, from dz2 as it stands
                    engine_f16x2.observe_gradients(self.vp3, self.vx3)
                    self._h2_check = False
                engine_f16x2.tower_backward(self.vp3, self.vx3)
            else:
                engine_x3.tower_backward(self.vp3, self.vx3)
        elif self.sk_bytes and joint_visual:
            v1, v2 = p.layers["layer_visual_1"], p.layers["layer_visual_2"]
            ops.fc_bwd_data(dp["layer_visual_2"], p.W("layer_visual_2"), A["layer_visual_1"], dp["layer_visual_1"],
                            R, v2.Kp, v2.Np)
            ops.fc_bwd_weight2(self.xv, dp["layer_visual_1"], p.gW("layer_visual_1"), p.gb("layer_visual_1"), v1.Kp, v1.Np,
                               A["layer_visual_1"], dp["layer_visual_2"], p.gW("layer_visual_2"), p.gb("layer_visual_2"),
                               v2.Kp, v2.Np, R, self.bw)
        else:
            self._bwd_fc("layer_visual_2", A["layer_visual_1"], dp["layer_visual_2"], dp["layer_visual_1"],
                         mask=A["layer_visual_1"])
            self._bwd_fc("layer_visual_1", self.xv, dp["layer_visual_1"])
        self._bwd_fc("layer_doc_2", A["layer_doc_1"], dp["layer_doc_2"], dp["layer_doc_1"], mask=A["layer_doc_1"])
        self._bwd_fc("layer_doc_1", self.xd, dp["layer_doc_1"])
        return p.grad


class FusionTrainStep:
    """sample -> raw gather -> fusion tower -> hinge loss -> backward -> Adam, all HIP
    (the reference's step with ``FLAGS.model`` set to a fusion net and
    ``feature_size`` 1628, online_data.py:38)."""

    def __init__(self, net, table, pairs, batch_size, margin=0.8, base_learning_rate=0.01, seed=1234,
                 weight_seed=42, device="cuda:0", exchange=None, grad_sync=None, slot0=0, batch_global=None,
                 precision="auto", **dims):
        """``exchange`` (a dist.RowExchange built with ``local_gather=dist.raw_local_gather``) and
        ``grad_sync``: the data-parallel hooks, as for train.TrainStep -- row-sharded table, this
        rank's slice [slot0, slot0+B) of the global batch, averaged gradients."""
        self.device = torch.device(device)
        self.exchange, self.grad_sync = exchange, grad_sync
        self.slot0 = int(slot0)
        self.batch_global = int(batch_size) if batch_global is None else int(batch_global)
        self.table, self.pairs, self.B, self.margin, self.seed = table, pairs, int(batch_size), margin, seed
        doc = table.feature_size - dims.get("visual_size", VISUAL)
        self.params = FusionParams(net, device, doc_size=doc, seed=weight_seed, **dims)
        self.tower = FusionTower(self.params, 3 * self.B, precision=precision)
        self.precision = ("f32" if self.tower.vx3 is None else
                          "f16x2 (visual branch) + f32" if self.tower.vh2 else "f32x3 (visual branch) + f32")
        dev, f32 = self.device, torch.float32
        self.idx = torch.zeros((self.B, 3), dtype=torch.int32, device=dev)
        self.x = torch.zeros((3 * self.B, table.data.shape[1]), dtype=f32, device=dev)
        self.pos, self.neg, self.hinge = (torch.zeros(self.B, dtype=f32, device=dev) for _ in range(3))
        self.stats = torch.zeros(4, dtype=f32, device=dev)
        self.m, self.v = torch.zeros_like(self.params.flat), torch.zeros_like(self.params.flat)
        self.lr, self.global_step = base_learning_rate, 0

    def step(self):
        t = self.tower
        ops.sample_uniform(self.pairs, self.table.n_rows_global, self.seed, self.global_step, self.B, self.idx,
                           slot0=self.slot0, batch_global=self.batch_global)
        if self.exchange is None:
            ops.gather_rows(self.table.data, self.table.row0, self.idx.view(-1), self.table.feature_size, self.x,
                            normalize=False)
        else:
            self.exchange.gather(self.table, self.idx.view(-1), self.x)
        t.refresh_planes(self.global_step)      # (the visual branch's weight planes follow whatever the weights are now: Adam's update, a load())
        t.forward(self.x)
        ops.triplet_hinge(t.e, self.B, self.params.Dp, self.margin, self.pos, self.neg, self.hinge, self.stats, t.de)
        t.backward()
        if self.grad_sync is not None:
            self.grad_sync(self.params.grad)
        self.global_step += 1
        ops.adam_step(self.params.flat, self.params.grad, self.m, self.v, self.lr, self.global_step)

    def loss(self):
        return float(self.stats[0].item())
