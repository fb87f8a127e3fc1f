"""Embedding towers behind the reference's model plug-in API (models.py).

``find_class_by_name("VNet", [models])()`` then
``model.create_model(model_input, output_size)`` -> {"layer_1","layer_2","l2_norm"}
exactly as train.py:105,127 consumes it.  Forward and backward run the HIP
kernels through the C ABI; autograd only routes the gradient.
"""
import torch

from . import engine, ops


class BaseModel(object):
    """Inherit from this class when implementing new models (models.py:33-38)."""

    def create_model(self, unused_model_input, **unused_params):
        raise NotImplementedError()


class _VNetFunction(torch.autograd.Function):
    """layer_1/layer_2 are returned for inspection only (the reference consumes
    just "l2_norm", train.py:127); they are marked non-differentiable."""

    @staticmethod
    def forward(ctx, model_input, flat, model):
        p, ws = model.params, model._workspace(model_input.shape[0])
        L = p.layout
        R = model_input.shape[0]
        x = model_input.contiguous()
        ops.l2norm_fwd(x, L.F, ws.x_hat)                 # models.py:58
        engine.tower_forward(p, ws, R)                   # models.py:59-61
        ctx.model, ctx.R = model, R
        l1, l2, out = ws.h1[:R, :L.H], ws.z[:R, :L.D], ws.e[:R, :L.D]
        ctx.mark_non_differentiable(l1, l2)
        return l1, l2, out.clone()

    @staticmethod
    def backward(ctx, _g1, _g2, g_out):
        model, R = ctx.model, ctx.R
        p, ws = model.params, model._ws
        L = p.layout
        ws.de[:R].zero_()
        ws.de[:R, :L.D] = g_out
        engine.tower_backward(p, ws, R)
        return None, p.grad, None


class VNet(BaseModel):
    """Visual Feature Network (models.py:41-62): l2norm -> FC 5000 leaky_relu ->
    FC output_size leaky_relu -> l2norm.  The output layer is activated too
    (models.py:60 uses fully_connected's default activation_fn)."""

    hidden_size = 5000          # hard-coded in the reference (models.py:59)

    def __init__(self, device="cuda:0", seed=42):
        self.device = torch.device(device)
        self.seed = seed
        self.params = None
        self._ws = None

    def build(self, feature_size, output_size=256):
        layout = engine.TowerLayout(feature_size, self.hidden_size, output_size)
        self.params = engine.VNetParams(layout, self.device, self.seed, bias_init=0.0)
        self.params.flat.requires_grad_(True)
        return self

    def _workspace(self, n_rows):
        if self._ws is None or self._ws.R < n_rows:
            self._ws = engine.TowerWorkspace(self.params.layout, n_rows, self.device)
        return self._ws

    def create_model(self, model_input, output_size=256):
        """model_input: float32 [batch*3, feature_size] device tensor (rows a,p,n per
        triplet).  Returns {"layer_1","layer_2","l2_norm"}."""
        if self.params is None:
            self.build(model_input.shape[1], output_size)
        L = self.params.layout
        if model_input.shape[1] != L.F or output_size != L.D:
            raise ValueError("model was built for feature_size=%d output_size=%d" % (L.F, L.D))
        l1, l2, out = _VNetFunction.apply(model_input, self.params.flat, self)
        return {"layer_1": l1, "layer_2": l2, "l2_norm": out}

    def variables(self):
        return self.params.state_dict()


# ---------------------------------------------------------------- fusion towers --
class _FusionFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model_input, flat, model):
        tower = model._tower(model_input.shape[0])
        x = model_input.contiguous()
        e = tower.forward(x)
        ctx.tower = tower
        return e[:, :model.params.D].clone()

    @staticmethod
    def backward(ctx, g_out):
        tower = ctx.tower
        tower.de.zero_()
        tower.de[:, :tower.p.D] = g_out
        tower.backward()
        return None, tower.p.grad, None


class _FusionNet(BaseModel):
    """Shared body of the fusion towers: input = visual (first 1500 columns) ++ doc
    features (feature_size 1628 in production, online_data.py:38)."""

    net = None
    visual_size = 1500

    def __init__(self, device="cuda:0", seed=42, **dims):
        self.device, self.seed, self.dims = torch.device(device), seed, dims
        self.params = None
        self._t = None

    def build(self, feature_size, output_size=256):
        from . import fusion
        visual = self.dims.get("visual_size", self.visual_size)
        dims = dict(self.dims, visual_size=visual, doc_size=feature_size - visual, output_size=output_size)
        self.params = fusion.FusionParams(self.net, self.device, seed=self.seed, **dims)
        self.params.flat.requires_grad_(True)
        return self

    def _tower(self, n_rows):
        from . import fusion
        if self._t is None or self._t.R != n_rows:
            # (the facade's weights are the caller's to change between calls: the fp32-MFMA tower reads them as they are;
            # fusion.FusionTrainStep, which owns its update, runs the visual branch on the plane kernels)
            self._t = fusion.FusionTower(self.params, n_rows, precision="f32")
        return self._t

    def create_model(self, model_input, output_size=256):
        """model_input float32 [batch*3, feature_size] -> {"l2_norm": [batch*3, output_size]}."""
        if self.params is None:
            self.build(model_input.shape[1], output_size)
        if model_input.shape[1] != self.params.visual + self.params.doc or output_size != self.params.D:
            raise ValueError("model was built for another feature / output size")
        return {"l2_norm": _FusionFunction.apply(model_input, self.params.flat, self)}


class MultiplyNet(_FusionNet):
    """Fusion by multiply (models.py:65-92)."""
    net = "MultiplyNet"


class MlpNet(_FusionNet):
    """Multiply fusion followed by an MLP 600 -> 256 (models.py:93-122)."""
    net = "MlpNet"


class ResNet(_FusionNet):
    """Multiply fusion with residual connections -- the tower the author reports to
    work best (models.py:125-157)."""
    net = "ResNet"


class ResNetV2(_FusionNet):
    """The wider ResNet: a deep and a shallow branch per modality, four cross products,
    the same residual head (models.py:205-243)."""
    net = "ResNetV2"
