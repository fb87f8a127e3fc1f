"""Device-resident state of the hot path and the kernel call sequences.

  FeatureTable   the catalogue in HBM (reference: module-global FEATURES ndarray,
                 inputs.py:19,73-74), rows padded to a 128-B multiple
  VNetParams     W1,b1,W2,b2 (+ gradients, optimizer slots) in ONE flat padded
                 buffer each, so the optimizer is one launch and the data-parallel
                 gradient exchange one all-reduce
  TowerWorkspace caller-owned activations / workspaces for a fixed row count
  tower_forward / tower_backward   the kernel sequence of VNet.create_model
                 (models.py:46-62) and of its autodiff (train.py:141)

Padding: F -> Fp (x64), H -> Hp (x128), D -> Dp (x64).  Padded weights, biases
and table columns are zero, every kernel keeps them zero (zero activations ->
zero gradients -> zero Adam/LARS update), so the padded problem computes the
reference's numbers exactly.
"""
import math
import os

import numpy as np
import torch

from . import ops


def round_up(x, m):
    return (x + m - 1) // m * m


class FeatureTable:
    """Catalogue shard [n_rows, row_stride] fp32 in HBM holding global rows
    [row0, row0+n_rows).  row_stride*4 is a multiple of 128 B so every row starts
    on a cache line and doubles as the GEMM's K padding."""

    def __init__(self, data, feature_size, row0=0, n_rows_global=None):
        if not data.is_cuda or data.dtype != torch.float32 or data.dim() != 2:
            raise ValueError("FeatureTable needs a 2-D fp32 device tensor")
        self.data = data
        self.feature_size = int(feature_size)
        self.row0 = int(row0)
        self.n_rows = data.shape[0]
        self.n_rows_global = self.n_rows if n_rows_global is None else int(n_rows_global)

    @staticmethod
    def padded_stride(feature_size):
        return round_up(feature_size, 64)

    @classmethod
    def from_numpy(cls, features, device, row0=0, n_rows_global=None):
        """features.npy contents (float32 [N,F], online_data.py:87-93) -> HBM."""
        f = np.asarray(features, dtype=np.float32)
        n, F = f.shape
        data = torch.zeros((n, cls.padded_stride(F)), dtype=torch.float32, device=device)
        data[:, :F] = torch.from_numpy(f).to(device)
        return cls(data, F, row0, n_rows_global)

    @classmethod
    def synthetic(cls, n_rows, feature_size, seed, device, row0=0, n_rows_global=None):
        """imitation_data.py-shaped U[0,1) table generated in HBM by the HIP fill
        kernel (a 6-60 GB table is not generated on the host)."""
        data = torch.empty((n_rows, cls.padded_stride(feature_size)), dtype=torch.float32,
                           device=device)
        ops.fill_uniform_table(data, row0, feature_size, seed)
        return cls(data, feature_size, row0, n_rows_global)

    def rows(self, idx):
        """Unpadded copy of the given LOCAL rows (debug / tests)."""
        return self.data[idx, :self.feature_size]


class TowerLayout:
    def __init__(self, feature_size, hidden=5000, output_size=256):
        self.F, self.H, self.D = int(feature_size), int(hidden), int(output_size)
        self.Fp = round_up(self.F, 64)
        self.Hp = round_up(self.H, 128)
        self.Dp = round_up(self.D, 64)
        self.sizes = (self.Fp * self.Hp, self.Hp, self.Hp * self.Dp, self.Dp)
        self.offsets = tuple(int(x) for x in np.cumsum((0,) + self.sizes[:-1]))
        self.numel = int(sum(self.sizes))
        self.numel_unpadded = self.F * self.H + self.H + self.H * self.D + self.D


class VNetParams:
    """Parameters of VNet (models.py:59-60: two slim.fully_connected layers,
    weights [in,out], slim variable names fully_connected{,_1}/{weights,biases})."""

    NAMES = ("fully_connected/weights", "fully_connected/biases",
             "fully_connected_1/weights", "fully_connected_1/biases")

    def __init__(self, layout, device, seed=42, bias_init=0.0):
        self.layout = layout
        self.device = torch.device(device)
        L = layout
        self.flat = torch.zeros(L.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.flat)
        self.W1, self.b1, self.W2, self.b2 = self._views(self.flat)
        self.gW1, self.gb1, self.gW2, self.gb2 = self._views(self.grad)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        # slim default initializer: Xavier uniform, +-sqrt(6/(fan_in+fan_out))
        for W, fi, fo in ((self.W1, L.F, L.H), (self.W2, L.H, L.D)):
            lim = math.sqrt(6.0 / (fi + fo))
            W[:fi, :fo] = (torch.rand((fi, fo), device=device, generator=gen) * 2 - 1) * lim
        self.b1[:L.H] = bias_init
        self.b2[:L.D] = bias_init

    def _views(self, flat):
        L = self.layout
        o = L.offsets
        return (flat[o[0]:o[0] + L.sizes[0]].view(L.Fp, L.Hp), flat[o[1]:o[1] + L.sizes[1]],
                flat[o[2]:o[2] + L.sizes[2]].view(L.Hp, L.Dp), flat[o[3]:o[3] + L.sizes[3]])

    def segments(self):
        """(offset, numel) of each variable inside the flat buffers."""
        return list(zip(self.layout.offsets, self.layout.sizes))

    def load(self, W1, b1, W2, b2):
        L = self.layout
        self.flat.zero_()
        for dst, src in ((self.W1[:L.F, :L.H], W1), (self.b1[:L.H], b1),
                         (self.W2[:L.H, :L.D], W2), (self.b2[:L.D], b2)):
            dst.copy_(torch.as_tensor(np.asarray(src, dtype=np.float32)).to(self.device))

    def unpadded(self, grads=False):
        L = self.layout
        t = (self.gW1, self.gb1, self.gW2, self.gb2) if grads else (self.W1, self.b1, self.W2, self.b2)
        return (t[0][:L.F, :L.H], t[1][:L.H], t[2][:L.H, :L.D], t[3][:L.D])

    def state_dict(self):
        return {n: t.detach().cpu().clone() for n, t in zip(self.NAMES, self.unpadded())}


class TowerWorkspace:
    """Activations and scratch for R rows.  Allocated once by the caller; the
    step path allocates nothing (hipGraph-capturable)."""

    def __init__(self, layout, n_rows, device, backward=True):
        L = layout
        self.layout, self.R = layout, int(n_rows)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.x_hat = z(n_rows, L.Fp)
        self.h1 = z(n_rows, L.Hp)
        self.z = z(n_rows, L.Dp)
        self.e = z(n_rows, L.Dp)
        if backward:
            self.de = z(n_rows, L.Dp)
            self.dz2 = z(n_rows, L.Dp)
            self.dz1 = z(n_rows, L.Hp)
            # both weight gradients in one stream-K launch when the shapes allow (0 = they do not)
            self.sk_bytes = ops.fc_bwd_weight2_workspace(n_rows, L.Fp, L.Hp, L.Hp, L.Dp)
            nbytes = max(self.sk_bytes, ops.fc_bwd_weight_workspace(n_rows, L.Hp, L.Dp),
                         ops.fc_bwd_weight_workspace(n_rows, L.Fp, L.Hp),
                         # dW1 in two row blocks (data-parallel runs, tower_backward w1_chunks=2)
                         ops.fc_bwd_weight_workspace(n_rows, L.Fp // 2, L.Hp) if L.Fp % 256 == 0 else 0)
            self.bw = torch.empty(nbytes // 4, dtype=torch.float32, device=device)


def tower_forward(p, ws, n_rows=None, normalize=True):
    """x_hat (already l2-normalised, models.py:58) -> h1 -> z -> e.
    models.py:59-61.  ``normalize=False`` stops at z: the training step's fused tail
    (ops.vnet_tail) normalises, takes the loss and starts the backward pass in one launch.

    The contractions run over round_up(F, 32) and round_up(H, 32), not over the padded leading
    dimensions (columns F.. of x_hat and H.. of h1 are zero): 47 instead of 48 K-tiles for
    F = 1500, 157 instead of 160 for H = 5000."""
    L = p.layout
    R = ws.R if n_rows is None else n_rows
    ops.fc_lrelu_fwd(ws.x_hat, p.W1, p.b1, ws.h1, R, round_up(L.F, 32), L.Hp)
    ops.fc_lrelu_fwd(ws.h1, p.W2, p.b2, ws.z, R, round_up(L.H, 32), L.Dp)
    ws.tail_done = False
    if normalize:
        ops.l2norm_fwd(ws.z[:R], L.Dp, ws.e)
    return ws.e


def tower_backward(p, ws, n_rows=None, after_w1=None, w1_chunks=1, after_w1_chunk=None):
    """ws.de (grad wrt e) -> p.grad (dW1, db1, dW2, db2).  No dX: the features are
    inputs, not variables (train.py:265).  The first layer's gradient (85 % of the
    bytes) is produced BEFORE the second layer's so that ``after_w1`` -- the
    data-parallel all-reduce of [dW1|db1] -- runs under the dW2 GEMM.

    ``w1_chunks`` > 1 with ``after_w1_chunk(lo, hi)``: dW1 is produced in row blocks of W1
    (contiguous ranges [lo, hi) of the flat gradient; the last one ends after db1) and the
    callback fires after each, so the all-reduce of block c runs under the GEMM of block
    c+1 and only the last, smaller one is left for the dW2 GEMM to cover."""
    L = p.layout
    R = ws.R if n_rows is None else n_rows
    if not getattr(ws, "tail_done", False):          # the fused tail has already produced dz2
        ops.l2norm_bwd(ws.z[:R], ws.de[:R], L.Dp, ws.dz2, lrelu_alpha=ops.LRELU_ALPHA)
    ops.fc_bwd_data(ws.dz2, p.W2, ws.h1, ws.dz1, R, L.Hp, L.Dp)
    rows = L.Fp // w1_chunks if w1_chunks > 1 else 0
    if (after_w1 is None and after_w1_chunk is None and getattr(ws, "sk_bytes", 0) and R == ws.R
            and not os.environ.get("CDML_NO_STREAMK")):
        # single GPU: nothing waits for dW1 alone, so both products share one stream-K launch
        ops.fc_bwd_weight2(ws.x_hat, ws.dz1, p.gW1, p.gb1, L.Fp, L.Hp, ws.h1, ws.dz2, p.gW2, p.gb2, L.Hp, L.Dp,
                           R, ws.bw)
        return p.grad
    if after_w1_chunk is not None and w1_chunks > 1 and rows % 128 == 0 and rows * w1_chunks == L.Fp:
        for c in range(w1_chunks):
            lo, hi = c * rows, (c + 1) * rows
            last = c == w1_chunks - 1
            ops.fc_bwd_weight(ws.x_hat[:, lo:hi], ws.dz1, p.gW1[lo:hi], p.gb1 if last else None, ws.bw,
                              R, rows, L.Hp)
            after_w1_chunk(lo * L.Hp, hi * L.Hp + (L.Hp if last else 0))
    else:
        ops.fc_bwd_weight(ws.x_hat, ws.dz1, p.gW1, p.gb1, ws.bw, R, L.Fp, L.Hp)
        if after_w1_chunk is not None:
            after_w1_chunk(0, L.Fp * L.Hp + L.Hp)
    if after_w1 is not None:
        after_w1()
    ops.fc_bwd_weight(ws.h1, ws.dz2, p.gW2, p.gb2, ws.bw, R, L.Hp, L.Dp)
    return p.grad
