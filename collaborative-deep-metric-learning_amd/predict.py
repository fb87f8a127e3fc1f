"""Batch inference behind the reference's API (predict.py:45-96): run the tower
forward over an [N,F] feature array in chunks and return float32 [N,D].

The reference restores a TF checkpoint or reuses the live session
(``Prediction(sess=sess)``, train.py:282); here ``Prediction`` wraps the live
``VNetParams`` (or a checkpoint written by ``train.Trainer``).  Forward only:
l2norm -> FC -> FC -> l2norm, the same HIP kernels as training.
"""
import os

import numpy as np
import torch

from . import engine, engine_bf16, engine_f16x2, engine_x3, ops


class Prediction():
    def __init__(self, params=None, ckpt=None, device="cuda:0", precision="f32", fc2_single_pass=False):
        """``precision``: "f32" (the reference's arithmetic) or "bf16" -- BASELINE config 4's
        precision for catalogue inference: an fp16 ``FeatureTableF16`` in, bf16 MFMA projection,
        fp32 accumulation and output normalisation (build-defined; tolerance 5e-3 on the unit-norm
        embeddings, tests/test_gpu_bf16.py).  ``fc2_single_pass`` (precision "f32x3"): see
        engine_x3.TowerWorkspaceX3 -- off, an embedding's bits do not depend on the chunk size."""
        if params is None:
            if ckpt is None or not os.path.exists(ckpt):
                raise IOError("Prediction __init__ Cannot find %s" % ckpt)      # predict.py:54-55
            state = torch.load(ckpt, map_location="cpu")
            mk = (engine_bf16.layout_bf16 if precision == "bf16" else engine_x3.layout_x3 if precision in ("f32x3", "f16x2")
                  else engine.TowerLayout)
            layout = mk(*state["layout"])
            params = engine.VNetParams(layout, device)
            params.load(*[state["variables"][n] for n in engine.VNetParams.NAMES])
        if precision not in ("f32", "bf16", "f32x3", "f16x2"):
            raise ValueError("precision must be 'f32', 'f32x3', 'f16x2' or 'bf16'")
        if precision in ("f32x3", "f16x2"):
            # ("f16x2": the same tower from two fp16 planes per operand, three plane products on the fp16 MFMA -- engine_f16x2;
            # the weights' and the hidden layer's scales are derived from the weights once per pass)
            # the fp32 tower on the bf16 MFMA (engine_x3: three exact bf16 planes per operand, six plane products)
            L = params.layout
            if L.Fp % 256 or L.Hp % 256 or L.Dp % 256:
                raise ValueError("precision 'f32x3' needs parameters built on engine_x3.layout_x3 (widths padded to 256)")
        self.params = params
        self.device = params.device
        self.precision = precision
        self.fc2_single_pass = bool(fc2_single_pass)
        self._ws = None

    X3_MAX_ROWS = 65536

    def _workspace(self, n_rows):
        if self.precision == "bf16":
            n_rows = engine.round_up(n_rows, 64)
        if self.precision in ("f32x3", "f16x2"):
            n_rows = engine.round_up(n_rows, 128)
        if self._ws is None or self._ws.R < n_rows:
            if self.precision == "f16x2":
                self._ws = engine_f16x2.TowerWorkspaceH2(self.params.layout, n_rows, self.device, planes_in=False, backward=False)
            elif self.precision == "f32x3":
                self._ws = engine_x3.TowerWorkspaceX3(self.params.layout, n_rows, self.device, planes_in=False, backward=False,
                                                      fc2_single_pass=self.fc2_single_pass)
            elif self.precision == "bf16":
                self._ws = engine_bf16.TowerWorkspaceBF16(self.params.layout, n_rows, self.device, backward=False)
                self._ids = torch.arange(n_rows, dtype=torch.int32, device=self.device)
                self._idx = torch.zeros(n_rows, dtype=torch.int32, device=self.device)
            else:
                self._ws = engine.TowerWorkspace(self.params.layout, n_rows, self.device, backward=False)
        return self._ws

    def predict(self, input_batch):
        """input_batch: [n,F] ndarray / device tensor (raw features) or rows of a
        FeatureTable's padded storage.  Returns a device tensor [n,D]."""
        L = self.params.layout
        if self.precision == "bf16":
            raise ValueError("bf16 inference reads an fp16 catalogue: use embed_table / run_features(FeatureTableF16)")
        x = input_batch if torch.is_tensor(input_batch) else torch.as_tensor(np.asarray(input_batch, np.float32))
        x = x.to(self.device, torch.float32)
        if x.stride(-1) != 1 or (x.stride(0) % 4) or (x.data_ptr() % 16):
            x = x.contiguous()
        n = x.shape[0]
        if self.precision in ("f32x3", "f16x2") and n > self.X3_MAX_ROWS:
            return torch.cat([self.predict(x[lo:lo + self.X3_MAX_ROWS]).clone() for lo in range(0, n, self.X3_MAX_ROWS)])
        ws = self._workspace(n)
        if L.F % 4:
            raise ValueError("feature size must be a multiple of 4")
        ops.l2norm_fwd(x[:, :L.F] if x.shape[1] != L.F else x, L.F, ws.x_hat)      # models.py:58
        if self.precision == "f16x2":
            if not getattr(self, "_planes_fresh", False):
                engine_f16x2.observe_weights(self.params, ws)                      # scales from the weights, then their planes
            engine_f16x2.tower_forward(self.params, ws)
        elif self.precision == "f32x3":
            if not getattr(self, "_planes_fresh", False):
                engine_x3.refresh_weights(self.params, ws)                         # (embed_table: once per pass)
            engine_x3.tower_forward(self.params, ws)
        else:
            engine.tower_forward(self.params, ws, n)                               # models.py:59-61
        return ws.e[:n, :L.D]

    def embed_table(self, table, batch_size, out=None):
        """Embeddings of every row of a device-resident catalogue, ``batch_size`` rows at a time,
        into a DEVICE tensor [N, D] (predict.py:71-96 without its host round trip: the reference
        converts every chunk with ``.tolist()``, predict.py:79-86).  Enqueue-only."""
        L, N = self.params.layout, table.n_rows
        if out is None:
            out = torch.empty((N, L.D), dtype=torch.float32, device=self.device)
        if self.precision == "bf16":
            if table.data.dtype != torch.float16:
                raise ValueError("precision 'bf16' reads an fp16 FeatureTableF16")
            ws = self._workspace(min(batch_size, N))
            engine_bf16.refresh_weights(self.params, ws)          # bf16 operand copies of the current weights
            for lo in range(0, N, batch_size):
                n = min(batch_size, N - lo)
                torch.add(self._ids[:n], lo, out=self._idx[:n])
                ops.gather_rows_f16(table.data, 0, self._idx[:n], table.feature_size, ws.x_hat)   # rows lo..lo+n, l2-normalised
                engine_bf16.tower_forward(self.params, ws)
                out[lo:lo + n] = ws.e[:n, :L.D]
            return out
        feats = table.data[:, :table.feature_size] if table.data.shape[1] == table.feature_size else table.data
        if self.precision in ("f32x3", "f16x2"):
            # h1 as three planes is 30 KB per row: the GEMMs' 2 GiB buffer descriptors take 65 536 rows at a time
            batch_size = min(batch_size, self.X3_MAX_ROWS)
            if self.precision == "f16x2":
                engine_f16x2.observe_weights(self.params, self._workspace(min(batch_size, N)))
            else:
                engine_x3.refresh_weights(self.params, self._workspace(min(batch_size, N)))
            self._planes_fresh = True
        try:
            for lo in range(0, N, batch_size):
                hi = min(lo + batch_size, N)
                out[lo:hi] = self.predict(feats[lo:hi])
        finally:
            self._planes_fresh = False
        return out

    def run_features(self, features, batch_size, output_dir='', suffix=''):
        """Embeddings of every row of ``features`` (ndarray, tensor or FeatureTable),
        ``batch_size`` rows at a time; float32 ndarray [N,D] like predict.py:71-96
        (saved to output_dir/output<suffix>.npy when output_dir is given)."""
        if isinstance(features, engine.FeatureTable):
            out = self.embed_table(features, batch_size)
        else:
            N = features.shape[0]
            out = torch.empty((N, self.params.layout.D), dtype=torch.float32, device=self.device)
            for lo in range(0, N, batch_size):
                hi = min(lo + batch_size, N)
                out[lo:hi] = self.predict(features[lo:hi])
        output_np = out.cpu().numpy()
        if output_dir:
            np.save(os.path.join(output_dir, "output" + suffix + ".npy"), output_np)
        return output_np
