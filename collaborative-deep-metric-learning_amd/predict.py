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

from . import engine, ops


class Prediction():
    def __init__(self, params=None, ckpt=None, device="cuda:0"):
        if params is None:
            if ckpt is None or not os.path.exists(ckpt):
                raise IOError("Prediction __init__ Cannot find %s" % ckpt)      # predict.py:54-55
            state = torch.load(ckpt, map_location="cpu")
            layout = engine.TowerLayout(*state["layout"])
            params = engine.VNetParams(layout, device)
            params.load(*[state["variables"][n] for n in engine.VNetParams.NAMES])
        self.params = params
        self.device = params.device
        self._ws = None

    def _workspace(self, n_rows):
        if self._ws is None or self._ws.R < n_rows:
            self._ws = engine.TowerWorkspace(self.params.layout, n_rows, self.device, backward=False)
        return self._ws

    def predict(self, input_batch):
        """input_batch: [n,F] ndarray / device tensor (raw features) or rows of a
        FeatureTable's padded storage.  Returns a device tensor [n,D]."""
        L = self.params.layout
        x = input_batch if torch.is_tensor(input_batch) else torch.as_tensor(np.asarray(input_batch, np.float32))
        x = x.to(self.device, torch.float32)
        if x.stride(-1) != 1 or (x.stride(0) % 4) or (x.data_ptr() % 16):
            x = x.contiguous()
        n = x.shape[0]
        ws = self._workspace(n)
        if L.F % 4:
            raise ValueError("feature size must be a multiple of 4")
        ops.l2norm_fwd(x[:, :L.F] if x.shape[1] != L.F else x, L.F, ws.x_hat)      # models.py:58
        engine.tower_forward(self.params, ws, n)                                   # models.py:59-61
        return ws.e[:n, :L.D]

    def run_features(self, features, batch_size, output_dir='', suffix=''):
        """Embeddings of every row of ``features`` (ndarray, tensor or FeatureTable),
        ``batch_size`` rows at a time; float32 ndarray [N,D] like predict.py:71-96
        (saved to output_dir/output<suffix>.npy when output_dir is given)."""
        if isinstance(features, engine.FeatureTable):
            features = features.data[:, :features.feature_size] if features.data.shape[1] == features.feature_size \
                else features.data
        N = features.shape[0]
        out = torch.empty((N, self.params.layout.D), dtype=torch.float32, device=self.device)
        for lo in range(0, N, batch_size):
            hi = min(lo + batch_size, N)
            out[lo:hi] = self.predict(features[lo:hi])
        output_np = out.cpu().numpy()
        if output_dir:
            np.save(os.path.join(output_dir, "output" + suffix + ".npy"), output_np)
        return output_np
