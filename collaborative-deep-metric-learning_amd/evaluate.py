"""In-loop evaluation metric behind the reference's API (evaluate.py).

``Evaluation(features, cowatches)`` re-indexes the catalogue to the rows the
held-out co-watch pairs touch (evaluate.py:34-55); ``mean_dist(vectors,
cowatches)`` is the mean squared L2 distance between the embeddings of each
pair (evaluate.py:57-73) -- the model-selection signal of train.py:224-252.
The per-pair reduction runs as a HIP kernel; the re-indexing is host
bookkeeping on the (small) pair list.
"""
import numpy as np
import torch

from . import ops


class Evaluation():
    def __init__(self, features, cowatches, device="cuda:0"):
        """features: ndarray [N,F] (or None); cowatches: list/array of [a,p] row ids."""
        self.device = torch.device(device)
        try:
            self.features, self.cowatches = self._rencode(features, cowatches)
        except Exception:                      # the reference logs and sets None (evaluate.py:30-32)
            self.features, self.cowatches = None, None

    def _rencode(self, features, cowatches):
        cw = np.asarray(cowatches, dtype=np.int64).reshape(-1, 2)
        sorted_indexes = np.unique(cw)          # == np.sort(get_unique_watched_guids(cowatches))
        eval_features = features[sorted_indexes]
        eval_cowatches = np.searchsorted(sorted_indexes, cw)   # old row id -> position in the subset
        return eval_features, eval_cowatches.tolist()

    def _pair_stats(self, vectors, cowatches):
        v = vectors if torch.is_tensor(vectors) else torch.as_tensor(np.asarray(vectors, np.float32))
        v = v.to(self.device, torch.float32).contiguous()
        D = v.shape[1]
        if D % 4:                                # kernels take 16-B rows: pad with zero columns
            v = torch.nn.functional.pad(v, (0, 4 - D % 4))
        pairs = torch.as_tensor(np.asarray(cowatches, np.int32).reshape(-1, 2)).to(self.device)
        if int(pairs.max()) >= v.shape[0] or int(pairs.min()) < 0:
            raise IndexError("co-watch index outside the embedding table")
        P = pairs.shape[0]
        sq, dot = torch.empty(P, device=self.device), torch.empty(P, device=self.device)
        means = torch.empty(4, device=self.device)
        ops.pair_dist(v, pairs, v.shape[1], sq, dot, means)
        return means

    def mean_dist(self, vectors, cowatches):
        """Mean over pairs of sum((a-b)^2) (evaluate.py:57-73)."""
        return float(self._pair_stats(vectors, cowatches)[1].item())

    def mean_cos_dist(self, vectors, cowatches):
        """Mean over pairs of sum(a*b) (evaluate.py:75-90)."""
        return float(self._pair_stats(vectors, cowatches)[2].item())
