"""Co-watch mining of the reference's ETL on the device (parse_data.py:181-289).

The reference walks Python lists and a dict keyed by ``"a,b"`` strings; here the pair list is an
int32 [P,2] device tensor and the graph statistics come from one sort + run-length encode
(``csrc/cowatch.hip``).  Results are the reference's, as sets / sequences of integer pairs:

  get_all_cowatch(all_watched_guids)   consecutive pairs of every watch history, shuffled once
                                       (parse_data.py:181-209; host, numpy)
  cowatch_graph(pairs)                 distinct undirected edges + multiplicities
                                       (get_cowatch_graph, :221-254; raises on a self pair)
  select_cowatch(pairs, threshold, unique=False)
                                       pairs whose edge was seen >= threshold times (:256-289);
                                       unique=True: each such edge once -- as (min, max) in
                                       ascending order where the reference shuffles
"""
import numpy as np
import torch

from . import ops


def get_all_cowatch(all_watched_guids, seed=None):
    """int32 [P,2]: (w[i], w[i+1]) for every history w, shuffled (np.random, or ``seed``)."""
    parts = [np.stack([np.asarray(w[:-1]), np.asarray(w[1:])], 1) for w in all_watched_guids if len(w) > 1]
    if not parts:
        return np.zeros((0, 2), dtype=np.int32)
    cow = np.concatenate(parts).astype(np.int32)
    (np.random if seed is None else np.random.RandomState(seed)).shuffle(cow)
    return cow


def lookup(batch_triplets, features):
    """Row-id triplets [b,3] -> feature triplets [b,3,F] (parse_data.py:353-376; the reference
    walks a guid -> ndarray dict, here ``features`` is an engine.FeatureTable or a [N,F] device
    tensor and the rows are gathered by the HIP kernel).  Ids outside the table raise -- the
    reference logs and silently drops the triplet."""
    from .engine import FeatureTable
    table = features if isinstance(features, FeatureTable) else FeatureTable(features, features.shape[1])
    idx = batch_triplets.reshape(-1).to(device=table.data.device, dtype=torch.int32)
    out = torch.empty((idx.numel(), table.data.shape[1]), dtype=torch.float32, device=table.data.device)
    oob = torch.zeros(1, dtype=torch.int32, device=table.data.device)
    ops.gather_rows(table.data, table.row0, idx, table.feature_size, out, normalize=False, oob_flag=oob)
    if int(oob.item()):
        raise IndexError("lookup: a row id is outside the feature table")
    return out[:, :table.feature_size].reshape(batch_triplets.shape[0], 3, table.feature_size)


def _device_pairs(pairs, device):
    t = pairs if torch.is_tensor(pairs) else torch.as_tensor(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
    return t.to(device=device, dtype=torch.int32).contiguous()


def _check_self_pairs(flag):
    if int(flag.item()):
        # parse_data.py:244-246: "cowatch 存在相邻重复元素，结果不合规"
        raise RuntimeError("get_cowatch_graph: a co-watch pair repeats one video (a == p); "
                           "remove adjacent duplicates from the watch histories first")


def cowatch_graph(pairs, device="cuda:0"):
    """(edges int32 [U,2] with a < b in ascending order, counts int32 [U]) device tensors."""
    p = _device_pairs(pairs, device)
    P = p.shape[0]
    if P == 0:
        return p.new_zeros((0, 2)), p.new_zeros((0,))
    edges, counts, n_edges, flag = ops.cowatch_graph(p)
    _check_self_pairs(flag)
    U = int(n_edges.item())
    return edges[:U], counts[:U]


def select_cowatch(pairs, threshold, unique=False, device="cuda:0"):
    """int32 [S,2] device tensor (see module docstring)."""
    p = _device_pairs(pairs, device)
    if p.shape[0] == 0:
        return p
    out, n, flag = ops.cowatch_select(p, int(threshold), bool(unique))
    _check_self_pairs(flag)
    return out[:int(n.item())]
