"""ORACLE (test infrastructure only -- never imported by the product path).

Co-watch graph statistics of the reference's ETL restated in numpy:
  cowatch_graph   = get_cowatch_graph (parse_data.py:221-254): multiplicity of every undirected
                    edge {a, p}; a pair with a == p raises, as there (:244-246);
  select_cowatch  = select_cowatch (parse_data.py:256-289): pairs whose edge multiplicity is
                    >= threshold, all occurrences in input order (threshold <= 1: everything);
                    unique=True: each qualifying edge once (the reference shuffles orientation
                    and order, so only the SET is defined; returned as (min, max), ascending).
Pinned by tests/golden/cowatch_graph_seed7.npz, produced by running the reference's own
functions (tests/golden/make_golden.py).
"""
import numpy as np


def _keys(pairs):
    p = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    if (p[:, 0] == p[:, 1]).any():
        raise RuntimeError("get_cowatch_graph: self pair")
    return (np.minimum(p[:, 0], p[:, 1]) << 32) | np.maximum(p[:, 0], p[:, 1])


def cowatch_graph(pairs):
    """(edges int64 [U,2] ascending with a < b, counts int64 [U])."""
    uniq, counts = np.unique(_keys(pairs), return_counts=True)
    return np.stack([uniq >> 32, uniq & 0xffffffff], 1), counts


def select_cowatch(pairs, threshold, unique=False):
    p = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    k = _keys(p)
    uniq, inv, counts = np.unique(k, return_inverse=True, return_counts=True)
    thr = max(int(threshold), 1)
    if unique:
        u = uniq[counts >= thr]
        return np.stack([u >> 32, u & 0xffffffff], 1)
    return p[counts[inv] >= thr]
