"""ORACLE (test infrastructure only -- never imported by the product path).

Exact k-nearest neighbours as the reference's ``calc_knn`` defines its result
(faiss_knn.py:82-131): l2-normalise rows (``x / np.linalg.norm(x)``, :99-104),
then for every query the ``nearest_num`` catalogue rows of smallest SQUARED L2
distance, nearest first -- what ``faiss.Index*.search`` returns as (D, I) (:128).

The reference obtains this from ``faiss.IndexHNSWFlat`` (third-party, not under
/root/reference, version unpinned by the reference; absent from this image), an
approximate index whose ground truth is the exact search restated here.  Parity
unpinned: the reference holds no test or golden vector for this function.
Ties are ordered by id (faiss leaves tie order unspecified).
"""
import numpy as np


def calc_knn_exact(embeddings, q_embeddings=None, nearest_num=51, l2_norm=True):
    b = np.asarray(embeddings, dtype=np.float32).astype(np.float64)
    q = b if q_embeddings is None else np.asarray(q_embeddings, dtype=np.float32).astype(np.float64)
    if l2_norm:
        b = b / np.maximum(np.linalg.norm(b, axis=1, keepdims=True), 1e-6)
        q = b if q_embeddings is None else q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-6)
    d = (q * q).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2.0 * q @ b.T
    d = np.maximum(d, 0.0)
    k = int(nearest_num)
    ids = np.arange(b.shape[0])
    D = np.full((q.shape[0], k), np.inf)
    I = np.full((q.shape[0], k), -1, dtype=np.int64)
    for r in range(q.shape[0]):
        order = np.lexsort((ids, d[r]))[:k]
        D[r, :len(order)] = d[r, order]
        I[r, :len(order)] = order
    return D, I, d
