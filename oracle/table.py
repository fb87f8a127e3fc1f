"""ORACLE (test infrastructure only -- never imported by the product path).

Trainable catalogue rows: numpy restatement of ``cdml_table_adam_rows`` (include/cdml.h).
The reference has no counterpart -- its features are frozen inputs (train.py:265) -- so this
is the build's own specification (BASELINE north_star: "the catalogue feature table and its
Adam states"); parity unpinned.

  G[row]  = sum of grad_xhat[r] over the batch rows r with idx[r] == row   (ascending r)
  dx      = inv * (G - x_hat * <x_hat, G>),  x_hat = x * inv,  inv = 1/sqrt(max(|x|^2, 1e-12))
            (the backward of tf.nn.l2_normalize, models.py:58)
  lazy Adam on the touched rows only (tf.contrib.opt.LazyAdamOptimizer's rule), with the
  arithmetic of oracle/tower.py::adam_step.
"""
import numpy as np

from . import tower


def grad_xhat(dz1, W1, dtype=np.float64):
    """dLoss/d x_hat = dz1 @ W1^T (the data gradient of the first layer)."""
    return dz1.astype(dtype) @ W1.astype(dtype).T


def table_adam_rows(table, m, v, idx, G, t, lr, row0=0, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
    """Returns updated copies (table, m, v); rows outside [row0, row0+len(table)) are skipped."""
    table, m, v = table.astype(dtype).copy(), m.astype(dtype).copy(), v.astype(dtype).copy()
    idx = np.asarray(idx, dtype=np.int64)
    local = idx - row0
    for row in np.unique(local[(local >= 0) & (local < table.shape[0])]):
        g = np.zeros(table.shape[1], dtype=dtype)
        for r in np.nonzero(local == row)[0]:          # ascending r
            g = g + G[r].astype(dtype)
        x = table[row]
        inv = 1.0 / np.sqrt(max(float(x @ x), 1e-12))
        xh = x * inv
        dx = inv * (g - xh * float(xh @ g))
        table[row], m[row], v[row] = tower.adam_step(x, dx, m[row], v[row], t, lr, beta1, beta2, eps, dtype=dtype)
    return table, m, v
