"""ORACLE (test infrastructure only -- never imported by the product path).

torch-CPU form of the reference training step, used to TIME the CPU baseline that BASELINE.md
section 4 promises next to every GPU number (TensorFlow 1.13 itself cannot run here).  Same
arithmetic as oracle/tower.py, which restates the reference (tests/test_oracle.py checks the two
against each other):

  fetch  = the pair stream + the reference's negative rule (uniform over the catalogue, redraw
           while the draw is the anchor or the positive: parse_data.py:292-298, inputs.py:125-127)
           + the numpy fancy-index gather ``FEATURES[idx]`` (inputs.py:158) + the reshape to
           [3B, F] (train.py:313)
  train  = tf.nn.l2_normalize -> FC(5000) + leaky_relu(0.2) -> FC(256) + leaky_relu(0.2) ->
           tf.nn.l2_normalize (models.py:46-62), squared-L2 hinge loss (losses.py:32-38),
           gradients wrt the four variables only (train.py:141), TF-form Adam (train.py:146)

mirroring the reference's own fetch / train timers (train.py:314-323).  Threads are torch's
intra-op threads (``torch.set_num_threads``)."""
import math
import time

import numpy as np
import torch

L2_EPS = 1e-12
LRELU_ALPHA = 0.2


def l2_normalize(t):
    return t * torch.rsqrt(torch.clamp((t * t).sum(-1, keepdim=True), min=L2_EPS))


def leaky_relu(t):
    return torch.maximum(LRELU_ALPHA * t, t)


class CpuStep:
    def __init__(self, table, pairs, batch, hidden=5000, out=256, margin=0.8, lr=0.01, seed=1234, weight_seed=42):
        self.table = np.ascontiguousarray(table, dtype=np.float32)       # FEATURES (inputs.py:19,73-74)
        self.pairs = np.asarray(pairs)
        self.B, self.margin, self.lr = int(batch), float(margin), float(lr)
        self.rng = np.random.RandomState(seed)
        n, F = self.table.shape
        g = torch.Generator().manual_seed(weight_seed)

        def xavier(fi, fo):                                              # slim default initializer
            lim = math.sqrt(6.0 / (fi + fo))
            return ((torch.rand((fi, fo), generator=g) * 2 - 1) * lim).requires_grad_(True)
        self.W = [xavier(F, hidden), torch.zeros(hidden, requires_grad=True),
                  xavier(hidden, out), torch.zeros(out, requires_grad=True)]
        self.m = [torch.zeros_like(w) for w in self.W]
        self.v = [torch.zeros_like(w) for w in self.W]
        self.t = 0
        self.pos = 0

    def load(self, W):
        with torch.no_grad():
            for dst, src in zip(self.W, W):
                dst.copy_(torch.as_tensor(np.asarray(src, dtype=np.float32)))

    # ---- inputs.py:102-166 ----------------------------------------------------
    def sample(self):
        P, N = len(self.pairs), len(self.table)
        q = (self.pos + np.arange(self.B)) % P
        self.pos = (self.pos + self.B) % P
        ap = self.pairs[q]
        neg = self.rng.randint(0, N, size=self.B)
        bad = (neg == ap[:, 0]) | (neg == ap[:, 1])
        while bad.any():                                                 # redraw while n in {a, p}
            neg[bad] = self.rng.randint(0, N, size=int(bad.sum()))
            bad = (neg == ap[:, 0]) | (neg == ap[:, 1])
        return np.concatenate([ap, neg[:, None]], axis=1)

    def fetch(self, idx=None):
        idx = self.sample() if idx is None else idx
        batch = self.table[idx]                                          # [B,3,F], inputs.py:158
        return torch.from_numpy(batch.reshape(-1, batch.shape[-1]))      # train.py:313

    # ---- models.py:46-62, losses.py:32-38, train.py:141,146 ---------------------
    def forward(self, x):
        W1, b1, W2, b2 = self.W
        h1 = leaky_relu(l2_normalize(x) @ W1 + b1)
        z = leaky_relu(h1 @ W2 + b2)
        e = l2_normalize(z)
        t = e.view(-1, 3, e.shape[-1])
        a, p, n = t[:, 0], t[:, 1], t[:, 2]
        pos = ((a - p) ** 2).sum(-1)
        neg = ((a - n) ** 2).sum(-1)
        loss = torch.clamp(pos - neg + self.margin, min=0).mean()
        return e, loss

    def train(self, x, b1=0.9, b2=0.999, eps=1e-8):
        e, loss = self.forward(x)
        grads = torch.autograd.grad(loss, self.W)
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for w, g, m, v in zip(self.W, grads, self.m, self.v):        # ApplyAdam
                m.add_((g - m) * (1.0 - b1))
                v.add_((g * g - v) * (1.0 - b2))
                w.sub_(lr_t * m / (v.sqrt() + eps))
        return e, float(loss.detach()), grads


def time_steps(step, max_steps, budget_s, warmup):
    """Median fetch / train seconds per step over up to max_steps timed steps (stops early when
    the time budget is spent, at least 3 steps)."""
    for _ in range(warmup):
        step.train(step.fetch())
    tf, tt = [], []
    t_end = time.perf_counter() + budget_s
    while len(tf) < max_steps and (len(tf) < 3 or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        x = step.fetch()
        t1 = time.perf_counter()
        step.train(x)
        t2 = time.perf_counter()
        tf.append(t1 - t0)
        tt.append(t2 - t1)
    return float(np.median(tf)), float(np.median(tt)), len(tf)
