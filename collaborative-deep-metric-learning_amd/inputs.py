"""Input pipe: triplet sampler + feature gather on the device, behind the
reference's pipe API (inputs.py:62-172).

Reference behaviour kept: ``MPTripletPipe(cowatch_file_patten, feature_file,
wait_times)``; ``.cowatch_num``; ``.create_pipe(num_epochs, batch_size)``;
``.get_batch()`` -> float32 [batch, 3, feature_size] (anchor, positive,
negative feature rows) or ``None`` once ``num_epochs`` sequential passes over
the co-watch pairs are exhausted (the final partial batch is dropped,
inputs.py:110-122); negatives are uniform over the catalogue and redrawn while
they equal the anchor or the positive (inputs.py:125-127).

What changed: no worker processes, queues, pickling, host gather or host->device
copy -- the catalogue and the pair list live in HBM and one HIP launch samples
and gathers a batch.  The returned batch is a device tensor.  The random stream
is the counter-based one specified in oracle/sampler.py (the reference's forked
workers all replay one un-reseeded MT19937 stream); ``get_batch(indices=...)``
replays caller-supplied triplets for reference-identical batches.
"""
import glob

import numpy as np
import torch

from . import ops
from .engine import FeatureTable

MODE_UNIFORM = 0
MODE_INBATCH = 1


class BasePipe(object):
    """Inherit from this class when implementing new readers (inputs.py:24-29)."""

    def create_pipe(self, unused_data, **unused_params):
        raise NotImplementedError()


def read_cowatch_files(files):
    """``a,p\\n`` ASCII lines (written by online_data.py:277-280) -> int32 [P,2]."""
    parts = []
    for f in files:
        a = np.loadtxt(f, delimiter=",", dtype=np.int64, ndmin=2)
        if a.size:
            parts.append(a.reshape(-1, 2))
    if not parts:
        return np.zeros((0, 2), dtype=np.int32)
    return np.concatenate(parts).astype(np.int32)


class TripletPipe(BasePipe):
    """In-memory pipe over precomputed triplet features (inputs.py:32-59: a tf.data
    ``from_tensor_slices(...).repeat(num_epochs).batch(batch_size).shuffle(buffer_size)``
    one-shot iterator; imitation_data.gen_triplets feeds it in the reference's tests).
    Same order of transformations: elements repeat ``num_epochs`` times (``None`` = forever),
    are cut into batches (a batch may straddle two passes, the last one may be short) and the
    BATCHES are shuffled through a ``buffer_size`` reservoir.  The triplets live on the device;
    ``get_next()`` returns device tensors [b, 3, F] and raises ``StopIteration`` at the end
    (tf: OutOfRangeError)."""

    def __init__(self, triplets, device="cuda:0", seed=0):
        t = triplets if torch.is_tensor(triplets) else torch.as_tensor(np.asarray(triplets))
        if t.dim() == 2 and t.shape[1] == 3:             # row-id triplets (tests/test_inputs.py feeds guids;
            self.triplets = t.to(device=device, dtype=torch.int32)    # parse_data.lookup turns a batch into features)
        elif t.dim() == 3 and t.shape[1] == 3:
            self.triplets = t.to(device=device, dtype=torch.float32)
        else:
            raise ValueError("triplets must be [N, 3] row ids or [N, 3, feature_size] features")
        self.seed = seed

    def create_pipe(self, batch_size=10, num_epochs=None, num_readers=1, buffer_size=1000):
        return _TripletIterator(self.triplets, int(batch_size), num_epochs, int(buffer_size), self.seed)


class _TripletIterator:
    def __init__(self, triplets, batch_size, num_epochs, buffer_size, seed):
        self.t, self.B, self.n = triplets, batch_size, triplets.shape[0]
        self.total = None if num_epochs is None else self.n * int(num_epochs)
        self.pos = 0                                   # next element of the repeated stream
        self.buf, self.cap = [], max(buffer_size, 1)
        self.rng = np.random.RandomState(seed)

    def _next_batch(self):
        if self.n == 0 or (self.total is not None and self.pos >= self.total):
            return None
        hi = self.pos + self.B if self.total is None else min(self.pos + self.B, self.total)
        idx = torch.arange(self.pos, hi, device=self.t.device) % self.n
        self.pos = hi
        return self.t[idx]

    def get_next(self):
        while len(self.buf) < self.cap:                # tf.data shuffle: fill the buffer, draw one at random
            b = self._next_batch()
            if b is None:
                break
            self.buf.append(b)
        if not self.buf:
            raise StopIteration
        return self.buf.pop(self.rng.randint(len(self.buf)))

    __next__ = get_next

    def __iter__(self):
        return self


class MPTripletPipe(BasePipe):
    def __init__(self, cowatch_file_patten=None, feature_file=None, wait_times=30,
                 device="cuda:0", seed=1234, pairs=None, table=None):
        """cowatch_file_patten / feature_file as in the reference (inputs.py:63-77);
        alternatively pass ``pairs`` (int [P,2]) and ``table`` (FeatureTable or
        ndarray) that are already in memory."""
        self.device = torch.device(device)
        self.seed = int(seed)
        self.wait_times = wait_times          # kept for signature parity; nothing waits
        if pairs is None:
            self.cowatch_files = sorted(glob.glob(cowatch_file_patten))
            pairs = read_cowatch_files(self.cowatch_files)
        else:
            self.cowatch_files = []
        pairs = np.ascontiguousarray(np.asarray(pairs, dtype=np.int32).reshape(-1, 2))
        if len(pairs) == 0:
            raise ValueError("no co-watch pairs")
        self.cowatch_num = int(len(pairs))                       # inputs.py:70,79-86
        self.pairs = torch.from_numpy(pairs).to(self.device)
        if table is None:
            table = np.load(feature_file)                        # online_data.py:87-93
        if not isinstance(table, FeatureTable):
            table = FeatureTable.from_numpy(table, self.device)
        self.table = table
        self.feature_size = table.feature_size
        self.batch_size = None

    # ------------------------------------------------------------------
    def create_pipe(self, num_epochs, batch_size, queue_length=None):
        self.batch_size = int(batch_size)
        self.num_epochs = num_epochs
        self.step = 0
        if num_epochs is None:
            self.num_batches = None
        else:
            self.num_batches = (self.cowatch_num * int(num_epochs)) // self.batch_size
        B = self.batch_size
        self._idx = torch.empty((B, 3), dtype=torch.int32, device=self.device)
        self._oob = torch.zeros(1, dtype=torch.int32, device=self.device)

    def exhausted(self):
        return self.num_batches is not None and self.step >= self.num_batches

    def sample_indices(self, step=None):
        """int32 [batch,3] (a,p,n) video ids of batch ``step`` (device tensor)."""
        s = self.step if step is None else step
        return ops.sample_uniform(self.pairs, self.table.n_rows_global, self.seed, s,
                                  self.batch_size, self._idx)

    def get_batch(self, indices=None):
        """Next batch [batch,3,feature_size] fp32 on the device, or None when the
        pair stream is exhausted (train.py:300-306 treats None as end of data)."""
        if self.batch_size is None:
            raise RuntimeError("create_pipe() first")
        if indices is None:
            if self.exhausted():
                return None
            idx = self.sample_indices()
            self.step += 1
        else:
            idx = torch.as_tensor(indices, dtype=torch.int32).to(self.device).contiguous()
        n = idx.numel()
        out = torch.empty((n, self.feature_size), dtype=torch.float32, device=self.device)
        ops.gather_rows(self.table.data, self.table.row0, idx.view(-1), self.feature_size, out,
                        normalize=False, oob_flag=self._oob)
        return out.view(-1, 3, self.feature_size)

    def check_indices(self):
        """Raise if any replayed index fell outside the catalogue (sync point)."""
        if int(self._oob.item()):
            raise IndexError("triplet index outside the feature table")

    def __del__(self):
        pass
