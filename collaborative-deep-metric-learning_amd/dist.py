"""Multi-GPU hooks: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-GPU (train.py:341-342: CUDA_VISIBLE_DEVICES="0",
"TODO Prepare distributed arguments here"); this is the build's data-parallel
extension of the hot path:

  * the catalogue is ROW-SHARDED: rank g owns global rows
    [g*rows_per_rank, (g+1)*rows_per_rank);
  * every rank samples its own slice of the global batch (the sampler is
    counter-based, so no communication is needed to agree on triplets);
  * ``RowExchange.gather``: all-to-all of requested row ids, owners gather +
    l2-normalise the rows from their shard (HIP kernel), all-to-all of the rows
    back -- the one real exchange step of the path;
  * ``GradSync``: one all-reduce (average) of the flat 35 MB gradient buffer.

Only torch.distributed plumbing lives here; the local gather is injected
(``local_gather``) so the routing logic is testable on CPU with gloo.
"""
import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n_rows_global, world, rank):
    """Rows [lo, hi) owned by ``rank``: equal blocks of ceil(N/world) rows."""
    per = (n_rows_global + world - 1) // world
    lo = min(rank * per, n_rows_global)
    return lo, min(lo + per, n_rows_global), per


def _host_staged(group):
    """gloo cannot move device tensors through all_to_all: a rehearsal run of the
    N>1 path on a box without RCCL peers stages the collectives through host
    memory.  With RCCL ("nccl") tensors go over xGMI directly."""
    return dist.get_backend(group) != "nccl"


def all_to_all(out, inp, out_splits=None, in_splits=None, group=None):
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_splits,
                               input_split_sizes=in_splits, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits,
                               input_split_sizes=in_splits, group=group)
    return out


def _hip_local_gather(table, ids, out):
    """Owner side: gather + l2-normalise local rows with the HIP kernel (fp32 table ->
    fp32 rows, fp16 table -> bf16 rows)."""
    if table.data.dtype == torch.float16:
        ops.gather_rows_f16(table.data, table.row0, ids, table.feature_size, out)
    else:
        ops.gather_rows(table.data, table.row0, ids, table.feature_size, out, normalize=True)
    return out


def raw_local_gather(table, ids, out):
    """Owner side for the fusion towers: rows as stored (they normalise the visual and the
    document part separately, models.py:86-88)."""
    ops.gather_rows(table.data, table.row0, ids, table.feature_size, out, normalize=False)
    return out


class RowExchange:
    """Fetch (normalised) feature rows by GLOBAL id from a row-sharded table."""

    def __init__(self, n_rows_global, group=None, local_gather=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows_global = int(n_rows_global)
        self.per = shard_bounds(n_rows_global, self.world, self.rank)[2]
        self.local_gather = local_gather or _hip_local_gather
        self._buf = {}

    def _scratch(self, name, shape, dtype, device):
        t = self._buf.get(name)
        if (t is None or t.shape[0] < shape[0] or t.shape[1:] != tuple(shape[1:]) or t.device != device
                or t.dtype != dtype):
            t = torch.empty(shape, dtype=dtype, device=device)
            self._buf[name] = t
        return t[:shape[0]]

    def plan(self, ids):
        """Route ids to owners: (send_ids sorted by owner, order, send_counts)."""
        owner = torch.div(ids, self.per, rounding_mode="floor").to(torch.int64)
        order = torch.argsort(owner, stable=True)
        counts = torch.bincount(owner, minlength=self.world)
        return ids[order].contiguous(), order, counts

    def gather(self, table, ids, out):
        """out[r] = l2norm(table_global[ids[r]]) for r < len(ids); out is [R, stride]."""
        dev = ids.device
        send_ids, order, send_counts = self.plan(ids)
        recv_counts = torch.empty_like(send_counts)
        all_to_all(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()       # host sync: split sizes
        n_req = int(sum(rc))
        req_ids = self._scratch("req_ids", (max(n_req, 1),), torch.int32, dev)[:n_req]
        all_to_all(req_ids, send_ids, rc, sc, self.group)
        stride = out.shape[1]
        rows_out = self._scratch("rows_out", (max(n_req, 1), stride), out.dtype, out.device)[:n_req]
        if n_req:
            self.local_gather(table, req_ids, rows_out)
        rows_in = self._scratch("rows_in", (ids.numel(), stride), out.dtype, out.device)
        all_to_all(rows_in, rows_out, sc, rc, self.group)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=dev)
        self.unpermute(rows_in, inv, out)
        # kept for scatter_back(): the same routing carries row gradients to their owners
        self.last_plan = (order, sc, rc, req_ids.clone() if n_req else req_ids)
        return out

    def scatter_back(self, rows):
        """The reverse trip of the last ``gather``: rows[r] (e.g. the gradient of the r-th
        requested row) goes to the rank that owns ids[r].  Returns (ids, rows) as the owner
        sees them: every request it served, in the order it served them (source rank, then the
        requester's order) -- duplicates included."""
        order, sc, rc, req_ids = self.last_plan
        stride = rows.shape[1]
        send = self._scratch("back_send", (order.numel(), stride), rows.dtype, rows.device)
        self.unpermute(rows, order, send)                    # send[j] = rows[order[j]]
        n_req = int(sum(rc))
        recv = self._scratch("back_recv", (max(n_req, 1), stride), rows.dtype, rows.device)[:n_req]
        all_to_all(recv, send, rc, sc, self.group)
        return req_ids, recv

    def unpermute(self, rows_in, inv, out):
        """out[r] = rows_in[inv[r]] -- a row gather of the receive buffer."""
        if rows_in.is_cuda:
            if rows_in.element_size() == 2:          # bf16 rows move as fp32 words (bitwise copy)
                rows_in, out = rows_in.view(torch.float32), out.view(torch.float32)
            ops.gather_rows(rows_in, 0, inv.to(torch.int32), out.shape[1], out[:inv.numel()],
                            normalize=False)
        else:
            out[:inv.numel()] = rows_in[inv]


class GradSync:
    """Average the flat gradient buffer over the data-parallel group."""

    def __init__(self, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self.avg = False
        if self.nccl and self.world > 1:
            # probe once whether the communicator implements ReduceOp.AVG (every rank
            # runs the same probe, so the collective order stays identical)
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            try:
                t = torch.ones(1, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                self.avg = abs(float(t.item()) - 1.0) < 1e-6
            except Exception:
                self.avg = False

    def start(self, flat_grad, lo, hi):
        """Begin averaging flat_grad[lo:hi]; with RCCL the collective is asynchronous
        (it runs on the communicator's stream behind everything enqueued so far and
        overlaps whatever the compute stream does next).  Returns a handle for finish."""
        seg = flat_grad[lo:hi]
        if self.world == 1:
            return None
        if self.nccl:
            op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
            return (dist.all_reduce(seg, op=op, group=self.group, async_op=True), seg)
        self(seg)                                    # gloo: synchronous, host-staged
        return None

    def finish(self, handles):
        """Make the compute stream wait for the started collectives."""
        for h in handles:
            if h is not None:
                work, seg = h
                work.wait()
                if not self.avg:
                    seg.div_(self.world)

    def __call__(self, flat_grad):
        if self.world == 1:
            return flat_grad
        if self.nccl:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=self.group)
            if not self.avg:
                flat_grad.div_(self.world)
        elif flat_grad.is_cuda:                      # gloo rehearsal: stage through host
            h = flat_grad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.copy_(h.div_(self.world))
        else:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            flat_grad.div_(self.world)
        return flat_grad


class Prefetcher:
    """Double-buffered run-ahead of the row exchange.

    The sampler is counter-based, so the rows of step t+1 are known while step t
    computes: ``launch`` enqueues sampling + exchange for one buffer on a side
    stream (the host sync for the all-to-all split sizes then waits only for that
    stream, not for the 2.7 ms of GEMMs queued on the compute stream), ``acquire``
    makes the compute stream wait for a filled buffer and ``release`` tells the
    side stream the buffer may be overwritten.  Give the exchange its own process
    group (communicator): collectives of one group execute in issue order, and the
    exchange of step t+1 must not queue behind the gradient all-reduce of step t."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            self.stream = torch.cuda.Stream(self.device)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self._released = [False, False]
        self._cold = True

    def launch(self, b, fill_fn):
        if not self.cuda:
            fill_fn()
            return
        if self._cold:
            # first use: everything set up on the compute stream so far (table fill,
            # uploads, weight init) must be visible to the side stream
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            self._cold = False
        with torch.cuda.stream(self.stream):
            if self._released[b]:
                self.stream.wait_event(self.free[b])
            fill_fn()
            self.ready[b].record(self.stream)

    def acquire(self, b):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(self.ready[b])

    def release(self, b):
        if self.cuda:
            self.free[b].record(torch.cuda.current_stream(self.device))
            self._released[b] = True
