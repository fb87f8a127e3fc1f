"""Multi-GPU hooks: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-GPU (train.py:341-342: CUDA_VISIBLE_DEVICES="0",
"TODO Prepare distributed arguments here"); this is the build's data-parallel
extension of the hot path:

  * the catalogue is ROW-SHARDED: rank g owns global rows
    [g*rows_per_rank, (g+1)*rows_per_rank);
  * every rank samples its own slice of the global batch (the sampler is
    counter-based, so no communication is needed to agree on triplets);
  * ``RowExchange.gather``: FIXED-CAPACITY all-to-all.  A HIP kernel routes the R
    requested ids into per-owner segments of ``capacity`` slots (unused slots -1),
    one all-to-all with equal splits carries the ids, the owners gather +
    l2-normalise the rows of their shard (HIP kernel; -1 slots skipped), a second
    equal-split all-to-all carries the rows back and a HIP gather puts them in
    request order.  No host-side counts anywhere: the step is enqueue-only and the
    same every step (hipGraph-capturable); a segment that overflows raises a device
    flag that ``check_overflow`` reads off the critical path;
  * ``GradSync``: bucketed all-reduce (average) of the flat 35 MB gradient buffer.

Only torch.distributed plumbing lives here; the local gather is injected
(``local_gather``) so the routing logic is testable on CPU with gloo.
"""
import math

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


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class EagerCollectiveStreams:
    """The streams that have carried an EAGER RCCL collective, so that none of them is ever used as the origin of a
    hipGraph capture.

    Why (the abort of round 5): torch runs a synchronous eager collective on the caller's current
    stream and records the work's end event THERE; the RCCL watchdog thread polls that event (hipEventQuery, every
    100 ms) until it has seen it complete.  A capture that begins on such a stream before the next poll makes HIP answer
    the query with hipErrorCapturedEvent, the watchdog rethrows, the process ends.  Rounds 3-4 captured the split
    form's exchange graph with the prefetch stream -- which carries the eager exchanges -- as its origin; round 5 moved
    the capture to a stream of its own.  This registry turns "no stream that ever carried an eager collective enters
    a capture" from a convention into a checked invariant: every collective entry point of this module notes the
    stream it was issued on (``note``), and ``TrainStep._capture`` refuses an origin that ``carried`` one.

    Streams are identified by (device index, native handle); objects without those attributes (the CPU tests' stand-ins)
    by their ``id``.  Process-wide: a stream is a property of the process, not of one hook."""
    _seen = set()

    @staticmethod
    def key(stream):
        h = getattr(stream, "cuda_stream", None)
        if h is None:
            return ("obj", id(stream))
        return (getattr(stream, "device_index", None), int(h))

    @classmethod
    def note(cls, stream=None):
        """An eager collective is being issued on ``stream`` (None: the current stream).  No-op inside a capture (captured
        collectives are never put on a watchdog work list) and without a GPU."""
        if stream is None:
            if not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
                return
            stream = torch.cuda.current_stream()
        cls._seen.add(cls.key(stream))

    @classmethod
    def carried(cls, stream):
        return cls.key(stream) in cls._seen

    @classmethod
    def assert_clean_origin(cls, stream, what="capture origin"):
        if cls.carried(stream):
            raise RuntimeError(
                "%s %r has carried an eager RCCL collective: the RCCL watchdog still polls work events recorded on it, and "
                "a capture that pulls it in ends the process with hipErrorCapturedEvent (DESIGN.md section 7).  Capture "
                "from a stream that never issues eager collectives (TrainStep._ex_origin)" % (what, stream))


def new_capture_group(like=None):
    """A process group over the ranks of ``like`` (None: the world) that is used ONLY inside hipGraph captures.

    Why a group of its own (the hazard round 3 slept around): torch.distributed's RCCL watchdog thread keeps every
    collective issued EAGERLY on a process group on that group's work list and polls its end event (hipEventQuery,
    every 100 ms) until it has seen it complete.  The event was recorded on the group's communicator stream; once a
    capture pulls that stream in (a captured collective on the same group), HIP answers the query of ANY event last
    recorded there with hipErrorCapturedEvent, the watchdog rethrows, and the process ends -- whenever a capture
    begins before the watchdog's next poll has retired the last eager work.  Collectives issued DURING a capture are
    never put on the work list.  So a group that carries no eager collective at all has an empty work list for ever,
    and its communicator stream may enter captures freely; the groups that do carry eager collectives never see a
    capture.  Deterministic: nothing depends on the watchdog's period or on how loaded the host is.

    The communicator must come up without a collective (a first collective would be an eager work, and communicator
    setup cannot run inside a capture): with the default group bound to its device -- init_process_group(...,
    device_id=torch.device("cuda", i)) -- torch connects a new group's communicator eagerly at creation
    (distributed_c10d._new_process_group_helper -> eager_connect_single_device / ncclCommSplit).  Without that binding
    this raises.  Collective call: every rank of ``like`` creates the group at the same point."""
    default = capture_groups_supported()
    if default is None:
        raise RuntimeError("capturing RCCL collectives into a hipGraph needs a process group bound to its device: "
                           "init_process_group(..., device_id=torch.device('cuda', local_rank))")
    ranks = None if like is None or like is default else dist.get_process_group_ranks(like)
    return dist.new_group(ranks=ranks, backend="nccl")


def capture_groups_supported():
    """The default process group when ``new_capture_group`` can work with it -- it is bound to its device, so torch
    connects a new group's communicator eagerly at creation -- else None.  Reads two private names of
    torch.distributed (``distributed_c10d._get_default_group``, ``ProcessGroup.bound_device_id``; present in torch
    2.3 .. 2.10): where either is missing the answer is None, and ``TrainStep`` then steps eagerly with a warning instead
    of failing at construction."""
    get = getattr(getattr(dist, "distributed_c10d", None), "_get_default_group", None)
    if get is None or not dist.is_initialized():
        return None
    try:
        default = get()
    except Exception:                                     # noqa: BLE001 -- no default group
        return None
    return default if getattr(default, "bound_device_id", None) is not None else None


def all_to_all(out, inp, group=None):
    """Equal-split all-to-all of the leading dimension."""
    if inp.is_cuda and _host_staged(group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        if inp.is_cuda:
            EagerCollectiveStreams.note()
        dist.all_to_all_single(out, inp, group=group)
    return out


def _hip_local_gather(table, ids, out):
    """Owner side: gather + l2-normalise local rows with the HIP kernel (fp32 table ->
    fp32 rows, fp16 table -> bf16 rows); id -1 = padding slot, left untouched."""
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


def exchange_capacity(n_requests, world, factor=1.25):
    """Slots per peer: the mean share with ``factor`` headroom plus six standard deviations of
    a uniform draw (small batches), a multiple of 8, never more than all requests."""
    if world == 1:
        return int(n_requests)
    mean = n_requests / world
    cap = int(math.ceil(mean * factor + 6.0 * math.sqrt(mean))) + 8
    return min(int(n_requests), (cap + 7) // 8 * 8)


class RowExchange:
    """Fetch (normalised) feature rows by GLOBAL id from a row-sharded table."""

    def __init__(self, n_rows_global, group=None, local_gather=None, capacity_factor=1.25, skip_self=True,
                 capture_group=None):
        """``skip_self``: with a single rank every request is local, so the two all-to-alls are
        skipped (False keeps them: the RCCL entry points then run even at world size 1).
        ``capture_group``: the process group the all-to-alls go through INSIDE a hipGraph capture
        (``new_capture_group``; TrainStep creates it when asked to replay an RCCL step from graphs)."""
        self.group = group
        self.capture_group = capture_group
        self.skip_self = bool(skip_self)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_rows_global = int(n_rows_global)
        self.per = shard_bounds(n_rows_global, self.world, self.rank)[2]
        self.local_gather = local_gather or _hip_local_gather
        self.capacity_factor = float(capacity_factor)
        self._buf = {}
        self.overflow = None
        self.last = None

    def _scratch(self, name, shape, dtype, device):
        t = self._buf.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device or t.dtype != dtype:
            t = torch.zeros(shape, dtype=dtype, device=device)
            self._buf[name] = t
        return t

    def capacity(self, n_requests):
        return exchange_capacity(n_requests, self.world, self.capacity_factor)

    def route(self, ids, cap):
        """(send_ids int32[world*cap] with -1 padding, slot int32[R]): request r sits in slot
        owner*cap + k, k counting the owner's requests in ascending r."""
        dev, R = ids.device, ids.numel()
        if self.overflow is None or self.overflow.device != dev:
            self.overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        send_ids = self._scratch("send_ids", (self.world * cap,), torch.int32, dev)
        slot = self._scratch("slot", (R,), torch.int32, dev)
        if ids.is_cuda:
            ops.route_rows(ids, self.per, self.world, cap, send_ids, slot, self.overflow)
        else:                                   # CPU tensors: the gloo unit test of this plumbing
            owner = torch.div(ids, self.per, rounding_mode="floor").to(torch.int64)
            bad = (ids < 0) | (owner >= self.world)
            if bool(bad.any()):
                self.overflow |= 2
            owner = owner.clamp(0, self.world - 1)
            onehot = torch.nn.functional.one_hot(owner, self.world)
            pos = (onehot.cumsum(0) - onehot).gather(1, owner[:, None])[:, 0]
            ok = (pos < cap) & ~bad
            if bool((~ok & ~bad).any()):
                self.overflow |= 1
            send_ids.fill_(-1)
            s = (owner * cap + pos).to(torch.int32)
            send_ids[s[ok].long()] = ids[ok]
            slot.copy_(torch.where(ok, s, torch.full_like(s, -1)))
        return send_ids, slot

    def gather(self, table, ids, out):
        """out[r] = l2norm(table_global[ids[r]]) for r < len(ids); out is [R, stride]."""
        dev, R = ids.device, ids.numel()
        cap = self.capacity(R)
        send_ids, slot = self.route(ids, cap)
        n_slots = self.world * cap
        local_only = self.world == 1 and self.skip_self
        recv_ids = send_ids if local_only else self._scratch("recv_ids", (n_slots,), torch.int32, dev)
        group = self.capture_group if (self.capture_group is not None and _capturing()) else self.group
        if not local_only:
            all_to_all(recv_ids, send_ids, group)
        # split-fp32 path (round 5): ``out`` is the GEMMs' plane buffer [R, 3 planes] bf16 while the table is fp32 -- the rows
        # TRAVEL as fp32 (6 KB, not 9 KB of planes) and the un-permute pass writes the planes: request order and operand
        # form in one launch on the prefetch stream, no fp32 x_hat and no split launch on the compute stream
        planes = out.dtype == torch.bfloat16 and table.data.dtype == torch.float32
        stride = out.shape[1] // 3 if planes else out.shape[1]
        row_dtype = torch.float32 if planes else out.dtype
        rows_out = self._scratch("rows_out", (n_slots, stride), row_dtype, out.device)
        self.local_gather(table, recv_ids, rows_out)
        rows_in = rows_out if local_only else self._scratch("rows_in", (n_slots, stride), row_dtype, out.device)
        if not local_only:
            all_to_all(rows_in, rows_out, group)
        if planes:
            ops.gather_rows_x3(rows_in, slot, stride, out[:slot.numel()], nan_missing=True)
        else:
            self.unpermute(rows_in, slot, out)
        # kept for scatter_back(): the same routing carries row gradients to their owners
        self.last = (slot, recv_ids, cap)
        return out

    def scatter_back(self, rows):
        """The reverse trip of the last ``gather``: rows[r] (e.g. the gradient of the r-th
        requested row) goes to the rank that owns ids[r].  Returns (ids, rows) as the owner
        sees them: world*capacity slots in the order it was asked (source rank, then the
        requester's order; duplicates included), unused slots carrying id -1."""
        slot, recv_ids, cap = self.last
        n_slots, stride = self.world * cap, rows.shape[1]
        send = self._scratch("back_send", (n_slots, stride), rows.dtype, rows.device)
        if rows.is_cuda:
            ops.scatter_rows(rows, slot, send, stride)
        else:
            ok = slot >= 0
            send[slot[ok].long()] = rows[ok]
        if self.world == 1 and self.skip_self:
            return recv_ids, send
        recv = self._scratch("back_recv", (n_slots, stride), rows.dtype, rows.device)
        all_to_all(recv, send, self.capture_group if (self.capture_group is not None and _capturing()) else self.group)
        return recv_ids, recv

    def unpermute(self, rows_in, slot, out):
        """out[r] = rows_in[slot[r]] -- a row gather of the receive buffer.  A request that found
        no slot (slot -1: its owner's segment was full) gets a NaN row, so the SAME step's loss is
        NaN and the overflow cannot train on a stale row unnoticed (``check_overflow`` names it)."""
        if rows_in.is_cuda:
            if rows_in.element_size() == 2:          # bf16 rows move as fp32 words (bitwise copy)
                rows_in, out = rows_in.view(torch.float32), out.view(torch.float32)
            ops.gather_rows(rows_in, 0, slot, out.shape[1], out[:slot.numel()], normalize=False,
                            nan_missing=True)
        else:
            ok = slot >= 0
            out[:slot.numel()][ok] = rows_in[slot[ok].long()]
            out[:slot.numel()][~ok] = float("nan")

    def check_overflow(self):
        """Host check of the device-side flag (a sync: call it off the critical path -- the
        trainer does at its logging cadence, bench.py after the timed region)."""
        if self.overflow is None:
            return
        f = int(self.overflow.item())
        if f & 2:
            raise IndexError("row exchange: a requested id lies outside the %d-row catalogue" % self.n_rows_global)
        if f & 1:
            raise RuntimeError("row exchange: a peer segment overflowed (requests are skewed towards one shard); "
                               "raise RowExchange(capacity_factor=%.2f)" % self.capacity_factor)

    def bytes_per_step(self, n_requests, x, table=None):
        """Bytes this rank sends per step (ids out + rows back), padding included.  (Split-fp32 path: ``x`` is the plane
        buffer, the rows travel as fp32 -- a third of its columns, four bytes each.)"""
        n_slots = self.world * self.capacity(n_requests)
        planes = table is not None and x.dtype == torch.bfloat16 and table.data.dtype == torch.float32
        row_bytes = (x.shape[1] // 3) * 4 if planes else x.shape[1] * x.element_size()
        return int(n_slots * 4 + n_slots * row_bytes)


class GradSync:
    """Average the flat gradient buffer over the data-parallel group."""

    def __init__(self, group=None, device=None, skip_self=True, capture_group=None):
        """``skip_self``: a single rank has nothing to average and issues no collective (False
        keeps the RCCL calls: they then run, and can be captured, even at world size 1).
        ``capture_group``: the group the all-reduces go through inside a hipGraph capture
        (``new_capture_group``)."""
        self.group = group
        self.capture_group = capture_group
        self.world = dist.get_world_size(group)
        self.nccl = dist.get_backend(group) == "nccl"
        self.active = self.world > 1 or (self.nccl and not skip_self)
        self.avg = False
        if self.nccl and self.active:
            # probe once whether the communicator implements ReduceOp.AVG (every rank
            # runs the same probe, so the collective order stays identical)
            dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            try:
                t = torch.ones(1, device=dev)
                EagerCollectiveStreams.note()
                dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
                self.avg = abs(float(t.item()) - 1.0) < 1e-6
            except Exception:
                self.avg = False

    def start(self, flat_grad, lo, hi):
        """Begin averaging flat_grad[lo:hi]; with RCCL the collective is asynchronous
        (it runs on the communicator's stream behind everything enqueued so far and
        overlaps whatever the compute stream does next).  Returns a handle for finish."""
        seg = flat_grad[lo:hi]
        if not self.active:
            return None
        if self.nccl:
            op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
            group = self.capture_group if (self.capture_group is not None and _capturing()) else self.group
            EagerCollectiveStreams.note()
            return (dist.all_reduce(seg, op=op, group=group, async_op=True), seg)
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
        if not self.active:
            return flat_grad
        if self.nccl:
            group = self.capture_group if (self.capture_group is not None and _capturing()) else self.group
            EagerCollectiveStreams.note()
            dist.all_reduce(flat_grad, op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=group)
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


def reduce_input_flags(oob, overflow, group=None, world=None, extra=None):
    """The device-side input flags of ONE rank -- ``oob`` (int32[1]: a pair id outside the catalogue) and ``overflow``
    (RowExchange.overflow, int32[1] bit field: 1 = a peer segment overflowed, 2 = a requested id outside the catalogue;
    None = no exchange) -- as (oob, overflowed, bad_id) over EVERY rank of ``group`` (maximum; a host read, a sync).
    Collective: an overflow on one rank poisons every rank's weights through the gradient all-reduce (its rows come back
    NaN), so every rank has to learn of it -- and raise -- at the same point; a rank that read only its own flag would
    save a NaN checkpoint while the overflowing rank raised alone and left the others waiting in their next collective.
    ``extra`` (int32[1] or None): one more per-rank flag carried by the SAME all-reduce (the trainer's "my weights are
    not finite"); with it the result has a fourth element, its maximum over the ranks."""
    dev = oob.device
    ov = overflow.to(torch.int32).view(1) if overflow is not None else torch.zeros(1, dtype=torch.int32, device=dev)
    parts = [oob.to(torch.int32).view(1), ov & 1, (ov >> 1) & 1]
    if extra is not None:
        parts.append(extra.to(torch.int32).view(1))
    f = torch.cat(parts)
    world = dist.get_world_size(group) if (world is None and dist.is_available() and dist.is_initialized()) else (world or 1)
    if world > 1:
        if f.is_cuda and _host_staged(group):
            h = f.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=group)
            f = h
        else:
            if f.is_cuda:
                EagerCollectiveStreams.note()
            dist.all_reduce(f, op=dist.ReduceOp.MAX, group=group)
    return tuple(int(v) for v in f.tolist())


class Prefetcher:
    """Double-buffered run-ahead of the row exchange.

    The sampler is counter-based, so the rows of step t+1 are known while step t
    computes: ``launch`` enqueues sampling + exchange for one buffer on a side
    stream (the transfer hides under step t's GEMMs), ``acquire`` makes the compute
    stream wait for a filled buffer and ``release`` tells the side stream the buffer
    may be overwritten.  Give the exchange its own process group (communicator):
    collectives of one group execute in issue order, and the exchange of step t+1
    must not queue behind the gradient all-reduce of step t.

    Inside a hipGraph capture the exchange is recorded on the CAPTURING stream itself, ahead of
    the step's forward pass: forking it onto the side stream would put RCCL's own stream two forks
    deep -- capturing stream -> side stream -> communicator stream -- and on this stack (torch
    2.10, RCCL 2.26, HIP 7.0) that never returns from the capture (profiles/
    r03_rccl_graph_capture_probe.txt: one fork deep every collective captures and replays
    correctly, two deep it hangs or crashes in hipStreamEndCapture).  The overlapped form of a
    replayed step is therefore SEVERAL graphs per step (train.TrainStep, ``use_graph="split"``): the
    exchange captured with THIS stream as the capture origin, the compute step (in two parts) with
    the compute stream as origin -- each communicator stream is then one fork from its origin -- ordered by
    events recorded eagerly between the two replays.  The event hand-over between steps is not
    needed inside a single-graph replay (consecutive replays are ordered by the stream) and could
    not be captured anyway: its events belong to earlier launches."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"
        if self.cuda:
            self.stream = torch.cuda.Stream(self.device)
            self.ready = [torch.cuda.Event(), torch.cuda.Event()]
            self.free = [torch.cuda.Event(), torch.cuda.Event()]
        self._released = [False, False]
        self._cold = True

    def _capturing(self):
        return self.cuda and torch.cuda.is_current_stream_capturing()

    def launch(self, b, fill_fn):
        if not self.cuda:
            fill_fn()
            return
        cur = torch.cuda.current_stream(self.device)
        if self._capturing():
            fill_fn()                                # on the capturing stream (see the class docstring)
            return
        if self._cold:
            # first use: everything set up on the compute stream so far (table fill,
            # uploads, weight init) must be visible to the side stream
            self.stream.wait_stream(cur)
            self._cold = False
        with torch.cuda.stream(self.stream):
            if self._released[b]:
                self.stream.wait_event(self.free[b])
            fill_fn()
            self.ready[b].record(self.stream)

    def acquire(self, b):
        if self.cuda and not self._capturing():
            torch.cuda.current_stream(self.device).wait_event(self.ready[b])

    def wait_ready(self, b):
        """The compute stream (and what is issued on it next: the gradient all-reduce) waits for
        the exchange into buffer b (inside a capture it sits on the capturing stream already)."""
        if self.cuda and not self._capturing():
            torch.cuda.current_stream(self.device).wait_event(self.ready[b])

    def release(self, b):
        if self.cuda and not self._capturing():
            self.free[b].record(torch.cuda.current_stream(self.device))
            self._released[b] = True

    def drain(self):
        """Order the two streams behind each other -- before switching between eager steps and
        graph replays, and after a resume.  Both directions: the compute stream waits for what the
        side stream has been given, AND the side stream waits for the compute stream -- a replay
        records no ``free`` events (release() is a no-op under capture), so the events on hand date
        from the last eager step and the next eager ``launch`` must not overwrite a buffer that a
        replay still queued on the compute stream reads.  The stale ``free`` events are dropped."""
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            self.stream.wait_stream(cur)
            self._released = [False, False]
