"""world_size-2 and -8 gloo (CPU) tests of the multi-GPU plumbing in cdml_amd.dist: row
routing over a row-sharded catalogue (all-to-all ids -> owner gather ->
all-to-all rows -> unpermute) and the gradient average.  The owner-side gather
is injected from the oracle here (the HIP kernel needs a GPU); everything else
is the product code path the N>1 bench runs over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, F, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cdml_amd import dist as cdist
        from oracle import sampler as osampler, synth as osynth, tower as otower

        full = osynth.features_philox(0, n_rows, F, seed=3)            # the global catalogue
        lo, hi, per = cdist.shard_bounds(n_rows, world, rank)
        # a default group that is not bound to a device cannot host capture-only RCCL groups: the probe says so (and
        # TrainStep then steps eagerly with a warning instead of raising); asking for one anyway raises
        assert cdist.capture_groups_supported() is None
        try:
            cdist.new_capture_group()
            raise AssertionError("new_capture_group() on an unbound default group must raise")
        except RuntimeError as e:
            assert "device_id" in str(e)

        class Shard:                                                   # stand-in for FeatureTable
            data = torch.from_numpy(full[lo:hi].copy())
            row0, feature_size = lo, F

        def oracle_gather(table, ids, out):
            used = ids.numpy() >= 0                                    # id -1 = padding slot: untouched
            rows = table.data.numpy()[ids.numpy()[used] - table.row0]
            out[torch.from_numpy(used), :F] = torch.from_numpy(otower.l2_normalize(rows, np.float32)[0])
            out[torch.from_numpy(used), F:] = 0
            return out

        ex = cdist.RowExchange(n_rows, local_gather=oracle_gather)
        assert ex.per == per
        pairs = osynth.cowatch_pairs(n_rows, 60, 1)
        B = 16
        for step in range(3):
            # this rank's slice of the global batch (counter-based sampler: no comms)
            idx = osampler.device_triplets_vec(pairs, n_rows, 7, step, B, slot0=rank * B,
                                               batch_global=world * B).reshape(-1)
            ids = torch.from_numpy(idx.astype(np.int32))
            out = torch.full((len(idx), F + 4), -1.0)
            ex.gather(Shard, ids, out)
            want = otower.l2_normalize(full[idx], np.float32)[0]
            np.testing.assert_allclose(out[:, :F].numpy(), want, atol=1e-7)
            assert float(out[:, F:].abs().max()) == 0
            # the reverse trip (row gradients to their owners): tag every requested row with
            # (requesting rank, position, id); the owner must see exactly the requests it served
            tag = torch.zeros((len(idx), 4))
            tag[:, 0], tag[:, 1], tag[:, 2] = rank, torch.arange(len(idx)), torch.from_numpy(idx.astype(np.float32))
            got_ids, got_rows = ex.scatter_back(tag)
            assert got_ids.numel() == got_rows.shape[0] == world * ex.capacity(len(idx))
            used = got_ids >= 0                                                # unused slots carry id -1
            got_ids, got_rows = got_ids[used], got_rows[used]
            assert ((got_ids >= lo) & (got_ids < hi)).all()                    # only rows this rank owns
            np.testing.assert_array_equal(got_rows[:, 2].numpy(), got_ids.numpy().astype(np.float32))
            mine = got_rows[got_rows[:, 0] == rank]                            # own requests come back in order
            own = np.nonzero((idx >= lo) & (idx < hi))[0]
            np.testing.assert_array_equal(mine[:, 1].numpy(), own.astype(np.float32))
            n_total = torch.tensor([got_ids.numel()])
            dist.all_reduce(n_total)
            assert int(n_total) == world * len(idx)                            # nothing lost, nothing doubled
        ex.check_overflow()
        # routing: per-owner segments of `cap` slots, requests in ascending order, -1 padding
        ids = torch.tensor([per + 1, 0, per, 3, 2 * per - 1], dtype=torch.int32).clamp(max=n_rows - 1)
        send, slot = ex.route(ids, 4)
        assert send.tolist()[:8] == [0, 3, -1, -1, per + 1, per, min(2 * per - 1, n_rows - 1), -1]
        assert send.tolist()[8:] == [-1] * (4 * (world - 2))          # nothing for the other owners
        assert slot.tolist() == [4, 0, 5, 1, 6]
        # a segment that is too small raises the flag (and only then)
        ex.check_overflow()
        send, slot = ex.route(ids, 2)
        assert slot.tolist() == [2, 0, 3, 1, -1]
        with pytest.raises(RuntimeError):
            ex.check_overflow()
        ex.overflow.zero_()
        assert cdist.exchange_capacity(16384, 8) == 2840 and cdist.exchange_capacity(100, 1) == 100
        # skewed ids: every request of every rank goes to shard 0 (popularity-ordered
        # ids), more of them than one segment holds -> the requests that found no slot come back as
        # NaN rows (never as the stale row of an earlier step), the others are right, and the flag
        # raises; a capacity_factor that covers the skew serves all of them
        hot = torch.arange(400, dtype=torch.int32) % min(per, n_rows)
        tight = cdist.RowExchange(n_rows, local_gather=oracle_gather)
        cap = tight.capacity(hot.numel())
        assert cap < hot.numel()
        out = torch.full((hot.numel(), F + 4), 5.0)                      # "stale" content
        tight.gather(Shard, hot, out)
        served = torch.arange(hot.numel()) < cap
        want = otower.l2_normalize(full[hot.numpy()], np.float32)[0]
        np.testing.assert_allclose(out[served, :F].numpy(), want[served.numpy()], atol=1e-7)
        assert torch.isnan(out[~served]).all()
        with pytest.raises(RuntimeError, match="capacity_factor"):
            tight.check_overflow()
        roomy = cdist.RowExchange(n_rows, local_gather=oracle_gather, capacity_factor=float(world))
        out = torch.full((hot.numel(), F + 4), 5.0)
        roomy.gather(Shard, hot, out)
        np.testing.assert_allclose(out[:, :F].numpy(), want, atol=1e-7)
        roomy.check_overflow()
        # the input flags are collective: ONLY the last rank overflowed, every rank learns of it -- one
        # rank's NaN rows reach every rank's weights through the gradient average, so all of them must raise together
        oob = torch.zeros(1, dtype=torch.int32)
        mine = torch.tensor([1 if rank == world - 1 else 0], dtype=torch.int32)
        assert cdist.reduce_input_flags(oob, mine) == (0, 1, 0)
        assert cdist.reduce_input_flags(oob, torch.tensor([2 if rank == 0 else 0], dtype=torch.int32)) == (0, 0, 1)
        assert cdist.reduce_input_flags(torch.tensor([rank % 2], dtype=torch.int32), None) == (1, 0, 0)
        assert cdist.reduce_input_flags(oob, torch.zeros(1, dtype=torch.int32)) == (0, 0, 0)
        # gradient average
        sync = cdist.GradSync()
        g = torch.full((10,), float(rank + 1))
        sync(g)
        np.testing.assert_allclose(g.numpy(), (world + 1) / 2)
        q.put((rank, "ok"))
    except Exception as e:                                             # surface in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(world, n_rows, F):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_row_exchange_and_grad_sync_world2():
    _run(2, 101, 12)


def test_row_exchange_and_grad_sync_world8():
    """The node's form (BASELINE configs 3/4): eight ranks, uneven shards (1003 rows -> 7 x 126 + 121),
    every pair of ranks exchanging rows and row gradients, the 1/8 gradient average."""
    _run(8, 1003, 12)


def test_shard_bounds_cover_catalogue():
    sys.path.insert(0, ROOT)
    from cdml_amd import dist as cdist
    for n, w in ((10000000, 8), (101, 2), (7, 8), (1000000, 1)):
        spans = [cdist.shard_bounds(n, w, r)[:2] for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        per = cdist.shard_bounds(n, w, 0)[2]
        for r in (0, n // 2, n - 1):
            assert spans[r // per][0] <= r < spans[r // per][1]


def test_capture_origin_that_carried_an_eager_collective_is_refused():
    """The arrangement of rounds 3-4 that ended a process in round 5: the split form's exchange graph
    captured with the PREFETCH stream -- which carries the eager exchanges' work events -- as capture origin.  Every
    collective entry point of cdml_amd.dist notes the stream it was issued on; TrainStep._capture refuses such an origin
    before it touches the device (stand-in stream objects: no GPU here)."""
    from cdml_amd import dist as cdist, train

    class Stream:
        def __init__(self, handle):
            self.cuda_stream, self.device_index = handle, 0

    prefetch, own = Stream(0x7f0000001000), Stream(0x7f0000002000)
    cdist.EagerCollectiveStreams.note(prefetch)            # what an eager exchange on the prefetch stream leaves behind
    ts = object.__new__(train.TrainStep)                   # (no buffers: the check comes first)
    with pytest.raises(RuntimeError, match="hipErrorCapturedEvent"):
        ts._capture(lambda: None, origin=prefetch)
    assert cdist.EagerCollectiveStreams.carried(Stream(0x7f0000001000))      # identity is the native handle, not the object
    cdist.EagerCollectiveStreams.assert_clean_origin(own)
    assert not cdist.EagerCollectiveStreams.carried(own)
