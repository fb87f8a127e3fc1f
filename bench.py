#!/usr/bin/env python3
"""Headline benchmark: triplets/s of the full training step (sample + gather + tower fwd +
hinge loss + backward + Adam) on synthetic imitation_data-shaped input.

    python bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`, 0-based):
  N = 1 (default)   config 1: 1 M videos x 1500-d fp32 in HBM, 5000 hidden, 256-d embedding,
                    batch 4096 triplets, in-batch negatives, margin 0.8, Adam.
  N > 1             config 3: 10 M videos row-sharded over the ranks, batch 8192 triplets per
                    GPU (65 536 global at N = 8), RCCL all-to-all of sampled rows + all-reduce of
                    the gradients.  Weak scaling (per-GPU work fixed).
  --precision bf16  config 4: fp16 table + bf16 MFMA, uniform (global) negatives, batch 8192 per
                    GPU on the 10 M catalogue; --graph replays the step from a hipGraph.

Prints ONE JSON line (rank 0).  Beside the contract's fields:
  roofline       the dominant kernel (weight-gradient GEMM), timed live with events on the
                 launch stream inside the timed region; `traffic` from the committed PMC summary
  gather         the HBM-bound fused sampler+gather: 10 launches per event pair (a pair around
                 one 20-40 us launch costs 10-25 % of it), same arguments as the timed steps
  cpu_baseline   BASELINE.md section 4: the CPU restatement of the reference step (oracle/
                 tower_torch.py) on config 0's 10k x 1500 table, B = 128 and 1024, 1 and all
                 threads, fetch / train split, on this box's host cores
  like_for_like  (N = 1) the per-GPU workload of the N > 1 line on ONE GPU holding the whole
                 10 M-row catalogue, so that 8-vs-1 compares one workload
  data_learnable (N = 1) the same step on a learnable catalogue (co-watched videos share a
                 cluster): the iid imitation_data features collapse the embeddings
  comm           (N > 1) what the compute stream waits for: all-reduce and row exchange
  order          what ran before the W warm-up steps: the first ~100 ms of MFMA work after an idle spell
                 run ~2 % slower (clock ramp), so the default N = 1 line measures like_for_like and
                 data_learnable first; the other lines run 0.3 s of GEMM launches on scratch buffers
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F, H, D = 1500, 5000, 256
MARGIN = 0.8
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: matrix FP32 (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: BF16 MFMA dense (spec)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def synth_pairs(n_videos, n_users, seed, lo=0):
    """imitation_data.py:56-85-shaped co-watch pairs (own generator, product side)."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(2, 31, size=n_users)
    vids = rng.randint(0, n_videos, size=int(lens.sum()))
    last = np.zeros(len(vids), dtype=bool)
    last[np.cumsum(lens) - 1] = True
    keep = ~last[:-1] & (vids[:-1] != vids[1:])
    pairs = np.stack([vids[:-1][keep], vids[1:][keep]], axis=1)
    pairs = pairs[rng.permutation(len(pairs))]      # one global shuffle (parse_data.py:206), vectorised
    return (pairs + lo).astype(np.int32)


class KernelTimer:
    """Event pairs around individual launches on the current stream."""

    def __init__(self):
        self.ev = {}
        self.on = False

    def wrap(self, name, fn):
        def timed(*a, **k):
            if not self.on:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            self.ev.setdefault(name(*a, **k) if callable(name) else name, []).append((s, e))
            return r
        return timed

    def calibrate(self):
        """An empty start/end pair: what the pair itself adds to a measured launch (reported,
        not subtracted)."""
        pairs = []
        for _ in range(20):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        self.overhead_ms = float(np.median([s.elapsed_time(e) for s, e in pairs]))

    overhead_ms = 0.0

    def mean_ms(self, name):
        v = [s.elapsed_time(e) for s, e in self.ev.get(name, [])]
        return float(np.mean(v)) if v else None

    def count(self, name):
        return len(self.ev.get(name, []))


def pmc_traffic(kernel, bf16=False):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (profiles/latest_pmc.csv, latest_pmc_bf16.csv for --precision bf16: separate FETCH_SIZE /
    WRITE_SIZE passes of this bench, KiB; FETCH_SIZE doubled per the gfx950 correction).  None if absent."""
    path = os.path.join(ROOT, "profiles", "latest_pmc_bf16.csv" if bf16 else "latest_pmc.csv")
    if not os.path.exists(path):
        return None
    import csv
    vals = {}
    for r in csv.DictReader(open(path)):
        # `kernel` may leave trailing template arguments (tile-depth / stage tuning) open
        if r["kernel"] == kernel or (kernel.endswith(",") and r["kernel"].startswith(kernel)):
            vals[r["counter"]] = float(r["mean_per_launch"])
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0


def usable_cores():
    """Cores this process may actually use: the affinity mask and the cgroup CPU quota (a GPU
    box hands one job a share of the host; 256 torch threads on a 16-core share crawl)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(full=False):
    """the CPU restatement of the reference step (numpy fancy-index gather
    as inputs.py:158 + torch-CPU fp32 tower / hinge loss / backward / TF-form Adam,
    oracle/tower_torch.py) on config 0's inputs -- 10 000 x 1500 table of round(U[0,1), 8)
    (imitation_data.py:41-53, np.random.seed(0)), imitation_data-shaped pairs, the reference's
    uniform-negative rule -- at B = 128 (config 0) and B = 1024 (the reference's production
    batch, train.py:360), with 1 thread and with all threads, fetch vs train split like
    train.py:314-323, median over the timed steps.  Bounded: each of the four runs stops at 50
    timed steps or at its share of ~25 s, whichever comes first (--cpu-baseline-full: 50 steps
    everywhere)."""
    from oracle import synth as osynth, tower_torch
    table = osynth.features_numpy(10000, F, seed=0).astype(np.float32)
    pairs = osynth.cowatch_pairs(10000, 3000, 0)
    all_threads = usable_cores()
    prev = torch.get_num_threads()
    runs = []
    for threads, B, budget, warm in ((all_threads, 128, 3.0, 5), (all_threads, 1024, 6.0, 3),
                                     (1, 128, 8.0, 2), (1, 1024, 8.0, 1)):
        torch.set_num_threads(threads)
        st = tower_torch.CpuStep(table, pairs, B, hidden=H, out=D, margin=MARGIN, lr=0.01)
        tf, tt, n = tower_torch.time_steps(st, 50, 1e9 if full else budget, 10 if full else warm)
        runs.append({"batch": B, "threads": threads, "timed_steps": n, "fetch_ms": round(tf * 1e3, 3),
                     "train_ms": round(tt * 1e3, 3), "triplets_per_s": round(B / (tf + tt), 1)})
    torch.set_num_threads(prev)
    head = runs[1]                                   # production batch on all threads
    return {"value": head["triplets_per_s"], "unit": "triplets/s", "cores": all_threads, "kind": "port",
            "cpu": cpu_model(), "host_cpu_count": os.cpu_count(),
            "sample": "CPU restatement of the TF1 path (oracle/tower_torch.py: numpy gather + torch-CPU fp32 "
                      "tower/loss/backward/Adam), config-0 table 10000x1500, uniform negatives; value = B=1024 on "
                      "%d threads, median of %d timed steps (fetch %.1f ms + train %.1f ms); all four runs in `runs`"
                      % (all_threads, head["timed_steps"], head["fetch_ms"], head["train_ms"]),
            "runs": runs}


def learnable_catalogue(n_rows, dev, n_clusters=2000, seed=0):
    """tools/train_demo.py's catalogue: co-watched videos share a cluster, so there is something
    to learn (loss 0.8 -> ~0) and the embeddings do not collapse."""
    from cdml_amd import engine
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    centers = torch.rand(n_clusters, F, device=dev, generator=g)
    cluster = torch.randint(0, n_clusters, (n_rows,), device=dev, generator=g)
    table = engine.FeatureTable(torch.zeros((n_rows, engine.FeatureTable.padded_stride(F)), device=dev), F)
    table.data[:, :F] = (centers[cluster] + 0.35 * torch.randn(n_rows, F, device=dev, generator=g)).clamp_(0.0, 1.0)
    order = torch.argsort(cluster)
    a, b = order[:-1], order[1:]
    same = cluster[a] == cluster[b]
    pairs = torch.stack([a[same], b[same]], 1).to(torch.int32)
    return table, pairs[torch.randperm(pairs.shape[0], device=dev, generator=g)].contiguous()


def settle_gpu(dev, seconds=0.3):
    """Weight-gradient GEMM launches on scratch buffers for ~`seconds`: the first ~100 ms of MFMA work after an
    idle spell run ~2 % slower on this part (clock ramp), and the default N = 1 line gets past that through
    its secondary measurements; every other line gets the same start through this loop.  Not a step of the
    measured job: no state of the TrainStep is touched."""
    from cdml_amd import ops
    M, K, N = 4096, 1536, 5120
    # the split-K weight-gradient entry point: an MFMA-bound launch that is NOT part of the N = 1 step
    # (stream-K there), so a rocprofv3 run of this bench keeps clean per-kernel averages for the step
    x = torch.rand((M, K), device=dev)
    dy = torch.randn((M, N), device=dev) * 0.02
    dW = torch.empty((K, N), device=dev)
    db = torch.empty(N, device=dev)
    ws = torch.empty(max(ops.fc_bwd_weight_workspace(M, K, N), 16) // 4, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            ops.fc_bwd_weight(x, dy, dW, db, ws, M, K, N)
        torch.cuda.synchronize(dev)


def timed_steps(ts, steps, warmup, dev):
    for _ in range(warmup):
        ts.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=None, help="catalogue rows (default: 1M at N=1 fp32, else 10M)")
    ap.add_argument("--mode", default=None, choices=["inbatch", "uniform", "semihard"])
    ap.add_argument("--batch", type=int, default=None, help="triplets per GPU per step (default 4096 for config 1, else 8192)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="bf16 = BASELINE config 4 path (fp16 table + bf16 MFMA); not the headline metric")
    ap.add_argument("--train-table", action="store_true",
                    help="also train the catalogue rows (lazy Adam; build-defined, not the headline metric)")
    ap.add_argument("--gather-ahead", type=int, default=4, help="steps fetched per sampler+gather launch (1 GPU, fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="50 timed CPU steps in all four runs (minutes)")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip like_for_like and data_learnable")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the 0.3 s GEMM loop before the warm-up steps (rocprof runs: keeps its launches out of the kernel averages)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback)")
    backend = os.environ.get("CDML_DIST_BACKEND", "nccl")     # "gloo": 1-GPU-box rehearsal of N>1
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from cdml_amd import dist as cdist, engine, engine_bf16, ops, train
    bf16 = args.precision == "bf16"
    Table = engine_bf16.FeatureTableF16 if bf16 else engine.FeatureTable

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a finite timeout: a stuck collective ends the run with a non-zero exit instead of hanging
        tmo = datetime.timedelta(seconds=int(os.environ.get("CDML_DIST_TIMEOUT_S", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    config1 = world == 1 and not bf16
    n_rows = args.rows or (1000000 if config1 else 10000000)
    B = args.batch or (4096 if config1 else 8192)
    mode = args.mode or ("uniform" if bf16 else "inbatch")
    phase = "setup"
    try:
        if world > 1:
            lo, hi, _ = cdist.shard_bounds(n_rows, world, rank)
            table = Table.synthetic(hi - lo, F, seed=0, device=dev, row0=lo, n_rows_global=n_rows)
            # own communicator for the exchange so it never queues behind the gradient all-reduce
            ex_group = dist.new_group()
            exchange, grad_sync = cdist.RowExchange(n_rows, group=ex_group), cdist.GradSync(device=dev)
            # bring both communicators up here (RCCL builds them on first use), not inside a step
            phase = "communicator warm-up"
            warm = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(warm)
            dist.all_reduce(warm, group=ex_group)
        else:
            table = Table.synthetic(n_rows, F, seed=0, device=dev)
            exchange = grad_sync = None
        # users: a third of the catalogue, capped -- the run consumes < 1 M pairs per rank, and every
        # rank builds the same list on its host while the others wait
        pairs = torch.from_numpy(synth_pairs(n_rows, max(min(n_rows // 3, 600000), 1000), seed=0)).to(dev)

        ts = train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode,
                             optimizer="adam", base_learning_rate=0.01, seed=1234, weight_seed=42,
                             device=dev, exchange=exchange, grad_sync=grad_sync, slot0=rank * B,
                             batch_global=world * B, use_graph=args.graph, precision=args.precision,
                             train_table=args.train_table, gather_ahead=args.gather_ahead)
        L = ts.layout

        # per-kernel event timers on the launch stream (off during graph replay)
        kt = KernelTimer()
        timers_on = not args.no_kernel_timers
        graph_run = ts.use_graph                        # timed region replays the graph; the per-kernel
                                                        # event timers then run on extra EAGER steps after it
        real = {}
        if timers_on:
            def patch(name, label):
                real[name] = getattr(ops, name)
                setattr(ops, name, kt.wrap(label, real[name]))
            if bf16:
                # gemm_bf16_nt(epilogue, A, B, C, M, N, K, ...): FC1 has N = Hp, K = Fp; FC2 N = Dp; dH1 K = Dp
                patch("gemm_bf16_nt", lambda e, A, Bm, C, M, N, K, **k:
                      "fc1_fwd" if (N == L.Hp and K == L.Fp) else ("fc2_fwd" if N == L.Dp else "dH1"))
                # gemm_bf16_tn(A, B, C, M, N, K, ...): dW1 has N = Hp
                patch("gemm_bf16_tn", lambda A, Bm, C, M, N, K, **k: "dW1" if N == L.Hp else "dW2")
            else:
                patch("fc_lrelu_fwd", lambda x, W, b, y, M, K, N, *a, **k: "fc1_fwd" if N == L.Hp else "fc2_fwd")
                patch("fc_bwd_weight", lambda x, dy, dW, db, ws, M, K, N: "dW1" if N == L.Hp else "dW2")
                patch("fc_bwd_weight2", "dW")             # single GPU: both products in one stream-K launch
                patch("fc_bwd_data", "dH1")
            if bf16:                                      # matrices (with their bf16 operand copies), then the biases
                patch("adam_matrix_bf16", lambda W, *a, **k: "adam_w1" if W.shape[0] == L.Fp else "adam_w2")
                patch("adam_step", "adam_bias")
            else:
                patch("adam_step", "adam")
            patch("vnet_tail", "tail")
        comm = {}
        if world > 1:                                   # what the compute stream waits for
            real_finish = grad_sync.finish
            grad_sync.finish = kt.wrap("allreduce_wait", real_finish)
            if ts.prefetch is not None:
                ts.prefetch.acquire = kt.wrap("exchange_wait", ts.prefetch.acquire)

        def sync_all():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        extras = {}
        # The two secondary measurements run BEFORE the headline: the first ~100 ms of MFMA work after an
        # idle spell run ~2 % slower (measured: --warmup 5 -> 2.51 ms/step, --warmup 100 -> 2.478), so the
        # headline's W warm-up + K timed steps follow them back to back, at the clocks a real run sees.
        if config1 and mode == "inbatch" and not args.no_extras and not args.train_table and rank == 0:
            phase = "secondary measurements"
            try:                                        # their failure must not cost the headline
                n_s, n_w = min(args.steps, 30), min(max(args.warmup, 3), 5)
                # (b) the N > 1 line's per-GPU workload, whole 10 M-row catalogue on this one GPU
                t10 = Table.synthetic(10000000, F, seed=0, device=dev)
                p10 = torch.from_numpy(synth_pairs(10000000, 600000, seed=0)).to(dev)
                ts10 = train.TrainStep(t10, p10, 8192, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch",
                                       optimizer="adam", base_learning_rate=0.01, device=dev, gather_ahead=args.gather_ahead)
                el = timed_steps(ts10, n_s, n_w, dev)
                extras["like_for_like"] = {"workload": "config3 per-GPU batch on ONE GPU: 10000000 videos x 1500-d fp32 (61 GB) in "
                                                    "HBM, batch 8192 triplets, in-batch negatives (the N>1 lines run this per GPU "
                                                    "on a row-sharded catalogue)",
                                        "value": round(8192 * n_s / el, 1), "unit": "triplets/s",
                                        "ms_per_step": round(el / n_s * 1e3, 4), "steps": n_s, "warmup": n_w}
                del ts10, t10, p10
                torch.cuda.empty_cache()
                # (c) the headline step on a learnable catalogue
                tl, pl = learnable_catalogue(200000, dev)
                tsl = train.TrainStep(tl, pl, B, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch",
                                      optimizer="adam", base_learning_rate=2e-4, device=dev, gather_ahead=args.gather_ahead)
                tsl.step()
                l0 = tsl.loss()
                n_l = max(n_s, 60)
                el = timed_steps(tsl, n_l, n_w, dev)
                extras["data_learnable"] = {"workload": "config1 step (batch %d in-batch) on a learnable 200000 x 1500 catalogue "
                                                     "(co-watched videos share one of 2000 clusters), Adam 2e-4" % B,
                                         "value": round(B * n_l / el, 1), "unit": "triplets/s",
                                         "ms_per_step": round(el / n_l * 1e3, 4), "steps": n_l,
                                         "loss_first_step": round(l0, 4), "loss_last_step": round(tsl.loss(), 4)}
                del tsl, tl, pl
                torch.cuda.empty_cache()
            except Exception as e:                      # noqa: BLE001
                extras.setdefault("secondary_error", repr(e)[:300])
                ts10 = t10 = p10 = tsl = tl = pl = None
                torch.cuda.empty_cache()
        if not (extras.get("like_for_like") or extras.get("data_learnable")) and not args.no_settle:
            phase = "clock settle"
            settle_gpu(dev)
        phase = "warm-up steps"
        for _ in range(args.warmup):
            ts.step()
        sync_all()
        phase = "timed steps"
        # kernel timers on every 4th timed step (every 8th in long runs): an event pair costs the
        # stream ~5 us, seven pairs a step are 1.5 % of it; the sampled launches are still launches
        # of the timed region
        stride = 4 if args.steps < 100 else 8
        t0 = time.perf_counter()
        for i in range(args.steps):
            kt.on = timers_on and not graph_run and (i % stride == 0)
            ts.step()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        kt.on = False
        sampled = (args.steps + stride - 1) // stride
        if graph_run and timers_on:                     # same kernels, launched eagerly, for the roofline
            ts.use_graph = False
            sampled = 8
            for i in range(2 * sampled):
                kt.on = i % 2 == 0
                ts.step()
            torch.cuda.synchronize(dev)
            kt.on = False
            ts.use_graph = True
        for name, fn in real.items():
            setattr(ops, name, fn)
        if timers_on:
            kt.calibrate()
        if world > 1:
            phase = "max over ranks"
            elapsed = reduce_max(elapsed)
        loss = ts.loss()
        assert np.isfinite(loss), "non-finite loss"
        if exchange is not None:
            exchange.check_overflow()
    except Exception as e:                               # noqa: BLE001 -- name the phase, exit non-zero
        sys.stderr.write("[bench rank %d] failed during %s: %r\n" % (rank, phase, e))
        raise

    if rank == 0:
        rpt = ts.rows_per_triplet
        R = B * rpt
        ms = elapsed / args.steps * 1e3
        if bf16:
            cfg_name = "config4" + (" per-GPU shape on 1 GPU" if world == 1 else "")
        elif world > 1:
            cfg_name = "config3"
        else:
            cfg_name = {"inbatch": "config1", "semihard": "config2", "uniform": "config1 size, reference negative rule"}[mode]
        out = {
            "metric": "triplets/sec", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if not bf16 else "bf16 (fp16 table, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s: %d videos x %d-d %s in HBM%s, %d hidden, %d-d embed, batch %d triplets/GPU "
                                   "(%d global), %s negatives, margin %.1f, Adam, full step (sample+gather+fwd+loss+bwd+opt)"
                                   % (cfg_name, n_rows, F, "fp16" if bf16 else "fp32",
                                      "" if world == 1 else " row-sharded over %d GPUs" % world, H, D, B, world * B,
                                      {"inbatch": "in-batch", "uniform": "uniform random", "semihard": "semi-hard mined"}[mode],
                                      MARGIN),
                       "global_batch": world * B, "rows_per_triplet": rpt,
                       "parallelism": "dp%d" % world + ("" if world == 1 else " row-sharded table, all-to-all rows + all-reduce grads"),
                       "hipgraph": bool(ts.use_graph), "trainable_table": bool(args.train_table),
                       "gather_steps_per_launch": ts.gather_ahead},
            "loss": round(loss, 6),
        }
        # dominant kernel by total time: the weight-gradient GEMM (dW1: 2*R*F*H flop, dW2: 2*R*H*D
        # flop, one kernel symbol); rocprof's per-kernel average is over all its launches, so the
        # roofline is too.  Algorithmic (unpadded) flop.
        flops_gemm = 2.0 * R * F * H
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        kname = "k_gemm_bf16_256<true, 3>" if bf16 else "k_gemm_f32<false, false, 2, 2, 3,"
        have_dw = kt.count("dW") or (kt.count("dW1") and kt.count("dW2"))
        if timers_on and have_dw and kt.count("fc1_fwd"):
            if kt.count("dW"):                          # stream-K: dW1 and dW2 (+ the fix-up pass) in one call
                n_launch, t_ms = kt.count("dW"), kt.mean_ms("dW")
                kname, klabel = "k_gemm_f32_sk", "k_gemm_f32_sk (dW1+dW2 in one stream-K launch, fix-up pass included)"
            else:
                n_launch = kt.count("dW1") + kt.count("dW2")
                t_ms = (kt.mean_ms("dW1") * kt.count("dW1") + kt.mean_ms("dW2") * kt.count("dW2")) / n_launch
                klabel = kname + (" ...> (dW1+dW2 launches)" if not bf16 else " (dW1+dW2 launches)")
            flop_launch = sampled * (2.0 * R * F * H + 2.0 * R * H * D) / n_launch
            ach = flop_launch / (t_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": klabel,
                               "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach / peak, 4),
                               "traffic": pmc_traffic(kname, bf16) if world == 1 else None,
                               "launch_ms": round(t_ms, 4), "flop_per_launch": flop_launch,
                               "launches_per_step": n_launch / sampled, "timed_steps": sampled,
                               "timed_how": ("event pairs on every %dth timed step" % stride) if not graph_run else
                                            "event pairs on 8 eager steps after the graph-replayed timed region"}
            ach1 = flops_gemm / (kt.mean_ms("fc1_fwd") * 1e-3) / 1e12
            k1 = "k_gemm_bf16_256<false, 0>" if bf16 else "k_gemm_f32<true, false, 2, 2, 1,"
            out["roofline_fc1_fwd"] = {"bound": "mfma", "kernel": k1 + ("" if bf16 else " ...>"),
                                       "achieved": round(ach1, 2), "peak": peak, "unit": "TFLOP/s",
                                       "frac": round(ach1 / peak, 4), "traffic": pmc_traffic(k1, bf16) if world == 1 else None,
                                       "launch_ms": round(kt.mean_ms("fc1_fwd"), 4), "flop_per_launch": flops_gemm}
            kern = {}
            for k in ("fc1_fwd", "fc2_fwd", "tail", "dH1", "dW", "dW1", "dW2", "adam", "adam_w1", "adam_w2", "adam_bias"):
                if kt.mean_ms(k) is not None:
                    kern[k + "_ms"] = round(kt.mean_ms(k), 4)
            kern["empty_event_pair_ms"] = round(kt.overhead_ms, 5)      # included in the figures above
            out["kernels"] = kern
        # the HBM-bound kernel the metric also names: the fused sampler+gather (fp32, 1 GPU) or
        # the fp16 row gather (config 4), 10 back-to-back launches per event pair
        if world == 1 and not args.train_table:
            reps, per = 5, 10
            n_st = ts.gather_ahead
            m = train._MODES[mode]
            if bf16:
                gk = "k_sample_gather<%d," % (1 if rpt == 2 else 0)       # <mode, RowF16<3>>
                gbytes = n_st * R * F * (2 + 2.0)          # fp16 rows read + bf16 normalised rows written
            else:
                gk = "k_sample_gather<%d," % (1 if rpt == 2 else 0)       # <mode, RowF32<6>>
                gbytes = n_st * 2.0 * R * F * 4             # rows read + normalised rows written
            nxt = [ts.global_step + 1000]                   # fresh steps every launch: re-reading the same rows
                                                            # would be served from the 256 MB Infinity Cache

            def launch():                                   # (training on `ts` is over: its buffers are scratch now)
                if n_st > 1:
                    ts._gather_block(nxt[0])
                else:
                    ops.sample_gather(m, ts.pairs, ts.seed, nxt[0], B, ts.table.data, F, ts.idx, ts.ws.x_hat,
                                      shift_out=ts.shift)
                nxt[0] += n_st
            for _ in range(3):
                launch()
            evs = []
            for _ in range(reps):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(per):
                    launch()
                e.record()
                evs.append((s, e))
            torch.cuda.synchronize(dev)
            t_g = float(np.median([s.elapsed_time(e) for s, e in evs])) / per
            g_ach = gbytes / (t_g * 1e-3) / 1e9
            out["gather"] = {"bound": "hbm", "kernel": gk + (" RowF16<3>>" if bf16 else " RowF32<6>>"), "achieved": round(g_ach, 1), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round(g_ach / PEAK_HBM_GBS, 4), "traffic": pmc_traffic(gk, bf16),
                             "bytes_per_launch": gbytes, "launch_ms": round(t_g, 4), "steps_per_launch": n_st,
                             "method": "%d back-to-back launches per event pair (inter-launch gaps included), median of %d"
                                       % (per, reps)}
        if world > 1:
            ar, exw = kt.mean_ms("allreduce_wait"), kt.mean_ms("exchange_wait")
            out["comm"] = {"allreduce_exposed_ms": None if ar is None else round(ar, 4),
                           "exchange_exposed_ms": None if exw is None else round(exw, 4),
                           "exchange_bytes": exchange.bytes_per_step(ts.R, ts.ws.x_hat),
                           "allreduce_bytes": int(ts.layout.numel * 4),
                           "note": "event pairs around GradSync.finish / Prefetcher.acquire on the compute stream "
                                   "(on the sampled timed steps): what the step waits for, not the collectives' own duration"}
        step_flops = R * (2.0 * F * H + 2.0 * H * D) + R * (2.0 * F * H + 4.0 * H * D)
        out["step_tflops"] = round(step_flops / (elapsed / args.steps) / 1e12, 2)

        out.update(extras)
        warmed = bool(extras.get("like_for_like") or extras.get("data_learnable"))
        if not warmed and args.no_settle:
            out["order"] = "warm-up + timed steps from a cold start (--no-settle)"
        elif not warmed:
            out["order"] = ("0.3 s of weight-gradient GEMM launches on scratch buffers (clock ramp after idle: the first ~100 ms "
                            "of MFMA work run ~2 % slower), then warm-up + timed steps")
        else:
            out["order"] = ("like_for_like and data_learnable were measured first, the headline's warm-up + timed steps "
                            "directly after them (GPU already at its sustained clocks), cpu_baseline last")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_full)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
