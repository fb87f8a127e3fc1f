#!/usr/bin/env python3
"""Headline benchmark: triplets/s of the full training step (sample + gather +
tower fwd + hinge loss + backward + Adam) on synthetic imitation_data-shaped
input, BASELINE.json config 1 at N=1:

    1 x MI355X, 1M videos x 1500-d fp32 in HBM, 5000 hidden, 256-d embedding,
    batch 4096 triplets, in-batch negatives, margin 0.8, Adam.

    python bench.py --gpus N --steps K --warmup W

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling --
every rank keeps batch 4096; the catalogue becomes config 3's 10M rows,
row-sharded, with an RCCL all-to-all of sampled rows and an all-reduce of the
gradients per step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the
1500x5000 projection GEMMs, MFMA-bound), timed live with events on the launch
stream over the timed steps; `gather` reports the HBM side the metric also
names; `cpu_baseline` is the CPU oracle (numpy restatement of the reference
step) timed on this box's host cores on a bounded sample.
"""
import argparse
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
BATCH = 4096
MARGIN = 0.8
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: matrix FP32 (spec)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def synth_pairs(n_videos, n_users, seed):
    """imitation_data.py:56-85-shaped co-watch pairs (own generator, product side)."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(2, 31, size=n_users)
    vids = rng.randint(0, n_videos, size=int(lens.sum()))
    last = np.zeros(len(vids), dtype=bool)
    last[np.cumsum(lens) - 1] = True
    keep = ~last[:-1] & (vids[:-1] != vids[1:])
    pairs = np.stack([vids[:-1][keep], vids[1:][keep]], axis=1)
    pairs = pairs[rng.permutation(len(pairs))]      # one global shuffle (parse_data.py:206), vectorised
    return pairs.astype(np.int32)


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
            self.ev.setdefault(name, []).append((s, e))
            return r
        return timed

    def calibrate(self):
        """Time an empty start/end pair: an upper bound of what the pair itself adds to a measured
        launch (a few us: it matters for the 20 us gather, not for the 1 ms GEMMs).  Reported, not
        subtracted -- subtracting it over-corrects (rocprofv3's per-kernel averages say so)."""
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


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC summary
    (profiles/latest_pmc.csv: separate FETCH_SIZE / WRITE_SIZE passes of this
    bench, KiB; FETCH_SIZE doubled per the gfx950 correction).  None if absent."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.csv")
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


def cpu_baseline(budget_s=12.0):
    """The oracle's restatement of one reference training step (numpy gather as
    inputs.py:158 + fp32 tower fwd/bwd + Adam), timed on the host cores."""
    from oracle import sampler as osampler, tower as otower
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n_rows, B = 50000, 256
    rng = np.random.RandomState(0)
    table = rng.random_sample((n_rows, F)).astype(np.float32)
    pairs = synth_pairs(n_rows, 5000, 0)
    W = [otower.xavier_uniform(rng, F, H), np.zeros(H, np.float32),
         otower.xavier_uniform(rng, H, D), np.zeros(D, np.float32)]
    m = [np.zeros_like(w) for w in W]
    v = [np.zeros_like(w) for w in W]

    def one_step(step):
        rows, tri, valid, _ = osampler.device_inbatch(pairs, 1234, step, B)
        x = osampler.gather(table, rows)
        fwd = otower.vnet_forward(x, *W, dtype=np.float32)
        otower.hinge_loss_indexed(fwd["l2_norm"], tri, valid.astype(bool), MARGIN, np.float32)
        dE = otower.hinge_loss_indexed_backward(fwd["l2_norm"], tri, valid.astype(bool), MARGIN, np.float32)
        g = otower.vnet_backward(fwd, W[2], dE, np.float32)
        for i, k in enumerate(("dW1", "db1", "dW2", "db2")):
            W[i], m[i], v[i] = otower.adam_step(W[i], g[k], m[i], v[i], step + 1, 0.01)

    one_step(0)
    t0, n = time.time(), 0
    while time.time() - t0 < budget_s:
        one_step(n + 1)
        n += 1
    dt = time.time() - t0
    return {"value": round(n * B / dt, 1), "unit": "triplets/s", "cores": int(threads), "kind": "port",
            "sample": "%d oracle steps (numpy fp32 restatement of the reference step incl. gather "
                      "and Adam) of %d in-batch triplets on a %dx%d host table, %.1f s"
                      % (n, B, n_rows, F, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=None, help="catalogue rows (default: 1M, or 10M when --gpus > 1)")
    ap.add_argument("--mode", default="inbatch", choices=["inbatch", "uniform", "semihard"])
    ap.add_argument("--batch", type=int, default=BATCH, help="triplets per GPU per step")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (1 GPU)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16"],
                    help="bf16 = BASELINE config 4 path (fp16 table + bf16 MFMA); not the headline metric")
    ap.add_argument("--train-table", action="store_true",
                    help="also train the catalogue rows (lazy Adam; build-defined, not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
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
    Table = engine_bf16.FeatureTableF16 if args.precision == "bf16" else engine.FeatureTable

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_rows = args.rows or (1000000 if world == 1 else 10000000)
    B = args.batch
    if world > 1:
        lo, hi, _ = cdist.shard_bounds(n_rows, world, rank)
        table = Table.synthetic(hi - lo, F, seed=0, device=dev, row0=lo, n_rows_global=n_rows)
        # own communicator for the exchange so it never queues behind the gradient all-reduce
        ex_group = dist.new_group()
        exchange, grad_sync = cdist.RowExchange(n_rows, group=ex_group), cdist.GradSync(device=dev)
        # bring both communicators up here (RCCL builds them on first use), not inside a step
        warm = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(warm)
        dist.all_reduce(warm, group=ex_group)
    else:
        table = Table.synthetic(n_rows, F, seed=0, device=dev)
        exchange = grad_sync = None
    # users: a third of the catalogue, capped -- the run consumes < 1 M pairs per rank, and every
    # rank builds the same list on its host while the others wait
    pairs = torch.from_numpy(synth_pairs(n_rows, max(min(n_rows // 3, 600000), 1000), seed=0)).to(dev)

    ts = train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=args.mode,
                         optimizer="adam", base_learning_rate=0.01, seed=1234, weight_seed=42,
                         device=dev, exchange=exchange, grad_sync=grad_sync, slot0=rank * B,
                         batch_global=world * B, use_graph=args.graph, precision=args.precision,
                         train_table=args.train_table)

    # per-kernel event timers on the launch stream (off during graph replay)
    kt = KernelTimer()
    timers_on = not args.no_kernel_timers and not args.graph and args.precision == "f32"
    if timers_on:
        real_fwd, real_bww = ops.fc_lrelu_fwd, ops.fc_bwd_weight
        L = ts.layout

        def fwd(x, W, b, y, M, K, N, *a, **k):
            name = "fc1_fwd" if K == L.Fp else "fc2_fwd"
            return kt.wrap(name, real_fwd)(x, W, b, y, M, K, N, *a, **k)

        def bww(x, dy, dW, db, ws, M, K, N):
            name = "dW1" if N == L.Hp else "dW2"         # by layer (data-parallel runs do dW1 in row blocks)
            return kt.wrap(name, real_bww)(x, dy, dW, db, ws, M, K, N)
        ops.fc_lrelu_fwd, ops.fc_bwd_weight = fwd, bww
        ts.fetch = kt.wrap("fetch", ts.fetch)

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        ts.step()
    sync_all()
    # kernel timers on every 4th timed step: an event pair costs the stream a few us, ten pairs a
    # step are ~1 % of it; the sampled launches are still launches of the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        kt.on = timers_on and (i % 4 == 0)
        ts.step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kt.on = False
    if timers_on:
        kt.calibrate()
    if world > 1:
        elapsed = reduce_max(elapsed)
    loss = ts.loss()
    assert np.isfinite(loss), "non-finite loss"

    if rank == 0:
        rpt = ts.rows_per_triplet
        R = B * rpt
        ms = elapsed / args.steps * 1e3
        if args.precision == "bf16":
            cfg_name = "config4" + (" (1 GPU)" if world == 1 else "")
        elif world > 1:
            cfg_name = "config3 (weak-scaled)"
        else:
            cfg_name = {"inbatch": "config1", "semihard": "config2", "uniform": "config1 size, reference negative rule"}[args.mode]
        out = {
            "metric": "triplets/sec", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "f32" else "bf16 (fp16 table, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s: %d videos x %d-d %s in HBM, %d hidden, %d-d embed, batch %d triplets/GPU, "
                                   "%s negatives, margin %.1f, Adam, full step (sample+gather+fwd+loss+bwd+opt)"
                                   % (cfg_name, n_rows, F, "fp16" if args.precision == "bf16" else "fp32", H, D, B,
                                      {"inbatch": "in-batch", "uniform": "uniform random",
                                       "semihard": "semi-hard mined"}[args.mode], MARGIN),
                       "global_batch": world * B, "rows_per_triplet": rpt,
                       "parallelism": "dp%d" % world + ("" if world == 1 else " row-sharded table, all-to-all rows + all-reduce grads"),
                       "hipgraph": bool(args.graph), "trainable_table": bool(args.train_table)},
            "loss": round(loss, 6),
        }
        # dominant kernel by total time: the bwd-weight GEMM k_gemm_f32<false, false, 2, 2, 3>,
        # launched twice per step (dW1: 2*R*F*H flop, dW2: 2*R*H*D flop); rocprof's per-kernel
        # average is over both launches, so the roofline is too.  Algorithmic (unpadded) flop.
        flops_gemm = 2.0 * R * F * H
        if timers_on and kt.count("dW1") and kt.count("dW2") and kt.count("fc1_fwd"):
            # mean over ALL launches of the kernel in the timed steps, like rocprof's average
            n_launch = kt.count("dW1") + kt.count("dW2")
            t_ms = (kt.mean_ms("dW1") * kt.count("dW1") + kt.mean_ms("dW2") * kt.count("dW2")) / n_launch
            timed_steps = (args.steps + 3) // 4
            flop_launch = timed_steps * (2.0 * R * F * H + 2.0 * R * H * D) / n_launch
            ach = flop_launch / (t_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_gemm_f32<false, false, 2, 2, 3, ...> (dW1+dW2 launches)",
                               "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                               "traffic": pmc_traffic("k_gemm_f32<false, false, 2, 2, 3,") if world == 1 else None,
                               "launch_ms": round(t_ms, 4), "flop_per_launch": flop_launch,
                               "launches_per_step": n_launch / timed_steps, "timed_steps": timed_steps}
            ach1 = flops_gemm / (kt.mean_ms("fc1_fwd") * 1e-3) / 1e12
            out["roofline_fc1_fwd"] = {"bound": "mfma", "kernel": "k_gemm_f32<true, false, 2, 2, 1, ...>",
                                       "achieved": round(ach1, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(ach1 / PEAK_F32_MFMA_TFLOPS, 4),
                                       "traffic": pmc_traffic("k_gemm_f32<true, false, 2, 2, 1,"),
                                       "launch_ms": round(kt.mean_ms("fc1_fwd"), 4),
                                       "flop_per_launch": flops_gemm}
            kern = {}
            for k in ("fc1_fwd", "fc2_fwd", "dW1", "dW2", "fetch"):
                if kt.mean_ms(k) is not None:
                    kern[k + "_ms"] = round(kt.mean_ms(k), 4)
            kern["empty_event_pair_ms"] = round(kt.overhead_ms, 5)      # included in the figures above
            out["kernels"] = kern
            t_f = kt.mean_ms("fetch")
            if t_f and world == 1:
                gbytes = 2.0 * R * F * 4                     # rows read + normalised rows written
                g_ach = gbytes / (t_f * 1e-3) / 1e9
                out["gather"] = {"bound": "hbm", "kernel": "k_sample_gather", "achieved": round(g_ach, 1),
                                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(g_ach / PEAK_HBM_GBS, 4),
                                 "traffic": pmc_traffic("k_sample_gather<1, 6>" if rpt == 2 else "k_sample_gather<0, 6>"),
                                 "bytes_per_launch": gbytes, "launch_ms": round(t_f, 4),
                                 # a 20-us kernel between two events: the launch time above includes
                                 # the pair's own cost (at most kernels.empty_event_pair_ms), so this
                                 # GB/s is a lower bound; rocprofv3's per-kernel average in profiles/
                                 # is the steadier figure
                                 "note": "event-pair cost included; see profiles/*kernel_stats.csv"}
        step_flops = R * (2.0 * F * H + 2.0 * H * D) + R * (2.0 * F * H + 4.0 * H * D)
        out["step_tflops"] = round(step_flops / (elapsed / args.steps) / 1e12, 2)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
