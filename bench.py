#!/usr/bin/env python3
"""Headline benchmark: triplets/s of the full training step (sample + gather + tower fwd +
hinge loss + backward + Adam) on synthetic imitation_data-shaped input.

    python bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`, 0-based):
  N = 1 (default)   the 1-GPU point of the scaling curve north_star quotes ("triplets/s at 8 GPUs vs 1 GPU on a
                    10M-video / 1500-d synthetic catalogue"): config 3's per-GPU workload on ONE GPU holding the whole
                    catalogue -- 10 M videos x 1500-d fp32 (61 GB) in HBM, 5000 hidden, 256-d embedding, batch 8192
                    triplets, in-batch negatives, margin 0.8, Adam -- so value(N) / value(1) of the bare driver commands
                    compares one workload (since round 5; rounds 1-4 timed config 1 here, which stays in the line as
                    the full `config1` record: 1 M videos, batch 4096).  --rows 1000000 --batch 4096 = config 1 as
                    the job.
  --precision       how the step's fp32 projection products are computed.  "f32x3" (the default since round 4,
                    by the round-3 ruling): every fp32 operand held as three exact bf16 planes hi | mid | lo
                    (hi + mid + lo == the fp32 value), every fp32 product as six bf16 plane products on the bf16
                    MFMA with fp32 accumulation -- fp32 results (held to the fp32 bounds by the parity tests), the
                    fp32-equivalent flop rate against the bf16 dense peak / 6.  "f32": the same step on the fp32
                    MFMA (v_mfma_f32_32x32x2_f32); the default line carries it as the `f32_mfma` record, measured
                    in the same run on the same table, so either reading has a measured number.
  N > 1             config 3: 10 M videos row-sharded over the ranks, batch 8192 triplets per
                    GPU (65 536 global at N = 8), RCCL all-to-all of sampled rows + all-reduce of
                    the gradients.  Weak scaling (per-GPU work fixed).  Started WITHOUT a launcher
                    (`python bench.py --gpus N`, no RANK in the environment) the command starts its
                    own N ranks -- before anything touches the GPU -- and relays rank 0's line;
                    under `python -m torch.distributed.run` it is one of the ranks.
  --precision bf16  config 4: fp16 table + bf16 MFMA, uniform (global) negatives, batch 8192 per
                    GPU on the 10 M catalogue; --graph replays the step from a hipGraph.
  --mode predict    catalogue inference (predict.py:71-96): forward-only tower over the 10 M-row
                    table in chunks, rows/s, fp32 and (--precision bf16) config-4 precision.

Prints ONE JSON line (rank 0).  Beside the contract's fields:
  roofline       the dominant kernel (weight-gradient GEMM), timed live with events on the
                 launch stream inside the timed region; `traffic` from the committed PMC summary
                 named in `traffic_source` (a builder-run rocprofv3 --pmc pass, not this run)
  gather         the HBM-bound fused sampler+gather: 10 launches per event pair (a pair around
                 one 20-40 us launch costs 10-25 % of it), same arguments as the timed steps
  cpu_baseline   BASELINE.md section 4: the CPU restatement of the reference step (oracle/
                 tower_torch.py) on config 0's 10k x 1500 table, B = 128 and 1024, 1 and all
                 threads, fetch / train split, on this box's host cores
  comm           (N > 1) what the compute stream waits for: all-reduce and row exchange; which
                 gradient-sync form ran (`grad_sync`: both forms are timed for a few steps before
                 the warm-up and the faster one is kept)
  order / warmup_effective   what ran on the GPU before the timed region (N = 1: a 0.3-s clock-settle loop of steps of a
                 SECOND TrainStep of the same shape on its own buffers -- the measured job's state is untouched -- then the
                 W warm-up steps; DESIGN.md section 8 says why bare GEMM launches were not enough)
Secondary records of the default N = 1 line (measured AFTER the headline; each in its own try; those that
train run at the headline's precision):
  config1            BASELINE config 1 (1 M videos, batch 4096, in-batch negatives): the headline of rounds 1-4, a full
                     record (value, roofline, FC1 roofline, kernel breakdown, gather) so the round-to-round series continues
  config2_semihard   BASELINE config 2 (the same 1 M catalogue, semi-hard mining over all pairs of the batch, batch 8192)
                     with its kernel breakdown
  f32_mfma           config 1 on the fp32 MFMA (precision "f32": the headline path of rounds 1-3), full
                     record with roofline, after one step of each path from identical weights on identical
                     triplets compared on the device
  dp_form_one_gpu    the headline workload through the N > 1 step FORM (row exchange + gradient sync hooks
                     over RCCL at world size 1, nothing skipped): each rank's compute floor at N = 8,
                     for both gradient-sync forms
  config4_per_gpu    BASELINE config 4's per-GPU workload (10 M-row fp16 table, bf16 MFMA, B = 8192
                     uniform) replayed from a hipGraph (the form config 4 names; `hipgraph`: true), the eager
                     step beside it (`eager`), with kernel timers, roofline and gather records
  reference_recipe   the reference's own run (train.py:354-364): B = 1024, uniform negatives, LARS
                     lr 1.0, margin 0.8 on the 1 M-row table
  fusion_resnet      the reference's production tower (models.py:125-157) at feature_size 1628
  predict            catalogue inference throughput (rows/s) over the 10 M-row table
  data_learnable     the headline step on a learnable catalogue (co-watched videos share a
                     cluster): the iid imitation_data features collapse the embeddings
"""
import argparse
import datetime
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
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


# ------------------------------------------------------------------ helpers ------
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_PHASE = {"name": "start"}


def set_phase(name):
    """Where this rank is: named in a failure message; mirrored to a file the self-launching
    parent reads when a rank hangs."""
    _PHASE["name"] = name
    d = os.environ.get("CDML_BENCH_PHASE_DIR")
    if d:
        try:
            with open(os.path.join(d, "rank%s" % os.environ.get("RANK", "0")), "w") as f:
                f.write(name)
        except OSError:
            pass


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


def csrc_hash():
    """sha256[:16] over the kernel sources (csrc/*.hip, *.h, sorted): the PMC summaries under profiles/ are
    stamped with it (profiles/latest_pmc.json, tools/pmc_stamp.py), so a summary taken from other kernels than
    the ones this run launches is recognised without git (the GPU box has none)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "collaborative-deep-metric-learning_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_DEFAULT_WORKLOAD = {"latest_pmc": "rows=1000000 batch=4096 mode=inbatch", "latest_pmc_x3_config1": "rows=1000000 batch=4096 mode=inbatch",
                        "latest_pmc_x3": "rows=1000000 batch=4096 mode=inbatch", "latest_pmc_bf16": "rows=10000000 batch=8192 mode=uniform"}


def workload_key(n_rows, batch, mode):
    return "rows=%d batch=%d mode=%s" % (n_rows, batch, mode)


def pmc_traffic(kernel, bf16=False, name=None, workload=None):
    """(bytes, source): HBM-side bytes per launch of `kernel` from the COMMITTED rocprofv3 PMC summary
    (profiles/latest_pmc.csv, latest_pmc_bf16.csv for the config-4 path: separate FETCH_SIZE /
    WRITE_SIZE passes of this bench, KiB; FETCH_SIZE doubled per the gfx950 correction) -- an earlier
    builder-run profiling pass, not a measurement of this run; `source` says which file and from when
    (profiles/latest_pmc.json, written by tools/profile_round.sh).  (None, None) if absent."""
    name = name or ("latest_pmc_bf16" if bf16 else "latest_pmc")
    path = os.path.join(ROOT, "profiles", name + ".csv")
    if not os.path.exists(path):
        return None, None
    import csv
    rows = list(csv.DictReader(open(path)))
    # `kernel` may leave trailing template arguments (tile-depth / stage tuning) open: an exact name wins, else the FIRST
    # kernel whose name starts with it (k_gemm_f32_sk must not pick up k_gemm_f32_sk_fixup's bytes)
    names = [r["kernel"] for r in rows]
    pick = kernel if kernel in names else next((n for n in names if not kernel.endswith(">") and n.startswith(kernel)), None)
    vals = {r["counter"]: float(r["mean_per_launch"]) for r in rows if r["kernel"] == pick}
    if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
        return None, None
    src = "profiles/%s.csv (builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run" % name
    try:
        meta = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json"))).get(name, {})
        src += "; taken %s at commit %s" % (meta.get("date", "?"), meta.get("commit", "?"))
    except (OSError, ValueError):
        meta = {}
    if meta.get("csrc_sha16") != csrc_hash():
        # the kernels changed since that pass (or it predates the stamp): a byte count of other code is not evidence
        return None, src + "; STALE: csrc/ differs from the sources that pass profiled, traffic withheld)"
    took = meta.get("workload", PMC_DEFAULT_WORKLOAD.get(name))
    if workload is not None and took != workload:
        # bytes per launch are those of the shapes that pass ran (stamps of rounds 3-4 carry none: config 1's)
        return None, src + "; OTHER WORKLOAD: that pass ran '%s', this record '%s', traffic withheld)" % (took, workload)
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, src + ")"


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
    train.py:314-323, median over the timed steps.  Bounded (~25 s in all): each run stops at 50
    timed steps or at its time share, whichever comes first -- the count is in `runs[].timed_steps`
    and in `sample` (--cpu-baseline-full: 50 steps everywhere, minutes)."""
    from oracle import synth as osynth, tower_torch
    table = osynth.features_numpy(10000, F, seed=0).astype(np.float32)
    pairs = osynth.cowatch_pairs(10000, 3000, 0)
    all_threads = usable_cores()
    prev = torch.get_num_threads()
    runs = []
    for threads, B, budget, warm in ((all_threads, 128, 3.0, 5), (all_threads, 1024, 6.0, 3),
                                     (1, 128, 6.0, 2), (1, 1024, 10.0, 1)):
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
                      "%d threads, median of %d timed steps (fetch %.1f ms + train %.1f ms); timed steps of the four "
                      "runs (threads x batch): %s -- each run is cut at its share of ~25 s, BASELINE.md's 50 only "
                      "with --cpu-baseline-full"
                      % (all_threads, head["timed_steps"], head["fetch_ms"], head["train_ms"],
                         ", ".join("%dx%d: %d" % (r["threads"], r["batch"], r["timed_steps"]) for r in runs)),
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


def settle_gpu(dev, seconds=0.3, precision="f32"):
    """GEMM launches on scratch buffers for ~`seconds`: the first ~100 ms of MFMA work after an idle spell run
    ~2 % slower on this part (clock ramp).  Not a step of the measured job: no state of the TrainStep is touched;
    disclosed in `order` / `warmup_effective`.  The launches are of the measured step's KIND of matrix work -- fp32 MFMA
    for precision "f32", bf16 MFMA for "f32x3" / "bf16" -- but of a kernel instantiation the step itself does not use, so a
    rocprofv3 run of this bench keeps clean per-kernel averages.  (Either kind leaves a 20-step run 3-4 % above the
    200-step rate, profiles/r04_settle_kinds.txt: the N = 1 lines settle on steps of a second TrainStep instead, main();
    this loop remains for N > 1, --mode predict, --only and CDML_SETTLE=gemm.)"""
    from cdml_amd import ops
    if precision == "f32":
        M, K, N = 4096, 1536, 5120
        # the split-K weight-gradient entry point: an MFMA-bound launch that is NOT part of the N = 1 step (stream-K there)
        x = torch.rand((M, K), device=dev)
        dy = torch.randn((M, N), device=dev) * 0.02
        dW = torch.empty((K, N), device=dev)
        db = torch.empty(N, device=dev)
        ws = torch.empty(max(ops.fc_bwd_weight_workspace(M, K, N), 16) // 4, device=dev)
        launch = lambda: ops.fc_bwd_weight(x, dy, dW, db, ws, M, K, N)
    else:
        M, K, N = 8192, 1536, 5120
        A = (torch.rand((M, K), device=dev) * 0.05).to(torch.bfloat16)
        Bm = (torch.randn((N, K), device=dev) * 0.03).to(torch.bfloat16)
        bias = torch.zeros(N, device=dev)
        if precision == "bf16":        # config 4's step runs the plain bf16 kernels: settle on the plane-walking one
            A3, B3 = torch.cat([A, A, A], 1).contiguous(), torch.cat([Bm, Bm, Bm], 1).contiguous()
            C = torch.empty((M, N), device=dev)
            launch = lambda: ops.gemm_bf16x3_nt(ops.BE_BIAS_LRELU_F32, A3, K, B3, K, C, M, N, K, bias=bias)
        else:                          # the f32x3 step runs the plane-walking kernels: settle on the plain bf16 one
            C = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            launch = lambda: ops.gemm_bf16_nt(ops.BE_BIAS_LRELU_BF16, A, Bm, C, M, N, K, bias=bias)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            launch()
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


# ------------------------------------------------- kernel timers and their records ------
def install_timers(kt, L, bf16):
    """Wrap the step's GEMM / tail / optimizer entry points of cdml_amd.ops with event pairs (active
    while kt.on).  Returns the function that puts the originals back."""
    from cdml_amd import ops
    real = {}

    def patch(name, label):
        if hasattr(ops, name):
            real[name] = getattr(ops, name)
            setattr(ops, name, kt.wrap(label, real[name]))
    if bf16 == "h2":                                # precision f16x2: the two-plane fp16 GEMMs
        # gemm_f16x2_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, out_scale, ...); _tn(A, pa, B, pb, C, M, N, K, out_scale, ...)
        patch("gemm_f16x2_nt", lambda e, *a, **k: {6: "fc1_fwd", 9: "fc1_fwd", 1: "fc2_fwd", 7: "dH1", 10: "dH1"}.get(e, "other"))
        patch("gemm_f16x2_tn", lambda A, pa, Bm, pb, C, M, N, K, *a, **k: "dW1" if N == L.Hp else "dW2")
        patch("split_f32_f16x2", "split_planes")
        patch("adam_matrix_bf16", lambda W, *a, **k: "adam_w1" if W.shape[0] == L.Fp else "adam_w2")
        patch("vnet_tail", "tail")

        def restore_h2():
            for name, fn in real.items():
                setattr(ops, name, fn)
        return restore_h2
    if bf16:
        # gemm_bf16_nt(epilogue, A, B, C, M, N, K, ...): FC1 has N = Hp, K = Fp; FC2 N = Dp; dH1 K = Dp
        patch("gemm_bf16_nt", lambda e, A, Bm, C, M, N, K, **k:
              "fc1_fwd" if (N == L.Hp and K == L.Fp) else ("fc2_fwd" if N == L.Dp else "dH1"))
        # gemm_bf16_tn(A, B, C, M, N, K, ...): dW1 has N = Hp
        patch("gemm_bf16_tn", lambda A, Bm, C, M, N, K, **k: "dW1" if N == L.Hp else "dW2")
        patch("gemm_bf16_tn2", "dW")                   # both weight gradients in one launch
        # matrices (with their bf16 operand copies), then the biases
        patch("adam_matrix_bf16", lambda W, *a, **k: "adam_w1" if W.shape[0] == L.Fp else "adam_w2")
        patch("adam_step", "adam_bias")
    elif bf16 is None:                             # precision f32x3: the split-fp32 GEMMs (+ the fp32 Adam)
        # gemm_bf16x3_nt(epilogue, A, plane_a, B, plane_b, C, M, N, K, ...); _tn(A, pa, B, pb, C, M, N, K, ...)
        # (by epilogue: 6 / 8 / 9 = FC1 with plane output, 1 = FC2, 7 / 10 = the data gradient, 3 = a weight gradient in the
        # k-contiguous form of the transposed activation layout; the k-strided form: FC2 when it carries a bias)
        patch("gemm_bf16x3_nt", lambda e, A, pa, Bm, pb, C, M, N, K, **k:
              {6: "fc1_fwd", 8: "fc1_fwd", 9: "fc1_fwd", 1: "fc2_fwd", 7: "dH1", 10: "dH1", 12: "dH1"}.get(
                  e, "dX" if N == L.Fp else "dW1" if N == L.Hp else "dW2"))      # (dX: the trainable table's row gradient dz1 . W1^T)
        patch("table_adam_rows", "table_adam")
        patch("gemm_bf16x3_tnk", lambda A, ma, a0, Bm, nb, b0, out, M, N, K, **k: "dW1" if N == L.Hp else "dW2")
        patch("gemm_bf16x3_tn", lambda A, pa, Bm, pb, C, M, N, K, **k:
              "fc2_fwd" if k.get("bias") is not None else ("dW1" if N == L.Hp else "dW2"))
        patch("transpose_to_bf16", "transpose_planes")
        patch("split_f32_bf16x3", "split_planes")
        patch("adam_matrix_bf16", lambda W, *a, **k: "adam_w1" if W.shape[0] == L.Fp else "adam_w2")
        patch("adam_step", "adam")
    else:
        patch("fc_lrelu_fwd", lambda x, W, b, y, M, K, N, *a, **k: "fc1_fwd" if N == L.Hp else "fc2_fwd")
        patch("fc_bwd_weight", lambda x, dy, dW, db, ws, M, K, N: "dW1" if N == L.Hp else "dW2")
        patch("fc_bwd_weight2", "dW")             # single GPU: both products in one stream-K launch
        patch("fc_bwd_data", lambda x, W, m, o, M, N, K, **k: "dX" if N == L.Fp else "dH1")
        patch("table_adam_rows", "table_adam")
        patch("adam_step", "adam")
        patch("lars_step", "lars")
        patch("lars_multi", "lars")
    patch("vnet_tail", "tail")
    # semi-hard mining (config 2): the B x 2B score product, the select, the indexed hinge, the separate l2norm kernels
    if not bf16:
        if bf16 is None:
            patch("fc_bwd_data", "score_gemm")           # (precision f32x3: its only fp32-MFMA launch, when not fused)
        patch("semihard_select", "semihard_select")
        patch("semihard_mine_x3", "semihard_mine")
        patch("triplet_hinge_indexed", "hinge_indexed")
        patch("l2norm_fwd", "l2norm_fwd")
        patch("l2norm_bwd", "l2norm_bwd")

    def restore():
        for name, fn in real.items():
            setattr(ops, name, fn)
    return restore


def gemm_records(kt, R, bf16, sampled, how, single_gpu, x3_products=0, pmc_name=None, workload=None):
    """roofline (dominant kernel: the weight-gradient GEMM), roofline_fc1_fwd, kernels -- from the
    event-timed launches.  Algorithmic (unpadded) flop: dW1 2*R*F*H, dW2 2*R*H*D.
    x3_products (precision f32x3): every algorithmic fp32 flop is that many bf16 MFMA flops, so the peak the
    fp32-equivalent rate is held against is the dense bf16 peak divided by it."""
    out = {}
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    if x3_products:
        peak = round(PEAK_BF16_MFMA_TFLOPS / x3_products, 1)
    have_dw = kt.count("dW") or (kt.count("dW1") and kt.count("dW2"))
    if not (have_dw and kt.count("fc1_fwd")):
        return out
    flops_gemm = 2.0 * R * F * H
    if kt.count("dW"):                          # dW1 and dW2 (+ the fix-up pass) in one call
        n_launch, t_ms = kt.count("dW"), kt.mean_ms("dW")
        if bf16:
            kname, klabel = "k_gemm_bf16_sk", "k_gemm_bf16_sk (dW1+dW2 in one launch, fix-up pass included)"
        else:
            kname, klabel = "k_gemm_f32_sk", "k_gemm_f32_sk (dW1+dW2 in one stream-K launch, fix-up pass included)"
    else:
        kname = "k_gemm_bf16_256<true, 3, true, false" if (bf16 or x3_products) else "k_gemm_f32<false, false, 2, 2, 3,"
        n_launch = kt.count("dW1") + kt.count("dW2")
        t_ms = (kt.mean_ms("dW1") * kt.count("dW1") + kt.mean_ms("dW2") * kt.count("dW2")) / n_launch
        klabel = kname + (" ...> (dW1+dW2 launches)" if not (bf16 or x3_products) else " (dW1+dW2 launches)")
        if x3_products:
            klabel = ("k_gemm_bf16_256<true, 3, true, true> (dW1+dW2 launches: fp32 products as %d bf16 plane products; "
                      "achieved = fp32-equivalent rate, peak = bf16 dense peak / %d)" % (x3_products, x3_products))
    flop_launch = sampled * (2.0 * R * F * H + 2.0 * R * H * D) / n_launch
    ach = flop_launch / (t_ms * 1e-3) / 1e12
    if x3_products:
        tr, src = (pmc_traffic("k_gemm_bf16_256<true, 3, true, true", name=pmc_name or "latest_pmc_x3", workload=workload)
                   if single_gpu else (None, None))
    else:
        tr, src = pmc_traffic(kname, bf16, name=pmc_name, workload=workload) if single_gpu else (None, None)
    out["roofline"] = {"bound": "mfma", "kernel": klabel, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                       "frac": round(ach / peak, 4), "traffic": tr, "traffic_source": src,
                       "launch_ms": round(t_ms, 4), "flop_per_launch": flop_launch,
                       "launches_per_step": n_launch / sampled, "timed_steps": sampled, "timed_how": how}
    ach1 = flops_gemm / (kt.mean_ms("fc1_fwd") * 1e-3) / 1e12
    k1 = "k_gemm_bf16_256<false, 0," if bf16 else "k_gemm_f32<true, false, 2, 2, 1,"
    if x3_products:
        # (full tiles + the last round's half tiles in one launch where the tile count leaves half a round or less;
        # k_gemm_bf16_256<false, 6, ..., R6> otherwise)
        k1 = "k_gemm_x3_rounds<6> / k_gemm_bf16_256<false, 6, true, true, false, false, true>"
    if x3_products:
        tr1, src1 = (None, None)
        if single_gpu:
            for kn in ("k_gemm_x3_rounds<6>", "k_gemm_bf16_256<false, 6, true, true, false, false, true"):
                tr1, src1 = pmc_traffic(kn, name=pmc_name or "latest_pmc_x3", workload=workload)
                if src1 is not None:
                    break
    else:
        tr1, src1 = pmc_traffic(k1, bf16, name=pmc_name, workload=workload) if single_gpu else (None, None)
    out["roofline_fc1_fwd"] = {"bound": "mfma", "kernel": k1 + ("" if (bf16 or x3_products) else " ...>"),
                               "achieved": round(ach1, 2), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(ach1 / peak, 4), "traffic": tr1, "traffic_source": src1,
                               "launch_ms": round(kt.mean_ms("fc1_fwd"), 4), "flop_per_launch": flops_gemm}
    kern = {}
    for k in ("fc1_fwd", "fc2_fwd", "tail", "dH1", "dW", "dW1", "dW2", "dX", "table_adam", "adam", "adam_w1", "adam_w2", "adam_bias", "lars",
              "split_planes", "transpose_planes", "score_gemm", "semihard_select", "semihard_mine", "hinge_indexed", "l2norm_fwd",
              "l2norm_bwd"):
        if kt.mean_ms(k) is not None:
            kern[k + "_ms"] = round(kt.mean_ms(k), 4)
            if kt.count(k) != sampled:
                kern[k + "_launches_per_step"] = round(kt.count(k) / sampled, 2)
    kern["empty_event_pair_ms"] = round(kt.overhead_ms, 5)      # included in the figures above
    out["kernels"] = kern
    return out


def gather_record(ts, mode, bf16, dev, pmc_name=None, workload=None):
    """The HBM-bound kernel the metric also names: the fused sampler+gather (fp32 rows, or fp16 -> bf16
    rows on the config-4 path), 10 back-to-back launches per event pair, fresh steps every launch."""
    from cdml_amd import ops, train
    reps, per = 5, 10
    n_st, rpt, B = ts.gather_ahead, ts.rows_per_triplet, ts.B
    R = B * rpt
    m = train._MODES[mode]
    gk = "k_sample_gather<%d," % (1 if rpt == 2 else 0)       # <mode, RowF32<6>> / <mode, RowF16<3>>
    gbytes = n_st * R * F * (2 + 2.0) if bf16 else n_st * 2.0 * R * F * 4     # rows read + normalised rows written
    x3 = (not bf16) and ts.ws.x_hat.dtype == torch.bfloat16
    h2 = (not bf16) and ts.ws.x_hat.dtype == torch.float16      # precision f16x2: fp32 rows read, two fp16 planes written
    ki = x3 and getattr(ts.ws, "xk", None) is not None
    if x3:
        # fp32 rows read, three bf16 planes written -- twice where the launch also writes the k8-interleaved copy the
        # first layer's weight gradient reads (round 5)
        gbytes = n_st * R * F * (4 + (12.0 if ki else 6.0))
        gk = "k_sample_gather<%d, RowF32X3" % (1 if rpt == 2 else 0)
    if h2:
        gbytes = n_st * R * F * (4 + 4.0)
        gk = "k_sample_gather<%d, RowF32H2" % (1 if rpt == 2 else 0)
    nxt = [ts.global_step + 1000]                   # fresh steps every launch: re-reading the same rows
                                                    # would be served from the 256 MB Infinity Cache

    def launch():                                   # (training on `ts` is over: its buffers are scratch now)
        if n_st > 1:
            ts._gather_block(nxt[0])
        else:
            ops.sample_gather(m, ts.pairs, ts.seed, nxt[0], B, ts.table.data, F, ts.idx, ts.ws.x_hat,
                              shift_out=ts.shift, x_ki=ts.ws.xk if ki else None)
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
    tr, src = ((None, None) if h2 else
               pmc_traffic("k_sample_gather<%d," % (1 if rpt == 2 else 0), name=pmc_name or "latest_pmc_x3", workload=workload) if x3
               else pmc_traffic(gk, bf16, name=pmc_name, workload=workload))
    # Which bytes.  `achieved` / `frac` divide the bytes the launch MOVES (rows read + the operand form
    # written: three bf16 planes on the split-fp32 path) by its time -- the HBM figure, what PMC FETCH + WRITE confirms.
    # SURVEY section 8(d) prices a gathered row at F*elt read + F*elt written (the normalised row in the table's own width):
    # `frac_algorithmic_8d` divides THOSE bytes by the same time -- lower on the plane path, whose rows cost 9 000 B of
    # writes for 6 000 B of algorithmic output.
    elt = 2 if bf16 else 4
    row_moved = gbytes / (n_st * R)
    row_alg = 2.0 * F * elt
    alg_ach = n_st * R * row_alg / (t_g * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": gk + ("<6>, true> (row-major planes + the k8-interleaved copy)" if ki else "<6>>" if (x3 or h2) else " RowF16<3>>" if bf16 else " RowF32<6>>"), "achieved": round(g_ach, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(g_ach / PEAK_HBM_GBS, 4), "traffic": tr,
            "traffic_source": src, "bytes_per_launch": gbytes, "launch_ms": round(t_g, 4), "steps_per_launch": n_st,
            "bytes_per_row_moved": row_moved, "bytes_per_row_algorithmic_8d": row_alg,
            "achieved_algorithmic_8d": round(alg_ach, 1), "frac_algorithmic_8d": round(alg_ach / PEAK_HBM_GBS, 4),
            "frac_is": "bytes MOVED per launch (rows read + operand form written) / time / 8 TB/s; frac_algorithmic_8d: SURVEY "
                       "8(d)'s F*elt read + F*elt written per row / the same time",
            "rows_per_launch": n_st * R,
            "method": "%d back-to-back launches per event pair (inter-launch gaps included), median of %d" % (per, reps)}


def measure_job(ts, steps, warmup, dev, bf16, timers=True, barrier=None):
    """W warm-up steps, then EXACTLY `steps` timed steps bracketed by (barrier +) synchronize on both
    sides; kernel timers on every 4th timed step (every 8th in long runs): an event pair costs the
    stream ~5 us, seven pairs a step are 1.5 % of it; the sampled launches are launches of the timed
    region.  Under graph replay the timers run on 8 extra EAGER steps after it.  Returns
    (elapsed_s, KernelTimer, sampled_steps, how)."""
    kt = KernelTimer()
    restore = install_timers(kt, ts.layout, bf16) if timers else (lambda: None)
    graph_run = ts.use_graph
    try:
        for _ in range(warmup):
            ts.step()
        torch.cuda.synchronize(dev)
        if barrier:
            barrier()
            torch.cuda.synchronize(dev)
        set_phase("timed steps")
        stride = 4 if steps < 100 else 8
        t0 = time.perf_counter()
        for i in range(steps):
            kt.on = timers and not graph_run and (i % stride == 0)
            ts.step()
        torch.cuda.synchronize(dev)
        if barrier:
            barrier()
            torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        kt.on = False
        sampled = (steps + stride - 1) // stride
        how = "event pairs on every %dth timed step" % stride
        if graph_run and timers:                     # same kernels, launched eagerly, for the roofline
            ts.use_graph = False
            sampled = 8
            for i in range(2 * sampled):
                kt.on = i % 2 == 0
                ts.step()
            torch.cuda.synchronize(dev)
            kt.on = False
            ts.use_graph = True
            how = "event pairs on 8 eager steps after the graph-replayed timed region"
    finally:
        kt.on = False
        restore()
    if timers:
        kt.calibrate()
    return elapsed, kt, sampled, how


# ------------------------------------------------------- self-launch (no launcher) ------
def self_launch(args):
    """`python bench.py --gpus N` with no launcher in the environment: start N ranks of this same
    command (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set; one process per GPU), relay rank 0's
    JSON line, exit non-zero naming the rank and its phase if a rank fails or the run exceeds
    CDML_BENCH_TIMEOUT_S.  This parent never touches the GPU."""
    n = args.gpus
    port = free_port()
    phase_dir = tempfile.mkdtemp(prefix="cdml_bench_")
    limit = float(os.environ.get("CDML_BENCH_TIMEOUT_S", "1500"))
    procs, lines = [], []
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CDML_BENCH_PHASE_DIR=phase_dir)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.getcwd(), stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      text=(r == 0)))

    def pump():
        for ln in procs[0].stdout:
            lines.append(ln.rstrip("\n"))
    reader = threading.Thread(target=pump, daemon=True)
    reader.start()

    def phases():
        out = []
        for r in range(n):
            try:
                out.append("rank %d: %s" % (r, open(os.path.join(phase_dir, "rank%d" % r)).read()))
            except OSError:
                out.append("rank %d: (not started)" % r)
        return "; ".join(out)

    t0, failed = time.time(), None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = "rank %d exited with code %d" % bad[0]
        elif all(c == 0 for c in codes):
            break
        elif time.time() - t0 > limit:
            failed = "no result within %.0f s (CDML_BENCH_TIMEOUT_S)" % limit
        else:
            time.sleep(0.25)
    if failed:
        where = phases()
        for p in procs:                               # these exact children, by handle
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
        sys.stderr.write("[bench] %d-rank run failed: %s [%s]\n" % (n, failed, where))
        sys.exit(1)
    reader.join(timeout=30)
    js = [ln for ln in lines if ln.startswith("{")]
    for ln in lines:
        if not ln.startswith("{"):
            sys.stderr.write(ln + "\n")
    if len(js) != 1:
        sys.stderr.write("[bench] rank 0 printed %d JSON lines (expected 1)\n" % len(js))
        sys.exit(1)
    print(js[0], flush=True)
    sys.exit(0)


# ------------------------------------------------------------ secondary records ------
LIKE_FOR_LIKE_WORKLOAD = ("config3 per-GPU batch on ONE GPU: 10000000 videos x 1500-d fp32 (61 GB) in HBM, batch 8192 "
                          "triplets, in-batch negatives (the N>1 lines run this per GPU on a row-sharded catalogue)")


def table_10m(keep, dev):
    """The 10 M-row fp32 catalogue (61 GB): the headline's own table at N = 1, shared with dp_form_one_gpu."""
    from cdml_amd import engine
    if "t10" not in keep:
        keep["t10"] = engine.FeatureTable.synthetic(10000000, F, seed=0, device=dev)
        keep["p10"] = torch.from_numpy(synth_pairs(10000000, 600000, seed=0)).to(dev)
    return keep["t10"], keep["p10"]


def rec_config1(dev, args, n_s, n_w, table, pairs):
    """BASELINE config 1 -- 1 M videos x 1500-d fp32, batch 4096, in-batch negatives -- the headline of rounds 1-4, as
    a full record at the headline's precision: its own warm-up, >= 100 timed steps (a 1.4-ms step: fewer would time
    the clock ramp), kernel timers, rooflines and the gather record."""
    from cdml_amd import train
    x3 = 6 if args.precision == "f32x3" else 0
    ts = train.TrainStep(table, pairs, 4096, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch", optimizer="adam",
                         base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev, precision=args.precision,
                         gather_ahead=args.gather_ahead)
    n, w = max(n_s, 100), max(n_w, 10)
    el, kt, sampled, how = measure_job(ts, n, w, dev, None if x3 else False)
    wk = workload_key(table.n_rows, 4096, "inbatch")
    out = {"workload": "config1: %d videos x 1500-d fp32 in HBM, 5000 hidden, 256-d embed, batch 4096 triplets, in-batch "
                       "negatives, margin 0.8, Adam, full step (the N = 1 headline of rounds 1-4)" % table.n_rows,
           "precision": args.precision, "value": round(4096 * n / el, 1), "unit": "triplets/s",
           "ms_per_step": round(el / n * 1e3, 4), "steps": n, "warmup": w, "loss": round(ts.loss(), 6)}
    out.update(gemm_records(kt, ts.R, False, sampled, how, True, x3_products=x3,
                            pmc_name="latest_pmc_x3_config1" if x3 else "latest_pmc", workload=wk))
    out["gather"] = gather_record(ts, "inbatch", False, dev, pmc_name="latest_pmc_x3_config1" if x3 else "latest_pmc",
                                  workload=wk)
    return out


def rec_config2(dev, args, n_s, n_w, table, pairs):
    """BASELINE config 2: the same 1 M catalogue, semi-hard negative mining over all pairs of the batch (every anchor
    against every embedded row: a B x 2B score product), batch 8192 -- at the headline's precision, >= 20 timed steps
    after its own warm-up, with the kernel breakdown (event pairs on sampled steps)."""
    from cdml_amd import train
    x3 = 6 if args.precision == "f32x3" else 0
    B2 = 8192
    ts = train.TrainStep(table, pairs, B2, output_size=D, hidden_size=H, margin=MARGIN, mode="semihard", optimizer="adam",
                         base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev, precision=args.precision,
                         gather_ahead=args.gather_ahead)
    n, w = max(n_s, 40), max(n_w, 5)
    el, kt, sampled, how = measure_job(ts, n, w, dev, None if x3 else False)
    out = {"workload": "config2: %d videos x 1500-d fp32 in HBM, batch %d triplets, semi-hard negatives mined over all "
                       "pairs of the batch (%d anchors x %d embedded rows), margin 0.8, Adam, full step"
                       % (table.n_rows, B2, B2, 2 * B2),
           "precision": args.precision, "value": round(B2 * n / el, 1), "unit": "triplets/s",
           "ms_per_step": round(el / n * 1e3, 4), "steps": n, "warmup": w, "loss": round(ts.loss(), 6),
           "mining": "fused into the score product's epilogue (no B x 2B score matrix in HBM)" if getattr(ts, "mine_fused", False)
                     else "score matrix S (B x 2B fp32, %d MB) written by the GEMM, scanned by k_semihard_select" % (B2 * 2 * B2 * 4 // 1000000)}
    out.update(gemm_records(kt, ts.R, False, sampled, how, False, x3_products=x3))
    return out


def rec_dp_form(dev, args, n_s, n_w, keep):
    """config 3's per-GPU workload through the N > 1 step FORM on one GPU: row exchange (routing kernel,
    two equal-split all-to-alls on their own communicator, un-permute) on the prefetch stream and the
    gradient sync over RCCL at world size 1, nothing skipped -- the collectives are self-copies, so this
    is each rank's COMPUTE floor at N = 8 (plus RCCL's launch costs), for both gradient-sync forms."""
    import torch.distributed as dist
    from cdml_amd import dist as cdist, train
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % free_port(), rank=0, world_size=1,
                                device_id=dev, timeout=datetime.timedelta(seconds=120))
    try:
        t10, p10 = table_10m(keep, dev)
        ex = cdist.RowExchange(10000000, group=dist.new_group(), skip_self=False)
        gs = cdist.GradSync(device=dev, skip_self=False)
        out = {"workload": "like_for_like's workload as TrainStep(exchange=RowExchange(skip_self=False), "
                           "grad_sync=GradSync(skip_self=False)) over RCCL at world size 1 (self-copies): the "
                           "per-rank compute floor of the N>1 step form; exchange capacity = all requests at "
                           "world 1 (1.39 x padded at world 8)",
               "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()), "precision": args.precision}
        forms = {"bucketed": "dW1 in two split-K row blocks + dW2, all-reduce per bucket under the next GEMM",
                 "two": "dW1 as one split-K launch, its all-reduce under the dW2 launch, [dW2|db2] after",
                 "single": "one stream-K launch for dW1+dW2, then ONE all-reduce"}
        for form in train.TrainStep.GRAD_SYNC_MODES:
            ts = train.TrainStep(t10, p10, 8192, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch",
                                 optimizer="adam", base_learning_rate=0.01, device=dev, exchange=ex, grad_sync=gs,
                                 batch_global=8192, grad_sync_mode=form, precision=args.precision)
            el = timed_steps(ts, n_s, n_w, dev)
            ts.check_inputs()
            out[form] = {"ms_per_step": round(el / n_s * 1e3, 4), "triplets_per_s": round(8192 * n_s / el, 1),
                         "form": forms[form]}
            del ts
        out["steps"], out["warmup"] = n_s, n_w
        return out
    finally:
        if own:
            dist.destroy_process_group()


def rec_train_table(dev, args, n_s, n_w, keep):
    """north_star's extension of the path ("the catalogue feature table and its Adam states shard row-wise"; the reference's
    features are a frozen placeholder, train.py:265): the headline workload -- 10 M rows, batch 8192, in-batch negatives, the
    headline's precision -- with the catalogue rows TRAINED: one more product per step (dLoss/dx_hat = dz1 . W1^T, 245.8 GFLOP
    at these shapes, on the plane kernels since round 6), then per touched row the l2norm backward and the lazy-Adam update
    (csrc/table_adam.hip).  Table + its two Adam states: 184 GB resident in one HBM.  Runs LAST on the shared catalogue: it
    changes its rows."""
    from cdml_amd import train
    x3 = 6 if args.precision == "f32x3" else 0
    t10, p10 = table_10m(keep, dev)
    ts = train.TrainStep(t10, p10, 8192, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch", optimizer="adam",
                         base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev, precision=args.precision,
                         train_table=True)
    n, w = max(n_s, 30), max(n_w, 5)
    el, kt, sampled, how = measure_job(ts, n, w, dev, None if x3 else False)
    R = ts.R
    out = {"workload": "the headline workload with a TRAINABLE catalogue: 10000000 videos x 1500-d fp32 + Adam m, v per row (184 GB "
                       "in one HBM), batch 8192 triplets, in-batch negatives, Adam on weights and on the gathered rows (lazy), "
                       "full step incl. dx_hat = dz1 . W1^T and the row update",
           "precision": args.precision, "value": round(8192 * n / el, 1), "unit": "triplets/s",
           "ms_per_step": round(el / n * 1e3, 4), "steps": n, "warmup": w, "loss": round(ts.loss(), 6),
           "gather_steps_per_launch": ts.gather_ahead}
    out.update(gemm_records(kt, R, False, sampled, how, False, x3_products=x3))
    if kt.mean_ms("dX") is not None:
        peak = round(PEAK_BF16_MFMA_TFLOPS / x3, 1) if x3 else PEAK_F32_MFMA_TFLOPS
        ach = 2.0 * R * F * H / (kt.mean_ms("dX") * 1e-3) / 1e12
        out["roofline_dx"] = {"bound": "mfma", "kernel": "k_gemm_bf16_256<false, 3, true, true, ..., R6> (dz1 planes . W1 planes^T, fp32 out)"
                                                           if x3 else "k_gemm_f32 NT (dz1 . W1^T)",
                              "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                              "launch_ms": round(kt.mean_ms("dX"), 4), "flop_per_launch": 2.0 * R * F * H}
    if kt.mean_ms("table_adam") is not None:
        # per touched row: the row, m, v read and written (6 x 6 KB) + its gathered gradient rows read (R x 6 KB in all)
        out["table_adam"] = {"launch_ms": round(kt.mean_ms("table_adam"), 4),
                             "bytes_bound": "<= %d rows x (6 KB gradient + 6 x 6 KB of row / m / v)" % R,
                             "achieved_GBs_upper": round(R * 7 * F * 4 / (kt.mean_ms("table_adam") * 1e-3) / 1e9, 1)}
    if x3:      # the same step on precision "f16x2" (the sixth product on the fp16 planes too), the rows as this run left them
        del ts
        torch.cuda.empty_cache()
        t2 = train.TrainStep(t10, p10, 8192, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch", optimizer="adam",
                             base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev, precision="f16x2", train_table=True)
        el2, _, _, _ = measure_job(t2, n, w, dev, "h2", timers=False)
        out["f16x2"] = {"ms_per_step": round(el2 / n * 1e3, 4), "value": round(8192 * n / el2, 1), "unit": "triplets/s", "steps": n,
                        "loss": round(t2.loss(), 6), "scale_moves": t2.ws.scales.changes}
    return out


def rec_config4(dev, args, n_s, n_w):
    from cdml_amd import engine_bf16, train
    t = engine_bf16.FeatureTableF16.synthetic(10000000, F, seed=0, device=dev)
    p = torch.from_numpy(synth_pairs(10000000, 600000, seed=0)).to(dev)
    mk = lambda g: train.TrainStep(t, p, 8192, output_size=D, hidden_size=H, margin=MARGIN, mode="uniform", optimizer="adam",
                                   base_learning_rate=0.01, device=dev, precision="bf16", gather_ahead=args.gather_ahead,
                                   use_graph=g)
    n_s, n_w = max(n_s, 100), max(n_w, 10)           # a 1-ms step: 20 of them would time the clock ramp
    wk = workload_key(10000000, 8192, "uniform")
    # BASELINE config 4 names a hipGraph-captured step: THAT form is the record's value; the eager step is timed beside
    # it on the same box, in the same process, in alternating blocks (graph, eager, graph, eager) so that neither form
    # owns the warmer half of the run (profiles/r05_graph_vs_eager.txt has the longer A/B and the kernel traces)
    tg, te = mk(True), mk(False)
    for ts_ in (tg, te):
        timed_steps(ts_, 0, n_w + 4, dev)              # (the replayed step: first step eager, then one capture per gather slot)
    half = max(n_s // 2, 50)
    blocks = {"graph": [], "eager": []}
    for _ in range(2):
        blocks["graph"].append(timed_steps(tg, half, 0, dev))
        blocks["eager"].append(timed_steps(te, half, 0, dev))
    ms_g, ms_e = (sum(blocks[k]) / (2 * half) * 1e3 for k in ("graph", "eager"))
    # kernel timers, roofline, gather: from the eager twin (same kernels, same shapes; a replay has no host-side launches
    # to bracket)
    el, kt, sampled, how = measure_job(te, 40, 0, dev, True)
    tg.check_inputs()
    te.check_inputs()
    out = {"workload": "config4 per-GPU shape on ONE GPU: 10000000 videos x 1500-d fp16 (30.7 GB) in HBM, bf16 MFMA "
                       "tower (f32 accumulate, fp32 master weights), batch 8192 triplets, uniform (global) negatives, Adam, "
                       "step replayed from a hipGraph",
           "hipgraph": True, "value": round(8192 / (ms_g * 1e-3), 1), "unit": "triplets/s", "ms_per_step": round(ms_g, 4),
           "steps": 2 * half, "warmup": n_w + 4, "dtype": "bf16 (fp16 table, f32 accumulate)", "loss": round(tg.loss(), 6),
           "eager": {"hipgraph": False, "value": round(8192 / (ms_e * 1e-3), 1), "ms_per_step": round(ms_e, 4), "steps": 2 * half,
                     "loss": round(te.loss(), 6)},
           "timed_how": "alternating blocks of %d steps: graph, eager, graph, eager (ms per block: graph %s, eager %s)"
                        % (half, [round(b * 1e3 / half, 4) for b in blocks["graph"]], [round(b * 1e3 / half, 4) for b in blocks["eager"]])}
    out.update(gemm_records(kt, te.R, True, sampled, how + " (the eager twin)", True, workload=wk))
    out["gather"] = gather_record(te, "uniform", True, dev, workload=wk)
    return out


def rec_reference_recipe(dev, args, n_s, n_w, table):
    """The reference's own run (train.py:354-364): batch 1024, uniform negatives with rejection, LARS
    (tf.contrib defaults) at learning rate 1.0, margin 0.8 -- on the 1 M-row table."""
    from cdml_amd import train
    pairs = torch.from_numpy(synth_pairs(table.n_rows, 300000, seed=0)).to(dev)
    x3 = 6 if args.precision == "f32x3" else 0
    ts = train.TrainStep(table, pairs, 1024, output_size=D, hidden_size=H, margin=MARGIN, mode="uniform",
                         optimizer="lars", base_learning_rate=1.0, device=dev, gather_ahead=args.gather_ahead,
                         precision="f32x3" if x3 else "f32")
    n = max(n_s, 200)                                  # a 0.6-ms step: 60 of them timed the clock ramp (0.674 against 0.643 ms as the job)
    el, kt, sampled, how = measure_job(ts, n, max(n_w, 20), dev, None if x3 else False)
    out = {"workload": "reference recipe (train.py:354-364): %d videos x 1500-d fp32, batch 1024 triplets (3072 rows), "
                       "uniform negatives, LARS lr 1.0, margin 0.8, full step" % table.n_rows,
           "precision": "f32x3" if x3 else "f32",
           "value": round(1024 * n / el, 1), "unit": "triplets/s", "ms_per_step": round(el / n * 1e3, 4), "steps": n,
           "loss": round(ts.loss(), 6)}
    out.update(gemm_records(kt, ts.R, False, sampled, how, False, x3_products=x3))
    return out


X3_DTYPE = ("f32 values as 3 exact bf16 planes (hi + mid + lo == the f32 value), 6 bf16-MFMA plane products per f32 product, "
            "f32 accumulate")


def rec_other_fp32_path(dev, args, n_s, n_w, table, pairs, B, mode, other):
    """Config 1 (`table`, `B`, `mode` of the caller) on the OTHER fp32 path -- `other` = "f32" (v_mfma_f32_32x32x2_f32: the headline of rounds
    1-3, record `f32_mfma`) when the headline runs precision "f32x3", and the reverse -- same table, batch, sampler and
    optimizer, with its own kernel timers and roofline.  Before timing, one step of each path is taken from the same
    weights on the same triplets and compared on the device."""
    from cdml_amd import train
    mk = lambda prec: train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode,
                                      optimizer="adam", base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev,
                                      precision=prec, gather_ahead=args.gather_ahead)
    a, b = mk("f32"), mk("f32x3")
    a.step(); b.step()
    torch.cuda.synchronize(dev)
    ga, gb = a.params.grad.double(), b.params.grad.double()
    check = {"same_triplets": bool(torch.equal(a.idx, b.idx)),
             "max_abs_embedding_diff": float((a.ws.e - b.ws.e).abs().max().item()),
             "loss_f32_mfma": round(a.loss(), 7), "loss_f32x3": round(b.loss(), 7),
             "gradient_rel_l2_diff": float(((ga - gb).norm() / ga.norm().clamp_min(1e-300)).item()),
             "note": "one step each from identical weights.  The gradient difference is what two fp32-accurate forward "
                     "passes give on this iid catalogue: where a pre-activation lies within rounding of zero leaky-relu' "
                     "flips between them and single gradient elements move by most of one term (|g| itself is a sum of "
                     "cancelling terms here).  The parity tests hold both paths to the same bounds against "
                     "the fp64 oracle (tests/test_gpu_parity.py::test_train_steps_config0, tests/test_gpu_fullsize.py::"
                     "test_gradients_well_conditioned_production_shape, tests/test_gpu_f32x3.py)"}
    ts = a if other == "f32" else b
    del a, b
    torch.cuda.empty_cache()
    x3 = 6 if other == "f32x3" else 0
    n = max(n_s, 60)
    el, kt, sampled, how = measure_job(ts, n, max(n_w, 10), dev, None if x3 else False)
    out = {"workload": "config 1 (%d videos, batch %d, %s negatives, Adam) with precision %s"
                       % (table.n_rows, B, mode, "f32x3" if x3 else "f32: the projection products on the fp32 MFMA "
                                                                    "(v_mfma_f32_32x32x2_f32), the headline path of rounds 1-3"),
           "value": round(B * n / el, 1), "unit": "triplets/s", "ms_per_step": round(el / n * 1e3, 4), "steps": n,
           "dtype": X3_DTYPE if x3 else "f32 (fp32 MFMA)",
           "loss": round(ts.loss(), 6), "f32_mfma_against_f32x3": check}
    out.update(gemm_records(kt, ts.R, False, sampled, how, True, x3_products=x3,
                            workload=workload_key(table.n_rows, B, mode)))
    del ts
    torch.cuda.empty_cache()
    # a run that LEARNS (data_learnable's catalogue): both paths from the same seeds, the loss along the way
    tl, pl = learnable_catalogue(200000, dev)
    curves = {}
    for prec in ("f32", "f32x3"):
        t = train.TrainStep(tl, pl, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode, optimizer="adam",
                            base_learning_rate=2e-4, device=dev, precision=prec, gather_ahead=args.gather_ahead)
        losses = []
        for i in range(61):
            t.step()
            if i % 15 == 0:
                losses.append(round(t.loss(), 4))
        curves[prec] = losses
        del t
        torch.cuda.empty_cache()
    out["learnable_catalogue_loss_every_15_steps"] = curves
    return out


def h2_records(kt, R, sampled, how, workload=None):
    """roofline / roofline_fc1_fwd / kernels of a precision-"f16x2" step from its event-timed launches (peak = the fp16 dense
    peak / 3: three fp16 MFMA flops per algorithmic fp32 flop)."""
    out = {}
    if not (kt.count("dW1") and kt.count("dW2") and kt.count("fc1_fwd")):
        return out
    peak = round(PEAK_BF16_MFMA_TFLOPS / 3.0, 1)          # (the fp16 dense peak of the part is the bf16 one)
    n_launch = kt.count("dW1") + kt.count("dW2")
    t_ms = (kt.mean_ms("dW1") * kt.count("dW1") + kt.mean_ms("dW2") * kt.count("dW2")) / n_launch
    flop_launch = sampled * (2.0 * R * F * H + 2.0 * R * H * D) / n_launch
    ach = flop_launch / (t_ms * 1e-3) / 1e12
    # (HBM-side bytes per launch from the committed PMC passes of the f16x2 step, when they are of these kernels and this workload)
    tr, src = pmc_traffic("k_gemm_f16x2_256<true, 3, true, true", name="latest_pmc_f16x2", workload=workload) if workload else (None, None)
    out["roofline"] = {"bound": "mfma", "kernel": "k_gemm_f16x2_256<true, 3, true, true> (dW1+dW2 launches: fp32 products as 3 fp16 "
                       "plane products; achieved = fp32-equivalent rate, peak = fp16 dense peak / 3)", "achieved": round(ach, 2),
                       "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": tr, "traffic_source": src,
                       "launch_ms": round(t_ms, 4),
                       "flop_per_launch": flop_launch, "launches_per_step": n_launch / sampled, "timed_steps": sampled,
                       "timed_how": how}
    ach1 = 2.0 * R * F * H / (kt.mean_ms("fc1_fwd") * 1e-3) / 1e12
    tr1, src1 = pmc_traffic("k_gemm_f16x2_256<false, 6, true, true", name="latest_pmc_f16x2", workload=workload) if workload else (None, None)
    out["roofline_fc1_fwd"] = {"bound": "mfma", "kernel": "k_gemm_f16x2_256<false, 6, true, true>", "achieved": round(ach1, 2),
                               "peak": peak, "unit": "TFLOP/s", "frac": round(ach1 / peak, 4), "traffic": tr1, "traffic_source": src1,
                               "launch_ms": round(kt.mean_ms("fc1_fwd"), 4), "flop_per_launch": 2.0 * R * F * H}
    kern = {}
    for k in ("fc1_fwd", "fc2_fwd", "tail", "dH1", "dW1", "dW2", "adam_w1", "adam_w2", "split_planes"):
        if kt.mean_ms(k) is not None:
            kern[k + "_ms"] = round(kt.mean_ms(k), 4)
            if kt.count(k) != sampled:
                kern[k + "_launches_per_step"] = round(kt.count(k) / sampled, 2)
    kern["empty_event_pair_ms"] = round(kt.overhead_ms, 5)
    out["kernels"] = kern
    return out


F16X2_DTYPE = ("f32 values as 2 fp16 planes of (value x a per-tensor power of two): hi + lo holds 22 significant bits; 3 fp16-MFMA "
               "plane products per f32 product, f32 accumulate; delayed per-tensor scales (engine_f16x2.py)")


def rec_f16x2(dev, args, n_s, n_w, table, pairs, B, mode):
    """The headline workload (`table`, `B`, `mode` of the caller) on precision "f16x2" -- every fp32 operand of the five
    projection products as two fp16 planes under a per-tensor power-of-two scale, three plane products on
    v_mfma_f32_16x16x32_f16 (csrc/gemm_f16x2_256.hip, engine_f16x2.py): a SECONDARY record (VERDICT r5 #3's ruling: dtype
    spelled out, peak = the fp16 dense peak / 3), the headline stays "f32x3".  Before timing, one step of this path and one
    of "f32x3" are taken from the same weights on the same triplets and compared on the device."""
    from cdml_amd import train
    mk = lambda prec: train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode,
                                      optimizer="adam", base_learning_rate=0.01, seed=1234, weight_seed=42, device=dev,
                                      precision=prec, gather_ahead=args.gather_ahead)
    lean = os.environ.get("CDML_F16X2_LEAN") == "1"       # profiling runs: the timed job only
    if lean:
        a = mk("f16x2")
        el, kt, sampled, how = measure_job(a, max(n_s, 4), n_w, dev, "h2", timers=not args.no_kernel_timers)
        return {"workload": "the headline's with precision f16x2 (lean: the timed job only)", "value": round(B * max(n_s, 4) / el, 1),
                "unit": "triplets/s", "ms_per_step": round(el / max(n_s, 4) * 1e3, 4), "steps": max(n_s, 4), "warmup": n_w}
    a, b = mk("f16x2"), mk("f32x3")
    a.step(); b.step()
    torch.cuda.synchronize(dev)
    ga, gb = a.params.grad.double(), b.params.grad.double()
    check = {"same_triplets": bool(torch.equal(a.idx, b.idx)),
             "max_abs_embedding_diff": float((a.ws.e - b.ws.e).abs().max().item()),
             "loss_f16x2": round(a.loss(), 7), "loss_f32x3": round(b.loss(), 7),
             "gradient_rel_l2_diff": float(((ga - gb).norm() / gb.norm().clamp_min(1e-300)).item()),
             "note": "one step each from identical weights (the same comparison f32_mfma.f32_mfma_against_f32x3 makes between the "
                     "two other fp32 paths, with the same caveat: leaky-relu' flips where a pre-activation is rounding noise).  "
                     "The bounds: tests/test_gpu_f16x2.py (every product against fp64 beside the fp32-MFMA kernel; a training run "
                     "across the scale checks against the fp64 oracle beside f32x3), tests/test_gpu_parity.py::test_train_steps_config0[f16x2]"}
    del b
    torch.cuda.empty_cache()
    ts = a
    n, w = max(n_s, 100), max(n_w, 10)
    el, kt, sampled, how = measure_job(ts, n, w, dev, "h2")
    R = ts.R
    out = {"workload": "the headline's: %d videos x 1500-d fp32 in HBM, batch %d triplets, %s negatives, Adam, full step -- with "
                       "precision f16x2" % (table.n_rows, B, mode),
           "value": round(B * n / el, 1), "unit": "triplets/s", "ms_per_step": round(el / n * 1e3, 4), "steps": n, "warmup": w,
           "dtype": F16X2_DTYPE, "loss": round(ts.loss(), 6), "f16x2_against_f32x3": check,
           "plane_scales": {k: (int(round(np.log2(v))) if v > 0 else None) for k, v in ts.ws.scales.state().items()},
           "plane_scales_are": "log2 of the per-tensor scales after the run; scale moves after calibration: %d in %d steps "
                               "(checked at steps 0, 1, 2, 4 .. 64 and every 64th: two device-to-host copies each, inside the timed "
                               "region); checks that found a plane tensor clamped at fp16's largest value: %d"
                               % (ts.ws.scales.changes, ts.global_step, ts.ws.scales.saturated),
           "gather_steps_per_launch": ts.gather_ahead}
    out.update(h2_records(kt, R, sampled, how, workload=workload_key(table.n_rows, B, mode)))
    # the gather writes 6 000 B of planes per row where the three bf16 planes are 9 000
    reps, per, n_st = 5, 10, ts.gather_ahead
    nxt = [ts.global_step + 1000]

    def launch():
        if n_st > 1:
            ts._gather_block(nxt[0])
        else:
            from cdml_amd import ops
            ops.sample_gather(train._MODES[mode], ts.pairs, ts.seed, nxt[0], B, ts.table.data, F, ts.idx, ts.ws.x_hat, shift_out=ts.shift)
        nxt[0] += n_st
    for _ in range(3):
        launch()
    evs = []
    for _ in range(reps):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(per):
            launch()
        e_.record()
        evs.append((s_, e_))
    torch.cuda.synchronize(dev)
    t_g = float(np.median([s_.elapsed_time(e_) for s_, e_ in evs])) / per
    gbytes = n_st * R * F * (4 + 4.0)
    out["gather"] = {"bound": "hbm", "kernel": "k_sample_gather<%d, RowF32H2<6>>" % (1 if ts.rows_per_triplet == 2 else 0),
                     "achieved": round(gbytes / (t_g * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(gbytes / (t_g * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "bytes_per_row_moved": 8.0 * F,
                     "launch_ms": round(t_g, 4), "rows_per_launch": n_st * R}
    del ts, a
    torch.cuda.empty_cache()
    # a run that LEARNS (data_learnable's catalogue): this path and f32x3 from the same seeds, the loss along the way
    tl, pl = learnable_catalogue(200000, dev)
    curves = {}
    for prec in ("f16x2", "f32x3"):
        t = train.TrainStep(tl, pl, 4096, output_size=D, hidden_size=H, margin=MARGIN, mode=mode, optimizer="adam",
                            base_learning_rate=2e-4, device=dev, precision=prec, gather_ahead=args.gather_ahead)
        losses = []
        for i in range(151):
            t.step()
            if i % 30 == 0:
                losses.append(round(t.loss(), 4))
        curves[prec] = losses
        if prec == "f16x2":
            curves["f16x2_scale_moves"] = t.ws.scales.changes
        del t
        torch.cuda.empty_cache()
    out["learnable_catalogue_loss_every_30_steps"] = curves
    del tl, pl
    torch.cuda.empty_cache()
    # BASELINE configs 1 and 2 on this precision (the 1 M-row catalogue; config 2's miner is the six-plane kernel of
    # cdml_semihard_mine_x3 -- it splits the fp32 embeddings itself -- its tower the fp16 kernels)
    from cdml_amd import engine
    t1 = engine.FeatureTable.synthetic(1000000, F, seed=0, device=dev)
    p1 = torch.from_numpy(synth_pairs(1000000, 333333, seed=0)).to(dev)
    others = {}
    for name, bc, mc, opt, lr_ in (("config1", 4096, "inbatch", "adam", 0.01), ("config2_semihard", 8192, "semihard", "adam", 0.01),
                                   ("reference_recipe", 1024, "uniform", "lars", 1.0)):      # (train.py:354-364: B = 1024, LARS lr 1.0)
        t = train.TrainStep(t1, p1, bc, output_size=D, hidden_size=H, margin=MARGIN, mode=mc, optimizer=opt,
                            base_learning_rate=lr_, seed=1234, weight_seed=42, device=dev, precision="f16x2",
                            gather_ahead=args.gather_ahead)
        nn_ = max(n_s, 200 if name == "reference_recipe" else 60)
        el2, _, _, _ = measure_job(t, nn_, max(n_w, 10), dev, "h2", timers=False)
        others[name] = {"ms_per_step": round(el2 / nn_ * 1e3, 4), "value": round(bc * nn_ / el2, 1), "unit": "triplets/s",
                        "steps": nn_, "loss": round(t.loss(), 6), "scale_moves": t.ws.scales.changes}
        del t
        torch.cuda.empty_cache()
    out["baseline_configs_on_f16x2"] = others
    return out


def rec_fusion(dev, args, n_s, n_w):
    """The reference's production tower (ResNet, models.py:125-157) at feature_size 1628 = 1500 visual +
    128 doc (online_data.py:38): batch 1024 uniform triplets, Adam."""
    from cdml_amd import engine, fusion
    n_rows = 1000000
    table = engine.FeatureTable.synthetic(n_rows, 1628, seed=0, device=dev)
    pairs = torch.from_numpy(synth_pairs(n_rows, 300000, seed=0)).to(dev)
    out = None
    for prec in ("auto", "f32", "f16x2"):            # round 6: the visual branch on the plane kernels; the fp32-MFMA tower and the fp16 planes beside it
        ts = fusion.FusionTrainStep("ResNet", table, pairs, 1024, device=dev, precision=prec)
        n = max(n_s, 100)
        el = timed_steps(ts, n, max(n_w, 10), dev)
        r = {"precision": ts.precision, "value": round(1024 * n / el, 1), "ms_per_step": round(el / n * 1e3, 4), "steps": n,
             "loss": round(ts.loss(), 6)}
        if out is None:
            out = dict(r, workload="ResNet fusion tower (models.py:125-157): %d videos x 1628-d fp32, batch 1024 uniform triplets "
                                   "(3072 rows), Adam, full step" % n_rows, unit="triplets/s")
        else:
            out["f32_mfma" if prec == "f32" else prec] = r
        del ts
    return out


def rec_predict(dev, precision, n_rows=10000000, chunk=100000, reps=2):
    """Catalogue inference (predict.py:71-96; the reference's chunk is 100 000 rows, predict.py:146):
    forward-only tower over every row of the HBM-resident table, embeddings left in HBM.  Algorithmic
    work per row: 2*F*H + 2*H*D flop; F*s bytes read + D*4 written."""
    from cdml_amd import engine, engine_bf16, predict
    bf16 = precision == "bf16"
    Table = engine_bf16.FeatureTableF16 if bf16 else engine.FeatureTable
    table = Table.synthetic(n_rows, F, seed=0, device=dev)
    from cdml_amd import engine_x3
    L = (engine_bf16.layout_bf16 if bf16 else engine_x3.layout_x3 if precision in ("f32x3", "f16x2") else engine.TowerLayout)(F, H, D)
    pr = predict.Prediction(params=engine.VNetParams(L, dev, 42), precision=precision)
    out = torch.empty((n_rows, D), dtype=torch.float32, device=dev)
    pr.embed_table(table, chunk, out=out)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        pr.embed_table(table, chunk, out=out)
    torch.cuda.synchronize(dev)
    el = (time.perf_counter() - t0) / reps
    nrm = float((out[::9973].norm(dim=1) - 1).abs().max())
    flops = n_rows * (2.0 * F * H + 2.0 * H * D)
    peak = (PEAK_BF16_MFMA_TFLOPS if bf16 else round(PEAK_BF16_MFMA_TFLOPS / 6, 1) if precision == "f32x3"
            else round(PEAK_BF16_MFMA_TFLOPS / 3, 1) if precision == "f16x2" else PEAK_F32_MFMA_TFLOPS)
    return {"workload": "catalogue inference: %d videos x 1500-d %s in HBM -> 256-d unit-norm embeddings in HBM, "
                        "%d-row chunks (l2norm + FC1 + FC2 + l2norm)" % (n_rows, "fp16" if bf16 else "fp32", chunk),
            "value": round(n_rows / el, 1), "unit": "rows/s", "seconds": round(el, 4), "passes_timed": reps,
            "tower_tflops": round(flops / el / 1e12, 2), "frac_of_mfma_peak": round(flops / el / 1e12 / peak, 4),
            "max_abs_norm_error": nrm,
            "dtype": ("bf16 (fp16 table, f32 accumulate)" if bf16 else
                      "f32 values as 2 fp16 planes under per-tensor scales, 3 plane products on the fp16 MFMA (peak = fp16 dense peak / 3)"
                      if precision == "f16x2" else "f32" if precision != "f32x3" else
                      "f32 values as 3 exact bf16 planes, 6 plane products on the bf16 MFMA (peak = bf16 dense peak / 6)")}


def rec_knn(dev, n=343455, k=51, precision="f32x3"):
    """The kNN export (faiss_knn.py:82-131 `calc_knn`; the reference's production catalogue: doc_location = 343455,
    faiss_knn.py:389; nearest_num = 51 = 50 neighbours + the query): EXACT self-kNN of n unit 256-d embeddings -- the
    reference builds an approximate HNSW index on the CPU.  Algorithmic work: 2 n^2 D flop of inner products.  Round 6: only the
    first 32 768 catalogue rows go through score blocks (they give every query its k-th best distance so far); the other
    90 % of the n^2 scores are never written -- the plane GEMM's epilogue appends the few elements within that distance
    to the query's candidate list (cdml_knn_filter_x3), one merge per query at the end."""
    from cdml_amd import knn
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    e = torch.randn(n, D, device=dev, generator=g)
    knn.knn_search(e[:8192], e[:8192], k, precision=precision)            # (loads the kernels)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    Dk, Ik = knn.knn_search(e, e, k, precision=precision)
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    flops = 2.0 * n * n * D
    peak = (round(PEAK_BF16_MFMA_TFLOPS / 6, 1) if precision == "f32x3" else round(PEAK_BF16_MFMA_TFLOPS / 3, 1) if precision == "f16x2"
            else PEAK_F32_MFMA_TFLOPS)
    self_first = bool((Ik[:, 0] == torch.arange(n, device=dev)).all())
    return {"workload": "exact self-kNN export: %d unit embeddings x %d-d, k = %d (faiss_knn.py:82-131, :389), l2-normalise + "
                        "inner products + per-query top-k merge, results in HBM" % (n, D, k),
            "precision": precision, "value": round(n / el, 1), "unit": "queries/s", "seconds": round(el, 4),
            "inner_product_tflops": round(flops / el / 1e12, 2), "frac_of_mfma_peak": round(flops / el / 1e12 / peak, 4),
            "peak_tflops": peak,
            "score_blocks": "the first %d catalogue rows only (%.0f %% of the scores: written and merged out of the Infinity Cache); "
                            "the rest filtered in the score product's epilogue, nothing written" % (knn.FIRST_BLOCK, 100.0 * knn.FIRST_BLOCK / n),
            "query_is_its_own_first_neighbour": self_first, "mean_second_neighbour_d2": round(float(Dk[:, 1].mean()), 5)}


def rec_learnable(dev, args, n_s, n_w, B):
    from cdml_amd import train
    tl, pl = learnable_catalogue(200000, dev)
    tsl = train.TrainStep(tl, pl, B, output_size=D, hidden_size=H, margin=MARGIN, mode="inbatch",
                          optimizer="adam", base_learning_rate=2e-4, device=dev, gather_ahead=args.gather_ahead,
                          precision=args.precision)
    tsl.step()
    l0 = tsl.loss()
    n_l = max(n_s, 60)
    el = timed_steps(tsl, n_l, n_w, dev)
    return {"workload": "config1 step (batch %d in-batch) on a learnable 200000 x 1500 catalogue "
                        "(co-watched videos share one of 2000 clusters), Adam 2e-4" % B, "precision": args.precision,
            "value": round(B * n_l / el, 1), "unit": "triplets/s", "ms_per_step": round(el / n_l * 1e3, 4),
            "steps": n_l, "loss_first_step": round(l0, 4), "loss_last_step": round(tsl.loss(), 4)}


# ------------------------------------------------------------------------ main ------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=None, help="catalogue rows (default 10M; 1000000 with --batch 4096 = config 1 as the job)")
    ap.add_argument("--mode", default=None, choices=["inbatch", "uniform", "semihard", "predict"])
    ap.add_argument("--batch", type=int, default=None, help="triplets per GPU per step (default 8192)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph")
    ap.add_argument("--precision", default="f32x3", choices=["f32x3", "f32", "bf16", "f32x3-3", "f16x2"],
                    help="f32x3 (default): fp32 operands as three exact bf16 planes, six plane products per fp32 product on "
                         "the bf16 MFMA, f32 accumulate; f32: the fp32 MFMA; bf16 = BASELINE config 4 path (fp16 table + "
                         "bf16 MFMA), not the headline metric; f32x3-3: three products (16-bit operands), measurement only; "
                         "f16x2: fp32 operands as two fp16 planes under per-tensor scales, three products on the fp16 MFMA -- a "
                         "secondary path (DESIGN section 5), the job of this line only when asked for")
    ap.add_argument("--train-table", action="store_true",
                    help="also train the catalogue rows (lazy Adam; build-defined, not the headline metric)")
    ap.add_argument("--gather-ahead", type=int, default=0,
                    help="steps fetched per sampler+gather launch (1 GPU); 0 = TrainStep's rule (1 .. 4 by the bytes a launch writes)")
    ap.add_argument("--grad-sync", default="auto", choices=["auto", "bucketed", "two", "single"],
                    help="N>1: how the gradient all-reduce is issued (auto: both forms are timed for 5 steps "
                         "before the warm-up, the faster one is kept)")
    ap.add_argument("--capacity-factor", type=float, default=1.25,
                    help="N>1: headroom of the fixed-capacity row exchange over the mean share per peer")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="50 timed CPU steps in all four runs (minutes)")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary records of the default N=1 line")
    ap.add_argument("--extras", default=None,
                    help="comma list of secondary records to run (default: all): config1,config2_semihard,dp_form_one_gpu,train_table,"
                         "config4_per_gpu,reference_recipe,fusion_resnet,predict,data_learnable,f32_mfma")
    ap.add_argument("--only", default=None, choices=["reference_recipe", "fusion_resnet", "data_learnable", "f16x2"],
                    help="run ONE secondary record as the job (its JSON line; for rocprofv3 runs of that workload)")
    ap.add_argument("--no-settle", action="store_true",
                    help="skip the 0.3 s GEMM loop before the warm-up steps (rocprof runs: keeps its launches out of the kernel averages)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    launched = "RANK" in os.environ and world >= 1
    if args.gpus > 1 and not launched:
        self_launch(args)                                # never returns
    world = max(world, 1) if launched else 1
    # ONE line on stdout: libraries write there too (RCCL prints a version banner when a communicator
    # comes up), so from here on file descriptor 1 is stderr and the JSON line goes to the saved one
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    set_phase("start")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback)")
    backend = os.environ.get("CDML_DIST_BACKEND", "nccl")     # "gloo": 1-GPU-box rehearsal of N>1
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    from cdml_amd import dist as cdist, engine, engine_bf16, train
    bf16 = args.precision == "bf16"
    x3 = 0 if not args.precision.startswith("f32x3") else (3 if args.precision.endswith("-3") else 6)
    h2 = args.precision == "f16x2"
    if h2 and args.graph and args.gpus > 1:
        sys.exit("--precision f16x2 replays from a hipGraph on one GPU only")
    Table = engine_bf16.FeatureTableF16 if bf16 else engine.FeatureTable

    if args.mode == "predict":                           # catalogue inference throughput (N = 1)
        if world != 1:
            sys.exit("--mode predict is a one-GPU measurement")
        set_phase("predict")
        if not args.no_settle:
            settle_gpu(dev, precision=args.precision[:5])
        r = rec_predict(dev, args.precision, n_rows=args.rows or 10000000)
        out = {"metric": "catalogue rows embedded/sec", "value": r["value"], "unit": "rows/s", "n_gpus": 1,
               "steps": r["passes_timed"], "warmup": 1, "ms_per_step": round(r["seconds"] * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": r["dtype"], "data": "synthetic",
               "config": {"workload": r["workload"]}, "predict": r}
        print(json.dumps(out), file=result_out, flush=True)
        return

    if args.only:                                        # one secondary record as the whole job (N = 1)
        if world != 1:
            sys.exit("--only is a one-GPU measurement")
        set_phase("only: " + args.only)
        if not args.no_settle:
            settle_gpu(dev, precision=args.precision[:5])
        n_s, n_w = args.steps, args.warmup
        if args.only == "reference_recipe":
            r = rec_reference_recipe(dev, args, n_s, n_w, engine.FeatureTable.synthetic(1000000, F, seed=0, device=dev))
        elif args.only == "fusion_resnet":
            r = rec_fusion(dev, args, n_s, n_w)
        elif args.only == "f16x2":
            nr = args.rows or 10000000
            r = rec_f16x2(dev, args, n_s, n_w, engine.FeatureTable.synthetic(nr, F, seed=0, device=dev),
                          torch.from_numpy(synth_pairs(nr, max(nr // 16, 1000), seed=0)).to(dev), args.batch or 8192, "inbatch")
        else:
            r = rec_learnable(dev, args, n_s, n_w, args.batch or 4096)
        out = {"metric": "triplets/sec", "value": r["value"], "unit": "triplets/s", "n_gpus": 1, "steps": r["steps"],
               "warmup": n_w, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": F16X2_DTYPE if args.only == "f16x2" else X3_DTYPE if args.precision == "f32x3" else "f32 (fp32 MFMA)",
               "data": "synthetic", "config": {"workload": r["workload"], "precision": "f16x2" if args.only == "f16x2" else args.precision},
               args.only: r}
        print(json.dumps(out), file=result_out, flush=True)
        return

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        set_phase("init_process_group")
        # a finite timeout: a stuck collective ends the run with a non-zero exit instead of hanging
        tmo = datetime.timedelta(seconds=int(os.environ.get("CDML_DIST_TIMEOUT_S", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)

    mode = args.mode or ("uniform" if bf16 else "inbatch")
    n_rows = args.rows or (1000000 if mode == "semihard" else 10000000)      # --mode semihard alone = BASELINE config 2
    B = args.batch or 8192
    # the default N = 1 job: config 3's per-GPU workload on one GPU (10 M rows, batch 8192, in-batch) -- the 1-GPU point
    # of the curve the N > 1 commands continue; it carries the secondary records
    default_job = world == 1 and not bf16 and args.rows is None and args.batch is None and mode == "inbatch"
    wkey = workload_key(n_rows, B, mode)
    set_phase("setup")
    probe = None
    try:
        if world > 1:
            lo, hi, _ = cdist.shard_bounds(n_rows, world, rank)
            table = Table.synthetic(hi - lo, F, seed=0, device=dev, row0=lo, n_rows_global=n_rows)
            # own communicator for the exchange so it never queues behind the gradient all-reduce
            ex_group = dist.new_group()
            exchange = cdist.RowExchange(n_rows, group=ex_group, capacity_factor=args.capacity_factor)
            grad_sync = cdist.GradSync(device=dev)
            # bring both communicators up here (RCCL builds them on first use), not inside a step
            set_phase("communicator warm-up")
            warm = torch.zeros(1, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(warm)
            dist.all_reduce(warm, group=ex_group)
            set_phase("setup")
        else:
            table = Table.synthetic(n_rows, F, seed=0, device=dev)
            exchange = grad_sync = None
        # users: a third of the catalogue, capped -- the run consumes < 1 M pairs per rank, and every
        # rank builds the same list on its host while the others wait
        pairs = torch.from_numpy(synth_pairs(n_rows, max(min(n_rows // 3, 600000), 1000), seed=0)).to(dev)

        ts = train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode,
                             optimizer="adam", base_learning_rate=0.01, seed=1234, weight_seed=42,
                             device=dev, exchange=exchange, grad_sync=grad_sync, slot0=rank * B,
                             batch_global=world * B, precision=args.precision,
                             # N > 1: the overlapped replay (exchange graph on the prefetch stream); config 4's form
                             use_graph=("split" if world > 1 else True) if args.graph else False,
                             train_table=args.train_table, gather_ahead=args.gather_ahead,
                             grad_sync_mode="bucketed" if args.grad_sync == "auto" else args.grad_sync)
        timers_on = not args.no_kernel_timers

        def barrier():
            dist.barrier()

        def reduce_max(x):
            t = torch.tensor([x], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        pre_steps = 0
        settle_how = None
        if not args.no_settle:
            set_phase("clock settle")
            settle_how = os.environ.get("CDML_SETTLE", "clone" if (world == 1 and not args.train_table) else "gemm")
            if settle_how == "clone":
                # 0.3 s of steps of a SECOND TrainStep of the same shape -- its own weights (another seed), optimizer state,
                # activations and gather buffers; it shares only the read-only catalogue and pair list -- then it is freed.
                # Why not GEMM launches alone (rounds 2-3): the chip takes ~100 ms of THIS mix of matrix and memory phases to
                # reach the clock it then holds; after 0.3 s of bare GEMMs (fp32 or bf16 alike) a 20-step run still read
                # 3-4 % above the 200-step rate (profiles/r04_settle_kinds.txt), after this it reads the same.  Nothing of the
                # measured job runs or changes before its W warm-up steps; `warmup_effective` says what did.
                clone = train.TrainStep(table, pairs, B, output_size=D, hidden_size=H, margin=MARGIN, mode=mode, optimizer="adam",
                                        base_learning_rate=0.01, seed=99, weight_seed=7, device=dev, precision=args.precision,
                                        gather_ahead=args.gather_ahead)
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.3:
                    for _ in range(20):
                        clone.step()
                    torch.cuda.synchronize(dev)
                del clone
                torch.cuda.empty_cache()
            else:
                settle_gpu(dev, precision=args.precision[:5])
        if world > 1 and args.grad_sync == "auto" and not args.train_table:
            # which gradient-sync form is faster HERE (link speed against kernel speed): a few steps of
            # each, max over ranks (so every rank picks the same one), before the warm-up steps
            set_phase("grad-sync probe")
            probe = {}
            for form in train.TrainStep.GRAD_SYNC_MODES:
                ts.grad_sync_mode = form
                for _ in range(2):
                    ts.step()
                torch.cuda.synchronize(dev)
                barrier()
                t0 = time.perf_counter()
                for _ in range(5):
                    ts.step()
                torch.cuda.synchronize(dev)
                barrier()
                probe[form + "_ms_per_step"] = round(reduce_max(time.perf_counter() - t0) / 5 * 1e3, 4)
                pre_steps += 7
            ts.grad_sync_mode = min(train.TrainStep.GRAD_SYNC_MODES, key=lambda f: probe[f + "_ms_per_step"])
            probe["picked"] = ts.grad_sync_mode
        set_phase("warm-up steps")
        comm_kt = None
        if world > 1:                                   # what the compute stream waits for
            comm_kt = KernelTimer()
            grad_sync.finish = comm_kt.wrap("allreduce_wait", grad_sync.finish)
            if ts.prefetch is not None:
                ts.prefetch.acquire = comm_kt.wrap("exchange_wait", ts.prefetch.acquire)
            comm_kt.on = True
        elapsed, kt, sampled, how = measure_job(ts, args.steps, args.warmup, dev, "h2" if h2 else None if x3 else bf16, timers_on,
                                                barrier if world > 1 else None)
        if comm_kt is not None:
            comm_kt.on = False
        elapsed_local = elapsed
        per_rank = None
        if world > 1:
            set_phase("max over ranks")
            elapsed = reduce_max(elapsed)
            # every rank's own facts in the one JSON line, so that the first real N > 1 run explains itself:
            # the flags are read BEFORE the check below raises on them -- an overflowing run still names its rank in the
            # failure message of check_inputs(), a clean one shows every rank's zeros here
            set_phase("per-rank facts")
            ov = 0 if exchange.overflow is None else int(exchange.overflow.item())
            ar_w, ex_w = comm_kt.mean_ms("allreduce_wait"), comm_kt.mean_ms("exchange_wait")
            mine = torch.tensor([rank, local_rank, elapsed_local / args.steps * 1e3, ov, int(ts.oob.item()),
                                 -1.0 if ar_w is None else ar_w, -1.0 if ex_w is None else ex_w,
                                 torch.cuda.memory_allocated(dev) / 2.0 ** 30], dtype=torch.float64,
                                device=dev if backend == "nccl" else "cpu")
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            per_rank = [{"rank": int(v[0]), "local_rank": int(v[1]), "ms_per_step": round(float(v[2]), 4),
                         "exchange_overflow_flag": int(v[3]), "pair_id_out_of_range_flag": int(v[4]),
                         "allreduce_exposed_ms": None if v[5] < 0 else round(float(v[5]), 4),
                         "exchange_exposed_ms": None if v[6] < 0 else round(float(v[6]), 4),
                         "hbm_allocated_gib": round(float(v[7]), 2)} for v in (t.cpu() for t in allv)]
        set_phase("checks")
        loss = ts.loss()                                 # (also reads the exchange-overflow / bad-id flags)
        assert np.isfinite(loss), "non-finite loss"
    except Exception as e:                               # noqa: BLE001 -- name the phase, exit non-zero
        sys.stderr.write("[bench rank %d] failed during %s: %r\n" % (rank, _PHASE["name"], e))
        raise

    if rank == 0:
        rpt = ts.rows_per_triplet
        R = B * rpt
        ms = elapsed / args.steps * 1e3
        if bf16:
            cfg_name = "config4" + (" per-GPU shape on 1 GPU" if world == 1 else "")
        elif world > 1:
            cfg_name = "config3"
        elif n_rows == 10000000 and B == 8192 and mode == "inbatch":
            cfg_name = "config3 per-GPU workload on 1 GPU (the whole 10 M-row catalogue in one HBM: the 1-GPU point of the scaling curve)"
        else:
            cfg_name = {"inbatch": "config1", "semihard": "config2", "uniform": "config1 size, reference negative rule"}[mode]
        out = {
            "metric": "triplets/sec", "value": round(world * B * args.steps / elapsed, 1),
            "unit": "triplets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("bf16 (fp16 table, f32 accumulate)" if bf16 else F16X2_DTYPE if h2 else "f32 (fp32 MFMA)" if not x3 else X3_DTYPE if x3 == 6 else
                      "f32 values as 3 bf16 planes, 3 plane products per f32 product on the bf16 MFMA (16-bit operands: NOT "
                      "an fp32 equivalent; measurement only)"),
            "data": "synthetic",
            "config": {"workload": "%s: %d videos x %d-d %s in HBM%s, %d hidden, %d-d embed, batch %d triplets/GPU "
                                   "(%d global), %s negatives (%d rows per triplet), margin %.1f, Adam, full step (sample+gather+fwd+loss+bwd+opt)"
                                   % (cfg_name, n_rows, F, "fp16" if bf16 else "fp32",
                                      "" if world == 1 else " row-sharded over %d GPUs" % world, H, D, B, world * B,
                                      {"inbatch": "in-batch", "uniform": "uniform random", "semihard": "semi-hard mined"}[mode],
                                      rpt, MARGIN),
                       "global_batch": world * B, "rows_per_triplet": rpt,
                       "rows_per_triplet_note": ("in-batch negatives: a triplet gathers and embeds 2 rows (anchor, positive; the negative "
                                                 "is another triplet's positive).  The reference's only negative rule is uniform (3 rows per "
                                                 "triplet, inputs.py:125-127): in rows embedded this value is value x 2/3 = %.1f "
                                                 "three-row triplets/s; the three-row step itself is timed as `reference_recipe` "
                                                 "(B = 1024) and `config4_per_gpu`" % (world * B * args.steps / elapsed * 2.0 / 3.0))
                                                if rpt == 2 else "uniform negatives: 3 rows per triplet, the reference's rule",
                       "value_three_row_equivalent": round(world * B * args.steps / elapsed * rpt / 3.0, 1),
                       "parallelism": "dp%d" % world + ("" if world == 1 else " row-sharded table, all-to-all rows + all-reduce grads"),
                       "hipgraph": ts.use_graph if ts.use_graph == "split" else bool(ts.use_graph), "trainable_table": bool(args.train_table),
                       "gather_steps_per_launch": ts.gather_ahead, "precision": args.precision},
            "loss": round(loss, 6),
        }
        if world > 1:
            # 1 -> N compares ONE workload only against this base: the bare N = 1 command times BASELINE config 1 (1 M rows,
            # batch 4096), the N > 1 commands config 3's per-GPU shape (10 M rows row-sharded, batch 8192 per GPU)
            out["scaling_base"] = {"workload": LIKE_FOR_LIKE_WORKLOAD, "precision": args.precision,
                                   "where": "the headline `value` of the N = 1 line (python bench.py --gpus 1; rounds 1-4: its "
                                            "`like_for_like` record): the 1-GPU triplets/s of the workload every rank of this "
                                            "line runs",
                                   "per_gpu_batch": B, "rows_global": n_rows}
            out["ranks_seen"] = dist.get_world_size()
            out["comm_backend"] = {"backend": dist.get_backend(), "launcher": "self" if os.environ.get("CDML_BENCH_PHASE_DIR") else "external",
                                   "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None,
                                   # what steers RCCL's algorithm / protocol / channel choice, as this process saw it (unset =
                                   # RCCL's own choice for the topology: ring over the xGMI mesh unless told otherwise)
                                   "env": {k: v for k, v in sorted(os.environ.items())
                                           if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "HSA_FORCE_FINE", "HIP_VISIBLE", "ROCR_VISIBLE"))},
                                   "algo": os.environ.get("NCCL_ALGO", "unset (RCCL decides)"),
                                   "proto": os.environ.get("NCCL_PROTO", "unset (RCCL decides)"),
                                   "channels": {"min": os.environ.get("NCCL_MIN_NCHANNELS"), "max": os.environ.get("NCCL_MAX_NCHANNELS")}}
            out["ranks"] = per_rank
            out["exchange"] = {"requests_per_rank": ts.R, "capacity_slots_per_peer": exchange.capacity(ts.R),
                               "capacity_factor": args.capacity_factor, "mean_share_per_peer": ts.R / world,
                               "overflow_flag_max": max(r["exchange_overflow_flag"] for r in per_rank),
                               "overflow_flag_bits": "1 = a peer segment was full (rows came back NaN), 2 = an id outside the catalogue"}
        if timers_on and h2:
            out.update(h2_records(kt, R, sampled, how, workload=wkey if world == 1 else None))
        elif timers_on:
            out.update(gemm_records(kt, R, bf16, sampled, how, world == 1, x3_products=x3, workload=wkey))
        if world == 1 and not args.train_table:
            set_phase("gather record")
            out["gather"] = gather_record(ts, mode, bf16, dev, workload=wkey)
            if "roofline" in out:
                # BASELINE's metric has an HBM half ("HBM GB/s on gather vs roofline"): carried INSIDE `roofline`, the object the
                # driver keeps, not only as a top-level extra
                gr = out["gather"]
                out["roofline"]["gather"] = {k: gr[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                                "bytes_per_row_moved", "bytes_per_row_algorithmic_8d",
                                                                "achieved_algorithmic_8d", "frac_algorithmic_8d", "launch_ms",
                                                                "rows_per_launch", "frac_is")}
        if world > 1:
            ar, exw = comm_kt.mean_ms("allreduce_wait"), comm_kt.mean_ms("exchange_wait")
            out["comm"] = {"allreduce_exposed_ms": None if ar is None else round(ar, 4),
                           "exchange_exposed_ms": None if exw is None else round(exw, 4),
                           "exchange_bytes": exchange.bytes_per_step(ts.R, ts.ws.x_hat, ts.table),
                           "allreduce_bytes": int(ts.layout.numel * 4),
                           "grad_sync": ts.grad_sync_mode, "grad_sync_probe": probe,
                           "exchange_capacity_factor": args.capacity_factor,
                           "note": "event pairs around GradSync.finish / Prefetcher.acquire on the compute stream (every "
                                   "step of the run): what the step waits for, not the collectives' own duration"}
        step_flops = R * (2.0 * F * H + 2.0 * H * D) + R * (2.0 * F * H + 4.0 * H * D)
        out["step_tflops"] = round(step_flops / (elapsed / args.steps) / 1e12, 2)
        out["warmup_effective"] = {"steps_of_this_job_before_the_timed_region": args.warmup + pre_steps,
                                   "of_which_grad_sync_probe": pre_steps,
                                   "settle_loop_s": 0.0 if args.no_settle else 0.3,
                                   "settle_kind": None if args.no_settle else settle_how,
                                   "note": "settle_kind clone: 0.3 s of steps of a SECOND TrainStep of the same shape on its own "
                                           "buffers (own weights, optimizer state, activations; only the read-only catalogue and "
                                           "pair list are shared), freed before the warm-up steps; gemm: 0.3 s of GEMM launches on "
                                           "scratch buffers.  No state of the measured job is touched: the chip needs ~100 ms of "
                                           "this kind of work to reach the clock it then holds (a 20-step run from idle reads "
                                           "3-4 % above the 200-step rate)"}
        out["order"] = ("%swarm-up + timed steps, then the secondary records, cpu_baseline last"
                        % ("" if args.no_settle else "0.3 s settle loop (%s), " % settle_how))

        # ---- secondary records (after the headline; the failure of one must not cost the line) ----
        run_extras = default_job and not args.no_extras and not args.train_table and x3 in (0, 6) and not h2
        if run_extras:
            del ts
            torch.cuda.empty_cache()
            other = "f32" if x3 else "f32x3"
            other_name = "f32_mfma" if x3 else "f32x3"
            want = set((args.extras or "config1,config2_semihard,dp_form_one_gpu,train_table,config4_per_gpu,reference_recipe,fusion_resnet,"
                                       "predict,knn,data_learnable,f16x2," + other_name).split(","))
            n_s, n_w = min(args.steps, 30), min(max(args.warmup, 3), 5)
            keep = {"t10": table, "p10": pairs}            # the headline's own catalogue serves dp_form_one_gpu
            del table, pairs
            # rounds 1-4 reported this workload as the `like_for_like` record beside a config-1 headline
            out["like_for_like"] = {"see": "the headline of this line IS that workload since round 5", "value": out["value"],
                                    "ms_per_step": out["ms_per_step"], "workload": LIKE_FOR_LIKE_WORKLOAD}

            def attempt(name, fn):
                if name not in want:
                    return
                set_phase("secondary: " + name)
                try:
                    out[name] = fn()
                except Exception as e:                   # noqa: BLE001
                    out[name] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
            if x3:
                attempt("f16x2", lambda: rec_f16x2(dev, args, n_s, n_w, keep["t10"], keep["p10"], B, mode))
            attempt("dp_form_one_gpu", lambda: rec_dp_form(dev, args, n_s, n_w, keep))
            attempt("train_table", lambda: rec_train_table(dev, args, n_s, n_w, keep))     # (last user of the 10 M-row table: it trains it)
            keep.clear()
            torch.cuda.empty_cache()
            # the 1 M-row catalogue of configs 1 and 2 (and of the reference's own recipe)
            t1 = engine.FeatureTable.synthetic(1000000, F, seed=0, device=dev)
            p1 = torch.from_numpy(synth_pairs(1000000, 333333, seed=0)).to(dev)
            attempt("config1", lambda: rec_config1(dev, args, n_s, n_w, t1, p1))
            attempt("config2_semihard", lambda: rec_config2(dev, args, n_s, n_w, t1, p1))
            attempt(other_name, lambda: rec_other_fp32_path(dev, args, n_s, n_w, t1, p1, 4096, "inbatch", other))
            attempt("reference_recipe", lambda: rec_reference_recipe(dev, args, n_s, n_w, t1))
            attempt("data_learnable", lambda: rec_learnable(dev, args, n_s, n_w, 4096))
            del t1, p1
            torch.cuda.empty_cache()
            attempt("fusion_resnet", lambda: rec_fusion(dev, args, n_s, n_w))
            attempt("config4_per_gpu", lambda: rec_config4(dev, args, n_s, n_w))
            attempt("predict", lambda: {"f32": rec_predict(dev, "f32"), "f32x3": rec_predict(dev, "f32x3"),
                                        "f16x2": rec_predict(dev, "f16x2"), "bf16": rec_predict(dev, "bf16")})
            attempt("knn", lambda: dict(rec_knn(dev), f16x2=rec_knn(dev, precision="f16x2")))
        if world == 1 and not args.no_cpu_baseline:
            set_phase("cpu_baseline")
            out["cpu_baseline"] = cpu_baseline(args.cpu_baseline_full)
        print(json.dumps(out), file=result_out, flush=True)
    if world > 1:
        set_phase("shutdown")
        dist.barrier()
        dist.destroy_process_group()
    set_phase("done")


if __name__ == "__main__":
    main()
