#!/usr/bin/env python3
"""hipGraph replay against eager launches of the SAME step, on one box, in one process.

    python tools/graph_vs_eager.py [--blocks 4] [--steps 200]            # alternating blocks, both workloads
    python tools/graph_vs_eager.py --trace eager|graph --workload x3|bf16 # one form only, 60 steps (run under
                                                                          # rocprofv3 --kernel-trace, program after `--`)
    python tools/graph_vs_eager.py --gaps trace.csv                       # per-step kernel time and idle gaps of a trace

Workloads: "x3" = BASELINE config 1 at precision f32x3 (1 M rows, B = 4096, in-batch); "bf16" = config 4's per-GPU shape
(10 M-row fp16 table, B = 8192, uniform negatives).  For each the eager TrainStep and the use_graph=True TrainStep are
built on the same table and pair list; blocks alternate eager, graph, eager, graph ... so neither owns the warm half.
Also timed: what the HOST spends enqueuing a step in each form (no device sync inside the loop)."""
import argparse
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(workload, dev, graph):
    import torch
    import bench
    from cdml_amd import engine, engine_bf16, train
    if workload == "bf16":
        t = engine_bf16.FeatureTableF16.synthetic(10000000, bench.F, seed=0, device=dev)
        p = torch.from_numpy(bench.synth_pairs(10000000, 600000, seed=0)).to(dev)
        mk = lambda g: train.TrainStep(t, p, 8192, output_size=bench.D, hidden_size=bench.H, margin=bench.MARGIN, mode="uniform",
                                       optimizer="adam", base_learning_rate=0.01, device=dev, precision="bf16", use_graph=g)
    else:
        t = engine.FeatureTable.synthetic(1000000, bench.F, seed=0, device=dev)
        p = torch.from_numpy(bench.synth_pairs(1000000, 333333, seed=0)).to(dev)
        mk = lambda g: train.TrainStep(t, p, 4096, output_size=bench.D, hidden_size=bench.H, margin=bench.MARGIN, mode="inbatch",
                                       optimizer="adam", base_learning_rate=0.01, device=dev, precision="f32x3", use_graph=g)
    return [mk(g) for g in graph]


def block(ts, steps, dev):
    import torch
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3, t_host / steps * 1e3


def gaps(path):
    """Per step of a rocprofv3 kernel trace (one stream of cdml kernels): sum of kernel durations, sum of the idle gaps
    between consecutive kernels, by splitting the trace at every sampler+gather / first GEMM of a step."""
    rows = []
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if "cdml" not in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
    rows.sort()
    if not rows:
        print("no cdml kernels in", path)
        return
    # steady state: the last 60 % of the launches
    rows = rows[int(len(rows) * 0.4):]
    busy = sum(e - s for s, e, _ in rows)
    idle = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
    span = rows[-1][1] - rows[0][0]
    big = sorted(idle)[-5:]
    print("%s: %d kernels, span %.3f ms, kernel time %.3f ms (%.1f %%), idle between kernels %.3f ms (%.1f %%): mean gap %.2f us, "
          "median %.2f us, five largest %s us"
          % (os.path.basename(path), len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(idle) / 1e6,
             100.0 * sum(idle) / span, sum(idle) / len(idle) / 1e3, sorted(idle)[len(idle) // 2] / 1e3,
             [round(b / 1e3, 1) for b in big]))
    per = {}
    for s, e, n in rows:
        k = n.split("(")[0][:70]
        per.setdefault(k, []).append(e - s)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("    %-72s n=%4d mean %.1f us" % (k, len(v), sum(v) / len(v) / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--trace", default=None, choices=["eager", "graph"])
    ap.add_argument("--workload", default=None, choices=["x3", "bf16"])
    ap.add_argument("--gaps", default=None)
    a = ap.parse_args()
    if a.gaps:
        gaps(a.gaps)
        return
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.trace:
        (ts,) = build(a.workload or "x3", dev, [a.trace == "graph"])
        for _ in range(100):
            ts.step()
        torch.cuda.synchronize(dev)
        print("traced 100 steps of the %s form (%s)" % (a.trace, a.workload or "x3"))
        return
    for wl in ([a.workload] if a.workload else ["x3", "bf16"]):
        te, tg = build(wl, dev, [False, True])
        for ts in (te, tg):
            for _ in range(30):
                ts.step()
        torch.cuda.synchronize(dev)
        res = {"eager": [], "graph": []}
        for b in range(a.blocks):
            for name, ts in (("eager", te), ("graph", tg)):
                res[name].append(block(ts, a.steps, dev))
        for name in ("eager", "graph"):
            print("%s %s: ms/step per block %s | host enqueue ms/step %s | mean %.4f"
                  % (wl, name, [round(r[0], 4) for r in res[name]], [round(r[1], 4) for r in res[name]],
                     sum(r[0] for r in res[name]) / len(res[name])), flush=True)
        del te, tg
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
