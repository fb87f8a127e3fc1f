#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference's
own importable (pure-numpy) modules from /root/reference.

Run in the build container only (``python tests/golden/make_golden.py``); the
reference does not exist on the GPU box and nothing at test time reads it.
Only data (inputs + expected outputs) is written -- never reference source.

What can be imported: imitation_data.py and utils.py as
they are; parse_data.py and evaluate.py use TensorFlow for nothing but
``tensorflow.logging`` (parse_data.py:10,13; evaluate.py:10,12), so they import
once ``sys.modules['tensorflow']`` exposes a logging shim.  models.py / losses.py
/ train.py need real TF 1.13 graph ops and are NOT imported or emulated: the
tower has no reference golden (parity unpinned) and the loss golden is the
hand-derived known answer of tests/test_losses.py's input.
"""
import json
import logging as _pylogging
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    tf = types.ModuleType("tensorflow")
    lg = types.ModuleType("tensorflow.logging")
    for name in ("debug", "info", "warning", "error"):
        setattr(lg, name, getattr(_pylogging, name))
    lg.DEBUG = _pylogging.DEBUG
    lg.set_verbosity = lambda *_a, **_k: None
    tf.logging = lg
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow.logging"] = lg
    sys.path.insert(0, REF)
    import imitation_data
    import parse_data
    import evaluate
    return imitation_data, parse_data, evaluate


def main():
    imitation_data, parse_data, evaluate = _import_reference()

    # G1 -- sampler: reference negative rule under the legacy global RNG
    rng = np.random.RandomState(99)
    for seed in (0, 1234):
        for n_rows in (3, 10000):
            pairs = rng.randint(0, n_rows, size=(64, 2))
            pairs = pairs[pairs[:, 0] != pairs[:, 1]]
            np.random.seed(seed)
            neg_iter = parse_data.yield_negative_index(n_rows, putback=True)
            trip = [parse_data.combine_cowatch_neg([int(a), int(p)], neg_iter)
                    for a, p in pairs]
            np.savez(os.path.join(OUT, f"sampler_ref_seed{seed}_n{n_rows}.npz"),
                     pairs=pairs.astype(np.int64),
                     triplets=np.asarray(trip, dtype=np.int64),
                     seed=seed, n_rows=n_rows)
    np.random.seed(1234)
    it = parse_data.yield_negative_index(10000, putback=True)
    stream = [int(next(it)) for _ in range(8)]

    # G2 -- synthetic feature distribution (imitation_data.gen_features)
    np.random.seed(0)
    feats = imitation_data.gen_features(8, 1500)
    np.savez(os.path.join(OUT, "imitation_features_seed0.npz"), features=feats)
    np.random.seed(3)
    trip = imitation_data.gen_triplets(4, 16)

    # G3 -- eval metric (next-row N1): Evaluation.mean_dist on seeded inputs
    r = np.random.RandomState(5)
    features = r.random_sample((50, 12)).astype(np.float32)
    cowatches = [[int(a), int(b)] for a, b in r.randint(0, 50, size=(30, 2))]
    ev = evaluate.Evaluation(features, cowatches)
    emb = r.standard_normal((len(ev.features), 8)).astype(np.float32)
    md = float(ev.mean_dist(emb, ev.cowatches))
    np.savez(os.path.join(OUT, "evaluate_mean_dist.npz"),
             features=features, cowatches=np.asarray(cowatches, dtype=np.int64),
             eval_features=np.asarray(ev.features, dtype=np.float32),
             eval_cowatches=np.asarray(ev.cowatches, dtype=np.int64),
             embeddings=emb, mean_dist=md)

    # G4 -- loss known answer: the fixed input of tests/test_losses.py:13-18 with
    # the values derived by hand from losses.py:32-38 (the reference asserts none)
    # G5 -- co-watch mining known answers of tests/test_parse_data.py:151-153
    cow = parse_data.get_all_cowatch([[0], [1, 2], [3, 4, 5, 6], [], [7, 8, 9]])
    kat = {
        "neg_stream_seed1234_n10000": stream,
        "gen_triplets_shape": list(trip.shape),
        "loss_input": [[[1, 1], [2, 2], [5, 5]], [[1, 1], [5, 5], [2, 2]],
                       [[1, 1], [2, 2], [5, 5]], [[1, 1], [5, 5], [2, 2]],
                       [[1, 1], [5, 5], [2, 2]]],
        "loss_pos_dist": [2, 32, 2, 32, 32],
        "loss_neg_dist": [32, 2, 32, 2, 2],
        "loss_margin_0.1": {"hinge_dist": [0, 30.1, 0, 30.1, 30.1],
                            "hinge_loss": 18.06},
        "loss_margin_0.8": {"hinge_dist": [0, 30.8, 0, 30.8, 30.8],
                            "hinge_loss": 18.48},
        "get_all_cowatch_len": len(cow),
        "get_all_cowatch_sorted": sorted([list(map(int, c)) for c in cow]),
    }
    # G6 -- co-watch graph and selection (next-row N3): the reference's own get_cowatch_graph /
    # select_cowatch on seeded watch histories (adjacent duplicates removed, as the ETL does
    # before pairing, online_data.py:114-116)
    r = np.random.RandomState(7)
    histories = []
    for _ in range(400):
        w = r.randint(0, 60, size=r.randint(2, 12)).tolist()
        histories.append([int(v) for i, v in enumerate(w) if i == 0 or v != w[i - 1]])
    np.random.seed(7)
    cowatches = [[int(a), int(b)] for a, b in parse_data.get_all_cowatch(histories)]
    graph, cowatches = parse_data.get_cowatch_graph(cowatches)
    edges = sorted(tuple(map(int, e.split(","))) for e in graph)
    counts = [graph["%d,%d" % e] for e in edges]
    sel = {t: parse_data.select_cowatch(graph, t, cowatches) for t in (1, 2, 3, 5)}
    np.random.seed(8)
    uniq3 = parse_data.select_cowatch(graph, 3, unique=True)
    np.savez(os.path.join(OUT, "cowatch_graph_seed7.npz"),
             cowatches=np.asarray(cowatches, dtype=np.int64), edges=np.asarray(edges, dtype=np.int64),
             counts=np.asarray(counts, dtype=np.int64),
             **{"select_t%d" % t: np.asarray(v, dtype=np.int64).reshape(-1, 2) for t, v in sel.items()},
             unique_t3_sorted=np.asarray(sorted(tuple(sorted(map(int, c))) for c in uniq3), dtype=np.int64))

    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(kat, f, indent=1)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
