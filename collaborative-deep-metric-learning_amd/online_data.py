"""On-disk formats of a reference dataset directory (reference online_data.py),
so a directory produced by the reference's ETL is consumed unchanged -- and a
synthetic one can be written for tests:

  features.npy        float32 [N,F], row id = encoded video index   (online_data.py:87-93,205-213)
  encode_map.json     guid -> row id,  decode_map.json  row id -> guid (online_data.py:214-222)
  cowatches.eval/.test, x??.train   ASCII "a,p\\n" pairs             (online_data.py:256-295)

Only the formats are mirrored; the ETL itself (text parsing, co-watch mining) is
one-off host work and out of scope.
"""
import glob
import json
import os

import numpy as np


def read_features_npy(filename):
    """online_data.py:87-93."""
    return np.load(filename)


def load_cowatches(filename):
    """List of [a, p] index pairs; malformed lines are skipped like the reference
    (online_data.py:125-142)."""
    cowatches = []
    with open(filename, "r") as f:
        for line in f:
            ids = line.strip().split(",")
            try:
                cowatches.append([int(ids[0]), int(ids[1])])
            except (ValueError, IndexError):
                continue
    return cowatches


def write_features(features, encode_map=None, decode_map=None, save_dir=""):
    """online_data.py:205-229 (both maps are written when encode_map is given)."""
    os.makedirs(save_dir, exist_ok=True)
    np.save(os.path.join(save_dir, "features.npy"), np.asarray(features, dtype=np.float32))
    if encode_map is not None:
        with open(os.path.join(save_dir, "encode_map.json"), "w") as f:
            json.dump(encode_map, f, ensure_ascii=False)
        with open(os.path.join(save_dir, "decode_map.json"), "w") as f:
            json.dump(decode_map, f, ensure_ascii=False)
    return True


def write_cowatches(cowatches, save_dir="", split_num=4, eval_num=100000, test_num=100000):
    """online_data.py:256-295: first eval_num pairs -> cowatches.eval, next test_num ->
    cowatches.test (15 % each when they would exceed 30 %), the rest split into
    ``split_num`` files of ceil(len(cowatches)/split_num) lines named like GNU split's
    output (xaa.train, xab.train, ...)."""
    os.makedirs(save_dir, exist_ok=True)
    n = len(cowatches)
    if eval_num + test_num > 0.3 * n:
        eval_num = int(n * 0.15)
        test_num = int(n * 0.15)

    def dump(path, rows):
        with open(path, "w") as f:
            for a, p in rows:
                f.write("%d,%d\n" % (a, p))
    if eval_num:
        dump(os.path.join(save_dir, "cowatches.eval"), cowatches[:eval_num])
    if test_num:
        dump(os.path.join(save_dir, "cowatches.test"), cowatches[eval_num:eval_num + test_num])
    train = cowatches[eval_num + test_num:]
    split_num = 1 if split_num < 1 else int(split_num)
    row_cnt = n // split_num if n % split_num == 0 else n // split_num + 1   # the reference sizes parts by ALL pairs
    for i in range(0, len(train), row_cnt):
        j = i // row_cnt
        name = "x" + chr(ord("a") + j // 26) + chr(ord("a") + j % 26) + ".train"
        dump(os.path.join(save_dir, name), train[i:i + row_cnt])
    return True


def load_dataset(train_dir):
    """Everything train.py:339-349 reads from FLAGS.train_dir."""
    out = {"train_files": sorted(glob.glob(os.path.join(train_dir, "*.train"))),
           "features": read_features_npy(os.path.join(train_dir, "features.npy"))}
    for name in ("eval", "test"):
        p = os.path.join(train_dir, "cowatches." + name)
        out[name + "_cowatches"] = load_cowatches(p) if os.path.exists(p) else []
    dm = os.path.join(train_dir, "decode_map.json")
    out["decode_map"] = json.load(open(dm)) if os.path.exists(dm) else None
    return out
