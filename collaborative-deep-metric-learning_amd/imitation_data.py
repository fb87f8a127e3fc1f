"""Synthetic test data behind the reference's generator API (imitation_data.py:14-110) --
BASELINE config 0's input ("imitation_data.py synthetic") -- plus the device-side generators
the large configurations use (a 6-60 GB catalogue is not produced on the host).

Host functions keep the reference's names, arguments, dtypes and random streams (legacy global
``np.random`` / ``random`` state, so ``np.random.seed(s)`` reproduces the reference's arrays:
pinned by tests/golden/imitation_features_seed0.npz):
    gen_features, gen_triplets, gen_watched_guids, gen_all_watched_guids,
    gen_unique_id_array, arrays_to_dict
Device functions:
    device_features   Philox catalogue in HBM (``engine.FeatureTable.synthetic``)
    cowatch_pairs     watched lists -> adjacent-duplicate removal -> consecutive pairs -> one
                      shuffle (online_data.py:114-116, parse_data.py:188-190,206) as int32 [P,2]
"""
import random

import numpy as np


def gen_unique_id_array(low, high, size, dtype=None):
    """``size`` distinct integers from [low, high] in random order (imitation_data.py:14-38)."""
    if low > high:
        raise ValueError("low is greater than high")
    if size > high - low + 1:
        raise ValueError("size is greater than the number of available ids")
    if size < 0:
        raise ValueError("size is negative")
    ids = np.array(random.sample(range(low, high + 1), size))
    return ids.astype(dtype) if dtype else ids


def gen_features(num_feature, feature_size, decimals=8):
    """float64 [num_feature, feature_size], U[0,1) rounded to ``decimals`` (imitation_data.py:41-53)."""
    return np.around(np.random.random((num_feature, feature_size)), decimals)


def gen_watched_guids(guids, low, high):
    """One watch history: randint(low, high) picks WITH replacement (imitation_data.py:56-68)."""
    return np.random.choice(guids, random.randint(low, high)).tolist()


def gen_all_watched_guids(guids, num_cowatch, low=2, high=30):
    """``num_cowatch`` watch histories (imitation_data.py:71-85)."""
    return [gen_watched_guids(guids, low, high) for _ in range(num_cowatch)]


def gen_triplets(batch_size, feature_size):
    """float64 [batch_size, 3, feature_size] of gen_features values (imitation_data.py:88-93)."""
    return np.reshape(gen_features(batch_size * 3, feature_size), (batch_size, 3, feature_size))


def arrays_to_dict(array_1d, array_2d):
    """{array_1d[i]: array_2d[i]} -- e.g. guid -> feature (imitation_data.py:96-110)."""
    if len(array_1d) != len(array_2d):
        raise ValueError("the arrays must have the same number of rows")
    return {k: v for k, v in zip(array_1d, array_2d)}


def device_features(n_rows, feature_size, seed=0, device="cuda:0", row0=0, n_rows_global=None):
    """The catalogue generated in HBM by the HIP fill kernel (same distribution, its own
    counter-based stream: row r is the same on every shard and for every shard count)."""
    from .engine import FeatureTable
    return FeatureTable.synthetic(n_rows, feature_size, seed, device, row0=row0, n_rows_global=n_rows_global)


def cowatch_pairs(n_videos, n_users, seed=0, low=2, high=30):
    """Co-watch pairs of ``n_users`` random watch histories over ``n_videos`` row ids."""
    rng = np.random.RandomState(seed)
    lens = rng.randint(low, high + 1, size=n_users)
    vids = rng.randint(0, n_videos, size=int(lens.sum()))
    last = np.zeros(len(vids), dtype=bool)
    last[np.cumsum(lens) - 1] = True
    keep = ~last[:-1] & (vids[:-1] != vids[1:])
    pairs = np.stack([vids[:-1][keep], vids[1:][keep]], axis=1)
    return pairs[rng.permutation(len(pairs))].astype(np.int32)
