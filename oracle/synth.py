"""Oracle: synthetic inputs shaped like the reference's imitation_data.py.
TEST INFRASTRUCTURE (see oracle/__init__.py).

  * features_numpy  -- imitation_data.py:41-53 exactly (legacy global-RNG stream,
    float64 U[0,1) rounded to 8 dp), cast to fp32 the way the reference stores
    features.npy (online_data.py:62).  Pinned by golden fixture G2.
  * features_philox -- spec of the build's on-device table generator (a 6 GB
    table is generated in HBM, not on the host): word j of row r is
    philox4x32_10(ctr=(j>>2, r, 0, TABLE_TAG), key=seed)[j&3]; value =
    (word >> 8) * 2^-24, i.e. U[0,1) on the fp32 grid (the 8-dp rounding of the
    reference is finer than fp32 resolution over most of [0,1)).
  * cowatch_pairs   -- imitation_data.py:56-85 shape: users with U{low..high}
    uniformly chosen videos (with replacement), adjacent duplicates removed
    (online_data.py:114-116), consecutive pairs (parse_data.py:179-190), one
    global shuffle (parse_data.py:206).
"""
import numpy as np
from .sampler import philox4x32_10

TABLE_TAG = 0x7AB1E000


def features_numpy(num_feature, feature_size, seed, decimals=8):
    np_state = np.random.RandomState(seed)     # == np.random.seed(seed) stream
    f = np.around(np_state.random_sample((num_feature, feature_size)), decimals)
    return f                                   # float64, like the reference


def features_philox(row0, n_rows, feature_size, seed):
    r = (row0 + np.arange(n_rows, dtype=np.uint64))[:, None]
    j = np.arange(feature_size, dtype=np.uint64)[None, :]
    out = philox4x32_10((j >> np.uint64(2), r & np.uint64(0xFFFFFFFF),
                         r >> np.uint64(32), TABLE_TAG),
                        (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    w = np.choose((j & np.uint64(3)).astype(np.int64), out)
    return ((w >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24))


def cowatch_pairs(n_videos, n_users, seed, low=2, high=30):
    rng = np.random.RandomState(seed)
    lens = rng.randint(low, high + 1, size=n_users)
    vids = rng.randint(0, n_videos, size=int(lens.sum()))
    ends = np.cumsum(lens)
    last = np.zeros(len(vids), dtype=bool)
    last[ends - 1] = True
    keep = ~last[:-1] & (vids[:-1] != vids[1:])
    pairs = np.stack([vids[:-1][keep], vids[1:][keep]], axis=1)
    rng.shuffle(pairs)
    return pairs.astype(np.int32)
