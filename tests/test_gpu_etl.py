"""Co-watch graph statistics on the device (parse_data.py:221-289) through the C
ABI: bit-exact against the reference's own outputs (fixture G6) and against oracle/etl.py on
larger inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import etl

pytestmark = pytest.mark.gpu


def test_cowatch_graph_and_select_match_reference_fixture(gpu, golden_dir):
    from cdml_amd import parse_data
    g = np.load(os.path.join(golden_dir, "cowatch_graph_seed7.npz"))
    cow = g["cowatches"].astype(np.int32)
    edges, counts = parse_data.cowatch_graph(cow, device=gpu)
    np.testing.assert_array_equal(edges.cpu().numpy(), g["edges"])
    np.testing.assert_array_equal(counts.cpu().numpy(), g["counts"])
    for t in (1, 2, 3, 5):
        got = parse_data.select_cowatch(cow, t, device=gpu).cpu().numpy()
        np.testing.assert_array_equal(got, g["select_t%d" % t])           # same pairs, same order
    got = parse_data.select_cowatch(cow, 3, unique=True, device=gpu).cpu().numpy()
    np.testing.assert_array_equal(got, g["unique_t3_sorted"])
    with pytest.raises(RuntimeError):
        parse_data.cowatch_graph(np.array([[1, 2], [4, 4], [2, 1]], dtype=np.int32), device=gpu)
    assert parse_data.select_cowatch(np.zeros((0, 2), np.int32), 2, device=gpu).shape == (0, 2)


@pytest.mark.parametrize("P,n_ids", [(1, 10), (1000, 30), (300000, 5000), (2000000, 1000000)])
def test_cowatch_select_vs_oracle(gpu, P, n_ids):
    rng = np.random.RandomState(P % 1000)
    a = rng.randint(0, n_ids, size=P)
    b = (a + 1 + rng.randint(0, n_ids - 1, size=P)) % n_ids                # never equal to a
    cow = np.stack([a, b], 1).astype(np.int32)
    from cdml_amd import parse_data
    edges, counts = parse_data.cowatch_graph(cow, device=gpu)
    we, wc = etl.cowatch_graph(cow)
    np.testing.assert_array_equal(edges.cpu().numpy(), we)
    np.testing.assert_array_equal(counts.cpu().numpy(), wc)
    for t, uniq in ((2, False), (3, False), (2, True)):
        got = parse_data.select_cowatch(torch.as_tensor(cow).to(gpu), t, unique=uniq, device=gpu).cpu().numpy()
        np.testing.assert_array_equal(got, etl.select_cowatch(cow, t, unique=uniq))


def test_get_all_cowatch_host():
    from cdml_amd import parse_data
    cow = parse_data.get_all_cowatch([[0], [1, 2], [3, 4, 5, 6], [], [7, 8, 9]], seed=0)
    assert sorted(map(tuple, cow.tolist())) == [(1, 2), (3, 4), (4, 5), (5, 6), (7, 8), (8, 9)]   # known_answers.json
