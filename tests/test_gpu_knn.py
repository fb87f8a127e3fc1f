"""kNN export (faiss_knn.py:82-131) -- HIP brute force through the
C ABI against the fp64 exact search of oracle/knn.py."""
import numpy as np
import pytest
import torch

from oracle import knn as oknn

pytestmark = pytest.mark.gpu

TOL = 1e-5    # squared-L2 distance of unit vectors, fp32 MFMA vs fp64


def check(D, I, Dref, Iref, dfull):
    k = Dref.shape[1]
    finite = np.isfinite(Dref)
    assert np.array_equal(I[~finite], Iref[~finite])          # -1 where the catalogue runs out
    assert np.abs(D[finite] - Dref[finite]).max() <= TOL
    with np.errstate(invalid="ignore"):
        assert (np.diff(D, axis=1)[finite[:, 1:]] >= 0).all() # ascending
    for r in range(D.shape[0]):
        if np.array_equal(I[r], Iref[r]):
            continue
        # ids may differ only where the exact distances are within the tolerance of each other
        n = int(finite[r].sum())
        assert len(set(I[r, :n])) == n
        assert np.abs(dfull[r, I[r, :n]] - Dref[r, :n]).max() <= 2 * TOL


def embeddings(n, D, seed):
    rng = np.random.RandomState(seed)
    return rng.randn(n, D).astype(np.float32)


PLANE_FORMS = ["f32x3", "f16x2"]      # three bf16 planes x six products | two fp16 planes x three products (round 6)


@pytest.mark.parametrize("precision", PLANE_FORMS)
@pytest.mark.parametrize("n,D,k", [(1000, 256, 51), (333, 48, 128), (20, 256, 51), (64, 32, 1)])
def test_self_knn_matches_exact_search(n, D, k, precision):
    from cdml_amd import knn
    e = embeddings(n, D, 0)
    Dg, Ig = knn.calc_knn(e.copy(), nearest_num=k, precision=precision)
    Dr, Ir, dfull = oknn.calc_knn_exact(e, nearest_num=k)
    assert Dg.dtype == np.float32 and Ig.dtype == np.int64 and Dg.shape == (n, k)
    check(Dg, Ig, Dr, Ir, dfull)
    assert (Ig[:, 0] == np.arange(n)).all()                    # the query is its own nearest neighbour


def test_separate_queries_many_blocks_and_ties():
    from cdml_amd import knn
    base = embeddings(20000, 256, 1)
    base[5000:5016] = base[17]                                  # exact duplicates: ties ordered by id
    q = np.concatenate([embeddings(300, 256, 2), base[17:18]])
    D, I = knn.knn_search(torch.from_numpy(base), torch.from_numpy(q), 81, q_block=128, b_block=4096)
    Dr, Ir, dfull = oknn.calc_knn_exact(base, q, 81)
    check(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, dfull)
    got = I[-1, :17].cpu().numpy()
    assert sorted(got.tolist()) == [17] + list(range(5000, 5016))


def test_unnormalised_distances():
    from cdml_amd import knn
    base, q = embeddings(500, 64, 3), embeddings(40, 64, 4)
    D, I = knn.calc_knn(base, q, nearest_num=10, l2_norm=False)
    Dr, Ir, dfull = oknn.calc_knn_exact(base, q, 10, l2_norm=False)
    finite = np.isfinite(Dr)
    assert np.abs(D - Dr)[finite].max() <= 1e-4 * Dr.max()     # |x|^2 ~ 64: relative tolerance
    assert (I == Ir).mean() > 0.99


def test_bad_k_is_refused():
    from cdml_amd import knn
    with pytest.raises(ValueError):
        knn.calc_knn(embeddings(10, 32, 0), nearest_num=129)


@pytest.mark.parametrize("precision", PLANE_FORMS)
def test_filter_epilogue_matches_exact_search_and_falls_back_on_overflow(precision):
    """Round 6: everything after the first block of the catalogue goes through ONE launch of the plane GEMM whose epilogue
    appends the elements within a query's k-th best distance to its candidate list (no score matrix).  (i) 70 000 rows,
    separate queries, exact (D, I) against the fp64 oracle, identical to the score-block form; (ii) a catalogue laid out
    AGAINST the filter -- the rows after the first block are all closer than anything in it, so every query's list
    overflows -- still returns the exact answer (the search falls back to score blocks: nothing is silently dropped)."""
    from cdml_amd import knn
    base = embeddings(70000, 64, 5)
    base[40000:40008] = base[123]                               # exact duplicates across the block boundary: ties by id
    q = np.concatenate([embeddings(200, 64, 6), base[123:124]])
    D, I = knn.knn_search(torch.from_numpy(base), torch.from_numpy(q), 51, precision=precision)
    Dr, Ir, dfull = oknn.calc_knn_exact(base, q, 51)
    check(D.cpu().numpy(), I.cpu().numpy(), Dr, Ir, dfull)
    assert sorted(I[-1, :9].cpu().tolist()) == [123] + list(range(40000, 40008))
    D2, I2 = knn.knn_search(torch.from_numpy(base), torch.from_numpy(q), 51, fused=False, precision=precision)
    assert torch.equal(I, I2) and torch.equal(D, D2)
    # queries and catalogue in several chunks (each merged, the thresholds tightened in between): the same answer
    D4, I4 = knn.knn_search(torch.from_numpy(base), torch.from_numpy(q), 51, q_block=64, q_chunk=128, c_chunk=8192, precision=precision)
    assert torch.equal(I, I4) and torch.equal(D, D4)
    rng = np.random.RandomState(7)
    centre = rng.randn(64).astype(np.float32)
    far = rng.randn(33000, 64).astype(np.float32)               # the first block: unrelated directions
    near = centre + 0.05 * rng.randn(37000, 64).astype(np.float32)   # the rest: a tight cluster around the queries
    base2 = np.concatenate([far, near])
    q2 = centre + 0.05 * rng.randn(50, 64).astype(np.float32)
    D3, I3 = knn.knn_search(torch.from_numpy(base2), torch.from_numpy(q2), 20, precision=precision)
    Dr3, Ir3, dfull3 = oknn.calc_knn_exact(base2, q2, 20)
    check(D3.cpu().numpy(), I3.cpu().numpy(), Dr3, Ir3, dfull3)
    assert (I3.cpu().numpy() >= 33000).all()


@pytest.mark.parametrize("precision", PLANE_FORMS)
def test_self_knn_at_the_reference_catalogue_size(precision):
    """the export at the reference's own scale -- doc_location = 343455 embeddings (faiss_knn.py:389), 256-d,
    nearest_num = 51 -- against a blocked EXACT search in fp64 on the device (torch: 2 048 queries x the whole catalogue
    per block).  Every returned neighbour's true distance lies within tolerance of the exact k-th distance (ids may swap
    only between candidates the fp32 arithmetic cannot tell apart), distances ascending and equal to the exact ones to
    TOL, the query its own first neighbour."""
    from cdml_amd import knn
    dev = torch.device("cuda:0")
    n, D, k = 343455, 256, 51
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    e = torch.randn(n, D, device=dev, generator=g)
    Dg, Ig = knn.knn_search(e, e, k, precision=precision)
    torch.cuda.synchronize()
    assert Dg.shape == (n, k) and Ig.dtype == torch.int64
    assert bool((Ig[:, 0] == torch.arange(n, device=dev)).all())
    assert bool((Dg[:, 1:] >= Dg[:, :-1]).all())
    b = e.double()
    b = b / b.norm(dim=1, keepdim=True).clamp_min(1e-6)
    bsq = (b * b).sum(1)
    worst_d = worst_set = 0.0
    exact_ids = 0
    for q0 in range(0, n, 2048):
        q = b[q0:q0 + 2048]
        d = (bsq[q0:q0 + 2048, None] + bsq[None, :] - 2.0 * (q @ b.t())).clamp_min(0.0)
        Dr, Ir = torch.topk(d, k, dim=1, largest=False, sorted=True)
        got_d, got_i = Dg[q0:q0 + 2048].double(), Ig[q0:q0 + 2048]
        worst_d = max(worst_d, float((got_d - Dr).abs().max()))
        true_d = d.gather(1, got_i)                                    # the exact distance of what was returned
        worst_set = max(worst_set, float((true_d - Dr[:, -1:]).max()))
        srt = got_i.sort(dim=1).values
        assert bool((srt[:, 1:] != srt[:, :-1]).all())                # no neighbour twice
        exact_ids += int((got_i == Ir).all(dim=1).sum())
        del d
    assert worst_d <= TOL, worst_d
    assert worst_set <= 2 * TOL, worst_set
    assert exact_ids >= 0.98 * n, exact_ids / n                       # (swaps only among near-ties)
