"""CPU: pin the oracle's retrieval restatement against SciPy and the literal
reference procedure (cdist + argsort per row)."""
import numpy as np
import pytest
from scipy.spatial.distance import cdist

from oracle import retrieval as oret


def _unit(rng, n, d=32):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("dim", list(range(1, 34)) + [48, 63, 64])
def test_cdist_bit_exact_vs_scipy(dim):
    rng = np.random.default_rng(dim)
    a = rng.standard_normal((37, dim)).astype(np.float32)
    b = rng.standard_normal((53, dim)).astype(np.float32)
    assert np.array_equal(oret.cdist_cosine64(a, b), cdist(a, b, metric="cosine"))


def _literal_eval_retrieval(lv1, lv2):
    """utils/train_dcca_pool.py:28-82 transcribed with scipy's cdist."""
    n1, n2 = lv1.shape[0], lv2.shape[0]
    k = n2 // n1 if n2 > n1 else 1
    h = n1 // n2 if n1 > n2 else 1
    dists = cdist(lv1, lv2, metric="cosine")
    ranks, hit = [], {1: 0, 5: 0, 10: 0, 25: 0}
    for i in range(n1):
        i_fixed = i // h
        sorted_idx = np.argsort(dists[i], kind="stable")
        for key in hit:
            if i_fixed in (sorted_idx[:key] // k):
                hit[key] += 1
        ranks.append(int(np.min(np.nonzero((sorted_idx // k) == i_fixed)[0]) + 1))
    return np.array(ranks), hit, float(np.diag(dists).mean())


@pytest.mark.parametrize("n1,n2", [(200, 200), (50, 150), (150, 50), (1, 1), (7, 1000)])
def test_ranks_by_counting_equals_argsort(n1, n2):
    rng = np.random.default_rng(n1 + n2)
    a, b = _unit(rng, n1), _unit(rng, n2)
    ranks_lit, hit_lit, mdist = _literal_eval_retrieval(a, b)
    ranks, dstar, ties = oret.ranks_by_counting(oret.cdist_cosine64(a, b))
    assert np.array_equal(ranks, ranks_lit) and ties.sum() == 0
    stats = oret.stats_from_ranks(ranks, dstar)
    ref = oret.eval_retrieval(a, b)
    assert stats[3] == hit_lit == ref[3]
    assert stats[0] == ref[0] and stats[1] == ref[1] and stats[4] == ref[4]
    if n1 == n2:
        assert stats[2] == ref[2] == mdist


def test_ties_and_offsets():
    rng = np.random.default_rng(0)
    a = _unit(rng, 40)
    b = a.copy()
    b[7] = b[2]
    a[7] = a[2]
    d = oret.cdist_cosine64(a, b)
    ranks, dstar, ties = oret.ranks_by_counting(d)
    assert ranks[2] == 1 and ranks[7] == 2 and ties[2] == 1 and ties[7] == 1
    lit, _, _ = _literal_eval_retrieval(a, b)
    assert np.array_equal(ranks, lit)
    # sharded query lists
    parts = [oret.ranks_by_counting(d[s:s + 10], k=1, h=1, query_offset=s)[0] for s in range(0, 40, 10)]
    assert np.array_equal(np.concatenate(parts), ranks)


def test_topk_matches_argsort():
    rng = np.random.default_rng(5)
    db, q = _unit(rng, 500), _unit(rng, 9)
    idx, dist = oret.topk(db, q, 25)
    full = cdist(q, db, metric="cosine")
    assert np.array_equal(idx, np.argsort(full, axis=1, kind="stable")[:, :25])
    assert np.array_equal(dist, np.take_along_axis(full, idx.astype(np.int64), axis=1))


@pytest.mark.parametrize("dim", [1, 2, 7, 31, 32, 33, 64])
def test_c_cdist_equals_numpy_form_bit_for_bit(dim):
    """oracle/cdist_ref.c (used where the NumPy form would take minutes: 2 M candidates) is the same restatement."""
    rng = np.random.default_rng(dim)
    A = rng.standard_normal((37, dim)).astype(np.float32)
    B = rng.standard_normal((501, dim)).astype(np.float32)
    B[3] = A[5]
    B[7] = -A[5]
    B[9] = 3.0 * A[5]
    assert np.array_equal(oret.cdist_cosine64(A, B), oret.cdist_cosine64_c(A, B))
    from scipy.spatial.distance import cdist
    assert np.array_equal(oret.cdist_cosine64_c(A, B), cdist(A.astype(np.float64), B.astype(np.float64), "cosine"))


def test_blocked_rank_and_topk_equal_the_plain_forms():
    rng = np.random.default_rng(4)
    A = rng.standard_normal((150, 32)).astype(np.float32)
    B = rng.standard_normal((3000, 32)).astype(np.float32)
    B[17] = B[5]
    B[2999] = B[5]
    for a, b in zip(oret.ranks_by_counting_blocked(A, B[:150], block=64), oret.ranks_by_counting(oret.cdist_cosine64(A, B[:150]))):
        assert np.array_equal(a, b)
    for a, b in zip(oret.ranks_by_counting_blocked(A, B, block=37), oret.ranks_by_counting(oret.cdist_cosine64(A, B))):
        assert np.array_equal(a, b)
    q = np.concatenate([A[:9], B[5:6]])
    for k in (1, 25, 128):
        i1, d1 = oret.topk_blocked(B, q, k, block=4)
        i2, d2 = oret.topk(B, q, k)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
