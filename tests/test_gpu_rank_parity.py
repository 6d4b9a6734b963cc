"""GPU parity: rank-by-counting kernel vs oracle (float64, bit-exact integers)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d=32):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


@pytest.fixture(scope="module")
def eng():
    from audio_sheet_retrieval_amd import _lib
    e = _lib.Engine("mutopia_ccal_cont")
    yield e
    e.close()


@pytest.mark.parametrize("n1,n2,dim", [(1000, 1000, 32), (257, 257, 32), (64, 64, 7), (100, 300, 32),
                                       (300, 100, 32), (1, 1, 32), (5, 1000, 31)])
def test_ranks_bit_exact(eng, n1, n2, dim):
    from oracle import retrieval as oret
    rng = np.random.default_rng(n1 * 7 + n2)
    a, b = _unit(rng, n1, dim), _unit(rng, n2, dim)
    if n1 == n2:
        b = (b + 1.5 * a).astype(np.float32)          # make the true pair closer than chance
    d = oret.cdist_cosine64(a, b)
    ranks_ref, dstar_ref, ties_ref = oret.ranks_by_counting(d)
    ranks, dstar, ties = eng.rank(a, b)
    assert np.array_equal(ranks, ranks_ref)
    assert np.array_equal(dstar, dstar_ref)             # float64 bit-exact
    assert np.array_equal(ties, ties_ref)
    # and the literal reference procedure (argsort per row) agrees
    stats_ref = oret.eval_retrieval(a, b)
    stats = oret.stats_from_ranks(ranks, dstar)
    assert stats[0] == stats_ref[0] and stats[1] == stats_ref[1]
    assert stats[3] == stats_ref[3] and stats[4] == stats_ref[4]


def test_ties_follow_stable_order(eng):
    """exact duplicates (e.g. zero-padded snippets): rank = stable-sort position."""
    from oracle import retrieval as oret
    rng = np.random.default_rng(3)
    a = _unit(rng, 50)
    b = a.copy()
    b[10] = b[3]
    b[20] = b[3]
    a[10] = a[3]
    a[20] = a[3]
    d = oret.cdist_cosine64(a, b)
    ranks_ref, dstar_ref, ties_ref = oret.ranks_by_counting(d)
    ranks, dstar, ties = eng.rank(a, b)
    assert np.array_equal(ranks, ranks_ref) and np.array_equal(ties, ties_ref)
    assert ties[3] == 2 and ranks[3] == 1 and ranks[10] == 2 and ranks[20] == 3


def test_sharded_queries_equal_unsharded(eng):
    """multi-GPU partitioning: ranking a query shard against all
    candidates with query_offset gives the same integers as the full problem."""
    rng = np.random.default_rng(11)
    a, b = _unit(rng, 400), _unit(rng, 400)
    full = eng.rank(a, b)
    parts = [eng.rank(a[s:s + 100], b, query_offset=s, n1_global=400) for s in range(0, 400, 100)]
    assert np.array_equal(np.concatenate([p[0] for p in parts]), full[0])
    assert np.array_equal(np.concatenate([p[1] for p in parts]), full[1])


def test_rank_rejects_bad_input(eng):
    from audio_sheet_retrieval_amd import _lib
    with pytest.raises(_lib.AsrError):
        eng.rank(np.zeros((3, 32), np.float32), np.zeros((0, 32), np.float32))
    r, d, t = eng.rank(np.zeros((0, 32), np.float32), np.ones((4, 32), np.float32))
    assert r.shape == (0,)
