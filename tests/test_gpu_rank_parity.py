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


@pytest.mark.parametrize("n_db,n_q,k,dim", [(5000, 33, 25, 32), (700, 5, 128, 32), (300, 4, 1, 16), (10, 3, 25, 32),
                                            (2049, 2, 25, 32)])
def test_topk_bit_exact(eng, n_db, n_q, k, dim):
    """audio_sheet_server.py:530-563: cdist + argsort[:k] == exact HIP top-k."""
    from oracle import retrieval as oret
    rng = np.random.default_rng(n_db + k)
    db, q = _unit(rng, n_db, dim), _unit(rng, n_q, dim)
    db[7] = db[3]                                  # exact duplicate: tie broken by index
    idx, dist = eng.topk(db, q, k)
    kk = min(k, n_db)
    idx_ref, dist_ref = oret.topk(db, q, kk)
    assert np.array_equal(idx[:, :kk], idx_ref)
    assert np.array_equal(dist[:, :kk], dist_ref)
    if k > n_db:
        assert (idx[:, n_db:] == -1).all() and np.isinf(dist[:, n_db:]).all()


def test_topk_sharded_merge_equals_global(eng):
    """config 5 partitioning: per-shard top-k with global indices, merged, equals the global top-k."""
    rng = np.random.default_rng(9)
    db, q = _unit(rng, 4000), _unit(rng, 20)
    full_idx, full_dist = eng.topk(db, q, 25)
    parts = [eng.topk(db[s:s + 1000], q, 25, idx_offset=s) for s in range(0, 4000, 1000)]
    cat_idx = np.concatenate([p[0] for p in parts], axis=1)
    cat_dist = np.concatenate([p[1] for p in parts], axis=1)
    order = np.lexsort((cat_idx, cat_dist), axis=1)[:, :25]
    assert np.array_equal(np.take_along_axis(cat_idx, order, axis=1), full_idx)
    assert np.array_equal(np.take_along_axis(cat_dist, order, axis=1), full_dist)


# ---- top-k with the fp32-MFMA filter stage in front of the exact scan (data bases >= 16384 codes) ----------------
def _topk_case(eng, db, q, k):
    from oracle import retrieval as oret
    idx, dist = eng.topk(db, q, k)
    ridx, rdist = oret.topk(db, q, k)
    assert np.array_equal(idx, ridx), "indices differ (k=%d)" % k
    assert np.array_equal(dist, rdist), "distances differ (k=%d)" % k


@pytest.mark.parametrize("k", [1, 25, 128])
def test_topk_filter_stage_is_exact_on_a_large_pool(k):
    from audio_sheet_retrieval_amd import _lib
    rng = np.random.default_rng(11)
    n_db, n_q = 40000, 45                       # 45: not a multiple of the 16-query groups
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = (db[rng.integers(0, n_db, n_q)] + 0.4 * rng.standard_normal((n_q, 32))).astype(np.float32)
    db[1234] = db[77]                           # exact duplicates: ties broken by index
    db[39999] = db[77]
    q[3] = db[77]
    eng = _lib.Engine("mutopia_ccal_cont")
    _topk_case(eng, db, q, k)
    eng.close()


def test_topk_filter_stage_adversarial_orders_and_tie_masses():
    from audio_sheet_retrieval_amd import _lib
    rng = np.random.default_rng(12)
    n_db = 20000
    eng = _lib.Engine("mutopia_ccal_cont")
    base = rng.standard_normal(32).astype(np.float32)
    # (a) every later item is closer to the query than all earlier ones: the threshold keeps moving
    noise = rng.standard_normal((n_db, 32)).astype(np.float32)
    w = np.linspace(3.0, 0.01, n_db, dtype=np.float32)[:, None]
    db = (base[None, :] + w * noise).astype(np.float32)
    q = np.stack([base, base + 0.1 * rng.standard_normal(32).astype(np.float32)]).astype(np.float32)
    _topk_case(eng, db, q, 25)
    _topk_case(eng, db, q, 128)
    # (b) 600 exact copies of the best match (more survivors than a candidate list holds -> exact scan for that
    # query), next to a query without ties
    db2 = rng.standard_normal((n_db, 32)).astype(np.float32)
    db2[rng.choice(n_db, 600, replace=False)] = base
    q2 = np.stack([base, rng.standard_normal(32).astype(np.float32)]).astype(np.float32)
    _topk_case(eng, db2, q2, 25)
    # (c) near-ties within the filter margin: distances that differ in the 7th digit only
    db3 = (base[None, :] * (1.0 + 1e-7 * rng.standard_normal((n_db, 1))) +
           1e-4 * rng.standard_normal((n_db, 32))).astype(np.float32)
    _topk_case(eng, db3, q2[:1], 25)
    eng.close()


# ---- ranking with the MFMA counting path (candidate sets >= 2048) -------------------------------------------------
@pytest.mark.parametrize("n1,n2", [(5000, 5000), (1500, 6000), (8192, 4096), (2500, 2500)])
def test_rank_counting_path_is_exact_on_large_sets(n1, n2):
    from audio_sheet_retrieval_amd import _lib
    from oracle import retrieval as oret
    rng = np.random.default_rng(21)
    lv2 = rng.standard_normal((n2, 32)).astype(np.float32)
    lv2 /= np.linalg.norm(lv2, axis=1, keepdims=True)
    k, h = oret.k_h(n1, n2)
    match = (np.arange(n1) // h) * k
    lv1 = (lv2[match] + 0.8 * rng.standard_normal((n1, 32))).astype(np.float32)
    # exact ties and near-ties around d*: duplicates of a matching candidate, and a candidate 1e-7 away from it
    lv2[n2 - 1] = lv2[match[3]]
    lv2[n2 - 2] = lv2[match[3]]
    lv2[n2 - 3] = (lv2[match[7]].astype(np.float64) * (1.0 + 1e-7)).astype(np.float32) + np.float32(1e-8)
    eng = _lib.Engine("mutopia_ccal_cont")
    ranks, dstar, ties = eng.rank(lv1, lv2)
    d = oret.cdist_cosine64(lv1, lv2)
    r_ranks, r_dstar, r_ties = oret.ranks_by_counting(d, k=k, h=h)
    assert np.array_equal(ranks, r_ranks)
    assert np.array_equal(dstar, r_dstar)
    assert np.array_equal(ties, r_ties)
    eng.close()
