"""GPU: the resident code data base (asr_db_*, include/asr_hip.h) - the shape of the reference's live server
(audio_sheet_server.py:496-522: the data base is loaded once; :530-563: every frame's queries are searched against
it) and of eval_retrieval (utils/train_dcca_pool.py:40-74: ONE distance row per query gives the top ranks and the rank
of the correct item).  Every result is compared bit for bit with the CPU oracle (small pools) and with the stateless
entry points asr_topk / asr_rank (all sizes), which are themselves oracle-pinned at these sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _unit(rng, n, d=32):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


class _Pool(object):
    def __init__(self, eng, codes):
        self.eng, self.codes = eng, np.ascontiguousarray(codes, np.float32)
        self.buf = eng.alloc(max(self.codes.nbytes, 4)).upload(self.codes)
        self.db = eng.db_create(self.buf.ptr, self.codes.shape[0], dim=self.codes.shape[1])

    def fused(self, q, k, query_offset=0, n1_global=None):
        eng, n = self.eng, q.shape[0]
        dq = eng.alloc(q.nbytes).upload(q)
        di, dd = eng.alloc(n * k * 4), eng.alloc(n * k * 8)
        dr, ds, dt = eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4)
        self.db.topk_rank_dev(dq.ptr, n, k, di.ptr, dd.ptr, dr.ptr, ds.ptr, dt.ptr, query_offset=query_offset,
                              n1_global=n1_global)
        eng.sync()
        out = (di.download((n, k), np.int32), dd.download((n, k), np.float64), dr.download((n,), np.int32),
               ds.download((n,), np.float64), dt.download((n,), np.int32))
        for b in (dq, di, dd, dr, ds, dt):
            b.free()
        return out

    def rank(self, q, query_offset=0, n1_global=None):
        eng, n = self.eng, q.shape[0]
        dq = eng.alloc(q.nbytes).upload(q)
        dr, ds, dt = eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4)
        self.db.rank_dev(dq.ptr, n, dr.ptr, ds.ptr, dt.ptr, query_offset=query_offset, n1_global=n1_global)
        eng.sync()
        out = dr.download((n,), np.int32), ds.download((n,), np.float64), dt.download((n,), np.int32)
        for b in (dq, dr, ds, dt):
            b.free()
        return out

    def close(self):
        self.db.close()
        self.buf.free()


@pytest.fixture()
def eng():
    from audio_sheet_retrieval_amd import _lib
    e = _lib.Engine("mutopia_ccal_cont")
    yield e
    e.close()


@pytest.mark.parametrize("n_db,n_q,k,dim", [(5000, 33, 25, 32), (700, 5, 128, 32), (300, 4, 1, 16), (10, 3, 25, 32),
                                            (20000, 45, 25, 32), (40000, 64, 128, 32), (40000, 7, 1, 32)])
def test_db_topk_equals_the_oracle(eng, n_db, n_q, k, dim):
    """small pools (exact scan), pools with the MFMA filter on unit-length rows, 7 / 45 / 64 queries (one workgroup
    for all four query groups), k = 1 / 25 / 128"""
    from oracle import retrieval as oret
    rng = np.random.default_rng(n_db + k)
    db, q = _unit(rng, n_db, dim), _unit(rng, n_q, dim)
    q[0] = db[3] + np.float32(0.05) * q[0]
    db[7] = db[3]                                  # exact duplicates: ties broken by index
    db[n_db - 1] = db[3]
    pool = _Pool(eng, db)
    idx, dist = pool.db.topk(q, k)
    kk = min(k, n_db)
    ridx, rdist = oret.topk(db, q, kk)
    assert np.array_equal(idx[:, :kk], ridx) and np.array_equal(dist[:, :kk], rdist)
    if k > n_db:
        assert (idx[:, n_db:] == -1).all() and np.isinf(dist[:, n_db:]).all()
    sidx, sdist = pool.db.topk(q, kk, idx_offset=1000)                       # a shard with global indices
    assert np.array_equal(sidx, ridx + 1000) and np.array_equal(sdist, rdist)
    pool.close()


@pytest.mark.parametrize("n_db,n_q", [(20011, 45), (40003, 64), (16397, 3), (40003, 301)])
def test_db_pool_sizes_off_the_tile_grid_and_zero_rows(eng, n_db, n_q):
    """pools whose size is no multiple of 16 or 4 (the filter pads the last MFMA tile with NaN rows, whose cosines fail
    every comparison) with the best matches in the very last rows: top-k and the fused ranks equal the oracle; then with
    a few ZERO rows (cosine NaN: never a candidate - scipy's nan sorts last) the top-k still equals the oracle's;
    301 queries: the seeding pass's four-queries-per-workgroup build with a remainder"""
    from oracle import retrieval as oret
    rng = np.random.default_rng(n_db)
    db, q = _unit(rng, n_db), _unit(rng, n_q)
    q[0] = db[n_db - 1] + np.float32(0.05) * q[0]
    q[1] = db[n_db - 2] + np.float32(0.05) * q[1]
    q[2] = db[n_db - 14] + np.float32(0.05) * q[2]
    pool = _Pool(eng, db)
    idx, dist, ranks, dstar, ties = pool.fused(q, 25)
    ridx, rdist = oret.topk(db, q, 25)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    assert idx[0, 0] == n_db - 1 and idx[1, 0] == n_db - 2 and idx[2, 0] == n_db - 14
    kk, hh = oret.k_h(n_q, n_db)
    o_ranks, o_dstar, o_ties = oret.ranks_by_counting(oret.cdist_cosine64(q, db), k=kk, h=hh)
    assert np.array_equal(ranks, o_ranks) and np.array_equal(dstar, o_dstar) and np.array_equal(ties, o_ties)
    pool.close()
    db[[5, 777, n_db - 3, n_db - 1]] = 0.0
    pool = _Pool(eng, db)
    idx, dist = pool.db.topk(q, 25)
    with np.errstate(invalid="ignore", divide="ignore"):
        ridx, rdist = oret.topk(db, q, 25)
    assert np.isfinite(rdist).all()
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    pool.close()


def test_db_filter_adversarial_orders_and_tie_masses(eng):
    """the cases the stateless filter is tested with, through the unit-length copy: a threshold that keeps moving, more
    exact ties than a candidate list holds (falls back to the exact scan), distances that differ in the 7th digit, rows
    of very different lengths (the cosine does not see them, the unit copy must not either)"""
    from oracle import retrieval as oret
    rng = np.random.default_rng(12)
    n_db = 20000
    base = rng.standard_normal(32).astype(np.float32)
    noise = rng.standard_normal((n_db, 32)).astype(np.float32)
    w = np.linspace(3.0, 0.01, n_db, dtype=np.float32)[:, None]
    db1 = (base[None, :] + w * noise).astype(np.float32)
    q1 = np.stack([base, base + 0.1 * rng.standard_normal(32).astype(np.float32)]).astype(np.float32)
    db2 = rng.standard_normal((n_db, 32)).astype(np.float32)
    db2[rng.choice(n_db, 600, replace=False)] = base
    q2 = np.stack([base, rng.standard_normal(32).astype(np.float32)]).astype(np.float32)
    db3 = (base[None, :] * (1.0 + 1e-7 * rng.standard_normal((n_db, 1))) +
           1e-4 * rng.standard_normal((n_db, 32))).astype(np.float32)
    db4 = (_unit(rng, n_db) * np.exp(rng.uniform(-8, 8, (n_db, 1)))).astype(np.float32)
    q4 = (db4[rng.integers(0, n_db, 30)] * np.float32(3.0) + 0.2 * rng.standard_normal((30, 32))).astype(np.float32)
    for db, q, ks in ((db1, q1, (25, 128)), (db2, q2, (25,)), (db3, q2[:1], (25,)), (db4, q4, (25,))):
        pool = _Pool(eng, db)
        for k in ks:
            idx, dist = pool.db.topk(q, k)
            ridx, rdist = oret.topk(db, q, k)
            assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), k
        pool.close()


@pytest.mark.parametrize("n1,n2,k", [(300, 300, 25), (5000, 20000, 25), (2048, 65536, 25), (40, 32768, 128),
                                     (64, 16384, 1), (2500, 40000, 100)])
def test_fused_topk_and_rank_equal_the_two_separate_calls_and_the_oracle(eng, n1, n2, k):
    """asr_topk_rank_db_dev: one walk over the pool for both results - identical to asr_topk + asr_rank (bit for bit),
    which equal the oracle; exact ties and a near-tie around d*, a query shard with its global offset"""
    from oracle import retrieval as oret
    rng = np.random.default_rng(n1 + n2)
    lv2 = _unit(rng, n2)
    kk, hh = oret.k_h(n1, n2)
    match = (np.arange(n1) // hh) * kk
    lv1 = (lv2[match] + 0.12 * rng.standard_normal((n1, 32))).astype(np.float32)
    lv2[n2 - 1] = lv2[match[3]]
    lv2[n2 - 2] = lv2[match[3]]
    lv2[n2 - 3] = (lv2[match[7]].astype(np.float64) * (1.0 + 1e-7)).astype(np.float32) + np.float32(1e-8)
    pool = _Pool(eng, lv2)
    idx, dist, ranks, dstar, ties = pool.fused(lv1, k)
    e_idx, e_dist = eng.topk(lv2, lv1, k)
    e_ranks, e_dstar, e_ties = eng.rank(lv1, lv2)
    assert np.array_equal(idx, e_idx) and np.array_equal(dist, e_dist)
    assert np.array_equal(ranks, e_ranks) and np.array_equal(dstar, e_dstar) and np.array_equal(ties, e_ties)
    r_ranks, r_dstar, r_ties = pool.rank(lv1)                                 # the ranking alone, norms from the handle
    assert np.array_equal(r_ranks, e_ranks) and np.array_equal(r_dstar, e_dstar) and np.array_equal(r_ties, e_ties)
    if n1 * n2 <= 5000 * 20000:
        d = oret.cdist_cosine64(lv1, lv2)
        o_ranks, o_dstar, o_ties = oret.ranks_by_counting(d, k=kk, h=hh)
        assert np.array_equal(ranks, o_ranks) and np.array_equal(dstar, o_dstar) and np.array_equal(ties, o_ties)
        o_idx, o_dist = oret.topk(lv2, lv1, k)
        assert np.array_equal(idx, o_idx) and np.array_equal(dist, o_dist)
    lo = n1 // 3
    s = pool.fused(lv1[lo:], k, query_offset=lo, n1_global=n1)
    assert np.array_equal(s[0], idx[lo:]) and np.array_equal(s[1], dist[lo:])
    assert np.array_equal(s[2], ranks[lo:]) and np.array_equal(s[3], dstar[lo:]) and np.array_equal(s[4], ties[lo:])
    assert ties[3] >= 2                          # the two copies of query 3's match tie with it
    pool.close()


def test_db_refresh_follows_rows_changed_in_place_and_handles_are_checked(eng):
    from audio_sheet_retrieval_amd import _lib
    from oracle import retrieval as oret
    rng = np.random.default_rng(5)
    a, b, q = _unit(rng, 30000), _unit(rng, 30000), _unit(rng, 20)
    pool = _Pool(eng, a)
    i0, d0 = pool.db.topk(q, 10)
    pool.buf.upload(b)                      # e.g. an all-gather landing new shards in the same buffer
    pool.db.refresh()
    i1, d1 = pool.db.topk(q, 10)
    r0, r1 = oret.topk(a, q, 10), oret.topk(b, q, 10)
    assert np.array_equal(i0, r0[0]) and np.array_equal(d0, r0[1])
    assert np.array_equal(i1, r1[0]) and np.array_equal(d1, r1[1])
    other = _lib.Engine("mutopia_ccal_cont")
    with pytest.raises(_lib.AsrError):      # a handle belongs to the context that made it
        other._check(other.lib.asr_topk_db_dev(other.ctx, pool.db.handle, pool.buf.ptr, 1, 32, 1, 0, pool.buf.ptr,
                                               pool.buf.ptr))
    other.close()
    with pytest.raises(_lib.AsrError):
        pool.rank(_unit(rng, 40000))        # more queries than candidates and no 1:1 layout: query without a match
    empty = eng.db_create(pool.buf.ptr, 0)
    empty.close()
    pool.close()


def test_embedding_db_keeps_its_pool_resident(eng):
    """piece_identification.EmbeddingDB: the codes, their norms and the query scratch live on the device for the life of
    the object; detect_* retrieve against the handle"""
    from audio_sheet_retrieval_amd.piece_identification import EmbeddingDB
    from oracle import retrieval as oret
    rng = np.random.default_rng(2)
    codes = _unit(rng, 50000)
    ids = (np.arange(50000) // 500).astype(np.int32)
    db = EmbeddingDB(eng, codes, ids, dict((i, "piece%d" % i) for i in range(100)))
    q = (codes[rng.integers(0, 50000, 32)] + 0.05 * rng.standard_normal((32, 32))).astype(np.float32)
    idx, dist = db.retrieve(q, 25)
    ridx, rdist = oret.topk(codes, q, 25)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    idx2, _ = db.retrieve(q[:5], 10)
    assert np.array_equal(idx2, ridx[:5, :10])


def test_query_sharded_retrieval_over_three_shards_equals_the_whole_pool(eng):
    """The pieces of `bench.py --workload pool2m --exchange queries` on one context: the pool cut into three shards (three
    asr_db handles), d* / j* from the shard that holds a query's correct candidates, top-k + rank counters of ALL queries
    per shard (asr_topk_count_db_dev), counters summed, k-lists merged (asr_topk_merge_dev), asr_rank_finish_dev - every
    array equal to the fused call on the whole pool (which equals asr_topk + asr_rank and the oracle, tests above)."""
    from oracle import retrieval as oret
    n1, shard, world, k = 768, 32768, 3, 25
    n2 = shard * world
    rng = np.random.default_rng(31)
    lv2 = _unit(rng, n2)
    kk, hh = oret.k_h(n1, n2)
    match = (np.arange(n1) // hh) * kk
    lv1 = (lv2[match] + 0.2 * rng.standard_normal((n1, 32))).astype(np.float32)
    lv2[n2 - 1] = lv2[match[3]]                  # a tie with a correct candidate of shard 0, sitting in shard 2
    lv2[shard + 5] = lv2[match[700]]             # ... and one of shard 2's in shard 1, BEFORE j* in index order
    whole = _Pool(eng, lv2)
    w_idx, w_dist, w_ranks, w_dstar, w_ties = whole.fused(lv1, k)
    whole.close()
    dq = eng.alloc(lv1.nbytes).upload(lv1)
    q_local = n1 // world
    d_ds, d_js = eng.alloc(n1 * 8), eng.alloc(n1 * 8)
    pools = [_Pool(eng, lv2[r * shard:(r + 1) * shard]) for r in range(world)]
    for r, p in enumerate(pools):                # every "rank" owns the d* of its own queries
        p.db.rank_dstar_dev(dq.offset(r * q_local * 128), q_local, r * shard, n2, r * q_local, n1,
                            d_ds.offset(r * q_local * 8), d_js.offset(r * q_local * 8))
    d_pidx, d_pdist = eng.alloc(world * n1 * k * 4), eng.alloc(world * n1 * k * 8)
    d_cnt = eng.alloc(n1 * 12)
    total = np.zeros((n1, 3), np.int64)
    for r, p in enumerate(pools):
        p.db.topk_count_dev(dq.ptr, n1, k, r * shard, d_pidx.offset(r * n1 * k * 4), d_pdist.offset(r * n1 * k * 8), d_ds.ptr,
                            d_js.ptr, d_cnt.ptr)
        eng.sync()
        total += d_cnt.download((n1, 3), np.int32)
    d_cnt.upload(total.astype(np.int32))
    di, dd = eng.alloc(n1 * k * 4), eng.alloc(n1 * k * 8)
    dr, dso, dt = eng.alloc(n1 * 4), eng.alloc(n1 * 8), eng.alloc(n1 * 4)
    eng.topk_merge_dev(d_pidx.ptr, d_pdist.ptr, world, n1, 0, n1, k, di.ptr, dd.ptr)
    eng.rank_finish_dev(d_cnt.ptr, d_ds.ptr, n1, dr.ptr, dso.ptr, dt.ptr)
    eng.sync()
    assert np.array_equal(di.download((n1, k), np.int32), w_idx)
    assert np.array_equal(dd.download((n1, k), np.float64), w_dist)
    assert np.array_equal(dr.download((n1,), np.int32), w_ranks)
    assert np.array_equal(dso.download((n1,), np.float64), w_dstar)
    assert np.array_equal(dt.download((n1,), np.int32), w_ties)
    assert w_ties[3] >= 1 and w_ties[700] >= 1
    # a shard asked for the d* of queries whose candidates it does not hold refuses
    from audio_sheet_retrieval_amd import _lib
    with pytest.raises(_lib.AsrError):
        pools[0].db.rank_dstar_dev(dq.offset(q_local * 128), q_local, 0, n2, q_local, n1, d_ds.ptr, d_js.ptr)
    for p in pools:
        p.close()


def test_db_argument_checks_and_handle_ownership():
    """index offsets that would leave int32 are refused (topk_db, topk_rank_db, topk_count_db), rank_db
    checks its output pointers like topk_db does, and closing the ENGINE first releases the data bases it still owns
    (their device buffers belong to the context): CodeDB.close() / EmbeddingDB.close() afterwards are no-ops."""
    from audio_sheet_retrieval_amd import _lib
    from audio_sheet_retrieval_amd.piece_identification import EmbeddingDB
    rng = np.random.default_rng(0)
    eng = _lib.Engine("mutopia_ccal_cont")
    codes = _unit(rng, 20000)
    q = _unit(rng, 4)
    buf = eng.alloc(codes.nbytes).upload(codes)
    dq = eng.alloc(q.nbytes).upload(q)
    di, dd = eng.alloc(4 * 25 * 4), eng.alloc(4 * 25 * 8)
    dr, ds, dt = eng.alloc(16), eng.alloc(32), eng.alloc(16)
    db = eng.db_create(buf.ptr, codes.shape[0])
    for call in (lambda: db.topk_dev(dq.ptr, 4, 25, di.ptr, dd.ptr, idx_offset=2 ** 31 - 10000),
                 lambda: db.topk_dev(dq.ptr, 4, 25, di.ptr, dd.ptr, idx_offset=-1),
                 lambda: db.topk_rank_dev(dq.ptr, 4, 25, di.ptr, dd.ptr, dr.ptr, ds.ptr, dt.ptr, idx_offset=2 ** 31 - 10000),
                 lambda: db.topk_count_dev(dq.ptr, 4, 25, 2 ** 31 - 10000, di.ptr, dd.ptr, ds.ptr, ds.ptr, dr.ptr),
                 lambda: db.rank_dev(dq.ptr, 4, None, ds.ptr, dt.ptr),
                 lambda: db.rank_dev(dq.ptr, 4, dr.ptr, None, dt.ptr)):
        with pytest.raises(_lib.AsrError) as ei:
            call()
        assert ei.value.code == _lib.ASR_ERR_INVALID
    db.topk_dev(dq.ptr, 4, 25, di.ptr, dd.ptr, idx_offset=2 ** 31 - 1 - 20000)       # the largest offset that fits
    eng.sync()
    assert di.download((4, 25), np.int32).max() <= 2 ** 31 - 1
    edb = EmbeddingDB(eng, codes[:500], np.arange(500) % 7, {i: "p%d" % i for i in range(7)})
    assert len(eng._open_dbs) == 2
    eng.close()                                   # destroys both data bases before the context
    assert not eng._open_dbs and db.handle is None
    db.close()
    edb.close()
    with EmbeddingDB(_lib.Engine("mutopia_ccal_cont"), codes[:500], np.arange(500) % 7, {}) as e2:
        idx, _ = e2.retrieve(q, 5)
        assert idx.shape == (4, 5)
        e2.engine.close()


def test_a_dropped_data_base_handle_releases_its_device_buffers(eng):
    """the engine's registry of open data bases is a WeakSet - a CodeDB the caller drops without close()
    is collected, its __del__ frees the ~136 B per row the handle owns (norms, reciprocal norms, unit rows), and a
    server that builds a data base per request (engine.db_create in a loop) does not pile them up until Engine.close()."""
    import ctypes
    import gc
    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    n = 1 << 20                                              # ~143 MB of handle-owned buffers
    codes = _unit(np.random.default_rng(5), n)
    buf = eng.alloc(codes.nbytes).upload(codes)
    eng.sync()
    base = free_bytes()
    db = eng.db_create(buf.ptr, n)
    eng.sync()
    held = base - free_bytes()
    assert held >= n * 100, held                             # norms + reciprocal norms + unit rows
    assert len(eng._open_dbs) == 1
    del db                                                   # no close()
    gc.collect()
    assert len(eng._open_dbs) == 0
    assert base - free_bytes() <= held // 8, (base, free_bytes(), held)
    for _ in range(6):                                       # the per-request pattern: nothing accumulates
        eng.db_create(buf.ptr, n)
    gc.collect()
    assert len(eng._open_dbs) == 0
    assert base - free_bytes() <= held // 8
    kept = eng.db_create(buf.ptr, n)                         # a live handle is still closed by the engine
    assert len(eng._open_dbs) == 1 and kept.handle is not None
    eng.close()
    assert kept.handle is None and len(eng._open_dbs) == 0


_SCAN_SCRIPT = r'''
import sys
import numpy as np
from audio_sheet_retrieval_amd import _lib
from oracle import retrieval as oret
rng = np.random.default_rng(int(sys.argv[1]))
eng = _lib.Engine("mutopia_ccal_cont")
bad = 0
for case in range(int(sys.argv[2])):
    n_db = int(rng.choice([16384, 16385, 20011, 65536, 100003, 262147, 524288 + 5]))
    n_q = int(rng.choice([1, 2, 3, 7, 16, 33, 100, 128]))
    k = int(rng.choice([1, 5, 25, 32, 33, 100, 128]))
    recipe = case % 6
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    q = rng.standard_normal((n_q, 32)).astype(np.float32)
    if recipe == 1:                     # a tight cluster around the query: masses of near-ties, slices that overflow
        m = int(rng.integers(2000, 6000))
        at = int(rng.integers(0, n_db - m))
        db[at:at + m] = q[0] + np.float32(10.0 ** rng.uniform(-7, -3)) * rng.standard_normal((m, 32)).astype(np.float32)
    if recipe == 2:                     # exact duplicates of the best match, far apart: ties broken by index
        db[rng.choice(n_db, 40, replace=False)] = q[0]
    if recipe == 3:                     # rows of wildly different lengths, some zero rows (NaN cosines sort last)
        db *= np.exp(rng.uniform(-8, 8, (n_db, 1))).astype(np.float32)
        db[rng.choice(n_db, 5, replace=False)] = 0.0
    if recipe == 4:                     # the best matches in the very last rows of the pool
        db[-3:] = q[0] + np.float32(1e-3) * rng.standard_normal((3, 32)).astype(np.float32)
    if recipe == 5:                     # fewer usable rows than k in most slices: a pool of (almost) all zero rows
        db[:] = 0.0
        db[rng.choice(n_db, 60, replace=False)] = rng.standard_normal((60, 32)).astype(np.float32)
    buf = eng.alloc(db.nbytes).upload(db)
    pool = eng.db_create(buf.ptr, n_db)
    idx, dist = pool.topk(q, k, idx_offset=7)
    with np.errstate(invalid="ignore", divide="ignore"):
        ridx, rdist = oret.topk(db, q, k)
    fin = np.isfinite(rdist)
    ok = np.array_equal(idx[fin], ridx[fin] + 7) and np.array_equal(dist[fin], rdist[fin])
    print("case %d: pool %d queries %d k %d recipe %d %s" % (case, n_db, n_q, k, recipe, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
    pool.close(); buf.free()
eng.close()
sys.exit(1 if bad else 0)
'''


@pytest.mark.parametrize("scan_nq,select", [("", ""), ("128", ""), ("0", ""), ("0", "0")])
def test_single_query_scan_path_equals_the_oracle(scan_nq, select, tmp_path):
    """The reference's own shape - ONE query against the whole data base (audio_sheet_server.py:530-563) - takes a
    single streaming pass since round 5 (topk_scan_kernel: per-slice fp32 keys, radix select, exact float64 survivors,
    head-pruned merge).  1-128 queries x pools on and off the tile grid x k = 1 .. 128 x adversarial recipes (tight clusters
    that overflow a slice's survivor buffer, exact duplicates, zero rows, best matches in the last rows, pools with
    fewer usable rows than k), bit for bit against the oracle; ASR_TOPK_SCAN=4 sends up to four queries down that path,
    =0 none (the general path on the same cases: its exact refine is topk_collect_kernel + topk_select_kernel, whose
    in-kernel exact scan the tight clusters trigger; ASR_TOPK_SELECT=0 is round 4's topk_kernel + merge)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    env.pop("ASR_TOPK_SCAN", None)
    env.pop("ASR_TOPK_SELECT", None)
    if scan_nq:
        env["ASR_TOPK_SCAN"] = scan_nq
    if select:
        env["ASR_TOPK_SELECT"] = select
    out = subprocess.run([sys.executable, "-c", _SCAN_SCRIPT, "77", "24"], env=env, cwd=root, capture_output=True, text=True,
                         timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 24
