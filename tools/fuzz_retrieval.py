#!/usr/bin/env python
"""Run ON THE GPU BOX: randomised shapes and data for the resident-data-base retrieval path (seeding kernels, bf16-split
filter, fused ranking) against the CPU oracle, bit for bit.  Test infrastructure (imports oracle/): not part of the
product.   python tools/fuzz_retrieval.py [cases] [seed] [sharded]

Every case draws a pool size (16 384 .. 300 000, deliberately off the 16-row tile grid), a query count (1 .. 700), k and
one of several data recipes (isotropic, one tight cluster, exact duplicates of the best match, rows of wildly different
lengths, a low-rank pool, queries that are pool rows); it compares asr_topk_rank_db_dev's top-k indices and float64
distances, ranks, d* and tie counts with oracle.retrieval (SciPy's summation order) and prints one line per case."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audio_sheet_retrieval_amd import _lib  # noqa: E402
from oracle import retrieval as oret  # noqa: E402


def make_case(rng):
    n_db = int(rng.integers(16384, 300000))
    if rng.random() < 0.3:
        n_db = int(rng.choice([16384, 32768, 65536, 131072])) + int(rng.integers(-3, 4))
    n_db = max(n_db, 16384)
    n_q = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 100, 255, 256, 257, 300, 511, 700]))
    k = int(rng.choice([1, 5, 25, 25, 25, 32, 33, 100, 128]))
    recipe = int(rng.integers(0, 6))
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    if recipe == 1:                                           # a tight cluster: many near-ties around the k-th distance
        c = rng.standard_normal(32).astype(np.float32)
        m = int(rng.integers(50, 3000))
        db[rng.choice(n_db, m, replace=False)] = c + np.float32(10.0 ** rng.uniform(-6, -2)) * rng.standard_normal((m, 32)).astype(np.float32)
    if recipe == 3:                                           # rows of very different lengths
        db *= np.exp(rng.uniform(-8, 8, (n_db, 1))).astype(np.float32)
    if recipe == 4:                                           # low-rank pool: the cosines crowd together
        basis = rng.standard_normal((4, 32)).astype(np.float32)
        db = (rng.standard_normal((n_db, 4)).astype(np.float32) @ basis + np.float32(1e-3) * db).astype(np.float32)
    kk, hh = oret.k_h(n_q, n_db)
    match = (np.arange(n_q) // hh) * kk
    q = (db[match] + np.float32(10.0 ** rng.uniform(-3, 0)) * rng.standard_normal((n_q, 32)).astype(np.float32) *
         np.linalg.norm(db[match], axis=1, keepdims=True).astype(np.float32) / np.float32(5.7)).astype(np.float32)
    if recipe == 2:                                           # exact duplicates of a match, at both ends of the pool
        db[n_db - 1] = db[match[0]]
        db[n_db - 2] = db[match[0]]
        db[1 if match[0] != 1 else 2] = db[match[0]]
    if recipe == 5:                                           # queries ARE pool rows (distance 0 / -1e-16)
        q = db[rng.integers(0, n_db, n_q)].copy()
    return n_db, n_q, k, recipe, np.ascontiguousarray(db), np.ascontiguousarray(q)


def sharded_case(eng, rng, ci):
    """the exchange of `bench.py --workload pool2m --exchange queries` on one context: `world` shards (one asr_db handle
    each), d* / j* from the shard that owns a query, top-k + rank counters of ALL queries per shard, counters summed,
    k-lists merged, counters -> ranks; against the oracle on the whole pool"""
    world = int(rng.choice([2, 3, 5, 8]))
    q_local = int(rng.choice([1, 7, 16, 40, 96]))
    target = int(rng.integers(16384, max(16385, 1200000 // world)))     # rows per shard (the entry point asks for >= 16384)
    kk = (target + q_local - 1) // q_local
    k = int(rng.choice([1, 25, 25, 100]))
    n1, shard = world * q_local, q_local * kk
    n2 = world * shard
    db = rng.standard_normal((n2, 32)).astype(np.float32)
    if rng.random() < 0.5:
        db *= np.exp(rng.uniform(-3, 3, (n2, 1))).astype(np.float32)
    match = np.arange(n1) * kk + rng.integers(0, kk, n1)
    q = (db[match] + np.float32(10.0 ** rng.uniform(-3, 0)) * rng.standard_normal((n1, 32)).astype(np.float32) *
         np.linalg.norm(db[match], axis=1, keepdims=True).astype(np.float32) / np.float32(5.7)).astype(np.float32)
    db[n2 - 1] = db[match[0]]                                   # a tie with query 0's match, in the last shard
    t0 = time.perf_counter()
    dq = eng.alloc(q.nbytes).upload(q)
    d_ds, d_js = eng.alloc(n1 * 8), eng.alloc(n1 * 8)
    bufs = [eng.alloc(shard * 128).upload(np.ascontiguousarray(db[r * shard:(r + 1) * shard])) for r in range(world)]
    pools = [eng.db_create(b.ptr, shard) for b in bufs]
    for r, p in enumerate(pools):
        p.rank_dstar_dev(dq.offset(r * q_local * 128), q_local, r * shard, n2, r * q_local, n1, d_ds.offset(r * q_local * 8),
                         d_js.offset(r * q_local * 8))
    d_pidx, d_pdist, d_cnt = eng.alloc(world * n1 * k * 4), eng.alloc(world * n1 * k * 8), eng.alloc(n1 * 12)
    total = np.zeros((n1, 3), np.int64)
    for r, p in enumerate(pools):
        p.topk_count_dev(dq.ptr, n1, k, r * shard, d_pidx.offset(r * n1 * k * 4), d_pdist.offset(r * n1 * k * 8), d_ds.ptr, d_js.ptr,
                         d_cnt.ptr)
        eng.sync()
        total += d_cnt.download((n1, 3), np.int32)
    d_cnt.upload(total.astype(np.int32))
    di, dd = eng.alloc(n1 * k * 4), eng.alloc(n1 * k * 8)
    dr, dso, dt = eng.alloc(n1 * 4), eng.alloc(n1 * 8), eng.alloc(n1 * 4)
    eng.topk_merge_dev(d_pidx.ptr, d_pdist.ptr, world, n1, 0, n1, k, di.ptr, dd.ptr)
    eng.rank_finish_dev(d_cnt.ptr, d_ds.ptr, n1, dr.ptr, dso.ptr, dt.ptr)
    eng.sync()
    got = (di.download((n1, k), np.int32), dd.download((n1, k), np.float64), dr.download((n1,), np.int32),
           dso.download((n1,), np.float64), dt.download((n1,), np.int32))
    for p in pools:
        p.close()
    for b in bufs + [dq, d_ds, d_js, d_pidx, d_pdist, d_cnt, di, dd, dr, dso, dt]:
        b.free()
    t1 = time.perf_counter()
    kq = min(k, n2)
    o_idx, o_dist = oret.topk_blocked(db, q, kq)
    o_ranks, o_dstar, o_ties = oret.ranks_by_counting_blocked(q, db)
    ok = (np.array_equal(got[0][:, :kq], o_idx) and np.array_equal(got[1][:, :kq], o_dist) and np.array_equal(got[2], o_ranks) and
          np.array_equal(got[3], o_dstar) and np.array_equal(got[4], o_ties))
    print("case %2d: %d shards of %6d rows, %3d queries per shard, k %3d  device %.2f s oracle %.1f s  %s" %
          (ci, world, shard, q_local, k, t1 - t0, time.perf_counter() - t1, "ok" if ok else "MISMATCH"), flush=True)
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    eng = _lib.Engine("mutopia_ccal_cont")
    bad = 0
    if len(sys.argv) > 3 and sys.argv[3] == "sharded":
        for ci in range(cases):
            bad += not sharded_case(eng, rng, ci)
        eng.close()
        print("%d sharded cases, %d mismatches" % (cases, bad))
        sys.exit(1 if bad else 0)
    for ci in range(cases):
        n_db, n_q, k, recipe, db, q = make_case(rng)
        t0 = time.perf_counter()
        ddb, dq = eng.alloc(db.nbytes).upload(db), eng.alloc(q.nbytes).upload(q)
        di, dd = eng.alloc(n_q * k * 4), eng.alloc(n_q * k * 8)
        dr, ds, dt = eng.alloc(n_q * 4), eng.alloc(n_q * 8), eng.alloc(n_q * 4)
        pool = eng.db_create(ddb.ptr, n_db)
        pool.topk_rank_dev(dq.ptr, n_q, k, di.ptr, dd.ptr, dr.ptr, ds.ptr, dt.ptr)
        eng.sync()
        idx, dist = di.download((n_q, k), np.int32), dd.download((n_q, k), np.float64)
        ranks, dstar, ties = dr.download((n_q,), np.int32), ds.download((n_q,), np.float64), dt.download((n_q,), np.int32)
        idx2, dist2 = pool.topk(q, k)                          # the top-k alone (no ranking riding along)
        pool.close()
        for b in (ddb, dq, di, dd, dr, ds, dt):
            b.free()
        t1 = time.perf_counter()
        o_idx, o_dist = oret.topk_blocked(db, q, k)
        o_ranks, o_dstar, o_ties = oret.ranks_by_counting_blocked(q, db)
        ok = (np.array_equal(idx, o_idx) and np.array_equal(dist, o_dist) and np.array_equal(idx2, o_idx) and
              np.array_equal(dist2, o_dist) and np.array_equal(ranks, o_ranks) and np.array_equal(dstar, o_dstar) and
              np.array_equal(ties, o_ties))
        bad += not ok
        print("case %2d: pool %6d queries %3d k %3d recipe %d  device %.2f s oracle %.1f s  %s" %
              (ci, n_db, n_q, k, recipe, t1 - t0, time.perf_counter() - t1, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            for name, a, b in (("idx", idx, o_idx), ("dist", dist, o_dist), ("idx(topk)", idx2, o_idx), ("ranks", ranks, o_ranks),
                               ("dstar", dstar, o_dstar), ("ties", ties, o_ties)):
                if not np.array_equal(a, b):
                    w = np.argwhere(a != b)
                    print("   %s differs at %d places, first %s: %r vs %r" % (name, len(w), w[0], a[tuple(w[0])], b[tuple(w[0])]))
    eng.close()
    print("%d cases, %d mismatches" % (cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
