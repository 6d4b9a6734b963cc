"""Oracle: all-pairs cosine retrieval metrics (NumPy float64, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows
  audio_sheet_retrieval/utils/train_dcca_pool.py:28-82   (eval_retrieval)
  audio_sheet_retrieval/audio_sheet_server.py:530-563    (top-k retrieval)
and restates scipy.spatial.distance.cdist(..., "cosine") (SciPy's C routine
cdist_cosine: per-row norms = sqrt(sum x*x), per pair
cos = sum(u*v) / (|u|*|v|), clipped to [-1,1], d = 1 - cos; all in float64 -
"third-party semantic").  The summation ORDER is the one the installed SciPy
1.15.3 binary uses (2-lane SSE2 reduction): even-k and odd-k terms are summed
left to right in two accumulators, the accumulators are added, a trailing odd
element is added last.  tests/test_oracle_retrieval.py checks this restatement
BIT-FOR-BIT against scipy.spatial.distance.cdist for D = 1..64.

For float32-valued inputs every product u_k*v_k is exact in float64, so the
result does not depend on whether a compiler contracts mul+add into fma; only
the summation order matters, and the HIP rank kernel uses the same order -
which is what makes integer ranks bit-exact.
"""
from __future__ import annotations

import numpy as np


def _sum2(P):
    """Two-accumulator (even k / odd k) left-to-right float64 sum over the last
    axis, accumulators added at the end, odd tail element added last."""
    n = P.shape[-1]
    m = n // 2 * 2
    a0 = np.zeros(P.shape[:-1], np.float64)
    a1 = np.zeros(P.shape[:-1], np.float64)
    for k in range(0, m, 2):
        a0 = a0 + P[..., k]
        a1 = a1 + P[..., k + 1]
    s = a0 + a1
    for k in range(m, n):
        s = s + P[..., k]
    return s


def row_norms64(X):
    """sqrt of the float64 sum of squares of every row (SciPy _row_norms)."""
    X = np.asarray(X, dtype=np.float64)
    return np.sqrt(_sum2(X * X))


def cdist_cosine64(A, B):
    """float64 cosine distance matrix with SciPy's operation order."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    na, nb = row_norms64(A), row_norms64(B)
    dot = np.empty((A.shape[0], B.shape[0]), np.float64)
    step = max(1, (1 << 22) // max(1, B.shape[0] * B.shape[1]))
    for s0 in range(0, A.shape[0], step):       # chunked: products are exact
        dot[s0:s0 + step] = _sum2(A[s0:s0 + step, None, :] * B[None, :, :])
    cos = dot / (na[:, None] * nb[None, :])
    cos = np.where(np.abs(cos) > 1.0, np.copysign(1.0, cos), cos)
    return 1.0 - cos


def cdist_cosine64_c(A, B):
    """cdist_cosine64 evaluated by oracle/cdist_ref.c (same operation order, bit-for-bit equal:
    tests/test_oracle_retrieval.py) - for candidate sets where the NumPy form takes minutes.  float32 inputs."""
    import ctypes
    from .network import _conv_lib
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    assert A.ndim == 2 and B.ndim == 2 and A.shape[1] == B.shape[1]
    lib = _conv_lib()
    fp, dp = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double)
    na, nb = np.empty(A.shape[0], np.float64), np.empty(B.shape[0], np.float64)
    lib.row_norms_f64(A.ctypes.data_as(fp), A.shape[0], A.shape[1], A.shape[1], na.ctypes.data_as(dp))
    lib.row_norms_f64(B.ctypes.data_as(fp), B.shape[0], B.shape[1], B.shape[1], nb.ctypes.data_as(dp))
    out = np.empty((A.shape[0], B.shape[0]), np.float64)
    lib.cdist_cosine_f64(A.ctypes.data_as(fp), A.shape[0], B.ctypes.data_as(fp), B.shape[0], A.shape[1],
                         na.ctypes.data_as(dp), nb.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def ranks_by_counting_blocked(lv1, lv2, block=64, query_offset=0, n1_global=None):
    """ranks_by_counting(cdist_cosine64(lv1, lv2)) without holding the whole distance matrix: query rows in blocks of
    `block` through cdist_cosine64_c (2 M candidates x 64 rows = 1 GB of float64)."""
    n1, n2 = lv1.shape[0], lv2.shape[0]
    k, h = k_h(n1 if n1_global is None else n1_global, n2)
    ranks, dstar, ties = np.zeros(n1, np.int32), np.zeros(n1, np.float64), np.zeros(n1, np.int32)
    for s0 in range(0, n1, block):
        d = cdist_cosine64_c(lv1[s0:s0 + block], lv2)
        r, ds, t = ranks_by_counting(d, k=k, h=h, query_offset=query_offset + s0)
        ranks[s0:s0 + block], dstar[s0:s0 + block], ties[s0:s0 + block] = r, ds, t
    return ranks, dstar, ties


def topk_blocked(db_codes, query_codes, k, block=64):
    """topk() for large data bases: query blocks through cdist_cosine64_c, stable order."""
    nq = query_codes.shape[0]
    idx, dist = np.zeros((nq, k), np.int32), np.zeros((nq, k), np.float64)
    for s0 in range(0, nq, block):
        d = cdist_cosine64_c(query_codes[s0:s0 + block], db_codes)
        # the k smallest by (distance, index): argpartition narrows, a stable sort of the survivors orders them
        for r in range(d.shape[0]):
            row = d[r]
            kth = np.partition(row, k - 1)[k - 1]
            cand = np.nonzero(row <= kth)[0]
            order = cand[np.argsort(row[cand], kind="stable")][:k]
            idx[s0 + r], dist[s0 + r] = order, row[order]
    return idx, dist


def k_h(n1, n2):
    """train_dcca_pool.py:35-36 (py2 integer division)."""
    k = n2 // n1 if n2 > n1 else 1
    h = n1 // n2 if n1 > n2 else 1
    return k, h


def ranks_by_counting(dists, k=None, h=None, query_offset=0):
    """rank_i (1-based) of the correct item in a stable ascending sort of row
    i, restating train_dcca_pool.py:35-36,46-66 without the sort:
        correct candidates of query i: j with j // k == (i + query_offset) // h
        rank_i = 1 + #{j: d_ij < d*} + #{j < j*: d_ij == d*}
    with d* the smallest distance among the correct candidates and j* the first
    index attaining it.  Also returns d* and the number of other candidates
    tied with d*; tie-free rows are the bit-exact contract (A.9).
    `query_offset` is the global index of row 0 (sharded query sets); k, h
    default to k_h(*dists.shape)."""
    n1, n2 = dists.shape
    if k is None or h is None:
        k, h = k_h(n1, n2)
    ranks = np.zeros(n1, np.int32)
    dstar = np.zeros(n1, np.float64)
    ties = np.zeros(n1, np.int32)
    for i in range(n1):
        i_fixed = (i + query_offset) // h
        row = dists[i]
        lo, hi = i_fixed * k, min(i_fixed * k + k, n2)
        jbest = lo + int(np.argmin(row[lo:hi]))      # first minimum
        d = row[jbest]
        less = int(np.count_nonzero(row < d))
        eq_before = int(np.count_nonzero(row[:jbest] == d))
        ranks[i] = 1 + less + eq_before
        dstar[i] = d
        ties[i] = int(np.count_nonzero(row == d)) - 1
    return ranks, dstar, ties


def eval_retrieval(lv1_cca, lv2_cca):
    """train_dcca_pool.py:28-82, same 5-tuple:
    (mean_rank, median_rank, mean(diag(dists)), hit_rates{1,5,10,25}, mean(1/rank)).
    Uses the literal reference procedure (full argsort per row) so that
    ranks_by_counting can be tested against it."""
    n_v1, n_v2 = lv1_cca.shape[0], lv2_cca.shape[0]
    k = n_v2 // n_v1 if n_v2 > n_v1 else 1
    h = n_v1 // n_v2 if n_v1 > n_v2 else 1
    dists = cdist_cosine64(lv1_cca, lv2_cca)
    ranks, aps = [], []
    hit_rates = {1: 0, 5: 0, 10: 0, 25: 0}
    for i in range(n_v1):
        i_fixed = np.floor_divide(i, h)
        sorted_idx = np.argsort(dists[i], kind="stable")
        for key in hit_rates:
            top_k_results = np.floor_divide(sorted_idx[0:key], k)
            if i_fixed in top_k_results:
                hit_rates[key] += 1
        fixed_sorted_idx = np.floor_divide(sorted_idx, k)
        rank = np.min(np.nonzero(fixed_sorted_idx == i_fixed)[0]) + 1
        ranks.append(rank)
        aps.append(1.0 / rank)
    mean_rank = np.mean(ranks)
    median_rank = np.median(ranks)
    mean_dist = np.diag(dists).mean()
    map_ = np.mean(aps)
    return mean_rank, median_rank, mean_dist, hit_rates, map_


def stats_from_ranks(ranks, dstar):
    """The 5-tuple of eval_retrieval from integer ranks (hit@k <=> rank <= k)."""
    ranks = np.asarray(ranks)
    hit_rates = {key: int(np.count_nonzero(ranks <= key)) for key in (1, 5, 10, 25)}
    return (np.mean(ranks), np.median(ranks), float(np.mean(dstar)), hit_rates,
            float(np.mean(1.0 / ranks.astype(np.float64))))


def topk(db_codes, query_codes, k):
    """audio_sheet_server.py:534-537: cdist(DB, q, 'cosine') -> argsort[:k]
    per query, stable order (index ascending among equal distances).
    Returns idx (Q,k) int32 and dist (Q,k) float64."""
    d = cdist_cosine64(query_codes, db_codes)
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int32), np.take_along_axis(d, idx, axis=1)
