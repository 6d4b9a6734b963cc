"""Outputs of the reference's own NumPy/SciPy code (tests/golden/reference_golden.npz, produced by
tests/golden/make_reference_golden.py in the build container) against the oracle (CPU) and the HIP library (GPU).

Pinned here: CCA.fit('svd') (utils/cca.py), eval_retrieval (utils/train_dcca_pool.py:28-82), dtw_by_dist
(utils/dtw_by_dist.py) and the alignment helpers (utils/alignment.py:112-190).
Tolerances: integer results (ranks statistics, hit counts, DTW paths, aligned indices) bit-exact; float64 DTW costs
bit-exact; CCA matrices 1e-4 relative (the reference accumulates the covariances in float32, the oracle and the
library in float64 - 1e-4 is north_star's embedding tolerance).
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden.npz")
CCA_CASES = ("cca_a", "cca_b")
EVAL_CASES = ("eval_a", "eval_b", "eval_c")
DTW_CASES = ("dtw_tall", "dtw_wide", "dtw_square")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _check_cca(g, tag, U, V, m1, m2):
    Ur, Vr = g[tag + "/U"], g[tag + "/V"]
    assert np.abs(m1 - g[tag + "/m1"]).max() <= 1e-4 and np.abs(m2 - g[tag + "/m2"]).max() <= 1e-4
    U, V = np.asarray(U, np.float64), np.asarray(V, np.float64)
    # the singular vectors are defined up to one joint sign per component
    sign = np.sign((U * Ur).sum(axis=0))
    scale = max(1.0, np.abs(Ur).max(), np.abs(Vr).max())
    assert np.abs(U * sign - Ur).max() <= 1e-4 * scale
    assert np.abs(V * sign - Vr).max() <= 1e-4 * scale
    # and what retrieval uses - cross-view scores of the projected training data - agrees without any sign fix
    a = (g[tag + "/H1"] - g[tag + "/m1"]).astype(np.float64)
    b = (g[tag + "/H2"] - g[tag + "/m2"]).astype(np.float64)
    assert np.abs((a @ U) @ (b @ V).T - (a @ Ur) @ (b @ Vr).T).max() <= 1e-3 * np.abs((a @ Ur) @ (b @ Vr).T).max()


def _check_eval(g, tag, result):
    mean_rank, median_rank, mean_dist, hits, mean_ap = result
    ref = g[tag + "/stats"]
    assert mean_rank == ref[0] and median_rank == ref[1]
    assert abs(mean_dist - ref[2]) <= 1e-12
    assert abs(mean_ap - ref[3]) <= 1e-12
    assert [hits[1], hits[5], hits[10], hits[25]] == list(g[tag + "/hits"])


# ---- CPU: the oracle against the reference's outputs --------------------------------------------------------------
@pytest.mark.parametrize("tag", CCA_CASES)
def test_oracle_cca_fit_matches_reference(gold, tag):
    from oracle import cca_np
    U, V, m1, m2, _ = cca_np.fit_f32(gold[tag + "/H1"], gold[tag + "/H2"])
    _check_cca(gold, tag, U, V, m1, m2)
    cca = cca_np.CCA(method="svd")
    cca.fit(gold[tag + "/H1"], gold[tag + "/H2"])
    _check_cca(gold, tag, cca.U, cca.V, cca.m1, cca.m2)


@pytest.mark.parametrize("tag", EVAL_CASES)
def test_oracle_eval_retrieval_matches_reference(gold, tag):
    from oracle import retrieval as oret
    lv1, lv2 = gold[tag + "/lv1"], gold[tag + "/lv2"]
    _check_eval(gold, tag, oret.eval_retrieval(lv1, lv2))
    dists = oret.cdist_cosine64(lv1, lv2)
    ranks, dstar, _ = oret.ranks_by_counting(dists)
    if lv1.shape[0] == lv2.shape[0]:
        _check_eval(gold, tag, oret.stats_from_ranks(ranks, dstar))


@pytest.mark.parametrize("tag", DTW_CASES)
def test_oracle_dtw_and_alignment_match_reference(gold, tag):
    from oracle import alignment as oa, retrieval as oret
    sheet, spec = gold[tag + "/sheet"], gold[tag + "/spec"]
    dists = oret.cdist_cosine64(sheet, spec)
    assert np.array_equal(dists, gold[tag + "/dists"])              # SciPy's cdist, bit for bit
    min_dist, _, acc, path = oa.dtw_by_dist(dists)
    assert min_dist == float(gold[tag + "/min_dist"])
    assert np.array_equal(acc, gold[tag + "/acc"])
    assert np.array_equal(path[0], gold[tag + "/path0"]) and np.array_equal(path[1], gold[tag + "/path1"])
    assert np.array_equal(oa.align_baseline(dists), gold[tag + "/baseline"])
    assert np.array_equal(oa.align_pydtw(dists), gold[tag + "/pydtw"])
    for how in ("baseline", "pydtw"):
        _, res = oa.compute_alignment(sheet, spec, gold[tag + "/sheet_idxs"], gold[tag + "/spec_idxs"], how)
        assert np.array_equal(res["aligned_sheet_idxs"], gold["%s/%s/aligned_idxs" % (tag, how)])
        assert np.array_equal(res["i_inter"], gold["%s/%s/i_inter" % (tag, how)])
        assert np.array_equal(res["a2s_alignment"], gold["%s/%s/a2s" % (tag, how)])


# ---- GPU: the library against the reference's outputs -------------------------------------------------------------
@pytest.fixture(scope="module")
def eng():
    from audio_sheet_retrieval_amd import _lib
    e = _lib.Engine("mutopia_ccal_cont")
    yield e
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CCA_CASES)
def test_device_cca_fit_matches_reference(gold, eng, tag):
    U, V, m1, m2, _ = eng.cca_fit(gold[tag + "/H1"], gold[tag + "/H2"])
    _check_cca(gold, tag, U, V, m1, m2)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CCA_CASES)
def test_host_cca_class_matches_reference(gold, eng, tag):
    from audio_sheet_retrieval_amd.utils.cca import CCA
    cca = CCA(method="svd", engine=eng)
    cca.fit(gold[tag + "/H1"], gold[tag + "/H2"])
    _check_cca(gold, tag, cca.U, cca.V, cca.m1, cca.m2)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", EVAL_CASES)
def test_device_eval_retrieval_matches_reference(gold, eng, tag):
    from audio_sheet_retrieval_amd.utils.train_dcca_pool import eval_retrieval
    _check_eval(gold, tag, eval_retrieval(gold[tag + "/lv1"], gold[tag + "/lv2"], engine=eng))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", DTW_CASES)
def test_device_dtw_and_alignment_match_reference(gold, eng, tag):
    from audio_sheet_retrieval_amd import alignment as al
    sheet, spec = gold[tag + "/sheet"], gold[tag + "/spec"]
    min_dist, dists, path = al.dtw_by_dist_codes(eng, sheet, spec)
    assert np.array_equal(dists, gold[tag + "/dists"])
    assert min_dist == float(gold[tag + "/min_dist"])
    assert np.array_equal(path[0], gold[tag + "/path0"]) and np.array_equal(path[1], gold[tag + "/path1"])
    positions, _ = al.align_pydtw(eng, sheet, spec)
    assert np.array_equal(positions, gold[tag + "/pydtw"])
    for how in ("baseline", "pydtw"):
        mapping, res = al.compute_alignment(eng, sheet, spec, gold[tag + "/sheet_idxs"], gold[tag + "/spec_idxs"], how)
        assert np.array_equal(res["aligned_sheet_idxs"], gold["%s/%s/aligned_idxs" % (tag, how)])
        assert np.array_equal(res["a2s_alignment"], gold["%s/%s/a2s" % (tag, how)])
        errors = al.estimate_alignment_error(gold["%s/%s/truth" % (tag, how)], gold["%s/%s/onsets" % (tag, how)],
                                             mapping)
        assert np.array_equal(errors, gold["%s/%s/errors" % (tag, how)])
