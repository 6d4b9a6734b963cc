"""Alignment: oracle DTW self-checks on CPU; device distance matrix / DTW path / alignment == oracle."""
import numpy as np
import pytest


def _codes(rng, n_sheet, n_spec, noise=0.3):
    base = rng.standard_normal((max(n_sheet, n_spec), 32))
    s = base[np.linspace(0, len(base) - 1, n_sheet).astype(int)]
    a = base[np.linspace(0, len(base) - 1, n_spec).astype(int)] + noise * rng.standard_normal((n_spec, 32))
    f = lambda x: (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    return f(s), f(a)


def test_oracle_dtw_properties():
    from oracle import alignment as oa
    rng = np.random.default_rng(0)
    d = rng.random((7, 5))
    md, C, D1, path = oa.dtw_by_dist(d)
    assert np.array_equal(C, d)
    p, q = path                       # not transposed: (cols, rows) order as the reference returns it
    assert q[0] == 0 and p[0] == 0 and q[-1] == 6 and p[-1] == 4
    assert np.all(np.diff(q) >= 0) and np.all(np.diff(p) >= 0) and np.all(np.diff(q) + np.diff(p) >= 1)
    assert abs(D1[-1, -1] - sum(d[i, j] for i, j in zip(q, p))) < 1e-12
    # identical sequences: the diagonal
    e = 1.0 - np.eye(6)
    _, _, _, (p2, q2) = oa.dtw_by_dist(e)
    assert np.array_equal(p2, np.arange(6)) and np.array_equal(q2, np.arange(6))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 90), (120, 50), (64, 64), (1, 9), (300, 700)])
def test_device_dtw_matches_oracle_bit_exact(shape):
    from audio_sheet_retrieval_amd import _lib, alignment as al
    from oracle import alignment as oa, retrieval as oret
    rng = np.random.default_rng(3)
    sheet, spec = _codes(rng, *shape)
    eng = _lib.Engine("mutopia_ccal_cont")
    md, dists, path = al.dtw_by_dist_codes(eng, sheet, spec)
    ref_d = oret.cdist_cosine64(sheet, spec)
    assert np.array_equal(dists, ref_d)
    rmd, _, _, rpath = oa.dtw_by_dist(ref_d)
    assert md == rmd
    assert np.array_equal(path[0], rpath[0]) and np.array_equal(path[1], rpath[1])
    eng.close()


@pytest.mark.gpu
def test_compute_alignment_matches_oracle_and_cca_workspace_survives_larger_calls():
    from audio_sheet_retrieval_amd import _lib, alignment as al
    from oracle import alignment as oa
    rng = np.random.default_rng(5)
    sheet, spec = _codes(rng, 150, 400, noise=0.2)
    sheet_idxs = np.linspace(100, 5000, 150).astype(np.int64)
    spec_idxs = np.sort(rng.choice(np.arange(10, 4000), size=400, replace=False))
    eng = _lib.Engine("mutopia_ccal_cont")
    for by in ("pydtw", "baseline"):
        m, res = al.compute_alignment(eng, sheet, spec, sheet_idxs, spec_idxs, by)
        rm, rres = oa.compute_alignment(sheet, spec, sheet_idxs, spec_idxs, by)
        assert np.array_equal(res["aligned_sheet_idxs"], rres["aligned_sheet_idxs"]), by
        assert np.array_equal(res["a2s_alignment"], rres["a2s_alignment"]), by
        assert m == rm
    err = al.estimate_alignment_error(sheet_idxs[:10].astype(float), res["i_inter"][:10], m)
    assert err.shape == (10,)
    # regression: a rank / top-k / dtw call that grows the norm buffers must leave the CCA-fit workspace alone
    H = rng.standard_normal((500, 32)).astype(np.float32)
    G = (H @ rng.standard_normal((32, 32)) * 0.3 + rng.standard_normal((500, 32))).astype(np.float32)
    U1 = eng.cca_fit(H, G)[0]
    big = rng.standard_normal((5000, 32)).astype(np.float32)
    eng.rank(big, big)
    U2 = eng.cca_fit(H, G)[0]
    assert np.array_equal(U1, U2)
    eng.close()
