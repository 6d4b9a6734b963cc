"""Piece-identification vote: oracle restatement checks on CPU, device parity on the GPU."""
import numpy as np
import pytest


def test_oracle_vote_matches_literal_reference_procedure():
    from oracle import piece_vote as pv
    rng = np.random.default_rng(0)
    ids = rng.integers(0, 7, size=300)
    pieces, counts, votes = pv.vote(ids, top_k=3)
    unique, c = np.unique(ids, return_counts=True)
    assert counts.tolist() == sorted(c.tolist(), reverse=True)[:3]
    assert all(c[list(unique).index(p)] == k for p, k in zip(pieces, counts))
    assert abs(votes.sum() - 1.0) < 1e-12
    # ties: larger id first
    p2, c2, _ = pv.vote([4, 4, 9, 9, 1], top_k=3)
    assert p2.tolist() == [9, 4, 1] and c2.tolist() == [2, 2, 1]
    assert pv.window_starts(1000, 42, 100)[0] == 0 and pv.window_starts(1000, 42, 100)[-1] == 958
    src = np.arange(20 * 300, dtype=np.float32).reshape(20, 300)
    w = pv.slice_windows(src, 4, 8, 16, pv.window_starts(300, 16, 5))
    assert w.shape == (5, 1, 8, 16) and w[4, 0, 0, 0] == src[4, 284]


@pytest.mark.gpu
def test_device_vote_and_window_slicing_match_oracle():
    from audio_sheet_retrieval_amd import _lib
    from oracle import piece_vote as pv
    rng = np.random.default_rng(1)
    eng = _lib.Engine("mutopia_ccal_cont")
    # window slicing
    src = rng.random((100, 777)).astype(np.float32)
    starts = pv.window_starts(777, 42, 100).astype(np.int32)
    d_src, d_out = eng.alloc(src.nbytes).upload(src), eng.alloc(100 * 92 * 42 * 4)
    eng.slice_windows_dev(d_src.ptr, 100, 777, 3, 92, 42, starts, d_out.ptr)
    assert np.array_equal(d_out.download((100, 1, 92, 42), np.float32), pv.slice_windows(src, 3, 92, 42, starts))
    # vote: top-k indices -> piece ids -> histogram -> top pieces
    n_db, n_pieces, k = 5000, 37, 25
    db = rng.standard_normal((n_db, 32)).astype(np.float32)
    ids = np.sort(rng.integers(0, n_pieces, size=n_db)).astype(np.int32)
    q = (db[rng.integers(0, n_db, size=100)] + 0.5 * rng.standard_normal((100, 32))).astype(np.float32)
    d_db, d_ids, d_q = eng.alloc(db.nbytes).upload(db), eng.alloc(ids.nbytes).upload(ids), eng.alloc(q.nbytes).upload(q)
    d_idx, d_dist = eng.alloc(100 * k * 4), eng.alloc(100 * k * 8)
    eng.topk_dev(d_db.ptr, n_db, d_q.ptr, 100, k, d_idx.ptr, d_dist.ptr)
    for top_k in (1, 5, 64):
        pieces, counts = eng.piece_vote_dev(d_idx.ptr, 100 * k, d_ids.ptr, n_db, n_pieces, top_k)
        got_ids, _ = pv.retrieve_ids(db, ids, q, k)
        rp, rc, _ = pv.vote(got_ids, top_k)
        assert np.array_equal(pieces, rp) and np.array_equal(counts, rc), top_k
    eng.close()


@pytest.mark.gpu
def test_detect_score_end_to_end_matches_oracle(tmp_path):
    from audio_sheet_retrieval_amd import _lib, piece_identification as pid
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    from oracle import network as onet, piece_vote as pv
    rng = np.random.default_rng(2)
    model = "mutopia_ccal_cont"
    params = synth_data.synth_params(param_shapes(model), seed=1, trained_like=True)
    eng = _lib.Engine(model)
    eng.set_params(params)
    n_db, n_pieces = 600, 11
    codes = rng.standard_normal((n_db, 32)).astype(np.float32)
    codes /= np.linalg.norm(codes, axis=1, keepdims=True)
    ids = np.sort(rng.integers(0, n_pieces, size=n_db))
    db = pid.EmbeddingDB(eng, codes, ids, {i: "piece_%02d" % i for i in range(n_pieces)})
    path = str(tmp_path / "sheet_db.pkl")
    db.save(path)
    db2 = pid.EmbeddingDB.load(eng, path)
    assert np.array_equal(db2.codes, codes) and np.array_equal(db2.ids, ids) and db2.id_to_name[3] == "piece_03"
    spec = (3.0 * rng.random((92, 400)) ** 2).astype(np.float32)
    names, votes = pid.detect_score(eng, db2, spec, top_k=3, n_candidates=5, n_samples=20)
    # oracle: same windows -> oracle tower -> oracle top-k -> vote
    starts = pv.window_starts(400, 42, 20)
    win = pv.slice_windows(spec, 0, 92, 42, starts)
    q = onet.compute_v2_latent(win, params)
    got_ids, _ = pv.retrieve_ids(codes, ids, q, 5)
    rp, rc, rv = pv.vote(got_ids, 3)
    assert names == ["piece_%02d" % p for p in rp]
    assert np.allclose(votes, rv)
    eng.close()
