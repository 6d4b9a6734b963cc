"""Golden vectors (tests/golden/hotpath_golden.npz, made by make_golden.py from
the oracle): CPU - the oracle still reproduces them; GPU - the HIP path matches
them (embeddings <= 1e-4, integer ranks bit-exact)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz")
MODELS = ("mutopia_ccal_cont", "mutopia_ccal_cont_rsz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _inputs(gold, model):
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    sheet, spec = synth_data.synth_pairs(gold["indices"], seed=23)
    params = synth_data.synth_params(param_shapes(model), seed=1, trained_like=True)
    return sheet, spec, params


@pytest.mark.parametrize("model", MODELS)
def test_oracle_reproduces_golden_embeddings(gold, model):
    from oracle import network as onet
    sheet, spec, params = _inputs(gold, model)
    lv1, lv2 = onet.compute_output(onet.prepare(sheet, model), spec, params)
    assert np.abs(lv1 - gold[model + "/lv1"]).max() <= 2e-6
    assert np.abs(lv2 - gold[model + "/lv2"]).max() <= 2e-6


def test_oracle_reproduces_golden_ranks_and_cca(gold):
    from oracle import cca_np, retrieval as oret
    ranks, dstar, ties = oret.ranks_by_counting(oret.cdist_cosine64(gold["rank/lv1"], gold["rank/lv2"]))
    assert np.array_equal(ranks, gold["rank/ranks"]) and np.array_equal(dstar, gold["rank/dstar"])
    assert np.array_equal(ties, gold["rank/ties"]) and ties.sum() == 2
    U, V, m1, m2, coeffs = cca_np.fit_f32(gold["cca/H1"], gold["cca/H2"])
    s = np.sign(U[np.abs(U).argmax(axis=0), np.arange(32)])
    assert np.abs(U * s - gold["cca/U"]).max() <= 1e-5 and np.abs(V * s - gold["cca/V"]).max() <= 1e-5
    assert np.abs(coeffs - gold["cca/coeffs"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("model", MODELS)
def test_hip_matches_golden_embeddings(gold, model):
    from audio_sheet_retrieval_amd import _lib
    sheet, spec, params = _inputs(gold, model)
    eng = _lib.Engine(model)
    eng.set_params(params)
    assert np.abs(eng.embed_view1(sheet, prepared=False) - gold[model + "/lv1"]).max() <= 1e-4
    assert np.abs(eng.embed_view2(spec) - gold[model + "/lv2"]).max() <= 1e-4
    f1 = eng.embed_view1(sheet, prepared=False, features=True)
    assert np.abs(f1 - gold[model + "/feat1"]).max() <= 1e-4 * max(1.0, np.abs(gold[model + "/feat1"]).max())
    eng.close()


@pytest.mark.gpu
def test_hip_matches_golden_ranks(gold):
    from audio_sheet_retrieval_amd import _lib
    eng = _lib.Engine("mutopia_ccal_cont")
    ranks, dstar, ties = eng.rank(gold["rank/lv1"], gold["rank/lv2"])
    assert np.array_equal(ranks, gold["rank/ranks"])
    assert np.array_equal(dstar, gold["rank/dstar"])
    assert np.array_equal(ties, gold["rank/ties"])
    eng.close()
