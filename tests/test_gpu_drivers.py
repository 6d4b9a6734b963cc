"""GPU: the reference's entry points keep working as a drop-in -
RetrievalWrapper.compute_view_1/2, refine_cca.py, run_eval.py on synthetic pools,
checked against the same pipeline evaluated by the oracle."""
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPLIT, CONFIG = "splits/all_split.yaml", "exp_configs/mutopia_full_aug.yaml"
TAG = "all_split_mutopia_full_aug"


@pytest.fixture()
def exp_root(tmp_path, monkeypatch):
    from audio_sheet_retrieval_amd.config import settings
    from audio_sheet_retrieval_amd import run_eval, refine_cca
    for mod in (settings, run_eval, refine_cca):
        monkeypatch.setattr(mod, "EXP_ROOT", str(tmp_path))
    import audio_sheet_retrieval_amd.run_train as rt
    monkeypatch.setattr(rt, "EXP_ROOT", str(tmp_path))
    return tmp_path


def _dump_params(exp_root, model_name):
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    params = synth_data.synth_params(param_shapes(model_name), seed=1, trained_like=True)
    d = exp_root / model_name
    d.mkdir()
    with open(d / ("params_%s.pkl" % TAG), "wb") as fp:
        pickle.dump(params, fp, protocol=2)          # py2-compatible like the reference's dumps
    return params


@pytest.mark.parametrize("model_name", ["mutopia_ccal_cont", "mutopia_ccal_cont_rsz"])
def test_retrieval_wrapper_matches_oracle(exp_root, model_name):
    import importlib
    from audio_sheet_retrieval_amd.retrieval_wrapper import RetrievalWrapper
    from audio_sheet_retrieval_amd.utils import synth_data
    from oracle import network as onet
    params = _dump_params(exp_root, model_name)
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + model_name)
    rw = RetrievalWrapper(model, str(exp_root / model_name / ("params_%s.pkl" % TAG)),
                          prepare_view_1=model.prepare, prepare_view_2=None)
    assert rw.code_dim == 32 and tuple(rw.shape_view2) == (1, 92, 42)
    sheet, spec = synth_data.synth_pairs(np.arange(7), seed=23)
    c1 = rw.compute_view_1(sheet)                     # uint8 like audio_sheet_server.py:331
    c2 = rw.compute_view_2(spec)
    r1, r2 = onet.compute_output(onet.prepare(sheet, model_name), spec, params)
    assert c1.shape == (7, 32) and np.abs(c1 - r1).max() <= 1e-4 and np.abs(c2 - r2).max() <= 1e-4
    c1b = rw.compute_view_1(sheet.astype(np.float32))  # float32 0..255 like the pools
    assert np.abs(c1b - r1).max() <= 1e-4


def test_refine_cca_then_run_eval(exp_root, capsys):
    from audio_sheet_retrieval_amd import refine_cca, run_eval
    from audio_sheet_retrieval_amd.utils import synth_data
    from oracle import cca_np, network as onet, retrieval as oret
    model_name = "mutopia_ccal_cont"
    params = _dump_params(exp_root, model_name)
    common = ["--model", "models/%s.py" % model_name, "--data", "synthetic:300:50:120",
              "--train_split", SPLIT, "--config", CONFIG]
    out = refine_cca.main(common + ["--n_train", "300", "--batch_size", "50"])
    assert os.path.exists(out) and out.endswith("%s_est_UV/params_%s.pkl" % (model_name, TAG))
    refined = pickle.load(open(out, "rb"))
    assert len(refined) == 97
    # oracle pipeline for the same 300 training pairs
    data = synth_data.load_synthetic_retrieval(300, 50, 120, seed=23)
    X1, X2 = data["train"][0:300]
    x = onet.prepare(X1, model_name)
    Ur, Vr, m1r, m2r, _ = cca_np.fit_f32(onet.features_view1(x, params), onet.features_view2(X2, params))
    s = np.sign((refined[90].astype(np.float64) * Ur).sum(axis=0))
    assert np.abs(refined[90] * s - Ur).max() <= 1e-3 * max(1.0, np.abs(Ur).max())
    assert np.abs(refined[92] - m1r).max() <= 1e-4
    for i in range(90):
        assert np.array_equal(refined[i], params[i])          # towers untouched

    res = run_eval.main(common + ["--estimate_UV", "--n_test", "100", "--dump_results"])
    res_a2s = run_eval.main(common + ["--estimate_UV", "--n_test", "100", "--V2_to_V1", "--max_dim", "16"])
    yaml_file = exp_root / (model_name + "_est_UV") / ("eval_%s_S2A.yaml" % TAG)
    assert yaml_file.exists()
    # oracle evaluation with the oracle's own refined projection
    p2 = [p.copy() for p in params]
    p2[90], p2[91], p2[92], p2[93] = Ur, Vr, m1r, m2r
    idx = np.linspace(0, 119, 100).astype(int)
    T1, T2 = data["test"][idx]
    lv1, lv2 = onet.compute_output(onet.prepare(T1, model_name), T2, p2)
    ref = oret.eval_retrieval(lv1, lv2)
    assert abs(res["map"] - ref[4]) <= 0.02 and abs(res["med_rank"] - ref[1]) <= 2
    ref_a2s = oret.eval_retrieval(lv2[:, :16], lv1[:, :16])
    assert abs(res_a2s["map"] - ref_a2s[4]) <= 0.02
    assert "Hit Rates" in capsys.readouterr().out


def test_run_train_two_epochs_then_eval(exp_root, capsys):
    """run_train.py (README.md:98) on a small synthetic pool: fit() trains from freshly drawn weights, keeps the
    best model, writes the reference's pickles; the dumped parameters load into run_eval."""
    from audio_sheet_retrieval_amd import run_eval, run_train
    model_name = "mutopia_ccal_cont"
    common = ["--model", "models/%s.py" % model_name, "--data", "synthetic:300:100:100",
              "--train_split", SPLIT, "--config", CONFIG]
    import audio_sheet_retrieval_amd.models.mutopia_ccal_cont as m
    import audio_sheet_retrieval_amd.utils.batch_iterators as bi
    orig = m.train_batch_iterator
    # 3 updates per sub-epoch instead of 100 keeps the test short (k_samples is a model constant, :203)
    m.train_batch_iterator = lambda batch_size=m.BATCH_SIZE: bi.MultiviewPoolIteratorUnsupervised(
        batch_size=batch_size, prepare=m.prepare, k_samples=300)
    try:
        best_map = run_train.main(common + ["--max_epochs", "2"])
    finally:
        m.train_batch_iterator = orig
    out = capsys.readouterr().out
    assert "Epoch 2 of 2" in out and "costs_tr" in out
    d = exp_root / model_name
    params = pickle.load(open(d / ("params_%s.pkl" % TAG), "rb"))
    results = pickle.load(open(d / ("results_%s.pkl" % TAG), "rb"))
    assert len(params) == 97 and len(results["pred_tr_err"]) == 2 and np.isfinite(results["pred_tr_err"]).all()
    assert results["pred_tr_err"][1] < results["pred_tr_err"][0]          # it learns
    assert 0.0 <= best_map <= 1.0
    assert np.abs(params[90]).max() > 0                                     # CCALayer wrote its projection
    res = run_eval.main(common + ["--n_test", "50"])
    assert 0.0 < res["map"] <= 1.0


def test_fit_refinement_restores_device_state_and_resume_continues(exp_root, capsys, monkeypatch):
    """fit() on the device through a forced refinement restart (utils/train_dcca_pool.py:492-520): with PATIENCE 1
    and an improvement rule that only the first epoch can meet, epoch 2 exhausts the patience -> the best parameters
    AND Adam's state (m, v, t) of epoch 1 are put back on the device, the learning rate is halved, training goes on;
    the parameter pickle holds the best model; `--resume` (run_train.py:96-101) starts the next run from it."""
    from audio_sheet_retrieval_amd import run_train
    from audio_sheet_retrieval_amd.utils import train_dcca_pool as tdp
    import audio_sheet_retrieval_amd.models.mutopia_ccal_cont as m
    import audio_sheet_retrieval_amd.utils.batch_iterators as bi
    model_name = "mutopia_ccal_cont"
    common = ["--model", "models/%s.py" % model_name, "--data", "synthetic:200:100:100",
              "--train_split", SPLIT, "--config", CONFIG]
    monkeypatch.setattr(m, "train_batch_iterator", lambda batch_size=m.BATCH_SIZE: bi.MultiviewPoolIteratorUnsupervised(
        batch_size=batch_size, prepare=m.prepare, k_samples=200))
    monkeypatch.setattr(m, "PATIENCE", 1)
    monkeypatch.setattr(m, "REFINEMENT_STEPS", 1)
    monkeypatch.setattr(m, "REFINEMENT_PATIENCE", 0, raising=False)

    # script the validation metric (the early-stopping rule is pinned against the reference on CPU; here the point
    # is what the restart does to the DEVICE state) and record the optimiser state / parameters around it
    real_train = tdp.train
    seen = []

    def scripted_train(iter_funcs, dataset, train_iter, valid_iter, fit_cca):
        for epoch in real_train(iter_funcs, dataset, train_iter, valid_iter, fit_cca):
            epoch["map_va"] = {1: 0.9}.get(epoch["number"], 0.1)
            eng = iter_funcs.engine
            seen.append(dict(number=epoch["number"], lr=float(iter_funcs.lr.get_value()), t=eng.get_opt_state()["t"],
                             w=eng.get_params()[0].copy(), m=eng.get_opt_state()["m"].copy()))
            yield epoch
    monkeypatch.setattr(tdp, "train", scripted_train)
    best_map = run_train.main(common + ["--max_epochs", "4"])
    out = capsys.readouterr().out
    assert "Early Stopping!" in out and "refining (1)" in out
    assert best_map == pytest.approx(0.9)
    assert [s["number"] for s in seen] == [1, 2, 3]            # epoch 3 runs after the restart, then patience ends it
    assert seen[0]["lr"] == pytest.approx(0.002) and seen[1]["lr"] == pytest.approx(0.002)
    assert seen[2]["lr"] == pytest.approx(0.001)               # LR_MULTIPLIER = 0.5
    # 2 updates per sub-epoch: t = 2 after epoch 1, 4 after epoch 2; the restart puts t back to 2 -> 4 after epoch 3
    assert [s["t"] for s in seen] == [2, 4, 4]
    assert not np.array_equal(seen[1]["w"], seen[0]["w"])
    d = exp_root / model_name
    params = pickle.load(open(d / ("params_%s.pkl" % TAG), "rb"))
    assert np.array_equal(params[0], seen[0]["w"])             # the pickle holds epoch 1's (best) parameters
    # --resume: the first epoch of the next run starts from the pickled parameters, not from a fresh draw
    seen.clear()
    monkeypatch.setattr(m, "REFINEMENT_STEPS", 0)
    starts = []
    real_create = tdp.create_iter_functions

    def recording_create(layers, *a, **k):
        funcs = real_create(layers, *a, **k)
        starts.append(funcs.engine.get_params()[0].copy())
        return funcs
    monkeypatch.setattr(tdp, "create_iter_functions", recording_create)
    run_train.main(common + ["--max_epochs", "1", "--resume", "--no_dump"])
    assert np.array_equal(starts[0], params[0])
    assert "resuming from" in capsys.readouterr().out


def test_fit_nan_loss_takes_the_patience_exit(exp_root, capsys, monkeypatch):
    """a NaN training loss (utils/train_dcca_pool.py:410-411) ends the run through the early-stopping branch and
    leaves the best (finite) model in place"""
    from audio_sheet_retrieval_amd import run_train
    from audio_sheet_retrieval_amd.utils import train_dcca_pool as tdp
    import audio_sheet_retrieval_amd.models.mutopia_ccal_cont as m
    import audio_sheet_retrieval_amd.utils.batch_iterators as bi
    common = ["--model", "models/mutopia_ccal_cont.py", "--data", "synthetic:200:100:100",
              "--train_split", SPLIT, "--config", CONFIG]
    monkeypatch.setattr(m, "train_batch_iterator", lambda batch_size=m.BATCH_SIZE: bi.MultiviewPoolIteratorUnsupervised(
        batch_size=batch_size, prepare=m.prepare, k_samples=200))
    monkeypatch.setattr(m, "REFINEMENT_STEPS", 0)
    # a learning rate large enough to blow the update up: the library reports the NaN loss, fit() handles it
    monkeypatch.setattr(m, "INI_LEARNING_RATE", 1e30)
    run_train.main(common + ["--max_epochs", "5"])
    out = capsys.readouterr().out
    d = exp_root / "mutopia_ccal_cont"
    results = pickle.load(open(d / ("results_%s.pkl" % TAG), "rb"))
    assert np.isnan(results["pred_tr_err"]).any()
    assert "Early Stopping!" in out and len(results["pred_tr_err"]) < 5


def test_fit_releases_the_training_state(exp_root, monkeypatch):
    """fit() ends with asr_train_end - afterwards the same network embeds snippets of another size
    (asr_set_input_size refuses that while a training state is alive) and holds the best model's parameters."""
    from audio_sheet_retrieval_amd import network
    from audio_sheet_retrieval_amd.utils import synth_data, train_dcca_pool as tdp
    import audio_sheet_retrieval_amd.models.mutopia_ccal_cont as m
    import audio_sheet_retrieval_amd.utils.batch_iterators as bi
    data = synth_data.load_synthetic_retrieval(200, 100, 100, seed=23)
    layers = m.build_model(show_model=False)
    created = []
    real_create = tdp.create_iter_functions
    monkeypatch.setattr(tdp, "create_iter_functions", lambda *a, **k: created.append(real_create(*a, **k)) or created[-1])
    tdp.fit(layers, data, m.objectives,
            train_batch_iter=bi.MultiviewPoolIteratorUnsupervised(batch_size=100, prepare=m.prepare, k_samples=200),
            valid_batch_iter=m.valid_batch_iterator(), num_epochs=1, patience=1, learn_rate=0.002,
            compute_updates=m.compute_updates, l_2=m.L2, l_1=m.L1, exp_name="t", out_path=str(exp_root / "t"),
            fit_cca=False)
    funcs = created[0]
    assert funcs.begun is False
    eng = funcs.engine
    best = network.get_all_param_values(layers)
    small = np.random.default_rng(0).random((5, 1, 80, 120)).astype(np.float32)       # another snippet size
    out = eng.embed_view1(small, prepared=True)
    assert out.shape == (5, 32) and np.isfinite(out).all()
    assert np.array_equal(network.get_all_param_values(layers)[0], best[0])
    eng.close()
