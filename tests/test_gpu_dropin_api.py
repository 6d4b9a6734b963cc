"""GPU: the reference's own API on the fast path.

RetrievalWrapper.compute_view_1/2 (retrieval_wrapper.py:47-77), run_eval.main (run_eval.py:102-108), refine_cca.main
(refine_cca.py:92-107) and batch_compute1/2 (utils/batch_iterators.py:17-111) hand the library whole unprepared arrays
when the preparation is the model's own `prepare` (evaluated in the first kernel, input pipelined through page-locked
staging slots).  These tests pin that path BIT FOR BIT to the reference-shaped one: `prepare` on the host in NumPy,
chunks of 100 (10 for refine_cca), zero-padded last chunk - on uint8 and float32 inputs, `cont` and `_rsz`."""
import importlib
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SPLIT, CONFIG = "splits/all_split.yaml", "exp_configs/mutopia_full_aug.yaml"
TAG = "all_split_mutopia_full_aug"
MODELS = ["mutopia_ccal_cont", "mutopia_ccal_cont_rsz"]


@pytest.fixture()
def exp_root(tmp_path, monkeypatch):
    # every engine of a test takes the same convolution schedules: the tuner's timed picks may differ between two
    # contexts, and with them the float32 summation order ("Autotuner")
    monkeypatch.setenv("ASR_TUNE_CACHE", str(tmp_path / "tune_cache.txt"))
    from audio_sheet_retrieval_amd.config import settings
    from audio_sheet_retrieval_amd import run_eval, refine_cca
    import audio_sheet_retrieval_amd.run_train as rt
    for mod in (settings, run_eval, refine_cca, rt):
        monkeypatch.setattr(mod, "EXP_ROOT", str(tmp_path))
    return tmp_path


def _dump_params(exp_root, model_name):
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    params = synth_data.synth_params(param_shapes(model_name), seed=1, trained_like=True)
    d = exp_root / model_name
    d.mkdir(exist_ok=True)
    path = d / ("params_%s.pkl" % TAG)
    with open(path, "wb") as fp:
        pickle.dump(params, fp, protocol=2)
    return params, str(path)


def _unmarked(prepare):
    """the same preparation as a callable the library does not recognise: forces the host-side, chunked route"""
    return lambda x, y=None: prepare(x, y)


@pytest.mark.parametrize("model_name", MODELS)
@pytest.mark.parametrize("dtype", [np.uint8, np.float32])
def test_compute_view_fast_path_is_bit_identical_to_the_chunked_host_path(exp_root, model_name, dtype):
    from audio_sheet_retrieval_amd.retrieval_wrapper import RetrievalWrapper
    from audio_sheet_retrieval_amd.utils import synth_data
    _, path = _dump_params(exp_root, model_name)
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + model_name)
    n = 637                                           # 6 chunks of 100 + a ragged one; 2 granules + a ragged one
    sheet, spec = synth_data.synth_pairs(np.arange(n) + 5000, seed=23)
    sheet = sheet.astype(dtype)
    fast = RetrievalWrapper(model, path, prepare_view_1=model.prepare)
    slow = RetrievalWrapper(model, path, prepare_view_1=_unmarked(model.prepare))
    keep = sheet.copy()
    a1, a2 = fast.compute_view_1(sheet), fast.compute_view_2(spec)
    b1, b2 = slow.compute_view_1(sheet), slow.compute_view_2(spec)
    assert np.array_equal(sheet, keep)               # callers' arrays are never modified (retrieval_wrapper.py:54)
    assert a1.shape == (n, 32) and a1.dtype == np.float32
    assert np.array_equal(a1, b1), float(np.abs(a1 - b1).max())
    assert np.array_equal(a2, b2)
    # the reference's own loop spelled out: batch_compute2 with an opaque callable, chunk 100, zero-padded tail
    from audio_sheet_retrieval_amd.utils.batch_iterators import batch_compute2
    dummy = np.zeros((n, 1, 92, 42), np.float32)
    c1 = batch_compute2(sheet, dummy, lambda x, z: fast.compute_v1_latent(x, z), 100, prepare1=_unmarked(model.prepare))
    assert np.array_equal(a1, c1)
    # ... and the same call with the compiled function + the model's prepare takes the one-call route
    d1 = batch_compute2(sheet, dummy, fast.compute_v1_latent, 100, prepare1=model.prepare)
    assert np.array_equal(a1, d1)
    # other lengths: empty, one sample, exactly one granule
    assert fast.compute_view_1(sheet[:0]).shape == (0, 32)
    assert np.array_equal(fast.compute_view_1(sheet[:1]), a1[:1])
    assert np.array_equal(fast.compute_view_1(sheet[:250]), a1[:250])
    # a caller-prepared float array with prepare_view_1=None (the tutorials' usage)
    plain = RetrievalWrapper(model, path)
    assert np.array_equal(plain.compute_view_1(model.prepare(sheet[:130])), a1[:130])


def test_host_pipeline_granule_and_staging_switches_give_the_same_bits(exp_root):
    """ASR_HOST_GRANULE / ASR_HOST_STAGE / ASR_COPY_THREADS / ASR_HOST_PIPE only move the overlap around"""
    import subprocess
    import sys
    code = """
import numpy as np, sys
sys.path.insert(0, %r)
from audio_sheet_retrieval_amd import _lib
from audio_sheet_retrieval_amd.utils import synth_data
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
m = "mutopia_ccal_cont"
eng = _lib.Engine(m, device=0)
eng.set_params(synth_data.synth_params(param_shapes(m), seed=1, trained_like=True))
sheet, spec = synth_data.synth_pairs(np.arange(700), seed=23)
a = eng.embed_view1(sheet, prepared=False)
b = eng.embed_view2(spec)
p = eng.host_array(sheet.shape, sheet.dtype); p[...] = sheet          # page-locked caller memory: no staging copy
c = eng.embed_view1(p, prepared=False)
o1, o2 = eng.embed_both(sheet, spec)
assert np.array_equal(a, c) and np.array_equal(a, o1) and np.array_equal(b, o2)
np.save(sys.argv[1], np.concatenate([a, b]))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    cache = str(exp_root / "tune.txt")
    for k, env in enumerate([{}, {"ASR_HOST_GRANULE": "64", "ASR_HOST_STAGE": "1", "ASR_COPY_THREADS": "0"},
                             {"ASR_HOST_STAGE": "1"}, {"ASR_HOST_PIPE": "0"}, {"ASR_HOST_GRANULE_FIRST": "1000"},
                             {"ASR_HOST_GRANULE": "1000", "ASR_HOST_STAGE": "1", "ASR_COPY_THREADS": "7"}]):
        out = str(exp_root / ("o%d.npy" % k))
        subprocess.run([sys.executable, "-c", code, out], check=True, env=dict(os.environ, ASR_TUNE_CACHE=cache, **env))
        outs.append(np.load(out))
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


@pytest.mark.parametrize("model_name", MODELS)
def test_run_eval_and_refine_cca_fast_path_equal_the_chunked_route(exp_root, model_name, monkeypatch):
    from audio_sheet_retrieval_amd import refine_cca, run_eval
    params, path = _dump_params(exp_root, model_name)
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + model_name)
    common = ["--model", "models/%s.py" % model_name, "--data", "synthetic:330:50:215",
              "--train_split", SPLIT, "--config", CONFIG]
    refined_file = os.path.join(str(exp_root), model_name + "_est_UV", "params_%s.pkl" % TAG)

    out = refine_cca.main(common + ["--n_train", "325"])          # default chunk 10 like the reference
    assert out == refined_file
    fast_refined = pickle.load(open(out, "rb"))
    fast_eval = run_eval.main(common + ["--estimate_UV", "--n_test", "215"])

    # the same two drivers with a preparation the library does not recognise: prepare on the host, chunks of 10 / 100
    monkeypatch.setattr(model, "prepare", _unmarked(model.prepare))
    pickle.dump(params, open(path, "wb"), protocol=2)
    out = refine_cca.main(common + ["--n_train", "325"])
    slow_refined = pickle.load(open(out, "rb"))
    for a, b in zip(fast_refined, slow_refined):
        assert np.array_equal(a, b)
    slow_eval = run_eval.main(common + ["--estimate_UV", "--n_test", "215"])
    assert fast_eval == slow_eval


def test_eval_batches_rejects_wrong_output_buffers():
    """caller-supplied output arrays are validated before their pointers reach C"""
    from audio_sheet_retrieval_amd import _lib
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    m = "mutopia_ccal_cont"
    eng = _lib.Engine(m, device=0)
    eng.set_params(synth_data.synth_params(param_shapes(m), seed=1, trained_like=True))
    sheet, spec = synth_data.synth_pairs(np.arange(24), seed=23)
    good = eng.eval_batches([sheet, sheet], [spec, spec])
    for bad in (dict(ranks=[np.empty(24, np.int32)]),                                   # list too short
                dict(ranks=[np.empty(23, np.int32), np.empty(24, np.int32)]),           # too small
                dict(ranks=[np.empty(24, np.int64), np.empty(24, np.int64)]),           # wrong dtype
                dict(dstar=[np.empty(48, np.float64)[::2], np.empty(24, np.float64)])):  # not contiguous
        with pytest.raises(ValueError):
            eng.eval_batches([sheet, sheet], [spec, spec], out=bad)
    again = eng.eval_batches([sheet, sheet], [spec, spec], out=dict(ranks=[np.empty(24, np.int32) for _ in range(2)]))
    assert np.array_equal(again["ranks"][1], good["ranks"][1])
    eng.close()
