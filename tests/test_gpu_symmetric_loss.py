"""GPU: get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69) inside the fused step.

The two models bind weight 1, one direction (models/mutopia_ccal_cont.py:152-155); the symmetric direction (:53-65) and
the weight (:67) are part of the cited function and run on the device since round 5 (asr_set_objective): direction 2 is
the same pair pass with the views swapped, its gradients added.  Checked against oracle.train.contrastive_cos_loss."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("weight,symmetric", [(1.0, True), (0.5, True), (2.0, False)])
def test_cca_loss_stage_with_symmetric_direction_and_weight(weight, symmetric):
    """CCALayer + LengthNorm + loss stage alone (asr_cca_train_debug): loss 1e-6, dL/dlv... observed through dL/dH 1e-5
    of its maximum, against the float64 oracle; B not a multiple of the kernels' row groups"""
    from audio_sheet_retrieval_amd import _lib
    from oracle import train as otrain
    rng = np.random.default_rng(8)
    B = 203
    z = rng.standard_normal((B, 32))
    H1 = (z @ rng.standard_normal((32, 32)) + 0.7 * rng.standard_normal((B, 32))).astype(np.float32)
    H2 = (z @ rng.standard_normal((32, 32)) + 0.7 * rng.standard_normal((B, 32))).astype(np.float32)
    cca0 = [np.zeros((32, 32), np.float32), np.zeros((32, 32), np.float32), np.zeros(32, np.float32), np.zeros(32, np.float32),
            np.zeros((32, 32), np.float32), np.zeros((32, 32), np.float32), np.zeros((32, 32), np.float32)]
    eng = _lib.Engine("mutopia_ccal_cont")
    eng.set_objective(weight, 0.7, symmetric)
    got = eng.cca_train_debug(H1, H2, cca0, backward=True)
    eng.close()
    H1d, H2d = H1.astype(np.float64), H2.astype(np.float64)
    out1, out2, corr, new, cache = otrain.cca_train_fwd(H1d, H2d, [c.astype(np.float64) for c in cca0])
    lv1 = out1 / np.linalg.norm(out1, axis=1, keepdims=True)
    lv2 = out2 / np.linalg.norm(out2, axis=1, keepdims=True)
    loss, dlv1, dlv2 = otrain.contrastive_cos_loss(lv1, lv2, 0.7, weight=weight, symmetric=symmetric)
    dH1, dH2 = otrain.cca_train_bwd(cache, otrain.length_norm_bwd(out1, dlv1), otrain.length_norm_bwd(out2, dlv2))
    g_loss, g_dH1, g_dH2 = got["loss"], got["dH1"], got["dH2"]
    print("weight %.1f symmetric %s: loss %.7f vs %.7f; dH1 err %.1e dH2 err %.1e" % (
        weight, symmetric, g_loss, float(loss), np.abs(g_dH1 - dH1).max() / np.abs(dH1).max(),
        np.abs(g_dH2 - dH2).max() / np.abs(dH2).max()))
    assert abs(g_loss - float(loss)) <= 1e-6 * max(1.0, abs(float(loss)))
    assert np.abs(g_dH1 - dH1).max() <= 1e-5 * np.abs(dH1).max()
    assert np.abs(g_dH2 - dH2).max() <= 1e-5 * np.abs(dH2).max()
    # the one-directional, weight-1 loss is a different number (0.5 x two directions is nearly the same VALUE, not the
    # same gradient)
    base = otrain.contrastive_cos_loss(lv1, lv2, 0.7)
    if weight * (2 if symmetric else 1) != 1.0:
        assert abs(float(loss) - float(base[0])) > 1e-3
    assert np.abs(dlv1 - base[1]).max() > 0.05 * np.abs(base[1]).max()


def test_create_iter_functions_accepts_symmetric_objective():
    """create_iter_functions(objectives = symmetric loss): train / valid / compute_gradients follow the oracle with
    symmetric=True (loss 1e-5, all 54 gradient tensors 1e-4 of their maximum with the device's pooling sets imposed)"""
    from audio_sheet_retrieval_amd import network
    from audio_sheet_retrieval_amd.models import mutopia_ccal_cont as model
    from audio_sheet_retrieval_amd.models.objectives import get_contrastive_cos_loss
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.train_dcca_pool import create_iter_functions
    from oracle import network as onet, train as otrain
    from tests.test_gpu_train_routed import device_routing
    B = 48
    layers = model.build_model(show_model=False)
    net = layers[0].net
    params = synth_data.synth_params(net.shapes, seed=1, trained_like=False)
    network.set_all_param_values(layers, params)
    funcs = create_iter_functions(layers, lambda: get_contrastive_cos_loss(1.0, model.GAMMA, symmetric=True),
                                  model.compute_updates, 0.002, model.L2, model.L1)
    sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
    x1 = onet.prepare(sheet, "mutopia_ccal_cont")
    grads = funcs["compute_gradients"](x1, spec)
    eng = funcs.engine
    routing = device_routing(eng, params, B, x1.shape[2:], spec.shape[2:])
    p64 = [q.astype(np.float64) for q in params]
    ref = otrain.loss_and_grads(x1.astype(np.float64), spec.astype(np.float64), p64, routing=routing, ties=eng.pool_ties,
                                symmetric=True)
    one = otrain.loss_and_grads(x1.astype(np.float64), spec.astype(np.float64), p64, routing=routing, ties=eng.pool_ties)
    gmax = max(float(np.abs(g).max()) for g in ref[2])
    errs = [float(np.abs(g - r).max() / max(1e-3 * gmax, np.abs(r).max())) for g, r in zip(grads, ref[2])]
    differs = max(float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(ref[2][:3], one[2][:3]))
    # valid: the deterministic graph with the CCA projection compute_gradients' default updates just stored
    v = float(funcs["valid"](x1, spec)[0])
    now = [q.astype(np.float32) for q in eng.get_params()]
    v_ref = float(otrain.valid_loss(x1, spec, now, symmetric=True))
    v_one = float(otrain.valid_loss(x1, spec, now))
    assert abs(v_ref - v_one) > 1e-3
    eng.set_params(params)                       # back to the start: the train step is compared with `ref`
    loss = float(funcs["train"](x1, spec)[0])
    funcs.close()
    print("symmetric objective: gradient errors worst %.1e; train loss %.7f vs %.7f; valid %.7f vs %.7f; symmetric vs "
          "one-directional gradients differ by %.2f" % (max(errs), loss, float(ref[0]), v, v_ref, differs))
    assert max(errs) <= 1e-4, errs
    assert abs(loss - float(ref[0])) <= 1e-5 and abs(v - v_ref) <= 1e-4 * max(1.0, abs(v_ref))
    assert differs > 0.05


def test_set_objective_checks_its_arguments_and_debug_exports_refuse_unpooled_blocks():
    """asr_set_objective: weight > 0, symmetric in {0, 1} - anything else is ASR_ERR_INVALID with a message, the previous
    objective stays; asr_debug_train_tensor kind 10 (tie sets) exists for pooled blocks only and needs a training state"""
    import ctypes
    from audio_sheet_retrieval_amd import _lib
    eng = _lib.Engine("mutopia_ccal_cont")
    for w, g, sy in ((0.0, 0.7, 0), (-1.0, 0.7, 0), (1.0, 0.7, 2), (float("nan"), 0.7, 0)):
        rc = eng.lib.asr_set_objective(eng.ctx, ctypes.c_float(w), ctypes.c_float(g), sy)
        assert rc == _lib.ASR_ERR_INVALID, (w, g, sy)
        assert b"set_objective" in eng.lib.asr_last_error(eng.ctx)
    eng.set_objective(2.0, 0.5, True)
    with pytest.raises(_lib.AsrError):
        eng.debug_train_tensor("pool_mask", view=1, index=1, batch=4)          # no training state yet
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    eng.set_params(synth_data.synth_params(param_shapes("mutopia_ccal_cont"), seed=1, trained_like=False))
    eng.train_begin(4)
    for blk in (0, 2, 4, 6):                                                   # blocks 1, 3, 5, 7 are not pooled
        with pytest.raises(_lib.AsrError) as ei:
            eng.debug_train_tensor("pool_mask", view=1, index=blk, batch=4)
        assert ei.value.code == _lib.ASR_ERR_INVALID
    eng.close()
