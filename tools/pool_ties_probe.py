"""Probe (GPU box): device gradients against the routed float64 oracle on synthetic / mostly-white pages under either
max-pool tie rule; prints the worst tensors.  Usage: python tools/pool_ties_probe.py <synthetic|white> <all|first> [B]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from audio_sheet_retrieval_amd import _lib
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    from oracle import network as onet
    from tests.test_gpu_pool_ties import whiten_pages
    from tests.test_gpu_train_routed import routed_gradient_errors
    pages, rule = sys.argv[1], sys.argv[2]
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    model = "mutopia_ccal_cont"
    sheet, spec = synth_data.synth_pairs(np.arange(B), seed=23)
    if pages == "white":
        sheet = whiten_pages(sheet)
    params = synth_data.synth_params(param_shapes(model), seed=1, trained_like=False)
    eng = _lib.Engine(model, pool_ties=rule)
    eng.set_params(params)
    eng.train_begin(B)
    x1 = onet.prepare(sheet, model)
    errs, loss, o_loss, routing, flips = routed_gradient_errors(eng, params, x1, spec)
    if os.environ.get("PROBE_DUMP"):
        from oracle import train as otrain
        flat, _ = eng.compute_gradients(x1, spec)
        ref = otrain.loss_and_grads(x1.astype(np.float64), spec.astype(np.float64), [q.astype(np.float64) for q in params],
                                    routing=routing, ties=rule)
        sizes = [int(np.prod(q.shape)) for q in params]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        for pi in (0, 1, 2):
            gi = otrain.TRAINABLE.index(pi)
            d = flat[offs[pi]:offs[pi + 1]].reshape(params[pi].shape).astype(np.float64)
            r = ref[2][gi]
            print("param %d: oracle" % pi, np.array2string(r.reshape(r.shape[0], -1)[:, :3].ravel()[:12], precision=5),
                  "\n   dev-oracle", np.array2string((d - r).reshape(r.shape[0], -1)[:, :3].ravel()[:12], precision=7),
                  "\n   max|oracle| %.4e" % np.abs(r).max())
    eng.close()
    top = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    print("%s %s B=%d env[%s]: worst %s; median %.2e; loss %.7f / %.7f; flips %.1e" % (
        pages, rule, B, " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ASR_")),
        ", ".join("p%d %.2e" % kv for kv in top), float(np.median(list(errs.values()))), loss, o_loss, flips))


if __name__ == "__main__":
    main()
