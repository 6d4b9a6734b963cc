#!/usr/bin/env python
"""Regenerate tests/golden/hotpath_golden.npz.

The reference ships no golden vectors and cannot run offline, so
these vectors are produced BY THE ORACLE (oracle/, the CPU restatement of the
reference path) on seeded synthetic inputs: they pin HIP <-> oracle and guard the
oracle against accidental edits; they do not pin HIP <-> Theano.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from audio_sheet_retrieval_amd.utils import synth_data  # noqa: E402
from audio_sheet_retrieval_amd.utils.param_layout import param_shapes  # noqa: E402
from oracle import cca_np, network as onet, retrieval as oret  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.npz")
N_EMBED = 6
INDICES = np.array([0, 1, 2, 1000, 123456, 2 ** 33 + 5], dtype=np.int64)


def main():
    g = {"indices": INDICES}
    for model in ("mutopia_ccal_cont", "mutopia_ccal_cont_rsz"):
        sheet, spec = synth_data.synth_pairs(INDICES, seed=23)
        params = synth_data.synth_params(param_shapes(model), seed=1, trained_like=True)
        x = onet.prepare(sheet, model)
        lv1, lv2 = onet.compute_output(x, spec, params)
        g[model + "/lv1"], g[model + "/lv2"] = lv1, lv2
        g[model + "/feat1"] = onet.features_view1(x, params)
        g[model + "/feat2"] = onet.features_view2(spec, params)
    rng = np.random.default_rng(20261002)
    a = rng.standard_normal((96, 32)).astype(np.float32)
    b = (rng.standard_normal((96, 32)) + 1.2 * a).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    b[40] = b[17]                               # one exact tie pair
    ranks, dstar, ties = oret.ranks_by_counting(oret.cdist_cosine64(a, b))
    g["rank/lv1"], g["rank/lv2"] = a, b
    g["rank/ranks"], g["rank/dstar"], g["rank/ties"] = ranks, dstar, ties
    z = rng.standard_normal((500, 32))
    H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((500, 32)) + 2.0).astype(np.float32)
    H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((500, 32)) - 1.0).astype(np.float32)
    U, V, m1, m2, coeffs = cca_np.fit_f32(H1, H2)
    s = np.sign(U[np.abs(U).argmax(axis=0), np.arange(32)])      # joint sign canonicalisation
    g["cca/H1"], g["cca/H2"] = H1, H2
    g["cca/U"], g["cca/V"], g["cca/m1"], g["cca/m2"], g["cca/coeffs"] = U * s, V * s, m1, m2, coeffs
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
