#!/usr/bin/env python
"""Golden vectors produced by RUNNING THE REFERENCE ITSELF (build container only; /root/reference never travels).

    python tests/golden/make_reference_golden.py        # -> tests/golden/reference_golden.npz

Most of the reference cannot run offline (Python-2 source on Theano/Lasagne/cv2/madmom/msmd, none installed), but
four NumPy/SciPy-only pieces of the hot path can, and they are executed here on seeded inputs:

  CCA.fit(method='svd')      audio_sheet_retrieval/utils/cca.py:24-...   (refine_cca.py's estimator; the file has
                             Python-2 print statements / xrange - it is converted IN MEMORY with lib2to3's `print`
                             and `xrange` fixers, nothing else is touched and nothing is written)
  eval_retrieval             utils/train_dcca_pool.py:28-82              (the module imports theano/lasagne at the
                             top, so only this function's source segment is compiled, with numpy as `np`; `xrange`
                             converted as above.  The sizes used keep n_v2 / n_v1 integral: Python 2's `/` on
                             ints and Python 3's agree there)
  dtw_by_dist                utils/dtw_by_dist.py:5-34                    (imported as is)
  align_baseline, align_pydtw, compute_alignment, estimate_alignment_error
                             utils/alignment.py:112-190                   (`xrange` converted; `np.int`, removed
                             from NumPy 1.24+, is restored as the builtin it always aliased)

The file holds inputs and the reference's outputs only.  tests/test_reference_golden.py checks the oracle (CPU) and
the HIP library (GPU) against them.  Theano-side code (network forward, CCALayer, loss, updates) stays unpinned.
"""
import ast
import os
import sys
from lib2to3 import refactor

import numpy as np

REF = "/root/reference/audio_sheet_retrieval"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_golden.npz")
_TOOL = refactor.RefactoringTool(["lib2to3.fixes.fix_print", "lib2to3.fixes.fix_xrange"])


def _py3(source, name):
    return str(_TOOL.refactor_string(source if source.endswith("\n") else source + "\n", name))


def _module(path, namespace=None):
    ns = dict(namespace or {}, __name__="reference_" + os.path.basename(path)[:-3])
    with open(path) as fp:
        exec(compile(_py3(fp.read(), path), path, "exec"), ns)
    return ns


def _function(path, name, namespace):
    with open(path) as fp:
        source = fp.read()
    node = [n for n in ast.parse(source).body if isinstance(n, ast.FunctionDef) and n.name == name][0]
    segment = "\n".join(source.splitlines()[node.lineno - 1:node.end_lineno])
    ns = dict(namespace)
    exec(compile(_py3(segment, path), path + ":" + name, "exec"), ns)
    return ns[name]


def _unit_rows(rng, n, dim, noise, base=None):
    x = rng.standard_normal((n, dim)) if base is None else base + noise * rng.standard_normal((n, dim))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def main():
    if not hasattr(np, "int"):
        np.int = int                                         # alias removed in NumPy 1.24
    sys.path.insert(0, os.path.join(REF, "utils"))           # alignment.py does `from dtw_by_dist import ...`
    out = {}
    rng = np.random.RandomState(20260)

    # ---- CCA.fit ------------------------------------------------------------------------------------------------
    CCA = _module(os.path.join(REF, "utils", "cca.py"))["CCA"]
    for tag, n in (("cca_a", 500), ("cca_b", 1200)):
        z = rng.standard_normal((n, 32))
        H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32)) + 0.3).astype(np.float32)
        H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32)) - 0.2).astype(np.float32)
        cca = CCA(method="svd")
        cca.fit(H1, H2, verbose=False)
        out.update({tag + "/H1": H1, tag + "/H2": H2, tag + "/m1": cca.m1, tag + "/m2": cca.m2,
                    tag + "/U": cca.U, tag + "/V": cca.V})

    # ---- eval_retrieval -----------------------------------------------------------------------------------------
    eval_retrieval = _function(os.path.join(REF, "utils", "train_dcca_pool.py"), "eval_retrieval", {"np": np})
    for tag, n1, n2, noise in (("eval_a", 300, 300, 0.25), ("eval_b", 1000, 1000, 0.6), ("eval_c", 200, 400, 0.4)):
        lv1 = _unit_rows(rng, n1, 32, 0.0)
        base = np.repeat(lv1, n2 // n1, axis=0).astype(np.float64)
        lv2 = _unit_rows(rng, n2, 32, noise, base=base)
        mean_rank, median_rank, mean_dist, hits, mean_ap = eval_retrieval(lv1, lv2)
        out.update({tag + "/lv1": lv1, tag + "/lv2": lv2,
                    tag + "/stats": np.array([mean_rank, median_rank, mean_dist, mean_ap], np.float64),
                    tag + "/hits": np.array([hits[1], hits[5], hits[10], hits[25]], np.int64)})

    # ---- DTW + alignment ----------------------------------------------------------------------------------------
    from dtw_by_dist import dtw_by_dist
    align = _module(os.path.join(REF, "utils", "alignment.py"))
    for tag, n_sheet, n_spec in (("dtw_tall", 90, 60), ("dtw_wide", 50, 120), ("dtw_square", 64, 64)):
        # a noisy monotone correspondence between sheet positions and audio excerpts
        sheet = _unit_rows(rng, n_sheet, 32, 0.0)
        walk = np.sort(rng.randint(0, n_sheet, n_spec))
        spec = _unit_rows(rng, n_spec, 32, 0.35, base=sheet[walk].astype(np.float64))
        from scipy.spatial.distance import cdist
        dists = cdist(sheet, spec, metric="cosine")
        min_dist, cost, acc, path = dtw_by_dist(dists.copy())
        sheet_idxs = np.cumsum(rng.randint(3, 9, n_sheet)).astype(np.int64)
        spec_idxs = np.cumsum(rng.randint(0, 3, n_spec)).astype(np.int64) + 5
        out.update({tag + "/sheet": sheet, tag + "/spec": spec, tag + "/dists": dists,
                    tag + "/min_dist": np.float64(min_dist), tag + "/acc": np.array(acc),
                    tag + "/path0": np.asarray(path[0], np.int64), tag + "/path1": np.asarray(path[1], np.int64),
                    tag + "/sheet_idxs": sheet_idxs, tag + "/spec_idxs": spec_idxs,
                    tag + "/baseline": align["align_baseline"](dists),
                    tag + "/pydtw": np.asarray(align["align_pydtw"](dists.copy()), np.int64)})
        for how in ("baseline", "pydtw"):
            mapping, res = align["compute_alignment"](sheet, spec, sheet_idxs, spec_idxs, how)
            onsets = np.arange(spec_idxs[0] - 2, spec_idxs[-1] + 3, 2)
            truth = np.interp(onsets, spec_idxs, sheet_idxs[np.minimum(walk, n_sheet - 1)])
            errors = align["estimate_alignment_error"](truth, onsets, mapping)
            out.update({"%s/%s/i_inter" % (tag, how): res["i_inter"],
                        "%s/%s/a2s" % (tag, how): res["a2s_alignment"],
                        "%s/%s/aligned_idxs" % (tag, how): res["aligned_sheet_idxs"].astype(np.int64),
                        "%s/%s/onsets" % (tag, how): onsets, "%s/%s/truth" % (tag, how): truth,
                        "%s/%s/errors" % (tag, how): errors})

    np.savez_compressed(OUT, **out)
    print("%d arrays, %.1f KiB -> %s" % (len(out), os.path.getsize(OUT) / 1024.0, OUT))


if __name__ == "__main__":
    main()
