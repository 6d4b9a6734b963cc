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

  detect_score, detect_performance, _retrieve_sheet_snippet_ids, _retrieve_perform_excerpt_ids
                             audio_sheet_server.py:213-300, :530-563      (methods of a class whose module needs
                             cv2 / madmom / msmd: the four method bodies are compiled on their own and bound to a
                             plain attribute holder carrying the code data base.  `self.embed_network` - Theano in
                             the reference - is a fixed random projection + L2 norm defined HERE; it is not what
                             is pinned.  Pinned: window slicing, top-n_candidates retrieval, vote counting and
                             vote normalisation, on the codes that projection produces, which are stored.)

  AudioScoreRetrievalPool    utils/data_pools.py:36-228                   (the module imports cv2 / msmd / matplotlib;
                             the class and the module-level constants are compiled on their own with numpy and
                             scipy's interp1d.  Run with `sheet_scaling` off - that branch needs cv2 - so pinned
                             are: interpolate, prepare_train_entities, the shuffle, the window arithmetic, the
                             system / onset translations, the spectrogram padding shift and the ORDER of the random
                             draws; the nearest-neighbour rescaling stays unpinned)

  batch_compute1, batch_compute2, MultiviewPoolIteratorUnsupervised
                             utils/batch_iterators.py:17-111, :163-221    (pure NumPy; `xrange` converted, and the one
                             Python-2 integer division of the iterator, `range((n_samples + bs - 1) / bs)` (:193), is
                             written `//` in memory - the value Python 2 computes.  Run with recording fake `compute`
                             callables and a fake pool (defined in tests/golden/iterator_fakes.py, shared with the
                             test): pinned are the chunking, the zero-padding of the last chunk, which rows are kept,
                             the sub-epoch windows, the wrap-around fill of short batches, the epoch counter and the
                             point at which the pool is reshuffled.  -> tests/golden/reference_golden_iter.npz)

  fit                        utils/train_dcca_pool.py:318-543            (only this function's source segment is
                             compiled.  Its collaborators - the epoch generator `train`, `create_iter_functions`,
                             lasagne's get/set_all_param_values, theano.shared - are the scripted stand-ins of
                             tests/golden/fit_fakes.py and are NOT what is pinned; `open(path, 'w')` is given Python 2's
                             meaning (binary-safe) for the pickles.  Pinned: the early-stopping rule, the best-model /
                             optimiser-state snapshots, refinement restarts and learning-rate handling, the NaN exit,
                             the epoch limit, the history and parameter pickles.  -> reference_golden_fit.npz)

  CCA.fit / eval_retrieval AT THE SIZES BASELINE.json QUOTES
                             the same two reference functions on the seeded inputs of tests/golden/size_inputs.py:
                             25 000 x 32 features (refine_cca.py --n_train 25000), 1000 and 2000 unit codes per side
                             (configs[1]; eval_models.sh:15).  Only the reference's OUTPUTS are stored; the test
                             regenerates the inputs.  -> reference_golden_sizes.npz

The files hold inputs and the reference's outputs only.  tests/test_reference_golden.py checks the oracle (CPU) and
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


def _class_with_constants(path, class_name, namespace):
    with open(path) as fp:
        source = fp.read()
    lines = source.splitlines()
    keep = [n for n in ast.parse(source).body
            if isinstance(n, ast.Assign) or (isinstance(n, ast.ClassDef) and n.name == class_name)]
    text = "\n".join("\n".join(lines[n.lineno - 1:n.end_lineno]) for n in keep)
    ns = dict(namespace)
    exec(compile(_py3(text, path), path + ":" + class_name, "exec"), ns)
    return ns[class_name]


def _methods(path, class_name, names, namespace):
    with open(path) as fp:
        source = fp.read()
    cls = [n for n in ast.parse(source).body if isinstance(n, ast.ClassDef) and n.name == class_name][0]
    lines = source.splitlines()
    found = {}
    for node in cls.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            body = "\n".join(line[4:] for line in lines[node.lineno - 1:node.end_lineno])      # drop class indent
            ns = dict(namespace)
            exec(compile(_py3(body, path), path + ":" + node.name, "exec"), ns)
            found[node.name] = ns[node.name]
    return found


class _Projection(object):
    """stands where the reference has its Theano network: flatten -> fixed matrix -> unit length"""

    def __init__(self, rng, n1, n2):
        self.w1 = rng.standard_normal((n1, 32)) / np.sqrt(n1)
        self.w2 = rng.standard_normal((n2, 32)) / np.sqrt(n2)
        self.seen = {}

    def _run(self, x, w, key):
        y = x.reshape(x.shape[0], -1).astype(np.float64) @ w
        y = (y / np.linalg.norm(y, axis=1, keepdims=True)).astype(np.float32)
        self.seen[key] = (x.copy(), y)
        return y

    def compute_view_1(self, x):
        return self._run(x, self.w1, "view1")

    def compute_view_2(self, x):
        return self._run(x, self.w2, "view2")


class _Holder(object):
    pass


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
    for tag, n in (("cca_a", 400), ("cca_b", 900)):
        z = rng.standard_normal((n, 32))
        H1 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32)) + 0.3).astype(np.float32)
        H2 = (z @ rng.standard_normal((32, 32)) + 0.5 * rng.standard_normal((n, 32)) - 0.2).astype(np.float32)
        cca = CCA(method="svd")
        cca.fit(H1, H2, verbose=False)
        out.update({tag + "/H1": H1, tag + "/H2": H2, tag + "/m1": cca.m1, tag + "/m2": cca.m2,
                    tag + "/U": cca.U, tag + "/V": cca.V})

    # ---- eval_retrieval -----------------------------------------------------------------------------------------
    eval_retrieval = _function(os.path.join(REF, "utils", "train_dcca_pool.py"), "eval_retrieval", {"np": np})
    for tag, n1, n2, noise in (("eval_a", 300, 300, 0.25), ("eval_b", 700, 700, 0.6), ("eval_c", 200, 400, 0.4)):
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

    # ---- piece identification (server methods) ------------------------------------------------------------------
    from scipy.spatial.distance import cdist
    names = ("detect_score", "detect_performance", "_retrieve_sheet_snippet_ids", "_retrieve_perform_excerpt_ids")
    methods = _methods(os.path.join(REF, "audio_sheet_server.py"), "AudioSheetServer", names,
                       {"np": np, "cdist": cdist})
    if not hasattr(np, "float"):
        np.float = float                                     # alias removed in NumPy 1.24
    srv = _Holder()
    for name, fn in methods.items():
        setattr(srv, name, fn.__get__(srv))
    srv.spec_shape, srv.sheet_shape = (92, 42), (40, 50)      # small snippet shape keeps the fixture small
    srv.embed_network = _Projection(rng, 40 * 50, 92 * 42)
    for tag, n_pieces, per_piece, n_cand, top_k in (("vote_a", 6, 40, 1, 3), ("vote_b", 12, 25, 5, 5),
                                                    ("vote_c", 30, 10, 25, 10)):
        # data base: per piece a cluster of codes; the query material is generated from piece `target`
        centers = _unit_rows(rng, n_pieces, 32, 0.0).astype(np.float64)
        ids = np.repeat(np.arange(n_pieces), per_piece)
        db = _unit_rows(rng, len(ids), 32, 0.45, base=centers[ids])
        id_to_name = dict((i, "piece_%02d" % i) for i in range(n_pieces))
        srv.sheet_snippet_codes, srv.sheet_snippet_ids, srv.id_to_piece = db, ids, id_to_name
        srv.perform_excerpt_codes, srv.perform_excerpt_ids, srv.id_to_perform = db, ids, id_to_name
        # integer-valued so that the fixture can hold them as uint8
        spectrogram = rng.randint(0, 256, (92, 300 + 13 * n_pieces)).astype(np.float32)
        sheet = rng.randint(0, 256, (64, 400 + 7 * n_pieces)).astype(np.float32)
        names_s, votes_s = srv.detect_score(spectrogram, top_k=top_k, n_candidates=n_cand)
        windows2, codes2 = srv.embed_network.seen["view2"]
        names_p, votes_p = srv.detect_performance(sheet, top_k=top_k, n_candidates=n_cand)
        windows1, codes1 = srv.embed_network.seen["view1"]
        out.update({tag + "/db": db, tag + "/ids": ids.astype(np.int64), tag + "/n_cand": np.int64(n_cand),
                    tag + "/top_k": np.int64(top_k), tag + "/spectrogram": spectrogram.astype(np.uint8), tag + "/sheet": sheet.astype(np.uint8),
                    tag + "/spec_codes": codes2, tag + "/sheet_codes": codes1,
                    tag + "/spec_window_sums": windows2.reshape(len(windows2), -1).sum(axis=1, dtype=np.float64),
                    tag + "/sheet_window_sums": windows1.reshape(len(windows1), -1).sum(axis=1, dtype=np.float64),
                    tag + "/score_pieces": np.array([int(n[-2:]) for n in names_s], np.int64),
                    tag + "/score_votes": np.asarray(votes_s, np.float64),
                    tag + "/perform_pieces": np.array([int(n[-2:]) for n in names_p], np.int64),
                    tag + "/perform_votes": np.asarray(votes_p, np.float64)})

    # ---- data pool ----------------------------------------------------------------------------------------------
    from scipy.interpolate import interp1d
    Pool = _class_with_constants(os.path.join(REF, "utils", "data_pools.py"), "AudioScoreRetrievalPool",
                                 {"np": np, "interp1d": interp1d})
    n_bins, spec_ctx, sheet_ctx, staff = 24, 42, 50, 40
    images, specs, o2c = [], [], []
    for piece in range(3):
        width = 360 + 70 * piece
        images.append(rng.randint(0, 256, (52, width)).astype(np.float32))
        per_piece_specs, per_piece_maps = [], []
        for perf in range(2):
            frames = 260 + 40 * perf + 30 * piece
            per_piece_specs.append(rng.randint(0, 256, (n_bins, frames)).astype(np.float32))
            onsets = np.unique(rng.randint(0, frames, 30))
            coords = np.sort(rng.randint(0, width, len(onsets)))
            per_piece_maps.append(np.stack((onsets, coords), axis=1).astype(np.int64))
        specs.append(per_piece_specs)
        o2c.append(per_piece_maps)
    out["pool/n_pieces"] = np.int64(len(images))
    for piece in range(3):
        out["pool/image%d" % piece] = images[piece].astype(np.uint8)
        for perf in range(2):
            out["pool/spec%d_%d" % (piece, perf)] = specs[piece][perf].astype(np.uint8)
            out["pool/o2c%d_%d" % (piece, perf)] = o2c[piece][perf]
    configs = {"plain": dict(system_translation=0, sheet_scaling=None, onset_translation=0, spec_padding=0,
                             interpolate=-1),
               "augmented": dict(system_translation=5, sheet_scaling=None, onset_translation=1, spec_padding=3,
                                 interpolate=2)}
    for tag, aug in configs.items():
        for shuffle in (False, True):
            name = "pool/%s_%s" % (tag, "shuffled" if shuffle else "ordered")
            np.random.seed(4711)
            pool = Pool(list(images), [list(x) for x in specs], [[m.copy() for m in x] for x in o2c],
                        spec_context=spec_ctx, sheet_context=sheet_ctx, staff_height=staff,
                        data_augmentation=dict(aug), shuffle=shuffle)
            out[name + "/entities"] = np.asarray(pool.train_entities, np.int64)
            np.random.seed(815)
            sheet_a, spec_a = pool[0:12]
            sheet_b, spec_b = pool[int(pool.shape[0]) - 1]
            out.update({name + "/sheet_a": sheet_a.astype(np.uint8), name + "/spec_a": spec_a.astype(np.uint8),
                        name + "/sheet_b": sheet_b.astype(np.uint8), name + "/spec_b": spec_b.astype(np.uint8)})

    np.savez_compressed(OUT, **out)
    print("%d arrays, %.1f KiB -> %s" % (len(out), os.path.getsize(OUT) / 1024.0, OUT))


def main_iterators():
    """utils/batch_iterators.py run on the fakes of tests/golden/iterator_fakes.py -> reference_golden_iter.npz"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import iterator_fakes as fakes
    path = os.path.join(REF, "utils", "batch_iterators.py")
    with open(path) as fp:
        source = fp.read()
    assert source.count("range((n_samples + bs - 1) / bs)") == 2
    source = source.replace("range((n_samples + bs - 1) / bs)", "range((n_samples + bs - 1) // bs)")   # py2 int division
    ns = {"__name__": "reference_batch_iterators"}
    exec(compile(_py3(source, path), path, "exec"), ns)
    out = {}
    for tag, kwargs in fakes.COMPUTE_CASES:
        X1, X2 = fakes.compute_inputs(**kwargs)
        rec1 = fakes.RecordingCompute()
        R1 = ns["batch_compute1"](X1, rec1.one, kwargs["batch_size"], prepare=fakes.prepare_one if kwargs["prepare"] else None)
        rec2 = fakes.RecordingCompute()
        R2 = ns["batch_compute2"](X1, X2, rec2.two, kwargs["batch_size"],
                                  prepare1=fakes.prepare_one if kwargs["prepare"] else None, prepare2=None)
        out.update({"bc/%s/R1" % tag: R1, "bc/%s/calls1" % tag: rec1.log(), "bc/%s/R2" % tag: R2,
                    "bc/%s/calls2" % tag: rec2.log()})
    for tag, kwargs in fakes.ITERATOR_CASES:
        np.random.seed(99)
        pool = fakes.FakePool(kwargs["n_pool"])
        it = ns["MultiviewPoolIteratorUnsupervised"](kwargs["batch_size"], prepare=fakes.prepare_two,
                                                    k_samples=kwargs["k_samples"], shuffle=kwargs["shuffle"])
        log = fakes.run_passes(it, pool, kwargs["passes"])
        out.update({"it/%s/%s" % (tag, k): v for k, v in log.items()})
    dst = os.path.join(os.path.dirname(OUT), "reference_golden_iter.npz")
    np.savez_compressed(dst, **out)
    print("%d arrays, %.1f KiB -> %s" % (len(out), os.path.getsize(dst) / 1024.0, dst))


def main_fit():
    """the reference's fit() driven by the scripted epochs of tests/golden/fit_fakes.py -> reference_golden_fit.npz"""
    import builtins
    import pickle
    import tempfile
    import time
    import types
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fit_fakes as fakes

    def open_py2(path, mode="r", *a, **k):               # Python 2: 'w' is binary-safe on POSIX
        return builtins.open(path, mode + "b" if mode in ("w", "r") else mode, *a, **k)

    lasagne = types.SimpleNamespace(layers=types.SimpleNamespace(get_all_param_values=fakes.get_all_param_values,
                                                                 set_all_param_values=fakes.set_all_param_values))
    theano = types.SimpleNamespace(shared=fakes.Shared)
    col = types.SimpleNamespace(print_colored=lambda text, colour: text)
    bcolors = types.SimpleNamespace(UNDERLINE=0, OKGREEN=1, WARNING=2)
    out = {}
    for tag, kwargs, epochs in fakes.CASES:
        kwargs = dict(kwargs)
        script = fakes.Script(epochs)
        ns = {"np": np, "os": os, "time": time, "pickle": pickle, "theano": theano, "lasagne": lasagne, "col": col,
              "BColors": bcolors, "create_iter_functions": script.create_iter_functions, "train": script.train,
              "pretrain": lambda *a, **k: None, "open": open_py2}
        fit = _function(os.path.join(REF, "utils", "train_dcca_pool.py"), "fit", ns)
        layers = fakes.Layers()
        with tempfile.TemporaryDirectory() as tmp:
            log_file, dump_file = os.path.join(tmp, "results.pkl"), os.path.join(tmp, "params.pkl")
            ret = fit(layers, None, None, None, None, update_learning_rate=fakes.schedule(kwargs.pop("decay", False)),
                      exp_name=tag, out_path=tmp, dump_file=dump_file, log_file=log_file, **kwargs)
            summary = fakes.summarize(script, layers, ret, log_file, dump_file)
        out.update({"fit/%s/%s" % (tag, k): v for k, v in summary.items()})
    dst = os.path.join(os.path.dirname(OUT), "reference_golden_fit.npz")
    np.savez_compressed(dst, **out)
    print("%d arrays, %.1f KiB -> %s" % (len(out), os.path.getsize(dst) / 1024.0, dst))


def main_sizes():
    """CCA('svd').fit at 25 000 samples and eval_retrieval at 1000 / 2000 codes -> reference_golden_sizes.npz"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import size_inputs
    out = {}
    CCA = _module(os.path.join(REF, "utils", "cca.py"))["CCA"]
    for tag in size_inputs.CCA_SIZES:
        H1, H2 = size_inputs.cca_inputs(tag)
        cca = CCA(method="svd")
        coeffs = cca.fit(H1, H2, verbose=False)
        out.update({tag + "/m1": cca.m1, tag + "/m2": cca.m2, tag + "/U": cca.U, tag + "/V": cca.V,
                    tag + "/coeffs": np.asarray(coeffs, np.float64)})
    eval_retrieval = _function(os.path.join(REF, "utils", "train_dcca_pool.py"), "eval_retrieval", {"np": np})
    for tag in size_inputs.EVAL_SIZES:
        lv1, lv2 = size_inputs.eval_inputs(tag)
        mean_rank, median_rank, mean_dist, hits, mean_ap = eval_retrieval(lv1, lv2)
        out.update({tag + "/stats": np.array([mean_rank, median_rank, mean_dist, mean_ap], np.float64),
                    tag + "/hits": np.array([hits[1], hits[5], hits[10], hits[25]], np.int64)})
    dst = os.path.join(os.path.dirname(OUT), "reference_golden_sizes.npz")
    np.savez_compressed(dst, **out)
    print("%d arrays, %.1f KiB -> %s" % (len(out), os.path.getsize(dst) / 1024.0, dst))


if __name__ == "__main__":
    if "--sizes-only" in sys.argv:
        main_sizes()
        sys.exit(0)
    if "--iterators-only" not in sys.argv and "--fit-only" not in sys.argv:
        main()
        main_sizes()
    if "--fit-only" not in sys.argv:
        main_iterators()
    if "--iterators-only" not in sys.argv:
        main_fit()
