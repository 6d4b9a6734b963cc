#!/usr/bin/env python
"""Shows that the path LEARNS: trains `mutopia_ccal_cont` from freshly drawn HeUniform weights on the synthetic paired
pool (utils/synth_data.py: 8 "notes" per pair, a dark blob in the sheet <-> an energy bump in the spectrogram) with the
reference's own recipe - batch 100, Adam, lr 0.002 (models/mutopia_ccal_cont.py:23-51), the epoch body of
utils/train_dcca_pool.py:203-205 - then re-estimates the CCA projection like refine_cca.py and reports Recall@k / MAP /
median rank on held-out pairs (utils/train_dcca_pool.py:296-299), next to the chance level.

    python tools/train_demo.py [--updates 300] [--n_test 1000] [--out params.pkl] [--model mutopia_ccal_cont]

`ASR_BENCH_PARAMS=<that pickle> python bench.py` then runs the headline bench with trained weights, so that its
recall_at_1/5 are real figures instead of the chance level of random weights."""
import argparse
import importlib
import json
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


TRAIN_FIRST_INDEX = 1 << 24           # training pairs: indices from here upwards; evaluation pairs: 0 .. n_test - 1


def train(model_name="mutopia_ccal_cont", updates=300, n_train=10000, n_refine=5000, n_test=1000, seed=23, lr=None,
          verbose=True, eval_every=100):
    """-> (params (97 arrays), history list of dicts, final metrics dict)"""
    from audio_sheet_retrieval_amd import network, refine_cca
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.train_dcca_pool import create_iter_functions, eval_retrieval
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + model_name)
    np.random.seed(seed)                                   # Lasagne draws the initial weights from NumPy's global RNG
    layers = model.build_model(show_model=False)
    # training pairs from index 2^24 upwards; the held-out pairs are 0 .. n_test-1 - batch 0 of bench.py, so that the
    # bench run with these parameters (ASR_BENCH_PARAMS) reports recall on pairs the model never saw
    data = dict(train=synth_data.SyntheticRetrievalPool(n_train, seed, shuffle=True, first_index=TRAIN_FIRST_INDEX),
                test=synth_data.SyntheticRetrievalPool(n_test, seed, shuffle=False, first_index=0))
    funcs = create_iter_functions(layers, model.objectives, model.compute_updates,
                                  model.INI_LEARNING_RATE if lr is None else lr, model.L2, model.L1)
    engine = funcs.engine
    X_te, Z_te = data["test"].get_u8(slice(0, n_test))     # uint8 sheets, like the servers pass them
    B = model.BATCH_SIZE

    def evaluate():
        lv1 = engine.embed_view1(X_te, prepared=False)
        lv2 = engine.embed_view2(Z_te)
        mean_rank, med_rank, dist, hits, mean_ap = eval_retrieval(lv1, lv2, engine=engine)
        return dict(recall_at_1=hits[1] / float(n_test), recall_at_5=hits[5] / float(n_test),
                    recall_at_10=hits[10] / float(n_test), recall_at_25=hits[25] / float(n_test), map=float(mean_ap),
                    median_rank=float(med_rank), mean_rank=float(mean_rank), mean_dist=float(dist))

    history, t0, done = [], time.time(), 0
    pool = data["train"]
    while done < updates:
        for lo in range(0, pool.shape[0] - B + 1, B):
            if done >= updates:
                break
            x, z = pool.get_u8(slice(lo, lo + B))
            loss, _corr = funcs["train"](model.prepare(x), z)
            done += 1
            if done % eval_every == 0 or done == updates:
                m = evaluate()
                m.update(update=done, loss=float(loss), seconds=time.time() - t0)
                history.append(m)
                if verbose:
                    print("update %4d  loss %.4f  R@1 %.3f  R@5 %.3f  MAP %.3f  median rank %.0f   (%.1f s)"
                          % (done, loss, m["recall_at_1"], m["recall_at_5"], m["map"], m["median_rank"], m["seconds"]),
                          flush=True)
        pool.reset_batch_generator()
    funcs.close()
    # refine_cca.py: the projection re-estimated on a larger sample than the last batch of 100
    Xr, Zr = pool.get_u8(slice(0, n_refine))
    refine_cca.estimate(layers, Xr, Zr, model.prepare, verbose=False)
    final = evaluate()
    final.update(chance_recall_at_1=1.0 / n_test, chance_recall_at_5=5.0 / n_test, chance_median_rank=n_test / 2.0,
                 updates=updates, batch=B, n_test=n_test, model=model_name, seconds=time.time() - t0)
    if verbose:
        print("after refine_cca on %d pairs: R@1 %.3f  R@5 %.3f  R@25 %.3f  MAP %.3f  median rank %.0f  (chance: R@1 %.3f, "
              "median rank %.0f)" % (n_refine, final["recall_at_1"], final["recall_at_5"], final["recall_at_25"], final["map"],
                                     final["median_rank"], 1.0 / n_test, n_test / 2.0), flush=True)
    params = network.get_all_param_values(layers)
    engine.close()
    return params, history, final


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="mutopia_ccal_cont")
    ap.add_argument("--updates", type=int, default=300)
    ap.add_argument("--n_train", type=int, default=10000)
    ap.add_argument("--n_refine", type=int, default=5000)
    ap.add_argument("--n_test", type=int, default=1000)
    ap.add_argument("--out", default=None, help="parameter pickle (the reference's 97-array format)")
    ap.add_argument("--json", default=None, help="write history + final metrics here")
    args = ap.parse_args()
    params, history, final = train(args.model, args.updates, args.n_train, args.n_refine, args.n_test)
    if args.out and args.out.endswith(".npz"):      # tests/golden/trained_cont_params.npz: what bench.py's recall leg loads
        # the index range the run trained on travels with the weights: bench.py refuses them unless its own pairs lie
        # below it ("held out" must be checkable from the file, not from this script's source)
        np.savez_compressed(args.out, train_first_index=np.int64(TRAIN_FIRST_INDEX), train_count=np.int64(args.n_train),
                            updates=np.int64(args.updates), **{"p%02d" % i: a for i, a in enumerate(params)})
    elif args.out:
        with open(args.out, "wb") as fp:
            pickle.dump(params, fp, protocol=2)
    rec = {"what": "train_demo", "final": final, "history": history}
    if args.json:
        with open(args.json, "w") as fp:
            json.dump(rec, fp)
    print(json.dumps(rec["final"]))


if __name__ == "__main__":
    main()
