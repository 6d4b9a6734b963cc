#!/usr/bin/env python
"""Evaluate a cross-modality retrieval model - same CLI and control flow as the
reference's audio_sheet_retrieval/run_eval.py (:34-212).

    python -m audio_sheet_retrieval_amd.run_eval --model models/mutopia_ccal_cont.py --data synthetic \
        --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml \
        --estimate_UV --n_test 2000 [--V2_to_V1] [--dump_results]
"""
from __future__ import print_function

import argparse
import os

import numpy as np
import yaml

from . import network
from .config.settings import EXP_ROOT
from .retrieval_wrapper import load_params
from .run_train import compile_tag, select_data, select_model
from .utils.batch_iterators import batch_compute2
from .utils.train_dcca_pool import eval_retrieval


def flip_variables(v1, v2):
    """ flip variables (:26-31) """
    tmp = v1.copy()
    v1 = v2
    v2 = tmp
    return v1, v2


def main(argv=None):
    parser = argparse.ArgumentParser(description='Evaluate cross-modality retrieval model.')
    parser.add_argument('--model', help='select model to evaluate.')
    parser.add_argument('--data', help='select evaluation data.', type=str)
    parser.add_argument('--show', help='show evaluation plots.', action='store_true')
    parser.add_argument('--n_test', help='number of test samples used.', type=int, default=None)
    parser.add_argument('--V2_to_V1', help='query direction.', action='store_true')
    parser.add_argument('--estimate_UV', help='load re-estimated U and V.', action='store_true')
    parser.add_argument('--max_dim', help='maximum dimension of retrieval space.', type=int, default=None)
    parser.add_argument('--seed', help='query direction.', type=int, default=23)
    parser.add_argument('--train_split', help='path to train split file.', type=str, default=None)
    parser.add_argument('--config', help='path to experiment config file.', type=str, default=None)
    parser.add_argument('--dump_results', help='dump results of current run to file.', action='store_true')
    args = parser.parse_args(argv)

    model, _ = select_model(args.model)
    if not hasattr(model, 'prepare'):
        model.prepare = None

    print("Building network %s ..." % model.EXP_NAME)
    layers = model.build_model(show_model=False)

    tag = compile_tag(args.train_split, args.config)
    print("Experimental Tag:", tag)

    exp_name = model.EXP_NAME
    if args.estimate_UV:
        exp_name += "_est_UV"
    out_path = os.path.join(os.path.join(EXP_ROOT), exp_name)
    dump_file = 'params.pkl' if tag is None else 'params_%s.pkl' % tag
    dump_file = os.path.join(out_path, dump_file)

    print("\n")
    print("Loading model parameters from:", dump_file)
    params = load_params(dump_file)
    if isinstance(params[0], list):
        # old redundant dump (:76-79): one full list per layer handle
        params = params[-1]
    network.set_all_param_values(layers, params)

    print("\nLoading data...")
    data = select_data(args.data, args.train_split, args.config, args.seed, test_only=True)

    print("\nCompiling prediction functions...")
    l_view1, l_view2, l_v1latent, l_v2latent = layers
    input_1 = input_2 = [l_view1.input_var, l_view2.input_var]
    compute_v1_latent = network.function(inputs=input_1,
                                         outputs=network.get_output(l_v1latent, deterministic=True))
    compute_v2_latent = network.function(inputs=input_2,
                                         outputs=network.get_output(l_v2latent, deterministic=True))

    print("Evaluating on test set...")
    eval_set = 'test'
    n_test = args.n_test if args.n_test is not None else data[eval_set].shape[0]
    indices = np.linspace(0, data[eval_set].shape[0] - 1, n_test).astype(int)
    X1, X2 = data[eval_set][indices]

    print("Computing embedding space...")
    lv1 = batch_compute2(X1, X2, compute_v1_latent, np.min([100, n_test]), prepare1=model.prepare)
    lv2 = batch_compute2(X1, X2, compute_v2_latent, np.min([100, n_test]), prepare1=model.prepare)
    lv1_cca = lv1
    lv2_cca = lv2

    if args.V2_to_V1:
        lv1_cca, lv2_cca = flip_variables(lv1_cca, lv2_cca)

    n_test = lv1_cca.shape[0]

    if args.show:
        print("--show: plotting is outside the accelerated hot path; skipped")

    # clip some dimensions (:160-162)
    max_dim = args.max_dim if args.max_dim is not None else lv1_cca.shape[1]
    lv1_cca = lv1_cca[:, 0:max_dim]
    lv2_cca = lv2_cca[:, 0:max_dim]

    print("V1.shape:", lv1_cca.shape)
    print("V2.shape:", lv2_cca.shape)

    print("Computing performance measures...")
    engine = compute_v1_latent.engine
    mean_rank_te, med_rank_te, dist_te, hit_rates, map_ = eval_retrieval(lv1_cca, lv2_cca, engine=engine)

    recall_at_k = dict()
    print("\nHit Rates:")
    for key in np.sort(list(hit_rates.keys())):
        recall_at_k[key] = float(100 * hit_rates[key]) / n_test
        pk = recall_at_k[key] / key
        print("Top %02d: %.3f (%d) %.3f" % (key, recall_at_k[key], hit_rates[key], pk))

    print("\n")
    print("Median Rank: %.2f (%d)" % (med_rank_te, lv2_cca.shape[0]))
    print("Mean Rank  : %.2f (%d)" % (mean_rank_te, lv2_cca.shape[0]))
    print("Mean Dist  : %.5f " % dist_te)
    print("MAP        : %.3f " % map_)

    _, dists, _ = engine.rank(lv1_cca, lv2_cca)          # diag of the distance matrix (:190)
    print("Min Dist   : %.5f " % np.min(dists))
    print("Max Dist   : %.5f " % np.max(dists))
    print("Med Dist   : %.5f " % np.median(dists))

    results = {"map": float(map_), 'med_rank': float(med_rank_te),
               'recall_at_k': dict(("%d" % k, float(v)) for k, v in recall_at_k.items())}
    if args.dump_results:
        ret_dir = "A2S" if args.V2_to_V1 else "S2A"
        res_file = dump_file.replace("params_", "eval_").replace(".pkl", "_%s.yaml")
        res_file = res_file % ret_dir
        with open(res_file, 'w') as fp:
            yaml.dump(results, fp, default_flow_style=False)
    return results


if __name__ == '__main__':
    main()
