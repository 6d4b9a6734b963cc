#!/usr/bin/env python
"""Re-estimate the CCA projection on a large sample - same CLI and control flow
as the reference's audio_sheet_retrieval/refine_cca.py (:24-111).

    python -m audio_sheet_retrieval_amd.refine_cca --n_train 25000 --model models/mutopia_ccal_cont.py \
        --data synthetic --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml
"""
from __future__ import print_function

import argparse
import os
import pickle

import numpy as np

from . import network
from .config.settings import EXP_ROOT
from .retrieval_wrapper import load_params
from .run_train import compile_tag, select_data, select_model
from .utils.batch_iterators import batch_compute1
from .utils.cca import CCA


def main(argv=None):
    parser = argparse.ArgumentParser(description='Train model.')
    parser.add_argument('--model', help='model parameters for evaluation.', default="flickr30")
    parser.add_argument('--data', help='select evaluation data.', type=str, default="flickr30")
    parser.add_argument('--n_train', help='number of train samples used for projection.', type=int, default=1000)
    parser.add_argument('--seed', help='query direction.', type=int, default=23)
    parser.add_argument('--train_split', help='path to train split file.', type=str, default=None)
    parser.add_argument('--config', help='path to experiment config file.', type=str, default=None)
    parser.add_argument('--batch_size', type=int, default=10,
                        help='(extension) forward batch; the reference hard-codes 10 (:96-97)')
    args = parser.parse_args(argv)

    model, _ = select_model(args.model)
    if not hasattr(model, 'prepare'):
        model.prepare = None

    print("Building network %s ..." % model.EXP_NAME)
    layers = model.build_model(show_model=False)

    tag = compile_tag(args.train_split, args.config)
    print("Experimental Tag:", tag)

    out_path = os.path.join(os.path.join(EXP_ROOT), model.EXP_NAME)
    dump_file_name = 'params.pkl' if tag is None else 'params_%s.pkl' % tag
    dump_file = os.path.join(out_path, dump_file_name)
    print("\n")
    print("Loading model parameters from:", dump_file)
    network.set_all_param_values(layers, load_params(dump_file))

    # reset model parameter file (:61-65)
    out_path = os.path.join(os.path.join(EXP_ROOT), model.EXP_NAME + "_est_UV")
    if not os.path.exists(out_path):
        os.makedirs(out_path)
    dump_file = os.path.join(out_path, dump_file_name)

    print("\nLoading data...")
    data = select_data(args.data, args.train_split, args.config, args.seed)

    print("\nCompiling prediction functions...")
    l_view1, l_view2, l_v1latent, l_v2latent = layers
    input_1, input_2 = [l_view1.input_var], [l_view2.input_var]

    # get cca layer input (:78-84)
    cca_layer = None
    for l in network.get_all_layers(l_v1latent):
        if isinstance(l, network.CCALayer):
            print("CCALayer found!")
            cca_layer = l
            l_v1_cca = cca_layer.input_layers[0]
            l_v2_cca = cca_layer.input_layers[1]
            break

    compute_v1_latent = network.function(inputs=input_1,
                                         outputs=network.get_output(l_v1_cca, deterministic=True))
    compute_v2_latent = network.function(inputs=input_2,
                                         outputs=network.get_output(l_v2_cca, deterministic=True))

    print("Computing train output...")
    X1, X2 = data['train'][0:args.n_train]
    bs = int(np.min([args.batch_size, args.n_train]))
    lv1_tr = batch_compute1(X1, compute_v1_latent, bs, prepare=model.prepare)
    lv2_tr = batch_compute1(X2, compute_v2_latent, bs)

    print("Fitting CCA model...")
    cca = CCA(method='svd', engine=compute_v1_latent.engine)
    cca.fit(lv1_tr, lv2_tr, verbose=True)

    # reset layer weights (:104-107)
    cca_layer.mean1.set_value(cca.m1.astype(np.float32))
    cca_layer.mean2.set_value(cca.m2.astype(np.float32))
    cca_layer.U.set_value(cca.U.astype(np.float32))
    cca_layer.V.set_value(cca.V.astype(np.float32))

    print("Dumping refined model...")
    with open(dump_file, 'wb') as fp:
        pickle.dump(network.get_all_param_values(layers), fp, protocol=-1)
    return dump_file


if __name__ == '__main__':
    main()
