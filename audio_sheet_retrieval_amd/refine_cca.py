#!/usr/bin/env python
"""Re-estimate the CCA projection of a trained model on a large training sample (25 000 pairs in the paper's
recipe, README.md:104-111) and write the refined parameter file next to the original one
(`<EXP_ROOT>/<model>_est_UV/params_<tag>.pkl`).  Command line of the reference's refine_cca.py (:24-35):

    python -m audio_sheet_retrieval_amd.refine_cca --n_train 25000 --model models/mutopia_ccal_cont.py \
        --data synthetic --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml

What it computes (reference :78-107): the deterministic tower outputs feeding the CCALayer for the first n_train
pairs, CCA('svd').fit on them, then U, V and the two means of the layer are overwritten.  Here the towers, the
covariance sums and the 32x32 float64 algebra all run on the GPU (asr_embed_view*, asr_cca_fit).
"""
import argparse
import os
import pickle

import numpy as np

from . import network
from .config.settings import EXP_ROOT
from .retrieval_wrapper import load_params
from .run_train import compile_tag, select_data, select_model
from .utils.cca import CCA


def _arguments(argv):
    p = argparse.ArgumentParser(description="Re-estimate U, V and the means of a trained model's CCALayer.")
    p.add_argument("--model", default="flickr30", help="model definition, e.g. models/mutopia_ccal_cont.py")
    p.add_argument("--data", type=str, default="flickr30", help="data set ('synthetic[:train:valid:test]')")
    p.add_argument("--n_train", type=int, default=1000, help="training pairs the projection is estimated on")
    p.add_argument("--seed", type=int, default=23)
    p.add_argument("--train_split", type=str, default=None, help="split file (only its name enters the tag)")
    p.add_argument("--config", type=str, default=None, help="experiment config (only its name enters the tag)")
    p.add_argument("--batch_size", type=int, default=10,
                   help="forward chunk; results do not depend on it (the reference uses 10, :96-97)")
    return p.parse_args(argv)


def _cca_layer_of(latent_layer):
    for layer in network.get_all_layers(latent_layer):
        if isinstance(layer, network.CCALayer):
            return layer
    raise ValueError("the model has no CCALayer to refine")


def _tower_features(fn, data, chunk, prepare=None):
    """batch_compute1(data, fn, chunk, prepare=prepare) of the reference (:96-97).  The tower outputs do not depend on
    the chunking (deterministic mode), so with the model's own `prepare` - or none - the whole array is handed to the
    library in one call (2 x 2500 calls of 10 samples in the reference's recipe); another callable is applied per
    chunk on the host."""
    if prepare is None or network.is_fused_prepare(fn.net, prepare):
        return fn.embed_raw(data) if (prepare is not None or fn.view == 2) else fn(data)
    parts = []
    for lo in range(0, data.shape[0], chunk):
        parts.append(fn(prepare(data[lo:lo + chunk])))
    return np.concatenate(parts, axis=0)


def estimate(layers, sheets, specs, prepare=None, batch_size=10, verbose=True):
    """refine_cca.py:78-107 on arrays: tower outputs feeding the CCALayer -> CCA('svd').fit -> the layer's mean1,
    mean2, U, V overwritten (float32).  Returns the fitted CCA object."""
    view1, view2, latent1, _latent2 = layers
    cca_layer = _cca_layer_of(latent1)
    feed1, feed2 = cca_layer.input_layers
    tower1 = network.function([view1.input_var], network.get_output(feed1, deterministic=True))
    tower2 = network.function([view2.input_var], network.get_output(feed2, deterministic=True))
    chunk = max(1, min(batch_size, sheets.shape[0]))
    h1 = _tower_features(tower1, sheets, chunk, prepare)
    h2 = _tower_features(tower2, specs, chunk)
    cca = CCA(method="svd", engine=tower1.engine)
    cca.fit(h1, h2, verbose=verbose)
    for shared, value in ((cca_layer.mean1, cca.m1), (cca_layer.mean2, cca.m2), (cca_layer.U, cca.U), (cca_layer.V, cca.V)):
        shared.set_value(np.asarray(value, dtype=np.float32))
    return cca


def main(argv=None):
    args = _arguments(argv)
    model, _ = select_model(args.model)
    prepare = getattr(model, "prepare", None)
    layers = model.build_model(show_model=False)
    tag = compile_tag(args.train_split, args.config)
    file_name = "params.pkl" if tag is None else "params_%s.pkl" % tag
    source = os.path.join(EXP_ROOT, model.EXP_NAME, file_name)
    target_dir = os.path.join(EXP_ROOT, model.EXP_NAME + "_est_UV")
    print("model %s, tag %s\nparameters: %s" % (model.EXP_NAME, tag, source))
    network.set_all_param_values(layers, load_params(source))

    data = select_data(args.data, args.train_split, args.config, args.seed)
    sheets, specs = data["train"][0:args.n_train]
    print("tower outputs of %d training pairs, fitting CCA ('svd') ..." % sheets.shape[0])
    estimate(layers, sheets, specs, prepare, batch_size=min(args.batch_size, args.n_train))

    os.makedirs(target_dir, exist_ok=True)
    target = os.path.join(target_dir, file_name)
    with open(target, "wb") as fp:
        pickle.dump(network.get_all_param_values(layers), fp, protocol=-1)
    print("refined parameters: %s" % target)
    return target


if __name__ == "__main__":
    main()
