#!/usr/bin/env python
"""Train a cross-modality retrieval model - same CLI as the reference's
audio_sheet_retrieval/run_train.py (flags :55-64, select_model :19-29,
select_data :32-41, compile_tag :44-48, main :52-118).

    python -m audio_sheet_retrieval_amd.run_train --model models/mutopia_ccal_cont.py \
        --data synthetic --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml
"""
from __future__ import print_function

import argparse
import importlib
import os
import pickle

from .config.settings import EXP_ROOT


def select_model(model_path):
    """ select model and train function (:19-29) """
    model_str = os.path.basename(model_path)
    model_str = model_str.split('.py')[0]
    model = importlib.import_module("audio_sheet_retrieval_amd.models." + model_str)
    from .utils.train_dcca_pool import fit
    model.EXP_NAME = model_str
    return model, fit


def select_data(data_name, split_file, config_file, seed=23, test_only=False):
    """ select train data (:32-41).  'mutopia' needs the `msmd` package and data
    set, which this repository does not ship; 'synthetic' gives MSMD-shaped
    synthetic pools (utils/synth_data.py)."""
    if str(data_name) == "mutopia":
        try:
            import msmd  # noqa: F401
        except ImportError:
            raise SystemExit("--data mutopia needs the `msmd` package and the MSMD data set "
                             "(audio_sheet_retrieval/utils/mutopia_data.py); use --data synthetic")
        raise SystemExit("MSMD loading is outside the accelerated hot path (SURVEY.md section 2 row 16)")
    if str(data_name).startswith("synthetic"):
        from .utils import synth_data
        sizes = dict(n_train=10000, n_valid=1000, n_test=2000)
        if ":" in str(data_name):      # synthetic:<n_train>:<n_valid>:<n_test>
            vals = [int(v) for v in str(data_name).split(":")[1:]]
            sizes = dict(zip(("n_train", "n_valid", "n_test"), vals + list(sizes.values())[len(vals):]))
        return synth_data.load_synthetic_retrieval(seed=seed, **sizes)
    return None


def compile_tag(train_split, config):
    """ compile model tag fom split and config file paths (:44-48) """
    tag = os.path.splitext(os.path.basename(train_split))[0]
    tag += "_" + os.path.splitext(os.path.basename(config))[0]
    return tag


def main(argv=None):
    parser = argparse.ArgumentParser(description='Train cross-modality retrieval model.')
    parser.add_argument('--model', help='select model to train.')
    parser.add_argument('--data', help='select data for training.')
    parser.add_argument('--resume', help='resume on pre-trained model.', action='store_true')
    parser.add_argument('--seed', help='query direction.', type=int, default=23)
    parser.add_argument('--no_dump', help='do not dump model file.', action='store_true')
    parser.add_argument('--show_architecture', help='print model architecture.', action='store_true')
    parser.add_argument('--train_split', help='path to train split file.', type=str, default=None)
    parser.add_argument('--config', help='path to experiment config file.', type=str, default=None)
    parser.add_argument('--max_epochs', help='(extension) cap on MAX_EPOCHS.', type=int, default=None)
    args = parser.parse_args(argv)

    model, fit = select_model(args.model)
    fit_cca = model.FIT_CCA if hasattr(model, 'FIT_CCA') else True
    refinement_patience = getattr(model, 'REFINEMENT_PATIENCE', 10)
    pretrain_epochs = getattr(model, 'PRETRAIN_EPOCHS', 0)

    print("\nLoading data...")
    data = select_data(args.data, args.train_split, args.config, args.seed)

    tag = compile_tag(args.train_split, args.config)
    print("Experimental Tag:", tag)

    out_path = os.path.join(os.path.join(EXP_ROOT), model.EXP_NAME)
    dump_file = 'params.pkl' if tag is None else 'params_%s.pkl' % tag
    dump_file = os.path.join(out_path, dump_file)
    log_file = 'results.pkl' if tag is None else 'results_%s.pkl' % tag
    log_file = os.path.join(out_path, log_file)

    print("\nBuilding network...")
    layers = model.build_model(show_model=args.show_architecture)

    from . import network
    if args.resume:
        print("\n")
        print("Loading model parameters from:", dump_file)
        from .retrieval_wrapper import load_params
        network.set_all_param_values(layers, load_params(dump_file))

    dump_file = None if args.no_dump else dump_file

    # data parallel (extension): one process per GPU under torch.distributed.run.  Every rank draws
    # the same batches (same seed) and trains on its rows; the library all-reduces BatchNorm sums and gradients
    # over its own RCCL communicator, so all ranks hold the same parameters.  Rank 0 writes the files.
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import numpy as np
        import torch.distributed as dist
        from . import distributed as D
        rank = int(os.environ["RANK"])
        os.environ.setdefault("ASR_DEVICE", os.environ.get("LOCAL_RANK", "0"))
        dist.init_process_group(backend="gloo")          # control plane only: the id broadcast
        np.random.seed(args.seed)
        D.init_data_parallel(layers[0].net.engine, rank, world, transport="rccl")
        if rank != 0:
            dump_file, log_file = None, os.devnull

    train_batch_iter = model.train_batch_iterator(model.BATCH_SIZE)
    valid_batch_iter = model.valid_batch_iterator()
    layers, va_loss = fit(layers, data, model.objectives,
                          train_batch_iter=train_batch_iter, valid_batch_iter=valid_batch_iter,
                          num_epochs=model.MAX_EPOCHS if args.max_epochs is None else args.max_epochs,
                          patience=model.PATIENCE,
                          learn_rate=model.INI_LEARNING_RATE, update_learning_rate=model.update_learning_rate,
                          compute_updates=model.compute_updates, l_2=model.L2, l_1=model.L1,
                          exp_name=model.EXP_NAME, out_path=out_path, dump_file=dump_file, log_file=log_file,
                          fit_cca=fit_cca, pretrain_epochs=pretrain_epochs,
                          refinement_steps=model.REFINEMENT_STEPS, lr_multiplier=model.LR_MULTIPLIER,
                          refinement_patience=refinement_patience)
    return va_loss


if __name__ == '__main__':
    main()
