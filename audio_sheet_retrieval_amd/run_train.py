#!/usr/bin/env python
"""Train an audio <-> sheet retrieval model.  Command line of the reference's run_train.py (flags :55-64):

    python -m audio_sheet_retrieval_amd.run_train --model models/mutopia_ccal_cont.py --data synthetic \
        --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml

The helpers `select_model` (:19-29), `select_data` (:32-41) and `compile_tag` (:44-48) are shared with run_eval and
refine_cca, as in the reference.  The training loop itself is `utils.train_dcca_pool.fit`; every update runs as HIP
kernels (forward, CCALayer, pairwise ranking loss, backward, Adam).

Several GPUs: `--gpus N` starts one process per GPU (or launch the ranks yourself, e.g. `python -m
torch.distributed.run --nproc-per-node N -m audio_sheet_retrieval_amd.run_train ...`: RANK / LOCAL_RANK / WORLD_SIZE
are read from the environment).  All ranks draw the same batches (same seed) and each trains on its contiguous share of
the rows (distributed.shard_range: BATCH_SIZE = 100 on 8 GPUs is 4 x 13 + 4 x 12 rows - no row is dropped); the library
all-reduces the BatchNorm sums and the gradients over its own RCCL communicator, so parameters stay identical
everywhere, and every rank follows rank 0's early-stopping decisions.  Only rank 0 writes files.  No PyTorch is
involved: the communicator id travels over a local TCP hub (distributed.HubComm).
"""
import argparse
import importlib
import os

from .config.settings import EXP_ROOT

_SYNTHETIC_SIZES = (("n_train", 10000), ("n_valid", 1000), ("n_test", 2000))


def select_model(model_path):
    """'models/<name>.py' -> (model module with EXP_NAME = <name>, fit function)"""
    from .utils.train_dcca_pool import fit
    name = os.path.basename(model_path)
    if name.endswith(".py"):
        name = name[:-3]
    model = importlib.import_module("%s.models.%s" % (__package__, name))
    model.EXP_NAME = name
    return model, fit


def select_data(data_name, split_file, config_file, seed=23, test_only=False):
    """Data pools {'train','valid','test'}.

    'synthetic[:<n_train>[:<n_valid>[:<n_test>]]]': MSMD-shaped synthetic pools (utils/synth_data.py).
    'mutopia': the reference loads the MSMD data set through the `msmd` package (utils/mutopia_data.py); neither is
    part of this repository, so this exits with a message.  Anything else gives None, like the reference."""
    name = str(data_name)
    if name == "mutopia":
        raise SystemExit("--data mutopia needs the `msmd` package and the MSMD data set, which are outside the "
                         "accelerated path (SURVEY.md section 2 row 16); use --data synthetic")
    if not name.startswith("synthetic"):
        return None
    from .utils import synth_data
    sizes = dict(_SYNTHETIC_SIZES)
    for (key, _), value in zip(_SYNTHETIC_SIZES, name.split(":")[1:]):
        sizes[key] = int(value)
    return synth_data.load_synthetic_retrieval(seed=seed, **sizes)


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


def compile_tag(train_split, config):
    """experiment tag '<split file stem>_<config file stem>'"""
    return "%s_%s" % (_stem(train_split), _stem(config))


def _arguments(argv):
    p = argparse.ArgumentParser(description="Train a cross-modality retrieval model.")
    p.add_argument("--model", help="model definition, e.g. models/mutopia_ccal_cont.py")
    p.add_argument("--data", help="data set ('synthetic[:train:valid:test]')")
    p.add_argument("--resume", action="store_true", help="start from the parameter file of an earlier run")
    p.add_argument("--seed", type=int, default=23)
    p.add_argument("--no_dump", action="store_true", help="do not write the parameter file")
    p.add_argument("--show_architecture", action="store_true", help="print the layer table")
    p.add_argument("--train_split", type=str, default=None)
    p.add_argument("--config", type=str, default=None)
    p.add_argument("--max_epochs", type=int, default=None, help="override the model's MAX_EPOCHS")
    p.add_argument("--gpus", type=int, default=1, help="data-parallel training over this many GPUs of the node")
    p.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                   help="exchange between the ranks: the library's RCCL communicator, or host callbacks over the TCP "
                        "hub (several ranks on ONE GPU, which RCCL refuses; tests)")
    return p.parse_args(argv)


def _join_data_parallel(layers, seed, transport="rccl"):
    """One process per GPU.  Returns the control-plane hub (rank / world on it).  On the RCCL
    transport the hub stays open for the whole of fit() as a dead-peer watchdog (launch.join): a rank that dies in
    epoch 3 ends the others within seconds instead of leaving them in an all-reduce that has no timeout."""
    import numpy as np
    from . import launch
    np.random.seed(seed)                             # same batch order on every rank
    return launch.join(layers[0].net.engine, transport=transport)


def _spawn_ranks(argv, n):
    """`--gpus N` from a plain shell: N children of this (GPU-free) process, one per device, polled - when one exits
    non-zero the others are terminated and its code is returned (launch.spawn_ranks)."""
    import sys
    from . import launch
    return launch.spawn_ranks([sys.executable, "-m", __package__ + ".run_train"] + list(argv), n)


def main(argv=None):
    args = _arguments(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        raise SystemExit(_spawn_ranks(sys.argv[1:] if argv is None else argv, args.gpus))
    from . import launch
    rank, local_rank, world = launch.world_from_env()
    if world > 1:
        os.environ["ASR_DEVICE"] = str(launch.device_for(local_rank))
    model, fit = select_model(args.model)
    if world > 1 and model.BATCH_SIZE < world:
        raise SystemExit("BATCH_SIZE %d cannot be sharded over %d GPUs" % (model.BATCH_SIZE, world))
    data = select_data(args.data, args.train_split, args.config, args.seed)
    tag = compile_tag(args.train_split, args.config)
    suffix = "" if tag is None else "_" + tag
    out_path = os.path.join(EXP_ROOT, model.EXP_NAME)
    dump_file = os.path.join(out_path, "params%s.pkl" % suffix)
    log_file = os.path.join(out_path, "results%s.pkl" % suffix)
    print("model %s, tag %s, output folder %s" % (model.EXP_NAME, tag, out_path))

    layers = model.build_model(show_model=args.show_architecture)
    if args.resume:
        from . import network
        from .retrieval_wrapper import load_params
        print("resuming from %s" % dump_file)
        network.set_all_param_values(layers, load_params(dump_file))
    if args.no_dump:
        dump_file = None
    hub = None
    if world > 1:
        hub = _join_data_parallel(layers, args.seed, args.comm)
        if hub.rank != 0:
            dump_file, log_file = None, os.devnull

    schedule = dict(num_epochs=model.MAX_EPOCHS if args.max_epochs is None else args.max_epochs,
                    patience=model.PATIENCE,
                    learn_rate=model.INI_LEARNING_RATE,
                    update_learning_rate=model.update_learning_rate,
                    refinement_steps=model.REFINEMENT_STEPS,
                    lr_multiplier=model.LR_MULTIPLIER,
                    refinement_patience=getattr(model, "REFINEMENT_PATIENCE", 10),
                    pretrain_epochs=getattr(model, "PRETRAIN_EPOCHS", 0),
                    fit_cca=getattr(model, "FIT_CCA", True))
    _, best_validation = fit(layers, data, model.objectives,
                             train_batch_iter=model.train_batch_iterator(model.BATCH_SIZE),
                             valid_batch_iter=model.valid_batch_iterator(),
                             compute_updates=model.compute_updates, l_2=model.L2, l_1=model.L1,
                             exp_name=model.EXP_NAME, out_path=out_path, dump_file=dump_file, log_file=log_file,
                             **schedule)
    launch.leave(hub)              # (an exception above leaves without the goodbye: the peers' watchdogs end them too)
    return best_validation


if __name__ == "__main__":
    main()
