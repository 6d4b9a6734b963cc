#!/usr/bin/env python
"""Re-estimate the CCA projection of a trained model on a large training sample (25 000 pairs in the paper's
recipe, README.md:104-111) and write the refined parameter file next to the original one
(`<EXP_ROOT>/<model>_est_UV/params_<tag>.pkl`).  Command line of the reference's refine_cca.py (:24-35):

    python -m audio_sheet_retrieval_amd.refine_cca --n_train 25000 --model models/mutopia_ccal_cont.py \
        --data synthetic --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml

What it computes (reference :78-107): the deterministic tower outputs feeding the CCALayer for the first n_train
pairs, CCA('svd').fit on them, then U, V and the two means of the layer are overwritten.  Here the towers, the
covariance sums and the 32x32 float64 algebra all run on the GPU (asr_embed_view*, asr_cca_fit).

Several GPUs: `--gpus N` (or ranks started by a launcher) shards the n_train pairs by contiguous ranges; every rank
runs the towers on its pairs, the 32-d tower outputs (128 bytes per sample and view - 6.4 MB at 25 000 pairs) are
all-gathered over the library's communicator and every rank fits the same CCA on the same array: U, V and the means
are bit-identical to the one-GPU run (an all-reduce of partial covariance sums would change the float64 summation
order; the all-gather costs nothing next to the towers).  Rank 0 writes the file.
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
    p.add_argument("--gpus", type=int, default=1, help="shard the training pairs over this many GPUs of the node")
    p.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                   help="exchange between the ranks: RCCL, or host callbacks over the TCP hub (ranks sharing one GPU)")
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


def estimate(layers, sheets, specs, prepare=None, batch_size=10, verbose=True, comm=None):
    """refine_cca.py:78-107 on arrays: tower outputs feeding the CCALayer -> CCA('svd').fit -> the layer's mean1,
    mean2, U, V overwritten (float32).  Returns the fitted CCA object.  comm (distributed.EngineComm): sheets / specs
    are this rank's shard; the tower outputs of all ranks are gathered in rank order before the fit."""
    view1, view2, latent1, _latent2 = layers
    cca_layer = _cca_layer_of(latent1)
    feed1, feed2 = cca_layer.input_layers
    tower1 = network.function([view1.input_var], network.get_output(feed1, deterministic=True))
    tower2 = network.function([view2.input_var], network.get_output(feed2, deterministic=True))
    chunk = max(1, min(batch_size, sheets.shape[0]))
    h1 = _tower_features(tower1, sheets, chunk, prepare)
    h2 = _tower_features(tower2, specs, chunk)
    if comm is not None and comm.world > 1:
        h1, h2 = comm.all_gather_rows(h1), comm.all_gather_rows(h2)
    cca = CCA(method="svd", engine=tower1.engine)
    cca.fit(h1, h2, verbose=verbose)
    for shared, value in ((cca_layer.mean1, cca.m1), (cca_layer.mean2, cca.m2), (cca_layer.U, cca.U), (cca_layer.V, cca.V)):
        shared.set_value(np.asarray(value, dtype=np.float32))
    return cca


def main(argv=None):
    args = _arguments(argv)
    from . import launch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        raise SystemExit(launch.spawn_ranks([sys.executable, "-m", __package__ + ".refine_cca"] +
                                            list(sys.argv[1:] if argv is None else argv), args.gpus))
    rank, local_rank, world = launch.world_from_env()
    if world > 1:
        os.environ["ASR_DEVICE"] = str(launch.device_for(local_rank))
    say = print if rank == 0 else (lambda *a, **k: None)
    model, _ = select_model(args.model)
    prepare = getattr(model, "prepare", None)
    layers = model.build_model(show_model=False)
    tag = compile_tag(args.train_split, args.config)
    file_name = "params.pkl" if tag is None else "params_%s.pkl" % tag
    source = os.path.join(EXP_ROOT, model.EXP_NAME, file_name)
    target_dir = os.path.join(EXP_ROOT, model.EXP_NAME + "_est_UV")
    say("model %s, tag %s\nparameters: %s" % (model.EXP_NAME, tag, source))
    network.set_all_param_values(layers, load_params(source))
    hub, comm = None, None
    if world > 1:
        from . import distributed
        engine = layers[0].net.engine
        hub = launch.join(engine, transport=args.comm)
        comm = distributed.EngineComm(engine)

    data = select_data(args.data, args.train_split, args.config, args.seed)
    n_train = min(args.n_train, data["train"].shape[0])
    lo, hi = 0, n_train
    if world > 1:
        lo, hi = distributed.shard_range(n_train, rank, world)
        if hi <= lo:
            raise SystemExit("--n_train %d is smaller than the number of GPUs (%d)" % (n_train, world))
    sheets, specs = data["train"][lo:hi]
    say("tower outputs of %d training pairs, fitting CCA ('svd') ..." % n_train)
    estimate(layers, sheets, specs, prepare, batch_size=min(args.batch_size, args.n_train), verbose=rank == 0, comm=comm)

    target = os.path.join(target_dir, file_name)
    if rank == 0:
        os.makedirs(target_dir, exist_ok=True)
        from .utils.train_dcca_pool import atomic_pickle_dump
        atomic_pickle_dump(network.get_all_param_values(layers), target, protocol=-1)
        print("refined parameters: %s" % target)
    launch.leave(hub)
    return target


if __name__ == "__main__":
    main()
