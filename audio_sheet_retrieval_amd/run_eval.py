#!/usr/bin/env python
"""Two-way snippet retrieval evaluation of a trained model: embed a test subset with both towers, rank all pairs by
cosine distance, report Recall@k / median and mean rank / MAP.  Command line of the reference's run_eval.py (:34-50):

    python -m audio_sheet_retrieval_amd.run_eval --model models/mutopia_ccal_cont.py --data synthetic \
        --train_split splits/all_split.yaml --config exp_configs/mutopia_full_aug.yaml \
        --estimate_UV --n_test 2000 [--V2_to_V1] [--max_dim D] [--dump_results]

Pipeline (reference :59-194): parameter pickle -> network; test subset = np.linspace(0, N-1, n_test) (:103);
deterministic embeddings of both views; optional query-direction swap and dimension clipping; eval_retrieval.
Embedding and ranking run on the GPU; the ranks are exact (float64 distances, stable tie order).

Several GPUs: `--gpus N` starts one process per GPU (or launch the ranks yourself: RANK / LOCAL_RANK / WORLD_SIZE are
read from the environment).  The n_test pairs are sharded by contiguous ranges (distributed.shard_range), every rank
embeds its own pairs, the candidate embeddings are all-gathered over the library's RCCL communicator and every rank
ranks its queries against all of them (`query_offset`); hit counters are all-reduced, the integer ranks and match
distances gathered.  Rank 0 prints and dumps exactly what one GPU prints - the ranks are integers and the distances
exact float64, so nothing depends on N.
"""
import argparse
import os

import numpy as np
import yaml

from . import network
from .config.settings import EXP_ROOT
from .retrieval_wrapper import load_params
from .run_train import compile_tag, select_data, select_model
from .utils.train_dcca_pool import eval_retrieval


def flip_variables(v1, v2):
    """swap the roles of the two views (audio -> sheet instead of sheet -> audio)"""
    return v2, v1.copy()


def _arguments(argv):
    p = argparse.ArgumentParser(description="Evaluate a cross-modality retrieval model.")
    p.add_argument("--model", help="model definition, e.g. models/mutopia_ccal_cont.py")
    p.add_argument("--data", type=str, help="data set ('synthetic[:train:valid:test]')")
    p.add_argument("--show", action="store_true", help="(plots are not part of this implementation)")
    p.add_argument("--n_test", type=int, default=None, help="size of the evaluated test subset")
    p.add_argument("--V2_to_V1", action="store_true", help="query with view 2 (audio), retrieve view 1 (sheet)")
    p.add_argument("--estimate_UV", action="store_true", help="use the parameters written by refine_cca")
    p.add_argument("--max_dim", type=int, default=None, help="keep only the first dimensions of the embedding space")
    p.add_argument("--seed", type=int, default=23)
    p.add_argument("--train_split", type=str, default=None)
    p.add_argument("--config", type=str, default=None)
    p.add_argument("--dump_results", action="store_true", help="write the measures to eval_<tag>_<dir>.yaml")
    p.add_argument("--gpus", type=int, default=1, help="shard the test pairs over this many GPUs of the node")
    p.add_argument("--comm", choices=["rccl", "host"], default="rccl",
                   help="exchange between the ranks: the library's RCCL communicator, or host callbacks over the TCP "
                        "hub (several ranks on ONE GPU, which RCCL refuses; tests)")
    return p.parse_args(argv)


def _embed(fn_view1, fn_view2, sheets, specs, prepare, chunk=100):
    """batch_compute2 x 2 of the reference (:107-108).  With the model's own `prepare` (or none) both arrays go to
    the library whole and unprepared (network.CompiledFunction.embed_raw); any other callable is applied per chunk
    of 100 on the host, like the reference.  Same rows either way."""
    if prepare is None or network.is_fused_prepare(fn_view1.net, prepare):
        lv1 = fn_view1.embed_raw(sheets) if prepare is not None else fn_view1(sheets, specs)
        return lv1, fn_view2.embed_raw(specs)
    out1, out2 = [], []
    for lo in range(0, sheets.shape[0], chunk):
        a, b = prepare(sheets[lo:lo + chunk]), specs[lo:lo + chunk]
        out1.append(fn_view1(a, b))
        out2.append(fn_view2(a, b))
    return np.concatenate(out1, axis=0), np.concatenate(out2, axis=0)


def main(argv=None):
    args = _arguments(argv)
    from . import launch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import sys
        raise SystemExit(launch.spawn_ranks([sys.executable, "-m", __package__ + ".run_eval"] +
                                            list(sys.argv[1:] if argv is None else argv), args.gpus))
    rank, local_rank, world = launch.world_from_env()
    if world > 1:
        os.environ["ASR_DEVICE"] = str(launch.device_for(local_rank))
    say = print if rank == 0 else (lambda *a, **k: None)
    model, _ = select_model(args.model)
    prepare = getattr(model, "prepare", None)
    layers = model.build_model(show_model=False)
    tag = compile_tag(args.train_split, args.config)
    folder = model.EXP_NAME + ("_est_UV" if args.estimate_UV else "")
    param_file = os.path.join(EXP_ROOT, folder, "params.pkl" if tag is None else "params_%s.pkl" % tag)
    say("model %s, tag %s\nparameters: %s" % (model.EXP_NAME, tag, param_file))
    params = load_params(param_file)
    if isinstance(params[0], list):            # very old dumps hold one full list per layer handle (:76-79)
        params = params[-1]
    network.set_all_param_values(layers, params)

    data = select_data(args.data, args.train_split, args.config, args.seed, test_only=True)
    view1, view2, latent1, latent2 = layers
    both = [view1.input_var, view2.input_var]
    fn1 = network.function(both, network.get_output(latent1, deterministic=True))
    fn2 = network.function(both, network.get_output(latent2, deterministic=True))
    engine = fn1.engine
    hub = launch.join(engine, transport=args.comm) if world > 1 else None

    pool = data["test"]
    n_test = pool.shape[0] if args.n_test is None else args.n_test
    subset = np.linspace(0, pool.shape[0] - 1, n_test).astype(int)
    if world > 1:                                  # this rank's contiguous share of the test pairs
        from . import distributed
        lo, hi = distributed.shard_range(len(subset), rank, world)
        if hi <= lo:
            raise SystemExit("--n_test %d is smaller than the number of GPUs (%d)" % (len(subset), world))
        subset = subset[lo:hi]
    sheets, specs = pool[subset]
    queries, candidates = _embed(fn1, fn2, sheets, specs, prepare)
    if args.V2_to_V1:
        queries, candidates = flip_variables(queries, candidates)
    if args.max_dim is not None:
        queries, candidates = queries[:, :args.max_dim], candidates[:, :args.max_dim]

    if world > 1:
        comm = distributed.EngineComm(engine)
        (mean_rank, median_rank, mean_dist, hits, mean_ap), _, _, match_dist = distributed.sharded_eval_retrieval(
            lambda q, c_all, off, n: engine.rank(q, c_all, query_offset=off, n1_global=n), queries, candidates, comm,
            details=True)
        n_test = n_cand = int(match_dist.shape[0])
        dim = queries.shape[1]
        say("queries %r, candidates %r" % ((n_test, dim), (n_cand, dim)))
    else:
        n_test, n_cand = queries.shape[0], candidates.shape[0]
        say("queries %r, candidates %r" % (queries.shape, candidates.shape))
        mean_rank, median_rank, mean_dist, hits, mean_ap = eval_retrieval(queries, candidates, engine=engine)
        _, match_dist, _ = engine.rank(queries, candidates)       # distance of every query to its own match

    recall = dict((int(k), 100.0 * hits[k] / n_test) for k in sorted(hits))
    say("\nHit Rates:")
    for k in sorted(recall):
        say("  R@%-2d %7.3f %%  (%d of %d)" % (k, recall[k], hits[k], n_test))
    say("median rank %.2f, mean rank %.2f of %d candidates" % (median_rank, mean_rank, n_cand))
    say("MAP %.3f; distance to the match: mean %.5f, min %.5f, median %.5f, max %.5f"
        % (mean_ap, mean_dist, match_dist.min(), np.median(match_dist), match_dist.max()))

    results = {"map": float(mean_ap), "med_rank": float(median_rank),
               "recall_at_k": dict((str(k), float(v)) for k, v in recall.items())}
    if args.dump_results and rank == 0:
        direction = "A2S" if args.V2_to_V1 else "S2A"
        out = param_file.replace("params_", "eval_").replace(".pkl", "_%s.yaml" % direction)
        with open(out, "w") as fp:
            yaml.dump(results, fp, default_flow_style=False)
    launch.leave(hub)
    return results


if __name__ == "__main__":
    main()
