"""Training / evaluation engine - mirrors audio_sheet_retrieval/utils/train_dcca_pool.py.

eval_retrieval (:28-82) runs on the GPU (float64 distances in SciPy's summation
order, rank by counting); the training functions (create_iter_functions :85-167,
train :185-315, fit :318-543) wrap the HIP training step."""
from __future__ import annotations

import numpy as np


def eval_retrieval(lv1_cca, lv2_cca, engine=None):
    """Compute retrieval eval measures (utils/train_dcca_pool.py:28-82).
    Returns (mean_rank, median_rank, mean_dist, hit_rates, map) with
    hit_rates = {1,5,10,25: count}; hit@k <=> rank <= k, rank = 1-based position
    of the (first) correct item in a stable ascending sort of the float64 cosine
    distances - identical to the reference's argsort procedure on tie-free rows
    (NumPy's default sort is not stable, so the reference leaves ties undefined)."""
    if engine is None:
        from .. import runtime
        engine = runtime.default_engine()
    lv1_cca = np.ascontiguousarray(lv1_cca, dtype=np.float32)
    lv2_cca = np.ascontiguousarray(lv2_cca, dtype=np.float32)
    ranks, dstar, _ties = engine.rank(lv1_cca, lv2_cca)
    hit_rates = {key: int(np.count_nonzero(ranks <= key)) for key in (1, 5, 10, 25)}
    mean_rank = np.mean(ranks)
    median_rank = np.median(ranks)
    # np.diag(dists).mean() (:77): for equal list sizes d* is the diagonal
    mean_dist = float(np.mean(dstar[:min(lv1_cca.shape[0], lv2_cca.shape[0])]))
    map_ = np.mean(1.0 / ranks.astype(np.float64))
    return mean_rank, median_rank, mean_dist, hit_rates, map_


def fit(*args, **kwargs):
    raise NotImplementedError("training (create_iter_functions/train/fit) is not built yet in this round")
