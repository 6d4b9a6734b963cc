"""Layout of the reference's parameter pickle
(lasagne.layers.get_all_param_values(layers); utils/train_dcca_pool.py:395-401,
loaded in run_train.py:99-101, run_eval.py:74-82, retrieval_wrapper.py:27-29).

97 float32 arrays: tower-1 blocks 1..9 each [W (O,I,kh,kw), beta, gamma, mean,
inv_std] (0..44), tower-2 likewise (45..89), CCALayer [U, V, mean1, mean2, S12,
S11, S22] (90..96; models/lasagne_extensions/layers/cca.py:69-77).
"""
from __future__ import annotations

DIM_LATENT = 32
NUM_FILTERS = {"mutopia_ccal_cont": 12, "mutopia_ccal_cont_rsz": 24}

# indices into the flat list
IDX_U, IDX_V, IDX_MEAN1, IDX_MEAN2, IDX_S12, IDX_S11, IDX_S22 = 90, 91, 92, 93, 94, 95, 96


def tower_channels(nf):
    """(C_in, C_out, ksize) of the nine conv blocks (models/mutopia_ccal_cont.py:76-94)."""
    return [(1, nf, 3), (nf, nf, 3), (nf, 2 * nf, 3), (2 * nf, 2 * nf, 3),
            (2 * nf, 4 * nf, 3), (4 * nf, 4 * nf, 3), (4 * nf, 4 * nf, 3), (4 * nf, 4 * nf, 3),
            (4 * nf, DIM_LATENT, 1)]


def param_shapes(model_name):
    nf = NUM_FILTERS[model_name]
    shapes = []
    for _tower in range(2):
        for ci, co, k in tower_channels(nf):
            shapes += [(co, ci, k, k), (co,), (co,), (co,), (co,)]
    d = DIM_LATENT
    return shapes + [(d, d), (d, d), (d,), (d,), (d, d), (d, d), (d, d)]


def model_name_from_params(params):
    """Infer the model variant from a loaded pickle (first conv has nf filters)."""
    nf = params[0].shape[0]
    for name, v in NUM_FILTERS.items():
        if v == nf:
            return name
    raise ValueError("no model variant with %d first-layer filters" % nf)
