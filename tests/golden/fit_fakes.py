"""Scripted collaborators for pinning the CONTROL FLOW of fit() (reference utils/train_dcca_pool.py:318-543) against
the mirror in audio_sheet_retrieval_amd/utils/train_dcca_pool.py: early stopping on `map_va >= best`, best-model and
optimiser-state snapshots, refinement restarts with a multiplied learning rate, the NaN exit, the epoch limit, the
results / parameter pickles.  The epoch generator `train` and `create_iter_functions` (Theano in the reference, HIP
here) are replaced on both sides by the same script, which is NOT what is pinned.  Used by
make_reference_golden.py (runs the REFERENCE's fit, build container only) and tests/test_reference_golden.py (runs the
mirror's).  Nothing here comes from the reference."""
import numpy as np

NAN = float("nan")

#: (tag, fit keyword arguments, per-epoch (map_va, train_loss) script)
CASES = [
    ("refine_twice", dict(num_epochs=40, patience=2, refinement_steps=2, lr_multiplier=0.5, refinement_patience=1,
                          learn_rate=0.002),
     [(0.10, 1.0), (0.20, 0.9), (0.20, 0.8), (0.15, 0.7), (0.10, 0.6), (0.10, 0.6), (0.30, 0.5), (0.10, 0.5),
      (0.10, 0.5), (0.10, 0.5), (0.10, 0.5), (0.10, 0.5), (0.10, 0.5), (0.10, 0.5)]),
    ("nan_stops", dict(num_epochs=40, patience=5, refinement_steps=0, lr_multiplier=0.5, refinement_patience=3,
                       learn_rate=0.01),
     [(0.10, 1.0), (0.12, 0.9), (0.11, NAN), (0.50, 0.5), (0.60, 0.4)]),
    ("nan_then_refine", dict(num_epochs=40, patience=5, refinement_steps=1, lr_multiplier=0.1, refinement_patience=2,
                             learn_rate=0.01),
     [(0.10, 1.0), (0.12, 0.9), (0.05, NAN), (0.50, 0.5), (0.40, 0.4), (0.40, 0.4), (0.40, 0.4), (0.40, 0.4)]),
    ("epoch_limit", dict(num_epochs=4, patience=20, refinement_steps=3, lr_multiplier=0.5, refinement_patience=10,
                         learn_rate=0.002),
     [(0.10, 1.0), (0.09, 0.9), (0.30, 0.8), (0.20, 0.7), (0.90, 0.1)]),
    ("decaying_schedule", dict(num_epochs=40, patience=1, refinement_steps=1, lr_multiplier=0.5, refinement_patience=1,
                               learn_rate=0.004, decay=True),
     [(0.10, 1.0), (0.05, 0.9), (0.05, 0.9), (0.20, 0.9), (0.05, 0.9), (0.05, 0.9), (0.05, 0.9)]),
]


class Shared(object):
    """theano.shared stand-in: like a Theano shared variable it keeps the dtype it was created with (the learning
    rate is created from np.float32(learn_rate), reference :343, so later set_value(python float) is stored as
    float32)"""

    def __init__(self, value):
        self.value = value

    def get_value(self):
        return self.value

    def set_value(self, value):
        if isinstance(self.value, np.generic):
            value = self.value.dtype.type(value)
        self.value = value


class Layers(list):
    """the four layer handles fit() passes around; `store` is the parameter list get/set_all_param_values act on"""

    def __init__(self):
        super(Layers, self).__init__(["l_view1", "l_view2", "l_v1latent", "l_v2latent"])
        self.store = [np.zeros(3, np.float32), np.full((2, 2), 10.0, np.float32)]
        self.sets = 0


def get_all_param_values(layers):
    return [p.copy() for p in layers.store]


def set_all_param_values(layers, values):
    layers.store = [np.array(v, copy=True) for v in values]
    layers.sets += 1


class Script(object):
    """create_iter_functions + train stand-ins: every scripted epoch moves the parameters and the optimiser state by
    one and records the learning rate, parameter tag and optimiser tag it started from"""

    def __init__(self, epochs):
        self.epochs = epochs
        self.seen = []
        self.lr = self.layers = self.opt = None

    def create_iter_functions(self, layers, objectives, compute_updates, learning_rate, l_2, l_1, init_cca=False):
        self.layers, self.lr = layers, learning_rate
        self.opt = Shared(np.zeros(2, np.float32))
        return dict(updates={self.opt: None}, init_cca=False)

    def train(self, iter_funcs, dataset, train_batch_iter, valid_batch_iter, fit_cca):
        for number, (map_va, train_loss) in enumerate(self.epochs, 1):
            self.seen.append([number, float(self.lr.get_value()), float(self.layers.store[0][0]),
                              float(self.opt.get_value()[0]), float(self.layers.sets)])
            self.layers.store = [p + 1 for p in self.layers.store]
            self.opt.set_value(self.opt.get_value() + 1)
            yield {"number": number, "train_loss": train_loss, "valid_loss": 2.0 - map_va,
                   "mean_cos_dist_tr": 1.0 - map_va / 2, "mean_cos_dist_va": 1.0 - map_va,
                   "mean_rank_tr": 0.5, "mean_rank_va": 0.6, "med_rank_tr": 3.0, "med_rank_va": 4.0 + number,
                   "map_tr": map_va / 2, "map_va": map_va, "evals_tr": np.arange(3.0) * number}


def schedule(decay):
    if decay:
        def update_learning_rate(lr, epoch=None):
            return lr if epoch is None else np.float32(lr * 0.9)
    else:
        def update_learning_rate(lr, epoch=None):
            return lr
    return update_learning_rate


def summarize(script, layers, returned, log_file, dump_file):
    """what both sides are compared on"""
    import pickle
    with open(log_file, "rb") as fp:
        hist = pickle.load(fp)
    with open(dump_file, "rb") as fp:
        dumped = pickle.load(fp)
    out = {"seen": np.array(script.seen, np.float64), "final0": layers.store[0], "final1": layers.store[1],
           "returned_map": np.float64(returned[1]), "dumped0": np.asarray(dumped[0]), "sets": np.int64(layers.sets),
           "opt_end": script.opt.get_value()}
    for key in ("pred_tr_err", "pred_val_err", "dist_tr", "dist_val", "rank_tr", "rank_val", "map_tr", "map_val"):
        out["hist_" + key] = np.asarray(hist[key], np.float64)
    out["hist_evals_tr"] = np.asarray(hist["evals_tr"], np.float64)
    return out
