"""Training / evaluation engine - mirrors audio_sheet_retrieval/utils/train_dcca_pool.py.

eval_retrieval (:28-82) runs on the GPU (float64 distances in SciPy's summation
order, rank by counting); the training functions (create_iter_functions :85-167,
train :185-315, fit :318-543) wrap the HIP training step."""
from __future__ import annotations

import numpy as np


def atomic_pickle_dump(obj, path, protocol=2):
    """pickle to `path` through a temporary file in the same directory + rename: a process that is killed while it
    writes (the dead-peer watchdog of a multi-GPU job ends a rank with os._exit) leaves the previous file intact
    instead of a truncated one.  Same bytes as the reference's `pickle.dump(..., file(dump_file, "wb"))`
    (utils/train_dcca_pool.py:395-401, 488-489)."""
    import os
    import pickle
    if os.path.exists(path) and not os.path.isfile(path):      # a device node (/dev/null on the ranks that do not log)
        with open(path, "wb") as fp:
            pickle.dump(obj, fp, protocol=protocol)
        return
    tmp = "%s.tmp.%d" % (path, os.getpid())
    try:
        with open(tmp, "wb") as fp:
            pickle.dump(obj, fp, protocol=protocol)
            fp.flush()
            os.fsync(fp.fileno())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


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
    # np.diag(dists).mean() (:77): for equal list sizes d* is the diagonal; for unequal sizes the diagonal pairs row i
    # with candidate i, which is not query i's match, so it is computed by a second (square) call
    n_diag = min(lv1_cca.shape[0], lv2_cca.shape[0])
    if lv1_cca.shape[0] != lv2_cca.shape[0]:
        dstar = engine.rank(lv1_cca[:n_diag], lv2_cca[:n_diag])[1]
    mean_dist = float(np.mean(dstar[:n_diag]))
    map_ = np.mean(1.0 / ranks.astype(np.float64))
    return mean_rank, median_rank, mean_dist, hit_rates, map_


# --------------------------------------------------------------------------
# compiled-callable protocol (utils/train_dcca_pool.py:85-167)
# --------------------------------------------------------------------------
class SharedScalar(object):
    """theano.shared(np.float32(lr)) stand-in: fit() mutates it with set_value (:343,520,525)."""

    def __init__(self, value):
        self._v = np.float32(value)

    def get_value(self):
        return self._v

    def set_value(self, value):
        self._v = np.float32(value)


class _OptStateHandle(object):
    """What fit() snapshots / restores through iter_funcs['updates'].keys() (:396,515-516): the Adam
    moments and step counter living on the device."""

    def __init__(self, funcs):
        self._funcs = funcs

    def get_value(self):
        eng = self._funcs.engine
        return eng.get_opt_state() if self._funcs.begun else None

    def set_value(self, state):
        if state is not None:
            self._funcs._ensure(self._funcs.batch_cap)
            self._funcs.engine.set_opt_state(state)


class _Updates(object):
    def __init__(self, funcs):
        self._handle = _OptStateHandle(funcs)

    def keys(self):
        return [self._handle]


class IterFunctions(dict):
    """dict(train=, valid=, test=, compute_output=, init_cca=, compute_gradients=, all_params=, updates=)
    like the reference returns (:166-167); the callables run on the HIP library."""

    def __init__(self, layers, learning_rate, init_cca=False):
        super(IterFunctions, self).__init__()
        self.net = layers[0].net
        self.engine = self.net.engine
        self.lr = learning_rate
        self.begun = False
        self.batch_cap = 0
        self["train"] = self._train
        self["valid"] = self["test"] = self._valid
        self["compute_output"] = self._compute_output
        self["init_cca"] = self._init_cca if init_cca else False
        self["compute_gradients"] = self._compute_gradients
        self["all_params"] = [i for i in range(90) if i % 5 <= 2]
        self["updates"] = _Updates(self)

    def _ensure(self, batch):
        if not self.begun or batch > self.batch_cap:
            state = self.engine.get_opt_state() if self.begun else None
            self.batch_cap = max(int(batch), self.batch_cap)
            rank, world = self.engine.comm_info()
            if world > 1:
                # rank 0 times the training step's schedules first; the others read them from the job's tune cache
                # (the barrier carries rank 0's outcome: if its train_begin fails, every rank stops)
                from ..distributed import engine_flag_barrier, tune_in_rank_order
                tune_in_rank_order(self.engine, engine_flag_barrier(self.engine), rank,
                                   trigger=lambda: self.engine.train_begin(self.batch_cap))
            else:
                self.engine.train_begin(self.batch_cap)
            if state is not None:
                self.engine.set_opt_state(state)
            self.begun = True
            self._global_batch = None           # a fresh training state starts with equal shards

    def close(self):
        """Release the device training state (Adam moments, activations of the training step, side streams).  fit()
        calls it when training ends: the context then accepts other snippet sizes again (asr_set_input_size refuses
        them while a training state is alive) and its parameters are the ones last set or trained."""
        if self.begun:
            self.engine.train_end()
            self.begun = False
            self.batch_cap = 0
            self._global_batch = None

    def _sizes(self, X1, X2):
        rsz = self.net.model_name.endswith("_rsz")
        eng = self.engine
        if X1.shape[2:] != (eng.net_h1, eng.net_w1) or X2.shape[2:] != (eng.cfg.h2, eng.cfg.w2):
            if self.begun:
                raise ValueError("input size changed after training started")
            eng.set_input_size(1, X1.shape[2] * (2 if rsz else 1), X1.shape[3] * (2 if rsz else 1))
            eng.set_input_size(2, X2.shape[2], X2.shape[3])

    def _shard(self, X1, X2):
        """data parallel (engine.comm_init*): every rank iterates the same batches and trains on its contiguous share
        of the rows (distributed.shard_range - no row is dropped: a batch of 100 on 8 ranks is 4 x 13 + 4 x 12 rows
        and the library is told the size of the whole batch).  Allocates the training state on first use: every rank
        for the LARGEST shard, so that all of them look up the same schedules in the job's tune cache."""
        rank, world = self.engine.comm_info()
        n = int(X1.shape[0])
        if world > 1:
            from ..distributed import shard_batch
            X1, X2 = shard_batch([X1, X2], rank, world)
        self._sizes(X1, X2)
        self._ensure(-(-n // world))
        if world > 1 and getattr(self, "_global_batch", None) != n:
            self.engine.train_set_global_batch(n)
            self._global_batch = n
        return X1, X2

    def _train(self, X1, X2):
        X1, X2 = self._shard(X1, X2)
        loss, corr = self.engine.train_step(X1, X2, float(self.lr.get_value()))
        return [np.float32(loss), corr]

    def _init_cca(self, X1, X2):
        """burn-in pass (:160-162): train-mode forward, only the running averages change."""
        X1, X2 = self._shard(X1, X2)
        return list(self.engine.burn_in(X1, X2))

    def _compute_gradients(self, X1, X2):
        """theano.function(input_vars, all_grads) (:164): one array per entry of all_params, no update applied."""
        X1, X2 = self._shard(X1, X2)
        flat, _ = self.engine.compute_gradients(X1, X2)
        sizes = self.engine.param_sizes()
        offs = np.concatenate([[0], np.cumsum(sizes)])
        shapes = self.net.shapes
        return [flat[offs[i]:offs[i + 1]].reshape(shapes[i]).copy() for i in self["all_params"]]

    def _valid(self, X1, X2):
        return [np.float32(self.engine.valid_loss(X1, X2))]

    def _compute_output(self, X1, X2):
        return list(self.engine.embed_both(X1, X2, prepared=True))


def create_iter_functions(layers, objectives, compute_updates, learning_rate, l_2, l_1, init_cca=False):
    """Create functions for training, validation and testing (:85-167).  `objectives`, `compute_updates`,
    l_2 describe what the fused HIP step implements; they are checked, not compiled."""
    net = layers[0].net
    obj = objectives()
    if abs(obj.gamma - net.hyper["gamma"]) > 1e-12:
        raise ValueError("objective margin %g differs from the network's GAMMA %g" % (obj.gamma, net.hyper["gamma"]))
    if not (getattr(obj, "weight", 1.0) > 0.0):
        raise ValueError("objective weight must be positive")

    if l_1 is not None:
        raise NotImplementedError("L1 penalty: the two models use L1 = None")
    if (l_2 or 0.0) != net.hyper["l2"]:
        raise ValueError("l_2=%r differs from the network's L2 %r" % (l_2, net.hyper["l2"]))
    upd = compute_updates(None, None, learning_rate)
    if upd.get("rule") != "adam":
        raise NotImplementedError("only lasagne.updates.adam is built")
    if not isinstance(learning_rate, SharedScalar):
        learning_rate = SharedScalar(learning_rate)
    funcs = IterFunctions(layers, learning_rate, init_cca=init_cca)
    # get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69): both directions and the weight run
    # inside the fused step (asr_set_objective); the two models bind weight 1, one direction
    funcs.engine.set_objective(getattr(obj, "weight", 1.0), obj.gamma, getattr(obj, "symmetric", False))
    return funcs


def pretrain(iter_funcs, dataset, train_batch_iter, epochs=3):
    """Run some epochs over the training data to initialise the CCALayer running averages (:170-182)."""
    if not iter_funcs["init_cca"]:
        return
    print("Pretraining for %d epochs..." % epochs)
    from .batch_iterators import threaded_generator_from_iterator
    for _ in range(epochs):
        iterator = train_batch_iter(dataset["train"])
        generator = threaded_generator_from_iterator(iterator)
        for X_b, Z_b in generator:
            iter_funcs["init_cca"](X_b, Z_b)


# --------------------------------------------------------------------------
# epoch generator and fit (:185-315, :318-543)
# --------------------------------------------------------------------------
def _collect_outputs(iter_funcs, generator, n_needed, with_loss=False):
    """Run `compute_output` (and optionally `valid`) over a batch generator, keeping the first n_needed rows."""
    V1, V2, losses = None, None, []
    for batch in generator:
        if with_loss:
            losses.append(iter_funcs["valid"](*batch)[0])
        if V1 is None or V1.shape[0] < n_needed:
            a, b = iter_funcs["compute_output"](*batch)
            V1 = a if V1 is None else np.vstack([V1, a])
            V2 = b if V2 is None else np.vstack([V2, b])
    return V1, V2, losses


def train(iter_funcs, dataset, train_batch_iter, valid_batch_iter, fit_cca):
    """Generator over epochs (:185-315): one sub-epoch of updates, retrieval metrics on >= 1000 train and
    validation pairs, yields the reference's result dict."""
    import copy
    import itertools
    import sys
    import time

    from .batch_iterators import threaded_generator_from_iterator
    from .cca import CCA

    for epoch in itertools.count(1):
        losses, evals = [], []
        recent = np.zeros(5, dtype=np.float32)
        t_start = t_last = time.time()
        gen = threaded_generator_from_iterator(train_batch_iter(dataset["train"]))
        for i_batch, batch in enumerate(gen):
            res = iter_funcs["train"](*batch)
            losses.append(res[0])
            if len(res) > 1:
                evals.append(res[1])
            now = time.time()
            recent = np.roll(recent, -1)
            recent[-1] = now - t_last
            t_last = now
            ups = 1.0 / recent.mean()                      # updates per second, mean of the last 5 (:221-224)
            perc = 100 * (float(i_batch + 1) / train_batch_iter.n_batches)
            bar = "|" + int(perc // 4) * "#" + (25 - int(perc // 4)) * "-" + "|"
            print(" (%d%%) %s time: %.2fs, ups: %.2f, loss: %.5f" % (perc, bar, now - t_start, ups, np.mean(losses)),
                  end="\r")
            sys.stdout.flush()

        n_valid_cca = int(np.min([1000, dataset["valid"].shape[0]]))
        it_copy = copy.copy(train_batch_iter)
        it_copy.epoch_counter = 0
        V1_tr, V2_tr, _ = _collect_outputs(iter_funcs, threaded_generator_from_iterator(it_copy(dataset["train"])),
                                           n_valid_cca)
        cca = None
        if fit_cca:
            cca = CCA(method="svd", engine=iter_funcs.engine)
            cca.fit(V1_tr, V2_tr, verbose=False)
            V1_tr, V2_tr = cca.transform_V1(V1_tr), cca.transform_V2(V2_tr)
        _, med_tr, dist_tr, hits_tr, map_tr = eval_retrieval(V1_tr, V2_tr, engine=iter_funcs.engine)
        rank_tr = 1.0 - float(hits_tr[10]) / len(V1_tr)

        print("\x1b[K", end="\r")
        print(" ")
        V1_va, V2_va, va_losses = _collect_outputs(
            iter_funcs, threaded_generator_from_iterator(valid_batch_iter(dataset["valid"])), n_valid_cca, True)
        if cca is not None:
            V1_va, V2_va = cca.transform_V1(V1_va), cca.transform_V2(V2_va)
        _, med_va, dist_va, hits_va, map_va = eval_retrieval(V1_va, V2_va, engine=iter_funcs.engine)
        rank_va = 1.0 - float(hits_va[10]) / 1000            # the reference hard-codes /1000 (:299)

        result = {"number": epoch, "train_loss": np.mean(losses), "valid_loss": np.mean(va_losses),
                  "mean_cos_dist_tr": dist_tr, "mean_cos_dist_va": dist_va,
                  "mean_rank_tr": rank_tr, "mean_rank_va": rank_va, "med_rank_tr": med_tr, "med_rank_va": med_va,
                  "map_tr": map_tr, "map_va": map_va,
                  "evals_tr": np.asarray(evals).mean(axis=0) if evals else None}
        # data parallel: fit() on every rank takes its early-stopping / refinement / learn-rate decisions from rank
        # 0's numbers (a one-rank flip of `map_va >= best` would let the parameters diverge or dead-lock the
        # gradient all-reduce); a no-op on one GPU
        from ..distributed import broadcast_epoch
        yield broadcast_epoch(iter_funcs.engine, result)


def fit(layers, data, objectives, train_batch_iter, valid_batch_iter, num_epochs=100, patience=20, learn_rate=0.01,
        update_learning_rate=None, l_2=None, l_1=None, compute_updates=None, exp_name="ff", out_path=None,
        dump_file=None, fit_cca=True, do_raise=True, pretrain_epochs=0, refinement_steps=0, lr_multiplier=0.1,
        refinement_patience=10, log_file=None):
    """Train model (:318-543): early stopping on map_va >= best (:391), best parameters (and optimiser state)
    kept and pickled (:392-401), on exhausted patience reload them, multiply the learning rate and continue up to
    refinement_steps times (:492-520), NaN loss forces the patience exit (:410-411), results pickled every epoch
    (:476-489).  Returns (l_out, best map_va)."""
    import os
    import pickle
    import time

    from .. import network

    os.makedirs(out_path, exist_ok=True)        # every rank of a data-parallel run gets here
    if log_file is None:
        log_file = os.path.join(out_path, "results.pkl")
    print("\n\nRunning Test Case: " + exp_name)

    learning_rate = SharedScalar(learn_rate)
    if update_learning_rate is None:
        def update_learning_rate(lr, e=None):
            return lr
    learning_rate.set_value(update_learning_rate(learn_rate))

    print("Building model and compiling functions...")
    iter_funcs = create_iter_functions(layers, objectives, compute_updates, learning_rate, l_2, l_1,
                                       init_cca=pretrain_epochs > 0)
    history = dict((k, []) for k in ("pred_tr_err", "pred_val_err", "dist_tr", "dist_val", "rank_tr", "rank_val",
                                     "map_tr", "map_val", "evals_tr"))
    best = dict(tr_loss=1e7, va_loss=1e7, tr_dist=1e7, va_dist=1e7, med_tr=1e7, med_va=1e7, map_tr=0.0, map_va=0.0)
    best_model = network.get_all_param_values(layers)
    best_opt_state, best_epoch = None, 0
    since_improvement = 0
    print("Starting training...")
    tick = time.time()
    try:
        if pretrain_epochs:                                             # :363-365
            pretrain(iter_funcs, data, train_batch_iter, pretrain_epochs)
        for epoch in train(iter_funcs, data, train_batch_iter, valid_batch_iter, fit_cca):
            if epoch["map_va"] >= best["map_va"]:                       # :391
                since_improvement = 0
                best_epoch = epoch["number"]
                best_model = network.get_all_param_values(layers)
                best_opt_state = [u.get_value() for u in iter_funcs["updates"].keys()]
                if dump_file is not None:
                    atomic_pickle_dump(best_model, dump_file, protocol=2)
            since_improvement += 1
            print("Epoch {} of {} took {:.3f}s (patience: {})".format(
                epoch["number"], num_epochs, time.time() - tick, patience - since_improvement + 1))
            tick = time.time()
            if np.isnan(epoch["train_loss"]):                            # :410-411
                since_improvement = patience + 1
            best["tr_loss"] = min(best["tr_loss"], epoch["train_loss"])
            best["va_loss"] = min(best["va_loss"], epoch["valid_loss"])
            best["tr_dist"] = min(best["tr_dist"], epoch["mean_cos_dist_tr"])
            best["va_dist"] = min(best["va_dist"], epoch["mean_cos_dist_va"])
            best["map_tr"] = max(best["map_tr"], epoch["map_tr"])
            best["map_va"] = max(best["map_va"], epoch["map_va"])
            best["med_tr"] = min(best["med_tr"], epoch["med_rank_tr"])
            best["med_va"] = min(best["med_va"], epoch["med_rank_va"])
            print("  lr: %.9f" % learn_rate)
            print("  costs_tr %.5f costs_va %.5f" % (epoch["train_loss"], epoch["valid_loss"]))
            print("  dist_tr %.5f dist_va %.5f" % (epoch["mean_cos_dist_tr"], epoch["mean_cos_dist_va"]))
            print("  map_tr %.2f map_va %.2f  | medr_tr %.2f medr_va %.2f" % (
                100 * epoch["map_tr"], 100 * epoch["map_va"], epoch["med_rank_tr"], epoch["med_rank_va"]))
            for key, src in (("pred_tr_err", "train_loss"), ("pred_val_err", "valid_loss"),
                             ("dist_tr", "mean_cos_dist_tr"), ("dist_val", "mean_cos_dist_va"),
                             ("rank_tr", "mean_rank_tr"), ("rank_val", "mean_rank_va"),
                             ("map_tr", "map_tr"), ("map_val", "map_va"), ("evals_tr", "evals_tr")):
                history[key].append(epoch[src])
            atomic_pickle_dump(history, log_file, protocol=2)

            if since_improvement > patience:                             # :492-520
                print("Early Stopping!")
                print("Best Epoch: %d, Validation Loss: %.5f: Dist: %.5f Map: %.2f" % (
                    best_epoch, best["va_loss"], best["va_dist"], 100 * best["map_va"]))
                if refinement_steps <= 0:
                    break
                print("Loading best parameters so far and refining (%d) with decreased learn rate ..."
                      % refinement_steps)
                since_improvement = 0
                patience = refinement_patience
                refinement_steps -= 1
                network.set_all_param_values(layers, best_model)
                for u, value in zip(iter_funcs["updates"].keys(), best_opt_state or []):
                    u.set_value(value)
                learn_rate = np.float32(learn_rate * lr_multiplier)
                learning_rate.set_value(learn_rate)
            learn_rate = update_learning_rate(learn_rate, epoch["number"])   # :523-525
            if learn_rate is not None:
                learning_rate.set_value(learn_rate)
            if epoch["number"] >= num_epochs:
                break
    except KeyboardInterrupt:
        pass
    except Exception:
        if do_raise:
            raise
        return layers[-1], best["map_va"]
    finally:
        close = getattr(iter_funcs, "close", None)      # the reference has nothing to release (Theano shared variables)
        if close is not None:
            close()
    network.set_all_param_values(layers, best_model)
    return layers[-1], best["map_va"]
