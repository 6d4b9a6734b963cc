"""Embedding service object: the public surface of the reference's `RetrievalWrapper`
(audio_sheet_retrieval/retrieval_wrapper.py:12-77) on top of the HIP engine.

Same constructor signature, the attributes its callers read (`code_dim`, `shape_view1`, `shape_view2`,
`dummy_in_v1/2`, `prepare_view_1/2`) and the two methods `compute_view_1(X)`, `compute_view_2(Z)` -> (n, 32) float32.
Differences: there is no compile step (the kernels are built ahead of time), only the tower an output depends on is
evaluated (the reference feeds a dummy second view; rows are independent in deterministic mode, so results agree).
When `prepare_view_1` is the model's own `prepare` (or None) the array goes to the library as it is - uint8 or float
0..255 - in ONE call: /255 (and the _rsz halving) run inside the first kernel and the library pipelines staging
copy, H2D and the tower over the array (csrc/asr_api.hip: embed_host).  Any other callable is applied on the host,
chunk by chunk like the reference (:54-60); the results are bit-identical either way (tests/test_gpu_dropin_api.py).
"""
import pickle

import numpy as np

from . import network

_FORWARD_CHUNK = 100          # the reference embeds in batches of 100 (:55, :71); results do not depend on it


def load_params(param_file):
    """Parameter pickle -> list of 97 arrays.  Files written by the Python-2 reference
    (cPickle, utils/train_dcca_pool.py:399-401) need latin1 decoding."""
    with open(param_file, "rb") as fp:
        blob = fp.read()
    try:
        return pickle.loads(blob)
    except UnicodeDecodeError:
        return pickle.loads(blob, encoding="latin1")


def _in_chunks(fn, arr, chunk):
    if arr.shape[0] == 0:
        return np.zeros((0, 32), np.float32)
    return np.concatenate([fn(arr[i:i + chunk]) for i in range(0, arr.shape[0], chunk)], axis=0)


def _whole(fn, prepared):
    """the compiled function on a whole prepared view-1 array (the library chunks and pipelines it itself)"""
    return fn.engine.embed_view1(prepared, prepared=True, features=fn.features)


class RetrievalWrapper(object):

    def __init__(self, model, param_file, prepare_view_1=None, prepare_view_2=None):
        self.prepare_view_1, self.prepare_view_2 = prepare_view_1, prepare_view_2
        self.code_dim = model.DIM_LATENT
        view1, view2, latent1, latent2 = model.build_model(show_model=False)
        network.set_all_param_values([view1, view2, latent1, latent2], load_params(param_file))
        both = [view1.input_var, view2.input_var]
        self.compute_v1_latent = network.function(both, network.get_output(latent1, deterministic=True))
        self.compute_v2_latent = network.function(both, network.get_output(latent2, deterministic=True))
        self.shape_view1, self.shape_view2 = view1.output_shape[1:], view2.output_shape[1:]
        # kept for callers that look at them (:41-42); this implementation never feeds them to a tower
        self.dummy_in_v1 = np.zeros((1,) + tuple(self.shape_view1), np.float32)
        self.dummy_in_v2 = np.zeros((1,) + tuple(self.shape_view2), np.float32)

    def _view(self, which, data, prepare):
        fn = self.compute_v1_latent if which == 1 else self.compute_v2_latent
        data = np.asarray(data)
        if data.shape[0] == 0:
            return np.zeros((0, self.code_dim), np.float32)
        if prepare is None or (which == 1 and network.is_fused_prepare(fn.net, prepare)):
            if which == 1 and prepare is None:
                return _whole(fn, data)                   # already prepared by the caller (tutorials pass prepared floats)
            return fn.embed_raw(data)
        other = self.dummy_in_v2 if which == 1 else self.dummy_in_v1

        def one(block):
            block = prepare(np.array(block))             # callers' arrays are never modified (:54)
            pad = np.broadcast_to(other, (block.shape[0],) + other.shape[1:])
            return fn(block, pad) if which == 1 else fn(pad, block)
        return _in_chunks(one, data, _FORWARD_CHUNK)

    def compute_view_1(self, X):
        """sheet snippets (n,1,H,W) -> (n, code_dim)"""
        return self._view(1, X, self.prepare_view_1)

    def compute_view_2(self, Z):
        """spectrogram excerpts (n,1,bins,frames) -> (n, code_dim)"""
        return self._view(2, Z, self.prepare_view_2)
