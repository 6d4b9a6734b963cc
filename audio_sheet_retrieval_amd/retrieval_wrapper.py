"""RetrievalWrapper - the drop-in embedding API (reference:
audio_sheet_retrieval/retrieval_wrapper.py:12-77, same constructor, attributes
and methods)."""
from __future__ import print_function

import pickle

import numpy as np

from . import network
from .utils.batch_iterators import batch_compute2


def load_params(param_file):
    """Pickles written by the reference (python 2, cPickle protocol -1,
    utils/train_dcca_pool.py:399-401) need latin1; ours load either way."""
    with open(param_file, "rb") as fp:
        try:
            return pickle.load(fp)
        except UnicodeDecodeError:
            fp.seek(0)
            return pickle.load(fp, encoding="latin1")


class RetrievalWrapper(object):
    """ Wrapper for cross modality retrieval networks """

    def __init__(self, model, param_file, prepare_view_1=None, prepare_view_2=None):
        """ Constructor """
        self.prepare_view_1 = prepare_view_1
        self.prepare_view_2 = prepare_view_2
        self.code_dim = model.DIM_LATENT

        print("Building network ...")
        layers = model.build_model(show_model=False)

        print("Loading model parameters from:", param_file)
        params = load_params(param_file)
        network.set_all_param_values(layers, params)

        print("Compiling prediction functions ...")
        l_view1, l_view2, l_v1latent, l_v2latent = layers
        self.compute_v1_latent = network.function(inputs=[l_view1.input_var, l_view2.input_var],
                                                  outputs=network.get_output(l_v1latent, deterministic=True))
        self.compute_v2_latent = network.function(inputs=[l_view1.input_var, l_view2.input_var],
                                                  outputs=network.get_output(l_v2latent, deterministic=True))

        # dummy inputs for respective second view (:41-42); never evaluated here:
        # rows are independent in deterministic mode, so only the tower the
        # output depends on is run
        self.dummy_in_v1 = np.zeros(([1] + list(l_view1.output_shape[1:])), dtype=np.float32)
        self.dummy_in_v2 = np.zeros(([1] + list(l_view2.output_shape[1:])), dtype=np.float32)

        self.shape_view1 = l_view1.output_shape[1:]
        self.shape_view2 = l_view2.output_shape[1:]

    def compute_view_1(self, X):
        """ compute network output of view 1 (:47-61) """
        X = X.copy()
        dummy_in_v2 = np.repeat(self.dummy_in_v2, X.shape[0], axis=0)
        return batch_compute2(X, dummy_in_v2, self.compute_v1_latent,
                              batch_size=min(100, X.shape[0]),
                              prepare1=self.prepare_view_1)

    def compute_view_2(self, Z):
        """ compute network output of view 2 (:63-77) """
        Z = Z.copy()
        dummy_in_v1 = np.repeat(self.dummy_in_v1, Z.shape[0], axis=0)
        return batch_compute2(dummy_in_v1, Z, self.compute_v2_latent,
                              batch_size=min(100, Z.shape[0]),
                              prepare2=self.prepare_view_2)
