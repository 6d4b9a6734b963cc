"""Host-side stand-ins for the Lasagne/Theano objects the reference's drivers
touch, backed by one libasr_hip.so context (`_lib.Engine`).

The reference's seam is
    layers = model.build_model()                       # 4 layer handles
    lasagne.layers.set_all_param_values(layers, params)
    f = theano.function([l_view1.input_var, l_view2.input_var],
                        lasagne.layers.get_output(l_v1latent, deterministic=True))
    f(X1, X2) -> (n, 32) float32
This module provides the same verbs:
    set_all_param_values / get_all_param_values / get_all_layers / get_output /
    function
so that run_eval.py / refine_cca.py / retrieval_wrapper.py keep the reference's
control flow line by line.  There is no CPU path: compiling a function creates
the HIP context and raises if that fails.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .utils.param_layout import (IDX_MEAN1, IDX_MEAN2, IDX_S11, IDX_S12, IDX_S22, IDX_U, IDX_V, param_shapes)


class Network(object):
    """State shared by the layer handles of one build_model() call."""

    def __init__(self, model_name, input_shape_1, input_shape_2, raw_shape_1=None, device=None,
                 r1=1e-3, r2=1e-3, rT=1e-3, alpha=1.0, gamma=0.7, l2=1e-5):
        self.model_name = model_name
        self.input_shape_1 = tuple(input_shape_1)      # network-resolution (C,H,W) of view 1
        self.input_shape_2 = tuple(input_shape_2)
        self.raw_shape_1 = tuple(raw_shape_1 or input_shape_1)
        self.device = device
        self.hyper = dict(r1=r1, r2=r2, rT=rT, alpha=alpha, gamma=gamma, l2=l2)
        self.shapes = param_shapes(model_name)
        self._engine = None
        self._pending_params = None

    # the engine is created lazily: build_model() itself must work without a GPU
    # (the reference also builds the symbolic graph before touching a device)
    @property
    def engine(self):
        if self._engine is None:
            import os
            dev = self.device if self.device is not None else int(os.environ.get("ASR_DEVICE", "0"))
            self._engine = _lib.Engine(self.model_name, device=dev,
                                       h1=self.raw_shape_1[1], w1=self.raw_shape_1[2],
                                       h2=self.input_shape_2[1], w2=self.input_shape_2[2], **self.hyper)
            if self._pending_params is None:
                self.init_params()          # sets them on the engine as well
            else:
                self._engine.set_params(self._pending_params)
        return self._engine

    def set_params(self, params):
        params = [np.ascontiguousarray(p, dtype=np.float32) for p in params]
        if len(params) != len(self.shapes):
            raise ValueError("mismatch: parameter list has %d arrays, the network %d" % (len(params), len(self.shapes)))
        for i, (p, shp) in enumerate(zip(params, self.shapes)):
            if tuple(p.shape) != tuple(shp):
                raise ValueError("mismatch: parameter %d has shape %r, expected %r" % (i, p.shape, shp))
        self._pending_params = params
        if self._engine is not None:
            self._engine.set_params(params)

    def init_params(self):
        """Fresh parameters as Lasagne draws them at build time: W ~ HeUniform(gain 1) = U(+-sqrt(3/fan_in)) from
        NumPy's global RNG (the reference never seeds it: SURVEY A.1), BN beta 0 / gamma 1 / mean 0 / inv_std 1,
        CCALayer parameters 0 (layers/cca.py:48-50)."""
        params = []
        for i, shp in enumerate(self.shapes):
            if i < 90 and i % 5 == 0:
                lim = np.sqrt(3.0 / (shp[1] * shp[2] * shp[3]))
                params.append(np.random.uniform(-lim, lim, size=shp).astype(np.float32))
            elif i < 90 and i % 5 in (2, 4):
                params.append(np.ones(shp, np.float32))
            else:
                params.append(np.zeros(shp, np.float32))
        self.set_params(params)

    def get_params(self):
        if self._pending_params is None:
            self.init_params()
        if self._engine is not None and self._pending_params is not None:
            got = self._engine.get_params()
            return [g.reshape(s) for g, s in zip(got, self.shapes)]
        if self._pending_params is None:
            raise RuntimeError("network parameters have not been set")
        return [p.copy() for p in self._pending_params]

    def set_param(self, index, value):
        params = self.get_params()
        value = np.asarray(value, dtype=np.float32)
        if value.shape != tuple(self.shapes[index]):
            raise ValueError("parameter %d: shape %r, expected %r" % (index, value.shape, self.shapes[index]))
        params[index] = value
        self._pending_params = params
        if self._engine is not None:
            if index in (IDX_U, IDX_V, IDX_MEAN1, IDX_MEAN2):
                self._engine.set_cca(params[IDX_U], params[IDX_V], params[IDX_MEAN1], params[IDX_MEAN2])
            else:
                self._engine.set_params(params)


class SharedParam(object):
    """theano shared variable stand-in: get_value()/set_value() on one array of
    the flat parameter list (refine_cca.py:104-107)."""

    def __init__(self, net, index, name):
        self.net, self.index, self.name = net, index, name

    def get_value(self):
        return self.net.get_params()[self.index]

    def set_value(self, value):
        self.net.set_param(self.index, value)


class Layer(object):
    def __init__(self, net, kind, view=0, name=None):
        self.net, self.kind, self.view, self.name = net, kind, view, name
        self.input_layers = []

    @property
    def input_var(self):          # placeholder identifying the positional input
        return ("input", self.view)

    @property
    def output_shape(self):
        if self.kind == "input":
            shp = self.net.input_shape_1 if self.view == 1 else self.net.input_shape_2
            return (None,) + tuple(shp)
        if self.kind == "cca":
            return (None, 64)
        return (None, 32)


class CCALayer(Layer):
    """Handle of models/lasagne_extensions/layers/cca.py:CCALayer (:43-209):
    exposes the shared parameters the drivers overwrite and the two tower
    outputs feeding it (refine_cca.py:78-84)."""

    def __init__(self, net, in1, in2):
        super(CCALayer, self).__init__(net, "cca", 0, "CCALayer")
        self.input_layers = [in1, in2]
        self.U = SharedParam(net, IDX_U, "U")
        self.V = SharedParam(net, IDX_V, "V")
        self.mean1 = SharedParam(net, IDX_MEAN1, "mean1")
        self.mean2 = SharedParam(net, IDX_MEAN2, "mean2")
        self.S12 = SharedParam(net, IDX_S12, "S12")
        self.S11 = SharedParam(net, IDX_S11, "S11")
        self.S22 = SharedParam(net, IDX_S22, "S22")


def build_layers(net):
    """The four handles build_model() returns (models/mutopia_ccal_cont.py:145)."""
    l_view1 = Layer(net, "input", 1, "view1")
    l_view2 = Layer(net, "input", 2, "view2")
    f1 = Layer(net, "features", 1, "Flatten")
    f2 = Layer(net, "features", 2, "Flatten")
    f1.input_layers, f2.input_layers = [l_view1], [l_view2]
    cca = CCALayer(net, f1, f2)
    l_v1latent = Layer(net, "latent", 1, "LengthNorm")
    l_v2latent = Layer(net, "latent", 2, "LengthNorm")
    l_v1latent.input_layers = [cca]
    l_v2latent.input_layers = [cca]
    return l_view1, l_view2, l_v1latent, l_v2latent


# ---- lasagne.layers.* stand-ins -------------------------------------------------
def _net_of(layers):
    if isinstance(layers, Layer):
        return layers.net
    return layers[0].net


def set_all_param_values(layers, params):
    """lasagne.layers.set_all_param_values (run_eval.py:82, retrieval_wrapper.py:29)."""
    _net_of(layers).set_params(params)


def get_all_param_values(layers):
    """lasagne.layers.get_all_param_values (utils/train_dcca_pool.py:395, refine_cca.py:111)."""
    return _net_of(layers).get_params()


def get_all_layers(layer):
    """lasagne.layers.helper.get_all_layers: topological list of the handles
    reachable from `layer` (refine_cca.py:78)."""
    seen, order = set(), []

    def visit(l):
        if id(l) in seen:
            return
        seen.add(id(l))
        for p in l.input_layers:
            visit(p)
        order.append(l)
    visit(layer)
    return order


class _Output(object):
    def __init__(self, layer, deterministic):
        self.layer, self.deterministic = layer, deterministic


def get_output(layer, deterministic=False):
    """lasagne.layers.get_output(layer, deterministic=...)."""
    return _Output(layer, deterministic)


class CompiledFunction(object):
    """What `function()` returns: callable like the Theano function it stands for - one C-contiguous NCHW array per
    input placeholder, view 1 ALREADY prepared - plus `embed_raw`, the whole-array entry the drivers use when the
    preparation is the model's own `prepare` (which the first kernel evaluates itself)."""

    def __init__(self, net, layer, views):
        self.net, self.layer, self.views = net, layer, list(views)
        self.view = layer.view
        self.features = layer.kind == "features"
        self.pos = self.views.index(layer.view)
        self.engine = net.engine          # create the HIP context now ("compile time")

    def __call__(self, *arrays):
        if len(arrays) != len(self.views):
            raise TypeError("expected %d inputs, got %d" % (len(self.views), len(arrays)))
        x = arrays[self.pos]
        if self.view == 1:
            return self.engine.embed_view1(x, prepared=True, features=self.features)
        return self.engine.embed_view2(x, features=self.features)

    def embed_raw(self, data):
        """(n, 32) outputs for `data` = the UNPREPARED array of this function's view, any length, in one library call:
        view 1 as the pool / the servers hold it (uint8 or float 0..255, raw size) - model.prepare runs inside the
        first kernel (ASR_IN_U8_RAW / ASR_IN_F32_RAW); view 2 as is.  Row for row the same values as
        `self(model.prepare(chunk), ...)` over any chunking (deterministic mode: rows are independent)."""
        if self.view == 1:
            if data.dtype != np.uint8:
                data = np.ascontiguousarray(data, dtype=np.float32)      # prepare's x.astype(np.float32)
            return self.engine.embed_view1(data, prepared=False, features=self.features)
        return self.engine.embed_view2(data, features=self.features)


def is_fused_prepare(net, prepare):
    """True when `prepare` is the model's own preparation, which the library's first kernel evaluates (bit-identical,
    tests/test_gpu_dropin_api.py) - the drivers then hand over raw arrays instead of preparing them on the host."""
    return prepare is not None and getattr(prepare, "asr_fused_prepare", None) == net.model_name


def function(inputs, outputs):
    """theano.function(inputs, outputs) for the deterministic embedding graphs
    (run_eval.py:92-95, refine_cca.py:86-89, retrieval_wrapper.py:33-38).
    `inputs`: list of input_var placeholders; the compiled callable takes one
    NumPy array per placeholder (C-contiguous NCHW float32, view 1 already
    prepared) and returns a float32 (n, 32) array."""
    if not isinstance(outputs, _Output):
        raise TypeError("outputs must come from get_output()")
    if not outputs.deterministic:
        raise NotImplementedError("training graphs are built by utils.train_dcca_pool.create_iter_functions")
    layer = outputs.layer
    if layer.kind not in ("latent", "features"):
        raise ValueError("cannot compile an output for layer %r" % layer.name)
    views = [v[1] for v in inputs]
    if layer.view not in views:
        raise ValueError("the output depends on view %d which is not an input" % layer.view)
    return CompiledFunction(layer.net, layer, views)
