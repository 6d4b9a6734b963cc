"""Shared body of the two model modules (the reference keeps two copies that
differ in four lines: models/mutopia_ccal_cont.py vs mutopia_ccal_cont_rsz.py)."""
from __future__ import annotations

import numpy as np

from .. import network


def make_build_model(model_name, input_shape_1, input_shape_2, raw_shape_1, r1, r2, rT, alpha, gamma, l2):
    def build_model(show_model=False, device=None):
        """Compile net architecture (models/mutopia_ccal_cont.py:64-145): returns
        (l_view1, l_view2, l_v1latent, l_v2latent)."""
        net = network.Network(model_name, input_shape_1, input_shape_2, raw_shape_1=raw_shape_1, device=device,
                              r1=r1, r2=r2, rT=rT, alpha=alpha, gamma=gamma, l2=l2)
        layers = network.build_layers(net)
        if show_model:
            print_architecture(net)
        return layers
    return build_model


def print_architecture(net):
    """utils/monitoring.py:print_architecture stand-in: one line per block."""
    from ..utils.param_layout import NUM_FILTERS, tower_channels
    nf = NUM_FILTERS[net.model_name]
    for view, shp in ((1, net.input_shape_1), (2, net.input_shape_2)):
        h, w = shp[1], shp[2]
        print("view %d: input 1x%dx%d" % (view, h, w))
        for b, (ci, co, k) in enumerate(tower_channels(nf)):
            pool = b in (1, 3, 5, 7)
            print("  conv%d %dx%d %3d->%3d + BN%s%s  -> %dx%d" % (
                b + 1, k, k, ci, co, " + ELU" if b < 8 else "", " + MaxPool2" if pool else "",
                h // 2 if pool else h, w // 2 if pool else w))
            if pool:
                h, w = h // 2, w // 2
        print("  GlobalPool -> 32 -> CCALayer -> LengthNorm")


def prepare_plain(x, y=None):
    """models/mutopia_ccal_cont.py:170-190: sheet snippet to float32 / 255."""
    x = x.astype(np.float32)
    x /= 255
    return x if y is None else (x, y)


prepare_plain.asr_fused_prepare = "mutopia_ccal_cont"      # conv1's ASR_IN_*_RAW input modes evaluate exactly this


def prepare_rsz(x, y=None):
    """models/mutopia_ccal_cont_rsz.py:170-190: /255, then cv2.resize to half
    size (bilinear; for an exact factor 2 this is the 2x2 box mean, evaluated
    as horizontal then vertical (.5,.5) blends)."""
    x = x.astype(np.float32)
    x /= 255
    h2, w2 = x.shape[2] // 2, x.shape[3] // 2
    x = x[:, :, :2 * h2, :2 * w2]
    xh = x[:, :, :, 0::2] * np.float32(0.5) + x[:, :, :, 1::2] * np.float32(0.5)
    x = np.ascontiguousarray(xh[:, :, 0::2, :] * np.float32(0.5) + xh[:, :, 1::2, :] * np.float32(0.5))
    return x if y is None else (x, y)


prepare_rsz.asr_fused_prepare = "mutopia_ccal_cont_rsz"
