"""Process-wide default HIP context for the entry points whose reference
signatures carry no network handle (eval_retrieval(lv1, lv2), CCA().fit(...)):
the reference calls SciPy/NumPy there; here they run on the GPU through a small
context that allocates no activation workspace."""
from __future__ import annotations

import os

from . import _lib

_DEFAULT = {}


def default_engine(r1=1e-3, r2=1e-3, rT=1e-3, device=None):
    dev = int(os.environ.get("ASR_DEVICE", "0")) if device is None else device
    key = (dev, float(r1), float(r2), float(rT))
    if key not in _DEFAULT:
        _DEFAULT[key] = _lib.Engine("mutopia_ccal_cont", device=dev, r1=r1, r2=r2, rT=rT)
    return _DEFAULT[key]


def shutdown():
    for e in _DEFAULT.values():
        e.close()
    _DEFAULT.clear()
