"""Canonical correlation analysis - same class/attributes as the reference's
audio_sheet_retrieval/utils/cca.py (constructor :10-23, fit :25-53 + 'svd'
:199-211, transform* :432-444), with fit() evaluated by the HIP library
(asr_cca_fit).  Only method='svd' exists: it is the only branch the reference
ever selects (refine_cca.py:100, utils/train_dcca_pool.py:250)."""
from __future__ import annotations

import numpy as np


class CCA(object):
    """Cannonical correlation analysis"""

    def __init__(self, r1=1e-3, r2=1e-3, rT=1e-3, method="svd", engine=None):
        if method != "svd":
            raise NotImplementedError("CCA method %r: only 'svd' is on the retrieval hot path" % method)
        self.r1, self.r2, self.rT, self.method = r1, r2, rT, method
        self.m1 = self.m2 = self.U = self.V = None
        self._engine = engine

    def _get_engine(self):
        if self._engine is None:
            from .. import runtime
            self._engine = runtime.default_engine(r1=self.r1, r2=self.r2, rT=self.rT)
        return self._engine

    def fit(self, H1, H2, verbose=False):
        """Compute projections into correlation space; returns the canonical
        correlation coefficients (descending)."""
        eng = self._get_engine()
        U, V, m1, m2, coeffs = eng.cca_fit(H1, H2)
        self.U, self.V, self.m1, self.m2 = U, V, m1, m2
        if verbose:
            print("\nCorrelation-Coeffs:  ", np.around(coeffs, 3))
            print("Canonical-Correlation:", np.sum(coeffs) / H1.shape[1])
        return coeffs

    def transform(self, X):
        """Project data into cca space (:432-435)"""
        return np.dot(X - self.m1, self.U)

    def transform_V1(self, X):
        return self.transform(X)

    def transform_V2(self, Y):
        """(:441-444)"""
        return np.dot(Y - self.m2, self.V)
