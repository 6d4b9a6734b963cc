"""Oracle: linear CCA re-estimation, method 'svd' (NumPy/SciPy, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows
  audio_sheet_retrieval/utils/cca.py:25-53      (fit, common part)
  audio_sheet_retrieval/utils/cca.py:199-211    (method == 'svd')
  audio_sheet_retrieval/utils/cca.py:432-444    (transform_V1 / transform_V2)
  audio_sheet_retrieval/refine_cca.py:100-107   (write-back cast to float32)
Third-party: numpy.linalg.svd / inv, scipy.linalg.sqrtm (LAPACK, float64).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import sqrtm


class CCA(object):
    def __init__(self, r1=1e-3, r2=1e-3, rT=1e-3, method="svd"):
        assert method == "svd", "only the branch the reference selects (refine_cca.py:100)"
        self.r1, self.r2, self.rT, self.method = r1, r2, rT, method
        self.m1 = self.m2 = self.U = self.V = None

    def fit(self, H1, H2, verbose=False):
        m = H1.shape[0]                                           # :28
        self.m1 = np.mean(H1, axis=0)                             # :31 (dtype of H1: float32)
        self.m2 = np.mean(H2, axis=0)                             # :32
        H1bar = (H1 - self.m1).T                                  # :35,39
        H2bar = (H2 - self.m2).T                                  # :36,40
        S12 = (1.0 / (m - 1)) * np.dot(H1bar, H2bar.T)            # :43 float32 gemm
        S11 = (1.0 / (m - 1)) * np.dot(H1bar, H1bar.T)            # :47
        S11 = S11 + self.r1 * np.identity(S11.shape[0])           # :48 -> float64
        S22 = (1.0 / (m - 1)) * np.dot(H2bar, H2bar.T)            # :51
        S22 = S22 + self.r2 * np.identity(S22.shape[0])           # :52
        S11i = np.linalg.inv(np.real(sqrtm(S11)))                 # :201
        S22i = np.linalg.inv(np.real(sqrtm(S22)))                 # :202
        Tnp = S11i.dot(S12).dot(S22i)                             # :204
        U, values, Vt = np.linalg.svd(Tnp)                        # :206
        self.U = S11i.dot(U)                                      # :210
        self.V = S22i.dot(Vt.T)                                   # :211
        return values                                             # :208,430

    def transform_V1(self, X):                                    # :432-439
        return np.dot(X - self.m1, self.U)

    def transform_V2(self, Y):                                    # :441-444
        return np.dot(Y - self.m2, self.V)


def fit_f32(H1, H2, r1=1e-3, r2=1e-3):
    """refine_cca.py:100-107: CCA('svd').fit then cast m1, m2, U, V to float32.
    Returns (U, V, m1, m2, coeffs)."""
    c = CCA(r1=r1, r2=r2, method="svd")
    coeffs = c.fit(H1, H2)
    return (c.U.astype(np.float32), c.V.astype(np.float32),
            c.m1.astype(np.float32), c.m2.astype(np.float32), coeffs)
