"""CPU: pin the oracle's CCA('svd') restatement (utils/cca.py:25-53,199-211)."""
import numpy as np

from oracle import cca_np


def _views(rng, n, d=32):
    z = rng.standard_normal((n, d))
    H1 = (z @ rng.standard_normal((d, d)) + 0.5 * rng.standard_normal((n, d)) + 3.0).astype(np.float32)
    H2 = (z @ rng.standard_normal((d, d)) + 0.5 * rng.standard_normal((n, d)) - 1.0).astype(np.float32)
    return H1, H2


def test_cca_fit_properties():
    rng = np.random.default_rng(0)
    H1, H2 = _views(rng, 5000)
    c = cca_np.CCA(method="svd")
    coeffs = c.fit(H1, H2)
    assert coeffs.shape == (32,) and np.all(np.diff(coeffs) <= 1e-12) and coeffs[0] <= 1.0 + 1e-9
    # independent float64 derivation of the same quantities
    A, B = H1.astype(np.float64), H2.astype(np.float64)
    A -= A.mean(0)
    B -= B.mean(0)
    S11 = A.T @ A / (len(A) - 1) + 1e-3 * np.eye(32)
    S22 = B.T @ B / (len(B) - 1) + 1e-3 * np.eye(32)
    S12 = A.T @ B / (len(A) - 1)
    assert np.abs(c.U.T @ S11 @ c.U - np.eye(32)).max() < 1e-3
    assert np.abs(c.V.T @ S22 @ c.V - np.eye(32)).max() < 1e-3
    assert np.allclose(np.diag(c.U.T @ S12 @ c.V), coeffs, atol=1e-3)
    w, Q = np.linalg.eigh(S11)
    S11i = (Q / np.sqrt(w)) @ Q.T
    w, Q = np.linalg.eigh(S22)
    S22i = (Q / np.sqrt(w)) @ Q.T
    s = np.linalg.svd(S11i @ S12 @ S22i, compute_uv=False)
    assert np.allclose(s, coeffs, atol=1e-4)
    # transform (:432-444)
    t1, t2 = c.transform_V1(H1), c.transform_V2(H2)
    assert t1.shape == (5000, 32)
    cc = [np.corrcoef(t1[:, i], t2[:, i])[0, 1] for i in range(32)]
    assert np.allclose(cc, coeffs, atol=2e-2)


def test_fit_f32_casts():
    rng = np.random.default_rng(1)
    H1, H2 = _views(rng, 300)
    U, V, m1, m2, coeffs = cca_np.fit_f32(H1, H2)
    assert U.dtype == V.dtype == m1.dtype == m2.dtype == np.float32
    assert np.allclose(m1, H1.mean(0), atol=1e-4)
