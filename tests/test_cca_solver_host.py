"""CPU: the 32x32 float64 algebra of CCA('svd') as the gfx950 kernel runs it (csrc/cca_solve.inl: one-sided Jacobi
for S^-1/2 and the SVD of T), built serially for the host with AddressSanitizer + UBSan (tests/cca_host_harness.cpp)
and compared with the NumPy / SciPy oracle (oracle/cca_np.py: utils/cca.py:199-211).  GPU sanitizers do not exist on the
pool; this is where the solver's indexing gets checked."""
import os
import subprocess

import numpy as np
import pytest
from scipy.linalg import sqrtm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("cca_host") / "cca_host_harness"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(ROOT, "tests", "cca_host_harness.cpp"), "-o", str(out)]
    subprocess.check_call(cmd)
    return str(out)


def _case(rng, n, cond):
    z = rng.standard_normal((n, 32))
    H1 = z @ rng.standard_normal((32, 32)) + cond * rng.standard_normal((n, 32))
    H2 = z @ rng.standard_normal((32, 32)) + cond * rng.standard_normal((n, 32))
    H1 -= H1.mean(0)
    H2 -= H2.mean(0)
    S11 = H1.T @ H1 / (n - 1) + 1e-3 * np.eye(32)
    S22 = H2.T @ H2 / (n - 1) + 1e-3 * np.eye(32)
    S12 = H1.T @ H2 / (n - 1)
    return S11, S22, S12


def test_host_build_of_the_jacobi_solver_under_sanitizers(harness, tmp_path):
    rng = np.random.default_rng(7)
    cases = [_case(rng, n, cond) for n, cond in ((400, 0.5), (64, 0.1), (5000, 2.0), (40, 1.0))]
    # a rank-deficient sample (n < 32): S11, S22 are r*I on a subspace - the regulariser keeps them positive definite
    cases.append(_case(rng, 20, 0.3))
    blob = np.concatenate([np.concatenate([m.ravel() for m in c]) for c in cases]).astype(np.float64)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    blob.tofile(fin)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    res = subprocess.run([harness, str(fin), str(fout)], capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-3000:]
    out = np.fromfile(fout, dtype=np.float64).reshape(len(cases), 3104)
    for (S11, S22, S12), row in zip(cases, out):
        U, V, coeffs, A = row[:1024].reshape(32, 32), row[1024:2048].reshape(32, 32), row[2048:2080], row[2080:].reshape(32, 32)
        S11i = np.linalg.inv(np.real(sqrtm(S11)))
        S22i = np.linalg.inv(np.real(sqrtm(S22)))
        assert np.abs(A - S11i).max() <= 1e-9 * max(1.0, np.abs(S11i).max())
        _, s, _ = np.linalg.svd(S11i @ S12 @ S22i)
        assert np.abs(coeffs - s).max() <= 1e-10
        assert (np.diff(coeffs) <= 1e-15).all()                         # descending
        # canonical directions: U^T S11 U = I, V^T S22 V = I, U^T S12 V = diag(s) (sign-free invariants)
        assert np.abs(U.T @ S11 @ U - np.eye(32)).max() <= 1e-9
        assert np.abs(V.T @ S22 @ V - np.eye(32)).max() <= 1e-9
        assert np.abs(U.T @ S12 @ V - np.diag(coeffs)).max() <= 1e-9
