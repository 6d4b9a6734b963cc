"""Seeded inputs of the reference pins at the sizes BASELINE.json quotes (reference_golden_sizes.npz holds only the
reference's OUTPUTS for them): shared by tests/golden/make_reference_golden.py and tests/test_reference_golden.py.
NumPy's legacy RandomState streams are frozen, so the arrays are identical wherever they are regenerated."""
import numpy as np

CCA_SIZES = {"cca_25000": 25000}                      # refine_cca.py --n_train 25000 (README.md:104-107)
EVAL_SIZES = {"eval_1000": (1000, 0.9), "eval_2000": (2000, 1.1)}      # configs[1]; eval_models.sh:15 (--n_test 2000)


def cca_inputs(tag):
    n = CCA_SIZES[tag]
    rng = np.random.RandomState(7000 + n)
    z = rng.standard_normal((n, 32))
    mix1, mix2 = rng.standard_normal((32, 32)), rng.standard_normal((32, 32))
    H1 = (z @ mix1 + 0.7 * rng.standard_normal((n, 32)) + 0.3).astype(np.float32)
    H2 = (z @ mix2 + 0.7 * rng.standard_normal((n, 32)) - 0.2).astype(np.float32)
    return H1, H2


def eval_inputs(tag):
    n, noise = EVAL_SIZES[tag]
    rng = np.random.RandomState(9000 + n)
    a = rng.standard_normal((n, 32))
    lv1 = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    b = lv1.astype(np.float64) + noise * rng.standard_normal((n, 32)) / np.sqrt(32.0)
    lv2 = (b / np.linalg.norm(b, axis=1, keepdims=True)).astype(np.float32)
    return lv1, lv2
