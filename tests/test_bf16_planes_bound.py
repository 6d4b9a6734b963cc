"""The error bound the two-plane bf16 filter rests on (csrc/tail_rank_kernels.hip: TF_EPS_BF2, RF_BAND_BF2), checked on
the CPU with an exact emulation of the split: x = x1 + x2 + x3, x1 = bf16(x), x2 = bf16(x - x1) (round to nearest even,
differences exact in float32), and the filter keeps  x2.y1 + x1.y2 + x1.y1.  What it drops is bounded by
3.02 * 2^-16 * sum |x_i y_i| <= 4.6e-5 for unit-length operands; with the 3e-6 of the fp32 accumulation the kernels use
EPS = 5.5e-5.  The test drives the bound with the vectors that come closest to it (every element just below a bf16
rounding boundary, equal signs so that nothing cancels) and with random ones."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    return (((u + r) >> 16) << 16).astype(np.uint32).view(np.float32)


def planes(x):
    x = np.asarray(x, np.float32)
    x1 = bf16_rne(x)
    r = (x - x1).astype(np.float32)
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - x1.astype(np.float64))     # exact
    x2 = bf16_rne(r)
    x3 = (r - x2).astype(np.float32)
    assert np.array_equal(x3.astype(np.float64), r.astype(np.float64) - x2.astype(np.float64))
    return x1, x2, x3


def two_plane_dot(x, y):
    x1, x2, _ = planes(x)
    y1, y2, _ = planes(y)
    f = np.float64
    return (x2.astype(f) * y1.astype(f) + x1.astype(f) * y2.astype(f) + x1.astype(f) * y1.astype(f)).sum(-1)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


BOUND = 3.02 * 2.0 ** -16          # x sum |x_i y_i| (<= 1 for unit-length operands)


def test_plane_sizes():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    x1, x2, x3 = planes(x)
    ax = np.abs(x.astype(np.float64))
    assert np.all(np.abs(x2.astype(np.float64)) <= 2.0 ** -8 * ax * (1 + 1e-12))
    assert np.all(np.abs(x3.astype(np.float64)) <= 2.0 ** -16 * ax * (1 + 1e-12))
    assert np.array_equal(x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64), x.astype(np.float64))


def test_two_plane_dot_stays_inside_the_bound():
    rng = np.random.default_rng(1)
    worst = 0.0
    cases = []
    # random unit vectors, near neighbours, one sign (sum |x_i y_i| = 1: nothing cancels)
    a = unit(rng.standard_normal((20000, 32)))
    b = unit(rng.standard_normal((20000, 32)))
    cases += [(a, b), (a, unit(a + 0.05 * rng.standard_normal(a.shape))), (np.abs(a), np.abs(b))]
    # the adversarial family: every element m * 2^e with the mantissa just below / above a bf16 rounding boundary
    # (x2 at its largest, 2^-8 |x|), and x2's own mantissa at ITS boundary (x3 at its largest), all signs equal
    for frac in (2.0 ** -8 - 2.0 ** -17, 2.0 ** -8 - 2.0 ** -23, 2.0 ** -9 + 2.0 ** -17, 2.0 ** -8 - 2.0 ** -16 - 2.0 ** -23):
        base = (1.0 + frac) * np.ones(32)
        v = unit(base[None, :] * np.exp2(rng.integers(-2, 1, (2000, 32))))
        w = unit(base[None, :] * np.exp2(rng.integers(-2, 1, (2000, 32))))
        cases.append((v, w))
        cases.append((v, v))
    for x, y in cases:
        exact = (x.astype(np.float64) * y.astype(np.float64)).sum(-1)
        err = np.abs(two_plane_dot(x, y) - exact)
        lim = BOUND * (np.abs(x.astype(np.float64)) * np.abs(y.astype(np.float64))).sum(-1)
        assert np.all(err <= lim * (1 + 1e-9)), float((err / lim).max())
        worst = max(worst, float(err.max()))
    # the bound is not vacuous (the adversarial family gets within a factor of a few of it) and sits below the kernels' EPS
    assert 5e-6 < worst <= 4.61e-5 < 5.5e-5 - 3e-6
