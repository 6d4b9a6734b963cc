"""The Winograd identities the HIP kernels hard-code, checked in float64 on the CPU against the direct convolution.

csrc/conv_wino_kernels.hip (F(2x2,3x3)) and csrc/conv_wino4_kernels.hip (F(4x4,3x3)) spell the transforms out as
factorised add / fma sequences (`in6`, `out6`, the B^T d B and A^T M A code) and the weight transform as G g G^T in
`wino_pack_kernel` / `wino4_pack_kernel`.  The same sequences are restated here line by line; if a constant or a sign
in the kernels' formulas were wrong, the GPU parity tests would catch it - this file says which identity they implement.
"""
import numpy as np


def _direct(d, g):
    """valid 3x3 correlation of a (n+2)x(n+2) patch -> n x n"""
    n = d.shape[0] - 2
    return np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(n)] for i in range(n)])


# ---- F(2x2, 3x3) -------------------------------------------------------------------------------------------------
def _in4(d0, d1, d2, d3):                       # B^T d, as in conv3x3_wino / conv3x3_winog
    return d0 - d2, d1 + d2, d2 - d1, d1 - d3


def _out4(m0, m1, m2, m3):                      # A^T m
    return (m0 + m1) + m2, (m1 - m2) - m3


def _g4(g):                                     # G g G^T, as in wino_pack_kernel
    t = np.stack([g[0], 0.5 * (g[0] + g[1] + g[2]), 0.5 * (g[0] - g[1] + g[2]), g[2]])          # (4, 3)
    return np.stack([t[:, 0], 0.5 * (t[:, 0] + t[:, 1] + t[:, 2]), 0.5 * (t[:, 0] - t[:, 1] + t[:, 2]), t[:, 2]], axis=1)


def test_f2x2_3x3_factorisation_equals_direct_convolution():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((4, 4)), rng.standard_normal((3, 3))
        cols = np.stack(_in4(*d), axis=0)                              # columns: B^T d   (rows indexed by xi)
        v = np.stack(_in4(*cols.T), axis=1)                            # rows:    (B^T d) B
        m = v * _g4(g)
        s = np.stack(_out4(*m), axis=0)                                # (2, 4)
        y = np.stack(_out4(*s.T), axis=1)                              # (2, 2)
        assert np.abs(y - _direct(d, g)).max() < 1e-12


# ---- F(4x4, 3x3) -------------------------------------------------------------------------------------------------
def _in6(d0, d1, d2, d3, d4, d5):               # `in6` of conv_wino4_kernels.hip
    r0 = 4 * d0 + (-5 * d2 + d4)
    t1, t2 = -4 * d2 + d4, -4 * d1 + d3
    t3, t4 = d4 - d2, d3 - d1
    r5 = 4 * d1 + (-5 * d3 + d5)
    return r0, t1 + t2, t1 - t2, 2 * t4 + t3, -2 * t4 + t3, r5


def _out6(m0, m1, m2, m3, m4, m5):              # `out6`
    s1, s2, s3, s4 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return (m0 + s1) + s3, 2 * s4 + s2, 4 * s3 + s1, 8 * s4 + s2 + m5


_G6 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])


def test_f4x4_3x3_factorisation_equals_direct_convolution():
    rng = np.random.default_rng(1)
    for _ in range(20):
        d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
        cols = np.stack(_in6(*d), axis=0)
        v = np.stack(_in6(*cols.T), axis=1)
        m = v * (_G6 @ g @ _G6.T)                                      # wino4_pack_kernel
        s = np.stack(_out6(*m), axis=0)                                # (4, 6)
        y = np.stack(_out6(*s.T), axis=1)                              # (4, 4)
        assert np.abs(y - _direct(d, g)).max() < 1e-11


def test_channel_order_of_the_transformed_weights_is_a_permutation():
    """k-steps 2t, 2t+1 of lane group g <-> channels 8t+2g, 8t+2g+1; a 4-channel remainder is one more k-step with
    channel 8*NB + g (wino_pack_kernel): every input channel must appear exactly once."""
    for cin in (12, 24, 48, 96):
        nb = cin // 8
        seen = []
        for k in range(cin):
            if k < nb * 8:
                t, w = k >> 3, k & 7
                g, ks = w >> 1, 2 * t + (w & 1)
            else:
                g, ks = k - nb * 8, 2 * nb
            seen.append(ks * 4 + g)
        assert sorted(seen) == list(range(cin))


def test_data_gradient_taps_are_the_unflipped_channel_swapped_filter():
    """dgrad mode of the pack kernels: dx[ci] = sum_co corr(dz[co], W[co][ci]) with the UN-flipped 3x3 filter, because the
    forward is a correlation with the flipped one (Lasagne convolves)."""
    rng = np.random.default_rng(2)
    W = rng.standard_normal((3, 2, 3, 3))                              # (co, ci, 3, 3), Lasagne convolution form
    x = rng.standard_normal((2, 7, 8))
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    y = np.zeros((3, 7, 8))
    for co in range(3):
        for ci in range(2):
            y[co] += _direct_same(xp[ci], W[co, ci, ::-1, ::-1])
    dz = rng.standard_normal(y.shape)
    # analytic gradient by finite differences of <y, dz>
    eps = 1e-6
    dx = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        xx = x.copy(); xx[idx] += eps
        xpp = np.pad(xx, ((0, 0), (1, 1), (1, 1)))
        yy = np.zeros_like(y)
        for co in range(3):
            for ci in range(2):
                yy[co] += _direct_same(xpp[ci], W[co, ci, ::-1, ::-1])
        dx[idx] = np.sum((yy - y) * dz) / eps
    dzp = np.pad(dz, ((0, 0), (1, 1), (1, 1)))
    got = np.zeros_like(x)
    for ci in range(2):
        for co in range(3):
            got[ci] += _direct_same(dzp[co], W[co, ci])                # un-flipped taps, roles of ci / co swapped
    assert np.abs(got - dx).max() < 1e-4


def _direct_same(padded, g):
    h, w = padded.shape[0] - 2, padded.shape[1] - 2
    return np.array([[np.sum(padded[i:i + 3, j:j + 3] * g) for j in range(w)] for i in range(h)])
