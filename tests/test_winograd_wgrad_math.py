"""CPU: the algebra behind `wgrad_wino_kernel` (csrc/train_bwd_kernels.hip) - the weight gradient of a 3x3 filter as
Winograd F(3x3, 2x2).

The kernel accumulates, per transform position, products of B^T X B (4x4 input patch) and G D G^T (2x2 block of dz) over
all blocks and images, and applies A^T . A once at the end.  Here the same matrices are checked in float64 against the
definition the packed-taps kernel and the oracle use (oracle/train.py: dWc[a][b] = sum_pix x[y+a-1][x+b-1] dz[y][x],
correlation form), including the zero padding at the map border and an odd map width (the kernel pads dz and x with
zeros, which contribute nothing)."""
import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)      # forward F(2x2,3x3) input transform
G = np.array([[1, 0], [1, 1], [1, -1], [0, -1]], np.float64)                                  # dz transform
AT = np.array([[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]], np.float64)                  # back to the 3x3 gradient


def direct(x, dz):
    H, W = dz.shape
    xp = np.pad(x, 1)
    return np.array([[(xp[a:a + H, b:b + W] * dz).sum() for b in range(3)] for a in range(3)])


def winograd(x, dz):
    H, W = dz.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2                 # whole 2x2 blocks: zero rows / columns past the map
    dzp = np.zeros((Hp, Wp)); dzp[:H, :W] = dz
    xp = np.zeros((Hp + 2, Wp + 2)); xp[1:H + 1, 1:W + 1] = x   # halo of one pixel, zeros outside the map
    S = np.zeros((4, 4))
    for i in range(0, Hp, 2):
        for j in range(0, Wp, 2):
            X = xp[i:i + 4, j:j + 4]                             # rows 2i-1 .. 2i+2 of the map
            D = dzp[i:i + 2, j:j + 2]
            S += (BT @ X @ BT.T) * (G @ D @ G.T)                 # what the MFMAs accumulate, position by position
    return AT @ S @ AT.T


def test_one_block_is_the_f32_exchange_of_roles():
    rng = np.random.RandomState(0)
    X, D = rng.randn(4, 4), rng.randn(2, 2)
    want = np.array([[sum(X[a + p, b + q] * D[p, q] for p in range(2) for q in range(2)) for b in range(3)] for a in range(3)])
    got = AT @ ((BT @ X @ BT.T) * (G @ D @ G.T)) @ AT.T
    assert np.abs(got - want).max() < 1e-13


def test_whole_maps_with_borders_and_odd_sizes():
    rng = np.random.RandomState(1)
    for H, W in ((8, 8), (20, 25), (11, 5), (10, 12), (2, 2), (3, 7)):
        x, dz = rng.randn(H, W), rng.randn(H, W)
        assert np.abs(winograd(x, dz) - direct(x, dz)).max() < 1e-11 * H * W, (H, W)


def test_sixteen_products_instead_of_thirty_six():
    # per 2x2 block and channel pair: 16 multiplications in the transform domain against 4 pixels x 9 taps
    assert BT.shape[0] * BT.shape[0] == 16 and 4 * 9 == 36
