"""Oracle: one training update (NumPy, CPU).  TEST INFRASTRUCTURE - see oracle/__init__.py.

Restates what the reference's compiled `train` function does
(audio_sheet_retrieval/utils/train_dcca_pool.py:85-167: outputs with
deterministic=False :100-101, contrastive loss :106, CCALayer loss/corr :121-128,
L2 penalty over all trainable params :141-142, theano.grad :148, Adam :151),
i.e. forward with batch statistics, the pairwise ranking loss
(models/objectives.py:30-69), the gradient through LengthNormLayer, the
CCALayer train branch (layers/cca.py:91-182, including the four eigh's with
Theano's EighGrad), global pooling, BN (batch statistics), ELU,
max-pool, the flipped-filter convolutions, and lasagne.updates.adam (A.7).
Third-party semantics (Theano grad rules, Lasagne Adam) are "unverified
offline"; tests/test_oracle_train.py checks every gradient against torch
autograd of an independently written forward.

Max-pool ties (MaxPool2DLayer, models/mutopia_ccal_cont.py:79,83,87,91,104,108,
112,116): Theano's CPU MaxPoolGrad - the path north_star names -
adds the upstream gradient to EVERY element of a 2x2 window that equals the
window maximum (third-party semantic, unverified offline); that is `ties="all"`,
the default everywhere below.  `ties="first"` feeds only the first maximal
element in row-major order (a cuDNN-style rule; which element cuDNN really
picks is unverified offline too).  Ties are not rare: white regions of a sheet
give bit-identical activations (5.5 % of the sheet tower's block-2 windows of
synth_pairs(seed=23) hold at least two equal maxima).
"""
from __future__ import annotations

import numpy as np

from . import network as net

F32 = np.float32


# ---------------------------------------------------------------------------
# loss  (models/objectives.py:30-69, one direction, weight 1)
# ---------------------------------------------------------------------------
def contrastive_cos_loss(lv1, lv2, gamma=0.7, weight=1.0, symmetric=False):
    """get_contrastive_cos_loss(weight, gamma, symmetric) (models/objectives.py:30-69).  Returns loss and the gradients
    wrt lv1, lv2.  symmetric adds direction 2 (:53-65): the same hinge on D = lv2 lv1^T, i.e. with the views swapped."""
    def one_direction(a, b):
        dt = a.dtype.type
        n = a.shape[0]
        D = a.dot(b.T)                                       # :39
        d = np.diag(D).reshape(-1, 1)                        # :40
        L = dt(gamma) - d + D                                # :45-47 (off-diagonal entries only)
        off = ~np.eye(n, dtype=bool)
        Lc = np.clip(L, 0, 1000)                             # :48
        loss = Lc[off].mean(dtype=a.dtype)                   # :50
        # d clip / dx = 1 on the closed interval [0, 1000] (Theano Clip grad)
        G = ((L >= 0) & (L <= 1000) & off).astype(a.dtype) / dt(n * (n - 1))
        dD = G.copy()
        dD[np.arange(n), np.arange(n)] = -G.sum(axis=1)
        return loss, dD.dot(b), dD.T.dot(a)
    loss, g1, g2 = one_direction(lv1, lv2)
    if symmetric:
        loss2, h2, h1 = one_direction(lv2, lv1)
        loss, g1, g2 = loss + loss2, g1 + h1, g2 + h2
    w = lv1.dtype.type(weight)
    return w * loss, w * g1, w * g2                          # :67


def length_norm_bwd(x, dy):
    """y = x / ||x||  ->  dx = (dy - y (y.dy)) / ||x||   (layers/cca.py:39-40)."""
    nrm = np.sqrt((x * x).sum(axis=1, keepdims=True))
    y = x / nrm
    return (dy - y * (y * dy).sum(axis=1, keepdims=True)) / nrm


# ---------------------------------------------------------------------------
# eigh gradient (Theano EighGrad.perform)
# ---------------------------------------------------------------------------
def eigh_grad(w, v, gw, gv):
    """g = v (diag(gw) + K) v^T, K[n,m] = (v^T gv)[m,n] / (w[n] - w[m]) for m != n;
    returned as tril(g) + triu(g,1)^T exactly like Theano (eigh reads the lower
    triangle only)."""
    vtgv = v.T.dot(gv)                                   # [m, n] = v_m . gv_n
    dw = w[:, None] - w[None, :]                         # [n, m] = w_n - w_m
    with np.errstate(divide="ignore", invalid="ignore"):
        K = np.where(np.eye(len(w), dtype=bool), 0, vtgv.T / dw)      # K[n,m]
    g = v.dot(np.diag(gw) + K).dot(v.T)
    return np.tril(g) + np.triu(g, 1).T


def _inv_sqrt_fwd(S):
    d, A = np.linalg.eigh(S)
    w = np.reciprocal(np.sqrt(d))
    return (A * w).dot(A.T), (d, A, w)


def _inv_sqrt_bwd(cache, Q):
    """S^-1/2 = (A * w) A^T with (d, A) = eigh(S), w = d^-1/2 (layers/cca.py:144-147).
    Q: gradient wrt S^-1/2.  Returns the gradient wrt S (lower-triangular form)."""
    d, A, w = cache
    dA = (Q + Q.T).dot(A * w)
    dw = np.einsum("ik,ij,jk->k", A, Q, A)
    dd = dw * (-0.5) * w / d                             # d(d^-1/2)/dd = -1/2 d^-3/2
    return eigh_grad(d, A, dd.astype(d.dtype), dA.astype(d.dtype))


# ---------------------------------------------------------------------------
# CCALayer train branch: forward with cache, backward
# ---------------------------------------------------------------------------
def cca_train_fwd(H1, H2, cca_params, r=(1e-3, 1e-3, 1e-3), alpha=1.0):
    """Same arithmetic as oracle.network.cca_layer_train plus a cache for the
    backward pass.  Returns out1, out2, corr, new_cca_params, cache."""
    dt = H1.dtype.type
    U0r, V0r, m1_0, m2_0, S12_0, S11_0, S22_0 = [p.astype(H1.dtype) for p in cca_params]
    a, oma = dt(alpha), dt(1.0 - alpha)
    m = dt(H1.shape[0])
    mean1 = oma * m1_0 + a * H1.mean(axis=0, dtype=H1.dtype)
    mean2 = oma * m2_0 + a * H2.mean(axis=0, dtype=H1.dtype)
    Hb1, Hb2 = H1 - mean1, H2 - mean2
    eye = np.eye(H1.shape[1], dtype=H1.dtype)
    c = dt(1.0) / (m - 1)
    S12 = oma * S12_0 + a * (c * Hb1.T.dot(Hb2))
    S11 = oma * S11_0 + a * (c * Hb1.T.dot(Hb1) + dt(r[0]) * eye)
    S22 = oma * S22_0 + a * (c * Hb2.T.dot(Hb2) + dt(r[1]) * eye)
    S11si, c11 = _inv_sqrt_fwd(S11)
    S22si, c22 = _inv_sqrt_fwd(S22)
    T = S11si.dot(S12).dot(S22si)
    M1 = T.dot(T.T) + dt(r[2]) * eye
    M2 = T.T.dot(T) + dt(r[2]) * eye
    E1, E = np.linalg.eigh(M1)
    F1, Fm = np.linalg.eigh(M2)
    corr = np.sqrt(np.clip(E1, 1e-7, 1.0)).astype(H1.dtype)
    U0 = S11si.dot(E)
    V = S22si.dot(Fm)
    s = np.sign(U0.T.dot(S12).dot(V).diagonal()).astype(H1.dtype)
    U = U0 * s
    out1, out2 = Hb1.dot(U), Hb2.dot(V)
    new = [U, V, mean1, mean2, S12, S11, S22]
    cache = dict(Hb1=Hb1, Hb2=Hb2, S12=S12, S11si=S11si, S22si=S22si, c11=c11, c22=c22, T=T, E1=E1, E=E,
                 F1=F1, F=Fm, U=U, V=V, s=s, c=c, a=a)
    return out1, out2, corr, new, cache


def cca_train_bwd(cache, dout1, dout2):
    """Gradient of the CCALayer train branch wrt H1, H2 (everything is
    differentiated, including the running-average mixing with alpha; the sign
    vector s is piecewise constant)."""
    k = cache
    Hb1, Hb2, c, a = k["Hb1"], k["Hb2"], k["c"], k["a"]
    dU = Hb1.T.dot(dout1)
    dV = Hb2.T.dot(dout2)
    dHb1 = dout1.dot(k["U"].T)
    dHb2 = dout2.dot(k["V"].T)
    dU0 = dU * k["s"]
    dS11si = dU0.dot(k["E"].T)
    dE = k["S11si"].T.dot(dU0)
    dS22si = dV.dot(k["F"].T)
    dF = k["S22si"].T.dot(dV)
    zero = np.zeros_like(k["E1"])
    dM1 = eigh_grad(k["E1"], k["E"], zero, dE)            # corr feeds no loss term (wl = 0)
    dM2 = eigh_grad(k["F1"], k["F"], zero, dF)
    T = k["T"]
    dT = (dM1 + dM1.T).dot(T) + T.dot(dM2 + dM2.T)
    S12 = k["S12"]
    dS11si = dS11si + dT.dot((S12.dot(k["S22si"])).T)
    dS12 = k["S11si"].T.dot(dT).dot(k["S22si"].T)
    dS22si = dS22si + (k["S11si"].dot(S12)).T.dot(dT)
    dS11 = _inv_sqrt_bwd(k["c11"], dS11si)
    dS22 = _inv_sqrt_bwd(k["c22"], dS22si)
    ac = a * c
    dHb1 = dHb1 + ac * (Hb1.dot(dS11 + dS11.T) + Hb2.dot(dS12.T))
    dHb2 = dHb2 + ac * (Hb2.dot(dS22 + dS22.T) + Hb1.dot(dS12))
    # Hb = H - ((1-alpha) m0 + alpha mean(H))
    dH1 = dHb1 - a * dHb1.mean(axis=0, keepdims=True)
    dH2 = dHb2 - a * dHb2.mean(axis=0, keepdims=True)
    return dH1.astype(Hb1.dtype), dH2.astype(Hb1.dtype)


# ---------------------------------------------------------------------------
# tower backward
# ---------------------------------------------------------------------------
def conv_bwd_nhwc(x, W, dz):
    """Gradients of z = conv2d_flip_nhwc(x, W) wrt x and W."""
    co, ci, k, _ = W.shape
    p = (k - 1) // 2
    # dx = conv_flip(dz, W') with W'[i,o,a',b'] = W[o,i,k-1-a',k-1-b']
    Wt = np.ascontiguousarray(np.transpose(W[:, :, ::-1, ::-1], (1, 0, 2, 3)))
    dx = net.conv2d_flip_nhwc(np.ascontiguousarray(dz), Wt) if x.dtype == F32 else \
        net.conv2d_flip_nhwc_f64(dz, Wt)
    n, h, w, _ = x.shape
    xp = np.zeros((n, h + 2 * p, w + 2 * p, ci), x.dtype)
    xp[:, p:p + h, p:p + w, :] = x
    dzf = dz.reshape(-1, co)
    dW = np.zeros_like(W)
    for a in range(k):
        for b in range(k):
            xs = xp[:, 2 * p - a:2 * p - a + h, 2 * p - b:2 * p - b + w, :].reshape(-1, ci)
            dW[:, :, a, b] = dzf.T.dot(xs)
    return dx, dW


def bn_train_bwd(z, gamma, mu, inv_std, dy):
    """y = (z - mu) * (gamma * inv_std) + beta with batch mu / inv_std (A.2);
    returns dz, dbeta, dgamma."""
    c = z.shape[-1]
    zf, dyf = z.reshape(-1, c), dy.reshape(-1, c)
    xhat = (zf - mu) * inv_std
    dbeta = dyf.sum(axis=0, dtype=z.dtype)
    dgamma = (dyf * xhat).sum(axis=0, dtype=z.dtype)
    mcount = z.dtype.type(zf.shape[0])
    dz = (gamma * inv_std) * (dyf - dbeta / mcount - xhat * (dgamma / mcount))
    return dz.reshape(z.shape).astype(z.dtype), dbeta, dgamma


def _windows(a):
    """(n, h, w, c) -> (n, h/2, w/2, c, 4): the 2x2 windows, elements in row-major order (rr = 2 dy + dx)"""
    n, h, w, c = a.shape
    h2, w2 = h // 2, w // 2
    return a[:, :2 * h2, :2 * w2, :].reshape(n, h2, 2, w2, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, h2, w2, c, 4)


def maxpool2_routed_nhwc(a, route):
    """Pooling with an IMPOSED selection (what a device did).  route (n, h/2, w/2, c) in 0..3 names the window element
    that is passed on; a boolean route (n, h/2, w/2, c, 4) names the SET of elements the device found equal to the
    maximum - its first member is passed on.  Where the selection names the maximum this is MaxPool2DLayer; where two
    elements tie to within a rounding error it is the same function up to that error, with a definite gradient routing."""
    route = np.asarray(route)
    if route.dtype == bool:
        route = route.argmax(axis=-1)
    return np.take_along_axis(_windows(a), route[..., None], axis=-1)[..., 0]


def maxpool2_bwd_nhwc(a, dpooled, route=None, ties="all"):
    """Gradient of the 2x2 max-pool wrt its input.  ties="all": every element equal to the window maximum receives the
    pooled gradient (Theano's CPU MaxPoolGrad); ties="first": the first maximum in row-major order only.  route: an
    imposed selection instead of the comparison - an index array (one element per window) or a boolean set
    (n, h/2, w/2, c, 4) whose members all receive the gradient."""
    if ties not in ("all", "first"):
        raise ValueError("ties must be 'all' or 'first'")
    n, h, w, c = a.shape
    h2, w2 = h // 2, w // 2
    win = _windows(a)
    route = None if route is None else np.asarray(route)
    if route is not None and route.dtype == bool:
        g = np.where(route, dpooled[..., None], 0).astype(a.dtype)
    elif route is None and ties == "all":
        g = np.where(win == win.max(axis=-1, keepdims=True), dpooled[..., None], 0).astype(a.dtype)
    else:
        arg = win.argmax(axis=-1) if route is None else route   # first occurrence
        g = np.zeros(win.shape, a.dtype)
        np.put_along_axis(g, arg[..., None], dpooled[..., None], axis=-1)
    da = np.zeros_like(a)
    da[:, :2 * h2, :2 * w2, :] = g.reshape(n, h2, w2, c, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(n, 2 * h2, 2 * w2, c)
    return da


def pool_tie_share(a):
    """share of the 2x2 windows of a (n, h, w, c) that hold at least two / four equal maxima"""
    win = _windows(a)
    cnt = (win == win.max(axis=-1, keepdims=True)).sum(axis=-1)
    return float((cnt >= 2).mean()), float((cnt == 4).mean())


def elu_f32(y):
    """lasagne.nonlinearities.elu in float32 (what MaxPool2DLayer pools in the reference): y if y > 0 else expm1(y)"""
    y = np.asarray(y, F32)
    return np.where(y > 0, y, np.expm1(np.minimum(y, F32(0)))).astype(F32)


def elu_tie_deviation(y):
    """The device decides "equal to the window maximum" on y = BatchNorm output (float32), the reference on the float32
    ELU output it pools.  ELU is monotone, so the maximum is the same element; it is not injective in
    float32 for y < 0 (around y = -3 about five neighbouring floats share one image, below about -17 everything is
    -1.0f), so distinct y CAN tie in the reference and not on the device.  Returns, for (n, h, w, c) float32 y,
    (share of windows with >= 2 maxima decided on y, the same decided on elu_f32(y), share of windows where the two tie
    sets differ, largest ELU'(y_max) = exp(y_max) over those windows - the factor that scales the gradient the
    reference would additionally route there)."""
    y = np.asarray(y, F32)
    wy, wa = _windows(y), _windows(elu_f32(y))
    ty = wy == wy.max(axis=-1, keepdims=True)
    ta = wa == wa.max(axis=-1, keepdims=True)
    differ = (ty != ta).any(axis=-1)
    ymax = wy.max(axis=-1)
    worst = float(np.exp(np.minimum(ymax[differ], 0)).max()) if differ.any() else 0.0
    return float((ty.sum(-1) >= 2).mean()), float((ta.sum(-1) >= 2).mean()), float(differ.mean()), worst


def route_check(a, route):
    """How an imposed pooling selection relates to this evaluation's own activations a (n, h, w, c):
    gap   - the largest amount by which a selected element falls short of its window's maximum, relative to max |a|
            (a device that selected true maxima or rounding-level near-ties of them gives ~1e-7; a pooling bug that
            picks other elements gives O(1));
    flips - the share of windows whose imposed selection differs from the free one (index route: not the first maximum;
            boolean route: not exactly the set of elements equal to the maximum)."""
    win = _windows(a)
    wmax = win.max(axis=-1)
    route = np.asarray(route)
    if route.dtype == bool:
        sel_min = np.where(route, win, np.inf).min(axis=-1)          # every member of the set has to be (nearly) maximal
        flips = (route != (win == wmax[..., None])).any(axis=-1).mean()
    else:
        sel_min = np.take_along_axis(win, route[..., None], axis=-1)[..., 0]
        flips = (route != win.argmax(axis=-1)).mean()
    scale = float(np.abs(a).max()) or 1.0
    return float((wmax - sel_min).max() / scale), float(flips)


def tower_forward_train(x_nchw, tparams, routing=None, diag=None):
    """Train-mode tower forward keeping what the backward pass needs.  routing: {block index: (n, h/2, w/2, c) int
    array in 0..3, or boolean (n, h/2, w/2, c, 4) sets} imposes the pooling selection of those blocks
    (maxpool2_routed_nhwc); diag (a dict) then receives {block index: route_check(...)}."""
    dtype = tparams[0].dtype
    x = np.ascontiguousarray(np.transpose(x_nchw, (0, 2, 3, 1)), dtype=dtype)
    cache, stats = [], []
    for blk in range(9):
        W, beta, gamma = tparams[5 * blk:5 * blk + 3]
        z = net.conv2d_flip_nhwc(x, W) if dtype == F32 else net.conv2d_flip_nhwc_f64(x, W)
        zf = z.reshape(-1, z.shape[-1])
        mu = zf.mean(axis=0, dtype=dtype)
        var = ((zf - mu) ** 2).mean(axis=0, dtype=dtype)
        inv_std = (1.0 / np.sqrt(var + dtype.type(1e-4))).astype(dtype)
        y = (z - mu) * (gamma * inv_std) + beta
        a = np.where(y > 0, y, np.expm1(np.minimum(y, 0))).astype(dtype) if blk < 8 else y
        route = routing.get(blk) if routing else None
        if blk in (1, 3, 5, 7):
            pooled = net.maxpool2_nhwc(a) if route is None else maxpool2_routed_nhwc(a, route).astype(dtype)
            if route is not None and diag is not None:
                diag[blk] = route_check(a, route)
        else:
            pooled = a
        cache.append(dict(x=x, z=z, y=y, a=a, mu=mu, inv_std=inv_std, route=route))
        stats.append((mu, inv_std))
        x = pooled
    n, h, w, c = x.shape
    H = x.reshape(n, h * w, c).mean(axis=1, dtype=dtype)
    return H, stats, cache, (n, h, w, c)


def tower_backward(tparams, cache, last_shape, dH, ties="all"):
    """Gradients wrt [W, beta, gamma] of the nine blocks (list of 27 arrays).  ties: maxpool2_bwd_nhwc's rule."""
    n, h, w, c = last_shape
    dtype = dH.dtype
    dx = np.broadcast_to((dH / dtype.type(h * w))[:, None, None, :], (n, h, w, c)).astype(dtype)
    grads = [None] * 27
    for blk in range(8, -1, -1):
        W, beta, gamma = tparams[5 * blk:5 * blk + 3]
        k = cache[blk]
        da = maxpool2_bwd_nhwc(k["a"], dx, k.get("route"), ties) if blk in (1, 3, 5, 7) else dx
        if blk < 8:
            dy = da * np.where(k["y"] > 0, 1.0, np.exp(np.minimum(k["y"], 0))).astype(dtype)     # ELU'
        else:
            dy = da
        dz, dbeta, dgamma = bn_train_bwd(k["z"], gamma, k["mu"], k["inv_std"], dy)
        dx, dW = conv_bwd_nhwc(k["x"], W, dz)
        grads[3 * blk:3 * blk + 3] = [dW.astype(dtype), dbeta.astype(dtype), dgamma.astype(dtype)]
    return grads


# ---------------------------------------------------------------------------
# the compiled `train` function
# ---------------------------------------------------------------------------
TRAINABLE = [i for i in range(90) if i % 5 in (0, 1, 2)]      # W, beta, gamma of the 18 blocks


def loss_and_grads(x_prepared, z, params, gamma=0.7, l2=1e-5, r=(1e-3, 1e-3, 1e-3), alpha=1.0, routing=None, ties="all",
                   diag=None, weight=1.0, symmetric=False):
    """Returns loss (incl. the L2 penalty), corr, gradients for TRAINABLE (54
    arrays, same order), and the parameter list after the running-stat side
    effects (BN EMA, CCALayer values).  routing = (tower 1's, tower 2's) dicts for
    tower_forward_train: the pooling selection a device made, imposed on this evaluation.  ties: the max-pool gradient
    rule at windows with equal maxima (maxpool2_bwd_nhwc; "all" = Theano's CPU MaxPoolGrad).  diag: a dict that receives
    route_check's (gap, flips) per (tower, block) when a routing is imposed."""
    dtype = params[0].dtype
    d1, d2 = ({}, {}) if diag is not None else (None, None)
    H1, st1, c1, ls1 = tower_forward_train(x_prepared, params[0:45], routing[0] if routing else None, d1)
    H2, st2, c2, ls2 = tower_forward_train(z, params[45:90], routing[1] if routing else None, d2)
    if diag is not None:               # {(tower, block): (gap, flips)} of the imposed pooling selection (route_check)
        diag.update({(0, b): v for b, v in d1.items()})
        diag.update({(1, b): v for b, v in d2.items()})
    out1, out2, corr, new_cca, cca_cache = cca_train_fwd(H1, H2, params[90:97], r, alpha)
    nrm1 = np.sqrt((out1 * out1).sum(axis=1, keepdims=True))
    nrm2 = np.sqrt((out2 * out2).sum(axis=1, keepdims=True))
    lv1, lv2 = out1 / nrm1, out2 / nrm2
    loss, dlv1, dlv2 = contrastive_cos_loss(lv1, lv2, gamma, weight, symmetric)
    dout1 = length_norm_bwd(out1, dlv1)
    dout2 = length_norm_bwd(out2, dlv2)
    dH1, dH2 = cca_train_bwd(cca_cache, dout1, dout2)
    g1 = tower_backward(params[0:45], c1, ls1, dH1.astype(dtype), ties)
    g2 = tower_backward(params[45:90], c2, ls2, dH2.astype(dtype), ties)
    grads = g1 + g2
    # weight decay over all trainable params incl. BN beta/gamma (train_dcca_pool.py:141-142)
    pen = dtype.type(0)
    for gi, pi in enumerate(TRAINABLE):
        pen = pen + (params[pi] * params[pi]).sum(dtype=dtype)
        grads[gi] = grads[gi] + dtype.type(2.0 * l2) * params[pi]
    total = loss + dtype.type(l2) * pen
    newp = [p.copy() for p in params]
    one = dtype.type(1)
    for t, st in enumerate((st1, st2)):
        for blk, (mu, istd) in enumerate(st):
            i = 45 * t + 5 * blk
            newp[i + 3] = ((one - net.BN_ALPHA) * params[i + 3] + net.BN_ALPHA * mu).astype(dtype)
            newp[i + 4] = ((one - net.BN_ALPHA) * params[i + 4] + net.BN_ALPHA * istd).astype(dtype)
    newp[90:97] = [p.astype(dtype) for p in new_cca]
    return total, corr, grads, newp, (lv1, lv2)


def adam_init(params):
    """lasagne.updates.adam state: t = 0, m = v = 0 per trainable parameter."""
    return dict(t=0, m=[np.zeros_like(params[i]) for i in TRAINABLE],
                v=[np.zeros_like(params[i]) for i in TRAINABLE])


def adam_update(params, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """lasagne.updates.adam (A.7): t += 1; a_t = lr sqrt(1-b2^t)/(1-b1^t);
    m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= a_t m / (sqrt(v) + eps)."""
    dtype = params[0].dtype.type
    t = state["t"] + 1
    a_t = dtype(lr) * np.sqrt(dtype(1) - dtype(beta2) ** dtype(t)) / (dtype(1) - dtype(beta1) ** dtype(t))
    newp = [p.copy() for p in params]
    nm, nv = [], []
    for gi, pi in enumerate(TRAINABLE):
        g = grads[gi]
        # (one - beta) on float32 graph constants (floatX = float32): 1 - float32(0.999) = 0.00099998713, whatever
        # precision the rest of this evaluation runs in
        m = dtype(beta1) * state["m"][gi] + dtype(F32(1) - F32(beta1)) * g
        v = dtype(beta2) * state["v"][gi] + dtype(F32(1) - F32(beta2)) * g * g
        newp[pi] = (params[pi] - a_t * m / (np.sqrt(v) + dtype(eps))).astype(params[pi].dtype)
        nm.append(m.astype(params[pi].dtype))
        nv.append(v.astype(params[pi].dtype))
    return newp, dict(t=t, m=nm, v=nv)


def routing_from_selected(z, zsel):
    """The window element a device selected, from its raw conv output z (n, h, w, c) and the raw value of the selected
    element zsel (n, h/2, w/2, c): the first element of the window that carries that value (two elements with the same
    raw value have the same activation - the first wins on every implementation)."""
    win = _windows(np.asarray(z))
    hit = win == np.asarray(zsel)[..., None]
    if not hit.any(axis=-1).all():
        raise ValueError("zsel holds values that are not in their window")
    return hit.argmax(axis=-1)


def routing_from_tie_sets(bits):
    """The boolean route (n, h/2, w/2, c, 4) from a device's 4-bit sets (asr_debug_train_tensor kind 10: bit rr = 2 dy +
    dx is set when window element rr equals the window maximum on the device)."""
    b = np.asarray(bits).astype(np.int64)
    if b.min() < 1 or b.max() > 15:
        raise ValueError("tie sets must be non-empty 4-bit sets")
    return ((b[..., None] >> np.arange(4)) & 1).astype(bool)


def train_step(x_prepared, z, params, state, lr=0.002, gamma=0.7, l2=1e-5, r=(1e-3, 1e-3, 1e-3), alpha=1.0, routing=None,
               ties="all", weight=1.0, symmetric=False):
    """iter_funcs['train'](X1, X2) -> [loss, corr]  (+ the updated shared state)."""
    loss, corr, grads, newp, _ = loss_and_grads(x_prepared, z, params, gamma, l2, r, alpha, routing, ties,
                                                weight=weight, symmetric=symmetric)
    # the Adam update reads the OLD parameter values; BN/CCA default_updates apply on top
    upd, state = adam_update(params, grads, state, lr)
    for pi in TRAINABLE:
        newp[pi] = upd[pi]
    return loss, corr, newp, state


def valid_loss(x_prepared, z, params, gamma=0.7, weight=1.0, symmetric=False):
    """iter_funcs['valid'] (train_dcca_pool.py:155): deterministic outputs, ranking
    loss only (no L2)."""
    lv1, lv2 = net.compute_output(x_prepared, z, params)
    return contrastive_cos_loss(lv1, lv2, gamma, weight, symmetric)[0]
