"""Oracle: twin-CNN forward, CCALayer, length-norm (NumPy float32, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Follows
  audio_sheet_retrieval/models/mutopia_ccal_cont.py:54-149      (architecture)
  audio_sheet_retrieval/models/mutopia_ccal_cont_rsz.py:68,77,179-185 (rsz)
  audio_sheet_retrieval/models/lasagne_extensions/layers/cca.py:29-40,82-203
plus the Lasagne 0.2.dev1 / Theano 1.0.1 layer semantics listed in
SURVEY.md Appendix A ("third-party semantic, unverified offline").

Public API works on NCHW float32 arrays like the reference's compiled
functions; internally the towers run NHWC (pure layout choice, no arithmetic
difference besides float32 summation order inside the BLAS calls).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32

BN_EPS = F32(1e-4)      # lasagne.layers.batch_norm default epsilon (A.2)
BN_ALPHA = F32(0.1)     # lasagne BatchNormLayer default alpha (A.2)
DIM_LATENT = 32         # mutopia_ccal_cont.py:35

#: model variants: name -> (num_filters_1, resize_view1, declared INPUT_SHAPE_1)
VARIANTS = {
    # mutopia_ccal_cont.py:32,74  (declares 120x200, is fed 160x200: SURVEY A.10)
    "mutopia_ccal_cont": dict(nf=12, rsz=False, in1=(1, 160, 200), in2=(1, 92, 42)),
    # mutopia_ccal_cont_rsz.py:32,68,77
    "mutopia_ccal_cont_rsz": dict(nf=24, rsz=True, in1=(1, 160, 200), in2=(1, 92, 42)),
}


# --------------------------------------------------------------------------
# parameter layout: lasagne.layers.get_all_param_values
# --------------------------------------------------------------------------
def tower_channels(nf: int):
    """(C_in, C_out, ksize) of the nine conv blocks of one tower
    (mutopia_ccal_cont.py:76-94)."""
    return [(1, nf, 3), (nf, nf, 3),
            (nf, 2 * nf, 3), (2 * nf, 2 * nf, 3),
            (2 * nf, 4 * nf, 3), (4 * nf, 4 * nf, 3),
            (4 * nf, 4 * nf, 3), (4 * nf, 4 * nf, 3),
            (4 * nf, DIM_LATENT, 1)]


def param_shapes(variant: str):
    """Shapes of the 97-array flat parameter list, in the reference's order:
    tower-1 blocks 1..9 [W, beta, gamma, mean, inv_std], tower-2 likewise,
    CCALayer [U, V, mean1, mean2, S12, S11, S22] (cca.py:69-77)."""
    nf = VARIANTS[variant]["nf"]
    shapes = []
    for _tower in range(2):
        for ci, co, k in tower_channels(nf):
            shapes += [(co, ci, k, k), (co,), (co,), (co,), (co,)]
    d = DIM_LATENT
    shapes += [(d, d), (d, d), (d,), (d,), (d, d), (d, d), (d, d)]
    return shapes


def default_params(variant: str, rng: np.random.Generator):
    """Fresh parameters as Lasagne would draw them: W ~ HeUniform(gain=1)
    = U(+-sqrt(3/fan_in)) (A.1); beta 0, gamma 1, mean 0, inv_std 1 (A.2);
    CCALayer params all zero (cca.py:48-50 init.Constant(0))."""
    out = []
    shapes = param_shapes(variant)
    for i, shp in enumerate(shapes[:90]):
        kind = i % 5
        if kind == 0:
            fan_in = shp[1] * shp[2] * shp[3]
            lim = np.sqrt(3.0 / fan_in)
            out.append(rng.uniform(-lim, lim, size=shp).astype(F32))
        elif kind in (1, 3):
            out.append(np.zeros(shp, F32))
        else:
            out.append(np.ones(shp, F32))
    for shp in shapes[90:]:
        out.append(np.zeros(shp, F32))
    return out


# --------------------------------------------------------------------------
# prepare  (mutopia_ccal_cont.py:170-190, _rsz.py:170-190)
# --------------------------------------------------------------------------
def prepare(x, variant: str):
    """x: (B,1,H,W) any dtype holding 0..255 -> float32 in 0..1; the rsz
    variant halves H and W with cv2.resize(bilinear), which for an exact
    factor-2 downscale is the 2x2 box mean."""
    x = x.astype(F32)
    x = x / F32(255)
    if VARIANTS[variant]["rsz"]:
        b, c, h, w = x.shape
        h2, w2 = h // 2, w // 2
        x = x[:, :, :2 * h2, :2 * w2]
        # cv2 INTER_LINEAR at scale 2: weights (.5,.5) horizontally, then vertically
        xh = x[:, :, :, 0::2] * F32(0.5) + x[:, :, :, 1::2] * F32(0.5)
        x = xh[:, :, 0::2, :] * F32(0.5) + xh[:, :, 1::2, :] * F32(0.5)
    return np.ascontiguousarray(x, dtype=F32)


# --------------------------------------------------------------------------
# layers (NHWC internal)
# --------------------------------------------------------------------------
def conv2d_flip_nhwc_numpy(x, W):
    """Lasagne Conv2DLayer(flip_filters=True, pad=(k-1)/2, stride 1, no bias)
    (A.1): y[n,h,w,o] = sum_{i,a,b} W[o,i,a,b] * x[n, h+p-a, w+p-b, i].
    x: (N,H,W,I) f32, W: (O,I,k,k) f32 -> (N,H,W,O) f32."""
    n, h, w, ci = x.shape
    co, ci2, k, _ = W.shape
    assert ci == ci2
    p = (k - 1) // 2
    if k == 1:
        return (x.reshape(-1, ci) @ W[:, :, 0, 0].T).reshape(n, h, w, co)
    xp = np.zeros((n, h + 2 * p, w + 2 * p, ci), x.dtype)
    xp[:, p:p + h, p:p + w, :] = x
    y = np.zeros((n * h * w, co), x.dtype)
    for a in range(k):
        for b in range(k):
            # tap (a,b) of the flipped kernel reads x[h + p - a, w + p - b]
            # = xp[h + 2p - a, w + 2p - b]
            xs = xp[:, 2 * p - a:2 * p - a + h, 2 * p - b:2 * p - b + w, :]
            y += np.ascontiguousarray(xs).reshape(-1, ci) @ np.ascontiguousarray(W[:, :, a, b].T)
    return y.reshape(n, h, w, co)


_CONV_LIB = None


def build_conv_lib():
    """gcc recipe for oracle/conv_ref.c -> oracle/_build/libconv_ref.so
    (also called by __graft_entry__.build())."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_build", "libconv_ref.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", so,
                           os.path.join(here, "conv_ref.c"), os.path.join(here, "cdist_ref.c"), "-lm"])
    return so


def _conv_lib():
    """ctypes handle of oracle/conv_ref.c (built on demand with gcc)."""
    global _CONV_LIB
    if _CONV_LIB is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libconv_ref.so")
        srcs = [os.path.join(here, "conv_ref.c"), os.path.join(here, "cdist_ref.c")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs):
            build_conv_lib()
        lib = ctypes.CDLL(so)
        fp = ctypes.POINTER(ctypes.c_float)
        lib.conv2d_corr_nhwc_f32.argtypes = [fp, fp, fp] + [ctypes.c_int] * 6
        lib.conv2d_corr_nhwc_f32.restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        lib.conv2d_corr_nhwc_f64.argtypes = [dp, dp, dp] + [ctypes.c_int] * 6
        lib.conv2d_corr_nhwc_f64.restype = None
        i64 = ctypes.c_int64
        lib.row_norms_f64.argtypes = [fp, i64, i64, ctypes.c_int, dp]
        lib.row_norms_f64.restype = None
        lib.cdist_cosine_f64.argtypes = [fp, i64, fp, i64, ctypes.c_int, dp, dp, dp]
        lib.cdist_cosine_f64.restype = None
        lib.elu_f32.argtypes = [fp, ctypes.c_size_t]
        lib.elu_f32.restype = None
        lib.bn_det_nhwc_f32.argtypes = [fp] * 6 + [ctypes.c_size_t, ctypes.c_int]
        lib.bn_det_nhwc_f32.restype = None
        _CONV_LIB = lib
    return _CONV_LIB


def conv2d_flip_nhwc(x, W):
    """Same sum as conv2d_flip_nhwc_numpy, evaluated by oracle/conv_ref.c
    (float32, accumulation order a', b', i)."""
    import ctypes
    n, h, w, ci = x.shape
    co, ci2, k, _ = W.shape
    assert ci == ci2
    x = np.ascontiguousarray(x, dtype=F32)
    # correlation-form taps wt[a'][b'][i][o] = W[o][i][k-1-a'][k-1-b']
    wt = np.ascontiguousarray(np.transpose(W[:, :, ::-1, ::-1], (2, 3, 1, 0)), dtype=F32)
    y = np.empty((n, h, w, co), F32)
    fp = ctypes.POINTER(ctypes.c_float)
    _conv_lib().conv2d_corr_nhwc_f32(x.ctypes.data_as(fp), wt.ctypes.data_as(fp),
                                     y.ctypes.data_as(fp), n, h, w, ci, co, k)
    return y


def conv2d_flip_nhwc_f64(x, W):
    """conv2d_flip_nhwc_numpy's sum in float64, evaluated by oracle/conv_ref.c:conv2d_corr_nhwc_f64 (the float64
    oracle of the training step)."""
    import ctypes
    n, h, w, ci = x.shape
    co, ci2, k, _ = W.shape
    assert ci == ci2 and co <= 128
    x = np.ascontiguousarray(x, dtype=np.float64)
    wt = np.ascontiguousarray(np.transpose(W[:, :, ::-1, ::-1], (2, 3, 1, 0)), dtype=np.float64)
    y = np.empty((n, h, w, co), np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    _conv_lib().conv2d_corr_nhwc_f64(x.ctypes.data_as(dp), wt.ctypes.data_as(dp), y.ctypes.data_as(dp),
                                     n, h, w, ci, co, k)
    return y


def batchnorm_det_nhwc_numpy(x, beta, gamma, mean, inv_std):
    """BatchNormLayer deterministic (A.2): (x-mean)*(gamma*inv_std)+beta."""
    return (x - mean) * (gamma * inv_std) + beta


def batchnorm_det_nhwc(x, beta, gamma, mean, inv_std):
    """Same expression evaluated by oracle/conv_ref.c:bn_det_nhwc_f32."""
    import ctypes
    fp = ctypes.POINTER(ctypes.c_float)
    x = np.ascontiguousarray(x, dtype=F32)
    c = x.shape[-1]
    args = [np.ascontiguousarray(a, dtype=F32) for a in (beta, gamma, mean, inv_std)]
    y = np.empty_like(x)
    _conv_lib().bn_det_nhwc_f32(x.ctypes.data_as(fp), *[a.ctypes.data_as(fp) for a in args],
                                y.ctypes.data_as(fp), x.size // c, c)
    return y


def batchnorm_train_nhwc(x, beta, gamma):
    """BatchNormLayer training branch (A.2): batch mean, biased variance over
    (N,H,W); returns y, batch mean, batch inv_std."""
    xm = x.reshape(-1, x.shape[-1])
    mu = xm.mean(axis=0, dtype=F32)
    var = ((xm - mu) ** 2).mean(axis=0, dtype=F32)
    inv_std = (F32(1) / np.sqrt(var + BN_EPS)).astype(F32)
    y = (x - mu) * (gamma * inv_std) + beta
    return y, mu, inv_std


def elu_numpy(x):
    """lasagne.nonlinearities.elu: switch(x > 0, x, expm1(x)) (A.3)."""
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0))).astype(F32)


def elu(x):
    """Same function evaluated by oracle/conv_ref.c:elu_f32 (expm1f)."""
    import ctypes
    y = np.array(x, dtype=F32, order="C", copy=True)
    _conv_lib().elu_f32(y.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), y.size)
    return y


def maxpool2_nhwc(x):
    """MaxPool2DLayer(pool_size=2): stride 2, ignore_border=True -> floor (A.3)."""
    n, h, w, c = x.shape
    h2, w2 = h // 2, w // 2
    x = x[:, :2 * h2, :2 * w2, :].reshape(n, h2, 2, w2, 2, c)
    return x.max(axis=(2, 4))


def tower_forward(x_nchw, tparams, deterministic=True, return_cache=False):
    """One tower (mutopia_ccal_cont.py:76-97): 8 x [conv3x3+BN+ELU], max-pool
    after every second block, 1x1 conv + BN (identity), global mean pool.
    tparams: 45 arrays [W,beta,gamma,mean,inv_std] x 9.
    Returns H (N,32) f32; with deterministic=False also the per-block batch
    statistics [(mu, inv_std)] x 9 needed for the running-average updates."""
    x = np.ascontiguousarray(np.transpose(x_nchw, (0, 2, 3, 1)), dtype=F32)
    stats, cache = [], []
    for blk in range(9):
        W, beta, gamma, mean, inv_std = tparams[5 * blk:5 * blk + 5]
        z = conv2d_flip_nhwc(x, W)
        if deterministic:
            y = batchnorm_det_nhwc(z, beta, gamma, mean, inv_std)
        else:
            y, mu, istd = batchnorm_train_nhwc(z, beta, gamma)
            stats.append((mu, istd))
        a = elu(y) if blk < 8 else y
        pooled = maxpool2_nhwc(a) if (blk in (1, 3, 5, 7)) else a
        if return_cache:
            cache.append(dict(x=x, z=z, y=y, a=a))
        x = pooled
    n, h, w, c = x.shape
    H = x.reshape(n, h * w, c).mean(axis=1, dtype=F32)   # GlobalPoolLayer (A.3)
    if return_cache:
        return H, stats, cache
    if deterministic:
        return H
    return H, stats


# --------------------------------------------------------------------------
# CCALayer + LengthNormLayer
# --------------------------------------------------------------------------
def _inv_sqrt_sym(S):
    """cca.py:144-147: d,A = eigh(S); (A * 1/sqrt(d)).dot(A.T)."""
    d, A = np.linalg.eigh(S)
    return ((A * np.reciprocal(np.sqrt(d))).dot(A.T)).astype(F32), d, A


def cca_layer_train(H1, H2, cca_params, r1=1e-3, r2=1e-3, rT=1e-3, alpha=1.0):
    """CCALayer.get_output_for(deterministic=False) (cca.py:91-182, 198-201).
    cca_params = [U, V, mean1, mean2, S12, S11, S22] (running values).
    Returns out (B,64), corr (32,), new running params (same order)."""
    U0, V0, m1_0, m2_0, S12_0, S11_0, S22_0 = cca_params
    a = F32(alpha)
    one_m_a = F32(1.0 - alpha)
    m = F32(H1.shape[0])
    mean1 = one_m_a * m1_0 + a * H1.mean(axis=0, dtype=F32)       # :94,98
    mean2 = one_m_a * m2_0 + a * H2.mean(axis=0, dtype=F32)       # :95,103
    H1bar = (H1 - mean1).T                                          # :109,113
    H2bar = (H2 - mean2).T
    eye = np.eye(H1.shape[1], dtype=F32)
    S12 = (F32(1.0) / (m - 1)) * H1bar.dot(H2bar.T)                 # :117
    S11 = (F32(1.0) / (m - 1)) * H1bar.dot(H1bar.T) + F32(r1) * eye  # :120-121
    S22 = (F32(1.0) / (m - 1)) * H2bar.dot(H2bar.T) + F32(r2) * eye  # :124-125
    S12 = one_m_a * S12_0 + a * S12                                 # :128
    S11 = one_m_a * S11_0 + a * S11                                 # :133
    S22 = one_m_a * S22_0 + a * S22                                 # :138
    S11si, _, _ = _inv_sqrt_sym(S11)                                # :144-145
    S22si, _, _ = _inv_sqrt_sym(S22)                                # :146-147
    Tnp = S11si.dot(S12).dot(S22si)                                 # :150
    M1 = Tnp.dot(Tnp.T) + F32(rT) * eye                             # :151,153
    M2 = Tnp.T.dot(Tnp) + F32(rT) * eye                             # :152,154
    E1, E = np.linalg.eigh(M1)                                      # :157
    _, Fm = np.linalg.eigh(M2)                                      # :158
    corr = np.sqrt(np.clip(E1, 1e-7, 1.0)).astype(F32)              # :161-162
    U = S11si.dot(E)                                                # :167
    V = S22si.dot(Fm)                                               # :168
    s = np.sign(U.T.dot(S12).dot(V).diagonal())                     # :172
    U = (U * s).astype(F32)                                         # :173
    out = np.hstack([H1bar.T.dot(U), H2bar.T.dot(V)]).astype(F32)   # :198-201
    new = [U, V.astype(F32), mean1.astype(F32), mean2.astype(F32),
           S12.astype(F32), S11.astype(F32), S22.astype(F32)]
    return out, corr, new


def cca_layer_det(H1, H2, cca_params):
    """CCALayer.get_output_for(deterministic=True) (cca.py:185-201)."""
    U, V, mean1, mean2 = cca_params[:4]
    return (H1 - mean1).dot(U).astype(F32), (H2 - mean2).dot(V).astype(F32)


def length_norm(x):
    """LengthNormLayer (cca.py:39-40): x / ||x||_2 per row, no epsilon."""
    return (x / np.sqrt((x * x).sum(axis=1, dtype=F32)).reshape(-1, 1)).astype(F32)


# --------------------------------------------------------------------------
# compiled-callable equivalents (run_eval.py:92-95, retrieval_wrapper.py:33-38)
# --------------------------------------------------------------------------
def features_view1(x_prepared, params):
    """pre-CCA 32-d output of tower 1 (refine_cca.py:86-87), deterministic."""
    return tower_forward(x_prepared, params[0:45], True)


def features_view2(z, params):
    """pre-CCA 32-d output of tower 2 (refine_cca.py:88-89), deterministic."""
    return tower_forward(z, params[45:90], True)


def compute_v1_latent(x_prepared, params):
    """deterministic l_v1latent output (run_eval.py:92-93)."""
    H1 = features_view1(x_prepared, params)
    return length_norm((H1 - params[92]).dot(params[90]).astype(F32))


def compute_v2_latent(z, params):
    """deterministic l_v2latent output (run_eval.py:94-95)."""
    H2 = features_view2(z, params)
    return length_norm((H2 - params[93]).dot(params[91]).astype(F32))


def compute_output(x_prepared, z, params):
    """iter_funcs['compute_output'] (train_dcca_pool.py:158)."""
    return compute_v1_latent(x_prepared, params), compute_v2_latent(z, params)


def train_forward(x_prepared, z, params, r=(1e-3, 1e-3, 1e-3), alpha=1.0):
    """deterministic=False forward of both towers + CCALayer + length-norm
    (train_dcca_pool.py:100-101).  Returns lv1, lv2, corr and the parameter
    list after the default_update side effects (BN EMA (A.2), CCALayer (A.4))."""
    H1, st1 = tower_forward(x_prepared, params[0:45], False)
    H2, st2 = tower_forward(z, params[45:90], False)
    out, corr, new_cca = cca_layer_train(H1, H2, params[90:97], r[0], r[1], r[2], alpha)
    d = H1.shape[1]
    lv1, lv2 = length_norm(out[:, :d]), length_norm(out[:, d:])
    newp = [p.copy() for p in params]
    for t, st in enumerate((st1, st2)):
        for blk, (mu, istd) in enumerate(st):
            i = 45 * t + 5 * blk
            newp[i + 3] = ((F32(1) - BN_ALPHA) * params[i + 3] + BN_ALPHA * mu).astype(F32)
            newp[i + 4] = ((F32(1) - BN_ALPHA) * params[i + 4] + BN_ALPHA * istd).astype(F32)
    newp[90:97] = new_cca
    return lv1, lv2, corr, newp
