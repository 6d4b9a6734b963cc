"""CPU: pin the oracle's training step (loss, backward incl. EighGrad, BN batch
statistics, max-pool routing, L2, Lasagne-Adam) against torch autograd of an
independently written float64 forward."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import network as onet, train as otrain


def _params(rng, nf, dtype):
    ps = []
    for _t in range(2):
        for ci, co, k in onet.tower_channels(nf):
            lim = np.sqrt(3.0 / (ci * k * k))
            ps += [rng.uniform(-lim, lim, (co, ci, k, k)), rng.standard_normal(co) * 0.1,
                   1.0 + 0.2 * rng.standard_normal(co), np.zeros(co), np.ones(co)]
    ps += [np.zeros((32, 32)), np.zeros((32, 32)), np.zeros(32), np.zeros(32),
           np.zeros((32, 32)), np.zeros((32, 32)), np.zeros((32, 32))]
    return [p.astype(dtype) for p in ps]


def _torch_loss(x1, x2, params, gamma=0.7, l2=1e-5, r=1e-3):
    tp = [torch.tensor(p, dtype=torch.float64, requires_grad=(i < 90 and i % 5 in (0, 1, 2)))
          for i, p in enumerate(params)]

    def tower(x, base):
        h = torch.tensor(x, dtype=torch.float64)
        for blk in range(9):
            W, beta, gamma_ = tp[base + 5 * blk], tp[base + 5 * blk + 1], tp[base + 5 * blk + 2]
            k = W.shape[-1]
            h = F.conv2d(h, torch.flip(W, dims=(2, 3)), padding=(k - 1) // 2)
            mu = h.mean(dim=(0, 2, 3), keepdim=True)
            var = ((h - mu) ** 2).mean(dim=(0, 2, 3), keepdim=True)
            h = (h - mu) * (gamma_[None, :, None, None] / torch.sqrt(var + 1e-4)) + beta[None, :, None, None]
            if blk < 8:
                h = F.elu(h)
            if blk in (1, 3, 5, 7):
                h = F.max_pool2d(h, 2)
        return h.flatten(2).mean(dim=2)

    H1, H2 = tower(x1, 0), tower(x2, 45)
    m = H1.shape[0]
    Hb1, Hb2 = H1 - H1.mean(0), H2 - H2.mean(0)
    eye = torch.eye(32, dtype=torch.float64)
    S12 = Hb1.T @ Hb2 / (m - 1)
    S11 = Hb1.T @ Hb1 / (m - 1) + r * eye
    S22 = Hb2.T @ Hb2 / (m - 1) + r * eye

    def inv_sqrt(S):
        d, A = torch.linalg.eigh(S)
        return (A / torch.sqrt(d)) @ A.T
    S11si, S22si = inv_sqrt(S11), inv_sqrt(S22)
    T = S11si @ S12 @ S22si
    _, E = torch.linalg.eigh(T @ T.T + r * eye)
    _, Fm = torch.linalg.eigh(T.T @ T + r * eye)
    U, V = S11si @ E, S22si @ Fm
    s = torch.sign(torch.diagonal(U.T @ S12 @ V)).detach()
    U = U * s
    o1, o2 = Hb1 @ U, Hb2 @ V
    lv1 = o1 / o1.norm(dim=1, keepdim=True)
    lv2 = o2 / o2.norm(dim=1, keepdim=True)
    D = lv1 @ lv2.T
    L = torch.clamp(gamma - torch.diagonal(D)[:, None] + D, 0, 1000)
    off = ~torch.eye(m, dtype=torch.bool)
    loss = L[off].mean()
    pen = sum((p * p).sum() for p in tp if p.requires_grad)
    total = loss + l2 * pen
    total.backward()
    return total.item(), [p.grad.numpy() for p in tp if p.requires_grad], lv1.detach().numpy(), lv2.detach().numpy()


@pytest.fixture(scope="module")
def problem():
    rng = np.random.default_rng(7)
    B = 48
    x1 = rng.random((B, 1, 32, 40))
    x2 = rng.random((B, 1, 24, 18)) * 2
    return rng, x1, x2, _params(rng, 12, np.float64)


def test_loss_and_every_gradient_match_torch_float64(problem):
    rng, x1, x2, params = problem
    total, corr, grads, newp, (lv1, lv2) = otrain.loss_and_grads(x1, x2, params)
    t_total, t_grads, t_lv1, t_lv2 = _torch_loss(x1, x2, params)
    assert abs(total - t_total) <= 1e-10
    # eigenvector sign/ordering conventions may differ per dimension; the score matrix may not
    assert np.abs(lv1 @ lv2.T - t_lv1 @ t_lv2.T).max() <= 1e-8
    assert len(grads) == len(t_grads) == 54
    for i, (g, tg) in enumerate(zip(grads, t_grads)):
        scale = max(1e-8, np.abs(tg).max())
        assert np.abs(g - tg).max() <= 1e-6 * scale + 1e-12, "gradient %d: %g vs scale %g" % (
            i, np.abs(g - tg).max(), scale)
    assert corr.shape == (32,) and (corr >= 0).all() and (corr <= 1).all()
    # running-stat side effects
    assert np.allclose(newp[3], 0.1 * newp[3] / 0.1) and not np.array_equal(newp[3], params[3])
    assert np.abs(newp[90].T @ newp[95] @ newp[90] - np.eye(32)).max() < 1e-6      # U' S11 U = I


def test_float32_run_agrees_with_float64(problem):
    rng, x1, x2, params = problem
    p32 = [p.astype(np.float32) for p in params]
    t32, _, g32, _, _ = otrain.loss_and_grads(x1.astype(np.float32), x2.astype(np.float32), p32)
    t64, _, g64, _, _ = otrain.loss_and_grads(x1, x2, params)
    assert abs(t32 - t64) <= 1e-4
    rel = [np.abs(a - b).max() / max(1e-6, np.abs(b).max()) for a, b in zip(g32, g64)]
    assert max(rel) <= 5e-2, max(rel)          # float32 eigh gradients: loose, documents the sensitivity
    assert np.median(rel) <= 2e-3


def test_adam_matches_hand_formula_and_lasagne_epsilon_placement():
    rng = np.random.default_rng(0)
    params = _params(rng, 12, np.float32)
    grads = [rng.standard_normal(params[i].shape).astype(np.float32) for i in otrain.TRAINABLE]
    st = otrain.adam_init(params)
    p1, st1 = otrain.adam_update(params, grads, st, lr=0.002)
    # first step: m = .1 g, v = .001 g^2, a_t = lr sqrt(.001)/.1 -> step = a_t * .1 g / (sqrt(.001) |g| + 1e-8)
    g, p0 = grads[0], params[0]
    a_t = 0.002 * np.sqrt(1 - 0.999) / (1 - 0.9)
    ref = p0 - a_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    assert np.abs(p1[0] - ref).max() <= 1e-7
    assert st1["t"] == 1 and np.allclose(st1["m"][0], 0.1 * g)
    assert np.array_equal(p1[3], params[3]) and np.array_equal(p1[90], params[90])    # non-trainables untouched
    p2, st2 = otrain.adam_update(p1, grads, st1, lr=0.002)
    assert st2["t"] == 2 and not np.array_equal(p2[0], p1[0])


def test_train_step_decreases_loss_and_valid_loss_runs():
    rng = np.random.default_rng(3)
    B = 40
    x1 = rng.random((B, 1, 32, 40)).astype(np.float32)
    x2 = (rng.random((B, 1, 24, 18)) * 2).astype(np.float32)
    params = _params(rng, 12, np.float32)
    st = otrain.adam_init(params)
    losses = []
    for _ in range(4):
        loss, corr, params, st = otrain.train_step(x1, x2, params, st, lr=0.002)
        losses.append(float(loss))
    assert losses[-1] < losses[0] and np.isfinite(losses).all()
    v = otrain.valid_loss(x1, x2, params)
    assert np.isfinite(v) and v >= 0


def test_loss_closed_form_small():
    rng = np.random.default_rng(1)
    a = rng.standard_normal((5, 32)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.standard_normal((5, 32)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    loss, d1, d2 = otrain.contrastive_cos_loss(a, b, 0.7)
    brute = np.mean([min(max(0.7 - a[i] @ b[i] + a[i] @ b[j], 0), 1000) for i in range(5) for j in range(5) if i != j])
    assert abs(loss - brute) < 1e-12 and loss >= 0
    eps = 1e-6
    a2 = a.copy(); a2[1, 3] += eps
    num = (otrain.contrastive_cos_loss(a2, b, 0.7)[0] - loss) / eps
    assert abs(num - d1[1, 3]) < 1e-5


def test_imposed_routing_reproduces_the_free_evaluation_and_moves_the_gradient():
    """oracle.train.loss_and_grads(routing=...): imposing the arg-max the free evaluation takes anyway changes nothing
    (bit for bit); routing_from_selected recovers it from (z, value of the selected element); imposing another element
    of one window changes the gradients - the hook tests/test_gpu_train_routed.py hangs the device's selection on"""
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    from oracle import train as otrain
    rng = np.random.default_rng(3)
    params = [p.astype(np.float64) for p in synth_data.synth_params(param_shapes("mutopia_ccal_cont"), seed=1,
                                                                    trained_like=True)]
    x1 = rng.random((6, 1, 32, 32))
    x2 = rng.random((6, 1, 32, 16)) * 2
    free = otrain.loss_and_grads(x1, x2, params)
    routing = ({}, {})
    for t, x in enumerate((x1, x2)):
        _, _, cache, _ = otrain.tower_forward_train(x, params[45 * t:45 * t + 45])
        for blk in (1, 3, 5, 7):
            a, z = cache[blk]["a"], cache[blk]["z"]
            arg = otrain._windows(a).argmax(axis=-1)
            zsel = np.take_along_axis(otrain._windows(z), arg[..., None], axis=-1)[..., 0]
            assert np.array_equal(otrain.routing_from_selected(z, zsel), arg)
            routing[t][blk] = arg
    same = otrain.loss_and_grads(x1, x2, params, routing=routing)
    assert same[0] == free[0]
    for a, b in zip(same[2], free[2]):
        assert np.array_equal(a, b)
    routing[0][1] = routing[0][1].copy()
    routing[0][1][0, 0, 0, :] = (routing[0][1][0, 0, 0, :] + 1) % 4
    other = otrain.loss_and_grads(x1, x2, params, routing=routing)
    assert any(not np.array_equal(a, b) for a, b in zip(other[2], free[2]))
    with pytest.raises(ValueError):
        otrain.routing_from_selected(np.zeros((1, 2, 2, 1)), np.ones((1, 1, 1, 1)))


def test_maxpool_backward_tie_rules_against_a_loop():
    """maxpool2_bwd_nhwc: ties="all" is Theano's CPU MaxPoolGrad (every element equal to the window maximum receives the
    pooled gradient, the full amount each - SURVEY 8a row 3), ties="first" feeds the first maximum in row-major order;
    both against a plain loop on an input full of ties, odd sizes (the last row / column belongs to no window)"""
    rng = np.random.default_rng(11)
    a = rng.integers(0, 3, (2, 5, 7, 3)).astype(np.float64)          # three distinct values: ties everywhere
    g = rng.standard_normal((2, 2, 3, 3))
    for ties in ("all", "first"):
        want = np.zeros_like(a)
        for n in range(2):
            for oy in range(2):
                for ox in range(3):
                    for c in range(3):
                        win = [(2 * oy + dy, 2 * ox + dx) for dy in (0, 1) for dx in (0, 1)]
                        m = max(a[n, y, x, c] for y, x in win)
                        hits = [(y, x) for y, x in win if a[n, y, x, c] == m]
                        for y, x in (hits if ties == "all" else hits[:1]):
                            want[n, y, x, c] += g[n, oy, ox, c]
        got = otrain.maxpool2_bwd_nhwc(a, g, ties=ties)
        assert np.array_equal(got, want), ties
    assert not np.array_equal(otrain.maxpool2_bwd_nhwc(a, g, ties="all"), otrain.maxpool2_bwd_nhwc(a, g, ties="first"))
    # an imposed boolean set: its members receive the gradient, its first member is what the forward passes on
    sets = otrain._windows(a) == otrain._windows(a).max(axis=-1, keepdims=True)
    assert np.array_equal(otrain.maxpool2_bwd_nhwc(a, g, route=sets), otrain.maxpool2_bwd_nhwc(a, g, ties="all"))
    assert np.array_equal(otrain.maxpool2_routed_nhwc(a, sets), onet.maxpool2_nhwc(a))
    bits = (sets * (1 << np.arange(4))).sum(axis=-1)
    assert np.array_equal(otrain.routing_from_tie_sets(bits.astype(np.float32)), sets)
    assert otrain.route_check(a, sets) == (0.0, 0.0)
    wrong = np.roll(sets, 1, axis=-1)
    gap, flips = otrain.route_check(a, wrong)
    assert gap > 0.1 and flips > 0.1              # a selection that is not the maximum is visible
    with pytest.raises(ValueError):
        otrain.maxpool2_bwd_nhwc(a, g, ties="every")


def whiten_pages(sheet, keep=(60, 100)):
    """pages with large white areas: everything outside the rows keep[0]:keep[1] is blank paper (255)"""
    out = np.full_like(sheet, 255)
    out[:, :, keep[0]:keep[1], :] = sheet[:, :, keep[0]:keep[1], :]
    return out


def test_synthetic_pages_hold_pooling_ties_and_the_two_rules_give_different_gradients():
    """White paper gives bit-identical activations: on synth_pairs(seed=23) several per cent of the sheet tower's first
    pooling windows (block 2) hold equal maxima, most of them on a page with large white areas, none in the spectrogram
    tower - and the gradients of the sheet tower's first two blocks depend on the rule by far more than the 1e-4 parity
    bar (19-95 % at batch 48).  The rule is therefore part of the contract: "all" = Theano CPU."""
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    params = [p.astype(np.float64) for p in synth_data.synth_params(param_shapes("mutopia_ccal_cont"), seed=1,
                                                                    trained_like=False)]
    sheet, spec = synth_data.synth_pairs(np.arange(6), seed=23)
    x1 = onet.prepare(sheet, "mutopia_ccal_cont").astype(np.float64)
    _, _, cache, _ = otrain.tower_forward_train(x1, params[0:45])
    two, four = otrain.pool_tie_share(cache[1]["a"])
    assert 0.02 <= two <= 0.2 and four > 0.002, (two, four)
    assert otrain.pool_tie_share(cache[3]["a"])[0] < 1e-3
    _, _, cache2, _ = otrain.tower_forward_train(spec.astype(np.float64), params[45:90])
    assert otrain.pool_tie_share(cache2[1]["a"])[0] == 0.0
    white = onet.prepare(whiten_pages(sheet), "mutopia_ccal_cont").astype(np.float64)
    _, _, cache_w, _ = otrain.tower_forward_train(white, params[0:45])
    assert otrain.pool_tie_share(cache_w[1]["a"])[0] >= 0.5
    # the gradients: crops (the rule matters wherever ties exist; a small geometry keeps this test fast, and more than 32
    # samples keep the 32-d CCA from aligning the two views perfectly, which would switch the ranking loss off)
    sheet, spec = synth_data.synth_pairs(np.arange(40), seed=23)
    xs = onet.prepare(sheet, "mutopia_ccal_cont").astype(np.float64)[:, :, 40:88, 60:124]
    zs = spec[:, :, :32, :24].astype(np.float64)
    g_all = otrain.loss_and_grads(xs, zs, params, ties="all")
    g_first = otrain.loss_and_grads(xs, zs, params, ties="first")
    assert g_all[0] == g_first[0]                                   # the loss does not see the rule
    rel = [float(np.abs(a - b).max() / np.abs(a).max()) for a, b in zip(g_all[2][:6], g_first[2][:6])]
    assert min(rel) > 0.02, rel                                     # W1, beta1, gamma1, W2, beta2, gamma2 of the sheet tower
    for a, b in zip(g_all[2][27:], g_first[2][27:]):                # the spectrogram tower has no ties: same gradients
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * max(1.0, float(np.abs(a).max())))


def test_ties_decided_before_or_after_the_float32_elu():
    """the device compares y (BatchNorm output), the reference the float32 ELU output.  (a) a constructed
    window of strongly negative, DISTINCT y whose float32 ELU images coincide: the reference rule (`all`, on the pooled
    activations) feeds both elements, the y rule one - the documented deviation, scaled by ELU'(y) = exp(y).  (b) on
    the synthetic pages, blank-paper pages included, no window's tie set depends on where the comparison is made: tied
    windows there are bit-identical patches, which tie before and after any function."""
    y0 = np.float32(-3.0)
    y1 = np.nextafter(y0, np.float32(0))                                  # one float above: same expm1 image
    assert y1 != y0 and otrain.elu_f32(y1) == otrain.elu_f32(y0)
    y = np.array([[[[y0], [y1]], [[-9.0], [-12.0]]]], np.float32)          # (1, 2, 2, 1): one window
    on_y, on_a, differ, worst = otrain.elu_tie_deviation(y)
    assert (on_y, on_a, differ) == (0.0, 1.0, 1.0) and abs(worst - np.exp(-3.0)) < 1e-6
    g = np.ones((1, 1, 1, 1), np.float32)
    np.testing.assert_array_equal(otrain.maxpool2_bwd_nhwc(otrain.elu_f32(y), g)[0, :, :, 0], [[1, 1], [0, 0]])
    np.testing.assert_array_equal(otrain.maxpool2_bwd_nhwc(y, g)[0, :, :, 0], [[0, 1], [0, 0]])
    sat = np.full((1, 2, 2, 1), -20.0, np.float32); sat[0, 0, 0, 0] = -18.0   # all images are -1.0f: four-way tie,
    assert otrain.elu_tie_deviation(sat)[3] < 2e-8                            # carrying exp(-18) of the gradient
    from audio_sheet_retrieval_amd.utils import synth_data
    from audio_sheet_retrieval_amd.utils.param_layout import param_shapes
    params = synth_data.synth_params(param_shapes("mutopia_ccal_cont"), seed=1, trained_like=True)
    sheet, _ = synth_data.synth_pairs(np.arange(4), seed=23)
    for pages in (sheet, whiten_pages(sheet)):
        x1 = onet.prepare(pages, "mutopia_ccal_cont").astype(np.float32)
        _, _, cache, _ = otrain.tower_forward_train(x1, [p.astype(np.float32) for p in params[0:45]])
        for blk in (1, 3, 5, 7):
            on_y, on_a, differ, worst = otrain.elu_tie_deviation(cache[blk]["y"])
            assert differ <= 2e-6 and on_a >= on_y, (blk, on_y, on_a, differ, worst)


def test_symmetric_and_weighted_loss_against_a_loop():
    """get_contrastive_cos_loss(weight, gamma, symmetric=True) (models/objectives.py:53-67): direction 2 is the same hinge
    on D = lv2 lv1^T, the sum is scaled by weight; value by brute force, gradients by finite differences"""
    rng = np.random.default_rng(4)
    a = rng.standard_normal((7, 32)); a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = a + 0.4 * rng.standard_normal((7, 32)); b /= np.linalg.norm(b, axis=1, keepdims=True)
    n, gam, w = 7, 0.7, 1.7

    def brute(a, b):
        d1 = np.mean([min(max(gam - a[i] @ b[i] + a[i] @ b[j], 0), 1000) for i in range(n) for j in range(n) if i != j])
        d2 = np.mean([min(max(gam - b[i] @ a[i] + b[i] @ a[j], 0), 1000) for i in range(n) for j in range(n) if i != j])
        return w * (d1 + d2)
    loss, g1, g2 = otrain.contrastive_cos_loss(a, b, gam, weight=w, symmetric=True)
    assert abs(loss - brute(a, b)) < 1e-12
    one = otrain.contrastive_cos_loss(a, b, gam)
    assert loss > w * one[0] > 0
    eps = 1e-6
    for (r, c) in ((1, 3), (4, 0), (6, 31)):
        a2 = a.copy(); a2[r, c] += eps
        assert abs((brute(a2, b) - brute(a, b)) / eps - g1[r, c]) < 1e-5
        b2 = b.copy(); b2[r, c] += eps
        assert abs((brute(a, b2) - brute(a, b)) / eps - g2[r, c]) < 1e-5
