"""The oracle's hand-derived backward passes vs an INDEPENDENT float64 autograd.

The Theano graphs (public/GRU_Spatial.py:127-229, public/GRU.py:313-385, public/BPR.py:201-237)
define a scalar cost and let T.grad differentiate it.  Theano cannot run here, so the closest
available pin is: restate only the *forward cost* in torch float64, let torch.autograd produce the
gradients, apply the reference's update rule and compare with oracle.*_step.  This file imports
the oracle (it is a test) and torch CPU only.
"""
import numpy as np
import pytest
import torch

from oracle import poi_oracle as O

F64 = torch.float64      # explicit dtype everywhere: a global set_default_dtype would leak into other test modules


def _toy_spatial(seed, N=23, B=7, D=5, LM=9, L=6):
    rng = np.random.default_rng(seed)
    P = O.init_spatial_params(rng, N, B, D)
    P['bi'] = rng.uniform(-0.3, 0.3, (3, D)); P['bs'] = rng.uniform(-0.3, 0.3, B + 1)
    P['h0'] = np.zeros(D)
    p = np.full(LM, N); q = np.full(LM, N); dp = np.full(LM, B); dq = np.full(LM, B)
    # repeated POIs on purpose (multiplicity-weighted L2, duplicate scatter)
    p[:L] = rng.integers(0, 6, L); q[:L] = rng.integers(4, N, L)
    dp[1:L] = rng.integers(0, B + 1, L - 1); dq[1:L] = rng.integers(0, B + 1, L - 1)
    mask = np.array([1] * L + [0] * (LM - L))
    return P, p, q, dp, dq, mask


def _torch_spatial_cost(T, p, q, dp, dq, L, lam):
    lt, di, ui, wh, bi, vs, bs, wd, lw = (T[k] for k in ('lt', 'di', 'ui', 'wh', 'bi', 'vs', 'bs', 'wd', 'loss_weight'))
    xps, xqs, xds = lt[p], lt[q], di[dp]
    xs = torch.cat((xps, xds), 1)
    ls = torch.softmax(lw, 0)
    h = torch.zeros(lt.shape[1], dtype=F64)
    sur = 0.0; bpr = 0.0
    for t in range(L - 1):
        zr = torch.sigmoid(torch.einsum('gij,j->gi', ui[:2], xs[t]) + torch.einsum('gij,j->gi', wh[:2], h) + bi[:2])
        z, r = zr[0], zr[1]
        c = torch.tanh(ui[2] @ xs[t] + wh[2] @ (r * h) + bi[2])
        h = (1 - z) * h + z * c
        s = torch.softmax(vs @ h + bs, 0)
        u = h @ (xps[t + 1] - xqs[t + 1]) + wd * (s[dp[t + 1]] - s[dq[t + 1]])
        bpr = bpr + torch.log(torch.sigmoid(u))
        sur = sur + s[:dp[t + 1] + 1].sum() - torch.log(s[dp[t + 1]])
    upq = -bpr
    los = ls[0] * sur + ls[1] * upq
    l2 = sum((v ** 2).sum() for v in (xps, xqs, ui, wh, bi, xds, vs, bs, wd, ls))
    return los + 0.5 * lam * l2, los, sur, upq, ls


@pytest.mark.parametrize("seed,L", [(0, 6), (1, 4), (2, 9)])
def test_spatial_step_matches_autograd(seed, L):
    alpha, lam = 0.01, 0.001
    P, p, q, dp, dq, mask = _toy_spatial(seed, L=L)
    T = {k: torch.tensor(np.asarray(v, float), dtype=F64, requires_grad=True) for k, v in P.items() if k != 'h0'}
    cost, los, sur, upq, ls = _torch_spatial_cost(T, p, q, dp, dq, L, lam)
    cost.backward()
    Pn, out = O.spatial_step(P, p, q, dp, dq, mask, alpha, lam)
    assert np.isclose(out[0], los.item(), rtol=1e-13)
    assert np.isclose(out[1], float(sur.detach()), rtol=1e-13) and np.isclose(out[2], float(upq.detach()), rtol=1e-13)
    assert np.allclose(out[3], ls.detach().numpy(), rtol=1e-13)
    for k in ('ui', 'wh', 'bi', 'vs', 'bs', 'wd', 'loss_weight'):
        exp = T[k].detach().numpy() - alpha * T[k].grad.numpy()
        assert np.allclose(np.asarray(Pn[k]), exp, rtol=1e-11, atol=1e-14), k
    R = np.unique(np.concatenate((p, q))); S = np.unique(dp)
    lt_exp = P['lt'].copy(); lt_exp[R] -= alpha * T['lt'].grad.numpy()[R]
    di_exp = P['di'].copy(); di_exp[S] -= alpha * T['di'].grad.numpy()[S]
    assert np.allclose(Pn['lt'], lt_exp, rtol=1e-11, atol=1e-14)
    assert np.allclose(Pn['di'], di_exp, rtol=1e-11, atol=1e-14)
    # rows outside R / S untouched, including gradient-free rows
    untouched = np.setdiff1d(np.arange(P['lt'].shape[0]), R)
    assert np.array_equal(Pn['lt'][untouched], P['lt'][untouched])


def test_spatial_forward_cost_consistent_and_fd():
    alpha, lam = 0.01, 0.001
    P, p, q, dp, dq, mask = _toy_spatial(5)
    c0, los, sur, upq, ls = O.spatial_forward_cost(P, p, q, dp, dq, mask, lam)
    Pn, out = O.spatial_step(P, p, q, dp, dq, mask, alpha, lam)
    assert np.isclose(out[0], los, rtol=1e-14)
    # finite-difference probe on a few wh / vs entries through the update rule
    eps = 1e-6
    for name, idx in (('wh', (1, 2, 3)), ('vs', (2, 1)), ('ui', (2, 0, 7))):
        Pp = {k: np.array(v, float, copy=True) for k, v in P.items()}
        Pm = {k: np.array(v, float, copy=True) for k, v in P.items()}
        Pp[name][idx] += eps; Pm[name][idx] -= eps
        gfd = (O.spatial_forward_cost(Pp, p, q, dp, dq, mask, lam)[0] - O.spatial_forward_cost(Pm, p, q, dp, dq, mask, lam)[0]) / (2 * eps)
        gor = (P[name][idx] - Pn[name][idx]) / alpha
        assert np.isclose(gfd, gor, rtol=1e-6, atol=1e-9), (name, gfd, gor)


def _torch_gru_cost(T, p, q, L, lam):
    lt, ui, wh, bi = T['lt'], T['ui'], T['wh'], T['bi']
    xps, xqs = lt[p], lt[q]
    h = torch.zeros(lt.shape[1], dtype=F64); tot = 0.0
    for t in range(L):
        tot = tot + torch.log(torch.sigmoid(h @ (xps[t] - xqs[t])))
        z = torch.sigmoid(ui[0] @ xps[t] + wh[0] @ h + bi[0])
        r = torch.sigmoid(ui[1] @ xps[t] + wh[1] @ h + bi[1])
        c = torch.tanh(ui[2] @ xps[t] + wh[2] @ (r * h) + bi[2])
        h = (1 - z) * h + z * c
    return -tot + 0.5 * lam * sum((v ** 2).sum() for v in (xps, xqs, ui, wh, bi)), -tot


@pytest.mark.parametrize("seed,L", [(0, 5), (3, 8)])
def test_gru_step_matches_autograd(seed, L):
    alpha, lam = 0.01, 0.001
    rng = np.random.default_rng(seed)
    N, D, LM = 19, 6, 8
    P = O.init_gru_params(rng, N, D); P['bi'] = rng.uniform(-0.2, 0.2, (3, D))
    p = np.full(LM, N); q = np.full(LM, N)
    p[:L] = rng.integers(0, 5, L); q[:L] = rng.integers(3, N, L)
    mask = np.array([1] * L + [0] * (LM - L))
    T = {k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in P.items() if k != 'h0'}
    cost, loss = _torch_gru_cost(T, p, q, L, lam)
    cost.backward()
    Pn, out = O.gru_step(P, p, q, mask, alpha, lam)
    assert np.isclose(out, loss.item(), rtol=1e-13)
    for k in ('ui', 'wh', 'bi'):
        assert np.allclose(Pn[k], P[k] - alpha * T[k].grad.numpy(), rtol=1e-11, atol=1e-14), k
    R = np.unique(np.concatenate((p, q)))
    lt_exp = P['lt'].copy(); lt_exp[R] -= alpha * T['lt'].grad.numpy()[R]
    assert np.allclose(Pn['lt'], lt_exp, rtol=1e-11, atol=1e-14)


def _torch_gru_minibatch_cost(T, p_rows, q_rows, masks, lam):
    """Independent float64 autograd restatement of the batched scan (public/GRU.py:425-459): batched state, unmasked state update,
    masked loss, mean over the batch, L2 over every gathered row."""
    lt, ui, wh, bi = T['lt'], T['ui'], T['wh'], T['bi']
    B = p_rows.shape[0]
    xps, xqs = lt[torch.as_tensor(p_rows)], lt[torch.as_tensor(q_rows)]          # (B, LM, D)
    m = torch.as_tensor(masks, dtype=F64)
    n_steps = int(masks.sum(axis=1).max())
    h = torch.zeros((B, lt.shape[1]), dtype=F64); tot = 0.0
    for t in range(n_steps):
        xp, xq = xps[:, t], xqs[:, t]
        tot = tot + (torch.log(torch.sigmoid((h * (xp - xq)).sum(dim=1))) * m[:, t]).sum()
        z = torch.sigmoid(xp @ ui[0].T + h @ wh[0].T + bi[0])
        r = torch.sigmoid(xp @ ui[1].T + h @ wh[1].T + bi[1])
        c = torch.tanh(xp @ ui[2].T + (r * h) @ wh[2].T + bi[2])
        h = (1 - z) * h + z * c
    l2 = sum((v ** 2).sum() for v in (xps, xqs, ui, wh)) + (bi.expand(B, 3, -1) ** 2).sum() / B
    return -tot / B + 0.5 * lam * l2, -tot


@pytest.mark.parametrize("seed,lens", [(0, (5, 2, 8)), (4, (1, 7, 7, 3, 6))])
def test_gru_minibatch_step_matches_autograd(seed, lens):
    """f4, mini-batch Gru (public/GRU.py:395-498): forward cost and hand-assembled update against autograd of the batched scan."""
    alpha, lam = 0.01, 0.001
    rng = np.random.default_rng(seed)
    N, D, LM, B = 19, 6, 9, len(lens)
    P = O.init_gru_params(rng, N, D); P['bi'] = rng.uniform(-0.2, 0.2, (3, D))
    p_rows = np.full((B, LM), N); q_rows = np.full((B, LM), N); masks = np.zeros((B, LM), np.int64)
    for b, L in enumerate(lens):
        p_rows[b, :L] = rng.integers(0, 6, L); q_rows[b, :L] = rng.integers(3, N, L); masks[b, :L] = 1
    T = {k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in P.items() if k != 'h0'}
    cost, loss = _torch_gru_minibatch_cost(T, p_rows, q_rows, masks, lam)
    cost.backward()
    c_or, l_or = O.gru_minibatch_forward_cost(P, p_rows, q_rows, masks, lam)
    assert np.isclose(c_or, cost.item(), rtol=1e-13) and np.isclose(l_or, loss.item(), rtol=1e-13)
    Pn, out = O.gru_minibatch_step(P, p_rows, q_rows, masks, alpha, lam)
    assert np.isclose(out, loss.item(), rtol=1e-13)
    for k in ('ui', 'wh', 'bi'):
        assert np.allclose(Pn[k], P[k] - alpha * T[k].grad.numpy(), rtol=1e-11, atol=1e-14), k
    R = np.unique(np.concatenate((p_rows.ravel(), q_rows.ravel())))
    lt_exp = P['lt'].copy(); lt_exp[R] -= alpha * T['lt'].grad.numpy()[R]
    assert np.allclose(Pn['lt'], lt_exp, rtol=1e-11, atol=1e-14)
    # a batch of one is the one-by-one step (public/GRU.py:313-385)
    P1, o1 = O.gru_minibatch_step(P, p_rows[:1], q_rows[:1], masks[:1], alpha, lam)
    P2, o2 = O.gru_step(P, p_rows[0], q_rows[0], masks[0], alpha, lam)
    assert o1 == o2 and all(np.array_equal(P1[k], P2[k]) for k in ('lt', 'ui', 'wh', 'bi'))


def test_bpr_step_matches_autograd():
    alpha, lam = 0.01, 0.001
    rng = np.random.default_rng(7)
    P = O.init_bpr_params(rng, 11, 13, 8)
    u, pi, qi = 4, 2, 9
    T = {k: torch.tensor(v, dtype=F64, requires_grad=True) for k, v in P.items()}
    usr, xpq = T['ux'][u], T['lt'][[pi, qi]]
    upq = torch.log(torch.sigmoid(usr @ (xpq[0] - xpq[1])))
    cost = -upq + 0.5 * lam * ((usr ** 2).sum() + (xpq ** 2).sum())
    cost.backward()
    Pn, loss = O.bpr_step(P, u, pi, qi, alpha, lam)
    assert np.isclose(loss, -upq.item(), rtol=1e-13)
    assert np.allclose(Pn['ux'], P['ux'] - alpha * T['ux'].grad.numpy(), rtol=1e-12, atol=1e-15)
    assert np.allclose(Pn['lt'], P['lt'] - alpha * T['lt'].grad.numpy(), rtol=1e-12, atol=1e-15)


# ---- CA-RNN (public/CA_RNN.py:105-170) ---------------------------------------------------------------------------------
def _toy_carnn(seed, N=19, B=6, D=5, LM=9, L=6):
    rng = np.random.default_rng(seed)
    P = O.init_carnn_params(rng, N, B, D)
    p = np.full(LM, N); q = np.full(LM, N); dp = np.full(LM, B); dq = np.full(LM, B)
    p[:L] = rng.integers(0, 6, L); q[:L] = rng.integers(4, N, L)                  # repeated POIs on purpose
    dp[1:L] = rng.integers(0, B + 1, L - 1); dq[1:L] = rng.integers(0, B + 1, L - 1)      # shared interval matrices
    mask = np.array([1] * L + [0] * (LM - L))
    return P, p, q, dp, dq, mask


def _torch_carnn_cost(T, p, q, dp, dq, L, lam):
    lt, M, wd = T['lt'], T['M'], T['wd']
    xps, xqs, wdps, wdqs = lt[p], lt[q], wd[dp], wd[dq]
    h = torch.zeros(lt.shape[1], dtype=F64)
    tot = 0.0
    for t in range(L - 1):
        h = torch.sigmoid(M @ xps[t] + wdps[t] @ h)
        yp = (wdps[t + 1] @ h) @ (M @ xps[t + 1])
        yq = (wdqs[t + 1] @ h) @ (M @ xqs[t + 1])
        tot = tot + torch.log(torch.sigmoid(yp - yq))
    los = -tot
    l2 = sum((v ** 2).sum() for v in (xps, xqs, M, wdps, wdqs))
    return los + 0.5 * lam * l2, los


@pytest.mark.parametrize("seed,L", [(0, 6), (1, 4), (2, 9)])
def test_carnn_step_matches_autograd(seed, L):
    alpha, lam = 0.01, 0.001
    P, p, q, dp, dq, mask = _toy_carnn(seed, L=L)
    T = {k: torch.tensor(np.asarray(v, float), dtype=F64, requires_grad=True) for k, v in P.items() if k != 'h0'}
    cost, los = _torch_carnn_cost(T, p, q, dp, dq, L, lam)
    cost.backward()
    Pn, out = O.carnn_step(P, p, q, dp, dq, mask, alpha, lam)
    assert np.isclose(out, los.item(), rtol=1e-13)
    c0, l0 = O.carnn_forward_cost(P, p, q, dp, dq, mask, lam)
    assert np.isclose(c0, cost.item(), rtol=1e-13) and np.isclose(l0, los.item(), rtol=1e-13)
    assert np.allclose(Pn['M'], P['M'] - alpha * T['M'].grad.numpy(), rtol=1e-11, atol=1e-14)
    R = np.unique(np.concatenate((p, q))); S = np.unique(np.concatenate((dp, dq)))
    lt_exp = P['lt'].copy(); lt_exp[R] -= alpha * T['lt'].grad.numpy()[R]
    wd_exp = P['wd'].copy(); wd_exp[S] -= alpha * T['wd'].grad.numpy()[S]
    assert np.allclose(Pn['lt'], lt_exp, rtol=1e-11, atol=1e-14)
    assert np.allclose(Pn['wd'], wd_exp, rtol=1e-11, atol=1e-14)
    assert np.array_equal(Pn['wd'][np.setdiff1d(np.arange(P['wd'].shape[0]), S)], P['wd'][np.setdiff1d(np.arange(P['wd'].shape[0]), S)])


def test_carnn_predict_and_scores_follow_the_literal_broadcast_sums():
    """The predict / scoring graphs of CA-RNN add-then-sum (public/CA_RNN.py:97-100,191): restated independently with
    explicit broadcasting in torch."""
    rng = np.random.default_rng(3)
    N, B, D, n, LM = 17, 5, 4, 3, 6
    P = O.init_carnn_params(rng, N, B, D)
    lens = [6, 4, 5]
    p_rows = rng.integers(0, N, (n, LM)); d_rows = rng.integers(0, B + 1, (n, LM))
    masks = np.array([[1] * L + [0] * (LM - L) for L in lens])
    hts = O.carnn_predict(P, P['lt'], P['wd'], p_rows, d_rows, masks)
    M, wd, lt = (torch.tensor(P[k], dtype=F64) for k in ('M', 'wd', 'lt'))
    for k in range(n):
        h = torch.zeros(1, D, dtype=F64)
        for t in range(lens[k]):
            p_t = lt[p_rows[k, t]][None, :]; wd_t = wd[d_rows[k, t]][None]
            h = torch.sigmoid(p_t @ M.T + torch.sum(wd_t + h.reshape(1, 1, D), 2))
        assert np.allclose(hts[k], h[0].numpy(), rtol=1e-13)
    ul = rng.integers(0, B + 1, (n, N))
    sc = O.carnn_score_all(hts, P['lt'], P['M'], P['wd'], ul)
    users = torch.tensor(hts, dtype=F64)
    h_W = torch.sum(wd[torch.tensor(ul)] + users.reshape(n, 1, 1, D), 3)
    r_M = (lt[:-1] @ M.T).reshape(1, N, D)
    assert np.allclose(sc, (-torch.sum(h_W + r_M, 2)).numpy(), rtol=1e-12)
