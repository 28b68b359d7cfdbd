"""CPU oracle (float64, numpy) for the next-POI hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy float64, the arithmetic that the reference's Theano graphs
define for the hot path named by BASELINE.json.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product
package (``point-of-interest-recommendation_amd/``) never imports anything from ``oracle/``.

Pinning status
--------------
* numpy half (top-K helpers, metrics, Haversine bins, masks, negative-bin construction): PINNED.
  ``tests/golden/make_golden.py`` imports the reference's own ``public/Valuate.py`` and
  ``public/Load_Data_by_length.py`` in the build container, runs them on seeded inputs and
  commits inputs+outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
  file against those vectors.
* Theano half (train steps, predict, scoring, AUC preference): PARITY UNPINNED.  Theano is a
  third-party dependency that is not vendored under /root/reference, has no pinned version
  (API use implies Theano 0.7-1.0.x on CPython 2.7) and cannot be installed here; the
  reference holds no golden vectors or tests for it.  The restatement follows the graph
  definitions line by line (citations below) and its hand-derived backward pass is checked
  against an independent float64 autograd of the same cost and against finite differences
  (``tests/test_oracle_autograd.py``).

All ``file:line`` citations are relative to /root/reference.
"""
from __future__ import annotations

import math
import numpy as np

F64 = np.float64


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def log_sigmoid(x):
    # log(sigmoid(x)), stable.  Theano rewrites log(sigmoid(x)) to -softplus(-x)
    # (SURVEY 8c); identical to <1e-16 in the tested range.
    return -np.logaddexp(0.0, -x)


def softmax0(x):
    """public/GRU_Spatial.py:31-37 - max-subtracted softmax along axis 0."""
    e = np.exp(x - x.max(axis=0, keepdims=True))
    return e / e.sum(axis=0, keepdims=True)


# ----------------------------------------------------------------------------------------------
# parameter construction (shapes + init ranges only; the reference init is unseeded)
# ----------------------------------------------------------------------------------------------
def init_spatial_params(rng, n_item, n_dist, d):
    """Shapes/ranges of public/GRU.py:58-70 and public/GRU_Spatial.py:50-71."""
    u = lambda *s: rng.uniform(-0.5, 0.5, s)
    return dict(
        lt=u(n_item + 1, d), di=u(n_dist + 1, d),
        ui=u(3, d, 2 * d), wh=u(3, d, d), bi=np.zeros((3, d)),
        vs=u(n_dist + 1, d), bs=np.zeros(n_dist + 1),
        wd=float(rng.uniform(0, 0.5)), loss_weight=u(2), h0=np.zeros(d))


def init_gru_params(rng, n_item, d):
    """public/GRU.py:58-70 (OboGru: ui is (3,D,D))."""
    u = lambda *s: rng.uniform(-0.5, 0.5, s)
    return dict(lt=u(n_item + 1, d), ui=u(3, d, d), wh=u(3, d, d), bi=np.zeros((3, d)),
                h0=np.zeros(d))


def init_bpr_params(rng, n_user, n_item, d):
    """public/BPR.py:50-54."""
    u = lambda *s: rng.uniform(-0.5, 0.5, s)
    return dict(ux=u(n_user, d), lt=u(n_item + 1, d))


# ----------------------------------------------------------------------------------------------
# GRU cell (shared by all recurrent paths)
# ----------------------------------------------------------------------------------------------
def _gru_cell(ui, wh, bi, x, hp):
    """public/GRU_Spatial.py:173-178 == public/GRU.py:345-350."""
    z = sigmoid(ui[0] @ x + wh[0] @ hp + bi[0])
    r = sigmoid(ui[1] @ x + wh[1] @ hp + bi[1])
    c = np.tanh(ui[2] @ x + wh[2] @ (r * hp) + bi[2])
    h = (1.0 - z) * hp + z * c
    return z, r, c, h


def _gru_cell_bwd(ui, wh, x, hp, z, r, c, dh, g):
    """Backward of one cell.  ``g`` is a dict of dense-gradient accumulators (ui, wh, bi).
    Returns (dx, dhp)."""
    dz = dh * (c - hp)
    dc = dh * z
    dhp = dh * (1.0 - z)
    da_c = dc * (1.0 - c * c)
    rh = r * hp
    g['ui'][2] += np.outer(da_c, x)
    g['wh'][2] += np.outer(da_c, rh)
    g['bi'][2] += da_c
    m = wh[2].T @ da_c
    dr = m * hp
    dhp = dhp + m * r
    dx = ui[2].T @ da_c
    da_z = dz * z * (1.0 - z)
    da_r = dr * r * (1.0 - r)
    g['ui'][0] += np.outer(da_z, x)
    g['ui'][1] += np.outer(da_r, x)
    g['wh'][0] += np.outer(da_z, hp)
    g['wh'][1] += np.outer(da_r, hp)
    g['bi'][0] += da_z
    g['bi'][1] += da_r
    dhp = dhp + wh[0].T @ da_z + wh[1].T @ da_r
    dx = dx + ui[0].T @ da_z + ui[1].T @ da_r
    return dx, dhp


# ----------------------------------------------------------------------------------------------
# a2: Distance2Pre training step  (public/GRU_Spatial.py:127-229)
# ----------------------------------------------------------------------------------------------
def spatial_forward_cost(P, p, q, dp, dq, mask, lam):
    """Forward only: returns (cost, los, sur, upq, ls).  Used by finite-difference tests."""
    lt, di, ui, wh, bi = P['lt'], P['di'], P['ui'], P['wh'], P['bi']
    vs, bs, wd = P['vs'], P['bs'], P['wd']
    L = int(np.sum(mask))
    xps, xqs, xds = lt[p], lt[q], di[dp]                     # :144-146 (all LM rows)
    xs = np.concatenate((xps, xds), axis=1)                  # :147
    ls = softmax0(np.asarray(P['loss_weight'], F64))         # :156
    h = P['h0'].copy()
    sur = 0.0
    bpr = 0.0
    for t in range(L - 1):                                   # :193-197, n_steps = L-1
        _, _, _, h = _gru_cell(ui, wh, bi, xs[t], h)
        s = softmax0(vs @ h + bs)                            # :180
        a, b = int(dp[t + 1]), int(dq[t + 1])
        u = h @ (xps[t + 1] - xqs[t + 1]) + wd * (s[a] - s[b])   # :184
        bpr += log_sigmoid(u)                                # :186
        sur += s[:a + 1].sum() - math.log(s[a])              # :189
    upq = -bpr                                               # :205
    los = ls[0] * sur + ls[1] * upq                          # :206
    l2 = sum(np.sum(v * v) for v in (xps, xqs, ui, wh, bi, xds, vs, bs)) + wd * wd + np.sum(ls * ls)
    return los + 0.5 * lam * l2, los, sur, upq, ls           # :202-209


def spatial_step(P, p, q, dp, dq, mask, alpha, lam):
    """One ``seq_train(uidx)`` call (public/GRU_Spatial.py:127-229).

    P: dict of float64 arrays (lt, di, ui, wh, bi, vs, bs, wd, loss_weight, h0);
    p, q, dp, dq, mask: the five padded int rows of length LM (givens, :225-229).
    Returns (P_new, [los, sur, upq, ls]); P is not modified (Theano ``updates`` semantics:
    every update is evaluated at the old values).
    """
    p, q, dp, dq = (np.asarray(v, np.int64) for v in (p, q, dp, dq))
    lt, di, ui, wh, bi = P['lt'], P['di'], P['ui'], P['wh'], P['bi']
    vs, bs, wd = P['vs'], P['bs'], float(P['wd'])
    D = lt.shape[1]
    L = int(np.sum(mask))
    xps, xqs, xds = lt[p], lt[q], di[dp]
    xs = np.concatenate((xps, xds), axis=1)
    ls = softmax0(np.asarray(P['loss_weight'], F64))
    nstep = max(L - 1, 0)
    hs = np.zeros((nstep + 1, D))
    hs[0] = P['h0']
    zs = np.zeros((nstep, D)); rs = np.zeros((nstep, D)); cs = np.zeros((nstep, D))
    ss = np.zeros((nstep, vs.shape[0])); us = np.zeros(nstep)
    sur = 0.0
    bpr = 0.0
    for t in range(nstep):
        zs[t], rs[t], cs[t], hs[t + 1] = _gru_cell(ui, wh, bi, xs[t], hs[t])
        h = hs[t + 1]
        s = softmax0(vs @ h + bs)
        ss[t] = s
        a, b = int(dp[t + 1]), int(dq[t + 1])
        us[t] = h @ (xps[t + 1] - xqs[t + 1]) + wd * (s[a] - s[b])
        bpr += log_sigmoid(us[t])
        sur += s[:a + 1].sum() - math.log(s[a])
    upq = -bpr
    los = ls[0] * sur + ls[1] * upq

    # ---- backward (hand-derived BPTT of :202-213) --------------------------------
    g = dict(ui=np.zeros_like(ui), wh=np.zeros_like(wh), bi=np.zeros_like(bi),
             vs=np.zeros_like(vs), bs=np.zeros_like(bs))
    g_wd = 0.0
    g_lt = np.zeros_like(lt)
    g_di = np.zeros_like(di)
    dh_next = np.zeros(D)
    for t in range(nstep - 1, -1, -1):
        h, s = hs[t + 1], ss[t]
        a, b = int(dp[t + 1]), int(dq[t + 1])
        gu = -ls[1] * sigmoid(-us[t])                        # d cost / d u_t
        e = xps[t + 1] - xqs[t + 1]
        dh = dh_next + gu * e
        g_lt[p[t + 1]] += gu * h
        g_lt[q[t + 1]] -= gu * h
        g_wd += gu * (s[a] - s[b])
        ds = np.zeros_like(s)
        ds[a] += gu * wd
        ds[b] -= gu * wd
        ds[:a + 1] += ls[0]
        ds[a] -= ls[0] / s[a]
        do = s * (ds - ds @ s)
        g['vs'] += np.outer(do, h)
        g['bs'] += do
        dh = dh + vs.T @ do
        dx, dh_next = _gru_cell_bwd(ui, wh, xs[t], hs[t], zs[t], rs[t], cs[t], dh, g)
        g_lt[p[t]] += dx[:D]
        g_di[dp[t]] += dx[D:]
    # L2 over ALL LM gathered rows, multiplicity-weighted (:202-203)
    np.add.at(g_lt, p, lam * xps)
    np.add.at(g_lt, q, lam * xqs)
    np.add.at(g_di, dp, lam * xds)
    dls = np.array([sur, upq]) + lam * ls
    g_lw = ls * (dls - dls @ ls)

    N = dict(P)
    N['ui'] = ui - alpha * (g['ui'] + lam * ui)              # :210-211
    N['wh'] = wh - alpha * (g['wh'] + lam * wh)
    N['bi'] = bi - alpha * (g['bi'] + lam * bi)
    N['vs'] = vs - alpha * (g['vs'] + lam * vs)
    N['bs'] = bs - alpha * (g['bs'] + lam * bs)
    N['wd'] = wd - alpha * (g_wd + lam * wd)
    N['loss_weight'] = np.asarray(P['loss_weight'], F64) - alpha * g_lw
    R = np.unique(np.concatenate((p, q)))                    # :149-151
    S = np.unique(dp)                                        # :152-153
    lt_new = lt.copy(); lt_new[R] = lt[R] - alpha * g_lt[R]  # :212,214
    di_new = di.copy(); di_new[S] = di[S] - alpha * g_di[S]  # :213,215
    N['lt'], N['di'] = lt_new, di_new
    return N, [los, sur, upq, ls]


# ----------------------------------------------------------------------------------------------
# a4: plain GRU + BPR step  (public/GRU.py:313-389)
# ----------------------------------------------------------------------------------------------
def gru_forward_cost(P, p, q, mask, lam):
    lt, ui, wh, bi = P['lt'], P['ui'], P['wh'], P['bi']
    L = int(np.sum(mask))
    xps, xqs = lt[p], lt[q]
    h = P['h0'].copy()
    tot = 0.0
    for t in range(L):                                       # :355-360, n_steps = L
        tot += log_sigmoid(h @ (xps[t] - xqs[t]))            # :352-353 uses h_{t-1}
        _, _, _, h = _gru_cell(ui, wh, bi, xps[t], h)
    l2 = sum(np.sum(v * v) for v in (xps, xqs, ui, wh, bi))
    return -tot + 0.5 * lam * l2, -tot


def _gru_seq_loss_grads(P, p, q, mask):
    """Gradients of -sum_t log sigmoid(h_{t-1} . (x_p - x_q)) of ONE sequence (no L2 terms): (dense grads, g_lt, sum of log-sigmoids).
    Shared by gru_step (public/GRU.py:313-385) and gru_minibatch_step (public/GRU.py:407-466): both scan the same recurrence."""
    p, q = np.asarray(p, np.int64), np.asarray(q, np.int64)
    lt, ui, wh, bi = P['lt'], P['ui'], P['wh'], P['bi']
    D = lt.shape[1]
    L = int(np.sum(mask))
    xps, xqs = lt[p], lt[q]
    hs = np.zeros((L + 1, D)); hs[0] = P['h0']
    zs = np.zeros((L, D)); rs = np.zeros((L, D)); cs = np.zeros((L, D)); us = np.zeros(L)
    tot = 0.0
    for t in range(L):
        us[t] = hs[t] @ (xps[t] - xqs[t])
        tot += log_sigmoid(us[t])
        zs[t], rs[t], cs[t], hs[t + 1] = _gru_cell(ui, wh, bi, xps[t], hs[t])
    g = dict(ui=np.zeros_like(ui), wh=np.zeros_like(wh), bi=np.zeros_like(bi))
    g_lt = np.zeros_like(lt)
    dh = np.zeros(D)                                         # d cost / d h_{L-1} = 0
    for t in range(L - 1, -1, -1):
        dx, dhp = _gru_cell_bwd(ui, wh, xps[t], hs[t], zs[t], rs[t], cs[t], dh, g)
        g_lt[p[t]] += dx
        gu = -sigmoid(-us[t])
        g_lt[p[t]] += gu * hs[t]
        g_lt[q[t]] -= gu * hs[t]
        dh = dhp + gu * (xps[t] - xqs[t])                    # gradient wrt h_{t-1}
    return g, g_lt, tot


def gru_step(P, p, q, mask, alpha, lam):
    """One ``OboGru.seq_train(uidx)`` (public/GRU.py:313-385).  Returns (P_new, -upq)."""
    p, q = np.asarray(p, np.int64), np.asarray(q, np.int64)
    lt, ui, wh, bi = P['lt'], P['ui'], P['wh'], P['bi']
    g, g_lt, tot = _gru_seq_loss_grads(P, p, q, mask)
    np.add.at(g_lt, p, lam * lt[p])                          # :365 all LM rows
    np.add.at(g_lt, q, lam * lt[q])
    N = dict(P)
    N['ui'] = ui - alpha * (g['ui'] + lam * ui)
    N['wh'] = wh - alpha * (g['wh'] + lam * wh)
    N['bi'] = bi - alpha * (g['bi'] + lam * bi)
    R = np.unique(np.concatenate((p, q)))                    # :329-331
    lt_new = lt.copy(); lt_new[R] = lt[R] - alpha * g_lt[R]  # :372-373
    N['lt'] = lt_new
    return N, -tot                                           # :380


# ----------------------------------------------------------------------------------------------
# f4: mini-batch ``Gru``  (public/GRU.py:395-498): a batch of users is ONE cost
# ----------------------------------------------------------------------------------------------
def gru_minibatch_forward_cost(P, p_rows, q_rows, masks, lam):
    """Forward only: (cost, -upq) of public/GRU.py:412-459.  The scan runs over the batch for seq_length = the longest valid length
    (:418, :450); the state is NOT masked, only the loss is (:446), so a shorter sequence keeps stepping on padding inputs without
    contributing to the cost.  L2: every gathered row of xps / xqs (all len_max positions of every sequence), ui, wh, and bi once
    (sum over the batch-broadcast bi divided by the batch size, :454-455).  Used by the autograd tests."""
    p_rows, q_rows, masks = np.asarray(p_rows, np.int64), np.asarray(q_rows, np.int64), np.asarray(masks)
    lt, ui, wh, bi = P['lt'], P['ui'], P['wh'], P['bi']
    B = p_rows.shape[0]
    seq_length = int(masks.sum(axis=1).max())
    h = np.tile(P['h0'], (B, 1))
    tot = 0.0
    for t in range(seq_length):
        xp, xq = lt[p_rows[:, t]], lt[q_rows[:, t]]
        upq = np.sum(h * (xp - xq), axis=1)                                  # :444
        tot += float(np.sum(np.array([log_sigmoid(u) for u in upq]) * masks[:, t]))      # :445-446
        hn = np.empty_like(h)
        for b in range(B):
            _, _, _, hn[b] = _gru_cell(ui, wh, bi, xp[b], h[b])             # :438-443
        h = hn
    l2 = sum(np.sum(v * v) for v in (lt[p_rows], lt[q_rows], ui, wh)) + np.sum(bi * bi)      # :453-455
    return -tot / B + 0.5 * lam * l2, -tot                                   # :457-459, :473


def gru_minibatch_step(P, p_rows, q_rows, masks, alpha, lam):
    """One ``Gru.seq_train(start_end)`` (public/GRU.py:407-485) on the padded rows of a batch of users.  Returns (P_new, -upq).
    Loss gradients are the per-sequence ones (the unmasked tail of a shorter sequence carries no gradient) averaged over the batch;
    the L2 term counts every gathered position; lt is written at the unique ids of the batch's padded rows (:429-431, :463)."""
    p_rows, q_rows, masks = np.asarray(p_rows, np.int64), np.asarray(q_rows, np.int64), np.asarray(masks)
    lt = P['lt']
    B = p_rows.shape[0]
    g = dict(ui=np.zeros_like(P['ui']), wh=np.zeros_like(P['wh']), bi=np.zeros_like(P['bi']))
    g_lt = np.zeros_like(lt)
    tot = 0.0
    for b in range(B):
        gb, glb, tb = _gru_seq_loss_grads(P, p_rows[b], q_rows[b], masks[b])
        for k in g:
            g[k] += gb[k] / B
        g_lt += glb / B
        tot += tb
    pf, qf = p_rows.ravel(), q_rows.ravel()
    np.add.at(g_lt, pf, lam * lt[pf])                        # :453 every gathered row, duplicates counted
    np.add.at(g_lt, qf, lam * lt[qf])
    N = dict(P)
    for k in g:
        N[k] = P[k] - alpha * (g[k] + lam * P[k])            # :460-461
    R = np.unique(np.concatenate((pf, qf)))                  # :429-430
    lt_new = lt.copy(); lt_new[R] = lt[R] - alpha * g_lt[R]  # :462-463
    N['lt'] = lt_new
    return N, -tot                                           # :473


# ----------------------------------------------------------------------------------------------
# f4: CA-RNN step / predict / scoring  (public/CA_RNN.py:46-227; flag 3 of prog_bpr_gru_spatial.py:141-151)
# ----------------------------------------------------------------------------------------------
def init_carnn_params(rng, n_item, n_dist, d):
    """public/GRU.py:58-63 (lt, h0) + public/CA_RNN.py:55-62 (M (H, D), wd (n_dist+1, H, D)); n_in == n_hidden."""
    u = lambda *s: rng.uniform(-0.5, 0.5, s)
    return dict(lt=u(n_item + 1, d), M=u(d, d), wd=u(n_dist + 1, d, d), h0=np.zeros(d))


def carnn_forward_cost(P, p, q, dp, dq, mask, lam):
    """Forward only: (cost, los) of public/CA_RNN.py:128-150.  Used by the autograd / finite-difference tests."""
    lt, M, wd = P['lt'], P['M'], P['wd']
    L = int(np.sum(mask))
    xps, xqs, wdps, wdqs = lt[p], lt[q], wd[dp], wd[dq]      # :117-121 (all LM rows / matrices)
    h = P['h0'].copy()
    tot = 0.0
    for t in range(L - 1):                                   # :138-142, n_steps = L-1
        h = sigmoid(M @ xps[t] + wdps[t] @ h)                # :131
        yp = (wdps[t + 1] @ h) @ (M @ xps[t + 1])            # :132
        yq = (wdqs[t + 1] @ h) @ (M @ xqs[t + 1])            # :133
        tot += log_sigmoid(yp - yq)                          # :134
    los = -tot                                               # :148
    l2 = sum(np.sum(v * v) for v in (xps, xqs, M, wdps, wdqs))   # :147
    return los + 0.5 * lam * l2, los


def carnn_step(P, p, q, dp, dq, mask, alpha, lam):
    """One ``OboCARNN.seq_train(uidx)`` (public/CA_RNN.py:105-170).  Returns (P_new, los); P is not modified
    (Theano `updates`: everything evaluated at the old values).  Backward hand-derived, checked against an independent
    float64 autograd of carnn_forward_cost (tests/test_oracle_autograd.py)."""
    p, q, dp, dq = (np.asarray(v, np.int64) for v in (p, q, dp, dq))
    lt, M, wd = P['lt'], P['M'], P['wd']
    D = lt.shape[1]
    L = int(np.sum(mask))
    ns = max(L - 1, 0)
    xps, xqs = lt[p], lt[q]
    hs = np.zeros((ns + 1, D)); hs[0] = P['h0']
    mp = np.zeros((ns, D)); mq = np.zeros((ns, D)); vp = np.zeros((ns, D)); vq = np.zeros((ns, D)); ys = np.zeros(ns)
    tot = 0.0
    for t in range(ns):
        hs[t + 1] = sigmoid(M @ xps[t] + wd[dp[t]] @ hs[t])
        h = hs[t + 1]
        mp[t], mq[t] = M @ xps[t + 1], M @ xqs[t + 1]
        vp[t], vq[t] = wd[dp[t + 1]] @ h, wd[dq[t + 1]] @ h
        ys[t] = vp[t] @ mp[t] - vq[t] @ mq[t]
        tot += log_sigmoid(ys[t])
    los = -tot
    g_M = np.zeros_like(M); g_lt = np.zeros_like(lt); g_wd = np.zeros_like(wd)
    dh_next = np.zeros(D)
    for t in range(ns - 1, -1, -1):
        h, hp = hs[t + 1], hs[t]
        a, b = dp[t + 1], dq[t + 1]
        g = -sigmoid(-ys[t])                                 # d cost / d (yp - yq)
        dh = dh_next + g * (wd[a].T @ mp[t] - wd[b].T @ mq[t])
        g_wd[a] += g * np.outer(mp[t], h)
        g_wd[b] -= g * np.outer(mq[t], h)
        g_M += g * (np.outer(vp[t], xps[t + 1]) - np.outer(vq[t], xqs[t + 1]))
        g_lt[p[t + 1]] += g * (M.T @ vp[t])
        g_lt[q[t + 1]] -= g * (M.T @ vq[t])
        da = dh * h * (1.0 - h)
        g_M += np.outer(da, xps[t])
        g_lt[p[t]] += M.T @ da
        g_wd[dp[t]] += np.outer(da, hp)
        dh_next = wd[dp[t]].T @ da
    # L2 over ALL LM gathered rows / matrices, multiplicity-weighted (:147)
    np.add.at(g_lt, p, lam * xps)
    np.add.at(g_lt, q, lam * xqs)
    np.add.at(g_wd, dp, lam * wd[dp])
    np.add.at(g_wd, dq, lam * wd[dq])
    N = dict(P)
    N['M'] = M - alpha * (g_M + lam * M)                     # :151-152
    R = np.unique(np.concatenate((p, q)))                    # :123-125
    S = np.unique(np.concatenate((dp, dq)))                  # :127-129
    lt_new = lt.copy(); lt_new[R] = lt[R] - alpha * g_lt[R]  # :153,155
    wd_new = wd.copy(); wd_new[S] = wd[S] - alpha * g_wd[S]  # :154,156
    N['lt'], N['wd'] = lt_new, wd_new
    return N, los                                            # :163


def carnn_predict(P, items, dists, p_rows, d_rows, masks):
    """public/CA_RNN.py:172-217, literally: h_t = sigmoid(p_t . M^T + sum(wd_t + h_{t-1}[None, :], axis = 2)), i.e. the
    ROW SUMS of the interval matrix plus the SUM of the previous state (the broadcast-add-then-sum of :191 - not the
    matrix-vector product of the training graph).  items / dists are the snapshots.  Returns hts (n, D)."""
    M = P['M']
    n = len(p_rows)
    H = M.shape[0]
    hts = np.zeros((n, H))
    for k in range(n):
        L = int(np.sum(masks[k]))
        h = P['h0'].copy()
        for t in range(L):
            wd_t = dists[d_rows[k][t]]                        # (H, D)
            h = sigmoid(items[p_rows[k][t]] @ M.T + np.sum(wd_t + h[None, :], axis=1))
        hts[k] = h
    return hts


def carnn_score_all(users, items, M, dists, ulptai_rows):
    """OboCARNN.compute_sub_all_scores (public/CA_RNN.py:91-101), literally:
    h_W[u, j, i] = sum_k (dists[ulptai[u, j]][i, k] + users[u, k]);  r_M[j, i] = (items[j] . M^T)[i];
    score[u, j] = - sum_i (h_W[u, j, i] + r_M[j, i]).  items includes the padding row (dropped)."""
    users = np.asarray(users, F64); it = np.asarray(items, F64)[:-1]
    Wsum = np.asarray(dists, F64).sum(axis=(1, 2))           # (n_dist + 1,)
    H = M.shape[0]
    m = (it @ M.T).sum(axis=1)                               # (N,)
    return -(Wsum[np.asarray(ulptai_rows)] + H * users.sum(axis=1)[:, None] + m[None, :])


# ----------------------------------------------------------------------------------------------
# a5: BPR-MF step  (public/BPR.py:201-237)
# ----------------------------------------------------------------------------------------------
def bpr_step(P, uidx, pi, qi, alpha, lam):
    """One ``bpr_train(uidx, [p, q])``.  Returns (P_new, -log sigmoid(u))."""
    ux, lt = P['ux'], P['lt']
    usr, xp, xq = ux[uidx], lt[pi], lt[qi]
    u = usr @ (xp - xq)                                      # :216
    g = -sigmoid(-u)
    N = dict(P)
    ux_new, lt_new = ux.copy(), lt.copy()
    ux_new[uidx] = usr - alpha * (g * (xp - xq) + lam * usr)     # :226-228
    # set_subtensor on lt[[p,q]] with both rows evaluated at old values
    lt_new[pi] = xp - alpha * (g * usr + lam * xp)
    lt_new[qi] = xq - alpha * (-g * usr + lam * xq)
    N['ux'], N['lt'] = ux_new, lt_new
    return N, -log_sigmoid(u)                                # :236


# ----------------------------------------------------------------------------------------------
# a6: predict (batched GRU forward over the whole train sequence)
# ----------------------------------------------------------------------------------------------
def spatial_predict(P, items, dists, p_rows, d_rows, masks):
    """public/GRU_Spatial.py:231-288.  items/dists are the *snapshots* trained_items /
    trained_dists.  Returns (hts (n,D), sts (n,B+1))."""
    ui, wh, bi, vs, bs = P['ui'], P['wh'], P['bi'], P['vs'], P['bs']
    n = len(p_rows)
    D = items.shape[1]
    hts = np.zeros((n, D)); sts = np.zeros((n, vs.shape[0]))
    for k in range(n):
        L = int(np.sum(masks[k]))
        h = P['h0'].copy()
        for t in range(L):                                   # h index L-1 (:274-276)
            x = np.concatenate((items[p_rows[k][t]], dists[d_rows[k][t]]))
            _, _, _, h = _gru_cell(ui, wh, bi, x, h)
        hts[k] = h
        sts[k] = softmax0(vs @ h + bs)                       # :278
    return hts, sts


def gru_predict(P, items, p_rows, masks):
    """public/GRU.py:154-202.  Returns hts (n,D)."""
    ui, wh, bi = P['ui'], P['wh'], P['bi']
    n = len(p_rows)
    hts = np.zeros((n, items.shape[1]))
    for k in range(n):
        L = int(np.sum(masks[k]))
        h = P['h0'].copy()
        for t in range(L):
            _, _, _, h = _gru_cell(ui, wh, bi, items[p_rows[k][t]], h)
        hts[k] = h
    return hts


# ----------------------------------------------------------------------------------------------
# a8 / a10: scoring and AUC preference
# ----------------------------------------------------------------------------------------------
def score_all(users, items, wd=None, prob=None):
    """public/GRU.py:93-96, public/BPR.py:76-79; spatial adds wd*prob
    (public/GRU_Spatial.py:117-125).  items includes the padding row (dropped here)."""
    sc = users @ items[:-1].T
    if prob is not None:
        sc = sc + wd * prob
    return sc


def auc_preference(users, items, tes_p, tes_q, tes_mask):
    """public/GRU.py:98-110."""
    d = items[np.asarray(tes_p)] - items[np.asarray(tes_q)]
    upq = np.einsum('nd,nld->nl', users, d) * np.asarray(tes_mask)
    return upq > 0


# ----------------------------------------------------------------------------------------------
# a9: top-K  (public/Valuate.py:91-100, loop :132-146)
# ----------------------------------------------------------------------------------------------
def topk_desc(scores, k):
    """Indices of the k largest scores per row, sorted by descending score.  The reference
    (argpartition + reversed argsort) leaves the order of exact ties unspecified; the build's
    rule - ties by ascending index - is stated here and used by the HIP kernel.  On tie-free
    rows this equals the reference helpers (golden-checked)."""
    scores = np.asarray(scores)
    out = np.empty((scores.shape[0], k), np.int64)
    for i, row in enumerate(scores):
        order = np.lexsort((np.arange(row.size), -row))
        out[i] = order[:k]
    return out


# ----------------------------------------------------------------------------------------------
# metrics  (public/Valuate.py:23-88, 149-172)
# ----------------------------------------------------------------------------------------------
def hit_zero_one(test_lst, recom_lst, test_mask):
    """public/Valuate.py:23-40."""
    t = list(test_lst[:int(np.sum(test_mask))])
    return np.array([1 if e in t else 0 for e in recom_lst])


def evaluate_map(test_lst, zero_one, test_mask):
    """public/Valuate.py:43-63."""
    n_t = int(np.sum(test_mask))
    zo = np.array(zero_one)
    if zo.sum() == 0:
        return 0.0
    cum = zo.cumsum() * zo
    s = sum(1.0 * cum[i] / (i + 1) for i in np.nonzero(cum)[0])
    return s / n_t


def evaluate_ndcg(test_lst, zero_one, test_mask):
    """public/Valuate.py:66-88."""
    n_t = int(np.sum(test_mask))
    zo = np.array(zero_one)
    if zo.sum() == 0:
        return 0.0
    s = sum(1.0 / np.log2(i + 2) for i in np.nonzero(zo)[0])
    m = sum(1.0 / np.log2(i + 2) for i in range(min(n_t, len(zo))))
    return s / m


def evaluate_ranks(all_ranks, tes_buys_masks, tes_masks, at_nums):
    """public/Valuate.py:149-172: recall/precision/F1/MAP/NDCG @ each k."""
    out = {}
    denom = float(np.sum(tes_masks))
    for k in at_nums:
        zo = np.array([hit_zero_one(t, r[:k], m) for t, r, m in zip(tes_buys_masks, all_ranks, tes_masks)])
        hits = float(zo.sum())
        rec = hits / denom
        pre = hits / (k * len(zo))
        f1 = 2.0 * rec * pre / (rec + pre) if rec + pre > 0 else 0.0
        mp = float(np.mean([evaluate_map(t, z, m) for t, z, m in zip(tes_buys_masks, zo, tes_masks)]))
        nd = float(np.mean([evaluate_ndcg(t, z, m) for t, z, m in zip(tes_buys_masks, zo, tes_masks)]))
        out[k] = dict(hits=hits, recall=rec, precision=pre, f1=f1, map=mp, ndcg=nd)
    return out


# ----------------------------------------------------------------------------------------------
# a11: input contract  (public/Load_Data_by_length.py)
# ----------------------------------------------------------------------------------------------
def cal_dis(lat1, lon1, lat2, lon2, dd, dist_num):
    """public/Load_Data_by_length.py:24-42 - Haversine ((1-cos)/2 form) -> distance bin."""
    d = 12742
    p = 0.017453292519943295
    a = (lat1 - lat2) * p
    b = (lon1 - lon2) * p
    c = (1.0 - math.cos(a)) / 2 + math.cos(lat1 * p) * math.cos(lat2 * p) * (1.0 - math.cos(b)) / 2
    dist = d * math.asin(math.sqrt(c))
    return min(int(dist * 1000 / dd), dist_num)


def data_buys_masks(all_usr_pois, all_usr_dist, item_tail, dist_tail):
    """public/Load_Data_by_length.py:115-124."""
    lens = [len(u) for u in all_usr_pois]
    lm = max(lens)
    pois = [list(u) + item_tail * (lm - le) for u, le in zip(all_usr_pois, lens)]
    dist = [list(u) + dist_tail * (lm - le) for u, le in zip(all_usr_dist, lens)]
    msks = [[1] * le + [0] * (lm - le) for le in lens]
    return pois, dist, msks


def compute_dist_neg(tra_buys_masks, tra_masks, tra_buys_neg_masks, pois_cordis, dd, dist_num):
    """public/Load_Data_by_length.py:165-180."""
    out = []
    for up, um, un in zip(tra_buys_masks, tra_masks, tra_buys_neg_masks):
        L = int(sum(um))
        dist = []
        for i in range(1, L):
            pre = pois_cordis[up[i - 1]]
            cur = pois_cordis[un[i]]
            dist.append(cal_dis(cur[0], cur[1], pre[0], pre[1], dd, dist_num))
        out.append([dist_num] + dist + [dist_num] * (len(up) - L))
    return out


def compute_distance(tra_pois_masks, tra_masks, pois_cordis, dd, dist_num):
    """public/Load_Data_by_length.py:183-215 - last train POI -> all POIs distance bins."""
    arr = np.asarray(tra_pois_masks)
    last = arr[np.arange(len(arr)), np.sum(tra_masks, axis=1) - 1]
    out = np.empty((len(arr), len(pois_cordis)), np.int64)
    for u, poi in enumerate(last):
        lc = pois_cordis[poi]
        for j, c in enumerate(pois_cordis):
            out[u, j] = cal_dis(lc[0], lc[1], c[0], c[1], dd, dist_num)
    return out


def acquire_prob(all_sus, ulptai, dist_num):
    """public/Load_Data_by_length.py:218-235 (not runnable under Python 3 - restated)."""
    all_sus = np.asarray(all_sus); ulptai = np.asarray(ulptai)
    return np.take_along_axis(all_sus, ulptai, axis=1) * (ulptai < dist_num)


def l2_value(P, lam, names):
    """model.l2.eval(): 0.5*lambda*sum of squares (public/GRU_Spatial.py:83-88, GRU.py:304-308,
    BPR.py:194-197)."""
    tot = 0.0
    for n in names:
        v = np.asarray(P[n], F64)
        tot += float(np.sum(v * v))
    return 0.5 * lam * tot
