"""-m gpu: the BPR-MF step (OboBpr.bpr_train, public/BPR.py:201-241) in SNAPSHOT mode on launches whose triples SHARE rows - the sorted,
atomic-free path of csrc/bpr.hip (round 5) - against the float64 oracle's per-triple step (oracle.bpr_step) combined by the batch rule of
include/poi_hip.h: a row touched by k triples moves by min(k, cap) / k of the sum of their reference updates.  Hot users and hot POIs give runs
longer than one 64-touch window (the opening / closing partial sums and their fixed-order combination), dims 32 .. 320 cover every lane
layout; identical launches must give bitwise identical tables; a half POI table == half(oracle from the half-rounded table) to one ulp."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, round_f32, toy_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    return poi_amd


def _triples(seed, n, n_user, n_item, hot_users, hot_items):
    rng = np.random.default_rng(seed)
    u = np.where(rng.random(n) < 0.5, rng.integers(0, hot_users, n), rng.integers(0, n_user, n))
    p = np.where(rng.random(n) < 0.5, rng.integers(0, hot_items, n), rng.integers(0, n_item, n))
    q = rng.integers(0, n_item, n)
    q = np.where(q == p, (q + 1) % n_item, q)
    return u.astype(np.int32), p.astype(np.int32), q.astype(np.int32)


def _expected(P, u, p, q, alpha, lam, cap):
    """batch rule on top of the oracle's single-triple step, every triple at the entry values"""
    acc = {k: np.zeros_like(P[k]) for k in ("ux", "lt")}
    cnt = {k: np.zeros(P[k].shape[0]) for k in ("ux", "lt")}
    losses = []
    for ui, pi, qi in zip(u, p, q):
        Pn, l = O.bpr_step(P, int(ui), int(pi), int(qi), alpha, lam)
        losses.append(l)
        for name, rows in (("ux", [ui]), ("lt", [pi, qi])):
            for r in rows:
                acc[name][r] += Pn[name][r] - P[name][r]; cnt[name][r] += 1
    out = {}
    for name in ("ux", "lt"):
        k = np.maximum(cnt[name], 1)
        out[name] = P[name] + acc[name] * (np.minimum(k, cap) / k)[:, None]
    return out, np.asarray(losses)


@pytest.mark.parametrize("dim,cap", [(32, 1.0), (64, 8.0), (128, 64.0), (256, 8.0), (320, 1e9), (20, 4.0)])
def test_bpr_snapshot_with_shared_rows_matches_the_batch_rule(pa, dim, cap):
    n_user, n_item, n = 120, 300, 3000
    T = toy_problem(50, n_user=n_user, n_item=n_item, dim=dim)
    P = round_f32(O.init_bpr_params(np.random.default_rng(7), n_user, n_item, dim))
    u, p, q = _triples(dim, n, n_user, n_item, hot_users=3, hot_items=4)      # ~500 touches on a hot user, ~400 on a hot POI: runs over many windows
    exp, el = _expected(P, u, p, q, 0.01, 0.001, cap)
    m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=dim, n_hidden=dim, init=P)
    m.ctx.set_batch_cap(cap)
    try:
        got_l = m.train_batch(u, p, q)
    finally:
        m.ctx.set_batch_cap(1.0)
    assert_close(got_l, el, "losses")
    assert_step_close({k: getattr(m, k).get_value() for k in ("ux", "lt")}, exp, P, ("ux", "lt"), "dim %d cap %g" % (dim, cap))
    # rows no triple touches are bit-identical
    lt = m.lt.get_value()
    untouched = np.setdiff1d(np.arange(n_item + 1), np.concatenate((p, q)))
    assert np.array_equal(lt[untouched], np.asarray(P["lt"], np.float32)[untouched])


def test_bpr_snapshot_is_bitwise_reproducible_and_order_defined(pa):
    import torch
    dim, n_user, n_item, n = 128, 500, 2000, 40000
    T = toy_problem(51, n_user=n_user, n_item=n_item, dim=dim)
    P = round_f32(O.init_bpr_params(np.random.default_rng(8), n_user, n_item, dim))
    u, p, q = _triples(3, n, n_user, n_item, hot_users=5, hot_items=6)
    runs = []
    for _ in range(3):
        m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=dim, n_hidden=dim, init=P)
        m.ctx.set_batch_cap(64.0)
        try:
            l = m.train_batch(u, p, q, sync=False)
            m.train_batch(u, p, q, sync=False)      # (a second launch on the moved tables: the workspace is reused)
        finally:
            m.ctx.set_batch_cap(1.0)
        runs.append((m.ux.t.clone(), m.lt.t.clone(), l.clone()))
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2])
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all()


@pytest.mark.parametrize("rounding", ["nearest", "stochastic"])
def test_bpr_snapshot_half_poi_table(pa, rounding):
    dim, n_user, n_item, n = 64, 60, 200, 1500
    T = toy_problem(52, n_user=n_user, n_item=n_item, dim=dim)
    P = round_f32(O.init_bpr_params(np.random.default_rng(9), n_user, n_item, dim))
    P["lt"] = np.asarray(P["lt"], np.float16).astype(np.float64)      # the stored values ARE halves
    u, p, q = _triples(4, n, n_user, n_item, hot_users=2, hot_items=3)
    exp, el = _expected(P, u, p, q, 0.01, 0.001, 8.0)
    m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=dim, n_hidden=dim, init=P,
                         table_dtype="f16")
    m.ctx.set_batch_cap(8.0); m.ctx.set_f16_rounding(rounding, seed=3)
    try:
        got_l = m.train_batch(u, p, q)
    finally:
        m.ctx.set_batch_cap(1.0); m.ctx.set_f16_rounding("nearest", seed=1)
    assert_close(got_l, el, "losses")
    assert_close(m.ux.get_value(), exp["ux"], "ux")
    lt = m.lt.get_value().astype(np.float64)
    want = np.asarray(exp["lt"], np.float16).astype(np.float64)
    ulp = np.spacing(np.abs(want).astype(np.float16)).astype(np.float64)
    assert (np.abs(lt - want) <= ulp).all()                    # one of the two half neighbours of the oracle's value, either rounding
    if rounding == "nearest":
        assert (lt == want).mean() > 0.97
    else:                                                      # stochastic: rows that moved land on both neighbours; untouched rows stay bit-identical
        untouched = np.setdiff1d(np.arange(n_item + 1), np.concatenate((p, q)))
        assert np.array_equal(lt[untouched], np.asarray(P["lt"], np.float64)[untouched])
        assert 0.2 < (lt == want).mean() < 1.0


def _expected_vec(P, u, p, q, alpha, lam, cap):
    """_expected, vectorised (np.add.at over the triples) - checked against the oracle loop below before it is used at sizes the loop cannot reach"""
    ux, lt = np.asarray(P["ux"], np.float64), np.asarray(P["lt"], np.float64)
    d = lt[p] - lt[q]
    m = np.einsum("nd,nd->n", ux[u], d)
    g = -1.0 / (1.0 + np.exp(m))
    loss = np.logaddexp(0.0, -m)
    G = {"ux": np.zeros_like(ux), "lt": np.zeros_like(lt)}
    C = {"ux": np.zeros(ux.shape[0]), "lt": np.zeros(lt.shape[0])}
    np.add.at(G["ux"], u, g[:, None] * d); np.add.at(C["ux"], u, 1)
    np.add.at(G["lt"], p, g[:, None] * ux[u]); np.add.at(C["lt"], p, 1)
    np.add.at(G["lt"], q, -g[:, None] * ux[u]); np.add.at(C["lt"], q, 1)
    out = {}
    for name, T in (("ux", ux), ("lt", lt)):
        k = C[name]
        sc = np.where(k > 0, alpha * np.minimum(k, cap), 0.0)
        out[name] = T - sc[:, None] * (G[name] / np.maximum(k, 1)[:, None] + lam * T)
    return out, loss


def test_bpr_snapshot_fuzz_sizes_and_dims(pa):
    """random launch sizes 1 .. 60000, dims 4 .. 1024 (every lane layout and column-pass count), user / POI tables small against the launch (long runs
    over many windows) and large (mostly single touches), caps 1 .. inf - against the vectorised batch rule (== the oracle loop, checked first)"""
    rng = np.random.default_rng(2026)
    T0 = toy_problem(60, n_user=9, n_item=40, dim=8)
    P0 = round_f32(O.init_bpr_params(np.random.default_rng(1), 9, 40, 8))
    u0, p0, q0 = _triples(1, 200, 9, 40, 3, 4)
    a, la = _expected(P0, u0, p0, q0, 0.01, 0.001, 4.0); b, lb = _expected_vec(P0, u0, p0, q0, 0.01, 0.001, 4.0)
    assert np.allclose(a["ux"], b["ux"], rtol=1e-12, atol=1e-14) and np.allclose(a["lt"], b["lt"], rtol=1e-12, atol=1e-14) and np.allclose(la, lb, rtol=1e-12)
    for it in range(24):
        dim = int(rng.choice([4, 20, 32, 64, 100, 128, 192, 256, 320, 512, 1024]))
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 20000, 60000]))
        if dim >= 512:
            n = min(n, 4097)
        n_user = int(rng.choice([3, 50, 3000])); n_item = int(rng.choice([5, 80, 5000]))
        cap = float(rng.choice([1.0, 3.0, 64.0, 1e9]))
        T = toy_problem(61 + it, n_user=n_user, n_item=n_item, dim=dim, hot=min(8, n_item))      # (only the model's bookkeeping tables: the triples come from below)
        P = round_f32(O.init_bpr_params(np.random.default_rng(100 + it), n_user, n_item, dim))
        u, p, q = _triples(200 + it, n, n_user, n_item, hot_users=max(1, n_user // 10), hot_items=max(2, n_item // 10))
        exp, el = _expected_vec(P, u, p, q, 0.01, 0.001, cap)
        m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=dim, n_hidden=dim, init=P)
        m.ctx.set_batch_cap(cap)
        try:
            got_l = m.train_batch(u, p, q)
        finally:
            m.ctx.set_batch_cap(1.0)
        what = "fuzz %d: dim %d n %d users %d items %d cap %g" % (it, dim, n, n_user, n_item, cap)
        assert_close(got_l, el, "losses " + what)
        assert_step_close({k: getattr(m, k).get_value() for k in ("ux", "lt")}, exp, P, ("ux", "lt"), what)


@pytest.mark.parametrize("mode", ["snapshot", "hogwild"])
def test_out_of_range_ids_raise_like_the_reference_and_move_nothing(pa, mode):
    """ADVICE r5 / include/poi_hip.h ABI 6: the reference's gather raises IndexError on an id outside its table (public/BPR.py:214-218).  Here the
    triple gets no gradient and a NaN loss, nothing is written outside the tables, the device counts it and the Python mirror raises IndexError."""
    import torch
    n_user, n_item, D = 40, 90, 64
    T = toy_problem(3, n_user=n_user, n_item=n_item, n_dist=10, dim=D, len_max=6)
    m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=D, n_hidden=D, seed=4)
    u, p, q = _triples(9, 300, n_user, n_item, 5, 9)
    guard_u, guard_i = m.ux.t.clone(), m.lt.t.clone()
    for bad in ((7, n_user, p[7], q[7]), (11, u[11], n_item + 1, q[11]), (13, u[13], p[13], -2)):
        ub, pb, qb = u.copy(), p.copy(), q.copy()
        ub[bad[0]], pb[bad[0]], qb[bad[0]] = bad[1], bad[2], bad[3]
        m.ux.t.copy_(guard_u); m.lt.t.copy_(guard_i)
        loss = m.train_batch(ub, pb, qb, mode=mode, sync=False)
        torch.cuda.synchronize()
        loss = loss.cpu().numpy()
        assert np.isnan(loss[bad[0]]) and np.isfinite(np.delete(loss, bad[0])).all()
        assert torch.isfinite(m.ux.t).all() and torch.isfinite(m.lt.t).all()
        assert m.ctx.take_bad_ids() >= 1 and m.ctx.take_bad_ids() == 0          # counted, then cleared
        m.ux.t.copy_(guard_u); m.lt.t.copy_(guard_i)
        with pytest.raises(IndexError):
            m.train_batch(ub, pb, qb, mode=mode)
    # a clean launch afterwards is the clean result
    m.ux.t.copy_(guard_u); m.lt.t.copy_(guard_i)
    assert np.isfinite(m.train_batch(u, p, q, mode=mode)).all()
