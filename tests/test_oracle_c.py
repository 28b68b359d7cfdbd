"""The plain-C restatement (cpu_baseline of bench.py) must agree with the numpy oracle."""
import numpy as np

from oracle import c_oracle as C
from oracle import poi_oracle as O
from poi_amd.data import padded_to_csr
from tests.gpu_util import spatial_params, toy_problem


def test_c_spatial_epoch_matches_numpy_oracle():
    T = toy_problem(77, n_user=6, n_item=40, n_dist=9, dim=8, len_max=9)
    P = spatial_params(77, T)
    Pm, Mm, Qm = T["train"]; DPm, _, DQm = T["dist"]
    order = [4, 0, 2, 4, 5]
    Pn = {k: (np.array(v, np.float64, copy=True) if not np.isscalar(v) else v) for k, v in P.items()}
    outs = []
    for u in order:
        Pn, out = O.spatial_step(Pn, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        outs.append([out[0], out[1], out[2], out[3][0], out[3][1]])
    lens = Mm.sum(1)
    off, p = padded_to_csr(Pm, lens); _, q = padded_to_csr(Qm, lens)
    _, dp = padded_to_csr(DPm, lens); _, dq = padded_to_csr(DQm, lens)
    Pc = {k: (np.array(v, np.float64, copy=True) if not np.isscalar(v) else v) for k, v in P.items()}
    got = C.spatial_epoch(Pc, off, p, q, dp, dq, order, T["len_max"], 0.01, 0.001)
    assert np.allclose(got, np.array(outs), rtol=1e-11, atol=1e-13)
    for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"):
        assert np.allclose(np.asarray(Pc[k]), np.asarray(Pn[k]), rtol=1e-10, atol=1e-13), k


def test_c_score_topk_matches_numpy_oracle():
    rng = np.random.default_rng(1)
    users = rng.normal(size=(7, 16)); items = rng.normal(size=(500, 16))
    exp = O.topk_desc(users @ items.T, 20)
    assert np.array_equal(C.score_topk(users, items, 20), exp)


def test_c_batch_rule_matches_numpy_emulation():
    """poi_oracle_spatial_batch (the full-size checker of the tile engine) == the numpy oracle run per sequence +
    the mean-of-touching-sequences rule (tests/gpu_util.batch_mean_update), threaded or not."""
    from tests.gpu_util import batch_mean_update
    T = toy_problem(5, n_user=40, n_item=60, n_dist=11, dim=8, len_max=10)
    P = spatial_params(5, T)
    lens = T["lens"]
    off, p = padded_to_csr(T["train"][0], lens); _, q = padded_to_csr(T["train"][2], lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    ids = np.random.default_rng(0).permutation(40)[:37]
    per, touched, outs = [], [], []
    for u in ids:
        Pn, out = O.spatial_step(P, T["train"][0][u], T["train"][2][u], T["dist"][0][u], T["dist"][2][u], T["train"][1][u], 0.01, 0.001)
        per.append(Pn); outs.append([out[0], out[1], out[2]])
        touched.append(dict(lt=np.unique(np.concatenate((T["train"][0][u], T["train"][2][u]))), di=np.unique(T["dist"][0][u])))
    E = batch_mean_update(P, per, touched, ("lt", "di"), ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"))
    for threads in (1, 3):
        N, out, tch = C.spatial_batch_mean(P, off, p, q, dp, dq, ids, 10, 0.01, 0.001, threads=threads)
        assert np.allclose(out[:, :3], np.array(outs), rtol=1e-11, atol=1e-13)
        for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"):
            assert np.allclose(np.asarray(N[k]), np.asarray(E[k]), rtol=1e-11, atol=1e-14), k
        assert np.array_equal(np.nonzero(tch["lt"])[0], np.unique(np.concatenate([t["lt"] for t in touched])))


def test_delta_tolerance_bites_where_the_weight_tolerance_does_not():
    """tests/gpu_util.assert_delta_close vs assert_close on a float32-rounded oracle result: an off-by-one in the
    padding rows' L2 multiplicity passes the 1e-5 max-norm bar and fails the per-row delta bar."""
    from tests.gpu_util import RTOL, delta_excess, rel_err
    T = toy_problem(13, n_user=4, n_item=300, n_dist=200, dim=64, len_max=12)
    P = spatial_params(13, T)
    u = 2
    Pm, Qm, DPm, DQm, Mm = (np.asarray(T["train"][0]), np.asarray(T["train"][2]), np.asarray(T["dist"][0]), np.asarray(T["dist"][2]),
                            np.asarray(T["train"][1]))
    exp, _ = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
    ap = lambda row, v: np.append(row, v)
    bad, _ = O.spatial_step(P, ap(Pm[u], T["n_item"]), ap(Qm[u], T["n_item"]), ap(DPm[u], T["n_dist"]), ap(DQm[u], T["n_dist"]), ap(Mm[u], 0),
                            0.01, 0.001)
    for k in ("lt", "di"):
        dev = np.asarray(exp[k], np.float32).astype(np.float64)          # what a correct float32 device would hold
        assert delta_excess(dev, exp[k], P[k])[0] <= 1.0
        if k == "di":         # one extra padding bin: alpha*lambda*|row| <= 5e-6 absolute, invisible at 1e-5 of max|theta| ~ 0.5
            assert rel_err(dev, bad[k]) <= RTOL
        assert delta_excess(dev, bad[k], P[k])[0] > 3.0
