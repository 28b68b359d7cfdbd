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
