"""Oracle vs the committed golden vectors (tests/golden/*.npz, produced by make_golden.py from the
reference's own numpy helpers).  CPU only."""
import os

import numpy as np

from oracle import poi_oracle as O


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_topk_matches_reference_helpers(golden_dir):
    g = _g(golden_dir, "topk.npz")
    assert np.array_equal(O.topk_desc(g["scores"], int(g["k"])), g["ranks"])
    assert np.array_equal(O.topk_desc(g["scores"], 5), g["ranks5"])


def test_metrics_match_reference_helpers(golden_dir):
    g = _g(golden_dir, "metrics.npz")
    for t, r, m, zo, mp, nd in zip(g["test_lst"], g["recom"], g["test_mask"], g["zero_one"], g["map"], g["ndcg"]):
        z = O.hit_zero_one(t, r, m)
        assert np.array_equal(z, zo)
        assert O.evaluate_map(t, z, m) == mp
        assert O.evaluate_ndcg(t, z, m) == nd


def test_cal_dis_matches_reference(golden_dir):
    g = _g(golden_dir, "cal_dis.npz")
    b200 = [O.cal_dis(a, b, c, d, 200, 200) for a, b, c, d in zip(g["lat1"], g["lon1"], g["lat2"], g["lon2"])]
    b25 = [O.cal_dis(a, b, c, d, 25, 1520) for a, b, c, d in zip(g["lat1"], g["lon1"], g["lat2"], g["lon2"])]
    assert np.array_equal(b200, g["bins_dd200_B200"])
    assert np.array_equal(b25, g["bins_dd25_B1520"])
    assert all(x == 0 for x in b200[:10]) and all(x == 200 for x in b200[10:20])


def test_masks_and_bins_match_reference(golden_dir):
    g = _g(golden_dir, "masks.npz")
    N, B, dd = int(g["n_item"]), int(g["n_dist"]), int(g["dd"])
    lens = g["lens"]
    off = np.concatenate(([0], np.cumsum(lens)))
    tra = [list(g["ragged_pois"][off[i]:off[i + 1]]) for i in range(len(lens))]
    trd = [list(g["ragged_dist"][off[i]:off[i + 1]]) for i in range(len(lens))]
    pm, dm, mk = O.data_buys_masks(tra, trd, [N], [B])
    assert np.array_equal(pm, g["pois_m"]) and np.array_equal(dm, g["dist_m"]) and np.array_equal(mk, g["msks"])
    cordis = [list(c) for c in g["coords"]]
    dn = O.compute_dist_neg(g["pois_m"], g["msks"], g["negs"], cordis, dd, B)
    assert np.array_equal(dn, g["dist_neg"])
    assert np.array_equal(O.compute_distance(g["pois_m"], g["msks"], cordis, dd, B), g["ulptai"])
    # sampler contract (public/Load_Data_by_length.py:127-143): pads stay pads, negatives avoid the user's items
    for prow, nrow, mrow in zip(g["pois_m"], g["negs"], g["msks"]):
        L = int(mrow.sum())
        assert np.all(nrow[L:] == N) and not set(nrow[:L]) & set(prow[:L])


def test_step_vectors_do_not_drift(golden_dir):
    g = _g(golden_dir, "spatial_step.npz")
    P = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    P["wd"] = float(P["wd"])
    Pn, out = O.spatial_step(P, g["p"], g["q"], g["dp"], g["dq"], g["mask"], float(g["alpha"]), float(g["lam"]))
    for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"):
        assert np.allclose(np.asarray(Pn[k]), g["out_" + k], rtol=1e-13, atol=1e-15), k
    assert np.isclose(out[0], g["los"], rtol=1e-13) and np.allclose(out[3], g["ls"], rtol=1e-13)
    g = _g(golden_dir, "gru_step.npz")
    P = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    Pn, loss = O.gru_step(P, g["p"], g["q"], g["mask"], float(g["alpha"]), float(g["lam"]))
    for k in ("lt", "ui", "wh", "bi"):
        assert np.allclose(Pn[k], g["out_" + k], rtol=1e-13, atol=1e-15), k
    assert np.isclose(loss, g["loss"], rtol=1e-13)
