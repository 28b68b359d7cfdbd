"""-m gpu: CA-RNN (flag 3, public/CA_RNN.py:46-227) through the model class -> ctypes C-ABI -> carnn.hip, against the
float64 oracle (oracle/poi_oracle.py carnn_*, itself checked against autograd): the per-user step with its sparse
write-back of POI rows and interval matrices, the batch rule, the literal predict / scoring graphs, top-K ranks."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, batch_mean_update, round_f32, toy_problem

pytestmark = pytest.mark.gpu
NAMES = ("lt", "wd", "M")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    poi_amd._lib.load()
    return poi_amd


def _params(seed, T):
    rng = np.random.default_rng(seed + 3000)
    return round_f32(O.init_carnn_params(rng, T["n_item"], T["n_dist"], T["dim"]))


def _model(pa, T, P, coords=None, alpha=0.01):
    return pa.models.OboCARNN(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[alpha, 0.001], n_user=T["n_user"],
                              n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=T["dim"], n_hidden=T["dim"], init=P, coords=coords)


def _get(m):
    return {k: np.asarray(getattr(m, k).get_value(), np.float64) for k in NAMES}


@pytest.mark.parametrize("seed,dim,n_dist", [(0, 20, 11), (1, 64, 37), (2, 128, 200)])
def test_carnn_step_parity_sequential(pa, seed, dim, n_dist):
    """model.train(uidx) user after user == OboCARNN.seq_train (prog_bpr_gru_spatial.py:246-247): loss and all three
    tensors at every step, both the 1e-5 weight bar and the per-row update bar (an interval matrix is one row)."""
    T = toy_problem(seed + 500, n_user=5, n_item=80, n_dist=n_dist, dim=dim, len_max=9)
    P = _params(seed, T)
    model = _model(pa, T, P)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    for u in [3, 0, 4, 3]:
        old = P
        P, los = O.carnn_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        got_los = model.train(np.int32(u))
        assert_close(got_los, los, "los")
        got = _get(model)
        flat = lambda d: {k: (v.reshape(v.shape[0], -1) if k == "wd" else v) for k, v in d.items()}
        assert_step_close(flat(got), flat(P), flat(old), NAMES, "after user %d" % u)
        P = round_f32({**P, **got})


@pytest.mark.parametrize("dim,n_dist,min_len", [(32, 11, 4), (64, 11, 4), (128, 200, 4), (64, 700, 4), (64, 11, 1), (128, 37, 1)])
def test_carnn_batch_matches_the_batch_rule(pa, dim, n_dist, min_len):
    """(dim 32: per-sequence kernel; 64 / 128: outer-product path; min_len 1: sequences of one and two positions - no step at all /
    a single step - mixed into the launch)"""
    T = toy_problem(520, n_user=40, n_item=90, n_dist=n_dist, dim=dim, len_max=10, hot=25, min_len=min_len)
    assert min_len > 1 or ((T["lens"] == 1).any() and (T["lens"] == 2).any())
    P = _params(520, T)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    users = np.random.default_rng(1).permutation(40)[:37].astype(np.int32)
    news, touched, losses = [], [], []
    for u in users:
        Pn, los = O.carnn_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); losses.append(los)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), wd=np.unique(np.concatenate((DPm[u], DQm[u])))))
    exp = batch_mean_update(P, news, touched, ("lt",), ("M",))
    # interval matrices: same rule with (n_dist + 1) "rows" of H x D
    acc = np.zeros_like(P["wd"]); cnt = np.zeros(P["wd"].shape[0])
    for Pn, tch in zip(news, touched):
        acc[tch["wd"]] += Pn["wd"][tch["wd"]] - P["wd"][tch["wd"]]; cnt[tch["wd"]] += 1
    exp["wd"] = P["wd"] + acc / np.maximum(cnt, 1)[:, None, None]
    model = _model(pa, T, P)
    got_los = model.train_batch(users)
    assert_close(got_los, losses, "losses", rtol=2e-5)
    got = _get(model)
    flat = lambda d: {k: (np.asarray(v).reshape(np.asarray(v).shape[0], -1) if k == "wd" else np.asarray(v)) for k, v in d.items() if k in NAMES}
    assert_step_close(flat(got), flat(exp), flat(P), NAMES, "batch")
    # second launch on the updated state: gradient tables / slabs were re-zeroed
    model.train_batch(users[:9])
    assert np.isfinite(_get(model)["wd"]).all()
    if dim >= 64:      # every gradient of the outer-product path is a sum in a fixed order: bitwise reproducible
        a, b = _model(pa, T, P), _model(pa, T, P)
        a.train_batch(users); b.train_batch(users)
        assert all(np.array_equal(_get(a)[k], _get(b)[k]) for k in NAMES)      # (lt too: its row gradients are sorted sums, no atomics)


def test_carnn_predict_scores_and_topk(pa):
    """seq_predict and compute_sub_all_scores literally as the reference's graphs define them (add-then-sum), with the
    last-POI interval matrix computed on the device == the reference's usrs_last_poi_to_all_intervals (oracle
    compute_distance); top-K ranks bit-exact on gap-checked rows."""
    T = toy_problem(540, n_user=21, n_item=150, n_dist=23, dim=32, len_max=9)
    rng = np.random.default_rng(9)
    coords = np.stack([40.0 + rng.random(150) * 0.05, -74.0 + rng.random(150) * 0.05], 1)
    P = _params(540, T)
    model = _model(pa, T, P, coords=coords)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(21, dtype=np.int32)
    hts = model.predict(ids)
    eh = O.carnn_predict(P, P["lt"], P["wd"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    assert_close(hts, eh, "hts")
    model.update_trained_users(hts)
    ul = O.compute_distance(T["train"][0], T["train"][1], [tuple(c) for c in coords], 200.0, 23)
    sub = np.array([3, 4, 5, 9, 20], np.int32)
    sc = model.compute_sub_all_scores(sub)
    es = O.carnn_score_all(np.asarray(hts, np.float64)[sub], P["lt"], P["M"], P["wd"], ul[sub])
    assert sc.shape == (5, 150)
    assert_close(sc, es, "scores")
    idx = model.compute_sub_topk(ids, 10).cpu().numpy()
    full = O.carnn_score_all(np.asarray(hts, np.float64), P["lt"], P["M"], P["wd"], ul)
    top = O.topk_desc(full, 11)
    tv = np.take_along_axis(full, top, axis=1)
    ok = (tv[:, :-1] - tv[:, 1:]).min(axis=1) > 1e-5 * np.abs(tv).max()
    assert ok.sum() >= 10
    assert np.array_equal(idx[ok], top[ok][:, :10])


def test_harness_runs_flag_3(pa):
    from poi_amd import harness
    from poi_amd.data import make_synthetic
    ds = make_synthetic(64, 300, 10, seed=31, local=0.8)
    p = harness.default_params(); p.update(latent_size=16, epochs=2, gru=3, batch_users=1, seed=3)
    model, best, hist = harness.train_valid_or_test(ds, p, log=lambda *a: None)
    assert model.__class__.__name__ == "OboCARNN" and len(hist) == 2
    assert all(np.isfinite(h["loss"]) and np.isfinite(h["l2"]) for h in hist) and 0.0 <= hist[-1]["auc"] <= 1.0
    p.update(batch_users=16)
    model, best, hist = harness.train_valid_or_test(ds, p, log=lambda *a: None)
    assert all(np.isfinite(h["loss"]) for h in hist)
