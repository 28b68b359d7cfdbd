"""-m gpu: the EXACT engine (poi_ctx_set_engine(ctx, 4), exact_engine.hip: float64 arithmetic end to end on float32 tables) against
the float64 oracle.  Where the float32 engines are held to BASELINE.json's 1e-5, this engine is held to what float32 STORAGE
allows: every tensor within 2e-7 of its max-norm after a step (half an ulp of the stored value + slack), every row's update
within 1e-6 of the update (+ the storage rounding), losses to 1e-6 - at every dim, for the reference step, the batch rule at
caps 1 / 4 / infinity, the mini-batch rule, the plain GRU and predict.  Full BASELINE shapes: tests/test_gpu_fullsize.py."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, batch_mean_update, gru_params, round_f32, spatial_params, toy_problem

pytestmark = pytest.mark.gpu

SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
GRU_NAMES = ("lt", "ui", "wh", "bi")
XTOL = 2e-7          # max-norm bar of a tensor stored in float32
XDELTA = 1e-6        # per-row bar of the update


@pytest.fixture()
def exact():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import poi_amd
    ctx = poi_amd._lib.context(0)
    ctx.set_engine("exact")
    yield poi_amd
    ctx.set_engine("auto"); ctx.set_batch_cap(1.0)


def _spatial_model(pa, T, P, **kw):
    return pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001],
                                   n_user=T["n_user"], n_item=T["n_item"], n_dists=[T["n_dist"], 0.2],
                                   n_in=T["dim"], n_hidden=T["dim"], init=P, **kw)


def _get(model, names):
    out = {}
    for k in names:
        v = getattr(model, k).get_value()
        out[k] = float(v) if k == "wd" else v
    return out


@pytest.mark.parametrize("pad_dim", [True, False])
@pytest.mark.parametrize("seed,dim,n_dist,n_item", [(0, 8, 11, 50), (1, 20, 37, 80), (2, 64, 200, 400), (3, 128, 200, 300), (5, 256, 300, 200)])
def test_exact_spatial_step_sequential(exact, seed, dim, n_dist, n_item, pad_dim):
    """model.train(uidx), one user after another (prog_bpr_gru_spatial.py:249-250), at the model's native dim and zero-padded."""
    T = toy_problem(seed, n_user=5, n_item=n_item, n_dist=n_dist, dim=dim, len_max=9 if dim > 32 else 10)
    P = spatial_params(seed, T)
    model = _spatial_model(exact, T, P, pad_dim=pad_dim)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    worst = 0.0
    for u in [3, 0, 4, 1, 0]:
        old = P
        P, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        los, sur, upq, ls = model.train(np.int32(u))
        assert_close([los, sur, upq], [out[0], out[1], out[2]], "losses", rtol=1e-6)
        assert_close(ls, out[3], "ls", rtol=XTOL)
        got = _get(model, SP_NAMES)
        worst = max(worst, assert_step_close(got, P, old, SP_NAMES, "after user %d" % u, rtol=XTOL, delta_rtol=XDELTA))
        P = round_f32({**P, **{k: got[k] for k in SP_NAMES}})
    print("exact engine, sequential: worst rel err %.2e" % worst)


@pytest.mark.parametrize("cap", [1.0, 4.0, 1e9])
def test_exact_batch_rule(exact, cap):
    """n_seq > 1: a row touched by k sequences moves by min(k, cap) / k x the sum of their reference updates."""
    T = toy_problem(11, n_user=9, n_item=40, n_dist=9, dim=16, len_max=8)
    P = spatial_params(11, T)
    model = _spatial_model(exact, T, P, pad_dim=False)
    model.ctx.set_batch_cap(cap)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    users = [5, 1, 2, 6, 0, 8, 7]
    news, touched, outs = [], [], []
    for u in users:
        Pn, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); outs.append(out)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), di=np.unique(DPm[u])))
    mean = batch_mean_update(P, news, touched, ("lt", "di"), ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"))
    # capped sum = base + min(k, cap) * (mean - base), k = touching sequences (dense: all of them)
    exp = {}
    n = len(users)
    for name in ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"):
        exp[name] = np.asarray(P[name], np.float64) + min(n, cap) * (np.asarray(mean[name], np.float64) - np.asarray(P[name], np.float64))
    exp["wd"] = float(exp["wd"])
    for name in ("lt", "di"):
        cnt = np.zeros(P[name].shape[0])
        for t in touched:
            cnt[t[name]] += 1
        exp[name] = P[name] + np.minimum(cnt, cap)[:, None] * (mean[name] - P[name])
    got_out = model.train_batch(np.array(users, np.int32))
    for k, out in enumerate(outs):
        assert_close(got_out[k][:3], out[:3], "losses[%d]" % k, rtol=1e-6)
    got = _get(model, SP_NAMES)
    assert_step_close(got, exp, P, SP_NAMES, "cap %g" % cap, rtol=XTOL, delta_rtol=XDELTA)


@pytest.mark.parametrize("seed,dim", [(0, 8), (1, 64), (2, 128)])
def test_exact_gru_step_sequential(exact, seed, dim):
    T = toy_problem(seed + 20, n_user=5, n_item=70, dim=dim)
    P = gru_params(seed, T)
    model = exact.models.OboGru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                n_item=T["n_item"], n_in=dim, n_hidden=dim, init=P)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    for u in [2, 0, 4, 2]:
        old = P
        P, loss = O.gru_step(P, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
        got_loss = model.train(np.int32(u))
        assert_close(got_loss, loss, "loss", rtol=1e-6)
        got = _get(model, GRU_NAMES)
        assert_step_close(got, P, old, GRU_NAMES, "after user %d" % u, rtol=XTOL, delta_rtol=XDELTA)
        P = round_f32({**P, **got})


def test_exact_minibatch_gru(exact):
    """The mini-batch rule (public/GRU.py:395-498, poi_ctx_set_batch_cap(0)) on the exact engine."""
    T = toy_problem(31, n_user=12, n_item=60, dim=32, len_max=9)
    P = gru_params(31, T)
    model = exact.models.Gru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                             n_in=32, n_hidden=32, init=P)
    Pm, Qm, Mm = np.asarray(T["train"][0]), np.asarray(T["train"][2]), np.asarray(T["train"][1])
    for ids in ([0, 3, 5, 7], [1, 2, 11], [4, 6, 8, 9, 10]):
        old = P
        P, loss = O.gru_minibatch_step(P, Pm[ids], Qm[ids], Mm[ids], 0.01, 0.001)
        got_loss = model.train(np.array(ids, np.int32))
        assert_close(got_loss, loss, "mini-batch loss", rtol=1e-6)
        got = _get(model, GRU_NAMES)
        assert_step_close(got, P, old, GRU_NAMES, "mini-batch %s" % ids, rtol=XTOL, delta_rtol=XDELTA)
        P = round_f32({**P, **got})


@pytest.mark.parametrize("dim,n_dist", [(32, 23), (128, 200), (256, 1520)])
def test_exact_predict(exact, dim, n_dist):
    T = toy_problem(40, n_user=37, n_item=333, n_dist=n_dist, dim=dim, len_max=12)
    P = spatial_params(40, T)
    model = _spatial_model(exact, T, P)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(3, 36, dtype=np.int32)
    hts, sts = model.predict(ids)
    eh, es = O.spatial_predict(P, P["lt"], P["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    assert_close(hts, eh, "hts", rtol=XTOL); assert_close(sts, es, "sts", rtol=XTOL)
