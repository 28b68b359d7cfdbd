"""-m gpu parity tests: HIP path (model classes -> ctypes C-ABI -> gfx950 kernels) vs the float64 oracle
on the same seeded inputs.  Tolerance for floating point: 1e-5 relative (max-norm per tensor), the
figure BASELINE.json's north_star states for f32 kernels against the f64 reference; integer / index
results (top-K ranks, AUC flags, distance bins) must be bit-exact."""
import os

import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import (RTOL, assert_close, assert_delta_close, assert_step_close, batch_mean_update, delta_excess, gru_params, rel_err, round_f32, spatial_params,
                            toy_problem)

pytestmark = pytest.mark.gpu

SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
GRU_NAMES = ("lt", "ui", "wh", "bi")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import poi_amd
    poi_amd._lib.load()
    return poi_amd


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


def test_selftest_primitives(pa):
    ctx = pa._lib.context(0)
    ctx.check(ctx.lib.poi_selftest(ctx.handle, None))


@pytest.mark.parametrize("engine", ["auto", "seq"])
@pytest.mark.parametrize("seed,dim,n_dist,n_item", [(0, 8, 11, 50), (1, 20, 37, 80), (4, 32, 200, 600), (2, 64, 200, 400), (3, 128, 200, 300)])
def test_spatial_step_parity_sequential(pa, seed, dim, n_dist, n_item, engine):
    """model.train(uidx) one user after another == the reference's hot loop #1
    (prog_bpr_gru_spatial.py:249-250): every step must match, and the state carried between steps
    (gradient tables re-zeroed, slabs consumed) must stay consistent."""
    T = toy_problem(seed, n_user=5, n_item=n_item, n_dist=n_dist, dim=dim, len_max=9 if dim > 32 else 10)
    P = spatial_params(seed, T)
    # auto: dims 8 / 20 / 32 are stored zero-padded to 64 and train on the tile engine (models.GruBasic pad_dim); seq: the
    # per-sequence engine at the model's NATIVE dim (pad_dim=False)
    model = _spatial_model(pa, T, P, pad_dim=(engine == "auto"))
    assert model.kdim == (dim if engine == "seq" or dim >= 64 else 64)
    model.ctx.set_engine(engine)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    worst = 0.0
    for u in [3, 0, 4, 1, 0]:
        old = P
        P, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        los, sur, upq, ls = model.train(np.int32(u))
        assert_close([los, sur, upq], [out[0], out[1], out[2]], "losses")
        assert_close(ls, out[3], "ls")
        got = _get(model, SP_NAMES)
        worst = max(worst, assert_step_close(got, P, old, SP_NAMES, "after user %d" % u))
        # continue BOTH sides from the device's float32 state so errors do not compound across steps
        P = round_f32({**P, **{k: got[k] for k in SP_NAMES}})
    model.ctx.set_engine("auto")
    print("spatial sequential worst rel err %.2e" % worst)


def test_spatial_step_touches_only_its_rows(pa):
    T = toy_problem(7, n_user=3, n_item=60, n_dist=11, dim=16)
    P = spatial_params(7, T)
    model = _spatial_model(pa, T, P)
    model.train(1)
    lt = model.lt.get_value()
    R = np.unique(np.concatenate((T["train"][0][1], T["train"][2][1])))
    untouched = np.setdiff1d(np.arange(T["n_item"] + 1), R)
    assert np.array_equal(lt[untouched], P["lt"][untouched].astype(np.float32))
    assert not np.array_equal(lt[R], P["lt"][R].astype(np.float32))


def test_spatial_batch_matches_mean_rule(pa):
    """n_seq > 1: rows move by the mean of the touching sequences' reference updates."""
    T = toy_problem(11, n_user=7, n_item=40, n_dist=9, dim=16, len_max=8)
    P = spatial_params(11, T)
    model = _spatial_model(pa, T, P, pad_dim=False)          # (the per-sequence engine at its native dim)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    users = [5, 1, 2, 6, 0]
    news, touched, outs = [], [], []
    for u in users:
        Pn, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); outs.append(out)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), di=np.unique(DPm[u])))
    exp = batch_mean_update(P, news, touched, ("lt", "di"), ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"))
    got_out = model.train_batch(np.array(users, np.int32))
    for k, out in enumerate(outs):
        assert_close(got_out[k][:3], out[:3], "losses[%d]" % k)
    got = _get(model, SP_NAMES)
    assert_step_close(got, exp, P, SP_NAMES)


def test_delta_tolerance_catches_a_padding_multiplicity_off_by_one(pa):
    """The 1e-5 max-norm bar on the weights cannot see an L2-decay multiplicity that is off by one on a padding
    row (alpha*lambda*|row| = 5e-6 absolute); the per-row delta bar must.  The oracle is run on the same sequence
    padded to len_max + 1 (one more padding id in p, q and dp: multiplicity +2 on lt[n_item], +1 on di[n_dist]):
    the device result has to FAIL the delta check against that perturbed expectation while it still passes the
    weight check - and pass both against the true one."""
    T = toy_problem(13, n_user=4, n_item=300, n_dist=200, dim=64, len_max=12)
    P = spatial_params(13, T)
    model = _spatial_model(pa, T, P)
    u = 2
    assert T["lens"][u] < T["len_max"]
    Pm, Qm, DPm, DQm, Mm = (np.asarray(T["train"][0]), np.asarray(T["train"][2]), np.asarray(T["dist"][0]), np.asarray(T["dist"][2]),
                            np.asarray(T["train"][1]))
    exp, _ = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
    pad = lambda row, v: np.append(row, v)
    bad, _ = O.spatial_step(P, pad(Pm[u], T["n_item"]), pad(Qm[u], T["n_item"]), pad(DPm[u], T["n_dist"]), pad(DQm[u], T["n_dist"]),
                            pad(Mm[u], 0), 0.01, 0.001)
    model.train(np.int32(u))
    got = _get(model, SP_NAMES)
    assert_step_close(got, exp, P, SP_NAMES)
    for k in ("lt", "di"):
        if k == "di":         # one extra padding bin: alpha*lambda*|row| <= 5e-6 absolute, invisible at 1e-5 of max|theta| ~ 0.5
            assert rel_err(got[k], bad[k]) <= 1.5 * RTOL, "the max-norm bar was expected to miss the off-by-one on " + k
        ex, row = delta_excess(got[k], bad[k], P[k])
        assert ex > 3.0 and row == (T["n_item"] if k == "lt" else T["n_dist"]), (k, ex, row)


@pytest.mark.parametrize("seed,dim", [(0, 8), (1, 64)])
def test_gru_step_parity_sequential(pa, seed, dim):
    T = toy_problem(seed + 20, n_user=5, n_item=70, dim=dim)
    P = gru_params(seed, T)
    model = pa.models.OboGru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                             n_item=T["n_item"], n_in=dim, n_hidden=dim, init=P)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    for u in [2, 0, 4, 2]:
        old = P
        P, loss = O.gru_step(P, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
        got_loss = model.train(np.int32(u))
        assert_close(got_loss, loss, "loss")
        got = _get(model, GRU_NAMES)
        assert_step_close(got, P, old, GRU_NAMES, "after user %d" % u)
        P = round_f32({**P, **got})


@pytest.mark.parametrize("mode", ["snapshot", "hogwild"])
@pytest.mark.parametrize("dim", [16, 64, 128, 256])
def test_bpr_step_parity(pa, mode, dim):
    T = toy_problem(30, n_user=9, n_item=40, dim=dim)
    rng = np.random.default_rng(5)
    P = round_f32(O.init_bpr_params(rng, T["n_user"], T["n_item"], dim))
    model = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                             n_item=T["n_item"], n_in=dim, n_hidden=dim, init=P)
    for (u, p, q) in [(0, 3, 17), (4, 3, 9), (0, 17, 2)]:
        old = P
        P, loss = O.bpr_step(P, u, p, q, 0.01, 0.001)
        got = float(model.train_batch([u], [p], [q], mode=mode)[0])
        assert_close(got, loss, "loss")
        assert_step_close({k: getattr(model, k).get_value() for k in ("ux", "lt")}, P, old, ("ux", "lt"))
        P = round_f32({k: getattr(model, k).get_value() for k in ("ux", "lt")})


def test_bpr_batch_collision_free_equals_sequential(pa):
    """A launch without shared rows must equal the reference's sequential loop in both modes."""
    dim = 32
    T = toy_problem(31, n_user=12, n_item=60, dim=dim)
    rng = np.random.default_rng(6)
    P0 = round_f32(O.init_bpr_params(rng, T["n_user"], T["n_item"], dim))
    us = np.arange(10); ps = np.arange(10) * 2; qs = np.arange(10) * 2 + 21
    P = P0
    losses = []
    for u, p, q in zip(us, ps, qs):
        P, l = O.bpr_step(P, u, p, q, 0.01, 0.001)
        losses.append(l)
    for mode in ("snapshot", "hogwild"):
        model = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                 n_item=T["n_item"], n_in=dim, n_hidden=dim, init=P0)
        got = model.train_batch(us, ps, qs, mode=mode)
        assert_close(got, losses, "losses " + mode)
        for k in ("ux", "lt"):
            assert_close(getattr(model, k).get_value(), P[k], k + " " + mode)


def test_predict_and_scores_parity(pa):
    T = toy_problem(40, n_user=37, n_item=333, n_dist=23, dim=32, len_max=12)
    P = spatial_params(40, T)
    model = _spatial_model(pa, T, P)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(3, 36, dtype=np.int32)
    hts, sts = model.predict(ids)
    eh, es = O.spatial_predict(P, P["lt"], P["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    assert_close(hts, eh, "hts"); assert_close(sts, es, "sts")
    # all users, then all-POI scores with a dense prob matrix (compat path, GRU_Spatial.py:117-125)
    allh, alls = model.predict(np.arange(T["n_user"], dtype=np.int32))
    model.update_trained_users(allh)
    rng = np.random.default_rng(3)
    prob = rng.random((T["n_user"], T["n_item"]))
    model.update_prob(prob)
    sc = model.compute_sub_all_scores(ids)
    users64 = allh.astype(np.float64)
    exp = O.score_all(users64[ids], P["lt"], P["wd"], prob.astype(np.float32).astype(np.float64)[ids])
    assert_close(sc, exp, "scores")
    # AUC preference flags: bit-exact on EVERY margin (the device sums the exact float64 products: misc.hip auc_kernel) - the small ones are
    # counted, not skipped
    flags = model.compute_sub_auc_preference(ids)
    eflags = O.auc_preference(users64[ids], P["lt"], T["test"][0][ids], T["test"][2][ids], T["test"][1][ids])
    assert np.array_equal(flags, eflags)


def test_auc_flags_bit_exact_on_tiny_margins(pa):
    """compute_sub_auc_preference (public/GRU.py:98-110) where a float32 margin has no reliable sign: every user's negative test POI is its
    positive one with ONE coordinate moved by one float32 ulp - margins of ~1e-8 x |u_j|, some exactly zero.  The device sums the exact
    float64 products (misc.hip auc_kernel): flags == the float64 oracle's on EVERY margin; the tiny ones are counted, not skipped."""
    T = toy_problem(77, n_user=400, n_item=900, n_dist=23, dim=64, len_max=8)
    P = spatial_params(77, T)
    rng = np.random.default_rng(5)
    T["test"][0][:, 0] = rng.permutation(900)[:400]                      # distinct positives, negatives = their perturbed copies in fresh rows
    T["test"][2][:, 0] = (T["test"][0][:, 0] + 1 + rng.integers(0, 3, 400)) % 900
    used = set(T["test"][0][:, 0].tolist())
    lt = np.asarray(P["lt"], np.float32)
    for u in range(400):
        p_, q_ = int(T["test"][0][u, 0]), int(T["test"][2][u, 0])
        if q_ in used:
            continue                                                        # (a row that is someone's positive keeps its values: ordinary margin)
        lt[q_] = lt[p_]
        j = int(rng.integers(0, 64))
        if u % 7:                                                           # every seventh pair stays identical: margin exactly 0 -> flag 0
            lt[q_, j] = np.nextafter(lt[q_, j], np.float32(np.inf if u % 2 else -np.inf))
    P["lt"] = lt.astype(np.float64)
    model = _spatial_model(pa, T, P)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(400, dtype=np.int32)
    allh, _ = model.predict(ids)
    model.update_trained_users(allh)
    flags = model.compute_sub_auc_preference(ids)
    users64 = allh.astype(np.float64)
    eflags = O.auc_preference(users64, P["lt"], T["test"][0], T["test"][2], T["test"][1])
    margins = np.einsum("nd,nld->nl", users64, P["lt"][T["test"][0]] - P["lt"][T["test"][2]])
    tiny = np.abs(margins) <= 1e-4
    assert tiny.sum() > 150 and (margins == 0).sum() > 10, (int(tiny.sum()), int((margins == 0).sum()))
    assert np.array_equal(flags, eflags), "%d of %d flags differ (%d tiny margins)" % ((flags != eflags).sum(), flags.size, tiny.sum())
    assert 0 < flags[tiny].sum() < tiny.sum()                              # both signs occur among the tiny margins


def test_gru_predict_parity(pa):
    T = toy_problem(41, n_user=9, n_item=50, dim=16)
    P = gru_params(41, T)
    model = pa.models.OboGru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                             n_item=T["n_item"], n_in=16, n_hidden=16, init=P)
    model.update_trained_items()
    ids = np.array([7, 2, 5], np.int32)          # non-contiguous ids
    hts = model.predict(ids)
    assert_close(hts, O.gru_predict(P, P["lt"], T["train"][0][ids], T["train"][1][ids]), "hts")


def test_topk_golden_bit_exact(pa, golden_dir):
    """Ranks from the reference's own helpers (tests/golden/topk.npz) reproduced bit-exactly by the
    HIP top-K on the same (float32-representable) scores."""
    import torch
    g = np.load(os.path.join(golden_dir, "topk.npz"))
    scores = g["scores"].astype(np.float32)
    # the golden scores are tie-free in float64; require the same after float32 rounding
    srt = np.sort(scores, axis=1)
    assert np.min(np.diff(srt, axis=1)) > 0
    ctx = pa._lib.context(0)
    dev = torch.as_tensor(scores).cuda()
    for k, key in ((20, "ranks"), (5, "ranks5")):
        exp = O.topk_desc(scores, k)
        assert np.array_equal(exp, g[key]), "float32 rounding changed the float64 order of the golden scores"
        idx = torch.empty((scores.shape[0], k), dtype=torch.int32, device="cuda")
        ctx.check(ctx.lib.poi_topk(ctx.handle, dev.data_ptr(), scores.shape[0], scores.shape[1], k, idx.data_ptr(), None, None))
        assert np.array_equal(idx.cpu().numpy(), g[key]), "HIP top-K differs from the ranks of the reference's own helpers"
        assert np.array_equal(idx.cpu().numpy(), exp)
    assert np.array_equal(O.topk_desc(g["scores"], 20), g["ranks"])


@pytest.mark.parametrize("n,n_item,dim,k", [(5, 100, 16, 5), (70, 1777, 64, 20), (33, 5000, 128, 20), (32, 640, 256, 10), (3, 50, 20, 20),
                                            (128, 3001, 128, 20), (300, 2500, 64, 20), (129, 700, 256, 7), (200, 999, 20, 20)])
def test_score_topk_fused_bit_exact_ranks(pa, n, n_item, dim, k):
    """Fused MFMA scoring + top-K: ranks bit-exact vs the oracle's ordering of the float64 scores on
    fixtures with a checked minimum score gap ("bit-exact ranks vs precision")."""
    import torch
    rng = np.random.default_rng(n * 7 + dim)
    users = rng.uniform(-0.5, 0.5, (n, dim)).astype(np.float32)
    items = rng.uniform(-0.5, 0.5, (n_item + 1, dim)).astype(np.float32)
    exp_sc = O.score_all(users.astype(np.float64), items.astype(np.float64))
    exp = O.topk_desc(exp_sc, k)
    ctx = pa._lib.context(0)
    du, di = torch.as_tensor(users).cuda(), torch.as_tensor(items).cuda()
    idx = torch.empty((n, k), dtype=torch.int32, device="cuda")
    sc = torch.empty((n, k), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.poi_score_topk(ctx.handle, du.data_ptr(), di.data_ptr(), n, n_item, dim, None, None, k,
                                     idx.data_ptr(), sc.data_ptr(), None))
    got, got_sc = idx.cpu().numpy(), sc.cpu().numpy()
    top_sc = np.take_along_axis(exp_sc, exp, axis=1)
    assert_close(got_sc, top_sc, "top-K scores")
    # rows whose top-(K+1) float64 scores are separated by more than the f32 error must match exactly
    kk = min(k + 1, n_item)
    top = -np.sort(-exp_sc, axis=1)[:, :kk]
    gap_ok = np.min(-np.diff(top, axis=1), axis=1) > 1e-5
    assert gap_ok.mean() > 0.5, "fixture too tie-prone"
    assert np.array_equal(got[gap_ok], exp[gap_ok])
    # every row: same set up to swaps of near-tied neighbours
    full = torch.empty((n, n_item), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.poi_score_all(ctx.handle, du.data_ptr(), di.data_ptr(), n, n_item, dim, None, None, full.data_ptr(), None))
    fsc = full.cpu().numpy()
    assert_close(fsc, exp_sc, "all scores")
    # the fused top-K must be EXACTLY the top-K of the scores the same kernel produced (integer work)
    assert np.array_equal(got, O.topk_desc(fsc, k))


def test_topk_cutoffs_beyond_32(pa):
    """at_nums such as [5, 10, 15, 20, 30, 50] (public/Valuate.py:126): k > 32 goes through explicit score rows + poi_topk
    (k <= 64); ranks bit-exact against the float64 oracle on gap-checked rows; metrics accept the cut-offs; k > 64 is loud."""
    from poi_amd.evaluate import device_rank_metrics
    T = toy_problem(77, n_user=40, n_item=700, n_dist=11, dim=32)
    P = spatial_params(77, T)
    model = _spatial_model(pa, T, P)
    model.update_trained_items()
    rng = np.random.default_rng(3)
    users = rng.uniform(-0.5, 0.5, (40, 32)).astype(np.float32)
    model.update_trained_users(users)
    ids = np.arange(40, dtype=np.int32)
    for k in (33, 50, 64):
        idx = model.compute_sub_topk(ids, k).cpu().numpy()
        full = users.astype(np.float64) @ np.asarray(P["lt"][:-1], np.float64).T
        top = O.topk_desc(full, k + 1)
        tv = np.take_along_axis(full, top, axis=1)
        ok = (tv[:, :-1] - tv[:, 1:]).min(axis=1) > 2e-6 * np.abs(tv).max()
        assert ok.sum() >= 10
        assert np.array_equal(idx[ok], top[ok][:, :k])
    m = device_rank_metrics(model, [ids], [5, 10, 15, 20, 30, 50])
    assert set(m) == {5, 10, 15, 20, 30, 50} and all(0.0 <= m[k]["recall"] <= 1.0 for k in m)
    with pytest.raises(pa._lib.PoiError):
        model.compute_sub_topk(ids, 65)
    with pytest.raises(ValueError):
        device_rank_metrics(model, [ids], [20, 10])


def test_topk_with_ties_uses_index_order(pa):
    import torch
    sc = np.zeros((4, 300), np.float32)
    sc[:, [3, 7, 9]] = 1.0
    sc[1, 250] = 1.0
    ctx = pa._lib.context(0)
    dev = torch.as_tensor(sc).cuda()
    idx = torch.empty((4, 5), dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.poi_topk(ctx.handle, dev.data_ptr(), 4, 300, 5, idx.data_ptr(), None, None))
    assert np.array_equal(idx.cpu().numpy(), O.topk_desc(sc, 5))
    assert list(idx.cpu().numpy()[0]) == [3, 7, 9, 0, 1]


def test_dist_prob_matches_reference_bins(pa, golden_dir):
    """Device Haversine bins == the reference's fun_compute_distance output (golden, bit-exact), and
    prob rows == fun_acquire_prob restated."""
    import torch
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    N, B, dd = int(g["n_item"]), int(g["n_dist"]), float(g["dd"])
    ul = g["ulptai"]
    U = ul.shape[0]
    last = g["pois_m"][np.arange(U), g["msks"].sum(1) - 1].astype(np.int32)
    rng = np.random.default_rng(0)
    sts = rng.random((U, B + 1)).astype(np.float32)
    ctx = pa._lib.context(0)
    dc = torch.as_tensor(g["coords"].astype(np.float64)).cuda()
    dl, ds = torch.as_tensor(last).cuda(), torch.as_tensor(sts).cuda()
    exp = O.acquire_prob(sts.astype(np.float64), ul, B)
    out = torch.empty((U, N), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.poi_dist_prob(ctx.handle, dc.data_ptr(), None, None, dl.data_ptr(), ds.data_ptr(), U, N, B, dd, out.data_ptr(), None))
    assert np.array_equal(out.cpu().numpy(), exp.astype(np.float32))          # literal cal_dis path
    from poi_amd.data import bin_thresholds, cos_lat
    cphi = torch.as_tensor(cos_lat(g["coords"])).cuda()
    thr = torch.as_tensor(bin_thresholds(dd, B)).cuda()
    out2 = torch.empty((U, N), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.poi_dist_prob(ctx.handle, dc.data_ptr(), cphi.data_ptr(), thr.data_ptr(), dl.data_ptr(), ds.data_ptr(), U, N, B, dd,
                                    out2.data_ptr(), None))
    assert np.array_equal(out2.cpu().numpy(), exp.astype(np.float32))         # threshold path


def test_dist_prob_threshold_path_matches_oracle_at_scale(pa):
    """3000 POIs x 40 users (120k pairs) of synthetic Gowalla-like coordinates: bins from the
    threshold path == oracle cal_dis (python libm) for every pair."""
    import torch
    from poi_amd.data import bin_thresholds, cos_lat, make_synthetic
    ds = make_synthetic(40, 3000, 12, seed=5)
    B, dd = ds.dist_num, ds.dd
    last = ds.last_pois().astype(np.int32)
    # probabilities = bin index itself, so that the output reveals the bin
    sts = np.tile(np.arange(B + 1, dtype=np.float32), (40, 1))
    ctx = pa._lib.context(0)
    dc = torch.as_tensor(ds.coords).cuda(); dl = torch.as_tensor(last).cuda(); dst = torch.as_tensor(sts).cuda()
    cphi = torch.as_tensor(cos_lat(ds.coords)).cuda(); thr = torch.as_tensor(bin_thresholds(dd, B)).cuda()
    out = torch.empty((40, 3000), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.poi_dist_prob(ctx.handle, dc.data_ptr(), cphi.data_ptr(), thr.data_ptr(), dl.data_ptr(), dst.data_ptr(), 40, 3000, B, dd,
                                    out.data_ptr(), None))
    got = out.cpu().numpy()
    exp = np.empty((40, 3000), np.float32)
    for u in range(40):
        lc = ds.coords[last[u]]
        for j in range(3000):
            b = O.cal_dis(lc[0], lc[1], ds.coords[j][0], ds.coords[j][1], dd, B)
            exp[u, j] = b if b < B else 0
    assert np.array_equal(got, exp)


def test_ulptai_matrix_and_fused_topk(pa, golden_dir):
    """poi_ulptai_build == the reference's fun_compute_distance output (golden ulptai, decoded from the
    tile layout, bit-exact) and == oracle cal_dis on synthetic coordinates; poi_score_topk_ulptai ranks ==
    oracle top-K of  users.items + wd * fun_acquire_prob(sus, ulptai)  (bit-exact, checked gaps)."""
    import torch
    from poi_amd.data import bin_thresholds, cos_lat, make_synthetic
    ctx = pa._lib.context(0)

    def build(coords, last, n_user, n_item, B, dd):
        bb = 1 if B <= 255 else 2
        nut, nt = (n_user + 31) // 32, (n_item + 31) // 32
        buf = torch.empty(nut * nt * 1024 * bb, dtype=torch.uint8, device="cuda")
        dc = torch.as_tensor(np.asarray(coords, np.float64)).cuda()
        cphi = torch.as_tensor(cos_lat(coords)).cuda(); thr = torch.as_tensor(bin_thresholds(dd, B)).cuda()
        dl = torch.as_tensor(np.asarray(last, np.int32)).cuda()
        ctx.check(ctx.lib.poi_ulptai_build(ctx.handle, dc.data_ptr(), cphi.data_ptr(), thr.data_ptr(), dl.data_ptr(), n_user, n_item, B, dd,
                                           buf.data_ptr(), bb, None))
        a = buf.cpu().numpy().view(np.uint8 if bb == 1 else np.uint16).reshape(nut, nt, 64, 16).astype(np.int32)
        lane, r = np.arange(64)[:, None], np.arange(16)[None, :]
        row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); col = np.broadcast_to(lane & 31, (64, 16))
        out = np.empty((nut * 32, nt * 32), np.int32)
        for ut in range(nut):
            blk = np.empty((32, nt, 32), np.int32)
            blk[row, :, col] = a[ut].transpose(1, 2, 0)
            out[ut * 32:(ut + 1) * 32] = blk.reshape(32, nt * 32)
        assert np.all(out[n_user:] == B) and np.all(out[:, n_item:] == B)      # padding = "too far"
        return buf, bb, out[:n_user, :n_item]

    g = np.load(os.path.join(golden_dir, "masks.npz"))
    N, B, dd = int(g["n_item"]), int(g["n_dist"]), float(g["dd"])
    ul = g["ulptai"]
    last = g["pois_m"][np.arange(ul.shape[0]), g["msks"].sum(1) - 1]
    _, _, got = build(g["coords"], last, ul.shape[0], N, B, dd)
    assert np.array_equal(got, ul)
    # uint16 bins (the reference's dd = 25 m / 1520 bins configuration)
    B2 = 1520
    _, bb2, got2 = build(g["coords"], last, ul.shape[0], N, B2, 25.0)
    assert bb2 == 2
    exp2 = np.array([[O.cal_dis(g["coords"][l][0], g["coords"][l][1], c[0], c[1], 25.0, B2) for c in g["coords"]] for l in last])
    assert np.array_equal(got2, exp2)

    ds = make_synthetic(150, 2100, 12, seed=11)
    B, dd = ds.dist_num, ds.dd
    last = ds.last_pois()
    buf, bb, bins = build(ds.coords, last, 150, 2100, B, dd)
    rng = np.random.default_rng(4)
    D, K = 64, 20
    users = rng.standard_normal((150, D)).astype(np.float32)
    items = rng.standard_normal((2100, D)).astype(np.float32)
    sus = rng.random((150, B + 1)).astype(np.float32)
    wd = np.float32(0.7)
    prob = O.acquire_prob(sus.astype(np.float64), bins, B)
    full = users.astype(np.float64) @ items.astype(np.float64).T + float(wd) * prob
    exp = O.topk_desc(full, K)
    srt = np.sort(full, axis=1)[:, ::-1][:, :K + 1]
    ok = np.min(srt[:, :-1] - srt[:, 1:], axis=1) > 1e-4                        # rows whose top-K order is unambiguous in f32
    assert ok.sum() > 100
    sus_m = np.zeros((160, B + 1), np.float32)             # whole 32-user tiles, "too far" column zeroed (ABI contract)
    sus_m[:150, :B] = sus[:, :B]
    du, di, dsus = torch.as_tensor(users).cuda(), torch.as_tensor(items).cuda(), torch.as_tensor(sus_m).cuda()
    dwd = torch.as_tensor(np.array([wd])).cuda()
    for lo, n in ((0, 150), (64, 86), (32, 64)):                               # whole matrix, ragged tail, interior batch
        idx = torch.empty((n, K), dtype=torch.int32, device="cuda")
        row_bytes = ((2100 + 31) // 32) * 1024 * bb
        ctx.check(ctx.lib.poi_score_topk_ulptai(ctx.handle, du[lo:lo + n].data_ptr(), di.data_ptr(), n, 2100, D, dwd.data_ptr(),
                                                dsus[lo:lo + n].data_ptr(), buf.data_ptr() + (lo // 32) * row_bytes, bb, B, K,
                                                idx.data_ptr(), None, None))
        got = idx.cpu().numpy()
        sel = ok[lo:lo + n]
        assert np.array_equal(got[sel], exp[lo:lo + n][sel])


@pytest.mark.parametrize("dim", [64, 256])
def test_geo_scoring_bins_on_the_fly_equal_the_bin_matrix_and_the_oracle(pa, dim):
    """poi_score_topk_geo: the distance bins of (last train POI, POI) are computed inside the scoring kernel - ranks must equal
    (a) the float64 oracle with the reference's usrs_last_poi_to_all_intervals (compute_distance) + fun_acquire_prob and
    (b) for dim <= 128 the resident-bin-matrix path, for unaligned and arbitrary id lists and at dim 256."""
    T = toy_problem(88, n_user=70, n_item=1500, n_dist=200, dim=dim, len_max=9)
    rng = np.random.default_rng(8)
    coords = np.stack([40.0 + rng.random(1500) * 0.3, -74.0 + rng.random(1500) * 0.3], 1)
    P = spatial_params(88, T)
    model = _spatial_model(pa, T, P, coords=coords)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(70, dtype=np.int32)
    hts, sts = model.predict(ids)
    model.update_trained_users(hts); model.update_trained_sus(sts)
    ul = O.compute_distance(T["train"][0], T["train"][1], [tuple(c) for c in coords], 200.0, 200)
    prob = O.acquire_prob(np.asarray(sts, np.float64), ul, 200)
    full = O.score_all(np.asarray(hts, np.float64), np.asarray(P["lt"], np.float64), float(P["wd"]), prob)
    top = O.topk_desc(full, 21)
    tv = np.take_along_axis(full, top, axis=1)
    ok = (tv[:, :-1] - tv[:, 1:]).min(axis=1) > (1e-5 if dim <= 128 else 6e-5) * np.abs(tv).max()
    assert ok.sum() >= 30
    model.use_bin_matrix = False
    for sel in (ids, np.arange(5, 47, dtype=np.int32), np.array([3, 60, 17, 18, 44], np.int32)):
        idx = model.compute_sub_topk(sel, 20).cpu().numpy()
        assert np.array_equal(idx[ok[sel]], top[sel][ok[sel]][:, :20]), "geo ranks differ from the oracle's"
    if dim <= 128:
        model.use_bin_matrix = True
        a = model.compute_sub_topk(ids, 20).cpu().numpy()
        model.use_bin_matrix = False
        b = model.compute_sub_topk(ids, 20).cpu().numpy()
        assert np.array_equal(a, b), "bin-matrix path and on-the-fly path disagree"


def test_l2_eval(pa):
    T = toy_problem(50, n_user=4, n_item=30, n_dist=7, dim=8)
    P = spatial_params(50, T)
    model = _spatial_model(pa, T, P)
    exp = O.l2_value(P, 0.001, SP_NAMES)
    assert abs(model.l2.eval() - exp) <= 1e-6 * exp


def test_out_of_range_ids_raise_index_error(pa):
    """Theano's advanced indexing raises IndexError on an id outside the table; so do the constructors and
    update_neg_masks (the kernels themselves never see an unchecked id)."""
    T = toy_problem(52, n_user=4, n_item=30, n_dist=7, dim=8)
    bad = [np.array(T["train"][0]), T["train"][1], T["train"][2]]
    bad[0] = bad[0].copy(); bad[0][1, 0] = 31
    kw = dict(alpha_lambda=[0.01, 0.001], n_user=4, n_item=30, n_in=8, n_hidden=8)
    with pytest.raises(IndexError):
        pa.models.OboGru(train=bad, test=T["test"], **kw)
    dist = [np.array(T["dist"][0]).copy(), T["dist"][1], T["dist"][2]]
    dist[0][2, 1] = 9
    with pytest.raises(IndexError):
        pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=dist, n_dists=[7, 0.2], **kw)
    m = pa.models.OboGru(train=T["train"], test=T["test"], **kw)
    neg = np.array(T["train"][2]).copy(); neg[0, 0] = -1
    with pytest.raises(IndexError):
        m.update_neg_masks(neg, T["test"][2])
    # user ids: a list with an id >= n_user, a negative id, and a contiguous range running past the table
    for ids in ([0, 4], [-1], np.arange(2, 6)):
        with pytest.raises(IndexError):
            m.train_batch(ids)
        with pytest.raises(IndexError):
            m.predict(ids)


def test_errors_are_loud(pa):
    ctx = pa._lib.context(0)
    rc = ctx.lib.poi_topk(ctx.handle, None, 1, 10, 5, None, None, None)
    assert rc < 0 and b"NULL" in ctx.lib.poi_last_error(ctx.handle)
    with pytest.raises(pa.PoiError):
        ctx.check(ctx.lib.poi_score_topk(ctx.handle, 1, 1, 1, 10, 6, None, None, 5, 1, None, None))   # dim % 4 != 0


def test_device_negative_sampling_contract_and_bins(pa):
    """On-device per-epoch negative refresh: the sampler's contract (Load_Data_by_length.py:127-162:
    support [0, n_item), never one of the user's train items, test negative also avoids the test item)
    and the negative distance bins bit-exact vs the oracle's fun_compute_dist_neg restatement on the
    sampled negatives; reproducible per seed, different across seeds, roughly uniform."""
    from poi_amd import harness
    from poi_amd.data import make_synthetic
    ds = make_synthetic(300, 400, 14, seed=21)
    p = harness.default_params(); p.update(latent_size=16, gru=2)
    model = harness.build_model(ds, p, seed=1)
    model.resample_negatives_device(1234)
    q1, dq1, tq1 = model.q.cpu().numpy(), model.dq.cpu().numpy(), model.tes_buys_neg_masks.cpu().numpy()
    model.resample_negatives_device(1234)
    assert np.array_equal(q1, model.q.cpu().numpy()) and np.array_equal(dq1, model.dq.cpu().numpy())
    model.resample_negatives_device(99)
    assert not np.array_equal(q1, model.q.cpu().numpy())
    off = ds.off.astype(np.int64)
    assert q1.min() >= 0 and q1.max() < ds.n_item
    for u in range(ds.n_user):
        own = set(ds.tra_p[off[u]:off[u + 1]])
        assert not own & set(q1[off[u]:off[u + 1]])
        assert tq1[u, 0] not in own and tq1[u, 0] != ds.tes_p[u] and 0 <= tq1[u, 0] < ds.n_item
    pad = ds.to_padded()
    from poi_amd.data import csr_to_padded
    exp = O.compute_dist_neg(pad["train"][0], pad["train"][1], csr_to_padded(ds.off, q1, ds.n_item, ds.len_max),
                             [list(c) for c in ds.coords], ds.dd, ds.dist_num)
    assert np.array_equal(csr_to_padded(ds.off, dq1, ds.dist_num, ds.len_max), np.array(exp))
    # uniformity: 400 items, ~2700 draws -> no item should be wildly over-represented
    cnt = np.bincount(q1, minlength=ds.n_item)
    assert cnt.max() < 8 * cnt.mean() + 10
    # the refreshed tables feed training
    out = model.train_batch(np.arange(64, dtype=np.int32))
    assert np.all(np.isfinite(out))


def test_device_rank_metrics_match_reference_golden(pa, golden_dir):
    """poi_rank_metrics vs the reference's own fun_hit_zero_one / fun_evaluate_map / fun_evaluate_ndcg
    outputs (tests/golden/metrics.npz) at several cut-offs, ragged test masks included."""
    import torch
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    ranks, tes, msk = g["recom"].astype(np.int32), g["test_lst"].astype(np.int32), g["test_mask"].astype(np.int32)
    at = [5, 10, 15, 20]
    ctx = pa._lib.context(0)
    dr, dt, dm = (torch.as_tensor(v).cuda() for v in (ranks, tes, msk))
    da = torch.as_tensor(np.array(at, np.int32)).cuda()
    acc = torch.zeros((4, 3), dtype=torch.float64, device="cuda")
    ctx.check(ctx.lib.poi_rank_metrics(ctx.handle, dr.data_ptr(), ranks.shape[0], 20, dt.data_ptr(), dm.data_ptr(), tes.shape[1],
                                       da.data_ptr(), 4, acc.data_ptr(), None))
    got = acc.cpu().numpy()
    exp = O.evaluate_ranks(ranks, tes, msk, at)
    for i, k in enumerate(at):
        assert got[i, 0] == exp[k]["hits"]
        assert np.isclose(got[i, 1] / len(ranks), exp[k]["map"], rtol=1e-12) and np.isclose(got[i, 2] / len(ranks), exp[k]["ndcg"], rtol=1e-12)
    # k = 20 equals the golden vectors produced by the reference helpers themselves
    assert got[3, 0] == g["zero_one"].sum()
    assert np.isclose(got[3, 1], g["map"].sum(), rtol=1e-12) and np.isclose(got[3, 2], g["ndcg"].sum(), rtol=1e-12)


@pytest.mark.parametrize("dim,f16", [(128, False), (256, False), (256, True)])
def test_geo_stream_kernel_ranks_match_the_oracle(pa, dim, f16):
    """poi_score_topk_geo with >= 1024 users and dim >= 128 runs the packed-stream GEO kernel (eight-wave workgroups, ring-buffered
    item stream, distance term only where dot + max_b wd * sts[b] can beat the user's K-th best): ranks must equal the float64
    oracle (cal_dis bins of (last POI, POI) -> fun_acquire_prob -> scores -> top-K), with a distance term large enough to decide
    ranks, for a ragged user count and a half item table."""
    import torch
    from poi_amd.data import bin_thresholds, cal_dis_vec, cos_lat
    ctx = pa._lib.context(0)
    rng = np.random.default_rng(40 + dim)
    n, N, B, dd, K = 1100, 4011, 200, 200.0, 20      # (ragged user and item counts)
    coords = np.stack([40.0 + rng.random(N) * 0.25, -74.0 + rng.random(N) * 0.25], 1)
    last = rng.integers(0, N, n).astype(np.int32)
    users = (rng.standard_normal((n, dim)) * (0.06 if dim == 128 else 0.045)).astype(np.float32)
    items = rng.standard_normal((N, dim)).astype(np.float16 if f16 else np.float32)
    sus = rng.random((n, B + 1)).astype(np.float32); sus[:, B] = 0.0
    wd = np.float32(0.9)
    bins = np.stack([cal_dis_vec(coords[l, 0], coords[l, 1], coords[:, 0], coords[:, 1], dd, B) for l in last])
    prob = np.take_along_axis(sus.astype(np.float64), bins, axis=1)
    dot = users.astype(np.float64) @ items.astype(np.float64).T
    full = dot + float(wd) * prob
    exp = O.topk_desc(full, K)
    assert (exp != O.topk_desc(dot, K)).mean() > 0.3, "the distance term must matter in this test"
    srt = np.sort(full, axis=1)[:, ::-1][:, :K + 1]
    ok = np.min(srt[:, :-1] - srt[:, 1:], axis=1) > 2e-5 * np.abs(srt).max()
    assert ok.sum() > 600
    pad = ((n + 31) // 32) * 32
    sus_m = np.zeros((pad, B + 1), np.float32); sus_m[:n] = sus
    du, dsus = torch.as_tensor(users).cuda(), torch.as_tensor(sus_m).cuda()
    di = torch.as_tensor(items).cuda()
    if f16:
        ctx.register_f16(di)
    try:
        dco, dcp = torch.as_tensor(coords).cuda(), torch.as_tensor(cos_lat(coords)).cuda()
        dth = torch.as_tensor(bin_thresholds(dd, B)).cuda()
        dlast, dwd = torch.as_tensor(last).cuda(), torch.as_tensor(np.array([wd])).cuda()
        idx = torch.empty((n, K), dtype=torch.int32, device="cuda")
        sc = torch.empty((n, K), dtype=torch.float32, device="cuda")
        ctx.check(ctx.lib.poi_score_topk_geo(ctx.handle, du.data_ptr(), di.data_ptr(), n, N, dim, dwd.data_ptr(), dsus.data_ptr(), dco.data_ptr(),
                                             dcp.data_ptr(), dth.data_ptr(), dlast.data_ptr(), B, dd, K, idx.data_ptr(), sc.data_ptr(), None))
        got, gsc = idx.cpu().numpy(), sc.cpu().numpy()
    finally:
        if f16:
            ctx.unregister_f16(di)
    assert np.array_equal(got[ok], exp[ok])
    assert_close(gsc[ok], np.take_along_axis(full, exp, axis=1)[ok], "top-K scores", rtol=2e-5)


@pytest.mark.parametrize("dim,with_prob", [(64, False), (128, True)])
def test_seeded_topk_is_exact_whatever_the_seed_holds(pa, dim, with_prob):
    """poi_ctx_set_topk_seed: the seed items' scores only RAISE the starting thresholds to a proven lower bound of the K-th best score,
    so the fused top-K must return exactly the unseeded result for good seeds (the true top-K), useless seeds (random items), and
    malformed rows (repeated / out-of-range ids are ignored) - and be consumed by one call."""
    import torch
    ctx = pa._lib.context(0)
    rng = np.random.default_rng(77 + dim)
    n, N, K = 300, 5000, 20
    users = (rng.standard_normal((n, dim)) * 0.3).astype(np.float32)
    items = rng.standard_normal((N, dim)).astype(np.float32)
    prob = rng.random((n, N)).astype(np.float32) if with_prob else None
    wd = np.array([0.8], np.float32)
    du, di = torch.as_tensor(users).cuda(), torch.as_tensor(items).cuda()
    dp = torch.as_tensor(prob).cuda() if with_prob else None
    dwd = torch.as_tensor(wd).cuda()

    def run(seed=None, k_seed=K):
        idx = torch.empty((n, K), dtype=torch.int32, device="cuda")
        sc = torch.empty((n, K), dtype=torch.float32, device="cuda")
        if seed is not None:
            ctx.set_topk_seed(seed, k_seed)
        ctx.check(ctx.lib.poi_score_topk(ctx.handle, du.data_ptr(), di.data_ptr(), n, N, dim, dwd.data_ptr() if with_prob else None,
                                         dp.data_ptr() if with_prob else None, K, idx.data_ptr(), sc.data_ptr(), None))
        return idx.cpu().numpy(), sc.cpu().numpy()

    base_idx, base_sc = run()
    full = users.astype(np.float64) @ items.astype(np.float64).T + (float(wd[0]) * prob if with_prob else 0.0)
    exp = O.topk_desc(full, K)
    srt = np.sort(full, axis=1)[:, ::-1][:, :K + 1]
    ok = np.min(srt[:, :-1] - srt[:, 1:], axis=1) > 1e-5 * np.abs(srt).max()
    assert np.array_equal(base_idx[ok], exp[ok])
    good = torch.as_tensor(base_idx).cuda()
    rnd = torch.as_tensor(np.stack([rng.choice(N, K, replace=False) for _ in range(n)]).astype(np.int32)).cuda()
    bad = good.clone(); bad[::3, 5] = bad[::3, 4]; bad[1::3, 0] = N + 7; bad[2::3, 2] = -1        # repeated / out of range / negative
    wide = torch.as_tensor(np.concatenate([base_idx[:, ::-1], rnd.cpu().numpy()], axis=1).copy()).cuda()      # k_seed = 40 > K (may repeat: ignored rows)
    for name, seed, ks in (("true top-K", good, K), ("random items", rnd, K), ("malformed rows", bad, K), ("k_seed > K", wide, 2 * K)):
        idx, sc = run(seed, ks)
        assert np.array_equal(idx, base_idx), name
        assert np.array_equal(sc, base_sc), name
    idx, _ = run()                       # the seed was consumed: this call is unseeded again
    assert np.array_equal(idx, base_idx)


@pytest.mark.parametrize("n_dist", [200, 300])          # uint8 / uint16 bin matrix
def test_model_level_seeding_across_evaluations(pa, n_dist):
    """models.compute_sub_topk seeds each evaluation with the previous one's lists (contiguous user ranges): identical ranks with the
    seeding on and off, for the bin-matrix path, the on-the-fly path and after the model has moved."""
    T = toy_problem(91, n_user=96, n_item=900, n_dist=n_dist, dim=64, len_max=9)
    rng = np.random.default_rng(9)
    coords = np.stack([40.0 + rng.random(900) * 0.3, -74.0 + rng.random(900) * 0.3], 1)
    P = spatial_params(91, T)
    model = _spatial_model(pa, T, P, coords=coords)
    ids = np.arange(96, dtype=np.int32)

    def evaluate():
        model.update_trained_items(); model.update_trained_dists()
        hts, sts = model.predict(ids)
        model.update_trained_users(hts); model.update_trained_sus(sts)
        out = {}
        for ubm in (True, False):
            model.use_bin_matrix = ubm
            model.topk_seeding = False
            ref = model.compute_sub_topk(ids, 20).cpu().numpy()
            model.topk_seeding = True
            a = model.compute_sub_topk(ids, 20).cpu().numpy()          # (first time: fills the seeds)
            b = model.compute_sub_topk(ids, 20).cpu().numpy()          # seeded with its own result
            assert np.array_equal(a, ref) and np.array_equal(b, ref)
            out[ubm] = ref
        return out

    first = evaluate()
    for u in range(0, 96, 7):
        model.train(np.int32(u))
    second = evaluate()                  # seeded with the lists of the first evaluation, under the moved model
    assert any(not np.array_equal(first[k], second[k]) for k in first), "training did not move any list: the test is vacuous"


@pytest.mark.parametrize("dim", [20, 100])
def test_padded_dim_is_exact_and_invisible(pa, dim):
    """A model whose dim is not 64 / 128 / 256 is stored zero-padded to the next of those (tile engine instead of the per-sequence
    engine).  The padding must stay EXACTLY zero through training, every logical view (get_value, predict, checkpoint values) must
    have the reference's shapes, and the padded and the native model must agree with the oracle and with each other."""
    import torch
    T = toy_problem(900 + dim, n_user=24, n_item=70, n_dist=37, dim=dim, len_max=9)
    P = spatial_params(900 + dim, T)
    a, b = _spatial_model(pa, T, P), _spatial_model(pa, T, P, pad_dim=False)
    K = 64 if dim < 64 else 128
    assert (a.kdim, b.kdim) == (K, dim) and a.dim == dim
    users = np.arange(24, dtype=np.int32)
    for m in (a, b):
        m.train_batch(users[:17]); m.train_batch(users[5:])
    ga, gb = _get(a, SP_NAMES), _get(b, SP_NAMES)
    for k in SP_NAMES:
        assert np.asarray(ga[k]).shape == np.asarray(P[k]).shape, k
        assert_close(ga[k], gb[k], "padded vs native " + k, rtol=2e-5)
    # the padding itself: exactly zero
    assert float(a.lt.t[:, dim:].abs().max()) == 0.0 and float(a.di.t[:, dim:].abs().max()) == 0.0 and float(a.vs.t[:, dim:].abs().max()) == 0.0
    assert float(a.wh.t[:, dim:, :].abs().max()) == 0.0 and float(a.wh.t[:, :, dim:].abs().max()) == 0.0 and float(a.bi.t[:, dim:].abs().max()) == 0.0
    ui = a.ui.t
    assert float(ui[:, dim:, :].abs().max()) == 0.0 and float(ui[:, :, dim:K].abs().max()) == 0.0 and float(ui[:, :, K + dim:].abs().max()) == 0.0
    for m in (a, b):
        m.update_trained_items(); m.update_trained_dists()
    (ha, sa), (hb, sb) = a.predict(users), b.predict(users)
    assert ha.shape == (24, dim) and sa.shape == (24, 38)
    assert_close(ha, hb, "hts", rtol=2e-5); assert_close(sa, sb, "sts", rtol=2e-5)
    a.update_trained_users(ha); b.update_trained_users(hb)          # logical (n, D) rows in
    assert float(a.trained_users.t[:, dim:].abs().max()) == 0.0
    ia, ib = a.compute_sub_topk(users, 10, return_scores=True), b.compute_sub_topk(users, 10, return_scores=True)
    assert_close(ia[1].cpu().numpy(), ib[1].cpu().numpy(), "top-K scores", rtol=2e-5)
    # load_params round trip through the logical shapes
    vals = [getattr(a, k).get_value() for k in ("loss_weight", "wd", "lt", "di", "ui", "wh", "bi", "vs", "bs")]
    c = _spatial_model(pa, T, P)
    c.load_params(vals)
    for k in SP_NAMES:
        assert np.array_equal(np.asarray(_get(c, SP_NAMES)[k]), np.asarray(ga[k])), k


@pytest.mark.parametrize("dim,n_dist,N", [(64, 0, 5000), (128, 0, 20000), (128, 200, 9000), (64, 300, 3000)])
def test_two_stage_topk_is_bitwise_the_one_stage_result(pa, dim, n_dist, N):
    """poi_ctx_set_topk_filter: a SEEDED fused top-K runs an f16 filter pass (v_mfma_f32_32x32x16_f16 on half-rounded users / items, a
    rigorous bound on |approximate - float32 score|) and rescores the survivors with the one-stage kernel's own float32 MFMA sequence:
    ids AND scores must equal the one-stage kernel's bit for bit - for good seeds (almost nothing survives), seeds from a moved model,
    random seeds and malformed rows (survivor lists overflow: the flagged tiles take the one-stage kernel), with and without the
    distance term (uint8 / uint16 bin matrix), ragged user counts, and against the float64 oracle on gap-checked rows."""
    import torch
    ctx = pa._lib.context(0)
    rng = np.random.default_rng(500 + dim + n_dist)
    n, K = 421, 20
    users = (rng.standard_normal((n, dim)) * 0.4).astype(np.float32)
    items = (rng.standard_normal((N, dim)) * 0.5).astype(np.float32)
    items[::97] *= 1e-6                                        # rows in the half-precision subnormal range
    du, di = torch.as_tensor(users).cuda(), torch.as_tensor(items).cuda()
    bins = None
    if n_dist:
        coords = np.stack([40.0 + rng.random(N) * 0.3, -74.0 + rng.random(N) * 0.3], 1)
        last = rng.integers(0, N, n).astype(np.int32)
        from poi_amd.data import bin_thresholds, cos_lat
        dd = 200.0
        thr = torch.as_tensor(bin_thresholds(dd, n_dist)).cuda(); cph = torch.as_tensor(cos_lat(coords)).cuda()
        dc = torch.as_tensor(coords).cuda(); dl = torch.as_tensor(last).cuda()
        bb = 1 if n_dist <= 255 else 2
        ntile = (N + 31) // 32
        bins = torch.zeros(((n + 31) // 32) * ntile * 1024 * bb, dtype=torch.uint8, device="cuda")
        ctx.check(ctx.lib.poi_ulptai_build(ctx.handle, dc.data_ptr(), cph.data_ptr(), thr.data_ptr(), dl.data_ptr(), n, N, n_dist, dd, bins.data_ptr(), bb, None))
        sts = rng.random((((n + 31) // 32) * 32, n_dist + 1)).astype(np.float32); sts /= sts.sum(axis=1, keepdims=True); sts[:, n_dist] = 0.0
        dsts = torch.as_tensor(sts).cuda()
        dwd = torch.as_tensor(np.array([3.5], np.float32)).cuda()

    def run(seed, two_stage, uu=du):
        idx = torch.full((n, K), -7, dtype=torch.int32, device="cuda")
        sc = torch.zeros((n, K), dtype=torch.float32, device="cuda")
        ctx.set_topk_filter(two_stage)
        if seed is not None:
            ctx.set_topk_seed(seed, seed.shape[1])
        try:
            if n_dist:
                ctx.check(ctx.lib.poi_score_topk_ulptai(ctx.handle, uu.data_ptr(), di.data_ptr(), n, N, dim, dwd.data_ptr(), dsts.data_ptr(), bins.data_ptr(), bb,
                                                        n_dist, K, idx.data_ptr(), sc.data_ptr(), None))
            else:
                ctx.check(ctx.lib.poi_score_topk(ctx.handle, uu.data_ptr(), di.data_ptr(), n, N, dim, None, None, K, idx.data_ptr(), sc.data_ptr(), None))
        finally:
            ctx.set_topk_filter(True)
        return idx.cpu().numpy(), sc.cpu().numpy()

    base_idx, base_sc = run(None, False)
    if not n_dist:
        full = users.astype(np.float64) @ items.astype(np.float64).T
        exp = O.topk_desc(full, K)
        srt = np.sort(full, axis=1)[:, ::-1][:, :K + 1]
        ok = np.min(srt[:, :-1] - srt[:, 1:], axis=1) > 1e-5 * np.abs(srt).max()
        assert ok.sum() > n // 2 and np.array_equal(base_idx[ok], exp[ok])
    good = torch.as_tensor(base_idx).cuda()
    # the lists of a slightly different model (what the next evaluation's seeds are)
    moved = torch.as_tensor(users + (rng.standard_normal(users.shape) * 0.02).astype(np.float32)).cuda()
    prev_idx, _ = run(None, False, uu=moved)
    prev = torch.as_tensor(prev_idx).cuda()
    rnd = torch.as_tensor(np.stack([rng.choice(N, K, replace=False) for _ in range(n)]).astype(np.int32)).cuda()
    bad = good.clone(); bad[::3, 5] = bad[::3, 4]; bad[1::3, 0] = N + 7
    # (unseeded: the two-stage path seeds itself with the one-stage kernel on the first 1/16 of the item tiles - tables of >= 256 tiles)
    for name, seed in (("true top-K", good), ("previous model's lists", prev), ("random items", rnd), ("malformed rows", bad), ("unseeded", None)):
        one_idx, one_sc = run(seed, False)
        two_idx, two_sc = run(seed, True)
        assert np.array_equal(one_idx, base_idx) and np.array_equal(one_sc, base_sc), name
        assert np.array_equal(two_idx, base_idx), "two-stage ids differ: " + name
        assert np.array_equal(two_sc.view(np.uint32), base_sc.view(np.uint32)), "two-stage scores differ: " + name


@pytest.mark.parametrize("dim,f16", [(128, False), (256, False), (256, True)])
def test_two_stage_geo_topk_is_bitwise_the_one_stage_result(pa, dim, f16):
    """The two-stage path of poi_score_topk_geo (bins computed on the fly, config X's dim 256 / half table): the f16 filter keeps a pair when
    approximate score + bound + the tile's LARGEST possible distance term can beat the threshold, computes the float64 Haversine bin only
    for those, and the rescoring kernel repeats the one-stage arithmetic: ids and scores bit for bit, for good / moved / random seeds."""
    import torch
    from poi_amd.data import bin_thresholds, cos_lat
    ctx = pa._lib.context(0)
    rng = np.random.default_rng(900 + dim + f16)
    n, N, K, n_dist, dd = 333, 9000, 20, 200, 200.0
    users = (rng.standard_normal((n, dim)) * 0.3).astype(np.float32)
    items = (rng.standard_normal((N, dim)) * 0.4).astype(np.float32)
    coords = np.stack([40.0 + rng.random(N) * 0.3, -74.0 + rng.random(N) * 0.3], 1)
    last = rng.integers(0, N, n).astype(np.int32)
    du = torch.as_tensor(users).cuda()
    di = torch.as_tensor(items).cuda().to(torch.float16 if f16 else torch.float32).contiguous()
    if f16:
        ctx.register_f16(di)
    thr = torch.as_tensor(bin_thresholds(dd, n_dist)).cuda(); cph = torch.as_tensor(cos_lat(coords)).cuda()
    dc = torch.as_tensor(coords).cuda(); dl = torch.as_tensor(last).cuda()
    sts = rng.random((((n + 31) // 32) * 32, n_dist + 1)).astype(np.float32) ** 4; sts /= sts.sum(axis=1, keepdims=True); sts[:, n_dist] = 0.0
    dsts = torch.as_tensor(sts).cuda()
    dwd = torch.as_tensor(np.array([6.0], np.float32)).cuda()

    def run(seed, two_stage, uu=du):
        idx = torch.full((n, K), -7, dtype=torch.int32, device="cuda")
        sc = torch.zeros((n, K), dtype=torch.float32, device="cuda")
        ctx.set_topk_filter(two_stage)
        if seed is not None:
            ctx.set_topk_seed(seed, seed.shape[1])
        try:
            ctx.check(ctx.lib.poi_score_topk_geo(ctx.handle, uu.data_ptr(), di.data_ptr(), n, N, dim, dwd.data_ptr(), dsts.data_ptr(), dc.data_ptr(), cph.data_ptr(),
                                                 thr.data_ptr(), dl.data_ptr(), n_dist, dd, K, idx.data_ptr(), sc.data_ptr(), None))
        finally:
            ctx.set_topk_filter(True)
        return idx.cpu().numpy(), sc.cpu().numpy()

    try:
        base_idx, base_sc = run(None, False)
        good = torch.as_tensor(base_idx).cuda()
        moved = torch.as_tensor(users + (rng.standard_normal(users.shape) * 0.02).astype(np.float32)).cuda()
        prev = torch.as_tensor(run(None, False, uu=moved)[0]).cuda()
        rnd = torch.as_tensor(np.stack([rng.choice(N, K, replace=False) for _ in range(n)]).astype(np.int32)).cuda()
        for name, seed in (("true top-K", good), ("previous model's lists", prev), ("random items", rnd), ("unseeded", None)):
            one_idx, one_sc = run(seed, False)
            assert np.array_equal(one_idx, base_idx) and np.array_equal(one_sc, base_sc), name
            # the user-stationary filter (every user tile walks the item table) and the item-stationary one (config X's shape: a workgroup
            # keeps four item tiles and walks the user tiles) - forced here, chosen by shape in production
            for mode in ("users", "items"):
                two_idx, two_sc = run(seed, mode)
                assert np.array_equal(two_idx, base_idx), "two-stage GEO ids differ: %s (%s-stationary filter)" % (name, mode)
                assert np.array_equal(two_sc.view(np.uint32), base_sc.view(np.uint32)), "two-stage GEO scores differ: %s (%s-stationary filter)" % (name, mode)
        st = ctx.topk_filter_stats()
        assert st["users"] == n
    finally:
        if f16:
            ctx.unregister_f16(di)
