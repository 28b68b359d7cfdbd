"""-m gpu: BASELINE.json's full shapes.  (1) The TIMED path against the oracle: a 12500-user launch of the
Gowalla shape in EXACTLY bench.py's configuration (configs[2]: data with 80 % local transitions - skewed distance
bins -, batch cap 64, users sorted by length) and the whole-shard launch + a sequential reference-schedule
run of the Foursquare shape (configs[1]) are compared with the plain-C float64 restatement of the batch rule /
the sequential epoch (oracle/poi_oracle_c.c, threaded over the launch), all nine tensors, weights within 1e-5
AND updates within 1e-4 of every ROW's own absolute mass (tests/gpu_util.delta_excess with absmass) - for the timed
tile engine (exact forward pass, float32 behind it) with no loosening of any tensor, 300 sequential steps inside 1e-5, and for the
EXACT engine (float64 arithmetic, poi_ctx_set_engine(4)): every update inside 1e-6 of its mass.  (2) Size-independent properties at the Gowalla shape (100 k POIs, 50 k users,
L <= 50, D = 128, 200 bins):
  * a 12500-user launch leaves every table row that no sequence of the launch touches bit-identical,
    moves every touched row, keeps everything finite and is bitwise reproducible (lt / di);
  * the per-sequence losses of a launch do not depend on which other sequences share the launch
    (forward values are evaluated at the launch-entry parameters): a 2048-user launch reports the same
    losses for its users as the 12500-user launch did;
  * fused scoring + top-K over all 100 k POIs == torch.topk of the explicit f32 score matrix (indices on
    rows whose top-21 scores are well separated), with and without the resident distance-bin matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    from poi_amd import data as pdata
    n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=77, local=0.8)      # bench.py's generator setting (--local 0.8)
    tab = ds.shard(0, n_user)

    def model():
        return poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                            n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, seed=3, coords=ds.coords)
    return poi_amd, ds, tab, model


def test_full_size_launch_touches_exactly_its_rows_and_is_reproducible(setup):
    import torch
    pa, ds, tab, make = setup
    rng = np.random.default_rng(5)
    users = rng.permutation(ds.n_user)[:12500].astype(np.int32)
    off = tab.off.astype(np.int64)
    sel = np.concatenate([np.arange(off[u], off[u + 1]) for u in users])
    touched = np.zeros(ds.n_item + 1, bool)
    touched[tab.p[sel]] = True; touched[tab.q[sel]] = True
    touched[ds.n_item] = True                                   # padding row: every user shorter than len_max
    runs = []
    for _ in range(2):
        m = make()
        lt0, di0 = m.lt.t.clone(), m.di.t.clone()
        out = m.train_batch(users)
        runs.append((m.lt.t.clone(), m.di.t.clone(), out, [getattr(m, k).t.clone() for k in ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")]))
    lt1, di1, out, _ = runs[0]
    assert torch.isfinite(lt1).all() and torch.isfinite(di1).all() and np.isfinite(out).all()
    same = (lt1 == lt0).all(dim=1).cpu().numpy()
    assert np.array_equal(~same, touched), "rows changed != rows touched"
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]), "lt / di differ between identical launches"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][3], runs[1][3])), "dense parameters differ between identical launches"
    assert np.array_equal(runs[0][2], runs[1][2]), "losses differ between identical launches"
    # launch composition does not change a sequence's forward values
    m = make()
    sub = users[:2048]
    out2 = m.train_batch(sub)
    assert np.allclose(out2[:, :3], out[:2048, :3], rtol=2e-5, atol=1e-6)


def test_full_size_topk_matches_explicit_scores(setup):
    import torch
    pa, ds, tab, make = setup
    m = make()
    m.update_trained_items(); m.update_trained_dists()
    ids = np.arange(0, 2048, dtype=np.int32)
    hts, sts = m.predict_device(np.arange(ds.n_user, dtype=np.int32))
    m.update_trained_users(hts)
    K = 20

    def check(idx, full):
        top = torch.topk(full, K + 1, dim=1)
        gap = (top.values[:, :-1] - top.values[:, 1:]).min(dim=1).values
        ok = gap > 1e-4 * top.values.abs().max()
        assert ok.sum() > 0.5 * len(ids)
        assert torch.equal(idx[ok].long(), top.indices[ok][:, :K])

    # no distance term
    idx = m.compute_sub_topk(ids, K)
    full = m.trained_users.t[:2048] @ m.trained_items.t[:ds.n_item].T
    check(idx, full)
    # distance term through the resident bin matrix vs the dense prob rows of poi_dist_prob
    m.update_trained_sus(sts)
    idx2 = m.compute_sub_topk(ids, K)
    wd, prob = pa.models.OboSpatialGru._prob_rows(m, torch.as_tensor(ids).cuda(), 0)
    full2 = full + wd[0] * prob
    check(idx2, full2)


def test_full_size_rank_agreement_with_the_float64_oracle(setup):
    """SURVEY section 7 "bit-exact ranks vs precision", public/Valuate.py:132-146 at the Gowalla shape: the device's top-20 lists of 2048 users over all
    100 k POIs (fused float16 filter + exact float32 rescoring) against the FLOAT64 oracle's (oracle.score_all + oracle.topk_desc from the same
    float32 user vectors / snapshot table), after one epoch of training, without and with the distance term.  Reported: the fraction of
    identical lists and of identical recall@{5, 10, 15, 20} flags; asserted: >= 99.5 % identical lists, EVERY differing position is a swap of
    two POIs whose float64 scores differ by less than 1e-6 of the row's largest score (below float32 resolution), recall flags identical on
    >= 99.9 % of the users."""
    import torch
    from oracle import poi_oracle as O
    from poi_amd import data as pdata
    pa, ds, tab, make = setup
    m = make()
    m.ctx.set_batch_cap(64.0)
    try:
        order = np.random.default_rng(1).permutation(ds.n_user).astype(np.int32)
        for b0 in range(0, ds.n_user, 12500):
            m.train_batch(order[b0:b0 + 12500], sync=False)
    finally:
        m.ctx.set_batch_cap(1.0)
    m.update_trained_items(); m.update_trained_dists()
    hts, sts = m.predict_device(np.arange(ds.n_user, dtype=np.int32))
    m.update_trained_users(hts)
    n, K = 2048, 20
    ids = np.arange(n, dtype=np.int32)
    users64 = hts[:n].double().cpu().numpy()
    items64 = m.trained_items.t.double().cpu().numpy()          # (n_item + 1, D): score_all drops the padding row
    sts64 = sts[:n].double().cpu().numpy()
    wd = float(m.wd.get_value())
    tes = np.asarray(tab.tes_p).reshape(-1)[:n]
    lens = np.diff(tab.off.astype(np.int64))[:n]
    last = np.asarray(tab.p)[tab.off[:n].astype(np.int64) + lens - 1]      # last train POI of every user

    def compare(idx, with_dist):
        """-> (identical lists, identical recall flags) per user; every differing position must be a near-tie in the oracle's float64 scores"""
        same = np.zeros(n, bool); rec_same = np.ones(n, bool)
        for c0 in range(0, n, 256):
            sl = slice(c0, c0 + 256)
            prob = None
            if with_dist:
                prob = np.empty((256, ds.n_item))
                for r, u in enumerate(range(c0, c0 + 256)):
                    b = pdata.cal_dis_vec(ds.coords[last[u], 0], ds.coords[last[u], 1], ds.coords[:, 0], ds.coords[:, 1], ds.dd, ds.dist_num)
                    prob[r] = np.where(b < ds.dist_num, sts64[u][np.minimum(b, ds.dist_num)], 0.0)
            sc = O.score_all(users64[sl], items64, wd, prob)
            cand = np.argpartition(-sc, K + 8, axis=1)[:, :K + 8]            # the oracle's top-K rule on the K + 8 best columns (they contain the top K)
            exp = np.take_along_axis(cand, O.topk_desc(np.take_along_axis(sc, cand, axis=1), K), axis=1)
            same[sl] = (idx[sl] == exp).all(axis=1)
            for k in (5, 10, 15, 20):
                rec_same[sl] &= (idx[sl][:, :k] == tes[sl, None]).any(axis=1) == (exp[:, :k] == tes[sl, None]).any(axis=1)
            for r in np.nonzero(~same[sl])[0]:
                pos = np.nonzero(idx[c0 + r] != exp[r])[0]
                d = np.abs(sc[r][idx[c0 + r][pos]] - sc[r][exp[r][pos]]).max()
                assert d < 1e-6 * np.abs(sc[r][exp[r]]).max(), "user %d: lists differ where the float64 scores are %.2e apart" % (c0 + r, d)
        return same, rec_same

    for with_dist in (False, True):
        if with_dist:
            m.update_trained_sus(sts)
        idx = m.compute_sub_topk(ids, K).cpu().numpy().astype(np.int64)
        same, rec_same = compare(idx, with_dist)
        print("rank agreement vs float64 (distance term %s): identical top-20 lists %.4f, identical recall@{5,10,15,20} %.4f" % (with_dist, same.mean(), rec_same.mean()))
        assert same.mean() >= 0.995 and rec_same.mean() >= 0.999


SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


def _state(m):
    out = {}
    for k in SP_NAMES:
        v = getattr(m, k).get_value()
        out[k] = float(v) if k == "wd" else np.asarray(v, np.float64)
    return out


BENCH_CAP = 64.0      # bench.py --batch-cap


def _gowalla_launch(setup, engine):
    """One 12500-user launch of bench.py's configuration (local transitions, cap 64, length-sorted) on `engine` and the float64 oracle
    of the same launch with the rows' absolute masses."""
    from oracle import c_oracle as C
    pa, ds, tab, make = setup
    users = np.random.default_rng(5).permutation(ds.n_user)[:12500].astype(np.int32)
    lens = np.diff(tab.off.astype(np.int64))
    users = users[np.argsort(-lens[users], kind="stable")]            # bench.py sorts a launch by length
    m = make()
    P = _state(m)
    P["h0"] = np.zeros(P["lt"].shape[1])
    m.ctx.set_engine(engine); m.ctx.set_batch_cap(BENCH_CAP)
    try:
        out = np.asarray(m.train_batch(users))
    finally:
        m.ctx.set_engine("auto"); m.ctx.set_batch_cap(1.0)
    got = _state(m)
    key = "gowalla_oracle"
    if key not in _CACHE:
        _CACHE[key] = C.spatial_batch_mean(P, tab.off, tab.p, tab.q, tab.dp, tab.dq, users, tab.len_max, 0.01, 0.001, cap=BENCH_CAP, absmass=True)
    exp, eout, touched = _CACHE[key]
    return P, out, got, exp, eout, touched


_CACHE = {}


def test_gowalla_timed_launch_matches_the_oracle_batch_rule(setup):
    """The tile engine (exact forward pass, per-bin tables, per-POI regrouping, three-level per-bin sums, sorted scatter, te_dapply) on the
    TIMED configuration against the float64 oracle of the capped-sum rule: per-sequence losses to 1e-5, ALL nine tensors to 1e-5 of their
    max-norm - every row of the POI table - and every row's UPDATE to 1e-4 of the row's own absolute mass."""
    from tests.gpu_util import assert_close, assert_step_close, rows_within
    P, out, got, exp, eout, touched = _gowalla_launch(setup, "auto")
    assert_close(out[:, :3], eout[:, :3], "losses of the launch")
    worst = assert_step_close(got, exp, P, SP_NAMES, "gowalla 12500-user launch", absmass=touched["absmass"])
    assert rows_within(got["lt"], exp["lt"]) == 1.0, "POI rows miss the 1e-5 bar"
    assert np.array_equal((got["lt"] != P["lt"]).any(axis=1), touched["lt"])
    print("gowalla launch vs oracle (tile engine, exact forward): worst weight rel err %.2e" % worst)


def test_gowalla_timed_launch_exact_engine_meets_the_contract_on_every_row(setup):
    """The same launch on the exact engine: NO loosening - all nine tensors within 1e-5 (max-norm, i.e. every row), every row's
    update within 1e-6 of its absolute mass + the float32 storage rounding, losses to 1e-6."""
    from tests.gpu_util import EXACT_DELTA_RTOL, assert_close, assert_step_close, rows_within
    P, out, got, exp, eout, touched = _gowalla_launch(setup, "exact")
    assert_close(out[:, :3], eout[:, :3], "losses of the launch (exact)", rtol=1e-6)
    worst = assert_step_close(got, exp, P, SP_NAMES, "gowalla 12500-user launch, exact engine", delta_rtol=EXACT_DELTA_RTOL, absmass=touched["absmass"])
    assert rows_within(got["lt"], exp["lt"], rtol=1e-6) == 1.0
    assert np.array_equal((got["lt"] != P["lt"]).any(axis=1), touched["lt"])
    print("gowalla launch vs oracle (exact engine): worst weight rel err %.2e" % worst)


def test_delta_bar_catches_a_padding_multiplicity_off_by_one_at_the_gowalla_shape(setup):
    """What the per-row bar is for, at full size.  A write-back whose L2 multiplicity of a padding row is off by one in every
    sequence (2 (len_max - L) + 1) shifts the row by min(k, cap) alpha lambda |row| = 64 x 5e-6; the exact engine's bar (1e-6 of the row's absolute mass) flags it on lt AND on di (whose padding row also sums every
    sequence's position-0 input gradient).  And ONE L2-decay term (alpha lambda |row| = 5e-6, the mean rule's off-by-one) on
    lt[n_item] is over the float32 engines' per-row bar, while the old whole-tensor term (1e-5 of the largest update of the
    tensor: single-occurrence rows move by 0.2 - 0.6 here) was as large as the error itself."""
    from tests.gpu_util import EXACT_DELTA_RTOL, assert_close, delta_excess
    P, out, got, exp, eout, touched = _gowalla_launch(setup, "exact")
    pa, ds, tab, make = setup
    for name, pad in (("lt", ds.n_item), ("di", ds.dist_num)):
        bad = np.array(exp[name], copy=True)
        bad[pad] -= BENCH_CAP * 0.01 * 0.001 * P[name][pad]
        ex, row = delta_excess(bad, exp[name], P[name], rtol=EXACT_DELTA_RTOL, absmass=touched["absmass"][name])
        assert ex > 3.0 and row == pad, (name, ex, row)
        print("systematic off-by-one on %s[pad] under cap 64: %.1fx over the exact engine's per-row bar" % (name, ex))
    bad = np.array(exp["lt"], copy=True)
    bad[ds.n_item] -= 0.01 * 0.001 * P["lt"][ds.n_item]
    assert_close(bad, exp["lt"], "lt with one extra L2-decay term")          # the weight bar does not see it
    ex_new, row = delta_excess(bad, exp["lt"], P["lt"], absmass=touched["absmass"]["lt"])
    ex_old, _ = delta_excess(bad, exp["lt"], P["lt"])
    assert ex_new > 1.5 and row == ds.n_item, (ex_new, row)
    print("one L2-decay term on lt[pad]: %.2fx over the per-row bar (old whole-tensor bar: %.2fx)" % (ex_new, ex_old))


def test_foursquare_shape_full_size_against_the_oracle():
    """configs[1] (10 k POIs, 5 k users, L <= 20, D = 64 - the two-table path of the tile engine) at FULL size:
    (a) the whole shard in one launch == the oracle's batch rule; (b) the reference schedule (one user per step,
    prog_bpr_gru_spatial.py:249-250) over 300 users of the shuffled order == the sequential float64 epoch, every tensor inside
    1e-5 after the 300 steps; (a') / (b') the same on the EXACT engine - updates inside 1e-6 of their mass, predict to 2e-7;
    (c) predict + all-POI top-20 == float64 scores' ranks on gap-checked rows."""
    import torch
    import poi_amd
    from oracle import c_oracle as C
    from oracle import poi_oracle as O
    from poi_amd import data as pdata
    from tests.gpu_util import EXACT_DELTA_RTOL, assert_close, assert_step_close, rows_within
    n_item, n_user, max_len, D = pdata.SHAPES["foursquare"]
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=78, local=0.8)
    tab = ds.shard(0, n_user)

    def make():
        return poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                            n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, seed=4, coords=ds.coords)
    m = make()
    P = _state(m); P["h0"] = np.zeros(D)
    users = np.random.default_rng(6).permutation(n_user).astype(np.int32)
    out = np.asarray(m.train_batch(users))
    got = _state(m)
    exp, eout, tch = C.spatial_batch_mean(P, tab.off, tab.p, tab.q, tab.dp, tab.dq, users, tab.len_max, 0.01, 0.001, absmass=True)
    assert_close(out[:, :3], eout[:, :3], "losses")
    assert_step_close(got, exp, P, SP_NAMES, "foursquare whole-shard launch", absmass=tch["absmass"])
    assert rows_within(got["lt"], exp["lt"]) == 1.0
    # (a') the same launch on the exact engine: the contract on every row, no loosening
    m = make()
    m.ctx.set_engine("exact")
    try:
        out = np.asarray(m.train_batch(users))
        got_x = _state(m)
        assert_close(out[:, :3], eout[:, :3], "losses (exact)", rtol=1e-6)
        assert_step_close(got_x, exp, P, SP_NAMES, "foursquare whole-shard launch, exact engine", delta_rtol=EXACT_DELTA_RTOL, absmass=tch["absmass"])
        # (b') reference schedule on the exact engine: 300 sequential steps stay inside the ONE-step bar
        m = make()
        Pc = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in P.items()}
        order = users[:300]
        eo = C.spatial_epoch(Pc, tab.off, tab.p, tab.q, tab.dp, tab.dq, order, tab.len_max, 0.01, 0.001)
        go = np.array([[r[0], r[1], r[2]] for r in (m.train(np.int32(u)) for u in order)])
        assert_close(go, eo[:, :3], "sequential losses (exact)", rtol=1e-5)
        got_x = _state(m)
        for k in SP_NAMES:
            assert_close(got_x[k], Pc[k], "after 300 sequential steps (exact): " + k, rtol=1e-5)
        # predict on the exact engine
        m.update_trained_items(); m.update_trained_dists()
        sub = np.arange(0, n_user, 16).astype(np.int32)
        hx, sx = m.predict(sub)
    finally:
        m.ctx.set_engine("auto")
    off = tab.off.astype(np.int64)
    rows = lambda flat, pad: [np.r_[flat[off[u]:off[u + 1]], np.full(max_len - (off[u + 1] - off[u]), pad)] for u in sub]
    masks = [np.r_[np.ones(off[u + 1] - off[u], int), np.zeros(max_len - (off[u + 1] - off[u]), int)] for u in sub]
    eh, es = O.spatial_predict(got_x | {"h0": np.zeros(D)}, got_x["lt"], got_x["di"], rows(tab.p, n_item), rows(tab.dp, ds.dist_num), masks)
    assert_close(hx, eh, "hts (exact)", rtol=2e-7); assert_close(sx, es, "sts (exact)", rtol=2e-7)
    # (b) reference schedule on the timed engine (exact forward pass): 300 sequential steps stay inside the ONE-step bar
    m = make()
    Pc = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in P.items()}
    eo = C.spatial_epoch(Pc, tab.off, tab.p, tab.q, tab.dp, tab.dq, order, tab.len_max, 0.01, 0.001)
    go = np.array([[r[0], r[1], r[2]] for r in (m.train(np.int32(u)) for u in order)])
    assert_close(go, eo[:, :3], "sequential losses")
    got = _state(m)
    for k in SP_NAMES:
        assert_close(got[k], Pc[k], "after 300 sequential steps: " + k)
    # (c) predict + score + top-K at full size
    m.update_trained_items(); m.update_trained_dists()
    ids = np.arange(n_user, dtype=np.int32)
    hts, sts = m.predict(ids)
    sub = np.arange(0, n_user, 16)
    off = tab.off.astype(np.int64)
    rows = lambda flat, pad: [np.r_[flat[off[u]:off[u + 1]], np.full(max_len - (off[u + 1] - off[u]), pad)] for u in sub]
    masks = [np.r_[np.ones(off[u + 1] - off[u], int), np.zeros(max_len - (off[u + 1] - off[u]), int)] for u in sub]
    eh, es = O.spatial_predict(got | {"h0": np.zeros(D)}, got["lt"], got["di"], rows(tab.p, n_item), rows(tab.dp, ds.dist_num), masks)
    assert_close(hts[sub], eh, "hts"); assert_close(sts[sub], es, "sts")
    m.update_trained_users(hts)
    idx = m.compute_sub_topk(ids, 20).cpu().numpy()
    sc = np.asarray(hts, np.float64) @ got["lt"][:-1].T
    top = O.topk_desc(sc[sub], 21)
    tv = np.take_along_axis(sc[sub], top, axis=1)
    ok = (tv[:, :-1] - tv[:, 1:]).min(axis=1) > 1e-5 * np.abs(tv).max()
    assert ok.sum() > 0.8 * len(sub)
    assert np.array_equal(idx[sub][ok], top[ok][:, :20])
