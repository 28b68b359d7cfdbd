"""-m gpu: BASELINE.json configs[4] at its TIMED launch size (bench.py --shape x1: 15625 users per launch, L <= 50, D = 256, 10 M-row half POI
table, batch cap 64, length-sorted - float64-MFMA forward recurrence, streaming backward kernels on split products), with the
REFERENCE'S OWN INIT, against the oracle:
  * the launch: per-sequence losses, `di` and the seven dense tensors against oracle/c_oracle.spatial_batch_mean (the float64 capped-sum
    rule) re-stated on the COMPACT table of the launch's POIs - the sequences are drawn over 300 k POI ids that are then spread over the
    10 M rows of the real table (the oracle cannot hold 10 M x 256 doubles), so the compact problem IS the drawn one; touched rows ==
    half(oracle) to one half ulp + the float32 noise bar; every untouched row of the 10 000 001 bit-identical;
  * evaluation: poi_score_topk_geo over all 10 M POIs against scores built from the ORACLE's predict (oracle.spatial_predict on the
    compact tables -> hts, sts), float64 products with the snapshot table and the distance term sts[bin] from the reference's Haversine
    expression vectorised in data.cal_dis_vec (pinned to the reference's cal_dis by tests/golden) - nothing of the product's own predict / prob rows on the
    expected side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ITEM, DIM, N_DIST = 10_000_000, 256, 200
N_SMALL, N_USER, LEN_MAX, CAP = 300_000, 15625, 50, 64.0
SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


def _f16(a):
    return np.asarray(a, np.float16).astype(np.float64)


@pytest.fixture(scope="module")
def launch():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    from tests.gpu_util import toy_problem
    T = toy_problem(777, n_user=N_USER, n_item=N_SMALL, n_dist=N_DIST, dim=DIM, len_max=LEN_MAX, min_len=4, hot=N_SMALL // 2)
    rng = np.random.default_rng(5)
    big_of = np.sort(rng.choice(N_ITEM, N_SMALL, replace=False)).astype(np.int64)      # compact id -> row of the real table (order kept)
    big_of = np.append(big_of, N_ITEM)                                                # the padding id
    Tb = dict(T)
    Tb["train"] = [big_of[np.asarray(T["train"][0])], T["train"][1], big_of[np.asarray(T["train"][2])]]
    Tb["test"] = [big_of[np.asarray(T["test"][0])], T["test"][1], big_of[np.asarray(T["test"][2])]]
    coords = np.stack([40.0 + rng.random(N_ITEM) * 0.36, -74.0 + rng.random(N_ITEM) * 0.47], 1)
    model = poi_amd.models.OboSpatialGru(train=Tb["train"], test=Tb["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=N_USER,
                                         n_item=N_ITEM, n_dists=[N_DIST, 0.2], n_in=DIM, n_hidden=DIM, seed=6, table_dtype="f16", coords=coords)
    # CONDITIONING.  The parameters keep the reference's own init (uniform(-0.5, 0.5): public/GRU_Spatial.py:50-71, public/GRU.py:60-62).  At dim 256
    # that recurrence is violently expansive - a float32 forward pass is 2e-3 off in the losses of the 40 .. 50-position sequences and O(1) off
    # in the updates (tools/x256_check.py), the gradients explode (updates of 10^2 .. 10^3 on weights of 0.5) - which is why the forward pass
    # of this configuration runs in float64 on the matrix cores (te_rec_fwdd, te_xfwd.hip) behind an int8-digit input product: every tensor
    # lands inside the standard bars below.  (Rounds 3 - 4 scaled ui / wh by 0.2 here to make the problem contractive: gone.)
    return poi_amd, T, big_of, coords, model


def _dense_state(m):
    out = {}
    for k in SP_NAMES:
        if k != "lt":
            v = getattr(m, k).get_value()
            out[k] = float(v) if k == "wd" else np.asarray(v, np.float64)
    return out


def test_configx_timed_launch_against_the_oracle(launch):
    import torch
    from oracle import c_oracle as C
    from poi_amd.data import padded_to_csr
    from tests.gpu_util import assert_close, assert_step_close
    pa, T, big_of, coords, m = launch
    rows = torch.as_tensor(big_of).cuda()
    lens = np.asarray(T["lens"])
    users = np.argsort(-lens, kind="stable").astype(np.int32)            # bench.py sorts a launch by length
    lt0 = m.lt.t.clone()
    before = _dense_state(m)
    before["lt"] = m.lt.t[rows].float().cpu().numpy().astype(np.float64)            # compact table: row i = real row big_of[i], last = padding row
    m.ctx.set_batch_cap(CAP)
    try:
        out = np.asarray(m.train_batch(users))
    finally:
        m.ctx.set_batch_cap(1.0)
    assert np.isfinite(out).all()
    off, p = padded_to_csr(np.asarray(T["train"][0]), lens); _, q = padded_to_csr(np.asarray(T["train"][2]), lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    Pin = dict(before); Pin["h0"] = np.zeros(DIM)
    exp, eout, tch = C.spatial_batch_mean(Pin, off, p, q, dp, dq, users, T["len_max"], 0.01, 0.001, cap=CAP, threads=8)
    # untouched rows of the whole table bit-identical; the launch's rows moved
    changed = (m.lt.t != lt0).any(dim=1)
    is_t = torch.zeros(N_ITEM + 1, dtype=torch.bool, device="cuda"); is_t[rows[torch.as_tensor(tch["lt"]).cuda()]] = True
    assert not bool((changed & ~is_t).any()), "a row no sequence of the launch touches changed"
    assert int(changed.sum()) > 0.9 * int(tch["lt"].sum())
    del lt0
    # the standard bars - 1e-5 on the weights, 1e-4 per row on the updates
    import os
    got = _dense_state(m)
    if os.environ.get("X_DIAG"):
        from tests.gpu_util import rel_err, delta_excess
        L = lens[users]
        for lo, hi in ((4, 10), (10, 20), (20, 30), (30, 40), (40, 51)):
            sel = (L >= lo) & (L < hi)
            print("losses L in [%d, %d): n %d  rel err %.2e" % (lo, hi, sel.sum(), rel_err(out[sel, :3], eout[sel, :3])))
        for k in SP_NAMES:
            if k != "lt":
                print(k, "w %.2e  delta excess(3e-4) %.2f" % (rel_err(got[k], exp[k]), delta_excess(got[k], exp[k], before[k], rtol=3e-4)[0]))
    assert_close(out[:, :3], eout[:, :3], "losses")
    assert_step_close(got, exp, before, [k for k in SP_NAMES if k != "lt"], "config X timed launch, dense tensors + di")
    lt = m.lt.t[rows].float().cpu().numpy().astype(np.float64)
    want = _f16(exp["lt"])
    ulp = np.spacing(np.abs(want).astype(np.float16)).astype(np.float64)
    noise = 1e-5 * np.abs(want).max()
    assert (np.abs(lt - want) <= ulp + noise).all(), "a touched row differs from half(oracle) by more than one half ulp"
    assert (lt == want).mean() > 0.97


def test_configx_geo_topk_against_oracle_side_scores(launch):
    import torch
    from oracle import poi_oracle as O
    from poi_amd import data as pdata
    pa, T, big_of, coords, m = launch
    n = 32
    ids = np.arange(n, dtype=np.int32)
    m.update_trained_items(); m.update_trained_dists()
    hts, sts = m.predict_device(np.arange(N_USER, dtype=np.int32))
    m.update_trained_users(hts); m.update_trained_sus(sts)
    idx, sc = m.compute_sub_topk(ids, 20, return_scores=True)
    # expected side: the oracle's predict on the compact tables (float64 from the stored values) ...
    rows = torch.as_tensor(big_of).cuda()
    P = _dense_state(m)
    P["lt"] = m.trained_items.t[rows].float().cpu().numpy().astype(np.float64); P["h0"] = np.zeros(DIM)
    di = np.asarray(m.trained_dists.t.float().cpu().numpy(), np.float64)
    p_rows = np.asarray(T["train"][0])[:n]; d_rows = np.asarray(T["dist"][0])[:n]; masks = np.asarray(T["train"][1])[:n]
    eh, es = O.spatial_predict(P, P["lt"], di, p_rows, d_rows, masks)
    # ... float64 products with the snapshot table, distance term from the exact bins of the last train POI to every POI
    lens = np.asarray(T["lens"])[:n]
    last_small = p_rows[np.arange(n), lens - 1]
    last_big = big_of[last_small]
    items = m.trained_items.t[:N_ITEM]
    full = torch.as_tensor(eh).cuda() @ items.double().T                                  # (n, 10 M) float64
    wd = float(P["wd"])
    for u in range(n):
        # the reference's Haversine expression, vectorised (data.cal_dis_vec: pinned to public/Load_Data_by_length.py:24-42 by tests/golden/cal_dis.npz)
        b = pdata.cal_dis_vec(coords[last_big[u], 0], coords[last_big[u], 1], coords[:, 0], coords[:, 1], 200.0, N_DIST)
        prob = np.where(b < N_DIST, es[u][np.minimum(b, N_DIST)], 0.0)
        full[u] += wd * torch.as_tensor(prob).cuda()
    top = torch.topk(full, 21, dim=1)
    gap = (top.values[:, :-1] - top.values[:, 1:]).min(dim=1).values
    ok = gap > 1e-5 * top.values.abs().max()
    assert int(ok.sum()) >= n // 2
    assert torch.equal(idx[ok].long(), top.indices[ok][:, :20])
    assert torch.allclose(sc[ok].double(), top.values[ok][:, :20], rtol=2e-4, atol=2e-4)
