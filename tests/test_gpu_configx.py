"""-m gpu: BASELINE.json configs[4] at its REAL table size on one GPU - 10 M POIs x dim 256, POI table stored as IEEE half (5.1 GB; element
offsets beyond 2^31) - through the model class and the C-ABI:
  * a launch touches exactly its rows: every other row of the 10 000 001 x 256 half table stays bit-identical, also rows whose element
    offsets exceed 2^31 / 2^32 (ids near the end of the table are planted in the sequences);
  * the touched rows against the float64 oracle: the launch is re-stated on a COMPACT table (the ~4 k touched rows, ids remapped; the
    oracle cannot hold 10 M x 256 doubles) with the capped-sum batch rule - expectation rounded to half, one half ulp + the float32 noise
    bar, as tests/test_gpu_tile_engine.py::test_fp16_poi_table_float32_math; dense tensors to the usual bars;
  * stochastic rounding (poi_ctx_set_f16_rounding): every element lands on one of the two half neighbours of the float32 result, the
    mean over many elements is unbiased, and a decay-only row - bit-identical under round-to-nearest - moves;
  * evaluation without any U x N matrix: poi_score_topk_geo over all 10 M POIs for 64 users == explicit float32 / float64 scores."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_ITEM, DIM, N_DIST = 10_000_000, 256, 200
SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


def _f16(a):
    return np.asarray(a, np.float16).astype(np.float64)


@pytest.fixture(scope="module")
def big():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    from tests.gpu_util import toy_problem
    T = toy_problem(4242, n_user=160, n_item=N_ITEM, n_dist=N_DIST, dim=DIM, len_max=14, hot=N_ITEM)
    P, Q = np.asarray(T["train"][0]), np.asarray(T["train"][2])
    L = T["lens"]
    for u in range(T["n_user"]):                      # ids near the end of the table (element offsets > 2^31, > 2^32) and shared rows
        if L[u] >= 6:
            P[u, 2] = N_ITEM - 5; Q[u, 3] = 9_000_001; P[u, 4] = 8_388_609 + (u % 3)
    T["train"][0], T["train"][2] = P, Q
    rng = np.random.default_rng(9)
    coords = np.stack([40.0 + rng.random(N_ITEM) * 0.36, -74.0 + rng.random(N_ITEM) * 0.47], 1)
    model = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                         n_item=N_ITEM, n_dists=[N_DIST, 0.2], n_in=DIM, n_hidden=DIM, seed=5, table_dtype="f16", coords=coords)
    assert model.lt.t.dtype == torch.float16 and model.lt.t.numel() > (1 << 31)
    return poi_amd, T, model


def _state(m, rows=None):
    out = {}
    for k in SP_NAMES:
        if k == "lt":
            continue
        v = getattr(m, k).get_value()
        out[k] = float(v) if k == "wd" else np.asarray(v, np.float64)
    return out


def test_configx_launch_on_the_10m_row_half_table(big):
    import torch
    from oracle import c_oracle as C
    from poi_amd.data import padded_to_csr
    from tests.gpu_util import assert_close, assert_step_close
    pa, T, m = big
    P, Q, M = (np.asarray(T["train"][i]) for i in (0, 2, 1))
    lens = T["lens"]
    users = np.arange(T["n_user"], dtype=np.int32)
    touched = np.unique(np.concatenate([P[M > 0], Q[M > 0], [N_ITEM]]))
    trows = torch.as_tensor(touched.astype(np.int64)).cuda()
    lt0 = m.lt.t.clone()
    before = _state(m)
    before["lt"] = m.lt.t[trows].float().cpu().numpy().astype(np.float64)
    cap = 4.0
    m.ctx.set_batch_cap(cap)
    try:
        out = np.asarray(m.train_batch(users))
    finally:
        m.ctx.set_batch_cap(1.0)
    assert np.isfinite(out).all()
    # (1) untouched rows bit-identical, over the whole table
    changed = (m.lt.t != lt0).any(dim=1)
    is_t = torch.zeros(N_ITEM + 1, dtype=torch.bool, device="cuda"); is_t[trows] = True
    assert not bool((changed & ~is_t).any()), "a row no sequence of the launch touches changed"
    assert int(changed.sum()) > 0.9 * len(touched)          # (decay-only rows may stay identical under round-to-nearest)
    for big_id in (N_ITEM - 5, 9_000_001, 8_388_609):
        assert bool(changed[big_id]), "planted row %d did not move" % big_id
    del lt0
    # (2) the touched rows against the float64 oracle on the compact table
    remap = {int(r): i for i, r in enumerate(touched)}       # the padding row N_ITEM is the largest id -> last compact row, as the oracle expects
    n_c = len(touched) - 1
    rm = np.vectorize(remap.get)
    Pc, Qc = rm(P), rm(Q)
    off, p = padded_to_csr(Pc, lens); _, q = padded_to_csr(Qc, lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    Pin = dict(before); Pin["h0"] = np.zeros(DIM)
    exp, eout, tch = C.spatial_batch_mean(Pin, off, p, q, dp, dq, users, T["len_max"], 0.01, 0.001, cap=cap)
    assert exp["lt"].shape == (n_c + 1, DIM)
    assert_close(out[:, :3], eout[:, :3], "losses", rtol=6e-5)
    got = _state(m)
    assert_step_close(got, exp, before, [k for k in SP_NAMES if k != "lt"], "config X launch, dense tensors + di", rtol=6e-5, delta_rtol=3e-4)
    lt = m.lt.t[trows].float().cpu().numpy().astype(np.float64)
    want = _f16(exp["lt"])
    ulp = np.spacing(np.abs(want).astype(np.float16)).astype(np.float64)
    noise = 6e-5 * np.abs(want).max()
    assert (np.abs(lt - want) <= ulp + noise).all(), "a touched row differs from half(oracle) by more than one half ulp"
    assert (lt == want).mean() > 0.97


def test_configx_stochastic_rounding_keeps_the_decay(big):
    """A row that only receives the L2 decay (alpha lambda |x| = 1e-5 |x| per step - 1/24 .. 1/49 of half an fp16 ulp): bit-identical
    under round-to-nearest, moves under stochastic rounding, every element lands on a half neighbour of the float32 result and the mean
    step is the decay (unbiased to the sampling error of 256 x 200 elements)."""
    import torch
    pa, T, m = big
    P, Q, M = (np.asarray(T["train"][i]) for i in (0, 2, 1))
    # q[0] of a sequence never enters the loss but is decayed: pick users whose q[0] appears nowhere else
    cnt = np.bincount(np.concatenate([P[M > 0], Q[M > 0]]), minlength=1)
    users = [u for u in range(T["n_user"]) if cnt[Q[u, 0]] == 1][:40]
    assert len(users) >= 20
    rows = torch.as_tensor(Q[users, 0].astype(np.int64)).cuda()
    ids = np.array(users, np.int32)
    x0 = m.lt.t[rows].clone()
    m.train_batch(ids)
    assert torch.equal(m.lt.t[rows], x0), "round to nearest was expected to lose a decay-only update"
    m.ctx.set_f16_rounding("stochastic", seed=17)
    try:
        steps = 200
        x0 = m.lt.t[rows].float()
        acc = torch.zeros_like(x0)
        prev = x0.clone()
        for _ in range(steps):
            m.train_batch(ids)
            cur = m.lt.t[rows].float()
            d = cur - prev
            ulp = torch.as_tensor(np.spacing(np.abs(prev.cpu().numpy()).astype(np.float16)).astype(np.float32)).cuda()
            assert bool(((d == 0) | (d.abs() <= ulp * 1.0001)).all()), "an element moved by more than one half ulp in one step"
            prev = cur
        moved = (prev - x0)
        assert bool((moved != 0).any())
        # expected: x (1 - alpha lambda)^steps - x ~ -steps alpha lambda x (each of these users contributes one decay per launch)
        expect = x0 * ((1.0 - 1e-5) ** steps - 1.0)
        rel = float((moved.sum() - expect.sum()).abs() / expect.abs().sum())
        sel = x0.abs() > 0.05
        slope = float((moved[sel] / x0[sel]).mean())
        assert abs(slope / (((1.0 - 1e-5) ** steps) - 1.0) - 1.0) < 0.15, "stochastic rounding is biased: mean relative step %.3e" % slope
        assert rel < 0.5
    finally:
        m.ctx.set_f16_rounding("nearest")


def test_configx_geo_topk_over_10m_pois(big):
    """poi_score_topk_geo over all 10 M POIs (half snapshot, dim 256, bins computed on the fly inside the scoring kernel, no U x N matrix)
    for 64 users against EXPLICIT scores: float64 users . items^T from torch + wd * the dense probability rows of poi_dist_prob (the
    reference-shaped path, itself bit-exact against the reference's bins: test_dist_prob_matches_reference_bins), top-21 by torch.topk,
    compared on the rows whose 21 best scores are separated by more than the float32 noise."""
    import torch
    pa, T, m = big
    ids = np.arange(64, dtype=np.int32)
    m.update_trained_items(); m.update_trained_dists()
    hts, sts = m.predict_device(np.arange(T["n_user"], dtype=np.int32))
    m.update_trained_users(hts); m.update_trained_sus(sts)
    idx, sc = m.compute_sub_topk(ids, 20, return_scores=True)
    wd, prob = pa.models.OboSpatialGru._prob_rows(m, torch.as_tensor(ids).cuda(), 0)
    full = hts[:64].double() @ m.trained_items.t[:N_ITEM].double().T
    full += wd[0].double() * prob.double()
    top = torch.topk(full, 21, dim=1)
    gap = (top.values[:, :-1] - top.values[:, 1:]).min(dim=1).values
    ok = gap > 2e-5 * top.values.abs().max()
    assert int(ok.sum()) >= 32
    assert torch.equal(idx[ok].long(), top.indices[ok][:, :20])
    assert torch.allclose(sc[ok].double(), top.values[ok][:, :20], rtol=1e-4, atol=1e-4)
