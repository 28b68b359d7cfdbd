"""-m gpu: dim 256 with the REFERENCE'S OWN INIT (uniform(-0.5, 0.5): public/GRU_Spatial.py:50-71, public/GRU.py:60-62) - the exact forward pass of
config X (te_gemmx<256>: int8 digit products over K = 256; te_rec_fwdd: the recurrence in float64 on the matrix cores) against the float64 oracle.
At this dim the recurrence expands perturbations ~10^6-fold over 50 positions: a float32 forward pass is O(1) off in the updates (tools/x256_check.py),
the updates themselves are 10^2 .. 10^3 on weights of 0.5 - the 1e-5 bar on the weights is a 1e-5 bar on the gradients.
  * a 192-user launch (per-step pre-activation rows) and a 1600-user launch (forward table over the launch's POIs: the FT variant of the kernel);
  * predict: final hidden states and distance softmax;
  * a NaN weight propagates to the losses and the updated tensors (the float64 gates clamp their argument: ADVICE r4)."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, spatial_params, toy_problem

pytestmark = pytest.mark.gpu
SP = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    return poi_amd


def _model(pa, T, P, dim):
    return pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                   n_dists=[T["n_dist"], 0.2], n_in=dim, n_hidden=dim, init=P)


@pytest.mark.parametrize("n_user,n_item", [(192, 4000), (1600, 700)])
def test_dim256_launch_with_the_reference_init(pa, n_user, n_item):
    from oracle import c_oracle as C
    from poi_amd.data import padded_to_csr
    dim, cap = 256, 64.0
    T = toy_problem(900 + n_user, n_user=n_user, n_item=n_item, n_dist=200, dim=dim, len_max=50, min_len=4, hot=n_item // 2)
    P0 = spatial_params(901, T)
    lens = np.asarray(T["lens"])
    users = np.argsort(-lens, kind="stable").astype(np.int32)
    off, p = padded_to_csr(np.asarray(T["train"][0]), lens); _, q = padded_to_csr(np.asarray(T["train"][2]), lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    Pin = {k: (float(P0[k]) if k == "wd" else np.asarray(P0[k], np.float64)) for k in SP}; Pin["h0"] = np.zeros(dim)
    exp, eout, _ = C.spatial_batch_mean(Pin, off, p, q, dp, dq, users, T["len_max"], 0.01, 0.001, cap=cap, threads=8)
    m = _model(pa, T, P0, dim)
    m.ctx.set_batch_cap(cap)
    try:
        out = np.asarray(m.train_batch(users))
    finally:
        m.ctx.set_batch_cap(1.0)
    assert np.abs(np.asarray(exp["wh"]) - Pin["wh"]).max() > 1.0, "the reference init at dim 256 should produce exploding updates (else this test is not testing the regime)"
    assert_close(out[:, :3], eout[:, :3], "losses")
    got = {k: (float(getattr(m, k).get_value()) if k == "wd" else np.asarray(getattr(m, k).get_value(), np.float64)) for k in SP}
    assert_step_close(got, exp, Pin, SP, "dim 256, reference init, %d users" % n_user)


@pytest.mark.parametrize("n_user,n_item", [(40, 900), (1700, 500)])      # (3 tiles; 107 tiles, every POI row read by many sequences)
def test_dim256_predict_with_the_reference_init(pa, n_user, n_item):
    dim = 256
    T = toy_problem(77, n_user=n_user, n_item=n_item, n_dist=200, dim=dim, len_max=30 if n_user > 100 else 50, min_len=4)
    P = spatial_params(78, T)
    m = _model(pa, T, P, dim)
    m.update_trained_items(); m.update_trained_dists()
    ids = np.arange(n_user, dtype=np.int32)
    hts, sts = m.predict(ids)
    eh, es = O.spatial_predict(P, P["lt"], P["di"], np.asarray(T["train"][0])[ids], np.asarray(T["dist"][0])[ids], np.asarray(T["train"][1])[ids])
    assert_close(hts, eh, "hts"); assert_close(sts, es, "sts")


@pytest.mark.parametrize("where", ["wh", "ui", "lt"])
@pytest.mark.parametrize("dim,n_user", [(64, 1), (64, 40), (128, 600), (128, 1700), (256, 40)])
def test_nan_weight_propagates_through_the_exact_forward(pa, dim, n_user, where):
    """the float64 reference propagates a NaN weight into every value that depends on it; so must the exact forward pass of te_xfwd.hip - one-sequence
    path, per-sequence kernel, 16-sequence tiles on int8 digits (a NaN has no digits: the row's scale carries it), forward table, float64-MFMA tiles"""
    T = toy_problem(5, n_user=max(n_user, 2), n_item=300, n_dist=50, dim=dim, len_max=12)
    P = spatial_params(6, T)
    P[where] = np.array(P[where])
    if where == "lt":
        P["lt"][int(np.asarray(T["train"][0])[0][0]), 7] = np.nan      # the first check-in of user 0
    else:
        P[where][1, 3, 5] = np.nan
    m = _model(pa, T, P, dim)
    out = np.asarray(m.train_batch(np.arange(n_user, dtype=np.int32)))
    if where == "lt":
        assert np.isnan(out[0, 0]), "the loss of a sequence whose first input row holds a NaN must be NaN"
    else:
        assert np.isnan(out[:, 0]).all(), "losses of sequences under a NaN weight must be NaN"
    assert np.isnan(np.asarray(m.wh.get_value())).any() and np.isnan(np.asarray(m.ui.get_value())).any()


def test_nan_flag_survives_graph_replay_with_interleaved_launch_sizes(pa):
    """a captured launch replays the launch id it was captured with.  Two launch sizes are captured while the weights are finite; then a
    NaN weight must poison the replays of BOTH graphs (the older graph's id is below whatever the newer launch left in the flag), and restoring
    the weights must give finite losses again on the same context."""
    dim = 128
    T = toy_problem(15, n_user=96, n_item=300, n_dist=50, dim=dim, len_max=12)
    P = spatial_params(16, T)
    m = _model(pa, T, P, dim)
    m.ctx.set_engine("tile")
    m.ctx.set_graph(True)
    try:
        a, b = np.arange(0, 40, dtype=np.int32), np.arange(40, 96, dtype=np.int32)
        r0 = m.ctx.graph_replays()
        for ids in (a, b):                                     # first sight eager, second sight captures, third replays - one size after the other
            for _ in range(3):                                 # (a launch is captured at its second consecutive sight)
                assert np.isfinite(np.asarray(m.train_batch(ids))).all()
        assert m.ctx.graph_replays() - r0 >= 4
        good = {k: getattr(m, k).t.clone() for k in ("wh", "ui", "lt", "di", "bi", "vs", "bs", "wd", "loss_weight")}
        m.wh.t[1, 3, 5] = float("nan")
        oa, ob = np.asarray(m.train_batch(a)), np.asarray(m.train_batch(b))          # replays of the OLDER and of the newer graph
        assert np.isnan(oa[:, 0]).all() and np.isnan(ob[:, 0]).all(), "a NaN weight must poison the replayed launches"
        for k, v in good.items():
            getattr(m, k).t.copy_(v)
        oa, ob = np.asarray(m.train_batch(a)), np.asarray(m.train_batch(b))
        assert np.isfinite(oa).all() and np.isfinite(ob).all(), "restored weights: the flag must not stick to a graph"
    finally:
        m.ctx.set_graph(False)
        m.ctx.set_engine("auto")
