"""-m gpu parity of the mini-batch `Gru` (public/GRU.py:395-498): `Gru.train(idxs)` - one SGD step on the batch cost -
against the float64 oracle restatement (oracle.gru_minibatch_step, itself checked against autograd of the batched scan in
tests/test_oracle_autograd.py), through the per-sequence engine (any dim) and the tile engine (dim 64 / 128 / 256)."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, gru_params, toy_problem

pytestmark = pytest.mark.gpu

GRU_NAMES = ("lt", "ui", "wh", "bi")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    poi_amd._lib.load()
    yield poi_amd
    poi_amd._lib.context(0).set_engine("auto")
    poi_amd._lib.context(0).set_batch_cap(1.0)


def _model(pa, T, P, **kw):
    return pa.models.Gru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                         n_item=T["n_item"], n_in=T["dim"], n_hidden=T["dim"], init=P, **kw)


@pytest.mark.parametrize("dim,engine,n_user,batch", [(20, "seq", 23, 7), (32, "seq", 40, 40), (20, "tile", 23, 7), (100, "tile", 40, 16), (64, "tile", 50, 16), (64, "seq", 50, 16),
                                                     (128, "tile", 90, 64), (256, "tile", 40, 24)])
def test_minibatch_gru_steps_match_oracle(pa, dim, engine, n_user, batch):
    T = toy_problem(300 + dim, n_user=n_user, n_item=70, dim=dim, len_max=11, hot=16)
    P = gru_params(300 + dim, T)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    model = _model(pa, T, P, pad_dim=(engine != "seq"))      # seq: the per-sequence engine at the model's native dim
    model.ctx.set_engine(engine)
    model.ctx.set_batch_cap(4.0)                      # the class switches to the mini-batch rule for its own launches and restores this
    order = np.random.default_rng(2).permutation(n_user).astype(np.int32)
    for b0 in range(0, n_user, batch):                # the last batch is ragged
        idxs = order[b0:b0 + batch]
        old = P
        P, loss = O.gru_minibatch_step(P, Pm[idxs], Qm[idxs], Mm[idxs], 0.01, 0.001)
        got_loss = model.train(idxs)
        assert_close(got_loss, loss, "batch loss", rtol=2e-5)
        got = {k: getattr(model, k).get_value() for k in GRU_NAMES}
        assert_step_close(got, P, old, GRU_NAMES, "%s batch at %d" % (engine, b0))
        P = {k: (np.asarray(got[k], np.float64) if k in got else v) for k, v in P.items()}      # continue from the device state
    assert model.ctx.batch_cap == 4.0


def test_minibatch_gru_of_one_user_is_the_one_by_one_step(pa):
    T = toy_problem(77, n_user=6, n_item=40, dim=20, len_max=9)
    P = gru_params(77, T)
    a, b = _model(pa, T, P), pa.models.OboGru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                               n_item=T["n_item"], n_in=20, n_hidden=20, init=P)
    for u in (4, 1, 4):
        la, lb = a.train([u]), b.train(u)
        assert la == lb
    for k in GRU_NAMES:
        assert np.array_equal(getattr(a, k).get_value(), getattr(b, k).get_value()), k


def test_minibatch_rule_is_rejected_where_it_does_not_apply(pa):
    ctx = pa._lib.context(0)
    T = toy_problem(5, n_user=4, n_item=30, dim=8, len_max=6)
    rng = np.random.default_rng(0)
    m = pa.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=4, n_item=30, n_in=8, n_hidden=8)
    ctx.set_batch_cap(0.0)
    try:
        with pytest.raises(pa.PoiError):
            m.train_batch(np.array([0, 1], np.int32), np.array([1, 2], np.int32), np.array([3, 4], np.int32))
    finally:
        ctx.set_batch_cap(1.0)
