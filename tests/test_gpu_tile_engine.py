"""-m gpu parity tests of the tile engine (batched f32-MFMA decomposition of the Distance2Pre step):
forced through poi_ctx_set_engine(tile) and compared with the float64 oracle (single sequence ==
the reference step; batches == mean-of-touching-sequences rule) and with the per-sequence engine."""
import numpy as np
import pytest

from oracle import poi_oracle as O
from tests.gpu_util import assert_close, assert_step_close, batch_mean_update, round_f32, spatial_params, toy_problem

pytestmark = pytest.mark.gpu

SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    poi_amd._lib.load()
    yield poi_amd
    poi_amd._lib.context(0).set_engine("auto")


@pytest.fixture(autouse=True, params=[0, 1280], ids=["regrouped", "default-threshold"])
def _regroup_min(request, pa):
    """Every test of this module runs twice: with the regrouped backward pass (per-bin tables, per-POI regrouping, forward table) forced
    for every launch size (poi_ctx_set_regroup_min(0) - what the 12500-user launches of the bench run), and with the product's default
    threshold, under which these small launches take the two-table path."""
    ctx = pa._lib.context(0)
    ctx.set_regroup_min(request.param)
    # (the exact forward pass likewise: 16-sequence int8 tiles for every launch size under "regrouped", the per-sequence float64 kernel for
    # these small launches under the default)
    ctx.set_exact_forward(True, 0 if request.param == 0 else 512)
    yield
    ctx.set_regroup_min(1280); ctx.set_exact_forward(True, 512)


def _model(pa, T, P):
    return pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001],
                                   n_user=T["n_user"], n_item=T["n_item"], n_dists=[T["n_dist"], 0.2],
                                   n_in=T["dim"], n_hidden=T["dim"], init=P)


def _get(model):
    out = {}
    for k in SP_NAMES:
        v = getattr(model, k).get_value()
        out[k] = float(v) if k == "wd" else v
    return out


def _oracle_batch(P, T, users):
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    news, touched, outs = [], [], []
    for u in users:
        Pn, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); outs.append(out)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), di=np.unique(DPm[u])))
    exp = batch_mean_update(P, news, touched, ("lt", "di"), ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"))
    return exp, outs


@pytest.mark.parametrize("dim,n_dist,engine", [(64, 11, "tile"), (64, 200, "tile"), (128, 40, "tile"), (128, 200, "tile"), (128, 200, "tile32"),
                                               (256, 40, "tile"), (256, 200, "tile"),
                                               (64, 1520, "tile"), (128, 1520, "tile"), (128, 300, "tile"), (256, 1520, "tile"),      # > 256 bins: chunked head
                                               (64, 256, "tile"), (64, 511, "tile"), (64, 512, "tile"), (64, 2046, "tile")])                # chunk boundaries, the maximum
def test_tile_single_sequence_is_the_reference_step(pa, dim, n_dist, engine):
    """(tile32 = the streaming recurrent kernels of dim 256 - 32-sequence tiles, weights streamed from L2 - at dim 128)"""
    T = toy_problem(60 + dim + n_dist, n_user=4, n_item=90, n_dist=n_dist, dim=dim, len_max=9)
    P = spatial_params(60 + dim, T)
    model = _model(pa, T, P)
    model.ctx.set_engine(engine)
    model.ctx.timing(True)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    for u in [2, 0, 2]:
        old = P
        P, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
        los, sur, upq, ls = model.train(np.int32(u))
        assert_close([los, sur, upq], out[:3], "losses")
        got = _get(model)
        assert_step_close(got, P, old, SP_NAMES, "after user %d" % u)
        P = round_f32({**P, **got})
    assert model.ctx.timing_get("te_head")[1] == 3, "the launches did not go through the tile engine"
    model.ctx.timing(False)
    model.ctx.set_engine("auto")


@pytest.mark.parametrize("dim,n_dist,n_user,tile_eng", [(64, 11, 45, "tile"), (64, 200, 70, "tile"), (128, 200, 37, "tile"), (128, 200, 70, "tile32"),
                                                        (64, 1520, 45, "tile"), (128, 1520, 70, "tile"),
                                                       (256, 200, 70, "tile")])
def test_tile_batch_matches_mean_rule_and_seq_engine(pa, dim, n_dist, n_user, tile_eng):
    T = toy_problem(70 + dim, n_user=n_user, n_item=120, n_dist=n_dist, dim=dim, len_max=11, hot=30)
    P = spatial_params(70 + dim, T)
    users = np.random.default_rng(0).permutation(n_user)[: n_user - 3].astype(np.int32)   # unsorted lengths, ragged last tile
    exp, outs = _oracle_batch(P, T, users)
    res = {}
    for eng in (tile_eng, "seq"):
        model = _model(pa, T, P)
        model.ctx.set_engine(eng)
        got_out = model.train_batch(users)
        for k, out in enumerate(outs):
            assert_close(got_out[k][:3], out[:3], "%s losses[%d]" % (eng, k), rtol=2e-5)
        got = _get(model)
        assert_step_close(got, exp, P, SP_NAMES, eng)
        res[eng] = got
        # a second launch on the updated state exercises the re-zeroed gradient tables / slabs
        model.train_batch(users[:40])
        res[eng + "2"] = _get(model)
    for k in SP_NAMES:     # (two launches compound the float32 noise; dim 256: see test_tile_predict_matches_oracle)
        assert_close(res[tile_eng + "2"][k], res["seq2"][k], "tile vs seq second launch " + k, rtol=6e-5 if dim >= 256 else 2e-5)
    pa._lib.context(0).set_engine("auto")


@pytest.mark.parametrize("dim,n_seq", [(64, 1), (128, 7), (256, 3)])
def test_tile_touched_row_list_for_tables_much_larger_than_the_launch(pa, dim, n_seq):
    """A POI table far larger than the launch's footprint (config X: 10 M rows against < 1 M touches): te_reduce walks the
    launch's touched-row list instead of scanning the table.  Same oracle bars; rows never touched stay bit-identical; the
    padding rows (touched only analytically: no segment) still get their L2 decay."""
    T = toy_problem(210 + dim, n_user=14, n_item=6000, n_dist=40, dim=dim, len_max=9, hot=3000)
    P = spatial_params(210 + dim, T)
    users = np.nonzero(T["lens"] < T["len_max"])[0][:n_seq].astype(np.int32)        # shorter than len_max: the padding rows are touched, analytically only
    assert len(users) == n_seq
    exp, outs = _oracle_batch(P, T, users)
    model = _model(pa, T, P)
    model.ctx.set_engine("tile")
    got_out = model.train_batch(users)
    for k, out in enumerate(outs):
        assert_close(got_out[k][:3], out[:3], "losses[%d]" % k, rtol=2e-5)
    got = _get(model)
    assert_step_close(got, exp, P, SP_NAMES, "touched-row list")
    same = (got["lt"] == np.asarray(P["lt"], np.float32)).all(axis=1)
    touched = np.zeros(T["n_item"] + 1, bool)
    for u in users:
        touched[T["train"][0][u]] = True; touched[T["train"][2][u]] = True
    touched[T["n_item"]] = True
    assert np.array_equal(~same, touched), "rows changed != rows touched (incl. the analytically touched padding row)"
    # second launch: segment bookkeeping was re-zeroed through the list as well
    model.train_batch(users)
    model.ctx.set_engine("auto")


@pytest.mark.parametrize("cap", [4.0, 1e9])
def test_batch_cap_generalises_the_mean_rule(pa, cap):
    """poi_ctx_set_batch_cap: a row touched by k sequences moves by min(k, cap) / k times the SUM of their reference
    updates (cap = 1: the mean; huge cap: the plain sum), dense tensors with k = n_seq - both engines, against the
    float64 oracle of the same rule."""
    from oracle import c_oracle as C
    from poi_amd.data import padded_to_csr
    T = toy_problem(170, n_user=120, n_item=150, n_dist=23, dim=64, len_max=11, hot=20)
    P = spatial_params(170, T)
    lens = T["lens"]
    off, p = padded_to_csr(T["train"][0], lens); _, q = padded_to_csr(T["train"][2], lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    users = np.random.default_rng(2).permutation(120)[:100].astype(np.int32)
    exp, eout, _ = C.spatial_batch_mean(P, off, p, q, dp, dq, users, T["len_max"], 0.001, 0.001, cap=cap)
    ctx = pa._lib.context(0)
    try:
        ctx.set_batch_cap(cap)
        for eng in ("tile", "seq"):
            model = pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.001, 0.001], n_user=T["n_user"],
                                            n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=T["dim"], n_hidden=T["dim"], init=P)
            ctx.set_engine(eng)
            out = model.train_batch(users)
            assert_close(np.asarray(out)[:, :3], eout[:, :3], eng + " losses", rtol=2e-5)
            assert_step_close(_get(model), exp, P, SP_NAMES, "%s cap %g" % (eng, cap))
            # a single sequence is the reference step whatever the cap
            u = int(users[0])
            Pn, _ = O.spatial_step(P, T["train"][0][u], T["train"][2][u], T["dist"][0][u], T["dist"][2][u], T["train"][1][u], 0.001, 0.001)
            m1 = pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.001, 0.001], n_user=T["n_user"],
                                         n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=T["dim"], n_hidden=T["dim"], init=P)
            m1.train(np.int32(u))
            assert_step_close(_get(m1), Pn, P, SP_NAMES, "%s single sequence under cap %g" % (eng, cap))
    finally:
        ctx.set_batch_cap(1.0); ctx.set_engine("auto")


@pytest.mark.parametrize("dim,n_dist,engine", [(64, 23, "tile"), (128, 200, "tile"), (128, 200, "tile32"), (256, 200, "tile"), (64, 1520, "tile"), (128, 1520, "tile")])
def test_tile_predict_matches_oracle(pa, dim, n_dist, engine):
    T = toy_problem(80 + dim, n_user=75, n_item=200, n_dist=n_dist, dim=dim, len_max=13)
    P = spatial_params(80 + dim, T)
    model = _model(pa, T, P)
    model.ctx.set_engine(engine)
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(2, 73, dtype=np.int32)
    hts, sts = model.predict(ids)
    eh, es = O.spatial_predict(P, P["lt"], P["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    # dim 256 with the reference's uniform(-0.5, 0.5) init saturates the gates (pre-activations are sums of 256 terms): 13 float32
    # steps land 3.6e-5 from float64 in the tile engine and 1.7e-5 in the per-sequence engine (tools/_diag_pred.py; unchanged with
    # libm-exact activations) - conditioning, as at the full BASELINE shapes (tests/gpu_util.FULL_SIZE_LT)
    rt = 6e-5 if dim >= 256 else 1e-5
    assert_close(hts, eh, "hts", rtol=rt); assert_close(sts, es, "sts", rtol=rt)
    model.ctx.set_engine("auto")


def test_tile_sorted_scatter_hot_rows_are_exact_and_reproducible(pa):
    """Few POIs / distance bins and many sequences: every table row collects hundreds of touches, so
    the write-back goes through the chunked ("hot row") path of te_scatter.hip.  The result must match
    the batch rule and - no float atomics, fixed summation order - be bitwise reproducible."""
    T = toy_problem(91, n_user=300, n_item=26, n_dist=11, dim=64, len_max=12)
    P = spatial_params(91, T)
    users = np.random.default_rng(3).permutation(300).astype(np.int32)
    exp, _ = _oracle_batch(P, T, users)
    runs = []
    for _ in range(2):
        model = _model(pa, T, P)
        model.ctx.set_engine("tile")
        model.train_batch(users)
        runs.append(_get(model))
    assert_step_close(runs[0], exp, P, SP_NAMES, "hot rows")
    for k in SP_NAMES:                      # no float atomics anywhere in the tile engine: every tensor is reproducible
        assert np.array_equal(runs[0][k], runs[1][k]), k + " differs between two identical launches"
    pa._lib.context(0).set_engine("auto")


GRU_NAMES = ("lt", "ui", "wh", "bi")


def _gru_model(pa, T, P):
    return pa.models.OboGru(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                            n_item=T["n_item"], n_in=T["dim"], n_hidden=T["dim"], init=P)


def _get_gru(model):
    return {k: getattr(model, k).get_value() for k in GRU_NAMES}


@pytest.mark.parametrize("dim", [64, 128, 256])
def test_tile_plain_gru_single_sequence_is_the_reference_step(pa, dim):
    """OboGru.seq_train (public/GRU.py:313-389) through the tile engine: one sequence per launch ==
    the reference step, several users in a row (state carried on the device)."""
    from tests.gpu_util import gru_params
    T = toy_problem(110 + dim, n_user=5, n_item=80, dim=dim, len_max=9)
    P = gru_params(110 + dim, T)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    P0 = P
    try:
        for one in (True, False):                 # the one-sequence path (dims 64 / 128) and the batched pipeline at one sequence per launch
            model = _gru_model(pa, T, P0)
            model.ctx.set_engine("tile"); model.ctx.set_one_sequence_path(one)
            model.ctx.timing(True)
            P = P0
            for u in [3, 0, 3, 1]:
                old = P
                P, loss = O.gru_step(P, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
                got_loss = model.train(np.int32(u))
                assert_close(got_loss, loss, "loss")
                got = _get_gru(model)
                assert_step_close(got, P, old, GRU_NAMES, "after user %d (one-sequence path %s)" % (u, one))
                P = round_f32({**P, **got})
            assert (model.ctx.timing_get("te_wgrad")[1] == 0) == (one and dim <= 128), "the launches did not take the expected path"
            model.ctx.timing(False)
    finally:
        pa._lib.context(0).set_one_sequence_path(True); pa._lib.context(0).set_engine("auto")


@pytest.mark.parametrize("dim,n_user", [(64, 77), (128, 150), (256, 77)])
def test_tile_plain_gru_batch_matches_mean_rule_and_seq_engine(pa, dim, n_user):
    from tests.gpu_util import gru_params
    T = toy_problem(120 + dim, n_user=n_user, n_item=60, dim=dim, len_max=12, hot=20)
    P = gru_params(120 + dim, T)
    users = np.random.default_rng(1).permutation(n_user)[: n_user - 2].astype(np.int32)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    news, touched, losses = [], [], []
    for u in users:
        Pn, loss = O.gru_step(P, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); losses.append(loss)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u])))))
    exp = batch_mean_update(P, news, touched, ("lt",), ("ui", "wh", "bi"))
    res = {}
    for eng in ("tile", "seq"):
        model = _gru_model(pa, T, P)
        model.ctx.set_engine(eng)
        got_loss = model.train_batch(users)
        assert_close(np.asarray(got_loss).reshape(-1), np.asarray(losses), eng + " losses", rtol=2e-5)
        got = _get_gru(model)
        assert_step_close(got, exp, P, GRU_NAMES, eng)
        model.train_batch(users[:70])
        res[eng] = _get_gru(model)
    for k in GRU_NAMES:
        assert_close(res["tile"][k], res["seq"][k], "tile vs seq second launch " + k, rtol=2e-5)
    pa._lib.context(0).set_engine("auto")


def test_tile_degenerate_lengths(pa):
    """Launch mixing sequences of length 1 (no GRU step at all: only the L2 decay of their rows), 2 (a single
    step) and longer ones, spatial and plain: == batch rule of the oracle, == per-sequence engine."""
    from tests.gpu_util import gru_params
    T = toy_problem(131, n_user=90, n_item=70, n_dist=11, dim=64, len_max=9, min_len=1)
    assert (T["lens"] == 1).any() and (T["lens"] == 2).any()
    users = np.arange(90, dtype=np.int32)
    P = spatial_params(131, T)
    exp, outs = _oracle_batch(P, T, users)
    for eng in ("tile", "seq"):
        model = _model(pa, T, P)
        model.ctx.set_engine(eng)
        got_out = model.train_batch(users)
        for k, out in enumerate(outs):
            assert_close(got_out[k][:3], out[:3], "%s losses[%d]" % (eng, k), rtol=2e-5)
        got = _get(model)
        assert_step_close(got, exp, P, SP_NAMES, eng)
    Pg = gru_params(131, T)
    Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
    news, touched, losses = [], [], []
    for u in users:
        Pn, loss = O.gru_step(Pg, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
        news.append(Pn); losses.append(loss)
        touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u])))))
    expg = batch_mean_update(Pg, news, touched, ("lt",), ("ui", "wh", "bi"))
    for eng in ("tile", "seq"):
        model = _gru_model(pa, T, Pg)
        model.ctx.set_engine(eng)
        got_loss = model.train_batch(users)
        assert_close(np.asarray(got_loss).reshape(-1), np.asarray(losses), eng + " gru losses", rtol=2e-5)
        got = _get_gru(model)
        assert_step_close(got, expg, Pg, GRU_NAMES, eng + " gru")
    pa._lib.context(0).set_engine("auto")


def test_tile_multi_tile_workgroups_match_seq_engine(pa):
    """~50 k packed rows: every persistent GEMM workgroup walks several 128-row tiles, i.e. the cross-tile
    operand pipeline of te_gemm_ntk (next tile's chunks and gather indices fetched inside the current tile)
    and the multi-iteration paths of te_wgrad / te_head run for real.  The toy sizes above give each workgroup
    at most one tile.  Reference: the per-sequence engine (itself pinned to the oracle above)."""
    T = toy_problem(901, n_user=2600, n_item=3000, n_dist=200, dim=128, len_max=41, hot=300)
    P = spatial_params(901, T)
    users = np.random.default_rng(3).permutation(2600)[:2500].astype(np.int32)
    res, outs = {}, {}
    for eng in ("tile", "seq"):
        model = _model(pa, T, P)
        model.ctx.set_engine(eng)
        outs[eng] = np.asarray(model.train_batch(users))
        res[eng] = _get(model)
    pa._lib.context(0).set_engine("auto")
    assert_close(outs["tile"][:, :3], outs["seq"][:, :3], "losses", rtol=2e-5)
    for k in SP_NAMES:
        assert_close(res["tile"][k], res["seq"][k], "tile vs seq " + k, rtol=2e-5)
    # ... and both against the float64 oracle of the batch rule (plain-C restatement, threaded over the launch)
    from oracle import c_oracle as C
    from poi_amd.data import padded_to_csr
    lens = T["lens"]
    off, p = padded_to_csr(T["train"][0], lens); _, q = padded_to_csr(T["train"][2], lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    exp, eout, _ = C.spatial_batch_mean(P, off, p, q, dp, dq, users, T["len_max"], 0.01, 0.001)
    for eng in ("tile", "seq"):
        assert_close(outs[eng][:, :3], eout[:, :3], eng + " losses vs oracle", rtol=2e-5)
        assert_step_close(res[eng], exp, P, SP_NAMES, eng + " vs oracle")


@pytest.mark.parametrize("dim", [64, 128])
def test_tile_plain_gru_multi_tile_workgroups_match_seq_engine(pa, dim):
    """Plain GRU at ~50 k packed rows: the one-table variant of te_gemm_ntk (gather indices double-buffered by
    tile parity, D = 128) and the runtime-K fallback (D = 64) with several tiles per workgroup."""
    from tests.gpu_util import gru_params
    T = toy_problem(930 + dim, n_user=2600, n_item=3000, dim=dim, len_max=41, hot=300)
    P = gru_params(930 + dim, T)
    users = np.random.default_rng(4).permutation(2600)[:2500].astype(np.int32)
    res, losses = {}, {}
    for eng in ("tile", "seq"):
        model = _gru_model(pa, T, P)
        model.ctx.set_engine(eng)
        losses[eng] = np.asarray(model.train_batch(users)).reshape(-1)
        res[eng] = _get_gru(model)
    pa._lib.context(0).set_engine("auto")
    assert_close(losses["tile"], losses["seq"], "losses", rtol=2e-5)
    for k in GRU_NAMES:
        assert_close(res["tile"][k], res["seq"][k], "tile vs seq " + k, rtol=2e-5)


def _f16(a):
    return np.asarray(a, np.float16).astype(np.float64)


@pytest.mark.parametrize("dim,n_user", [(64, 60), (128, 60), (128, 130), (256, 40)])      # (128, 130): the forward table is active
def test_fp16_poi_table_float32_math(pa, dim, n_user):
    """Config X's "fp16 embeddings": lt stored as IEEE half, arithmetic float32.  Oracle: the float64 batch rule run from the
    HALF-rounded table; expectation for lt = that result rounded to half (the device rounds once, at the write-back) - allowed to
    differ by one half ulp where float32 lands on the other side of a rounding boundary; the float32 tensors keep the usual bars;
    untouched rows stay bit-identical; predict + top-K read the half snapshot."""
    T = toy_problem(300 + dim, n_user=n_user, n_item=400, n_dist=40, dim=dim, len_max=10, hot=100)
    P = spatial_params(300 + dim, T)
    P["lt"] = _f16(P["lt"])
    users = np.random.default_rng(4).permutation(n_user)[: n_user - 3].astype(np.int32)
    exp, outs = _oracle_batch(P, T, users)
    model = pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                    n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=dim, n_hidden=dim, init=P, table_dtype="f16")
    import torch
    assert model.lt.t.dtype == torch.float16 and model.trained_items.t.dtype == torch.float16
    got_out = model.train_batch(users)
    for k, out in enumerate(outs):
        assert_close(got_out[k][:3], out[:3], "losses[%d]" % k, rtol=2e-5)
    got = _get(model)
    dense = [k for k in SP_NAMES if k != "lt"]
    assert_step_close(got, exp, P, dense, "fp16 table")
    lt = np.asarray(got["lt"], np.float64)
    want = _f16(exp["lt"])
    ulp = np.spacing(np.abs(want).astype(np.float16)).astype(np.float64)
    # + the float32-vs-float64 noise of the update itself (elements near zero have a tiny half ulp): the weight bar of the float32 tests
    noise = (6e-5 if dim >= 256 else 1e-5) * np.abs(want).max()
    assert (np.abs(lt - want) <= ulp + noise).all(), "lt differs from half(oracle) by more than one half ulp"
    assert (lt == want).mean() > 0.98, "more than 2 % of the half elements rounded the other way"
    touched = np.zeros(T["n_item"] + 1, bool)
    for u in users:
        touched[T["train"][0][u]] = True; touched[T["train"][2][u]] = True
    assert np.array_equal(lt[~touched], P["lt"][~touched])
    # evaluation path on the half snapshot
    model.update_trained_items(); model.update_trained_dists()
    ids = np.arange(n_user, dtype=np.int32)
    hts, sts = model.predict(ids)
    Pn = {**exp, "lt": lt, "h0": np.zeros(dim)}
    for k in dense:
        Pn[k] = np.asarray(got[k], np.float64) if k != "wd" else float(got[k])
    eh, es = O.spatial_predict(Pn, lt, Pn["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    rt = 6e-5 if dim >= 256 else 1e-5
    assert_close(hts, eh, "hts", rtol=rt); assert_close(sts, es, "sts", rtol=rt)
    model.update_trained_users(hts)
    idx = super(pa.models.OboSpatialGru, model).compute_sub_topk(ids, 10).cpu().numpy()          # no distance term: users . items
    full = np.asarray(hts, np.float64) @ lt[:-1].T
    top = O.topk_desc(full, 11)
    tv = np.take_along_axis(full, top, axis=1)
    ok = (tv[:, :-1] - tv[:, 1:]).min(axis=1) > 1e-5 * np.abs(tv).max()
    assert ok.sum() >= n_user // 2
    assert np.array_equal(idx[ok], top[ok][:, :10])
    # the per-sequence engine refuses a half table
    model.ctx.set_engine("seq")
    with pytest.raises(pa._lib.PoiError):
        model.train(np.int32(1))
    model.ctx.set_engine("auto")
    assert abs(model.l2.eval() - O.l2_value({**Pn, "loss_weight": got["loss_weight"]}, 0.001, SP_NAMES)) <= 1e-5 * model.l2.eval()


@pytest.mark.parametrize("dim,n_user,batch", [(128, 96, 32), (64, 40, 1), (256, 48, 16), (128, 2200, 2200)])      # (2200: the side-stream forks of large launches)
def test_graph_replay_is_bitwise_the_eager_launch(pa, dim, n_user, batch):
    """poi_ctx_set_graph: the captured launch replays the same kernels in the same order - parameters and per-sequence outputs are
    bitwise those of eager launches, whatever uidx / out pointers the caller hands over (they are staged)."""
    T = toy_problem(900 + dim, n_user=n_user, n_item=300, n_dist=200, dim=dim, len_max=12)
    P = spatial_params(901 + dim, T)
    rng = np.random.default_rng(5)
    order = rng.permutation(n_user).astype(np.int32)
    res = {}
    for mode in ("eager", "graph"):
        model = _model(pa, T, P)
        model.ctx.set_engine("tile")
        model.ctx.set_graph(mode == "graph")
        r0 = model.ctx.graph_replays()
        outs = []
        for rep in range(2):
            for b0 in range(0, n_user, batch):
                outs.append(np.array(model.train_batch(order[b0:b0 + batch])))
        res[mode] = (_get(model), outs, model.ctx.graph_replays() - r0)
        model.ctx.set_graph(False)
    n_launch = 2 * ((n_user + batch - 1) // batch)
    assert res["eager"][2] == 0
    assert res["graph"][2] >= n_launch - 3, res["graph"][2]          # first sight eager, second sight captures + replays
    for k in SP_NAMES:
        assert np.array_equal(np.asarray(res["eager"][0][k]), np.asarray(res["graph"][0][k])), k
    for a, b in zip(res["eager"][1], res["graph"][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("n_item", [127, 128, 129, 255, 400, 641])
def test_forward_table_any_table_size(pa, n_item):
    """Forward table (the n_item + 1 table rows times the POI half of ui - te_ptab_s3 on split products, te_gemm_ntk on float32-input
    MFMAs -, te_rec_fwd16 gathers): table sizes around the 128-row tile boundaries, training batch against the oracle's mean rule and
    predict against the oracle - a table whose last tile is partial once wrote its spare rows behind a too-small buffer.  (The 16-sequence
    tile kernels are forced: launches this small otherwise take the per-sequence kernels, which read no forward table.)"""
    T = toy_problem(500 + n_item, n_user=160, n_item=n_item, n_dist=40, dim=128, len_max=11, hot=min(60, n_item // 2))
    P = spatial_params(500 + n_item, T)
    users = np.random.default_rng(3).permutation(160)[:150].astype(np.int32)
    exp, outs = _oracle_batch(P, T, users)
    try:
        for split in (True, False):
            model = _model(pa, T, P)
            model.ctx.set_engine("tile"); model.ctx.set_small_launch(0); model.ctx.set_split_products(split)
            got_out = model.train_batch(users)
            for k, out in enumerate(outs):
                assert_close(got_out[k][:3], out[:3], "losses[%d]" % k, rtol=2e-5)
            got = _get(model)
            assert_step_close(got, exp, P, SP_NAMES, "forward table, n_item %d, split %s" % (n_item, split))
            model.update_trained_items(); model.update_trained_dists()
            ids = np.arange(160, dtype=np.int32)
            hts, sts = model.predict(ids)
            Pn = {k: (np.asarray(v, np.float64) if k != "wd" else float(v)) for k, v in got.items()}
            Pn["h0"] = np.zeros(128)
            eh, es = O.spatial_predict(Pn, Pn["lt"], Pn["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
            assert_close(hts, eh, "hts"); assert_close(sts, es, "sts")
    finally:
        pa._lib.context(0).set_engine("auto"); pa._lib.context(0).set_small_launch(1800); pa._lib.context(0).set_split_products(True)


@pytest.mark.parametrize("dim,n_dist,spatial", [(64, 11, True), (128, 255, True), (128, 300, True), (256, 40, True), (64, 0, False), (128, 0, False)])
def test_launch_without_a_single_step(pa, dim, n_dist, spatial):
    """A launch whose sequences all hold ONE position has no step at all: only the L2 decay of the touched rows.  The gather index
    arrays of such a launch are never written - whatever an earlier, larger launch left in the workspace is read (and must be
    harmless): run a big launch first, then the empty one, against the oracle's batch rule.  (tools/fuzz_engines.py found an
    aperture violation here: te_wgrad gathered a table row through a stale index.)"""
    from tests.gpu_util import gru_params
    big = toy_problem(77, n_user=700, n_item=900, n_dist=max(n_dist, 300) if spatial else 11, dim=dim, len_max=12)
    if spatial:
        pol = _model(pa, big, spatial_params(77, big))
    else:
        pol = _gru_model(pa, big, gru_params(77, big))
    pol.ctx.set_engine("tile")
    pol.train_batch(np.arange(700, dtype=np.int32))            # fills the workspace with large row ids / float bits
    T = toy_problem(78, n_user=40, n_item=60, n_dist=max(n_dist, 1), dim=dim, len_max=6, min_len=1)
    ones = np.nonzero(T["lens"] == 1)[0].astype(np.int32)
    assert len(ones) >= 3
    if spatial:
        P = spatial_params(78, T)
        exp, outs = _oracle_batch(P, T, ones)
        model = _model(pa, T, P)
        names = SP_NAMES
    else:
        P = gru_params(78, T)
        Pm, Qm, Mm = T["train"][0], T["train"][2], T["train"][1]
        news, touched = [], []
        for u in ones:
            Pn, _ = O.gru_step(P, Pm[u], Qm[u], Mm[u], 0.01, 0.001)
            news.append(Pn); touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u])))))
        exp = batch_mean_update(P, news, touched, ("lt",), ("ui", "wh", "bi"))
        model = _gru_model(pa, T, P)
        names = GRU_NAMES
    model.ctx.set_engine("tile")
    model.train_batch(ones)
    got = _get(model) if spatial else _get_gru(model)
    assert_step_close(got, exp, P, names, "launch without steps")
    model.ctx.set_engine("auto")


def test_forked_write_back_is_bitwise_the_serial_one(pa, monkeypatch):
    """Launches of >= 2048 sequences run the distance-bin chain (te_dsum .. te_dapply) on the side stream next to the POI rows' reduction
    (te_scatter.hip).  The two touch disjoint rows and share only read-only inputs, so the result must be the serial launch's
    (POI_TE_DBG=1) bit for bit - weights, tables and per-sequence losses - over two consecutive launches."""
    T = toy_problem(1901, n_user=2400, n_item=2500, n_dist=200, dim=128, len_max=21, hot=300)
    P = spatial_params(1901, T)
    users = np.random.default_rng(5).permutation(2400).astype(np.int32)
    res, outs = {}, {}
    for mode in ("fork", "serial"):
        if mode == "serial":
            monkeypatch.setenv("POI_TE_DBG", "1")
        else:
            monkeypatch.delenv("POI_TE_DBG", raising=False)
        model = _model(pa, T, P)
        model.ctx.set_engine("tile")
        o1 = np.asarray(model.train_batch(users))
        o2 = np.asarray(model.train_batch(users[::-1].copy()))
        outs[mode] = np.concatenate([o1, o2])
        res[mode] = _get(model)
    monkeypatch.delenv("POI_TE_DBG", raising=False)
    pa._lib.context(0).set_engine("auto")
    assert np.array_equal(outs["fork"], outs["serial"])
    for k in SP_NAMES:
        assert np.array_equal(res["fork"][k], res["serial"][k]), k


@pytest.mark.parametrize("dim,n_item,n_user,n_dist", [(128, 120, 150, 40), (128, 5000, 150, 40), (64, 120, 70, 40), (128, 300, 150, 1520), (256, 300, 40, 1520)])
def test_recurrent_kernel_variants_all_meet_the_oracle(pa, dim, n_item, n_user, n_dist):
    """The three arithmetic forms of the recurrence - 16-sequence MFMA tiles on bf16 x 3 split products (poi_ctx_set_split_products, default
    for launches above the small-launch threshold), on float32-input MFMAs, and one sequence per workgroup on the vector ALUs
    (poi_ctx_set_small_launch, default for launches of <= 1800 sequences) - with the forward table (n_item small against the launch) and
    without (n_item 5000): training launch against the oracle's mean rule at the usual bars, predict against the oracle, and the
    variants against each other (same bars: float32-accurate evaluations of the same step).  1520 bins: the chunked head on split products
    (te_head_big3) against the float32-input one (te_head_big); dim 256: the streaming recurrent kernels in both forms (no per-sequence form)."""
    T = toy_problem(900 + dim + n_item, n_user=n_user, n_item=n_item, n_dist=n_dist, dim=dim, len_max=11, hot=60)
    P = spatial_params(900 + dim, T)
    users = np.random.default_rng(5).permutation(n_user)[: n_user - 5].astype(np.int32)
    exp, outs = _oracle_batch(P, T, users)
    res = {}
    try:
        for split in (True, False, "per-sequence"):
            model = _model(pa, T, P)
            model.ctx.set_engine("tile"); model.ctx.set_split_products(split is True); model.ctx.set_small_launch(1800 if split == "per-sequence" else 0)
            got_out = model.train_batch(users)
            for k, out in enumerate(outs):
                assert_close(got_out[k][:3], out[:3], "losses[%d]" % k, rtol=2e-5)
            got = _get(model)
            assert_step_close(got, exp, P, SP_NAMES, "split products %s" % split)
            model.update_trained_items(); model.update_trained_dists()
            ids = np.arange(n_user, dtype=np.int32)
            hts, sts = model.predict(ids)
            Pn = {k: (np.asarray(v, np.float64) if k != "wd" else float(v)) for k, v in got.items()}
            Pn["h0"] = np.zeros(dim)
            eh, es = O.spatial_predict(Pn, Pn["lt"], Pn["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
            assert_close(hts, eh, "hts", rtol=6e-5 if dim == 256 else 1e-5); assert_close(sts, es, "sts")      # (dim 256: the predict bar of the other tests)
            res[split] = (got, hts)
        for other in (False, "per-sequence"):
            assert_step_close(res[True][0], {k: np.asarray(v, np.float64) if k != "wd" else float(v) for k, v in res[other][0].items()}, P, SP_NAMES, "split vs %s" % other)
            assert not all(np.array_equal(res[True][0][k], res[other][0][k]) for k in ("wh", "ui")), "the switch did not change the arithmetic"
    finally:
        pa._lib.context(0).set_split_products(True); pa._lib.context(0).set_small_launch(1800); pa._lib.context(0).set_engine("auto")


@pytest.mark.parametrize("dim,n_dist,len_max", [(128, 200, 50), (64, 40, 9), (128, 1520, 12), (20, 11, 7), (128, 40, 65), (64, 40, 161)])
def test_one_sequence_path_is_the_reference_step(pa, dim, n_dist, len_max):
    """poi_ctx_set_one_sequence_path: the five-kernel step of one sequence (te_one_*) == the reference step of the oracle, sequentially over
    users whose sequences repeat POIs (hot = 6: the same table row as input, positive and negative target), are as short as one or two
    positions (decay only / a single step) and as long as len_max (161: the path's limit - the reference's Foursquare sequences reach 157), with the padding rows' analytic multiplicities;
    and == the batched pipeline on the same launches (same bars).  dim 20: stored zero-padded to 64."""
    T = toy_problem(1700 + dim + len_max, n_user=14, n_item=60, n_dist=n_dist, dim=dim, len_max=len_max, min_len=1, hot=6)
    for u, L in ((3, 1), (5, 2)):            # a sequence of one position (decay only) and one of two (a single step)
        T["train"][0][u, L:] = T["n_item"]; T["train"][2][u, L:] = T["n_item"]; T["train"][1][u, L:] = 0
        T["dist"][0][u, L:] = n_dist; T["dist"][2][u, L:] = n_dist
        T["lens"][u] = L
    P0 = spatial_params(1700 + dim, T)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    order = [int(np.argmin(T["lens"])), int(np.argmax(T["lens"]))] + list(range(T["n_user"]))
    res = {}
    try:
        for one in (True, False):
            model = _model(pa, T, P0)
            model.ctx.set_engine("tile"); model.ctx.set_one_sequence_path(one)
            model.ctx.timing(True)
            P = P0
            for u in order:
                old = P
                P, out = O.spatial_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
                los, sur, upq, ls = model.train(np.int32(u))
                assert_close([los, sur, upq], out[:3], "losses of user %d (one-sequence path %s)" % (u, one))
                got = _get(model)
                # (len_max >= 50 with six hot POIs: the updates are as large as the weights, |d bi| 0.44; a float32 forward pass lands
                # 0.7e-5 .. 1.4e-5 off here on every engine - the exact forward pass, te_xfwd.hip, holds the bar with no loosening)
                assert_step_close(got, P, old, SP_NAMES, "after user %d (one-sequence path %s, length %d)" % (u, one, T["lens"][u]))
                P = round_f32({**P, **got})
            # the batched pipeline runs te_wgrad, the one-sequence path does not
            assert (model.ctx.timing_get("te_wgrad")[1] == 0) == one, "the launches did not take the expected path"
            model.ctx.timing(False)
            res[one] = _get(model)
        if len_max < 50:      # (the two runs follow their own float32 trajectories: comparable after 16 steps only where the steps are well conditioned)
            for k in SP_NAMES:
                assert_close(res[True][k], res[False][k], "one-sequence path vs batched pipeline " + k, rtol=3e-5)
    finally:
        pa._lib.context(0).set_one_sequence_path(True); pa._lib.context(0).set_engine("auto")


@pytest.mark.parametrize("per_seq", [512, 0], ids=["per-sequence-f64", "int8-tiles"])
@pytest.mark.parametrize("dim,len_max", [(128, 50), (64, 20)])
def test_exact_forward_pass_is_far_inside_the_bar_and_the_switch_changes_the_arithmetic(pa, dim, len_max, per_seq):
    """poi_ctx_set_exact_forward (default on): input product + forward recurrence in fixed point on the int8 matrix cores, float64 gates
    (te_xfwd.hip).  On the step the float32 forward passes miss - one sequence of 50 positions at dim 128, six hot POIs: updates as large
    as the weights - every tensor lands within 2e-6 of the float64 oracle (measured 5e-7: what is left is the float32 BPTT and the
    float32 roundings of z, r, c, h), on the one-sequence path, the batched pipeline and a 70-user launch; with the switch off the same
    launches run the float32 forward kernels (different bits, same toy-size bar).  Both forms of the exact recurrence: one workgroup per sequence
    in float64 on the vector ALUs (te_rec_fwd1x, the default for these launch sizes) and 16-sequence tiles on the int8 matrix cores (te_rec_fwdx)."""
    T = toy_problem(1700 + dim + 50, n_user=80, n_item=60, n_dist=200, dim=dim, len_max=len_max, min_len=1, hot=6)
    P0 = spatial_params(1700 + dim, T)
    users = np.arange(70, dtype=np.int32)
    exp_b, _ = _oracle_batch(P0, T, users)
    Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
    exp_1, _ = O.spatial_step(P0, Pm[0], Qm[0], DPm[0], DQm[0], Mm[0], 0.01, 0.001)
    ctx = pa._lib.context(0)
    res = {}
    try:
        for xf in (True, False):
            for mode in ("one", "batched", "batch-70"):
                model = _model(pa, T, P0)
                ctx.set_engine("tile"); ctx.set_exact_forward(xf, per_seq); ctx.set_one_sequence_path(mode == "one")
                if mode == "batch-70":
                    model.train_batch(users); exp = exp_b
                else:
                    model.train(np.int32(0)); exp = exp_1
                got = _get(model)
                worst = assert_step_close(got, exp, P0, SP_NAMES, "exact forward %s, %s" % (xf, mode), rtol=2e-6 if xf else (1e-5 if len_max < 50 else 4e-5),
                                          delta_rtol=1e-4)
                res[(xf, mode)] = (got, worst)
                if mode == "batch-70":      # predict runs the same exact forward pass: hts of the 50-position sequences to 2e-6 (float32 forward: 1e-5 class)
                    model.update_trained_items(); model.update_trained_dists()
                    ids = np.arange(T["n_user"], dtype=np.int32)
                    hts, sts = model.predict(ids)
                    Pn = {k: (np.asarray(v, np.float64) if k != "wd" else float(v)) for k, v in got.items()}
                    Pn["h0"] = np.zeros(dim)
                    eh, es = O.spatial_predict(Pn, Pn["lt"], Pn["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
                    assert_close(hts, eh, "hts (exact forward %s)" % xf, rtol=2e-6 if xf else 4e-5)
                    assert_close(sts, es, "sts (exact forward %s)" % xf, rtol=2e-6 if xf else 4e-5)
        for mode in ("one", "batched", "batch-70"):
            assert not all(np.array_equal(res[(True, mode)][0][k], res[(False, mode)][0][k]) for k in ("wh", "ui", "lt")), "the switch did not change the arithmetic"
    finally:
        ctx.set_exact_forward(True, 512); ctx.set_one_sequence_path(True); ctx.set_engine("auto")


def test_forward_table_over_step_input_pois_is_bitwise_invisible_and_head_switch_meets_the_oracle(pa):
    """poi_ctx_set_option: (1) "forward_table_compact" - the exact forward pass forms its float64 input table over the launch's step-input
    POIs only (te_xcount / te_xassign rank the rows te_slots marked, te_gather translates the ids): the same arithmetic per row, so the
    update is BITWISE the one of the table over every row of the POI table, and both meet the oracle; (2) "head_split" - the training head
    on bf16 split products (te_head3, 201 bins: seven bin tiles) and on float32-input matrix instructions (te_head): both within the bars,
    different bits; (3) unknown names and values out of range are refused."""
    T = toy_problem(4100, n_user=1800, n_item=2500, n_dist=200, dim=128, len_max=10, hot=400)
    P = spatial_params(4101, T)
    users = np.random.default_rng(9).permutation(1800)[:1700].astype(np.int32)
    exp, outs = _oracle_batch(P, T, users)
    ctx = pa._lib.context(0)
    res = {}
    try:
        for name, opts in (("default", {}), ("full-table", {"forward_table_compact": 0}), ("f32-head", {"head_split": 0})):
            model = _model(pa, T, P)
            ctx.set_engine("tile")
            for k, v in opts.items():
                ctx.set_option(k, v)
            got_out = np.asarray(model.train_batch(users))
            assert_close(got_out[:, :3], np.array([[float(o[0]), float(o[1]), float(o[2])] for o in outs]), "losses (%s)" % name, rtol=2e-5)
            res[name] = (_get(model), got_out)
            assert_step_close(res[name][0], exp, P, SP_NAMES, name)
            ctx.set_option("forward_table_compact", 1); ctx.set_option("head_split", 1)
        assert np.array_equal(res["default"][1], res["full-table"][1])
        for k in SP_NAMES:
            assert np.array_equal(res["default"][0][k], res["full-table"][0][k]), "the compact forward table changed " + k
        assert not all(np.array_equal(res["default"][0][k], res["f32-head"][0][k]) for k in ("vs", "wh", "ui")), "the head switch did not change the arithmetic"
        with pytest.raises(pa.PoiError):
            ctx.set_option("no_such_option", 1)
        with pytest.raises(pa.PoiError):
            ctx.set_option("head_split", 2)
    finally:
        ctx.set_option("forward_table_compact", 1); ctx.set_option("head_split", 1); ctx.set_engine("auto")


def test_stream_placements_and_transposed_bptt_are_bitwise_invisible(pa):
    """Round 4 moved work without changing arithmetic: the BPTT tile kernel with transposed products (te_rec_bwd16t: a lane owns four
    consecutive units of one sequence) evaluates the same formulas on the same MFMA products as te_rec_bwd16<SP>, and the side-stream placements
    (weight packs next to the index preparation, slot sort forked behind it, S-row assignment behind the sort) only reorder independent
    kernels.  The tuning bits of POI_TE_DBG (read at every launch) switch each of them back: the placements must not change the update by one
    bit; the two BPTT kernels agree to float32 rounding (1e-6 of the max-norm: the compiler contracts the gate-derivative expressions
    differently, and d bi is summed over the sequences in another order)."""
    import os
    T = toy_problem(5200, n_user=1800, n_item=2500, n_dist=200, dim=128, len_max=10, hot=400)
    P = spatial_params(5201, T)
    users = np.random.default_rng(11).permutation(1800)[:1700].astype(np.int32)
    ctx = pa._lib.context(0)
    res = {}
    old = os.environ.get("POI_TE_DBG")
    try:
        for name, bits in (("default", 0), ("bptt one unit of four sequences per lane", 128), ("packs inline", 256), ("S rows on the main stream", 1024),
                           ("sort behind te_gemm_ax", 2048)):
            os.environ["POI_TE_DBG"] = str(bits)
            model = _model(pa, T, P)
            ctx.set_engine("tile")
            out = np.asarray(model.train_batch(users))
            res[name] = (_get(model), out)
        ref, ref_out = res["default"]
        for name, (got, out) in res.items():
            assert np.array_equal(out, ref_out), name
            for k in SP_NAMES:
                if "bptt" in name:      # same formulas, another instruction selection (fma contraction) and d bi summation order: float32 rounding apart
                    assert_close(got[k], np.asarray(ref[k], np.float64) if k != "wd" else float(ref[k]), k + ", " + name, rtol=1e-6)
                else:
                    assert np.array_equal(got[k], ref[k]), "%s changed %s" % (name, k)
    finally:
        if old is None:
            os.environ.pop("POI_TE_DBG", None)
        else:
            os.environ["POI_TE_DBG"] = old
        ctx.set_engine("auto")


@pytest.mark.parametrize("dim,force", [(128, 0), (128, 48), (128, 16), (128, 80), (64, 80), (128, 150)])      # (dim 64: the hybrid is off there - the option must be harmless)
def test_hybrid_recurrences_match_the_oracle_and_the_plain_tiles(pa, dim, force):
    """Round 6, TeArgs.hyb: the leading sequences of a launch run on the per-sequence recurrent kernels (te_rec_fwd1x / te_rec_bwd1) WHILE the rest
    runs in 16-sequence tiles (te_rec_fwdx / te_rec_bwd16t) on another stream.  Forced splits (option "hybrid_force": 16, 48, 80 leading
    sequences, all 150 = no tile at all; 0 = the device's own cost model) against the float64 oracle's batch and against the launch without
    the hybrid - same bars as every other engine test; launches are repeated to check reproducibility across the two streams."""
    T = toy_problem(700 + dim + force, n_user=160, n_item=220, n_dist=60, dim=dim, len_max=14, hot=40)
    P = spatial_params(701 + dim, T)
    lens = np.asarray(T["train"][1]).sum(axis=1)
    users = np.random.default_rng(3).permutation(160)[:150].astype(np.int32)
    if force != 48:
        users = users[np.argsort(-lens[users], kind="stable")]            # a length-sorted launch, as bench.py / the harness build them
    # (force 48: the caller's order, unsorted - the split is then merely not the best one; every sequence still goes through exactly one kernel family)
    exp, outs = _oracle_batch(P, T, users)
    got = {}
    for mode in ("hybrid", "hybrid again", "tiles"):
        model = _model(pa, T, P)
        model.ctx.set_engine("tile")
        model.ctx.set_option("hybrid_min", 2); model.ctx.set_option("hybrid_force", force)
        model.ctx.set_option("hybrid", 0 if mode == "tiles" else 1)
        model.ctx.set_small_launch(0); model.ctx.set_exact_forward(True, 0)      # (no all-per-sequence launch: tiles unless the hybrid says otherwise)
        try:
            out = np.asarray(model.train_batch(users))
            got[mode] = (_get(model), out)
        finally:
            model.ctx.set_option("hybrid", 1); model.ctx.set_option("hybrid_min", 1150); model.ctx.set_option("hybrid_force", 0)
            model.ctx.set_small_launch(1800); model.ctx.set_exact_forward(True, 1100); model.ctx.set_engine("auto")
    for mode in ("hybrid", "tiles"):
        for k, out in enumerate(outs):
            assert_close(got[mode][1][k][:3], out[:3], "%s losses[%d]" % (mode, k), rtol=2e-5)
        assert_step_close(got[mode][0], exp, P, SP_NAMES, mode)
    for k in SP_NAMES:          # two streams, one result
        assert np.array_equal(np.asarray(got["hybrid"][0][k]), np.asarray(got["hybrid again"][0][k])), k
    assert np.array_equal(got["hybrid"][1], got["hybrid again"][1])
