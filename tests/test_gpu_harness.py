"""-m gpu: the driver-shaped epoch loop (harness.train_valid_or_test ~ prog_bpr_gru_spatial.py:182-334)
end to end: one-user-per-step training must reproduce the reference's sequential epoch (checked against
the plain-C float64 oracle on the same shuffled order), and the evaluator's ranks / metrics must match
the oracle's on the scores the model produces."""
import os

import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import poi_oracle as O
from tests.gpu_util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    return poi_amd


def test_one_by_one_epoch_equals_reference_sequential_epoch(pa):
    from poi_amd import harness
    from poi_amd.data import make_synthetic
    ds = make_synthetic(48, 200, 10, seed=11)
    p = harness.default_params()
    p.update(latent_size=16, epochs=1, gru=2, batch_users=1, seed=5)
    probe = harness.build_model(ds, p, seed=5)                    # same seed -> same initial parameters
    names = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
    P = {k: np.asarray(getattr(probe, k).get_value(), np.float64) for k in names}
    P["wd"] = float(P["wd"]); P["h0"] = np.zeros(16)
    order = np.random.default_rng(123).permutation(ds.n_user).astype(np.int32)
    exp_out = C.spatial_epoch(P, ds.off, ds.tra_p, ds.tra_q, ds.tra_dp, ds.tra_dq, order, ds.len_max, 0.01, 0.001)
    model, best, hist = harness.train_valid_or_test(ds, p, log=lambda *a: None)
    # 48 sequential steps (exact forward pass, float32 behind it) vs float64: still inside the ONE-step bar
    for k in names:
        assert_close(np.asarray(getattr(model, k).get_value(), np.float64), np.asarray(P[k]), k)
    assert np.isclose(hist[0]["loss"], exp_out[:, 0].sum(), rtol=1e-4)
    assert np.isclose(hist[0]["l2"], O.l2_value(P, 0.001, names), rtol=1e-4)
    # evaluation: fused top-K == oracle ordering of the scores the same model returns
    ids = np.arange(ds.n_user, dtype=np.int32)
    sc = model.compute_sub_all_scores(ids)
    ranks = model.compute_sub_topk(ids, 20).cpu().numpy()
    assert np.array_equal(ranks, O.topk_desc(sc, 20))
    from poi_amd.evaluate import device_rank_metrics
    dev_m = device_rank_metrics(model, [ids], [5, 10, 15, 20])
    exp_m = O.evaluate_ranks(ranks, ds.tes_p.reshape(-1, 1), np.ones((ds.n_user, 1), int), [5, 10, 15, 20])
    from poi_amd.evaluate import rank_metrics
    got_m = rank_metrics(ranks, ds.tes_p.reshape(-1, 1), np.ones((ds.n_user, 1), int), [5, 10, 15, 20])
    for k in (5, 10, 15, 20):
        for key in ("hits", "recall", "precision", "f1", "map", "ndcg"):
            assert np.isclose(got_m[k][key], exp_m[k][key], rtol=1e-12, atol=1e-15), (k, key)
            assert np.isclose(dev_m[k][key], exp_m[k][key], rtol=1e-12, atol=1e-15), ("device", k, key)
    # the spatial score includes wd * prob rebuilt on the device: check one row against the oracle path
    hts, sts = model.predict(ids)
    ul = O.compute_distance(ds.to_padded()["train"][0], ds.to_padded()["train"][1], [list(c) for c in ds.coords], ds.dd, ds.dist_num)
    prob = O.acquire_prob(sts.astype(np.float64), ul, ds.dist_num)
    exp_sc = O.score_all(hts.astype(np.float64), model.trained_items.get_value().astype(np.float64), float(model.wd.get_value()), prob)
    assert_close(sc, exp_sc, "spatial scores incl. distance term")


@pytest.mark.parametrize("gru,batch", [(0, 1), (0, 4096), (1, 1), (1, 16), (2, 64)])
def test_harness_runs_all_model_flags(pa, gru, batch):
    from poi_amd import harness
    from poi_amd.data import make_synthetic
    ds = make_synthetic(40 if batch == 1 else 96, 150, 9, seed=3)
    p = harness.default_params()
    p.update(latent_size=64 if gru == 2 else 16, epochs=2, gru=gru, batch_users=batch)
    model, best, hist = harness.train_valid_or_test(ds, p, log=lambda *a: None)
    assert len(hist) == 2 and all(np.isfinite(h["loss"]) and np.isfinite(h["l2"]) for h in hist)
    assert 0.0 <= hist[-1]["auc"] <= 1.0 and best.best_auc >= hist[-1]["auc"] - 1e-12
    assert hist[1]["loss"] != hist[0]["loss"]


def test_replica_sync_hip_delta_kernels(pa):
    """The product's delta arithmetic (poi_delta_make / poi_delta_apply) on device tensors: with a
    simulated second replica the reconciliation must give theta_start + sum of deltas."""
    import torch
    ctx = pa._lib.context(0)
    g = torch.Generator(device="cuda").manual_seed(3)
    base = torch.rand(100003, device="cuda", generator=g, dtype=torch.float32)
    cur = base + torch.rand(100003, device="cuda", generator=g, dtype=torch.float32) * 0.1
    other_delta = torch.rand(100003, device="cuda", generator=g, dtype=torch.float32) * 0.1
    delta = torch.empty_like(base)
    ctx.check(ctx.lib.poi_delta_make(ctx.handle, cur.data_ptr(), base.data_ptr(), delta.data_ptr(), cur.numel(), None))
    assert torch.equal(delta, cur - base)
    summed = delta + other_delta                       # what the all-reduce would deliver
    out = cur.clone()
    ctx.check(ctx.lib.poi_delta_apply(ctx.handle, out.data_ptr(), base.data_ptr(), summed.data_ptr(), out.numel(), None))
    assert torch.equal(out, base + summed)


def test_checkpoint_resume_reproduces_an_uninterrupted_run(pa, tmp_path):
    """Save at epoch 1 (reference cadence/naming), resume from it, and land on the same parameters as the
    uninterrupted 3-epoch run (same per-epoch shuffles and device negative seeds)."""
    from poi_amd import harness
    from poi_amd.data import make_synthetic
    ds = make_synthetic(96, 300, 10, seed=21)
    base = harness.default_params()
    base.update(latent_size=64, epochs=3, gru=2, batch_users=96, seed=9, dataset="ck", model_root=str(tmp_path), save_per_epoch=1)
    full, _, _ = harness.train_valid_or_test(ds, dict(base), log=lambda *a: None)
    resumed, _, _ = harness.train_valid_or_test(make_synthetic(96, 300, 10, seed=21), dict(base, load_epoch=1), log=lambda *a: None)
    for k in harness.CKPT_ORDER:
        assert_close(np.asarray(resumed.__dict__[k].get_value(), np.float64), np.asarray(full.__dict__[k].get_value(), np.float64), k, rtol=1e-5)


def test_cal_s_mode_saves_the_bin_probabilities_of_a_checkpoint(pa, tmp_path):
    """Mode 's' of the reference driver (prog_bpr_gru_spatial.py:337-362): load a checkpoint, predict every user, np.save(sts)."""
    from poi_amd import data as pdata, harness
    ds = pdata.make_synthetic(60, 120, 9, seed=5)
    p = harness.default_params(); p.update(latent_size=64, epochs=3, batch_users=20, save_per_epoch=1, model_root=str(tmp_path / "model"), dataset="toy")
    model, _, _ = harness.train_valid_or_test(ds, p, log=lambda *a: None)
    p2 = dict(p); p2["load_epoch"] = 2
    path, sts = harness.cal_s(ds, p2, out_root=str(tmp_path / "Lmdd"), log=lambda *a: None)
    assert path.endswith("toy_size64_UD40_dd200_epoch2last1.npy") and os.path.exists(path)
    saved = np.load(path)
    assert saved.shape == (60, ds.dist_num + 1) and np.allclose(saved.sum(axis=1), 1.0, atol=1e-5)
    # the same rows as predicting with the trained model itself (the checkpoint of the last epoch IS the final state)
    model.update_trained_items(); model.update_trained_dists()
    _, exp = model.predict(np.arange(60, dtype=np.int32))
    assert np.allclose(saved, exp, rtol=1e-6, atol=1e-7)


def test_reference_sequence_file_through_the_driver_loop_dim32(pa, golden_dir):
    """BASELINE.json configs[0] in miniature: a sequence file in the ETL's format -> data.load_sequence_file (== the reference's load_data,
    tests/test_host_cpu.py) -> the driver loop, Distance2Pre at dim 32, one user per step - against the plain-C float64 oracle's
    sequential epoch on the same shuffled order."""
    from poi_amd import harness
    path = os.path.join(golden_dir, "sequences_small.txt")
    p = harness.default_params()
    p.update(latent_size=32, epochs=1, gru=2, batch_users=1, seed=5, dataset=path, split=-1, UD=40, dd=200)
    ds = harness.load_dataset(p)
    assert (ds.n_user, ds.n_item, ds.dist_num) == (14, 45, 200)
    probe = harness.build_model(ds, p, seed=5)
    names = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
    P = {k: np.asarray(getattr(probe, k).get_value(), np.float64) for k in names}
    P["wd"] = float(P["wd"]); P["h0"] = np.zeros(32)
    order = np.random.default_rng(123).permutation(ds.n_user).astype(np.int32)
    exp_out = C.spatial_epoch(P, ds.off, ds.tra_p, ds.tra_q, ds.tra_dp, ds.tra_dq, order, ds.len_max, 0.01, 0.001)
    model, best, hist = harness.train_valid_or_test(None, p, log=lambda *a: None)
    for k in names:
        assert_close(np.asarray(getattr(model, k).get_value(), np.float64), np.asarray(P[k]), k)
    assert np.isclose(hist[0]["loss"], exp_out[:, 0].sum(), rtol=1e-4)
    ids = np.arange(ds.n_user, dtype=np.int32)
    assert np.array_equal(model.compute_sub_topk(ids, 20).cpu().numpy(), O.topk_desc(model.compute_sub_all_scores(ids), 20))
