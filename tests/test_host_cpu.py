"""CPU-only tests: the C-ABI library loads and exports exactly what include/poi_hip.h declares, and the
host-side input-contract code (data.py) agrees with the reference's golden vectors.  No GPU compute."""
import os
import re

import numpy as np
import pytest

import poi_amd
from poi_amd import data as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    poi_amd.build.build_lib()
    return poi_amd._lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "poi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(poi_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(poi_amd._lib.SIGNATURES), declared ^ set(poi_amd._lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.poi_abi_version() == poi_amd._lib.ABI_VERSION


def test_argument_counts_match_header(lib):
    hdr = open(os.path.join(ROOT, "include", "poi_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for name, (_, args) in poi_amd._lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, hdr)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), (name, params, len(args))


def test_models_refuse_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    T = dict(train=[[[0, 1]], [[1, 1]], [[1, 0]]], test=[[[1]], [[1]], [[0]]])
    with pytest.raises(poi_amd.PoiError):
        poi_amd.models.OboBpr(train=T["train"], test=T["test"], alpha_lambda=[0.01, 0.001], n_user=1, n_item=2, n_in=4, n_hidden=4)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "point-of-interest-recommendation_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_cal_dis_vec_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cal_dis.npz"))
    assert np.array_equal(D.cal_dis_vec(g["lat1"], g["lon1"], g["lat2"], g["lon2"], 200, 200), g["bins_dd200_B200"])
    assert np.array_equal(D.cal_dis_vec(g["lat1"], g["lon1"], g["lat2"], g["lon2"], 25, 1520), g["bins_dd25_B1520"])


def test_csr_roundtrip_and_bins_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    N, B, dd = int(g["n_item"]), int(g["n_dist"]), float(g["dd"])
    lens = g["lens"]
    off, p = D.padded_to_csr(g["pois_m"], lens)
    assert np.array_equal(p, g["ragged_pois"])
    assert np.array_equal(D.csr_to_padded(off, p, N), g["pois_m"])
    _, q = D.padded_to_csr(g["negs"], lens)
    assert np.array_equal(D.dist_pos_bins(off, p, g["coords"], dd, B), g["ragged_dist"])
    dq = D.dist_neg_bins(off, p, q, g["coords"], dd, B)
    assert np.array_equal(D.csr_to_padded(off, dq, B), g["dist_neg"])


def test_negative_sampler_contract():
    ds = D.make_synthetic(200, 500, 15, seed=3)
    off = ds.off.astype(np.int64)
    assert ds.lens.min() >= 4 and ds.lens.max() <= 15
    assert np.all(ds.tra_dp[off[:-1]] == ds.dist_num) and np.all(ds.tra_dq[off[:-1]] == ds.dist_num)
    for u in range(ds.n_user):
        own = set(ds.tra_p[off[u]:off[u + 1]])
        assert not own & set(ds.tra_q[off[u]:off[u + 1]])
        assert ds.tes_q[u] not in own and ds.tes_q[u] != ds.tes_p[u]
    assert ds.tra_q.min() >= 0 and ds.tra_q.max() < ds.n_item
    pad = ds.to_padded()
    assert pad["train"][0].shape == (200, ds.len_max) and pad["train"][0].max() == ds.n_item
    assert pad["dist"][0].max() == ds.dist_num
    before = ds.tra_q.copy()
    ds.resample_negatives(np.random.default_rng(9))
    assert not np.array_equal(before, ds.tra_q)


def test_shard_users_partitions_everything():
    lens = np.random.default_rng(0).integers(4, 50, 1000)
    for ws in (1, 2, 3, 8):
        cover = []
        for r in range(ws):
            lo, hi = D.shard_users(1000, ws, r, lens)
            cover.extend(range(lo, hi))
        assert cover == list(range(1000))
        los = [D.shard_users(1000, ws, r)[0] for r in range(ws)]
        assert los == sorted(los)
    # balanced by check-ins within 5 %
    work = [lens[slice(*D.shard_users(1000, 8, r, lens))].sum() for r in range(8)]
    assert max(work) / (sum(work) / 8) < 1.05


def test_bin_thresholds_reproduce_cal_dis(golden_dir):
    """bin(c) = #{k: c >= thr[k-1]} must equal int(d*asin(sqrt(c))*1000/dd) for every c, in particular
    right at and next to each threshold, and on the reference's golden pairs."""
    import math
    thr = D.bin_thresholds(200, 200)
    assert np.all(np.diff(thr) > 0)

    def f(c):
        return min(int(12742 * math.asin(math.sqrt(c)) * 1000 / 200), 200)
    rng = np.random.default_rng(0)
    cs = np.concatenate((thr, np.nextafter(thr, 0), np.nextafter(thr, 1), rng.uniform(0, thr[-1] * 1.2, 20000)))
    got = np.searchsorted(thr, cs, side="right")
    assert np.array_equal(got, [f(c) for c in cs])
    g = np.load(os.path.join(golden_dir, "cal_dis.npz"))
    p = D.DEG
    a = (g["lat1"] - g["lat2"]) * p; b = (g["lon1"] - g["lon2"]) * p
    c = (1.0 - np.cos(a)) / 2 + np.cos(g["lat1"] * p) * np.cos(g["lat2"] * p) * (1.0 - np.cos(b)) / 2
    assert np.array_equal(np.searchsorted(thr, c, side="right"), g["bins_dd200_B200"])
    assert np.array_equal(D.cos_lat(np.stack([g["lat1"], g["lon1"]], 1)), np.cos(g["lat1"] * p))


def test_rank_metrics_match_reference_golden(golden_dir):
    """evaluate.rank_metrics (vectorised Valuate.py:149-172) vs the reference's per-row helpers."""
    from poi_amd.evaluate import rank_metrics
    g = np.load(os.path.join(golden_dir, "metrics.npz"))
    m = rank_metrics(g["recom"], g["test_lst"], g["test_mask"], [20])[20]
    assert m["hits"] == g["zero_one"].sum()
    assert np.isclose(m["map"], g["map"].mean(), rtol=1e-12) and np.isclose(m["ndcg"], g["ndcg"].mean(), rtol=1e-12)
    assert np.isclose(m["recall"], g["zero_one"].sum() / g["test_mask"].sum())


def test_checkpoint_file_is_the_reference_pickle(tmp_path):
    """prog_bpr_gru_spatial.py:323-330 writes cPickle.dump([loss_weight, wd, lt, di, ui, wh, bi, vs, bs], protocol 2);
    :210-213 reads it back into load_params.  Same bytes-level contract here (Python-2-readable protocol, float64,
    reference order), and a Python 2 style pickle is accepted on load."""
    import pickle
    from poi_amd import harness
    rng = np.random.default_rng(0)
    vals = [rng.random(2), np.array(0.3), rng.random((7, 4)), rng.random((5, 4)), rng.random((3, 4, 8)), rng.random((3, 4, 4)),
            rng.random((3, 4)), rng.random((5, 4)), rng.random(5)]
    p = harness.default_params(); p.update(latent_size=4, dataset="toy")
    path = harness.checkpoint_path(p, "OboSpatialGru", 20, root=str(tmp_path))
    assert path.endswith("toy/OboSpatialGru_size4_UD40_dd200_epoch20")
    harness.dump_checkpoint([v.astype(np.float32) for v in vals], path)
    raw = open(path, "rb").read()
    assert raw[:2] == b"\x80\x02"                                  # pickle protocol 2
    # every GLOBAL must name a module Python 2 + an old numpy can import: numpy >= 2 pickles arrays through
    # numpy._core.multiarray, which numpy < 1.26 does not have (the reference's cPickle.load would fail)
    import pickletools
    mods = {arg.split(" ")[0] for op, arg, _ in pickletools.genops(raw) if op.name == "GLOBAL"}
    assert mods <= {"numpy.core.multiarray", "numpy", "_codecs"}, mods
    assert not any(op.name == "STACK_GLOBAL" for op, _, _ in pickletools.genops(raw))
    back = harness.read_checkpoint(path)
    assert len(back) == 9 and all(b.dtype == np.float64 for b in back)
    for a, b in zip(vals, back):
        assert np.allclose(a, b, rtol=1e-6) and a.shape == b.shape
    with open(path, "wb") as f:                                     # what a Python 2 reference run would have left
        pickle.dump([np.asarray(v) for v in vals], f, protocol=2)
    for a, b in zip(vals, harness.read_checkpoint(path)):
        assert np.array_equal(a, b)
    with open(path, "wb") as f:
        pickle.dump(vals[:8], f, protocol=2)
    with pytest.raises(ValueError):
        harness.read_checkpoint(path)


def test_coalesce_ranges_merges_contiguous_batches_only():
    import importlib
    ev = importlib.import_module("poi_amd.evaluate")
    batches = [np.arange(s, min(s + 32, 100), dtype=np.int32) for s in range(0, 100, 32)]
    out = ev.coalesce_ranges(batches, target=70)
    assert [len(o) for o in out] == [64, 36] and np.array_equal(np.concatenate(out), np.arange(100))
    odd = [np.array([5, 6, 7]), np.array([9, 10]), np.array([11, 12]), np.array([3, 1])]
    out = ev.coalesce_ranges(odd, target=100)
    assert [list(o) for o in out] == [[5, 6, 7], [9, 10, 11, 12], [3, 1]]


@pytest.mark.parametrize("split", [-1, -2])
def test_load_sequence_file_matches_the_reference_load_data(golden_dir, split):
    """data.load_sequence_file == public/Load_Data_by_length.py:45-112 on the same file (golden = the reference's own output, made by
    tests/golden/make_golden.py), modulo the POI relabelling: the reference numbers POIs in the iteration order of a set of strings."""
    g = np.load(os.path.join(golden_dir, "load_data_split%d.npz" % -split))
    ds = D.load_sequence_file(os.path.join(golden_dir, "sequences_small.txt"), split=split, dd=200, dist_num=200, seed=3)
    assert (ds.n_user, ds.n_item) == (int(g["user_num"]), int(g["item_num"]))
    # reference alias -> alias here through the coordinate rows (distinct per POI in this file)
    ref_c = g["pois_cordis"]
    key = {tuple(c): i for i, c in enumerate(ds.coords.tolist())}
    assert len(key) == ds.n_item
    to_here = np.array([key[tuple(c)] for c in ref_c.tolist()])                 # every reference coordinate is here, bit for bit
    assert sorted(to_here.tolist()) == list(range(ds.n_item))
    assert np.array_equal(np.diff(ds.off.astype(np.int64)), g["lens"])
    assert np.array_equal(ds.tra_p, to_here[g["tra_pois"]])
    assert np.array_equal(ds.tes_p, to_here[g["tes_pois"]])
    assert np.array_equal(ds.tra_dp, g["tra_dist"])
    assert np.array_equal(ds.tes_dp, g["tes_dist"])
    # the driver's tables right after loading (prog_bpr_gru_spatial.py:86-91): padded form, negatives outside the user's own POIs
    pad = ds.to_padded()
    assert pad["train"][0].shape == (ds.n_user, int(g["lens"].max())) and pad["train"][0].max() == ds.n_item
    off = ds.off.astype(np.int64)
    for u in range(ds.n_user):
        assert not set(ds.tra_q[off[u]:off[u + 1]].tolist()) & set(ds.tra_p[off[u]:off[u + 1]].tolist())
        assert ds.tra_dp[off[u]] == 200 and ds.tra_dq[off[u]] == 200


def test_sequence_file_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    coords = np.stack([40.0 + rng.uniform(0, 0.3, 30), -74.0 + rng.uniform(0, 0.4, 30)], 1)
    seqs = [list(map(int, rng.integers(0, 30, L))) for L in (5, 9, 6, 12)]
    path = str(tmp_path / "seq.txt")
    D.write_sequence_file(path, seqs, coords)
    ds, alias = D.load_sequence_file(path, split=-1, return_aliases=True)
    assert ds.n_user == 4 and ds.n_item == len({i for s in seqs for i in s})
    back = {v: int(k) for k, v in alias.items()}
    off = ds.off.astype(np.int64)
    for u, s in enumerate(seqs):
        assert [back[int(i)] for i in ds.tra_p[off[u]:off[u + 1]]] == s[:-1] and back[int(ds.tes_p[u])] == s[-1]
    assert np.array_equal(ds.coords, coords[[back[a] for a in range(ds.n_item)]])
    with pytest.raises(IndexError):
        D.load_sequence_file(path, split=-13)
