"""-m gpu: the product's multi-GPU arithmetic run for real on ONE device.  Replicas are emulated as separate model
instances on the same GPU that start from one snapshot and train different user shards; the reconciliation goes
through the library's poi_sync_* kernels (the flat buffers of the replicas are added where RCCL would add them) and is
compared with the oracle's  theta_start + combine(sum of the shards' deltas).  Plus: the RCCL entry points themselves
(poi_comm_*, poi_allreduce_tables, poi_sync_end_epoch) at world size 1 under torch.distributed."""
import os

import numpy as np
import pytest

from oracle import c_oracle as C
from poi_amd.data import padded_to_csr
from tests.gpu_util import assert_close, assert_delta_close, spatial_params, toy_problem

pytestmark = pytest.mark.gpu
SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")


@pytest.fixture(scope="module")
def pa():
    import torch
    assert torch.cuda.is_available()
    import poi_amd
    poi_amd._lib.load()
    return poi_amd


def test_sync_kernels_three_emulated_replicas_all_rules(pa):
    """make_delta / apply for the three combine rules against the same arithmetic in torch (float32, same order)."""
    import torch
    ctx = pa._lib.context(0)
    g = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g, dtype=torch.float32)
    shapes = [(1001, 64), (203, 128), (3, 64, 128), (7,), (1,)]
    rules = ["sum", "mean_touched", "mean", "mean", "sum"]
    base = [rnd(*s) for s in shapes]
    cur = [b.clone() for b in base]
    sync = pa.dist.ReplicaSync(cur, rules=rules, ctx=ctx, force_backend=True)
    sync.backend.begin_epoch()                              # (world size 1: ReplicaSync itself stays passive)
    world = 3
    total = torch.zeros_like(sync.backend.flat)
    deltas = []
    for r in range(world):
        d = [rnd(*s) * 0.1 - 0.05 for s in shapes]
        d[0][torch.arange(1001, device="cuda") % 3 != r] = 0.0        # table 0: every row moved by exactly one replica
        d[1][torch.arange(203, device="cuda") % (r + 2) == 0] = 0.0    # table 1: rows moved by 0..3 replicas
        for c, b, x in zip(cur, base, d):
            c.copy_(b + x)
        total += sync.backend.make_delta()
        deltas.append([c - b for c, b in zip(cur, base)])
    sync.backend.flat.copy_(total)
    sync.backend.apply(world)
    for i, (c, b, rule) in enumerate(zip(cur, base, rules)):
        s = deltas[0][i] + deltas[1][i] + deltas[2][i]
        if rule == "mean":
            s = s / world
        elif rule == "mean_touched":
            cnt = sum((dl[i].reshape(dl[i].shape[0], -1) != 0).any(dim=1).float() for dl in deltas).clamp(min=1.0)
            s = s / cnt[:, None]
        exp = b + s
        assert torch.allclose(c, exp, rtol=0, atol=2e-7), (i, rule, float((c - exp).abs().max()))
    # the result is the next epoch's snapshot: an immediate second reconciliation with no training changes nothing
    before = [c.clone() for c in cur]
    sync.backend.flat.copy_(sync.backend.make_delta())
    assert float(sync.backend.flat.abs().max()) == 0.0
    sync.backend.apply(world)
    assert all(torch.equal(a, b) for a, b in zip(before, cur))
    cs = sync.backend.checksum()
    cur[2][1, 2, 3] += 1e-3
    assert sync.backend.checksum() != cs
    sync.close()


def test_sync_kernels_half_table_travels_as_half(pa):
    """A table STORED as half (config X's POI table): snapshot and deltas are half (poi_sync_buffer16: half the snapshot memory, half the
    all-reduce bytes), the float32 tensors beside it are unaffected; three emulated replicas, rule mean_touched on the half table -
    result == half(base + sum of the half deltas / touching replicas); the touch count sees a row whose delta is below the
    half resolution of the DELTA buffer; the result is the next snapshot."""
    import torch
    ctx = pa._lib.context(0)
    g = torch.Generator(device="cuda").manual_seed(11)
    rnd = lambda *s: torch.rand(*s, device="cuda", generator=g, dtype=torch.float32)
    tab0 = (rnd(5003, 256) - 0.5).half()
    dense0 = rnd(384, 256)
    cur = [tab0.clone(), dense0.clone()]
    sync = pa.dist.ReplicaSync(cur, rules=["mean_touched", "mean"], ctx=ctx, force_backend=True)
    be = sync.backend
    assert be.flat16 is not None and be.flat16.dtype == torch.float16 and be.n16 >= tab0.numel()
    assert be.n < dense0.numel() + 5003 + 64, "the half table's deltas must not sit in the float32 buffer as well"
    be.begin_epoch()
    world = 3
    tot32, tot16 = torch.zeros_like(be.flat), torch.zeros(be.n16, device="cuda", dtype=torch.float32)
    d16, cnt = [], torch.zeros(5003, device="cuda")
    for r in range(world):
        moved = torch.arange(5003, device="cuda") % (r + 2) != 0
        new = (tab0.float() + (rnd(5003, 256) * 0.02 - 0.01) * moved[:, None]).half()
        cur[0].copy_(new); cur[1].copy_(dense0 + 0.01 * (r + 1))
        tot32 += be.make_delta(); tot16 += be.flat16.float()
        d16.append((new.float() - tab0.float()).half())
        cnt += (new != tab0).any(dim=1).float()
    be.flat.copy_(tot32); be.flat16.copy_(tot16.half())
    be.apply(world)
    ssum = (d16[0].float() + d16[1].float() + d16[2].float()).half().float()          # (what was put into the half buffer)
    exp = (tab0.float() + ssum * (1.0 / cnt.clamp(min=1.0))[:, None]).half()
    # (the kernel fuses the multiply-add: one float32 rounding less than this expression - a half-ulp tie may fall the other way)
    diff = (cur[0].float() - exp.float()).abs()
    assert float((diff == 0).float().mean()) > 0.999 and bool((diff <= 1.0001 * torch.as_tensor(np.spacing(exp.abs().cpu().numpy())).cuda().float()).all())
    assert torch.allclose(cur[1], dense0 + 0.02, rtol=0, atol=2e-7)
    before = cur[0].clone()
    be.make_delta()
    assert float(be.flat16.float().abs().max()) == 0.0 and float(be.flat.abs().max()) == 0.0
    be.apply(world)
    assert torch.equal(cur[0], before)
    sync.close()


@pytest.mark.parametrize("engine,rules", [("seq", None), ("tile", None), ("tile", {"lt": "mean_touched", "di": "sum", "ui": "sum"})])
def test_two_shards_from_one_snapshot_reconcile_to_the_oracle(pa, engine, rules):
    """Shard A and shard B are trained (batch rule, one launch each) by two model instances that start from the same
    parameters; reconciling A's replica with B's deltas must give the oracle's theta_start + combine(dA + dB)."""
    import torch
    dim, n_user = 64, 90
    T = toy_problem(301, n_user=n_user, n_item=150, n_dist=23, dim=dim, len_max=11, hot=40)
    P = spatial_params(301, T)
    lens = T["lens"]
    off, p = padded_to_csr(T["train"][0], lens); _, q = padded_to_csr(T["train"][2], lens)
    _, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
    shards = [np.arange(0, 48, dtype=np.int32), np.arange(48, n_user, dtype=np.int32)]
    mk = lambda: pa.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=n_user,
                                         n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=dim, n_hidden=dim, init=P)
    models = [mk(), mk()]
    models[0].ctx.set_engine(engine)
    syncs = [pa.dist.model_sync(m, rules=rules, force_backend=True) for m in models]
    r = dict(pa.dist.DEFAULT_RULES); r.update(rules or {})
    flats = []
    for m, s, ids in zip(models, syncs, shards):
        s.backend.begin_epoch()
        m.train_batch(ids)
        flats.append(s.backend.make_delta().clone())
    models[0].ctx.set_engine("auto")
    for s in syncs:                                          # both replicas receive the same all-reduced buffer
        s.backend.flat.copy_(flats[0] + flats[1])
        s.backend.apply(2)
    got = [{k: (float(getattr(m, k).get_value()) if k == "wd" else np.asarray(getattr(m, k).get_value(), np.float64)) for k in SP_NAMES}
           for m in models]
    assert syncs[0].backend.checksum() == syncs[1].backend.checksum(), "replicas differ after the reconciliation"
    # oracle: each shard's launch by the batch rule from the common snapshot, then the combine rule
    news = [C.spatial_batch_mean(P, off, p, q, dp, dq, ids, T["len_max"], 0.01, 0.001) for ids in shards]
    exp = {}
    for k in SP_NAMES:
        b = np.asarray(P[k], np.float64)
        d = [np.asarray(n[0][k], np.float64) - b for n in news]
        if r[k] == "sum":
            exp[k] = b + d[0] + d[1]
        elif r[k] == "mean":
            exp[k] = b + (d[0] + d[1]) / 2
        else:
            cnt = np.maximum(sum((np.abs(x).reshape(x.shape[0], -1) > 0).any(axis=1).astype(float) for x in d), 1.0)
            exp[k] = b + (d[0] + d[1]) / cnt.reshape(-1, *([1] * (b.ndim - 1)))
    for k in SP_NAMES:
        assert_close(got[0][k], exp[k], "%s after reconciling two shards (%s)" % (k, r[k]))
        assert_delta_close(got[0][k], exp[k], P[k], "%s after reconciling two shards (%s)" % (k, r[k]), rtol=3e-4)
    for s in syncs:
        s.close()


def test_carnn_replicas_reconcile_every_trainable_tensor(pa):
    """model_sync must cover M (and the interval matrices with the per-matrix mean_touched rule) for OboCARNN, and
    report() must checksum them - two replicas trained on different shards are bit-identical after the reconciliation and M is
    theta_start + mean of the two deltas."""
    import torch
    from oracle import poi_oracle as O
    from tests.gpu_util import round_f32
    T = toy_problem(77, n_user=40, n_item=90, n_dist=11, dim=64, len_max=10, hot=25)
    P = round_f32(O.init_carnn_params(np.random.default_rng(5), T["n_item"], T["n_dist"], T["dim"]))
    mk = lambda: pa.models.OboCARNN(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"],
                                    n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=64, n_hidden=64, init=P)
    models = [mk(), mk()]
    syncs = [pa.dist.model_sync(m, force_backend=True) for m in models]
    assert syncs[0].names == ["lt", "wd", "M"] and syncs[0].rules == ["mean_touched", "mean_touched", "mean"]
    base = {k: getattr(models[0], k).t.clone() for k in ("lt", "wd", "M")}
    shards = [np.arange(0, 22, dtype=np.int32), np.arange(22, 40, dtype=np.int32)]
    flats, after = [], []
    for m, s, ids in zip(models, syncs, shards):
        s.backend.begin_epoch()
        m.train_batch(ids)
        after.append({k: getattr(m, k).t.clone() for k in ("lt", "wd", "M")})
        flats.append(s.backend.make_delta().clone())
    assert not torch.equal(after[0]["M"], after[1]["M"])
    for s in syncs:
        s.backend.flat.copy_(flats[0] + flats[1])
        s.backend.apply(2)
    assert syncs[0].backend.checksum() == syncs[1].backend.checksum(), "CA-RNN replicas differ after the reconciliation"
    for k in ("lt", "wd", "M"):
        assert torch.equal(getattr(models[0], k).t, getattr(models[1], k).t), k
    expM = base["M"] + ((after[0]["M"] - base["M"]) + (after[1]["M"] - base["M"])) / 2
    assert torch.allclose(models[0].M.t, expM, rtol=0, atol=3e-7)
    # an interval matrix only shard A moved keeps its whole update; one both moved takes the mean
    dA = (after[0]["wd"] - base["wd"]).flatten(1).abs().amax(dim=1) > 0
    dB = (after[1]["wd"] - base["wd"]).flatten(1).abs().amax(dim=1) > 0
    cnt = (dA.float() + dB.float()).clamp(min=1.0)[:, None, None]
    expW = base["wd"] + ((after[0]["wd"] - base["wd"]) + (after[1]["wd"] - base["wd"])) / cnt
    assert torch.allclose(models[0].wd.t, expW, rtol=0, atol=3e-7)
    rep = syncs[0].report()
    assert rep["replica_checksums_equal"] and set(rep["rules"]) == {"lt", "wd", "M"}
    for s in syncs:
        s.close()


def test_rccl_entry_points_world_size_1(pa):
    """poi_comm_unique_id / poi_comm_init_rank / poi_allreduce_tables / poi_sync_end_epoch through the library's own RCCL
    communicator, under a one-rank torch.distributed group (the all-reduce of one rank is the identity)."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = pa._lib.context(0)
        t = [torch.rand(300, 64, device="cuda"), torch.rand(3, 64, 64, device="cuda")]
        before = [x.clone() for x in t]
        sync = pa.dist.ReplicaSync(t, rules=["sum", "mean"], ctx=ctx, force=True)
        assert sync.own_comm and sync.backend.lib.poi_comm_world(sync.backend.comm) == 1
        t[0] += 0.5; t[1] *= 2.0
        want = [x.clone() for x in t]
        sync.end_epoch()
        torch.cuda.synchronize()
        assert all(torch.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(t, want))
        rep = sync.report()
        assert rep["rccl_world_size"] == 1 and rep["replica_checksums_equal"] and rep["allreduce_bytes"] == 4 * (300 * 64 + 3 * 64 * 64)
        assert rep["allreduce_ms_last"] >= 0.0
        sync.end_epoch()                                   # nothing trained since: identity
        assert all(torch.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(t, want))
        sync.close()
        del before
        # a half-stored table next to a float32 tensor: the second (float16) all-reduce of poi_sync_end_epoch
        th = [(torch.rand(1000, 256, device="cuda") - 0.5).half(), torch.rand(128, 64, device="cuda")]
        s2 = pa.dist.ReplicaSync(th, rules=["mean_touched", "mean"], ctx=ctx, force=True)
        assert s2.own_comm and s2.backend.flat16 is not None
        th[0][::3] += 0.01; th[1] += 0.25
        want = [x.clone() for x in th]
        s2.end_epoch()
        torch.cuda.synchronize()
        # one rank: its own delta comes back - half(base + half(cur - base)) is cur up to the half rounding of the DELTA (2^-12 |delta|,
        # plus the final rounding; the untouched rows are bit-identical)
        dh = (th[0].float() - want[0].float()).abs()
        ulp = torch.as_tensor(np.spacing(want[0].abs().cpu().numpy())).cuda().float()
        assert bool((dh <= 0.01 * 2.0 ** -11 + ulp).all()) and float((dh == 0).float().mean()) > 0.95 and torch.equal(th[0][1::3], want[0][1::3])
        assert torch.allclose(th[1], want[1], rtol=0, atol=1e-7)
        assert s2.report()["allreduce_bytes"] == 4 * (128 * 64 + 1000) + 2 * 1000 * 256
        s2.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_emulated_replicas_converge_like_one_gpu(pa):
    """Convergence of the per-epoch reconciliation with DEFAULT_RULES: 4 replicas emulated on one GPU (each trains its user
    shard from the epoch-start snapshot in two launches, deltas combined by poi_sync_*) against one GPU making the same
    number of sequential launches per epoch, on data with a next-POI signal.  Same epochs -> recall@20 within 15 %, and
    far above the popularity baseline's neighbourhood (measured at the Gowalla shape: DESIGN.md "Multi-GPU")."""
    from tools import quality
    common = ["--shape", "foursquare", "--users", "4000", "--cap", "16", "--epochs", "40", "--eval-every", "40"]
    one = quality.main(common + ["--batch", "2000"])[-1]
    four = quality.main(common + ["--batch", "500", "--world", "4"])[-1]
    assert one["recall"] > 0.25 and four["recall"] > 0.25, (one, four)
    assert four["recall"] >= 0.85 * one["recall"], (one, four)


def test_bench_two_ranks_end_to_end_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on THIS GPU and gloo as the
    host-side collective (RCCL refuses two ranks on one device): user sharding, the quality schedule, the library's delta / combine
    kernels around a real cross-process all-reduce every epoch, the evaluation on each shard, and the self-check that makes a SCALE run
    validate itself (world size seen, replica checksums equal, exit code 0)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--same-device", "--shape", "foursquare", "--steps", "5", "--warmup", "1",
           "--eval-steps", "1", "--no-quality", "--no-secondary", "--no-cpu-baseline", "--no-exact", "--no-x1", "--full-out", str(tmp_path / "full.json")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 6000                                   # the compact record: a driver that keeps a few KB of the tail still parses it
    d = json.loads(last)
    assert {"metric", "value", "unit", "roofline", "roofline_gather_scatter", "cpu_baseline", "headline", "config"} <= set(d)
    assert json.load(open(tmp_path / "full.json"))["kernels"]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    mg = d["multi_gpu"]
    assert mg["world_size"] == 2 and mg["world_size_seen_by_all_gather"] == 2 and mg["replica_checksums_equal"] and mg["epochs_synced"] >= 5
    assert d["config"]["replica_schedule"] == "quality" and mg["throughput_schedule"]["launches_per_epoch_per_replica"] >= 1


def test_half_table_reconciliation_across_two_processes():
    """poi_sync_buffer16 across REAL processes at config X's scale per call: two ranks (gloo, both on this GPU) reconcile a 1 GB half table
    (4 M x 128) + a float32 tensor - the library's delta / touch-count / combine kernels around cross-process all-reduces of the float32
    buffer and of the 1 GB half buffer in 256 MB slices (dist.ReplicaSync._all_reduce_chunked).  Every rank checks
    half(base + half(d_0 + d_1) / touching replicas) element by element, untouched rows bit-identical, replica checksums equal
    (tests/_sync16_worker.py)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "_sync16_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("[")][-1])
    assert len(res) == 2 and all(x["ok_table"] and x["ok_untouched"] and x["ok_dense"] and x["checksums_equal"] for x in res)
    assert res[0]["allreduce_bytes"] >= (1 << 30) and res[0]["rows_moved_by_both"] > 1000 and res[0]["rows_moved_by_one"] > 1000
