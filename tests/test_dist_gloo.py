"""N > 1 path on CPU: user sharding + per-epoch replica reconciliation under torch.distributed (gloo, world size 2).
The protocol - snapshot, delta (+ per-row touch flags), ONE flat all-reduce SUM, combine rule per tensor, rebuild,
re-snapshot - is the product's ReplicaSync; the elementwise arithmetic is a host backend here (the product's is
libpoi_hip.so's poi_sync_* kernels, checked against this same arithmetic in tests/test_gpu_multi.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import poi_amd
from poi_amd.data import make_synthetic, shard_users


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class HostBackend:
    """Same contract as poi_amd.dist.HipSyncBackend on CPU tensors: flat = [deltas of all tensors | touch flags of the
    mean_touched tensors]; apply() combines by rule and re-snapshots."""

    def __init__(self, tensors, rules):
        self.t, self.rules = list(tensors), list(rules)
        self.base = [x.clone() for x in self.t]
        n = sum(x.numel() for x in self.t) + sum(x.shape[0] for x, r in zip(self.t, rules) if r == "mean_touched")
        self.flat = torch.zeros(n)

    def begin_epoch(self):
        self.base = [x.clone() for x in self.t]

    def make_delta(self):
        o = 0
        for x, b in zip(self.t, self.base):
            self.flat[o:o + x.numel()] = (x - b).reshape(-1); o += x.numel()
        for x, b, r in zip(self.t, self.base, self.rules):
            if r == "mean_touched":
                self.flat[o:o + x.shape[0]] = ((x - b).reshape(x.shape[0], -1) != 0).any(dim=1).float(); o += x.shape[0]
        return self.flat

    def apply(self, world):
        o, co = 0, sum(x.numel() for x in self.t)
        for x, b, r in zip(self.t, self.base, self.rules):
            d = self.flat[o:o + x.numel()].reshape(x.shape); o += x.numel()
            if r == "mean":
                d = d / world
            elif r == "mean_touched":
                cnt = self.flat[co:co + x.shape[0]].clamp(min=1.0); co += x.shape[0]
                d = d / cnt.reshape(-1, *([1] * (x.dim() - 1)))
            x.copy_(b + d)
        self.begin_epoch()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lt = torch.arange(40, dtype=torch.float32).reshape(10, 4).clone()
    di = torch.ones(3, 4)
    wh = torch.ones(6)
    rules = ["sum", "mean_touched", "mean"]
    sync = poi_amd.dist.ReplicaSync([lt, di, wh], rules=rules, backend=HostBackend([lt, di, wh], rules))
    ds = make_synthetic(10, 50, 8, seed=1)
    lo, hi = shard_users(10, world, rank, ds.lens)
    for epoch in range(2):
        # each rank "trains" its own user shard: disjoint rows of lt, shared rows of di (row 2 only on rank 1), dense wh
        for u in range(lo, hi):
            lt[u] += (rank + 1) * 0.5 + epoch
        di[0] += 0.5 * (rank + 1)
        if rank == 1:
            di[2] += 4.0
        wh += 0.25 * (rank + 1)
        sync.end_epoch()
    q.put((rank, lt.numpy().copy(), di.numpy().copy(), wh.numpy().copy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_rules_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    (_, lt0, di0, wh0, s0), (_, lt1, di1, wh1, s1) = res
    assert np.array_equal(lt0, lt1) and np.array_equal(wh0, wh1) and np.array_equal(di0, di1)   # replicas identical after sync
    assert s0[0] == 0 and s0[1] == s1[0] and s1[1] == 10                     # shards partition the users
    exp = np.arange(40, dtype=np.float32).reshape(10, 4)
    for (lo, hi), rank in ((s0, 0), (s1, 1)):
        for epoch in range(2):
            exp[lo:hi] += (rank + 1) * 0.5 + epoch
    assert np.allclose(lt0, exp)                                             # sum: every shard's rows keep their full update
    assert np.allclose(wh0, 1.0 + 2 * (0.25 + 0.5) / 2)                      # mean over the world
    assert np.allclose(di0[0], 1.0 + 2 * (0.5 + 1.0) / 2)                    # row moved by both replicas: mean of the two
    assert np.allclose(di0[2], 1.0 + 2 * 4.0)                                # row moved by one replica only: its full update
    assert np.allclose(di0[1], 1.0)


def test_replica_sync_is_identity_at_world1():
    lt = torch.randn(5, 3)
    before = lt.clone()
    sync = poi_amd.dist.ReplicaSync([lt], rules=["mean_touched"], backend=HostBackend([lt], ["mean_touched"]))
    lt += 1.0
    sync.end_epoch()
    assert torch.equal(lt, before + 1.0)


def _bench_worker(rank, world, port, q):
    """bench.py's own N > 1 plumbing on CPU tensors: plan_shard (sharding + launch schedule), the per-epoch ReplicaSync with the default
    rules, and the self-validation block bench.py prints (report(): world size seen, replica checksums)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    n_user, n_item = 400, 300
    ds = make_synthetic(n_user, n_item, 12, seed=3)
    plans = {sch: bench.plan_shard(n_user, ds.lens, world, rank, 100, sch) for sch in ("quality", "throughput")}
    lo, hi, B, batches = plans["quality"]
    lt = torch.zeros(n_item + 1, 4); wh = torch.zeros(8)
    names, rules = ["lt", "wh"], [poi_amd.dist.DEFAULT_RULES["lt"], poi_amd.dist.DEFAULT_RULES["wh"]]

    class B2(HostBackend):
        def checksum(self):
            return int(sum(int(x.view(torch.int32).to(torch.int64).sum()) for x in self.t))
    sync = poi_amd.dist.ReplicaSync([lt, wh], rules=rules, backend=B2([lt, wh], rules), names=names)
    tab = ds.shard(lo, hi)
    for epoch in range(2):
        for ids in batches:                       # one "launch" per batch: every user of the launch moves its first POI's row
            for u in ids:
                lt[int(tab.p[tab.off[u]])] += 1.0
            wh += 0.125
        sync.end_epoch()
    rep = sync.report()
    q.put((rank, {k: (v[0], v[1], v[2], [np.asarray(b) + v[0] for b in v[3]]) for k, v in plans.items()}, lt.numpy().copy(), wh.numpy().copy(), rep))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_sharding_schedule_and_self_check_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    import bench
    ds = make_synthetic(400, 300, 12, seed=3)
    one = bench.plan_shard(400, ds.lens, 1, 0, 100)
    assert len(one[3]) == 4
    for sch, n_launch in (("quality", 4), ("throughput", 2)):
        (lo0, hi0, B0, b0), (lo1, hi1, B1, b1) = res[0][1][sch], res[1][1][sch]
        assert lo0 == 0 and hi0 == lo1 and hi1 == 400                        # the shards partition the users
        assert len(b0) == n_launch and len(b1) == n_launch                   # quality: as many launches per replica as the one-GPU run
        allu = np.sort(np.concatenate(b0 + b1))
        assert np.array_equal(allu, np.arange(400))                          # every user trained exactly once per epoch
        for b in b0:                                                         # launches sorted by descending length
            assert np.all(np.diff(ds.lens[b]) <= 0)
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    for r in res:
        rep = r[4]
        assert rep["world_size"] == 2 and rep["replica_checksums_equal"] and rep["world_size_seen_by_all_gather"] == 2 and rep["epochs_synced"] == 2
        assert rep["rules"] == {"lt": "mean_touched", "wh": "mean"}
    assert np.allclose(res[0][3], 2 * 4 * 0.125)                              # dense: mean over the replicas of 4 launches x 2 epochs
