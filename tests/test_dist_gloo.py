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
