"""N > 1 path on CPU: user sharding + per-epoch replica reconciliation under torch.distributed (gloo,
world size 2).  The delta arithmetic is injected (the product uses the HIP kernels poi_delta_*), the
protocol - snapshot, delta, all-reduce SUM, rebuild, re-snapshot - is the product's ReplicaSync."""
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


def _cpu_delta_ops():
    def make(cur, base, out):
        out.copy_(cur - base)

    def apply(cur, base, dsum):
        cur.copy_(base + dsum)
    return make, apply


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    lt = torch.arange(40, dtype=torch.float32).reshape(10, 4).clone()
    wh = torch.ones(6)
    sync = poi_amd.dist.ReplicaSync([lt, wh], delta_ops=_cpu_delta_ops())
    ds = make_synthetic(10, 50, 8, seed=1)
    lo, hi = shard_users(10, world, rank, ds.lens)
    for epoch in range(2):
        # each rank "trains" its own user shard: touches disjoint rows of lt and the shared dense tensor
        for u in range(lo, hi):
            lt[u] += (rank + 1) * 0.5 + epoch
        wh += 0.25 * (rank + 1)
        sync.end_epoch()
    q.put((rank, lt.numpy().copy(), wh.numpy().copy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_sync_sums_deltas_world2():
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
    (_, lt0, wh0, s0), (_, lt1, wh1, s1) = res
    assert np.array_equal(lt0, lt1) and np.array_equal(wh0, wh1)            # replicas identical after sync
    assert s0[0] == 0 and s0[1] == s1[0] and s1[1] == 10                     # shards partition the users
    exp = np.arange(40, dtype=np.float32).reshape(10, 4)
    for (lo, hi), rank in ((s0, 0), (s1, 1)):
        for epoch in range(2):
            exp[lo:hi] += (rank + 1) * 0.5 + epoch
    assert np.allclose(lt0, exp)
    assert np.allclose(wh0, 1.0 + 2 * (0.25 + 0.5))                          # both ranks' dense deltas summed


def test_replica_sync_is_identity_at_world1():
    lt = torch.randn(5, 3)
    before = lt.clone()
    sync = poi_amd.dist.ReplicaSync([lt], delta_ops=_cpu_delta_ops())
    lt += 1.0
    sync.end_epoch()
    assert torch.equal(lt, before + 1.0)
