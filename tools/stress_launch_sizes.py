#!/usr/bin/env python
"""Stress: 40 s of training launches of mixed sizes (1 .. 20000 users: every path - one sequence, per-sequence kernels, two-table, regrouped with the
side-stream forks) enqueued without waiting, predict passes in between; must drain and leave finite parameters.   usage: python tools/stress_launch_sizes.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import poi_amd
from poi_amd import data as pdata
n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=5, local=0.8)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device="cuda:0", seed=7, coords=ds.coords)
m.ctx.set_batch_cap(64.0)
rng = np.random.default_rng(1)
t0 = time.time(); n = 0
sizes = [1, 7, 300, 1024, 1300, 2047, 2048, 2049, 4096, 9000, 12500, 20000]
while time.time() - t0 < 40:
    B = int(rng.choice(sizes))
    ids = rng.permutation(n_user)[:B].astype(np.int32)
    m.train_batch(ids, sync=(n % 17 == 0))
    if n % 50 == 49:
        m.update_trained_items(); m.update_trained_dists()
        h, s = m.predict_device(np.arange(4096, dtype=np.int32))
    n += 1
torch.cuda.synchronize()
ok = all(bool(torch.isfinite(getattr(m, k).t).all()) for k in ("lt", "di", "ui", "wh", "vs"))
print("launches", n, "finite", ok)
