#!/usr/bin/env python
"""Throughput of the plain GRU + BPR model (OboGru, public/GRU.py) on the synthetic Gowalla shape:
tile engine vs per-sequence engine.  Usage: python tools/bench_gru.py [epochs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402
from poi_amd import data as pdata  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260930)
tab = ds.shard(0, n_user)
model = poi_amd.models.OboGru(train=tab, test=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=D, n_hidden=D, seed=7)
lens = np.diff(tab.off.astype(np.int64))
perm = np.random.default_rng(1).permutation(n_user)
B = 12500
order = torch.as_tensor(np.concatenate([ids[np.argsort(-lens[ids], kind="stable")] for ids in np.split(perm, range(B, n_user, B))]).astype(np.int32)).cuda()
for eng, n_ep, users in (("tile", epochs, n_user), ("seq", 1, 4096)):
    model.ctx.set_engine(eng)
    def epoch():
        for b0 in range(0, users, B):
            model.train_batch(order[b0:min(b0 + B, users)], sync=False)
    epoch(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_ep):
        epoch()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_ep
    print("%-5s engine: %9.0f seq/s  (%.2f ms per %d users)" % (eng, users / dt, dt * 1e3, users))
model.ctx.set_engine("auto")
