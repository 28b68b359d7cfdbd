#!/usr/bin/env python
"""Throughput of CA-RNN (OboCARNN, public/CA_RNN.py, flag 3) on the synthetic Gowalla shape (per-sequence kernels, carnn.hip).
Usage: python tools/bench_carnn.py [dim] [users]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402
from poi_amd import data as pdata  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 128
users = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
n_item, n_user, max_len, _ = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260930)
tab = ds.shard(0, n_user)
model = poi_amd.models.OboCARNN(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, seed=7, coords=ds.coords)
order = torch.as_tensor(np.random.default_rng(1).permutation(n_user)[:users].astype(np.int32)).cuda()
for B in (1, 256, users):
    n = min(users, 2000 if B == 1 else users)
    def run():
        for b0 in range(0, n, B):
            model.train_batch(order[b0:b0 + B], sync=(B == 1))
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("CA-RNN dim %d, %5d sequences per launch: %9.0f seq/s (%.2f ms per %d users)" % (D, B, n / dt, dt * 1e3, n))
    if B == users:
        model.ctx.timing(True); run(); torch.cuda.synchronize()
        print("   regions (ms per launch): " + ", ".join("%s %.2f" % (k, model.ctx.timing_get(k)[0]) for k in ("carnn_train", "carnn_outer", "carnn_ltgrad", "carnn_apply")))
        model.ctx.timing(False)
