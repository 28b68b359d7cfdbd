#!/usr/bin/env python
"""Learning check of the half-stored POI table (config X): the same training run - data with a next-POI signal (local = 0.8), dim 256,
capped-sum launches - with the table stored as float32, as half with round-to-nearest write-back, and as half with STOCHASTIC rounding
(poi_ctx_set_f16_rounding).  Prints recall@20 / AUC after the same number of epochs.

    python tools/quality_fp16.py [--shape foursquare] [--dim 256] [--epochs 300] [--alpha 0.01]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poi_amd
from poi_amd import data as pdata
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="foursquare")
ap.add_argument("--dim", type=int, default=256)
ap.add_argument("--epochs", type=int, default=300)
ap.add_argument("--alpha", type=float, default=0.01)
ap.add_argument("--cap", type=float, default=64.0)
ap.add_argument("--batch", type=int, default=12500)
a = ap.parse_args()
n_item, n_user, max_len, _ = pdata.SHAPES[a.shape]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 5, local=0.8)
tab = ds.shard(0, n_user)
dev = torch.device("cuda", 0)
lens = np.diff(tab.off.astype(np.int64))
_, B, batches = bench.make_batches(n_user, lens, a.batch, seed=321)
order = torch.as_tensor(np.concatenate(batches).astype(np.int32)).to(dev)
pop = np.bincount(tab.p, minlength=n_item); top = np.argsort(-pop)[:20]
print(json.dumps({"shape": a.shape, "dim": a.dim, "epochs": a.epochs, "users_per_launch": B, "cap": a.cap,
                  "popularity_recall_at_20": float(np.isin(tab.tes_p.reshape(-1), top).mean())}), flush=True)
for name, tdt, mode in (("float32 table", "f32", "nearest"), ("half table, round to nearest", "f16", "nearest"), ("half table, stochastic rounding", "f16", "stochastic")):
    m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[a.alpha, 0.001], n_user=n_user, n_item=n_item,
                                     n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=a.dim, n_hidden=a.dim, device=dev, seed=11, coords=ds.coords, table_dtype=tdt)
    m.ctx.set_batch_cap(a.cap); m.ctx.set_f16_rounding(mode, seed=3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for ep in range(a.epochs):
        if ep:
            m.resample_negatives_device(99 * 1000003 + ep)
        for b0 in range(0, n_user, B):
            m.train_batch(order[b0:b0 + B], sync=False)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    rec, auc = bench.evaluate_model(m, tab, n_user, dev)
    l2 = m.l2.eval()
    print(json.dumps({"table": name, "recall_at_20": rec, "auc": auc, "train_seconds": t, "l2_term": l2}), flush=True)
    m.ctx.set_f16_rounding("nearest"); m.ctx.set_batch_cap(1.0)
    del m
