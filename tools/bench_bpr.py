#!/usr/bin/env python
"""BPR-MF step (poi_bpr_step, public/BPR.py:201-241) at the Gowalla shape: every (user, positive, negative) triple of an epoch (prog_bpr_gru_spatial.py:240-244)
in launches of --batch triples; triples/s, HBM fraction on SURVEY.md 8(d)'s 6 D e + 12 bytes per triple, per-region times.
    python tools/bench_bpr.py [--batch N] [--mode snapshot|hogwild] [--shape gowalla|foursquare]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poi_amd
from poi_amd import data as pdata
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=0); ap.add_argument("--mode", default="snapshot"); ap.add_argument("--shape", default="gowalla")
ap.add_argument("--epochs", type=int, default=20); ap.add_argument("--cap", type=float, default=64.0); ap.add_argument("--table-dtype", default="f32")
a = ap.parse_args()
n_item, n_user, max_len, D = pdata.SHAPES[a.shape]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=0.8)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboBpr(train=tab, test=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_in=D, n_hidden=D, device="cuda:0", seed=7,
                          table_dtype=a.table_dtype)
m.ctx.set_batch_cap(a.cap)
u, p, q = m.epoch_triples()
n = u.numel()
B = a.batch or n
def epoch():
    for b0 in range(0, n, B):
        m.train_batch(u[b0:b0 + B], p[b0:b0 + B], q[b0:b0 + B], mode=a.mode, sync=False)
for _ in range(3): epoch()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.epochs): epoch()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.epochs
m.ctx.timing(True)
for _ in range(5): epoch()
kt = {k: m.ctx.timing_get(k) for k in ("bpr_sort", "bpr_users", "bpr_items", "bpr_hogwild")}
m.ctx.timing(False)
e = 2 if a.table_dtype == "f16" else 4
by = n * (3 * D * 4 + 3 * D * 4 + 12) if e == 4 else n * (2 * D * 4 + 4 * D * 2 + 12)      # user row r/w float32, the two POI rows r/w in the table's storage
print(json.dumps({"triples_per_epoch": n, "launch_triples": B, "mode": a.mode, "ms_per_epoch": 1e3 * dt, "triples_per_s": n / dt,
                  "hbm_frac_survey_8d": by / dt / 8e12, "GBps": by / dt / 1e9,
                  "regions_us_per_launch": {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in kt.items() if v[1]}, "finite": bool(torch.isfinite(m.lt.t.float()).all())}))
