#!/usr/bin/env python
"""Kernel breakdown of the FIRST (unseeded) evaluation against the seeded steady state (Gowalla shape, a model trained for some epochs).
    python tools/eval_cold_breakdown.py [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import poi_amd, bench
from poi_amd import data as pdata
n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260930, local=0.8)
tab = ds.shard(0, n_user)
dev = torch.device("cuda", 0)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device=dev, seed=7, coords=ds.coords)
m.ctx.set_batch_cap(64.0)
lens = np.diff(tab.off.astype(np.int64))
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 100):
    _, B, bt = bench.make_batches(n_user, lens, 12500, seed=ep)
    if ep: m.resample_negatives_device(ep)
    for b in bt: m.train_batch(torch.as_tensor(b.astype(np.int32)).to(dev), sync=False)
ids = np.arange(n_user, dtype=np.int32)
m.update_trained_items(); m.update_trained_dists()
h, s = m.predict_device(ids); m.update_trained_users(h); m.update_trained_sus(s)
m.compute_sub_topk(ids, 20)
KN = ["score_topk", "score_maxpass", "score_filter", "score_rescore", "pack_items", "topk_seed", "topk_merge"]
if os.environ.get("EV_GEO"):      # bins from the N coordinates inside the kernel (poi_score_topk_geo) instead of the resident U x N bin matrix
    m.use_bin_matrix = False
    m.compute_sub_topk(ids, 20)
for name in ("unseeded", "seeded", "unseeded"):
    if name == "unseeded": m.reset_topk_seeds()
    m.ctx.timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.compute_sub_topk(ids, 20)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    kt = {k: m.ctx.timing_get(k) for k in KN}
    m.ctx.timing(False)
    st = m.ctx.topk_filter_stats()
    print(name, "%.2f ms" % (1e3 * dt), "survivors/user %.1f" % (st["survivors"] / max(st["users"], 1)), {k: (round(v[0], 3), v[1]) for k, v in kt.items() if v[1]}, flush=True)
