#!/usr/bin/env python
"""Wall time of every piece of one evaluation pass (bench.py's eval_epoch), synchronised per piece: where the first evaluation's ms go.
    python tools/eval_pieces.py"""
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
for ep in range(10):
    _, B, bt = bench.make_batches(n_user, lens, 12500, seed=ep)
    for b in bt: m.train_batch(torch.as_tensor(b.astype(np.int32)).to(dev), sync=False)
ids = np.arange(n_user, dtype=np.int32)
tes = torch.as_tensor(tab.tes_p.reshape(-1).astype(np.int32)).to(dev)
def piece(name, fn, acc):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + 1e3 * (time.perf_counter() - t0); return r
for rnd in range(4):
    acc = {}
    m.reset_topk_seeds()
    t0 = time.perf_counter()
    piece("update_trained_items", m.update_trained_items, acc)
    piece("update_trained_dists", m.update_trained_dists, acc)
    h, s = piece("predict_device", lambda: m.predict_device(ids), acc)
    piece("update_trained_users", lambda: m.update_trained_users(h), acc)
    piece("update_trained_sus", lambda: m.update_trained_sus(s), acc)
    idx = piece("compute_sub_topk", lambda: m.compute_sub_topk(ids, 20), acc)
    piece("hits", lambda: (idx == tes[:, None]).any(dim=1).sum(), acc)
    tot = 1e3 * (time.perf_counter() - t0)
    # the same pass without per-piece synchronisation
    m.reset_topk_seeds(); torch.cuda.synchronize(); t0 = time.perf_counter()
    m.update_trained_items(); m.update_trained_dists(); h, s = m.predict_device(ids); m.update_trained_users(h); m.update_trained_sus(s)
    idx = m.compute_sub_topk(ids, 20); hits = (idx == tes[:, None]).any(dim=1).sum(); torch.cuda.synchronize()
    print({k: round(v, 3) for k, v in acc.items()}, "sum %.2f ms; unsynchronised pass %.2f ms" % (tot, 1e3 * (time.perf_counter() - t0)), flush=True)
