#!/usr/bin/env python
"""Bitwise reproducibility of a real mid-size launch (Gowalla shape, 1563 / 2048 users: the hybrid recurrences' home) repeated from the same parameters.
    python tools/repro_big.py [users] [reps] [shape]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import poi_amd
from poi_amd import data as pdata
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1563
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
shape = sys.argv[3] if len(sys.argv) > 3 else "gowalla"
n_item, n_user, max_len, D = pdata.SHAPES[shape]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=0.8)
tab = ds.shard(0, n_user)
lens = np.diff(tab.off.astype(np.int64))
ids = np.random.default_rng(B).permutation(n_user)[:B]
ids = torch.as_tensor(ids[np.argsort(-lens[ids], kind="stable")].astype(np.int32)).cuda()
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item, n_dists=[ds.dist_num, ds.dd / 1000.0],
                                 n_in=D, n_hidden=D, device="cuda:0", seed=7, coords=ds.coords)
m.ctx.set_batch_cap(64.0)
NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
start = {k: getattr(m, k).t.clone() for k in NAMES}
ref, bad = None, {}
for r in range(reps):
    for k in NAMES: getattr(m, k).t.copy_(start[k])
    out = m.train_batch(ids, sync=False)
    torch.cuda.synchronize()
    got = {k: getattr(m, k).t.clone() for k in NAMES}; got["out"] = out.clone()
    if ref is None: ref = got
    else:
        for k in got:
            if not torch.equal(ref[k], got[k]): bad[k] = bad.get(k, 0) + 1
print(shape + " shape, %d-user launch, %d repetitions from the same parameters: tensors that differed from the first run: %s" % (B, reps, bad or "none"))
