#!/usr/bin/env python
"""Three processes x 60 full-size training launches (12500 users, everything enqueued without waiting): the parameter tables must come out
bit-identical - a race in the LDS-DMA ring of te_ptab_s3 or in the side-stream forks would show up here.   usage: python tools/determinism_check.py"""
import hashlib, json, os, subprocess, sys
CHILD = r'''
import hashlib, json, sys
sys.path.insert(0, ".")
import numpy as np, torch
import poi_amd
from poi_amd import data as pdata
n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=11, local=0.8)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device="cuda:0", seed=7, coords=ds.coords)
m.ctx.set_batch_cap(64.0)
rng = np.random.default_rng(3)
for it in range(60):
    ids = rng.permutation(n_user)[:12500].astype(np.int32)
    out = m.train_batch(ids, sync=False)
torch.cuda.synchronize()
h = {k: hashlib.sha1(getattr(m, k).get_value().tobytes()).hexdigest()[:10] for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd")}
print(json.dumps(h))
'''
res = []
for v in range(3):
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True)
    if out.returncode != 0:
        print(out.stderr[-1500:]); sys.exit(1)
    res.append(json.loads(out.stdout.strip().splitlines()[-1]))
print("deterministic" if res[0] == res[1] == res[2] else "DIFFERENT", res[0])
