#!/usr/bin/env python
"""Side-stream placement must not change a bit: train a few large launches under POI_TE_EARLY_BINS=0 and =1 (separate processes: the switch is
read at context creation) and compare checksums of every parameter tensor.   usage: python tools/placement_check.py"""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import hashlib, json, sys
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
for B in (4096, 12500, 2048):
    ids = rng.permutation(n_user)[:B].astype(np.int32)
    out = m.train_batch(ids)
torch.cuda.synchronize()
h = {k: hashlib.sha1(getattr(m, k).get_value().tobytes()).hexdigest() for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")}
h["out"] = hashlib.sha1(np.asarray(out).tobytes()).hexdigest()
print(json.dumps(h))
'''
res = {}
for v in ("0", "1"):
    env = dict(os.environ, POI_TE_EARLY_BINS=v)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    res[v] = json.loads(out.stdout.strip().splitlines()[-1])
same = res["0"] == res["1"]
print("bit-identical" if same else "DIFFERENT", json.dumps({k: (res["0"][k][:8], res["1"][k][:8]) for k in res["0"]}))
sys.exit(0 if same else 1)
