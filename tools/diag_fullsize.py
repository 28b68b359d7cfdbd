#!/usr/bin/env python
"""Diagnostic: one Gowalla-shape launch vs the C batch oracle; prints per-tensor errors and the worst rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import poi_amd
from poi_amd import data as pdata
from oracle import c_oracle as C
from tests.gpu_util import rel_err, delta_excess

shape = sys.argv[1] if len(sys.argv) > 1 else "gowalla"
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 12500
n_item, n_user, max_len, D = pdata.SHAPES[shape]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=77)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, seed=3, coords=ds.coords)
if len(sys.argv) > 3:
    m.ctx.set_engine(sys.argv[3])
NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
st = lambda: {k: (float(getattr(m, k).get_value()) if k == "wd" else np.asarray(getattr(m, k).get_value(), np.float64)) for k in NAMES}
users = np.random.default_rng(5).permutation(n_user)[:nl].astype(np.int32)
lens = np.diff(tab.off.astype(np.int64))
users = users[np.argsort(-lens[users], kind="stable")]
P = st(); P["h0"] = np.zeros(D)
out = np.asarray(m.train_batch(users))
got = st()
exp, eout, touched = C.spatial_batch_mean(P, tab.off, tab.p, tab.q, tab.dp, tab.dq, users, tab.len_max, 0.01, 0.001)
print("losses rel", rel_err(out[:, :3], eout[:, :3]))
for k in NAMES:
    print("%-12s rel %.3e  delta-excess %s  max|d| %.3e" % (k, rel_err(got[k], exp[k]), delta_excess(got[k], exp[k], P[k]), np.abs(np.asarray(exp[k]) - np.asarray(P[k])).max()))
# worst lt rows
off = tab.off.astype(np.int64)
sel = np.concatenate([np.arange(off[u], off[u + 1]) for u in users])
cnt_p = np.bincount(tab.p[sel], minlength=n_item + 1); cnt_q = np.bincount(tab.q[sel], minlength=n_item + 1)
err = np.abs(got["lt"] - exp["lt"]).max(axis=1)
for r in np.argsort(-err)[:15]:
    print("lt row %6d err %.3e |d_or| %.3e  occurrences p %d q %d" % (r, err[r], np.abs(exp["lt"][r] - P["lt"][r]).max(), cnt_p[r], cnt_q[r]))
tot = cnt_p + cnt_q
for lo, hi in ((1, 2), (2, 8), (8, 64), (64, 65), (65, 256), (256, 1024), (1024, 1 << 30)):
    msk = (tot >= lo) & (tot < hi)
    if msk.any():
        print("entries in [%d,%d): rows %d  max err %.3e  mean err %.3e" % (lo, hi, msk.sum(), err[msk].max(), err[msk].mean()))
errd = np.abs(got["di"] - exp["di"]).max(axis=1)
print("di worst rows", np.argsort(-errd)[:5], errd[np.argsort(-errd)[:5]])
