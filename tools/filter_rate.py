#!/usr/bin/env python
"""How often would an upper-bound filter skip the distance term in the scoring kernel?  (Gowalla shape, a model trained for a few dozen
epochs; 2048 users.)  Tile = 32 users x 32 items; a tile passes when dot + max_b wd * sts[u][b] exceeds the user's FINAL K-th best score for
any of its pairs (the best case: thresholds already at their final value).  Prints the pass rate per tile and per pair."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import poi_amd
from poi_amd import data as pdata
import bench

dev = torch.device("cuda:0")
ni, nu, ml, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(nu, ni, ml, seed=20260928, local=0.8)
tab = ds.shard(0, nu)
model = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=nu, n_item=ni,
                                     n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device=dev, seed=7, coords=ds.coords)
model.ctx.set_batch_cap(64.0)
lens = np.diff(tab.off.astype(np.int64))
_, B, batches = bench.make_batches(nu, lens, 12500)
order = torch.as_tensor(np.concatenate(batches).astype(np.int32)).to(dev)
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for _ in range(epochs):
    for b0 in range(0, nu, B):
        model.train_batch(order[b0:b0 + B])
ids = np.arange(2048, dtype=np.int32)
model.update_trained_items(); model.update_trained_dists()
hts, sts = model.predict_device(ids)
hts = hts[:, :D].float(); sts = sts.float()
lt = model.lt.t[:ni, :D].float()
wd = float(model.wd.get_value().reshape(-1)[0])
dot = hts @ lt.T                                            # 2048 x N
# exact distance term: bins of (last train POI, item)
last = torch.as_tensor(np.array([tab.p[tab.off[u + 1] - 1] for u in ids]).astype(np.int64)).to(dev)
co = torch.as_tensor(ds.coords).to(dev).double()
def bins(u_idx):
    a = co[last[u_idx]][:, None, :] * (np.pi / 180.0); b = co[None, :, :] * (np.pi / 180.0)
    c = torch.sin((b[..., 0] - a[..., 0]) / 2) ** 2 + torch.cos(a[..., 0]) * torch.cos(b[..., 0]) * torch.sin((b[..., 1] - a[..., 1]) / 2) ** 2
    d = 12742.0 * torch.asin(torch.sqrt(c.clamp(0, 1))) * 1000.0 / ds.dd
    return d.long().clamp(max=ds.dist_num)
score = torch.empty_like(dot)
for c0 in range(0, 2048, 256):
    bb = bins(torch.arange(c0, c0 + 256, device=dev))
    score[c0:c0 + 256] = dot[c0:c0 + 256] + wd * torch.gather(sts[c0:c0 + 256], 1, bb)
thr = score.topk(20, dim=1).values[:, -1]
ub = (wd * sts).max(dim=1).values.clamp(min=0)
passed = (dot + ub[:, None]) > thr[:, None]
n32 = (ni // 32) * 32
t = passed[:, :n32].reshape(64, 32, n32 // 32, 32).any(dim=3).any(dim=1)
print("wd %.3f  ub mean %.3f  score spread (std of dot) %.3f" % (wd, ub.mean().item(), dot.std().item()))
print("pairs passing the bound: %.4f %%   tiles (32 x 32) with any passing pair: %.2f %%   (exact candidates per pair: %.4f %%)"
      % (100 * passed.float().mean().item(), 100 * t.float().mean().item(), 100 * 20.0 / ni))
