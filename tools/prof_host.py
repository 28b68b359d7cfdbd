import sys, os, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import poi_amd
from poi_amd import data as pdata
shape = sys.argv[1]
n_item, n_user, max_len, D = pdata.SHAPES[shape]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=1)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, seed=7, coords=ds.coords)
lens = np.diff(tab.off.astype(np.int64))
B = min(6250, n_user)
order = torch.as_tensor(np.argsort(-lens[:B], kind="stable").astype(np.int32)).cuda()
for _ in range(3): m.train_batch(order, sync=False)
torch.cuda.synchronize()
for timing in (False, True):
    m.ctx.timing(timing)
    t0 = time.perf_counter()
    for _ in range(50): m.train_batch(order, sync=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s B=%d timing=%s: host enqueue %.3f ms/launch, total %.3f ms/launch" % (shape, B, timing, (t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
m.ctx.timing(False)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): m.train_batch(order, sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
