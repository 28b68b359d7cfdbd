#!/usr/bin/env python
"""Per-kernel time of one training launch as a function of the launch size (Gowalla shape): where small launches lose their time.
    python tools/launch_sizes.py [sizes ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poi_amd
from poi_amd import data as pdata
import bench

sizes = [int(x) for x in sys.argv[1:]] or [1, 16, 64, 256, 1024, 1580, 4096, 12500]
n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
n_item = int(os.environ.get("LS_NITEM", n_item))      # (experiments: a smaller POI table under the same steps)
DD = float(os.environ.get("LS_DD", "200"))        # LS_DD=25 LS_UD=38: the reference's 1520-bin configuration
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=0.8, dd=DD, ud_km=float(os.environ.get("LS_UD", "40")))
tab = ds.shard(0, n_user)
dev = torch.device("cuda", 0)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device=dev, seed=7, coords=ds.coords)
ctx = m.ctx
ctx.set_batch_cap(64.0)
if os.environ.get("LS_GRAPH"):      # hipGraph replay of the launches (poi_ctx_set_graph)
    ctx.set_graph(True)
lens = np.diff(tab.off.astype(np.int64))
KN = ["te_prep", "te_gather", "te_gemm_ax", "te_rec_fwd", "te_head", "te_rec_bwd", "te_psum", "te_wgrad", "te_gemm_dx", "te_finalize", "te_dsum", "te_bin_gemm",
      "te_scatter", "te_tail", "dense_apply"]
for B in sizes:
    ids = np.random.default_rng(B).permutation(n_user)[:B]
    ids = torch.as_tensor(ids[np.argsort(-lens[ids], kind="stable")].astype(np.int32)).to(dev)
    for _ in range(5):
        m.train_batch(ids, sync=False)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        m.train_batch(ids, sync=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    ctx.timing(True)
    for _ in range(10):
        m.train_batch(ids, sync=False)
    kt = {k: ctx.timing_get(k) for k in KN}
    ctx.timing(False)
    print(json.dumps({"launch_users": B, "us_per_launch": round(1e6 * wall, 1), "seq_per_s": round(B / wall),
                      "kernels_us": {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in kt.items() if v[1]}}), flush=True)
