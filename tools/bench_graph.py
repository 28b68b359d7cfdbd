#!/usr/bin/env python
"""Launch-bound regime: training steps/s at n sequences per launch, eager launches against hipGraph replay (poi_ctx_set_graph).
usage: python tools/bench_graph.py [--shape gowalla] [--n 1 4 16 64 256]"""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import poi_amd  # noqa: E402
from poi_amd import data  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="gowalla")
    ap.add_argument("--n", type=int, nargs="+", default=[1, 4, 16, 64, 256, 1024])
    ap.add_argument("--seconds", type=float, default=1.5)
    ap.add_argument("--sync", type=int, default=1, help="1: fetch the losses after every launch (the reference's model.train), 0: enqueue only")
    a = ap.parse_args()
    n_item, n_user, L, D = data.SHAPES[a.shape]
    n_user = min(n_user, 20000)
    ds = data.make_synthetic(n_user, n_item, L, seed=1, local=0.8)
    tab = ds.shard(0, n_user)
    rows = []
    for n in a.n:
        row = {"n": n}
        for mode in ("eager", "graph"):
            model = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=len(ds.coords),
                                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device=torch.device("cuda", 0), seed=3, coords=ds.coords)
            model.ctx.set_graph(mode == "graph")
            order = np.random.default_rng(0).permutation(n_user).astype(np.int32)
            dev_order = torch.as_tensor(order).cuda()
            k = 0
            def step():
                nonlocal k
                b0 = (k * n) % (n_user - n)
                k += 1
                return model.train_batch(dev_order[b0:b0 + n], sync=bool(a.sync))
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); cnt = 0
            while time.perf_counter() - t0 < a.seconds:
                for _ in range(20):
                    step()
                cnt += 20
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            row[mode + "_launches_per_s"] = cnt / dt
            row[mode + "_us_per_launch"] = 1e6 * dt / cnt
            model.ctx.set_graph(False)
        row["speedup"] = row["graph_launches_per_s"] / row["eager_launches_per_s"]
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
