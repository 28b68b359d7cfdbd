#!/usr/bin/env python
"""Bitwise reproducibility of one mid-size launch repeated from the same parameters (race hunting): python tools/repro_check.py <dim> <hybrid 0|1> <force> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import poi_amd
from tests.gpu_util import spatial_params, toy_problem
dim, hyb, force = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
T = toy_problem(700 + dim + force, n_user=160, n_item=220, n_dist=60, dim=dim, len_max=14, hot=40)
P = spatial_params(701 + dim, T)
lens = np.asarray(T["train"][1]).sum(axis=1)
users = np.random.default_rng(3).permutation(160)[:150].astype(np.int32)
users = users[np.argsort(-lens[users], kind="stable")]
NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
ref = None; bad = {}
for r in range(reps):
    m = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                     n_dists=[T["n_dist"], 0.2], n_in=dim, n_hidden=dim, init=P)
    m.ctx.set_engine("tile"); m.ctx.set_option("hybrid_min", 2); m.ctx.set_option("hybrid_force", force); m.ctx.set_option("hybrid", hyb)
    m.ctx.set_small_launch(0); m.ctx.set_exact_forward(True, 0)
    out = m.train_batch(users)
    got = {k: getattr(m, k).t.clone() for k in NAMES}; got["out"] = torch.as_tensor(out)
    if ref is None: ref = got
    else:
        for k in got:
            if not torch.equal(ref[k], got[k]): bad[k] = bad.get(k, 0) + 1
print("dim %d hybrid %d force %d: %d repetitions, tensors that differed from the first run: %s" % (dim, hyb, force, reps, bad or "none"))
