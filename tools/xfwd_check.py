#!/usr/bin/env python
"""Exact forward (poi_ctx_set_exact_forward) against the float32 forward on the step the float32 engines miss: one Distance2Pre step of a
50-position sequence at dim 128 (six hot POIs: updates as large as the weights) on the one-sequence path, the batched pipeline with one
sequence and with a 70-user launch - every tensor against the float64 oracle.
    python tools/xfwd_check.py [len_max] [dim]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import poi_amd
from oracle import poi_oracle as O
from tests.gpu_util import spatial_params, toy_problem, batch_mean_update
SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
len_max = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n_dist = 200
T = toy_problem(1700 + dim + 50, n_user=80, n_item=60, n_dist=n_dist, dim=dim, len_max=len_max, min_len=1, hot=6)
P0 = spatial_params(1700 + dim, T)
Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
def get(m):
    return {k: (float(getattr(m, k).get_value()) if k == "wd" else getattr(m, k).get_value()) for k in SP_NAMES}
def report(tag, g, Pn):
    worst = 0.0
    line = []
    for k in SP_NAMES:
        a, b, o = np.asarray(g[k], np.float64), np.asarray(Pn[k], np.float64), np.asarray(P0[k], np.float64)
        w = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        worst = max(worst, w)
        line.append("%s %.1e" % (k, w))
    print("%-34s worst %.2e | %s" % (tag, worst, "  ".join(line)), flush=True)
users = np.arange(70, dtype=np.int32)
news, touched = [], []
for u in users:
    Pn, _ = O.spatial_step(P0, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
    news.append(Pn); touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), di=np.unique(DPm[u])))
exp_b = batch_mean_update(P0, news, touched, ("lt", "di"), ("ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"))
ctx = poi_amd._lib.context(0)
for xf in (True, False):
    for mode in ("one", "batched-1", "batched-mfma-1", "batch-70", "batch-70-mfma", "batch-70-regrouped"):
        m = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                         n_dists=[n_dist, 0.2], n_in=dim, n_hidden=dim, init=P0)
        ctx.set_engine("tile"); ctx.set_exact_forward(xf); ctx.set_one_sequence_path(mode == "one"); ctx.set_small_launch(0 if "mfma" in mode else 1024)
        ctx.set_regroup_min(0 if "regrouped" in mode else 1280)
        if mode.startswith("batch-70"):
            m.train_batch(users); report("xfwd=%d %s" % (xf, mode), get(m), exp_b)
        else:
            u = 0
            r = m.train(np.int32(u)); report("xfwd=%d %s" % (xf, mode), get(m), news[0])
ctx.set_exact_forward(True); ctx.set_one_sequence_path(True); ctx.set_small_launch(1800); ctx.set_engine("auto"); ctx.set_regroup_min(1280)
