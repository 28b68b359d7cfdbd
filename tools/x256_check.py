#!/usr/bin/env python
"""Dim 256 with the REFERENCE's init (uniform(-0.5, 0.5): public/GRU_Spatial.py:50-71) - the exact forward pass (te_gemmx<256> + te_rec_fwdd) against the
float32 / split-product forward pass, every tensor against the float64 oracle (oracle/c_oracle.spatial_batch_mean, capped-sum rule).
    python tools/x256_check.py [n_user] [len_max] [cap] [dim]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import poi_amd
from oracle import poi_oracle as O, c_oracle as C
from poi_amd.data import padded_to_csr
from tests.gpu_util import spatial_params, toy_problem, rel_err, delta_excess
SP = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
n_user = int(sys.argv[1]) if len(sys.argv) > 1 else 256
len_max = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cap = float(sys.argv[3]) if len(sys.argv) > 3 else 64.0
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 256
n_dist, n_item = 200, 4000
hot = int(sys.argv[5]) if len(sys.argv) > 5 else 8
T = toy_problem(4242, n_user=n_user, n_item=n_item, n_dist=n_dist, dim=dim, len_max=len_max, min_len=4, hot=hot)
P0 = spatial_params(4243, T)
lens = np.asarray(T["lens"])
users = np.argsort(-lens, kind="stable").astype(np.int32)
off, p = padded_to_csr(np.asarray(T["train"][0]), lens); _, q = padded_to_csr(np.asarray(T["train"][2]), lens)
_, dp = padded_to_csr(T["dist"][0], lens); _, dq = padded_to_csr(T["dist"][2], lens)
Pin = {k: np.asarray(P0[k], np.float64) if k != "wd" else float(P0[k]) for k in SP}; Pin["h0"] = np.zeros(dim)
exp, eout, tch = C.spatial_batch_mean(Pin, off, p, q, dp, dq, users, T["len_max"], 0.01, 0.001, cap=cap, threads=8)
print("max |update|: " + "  ".join("%s %.2e" % (k, np.abs(np.asarray(exp[k], np.float64) - np.asarray(Pin[k], np.float64)).max()) for k in SP))
ctx = poi_amd._lib.context(0)
for xf in (True, False):
    m = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                     n_dists=[n_dist, 0.2], n_in=dim, n_hidden=dim, init=P0)
    ctx.set_exact_forward(xf); ctx.set_batch_cap(cap)
    out = np.asarray(m.train_batch(users))
    got = {k: (float(getattr(m, k).get_value()) if k == "wd" else np.asarray(getattr(m, k).get_value(), np.float64)) for k in SP}
    L = lens[users]
    print("xfwd=%d losses by length: " % xf + "  ".join("[%d,%d) %.1e" % (lo, hi, rel_err(out[(L >= lo) & (L < hi), :3], eout[(L >= lo) & (L < hi), :3]))
                                                     for lo, hi in ((4, 10), (10, 20), (20, 30), (30, 40), (40, 51)) if ((L >= lo) & (L < hi)).any()))
    print("xfwd=%d weights (1e-5 bar) | update excess (1e-4 per row): " % xf + "  ".join("%s %.1e|%.2f" % (k, rel_err(got[k], exp[k]), delta_excess(got[k], exp[k], Pin[k])[0]) for k in SP), flush=True)
    # predict: final hidden states against the oracle's
    m2 = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                      n_dists=[n_dist, 0.2], n_in=dim, n_hidden=dim, init=P0)
    m2.update_trained_items(); m2.update_trained_dists()
    ids = np.arange(min(32, n_user), dtype=np.int32)
    hts, sts = m2.predict(ids)
    Pp = dict(Pin)
    eh, es = O.spatial_predict(Pp, Pp["lt"], Pp["di"], np.asarray(T["train"][0])[ids], np.asarray(T["dist"][0])[ids], np.asarray(T["train"][1])[ids])
    print("xfwd=%d predict: hts %.1e  sts %.1e" % (xf, rel_err(hts, eh), rel_err(sts, es)), flush=True)
ctx.set_exact_forward(True); ctx.set_batch_cap(1.0)
