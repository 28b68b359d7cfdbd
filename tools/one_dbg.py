#!/usr/bin/env python
"""One Distance2Pre step of a long toy sequence (six hot POIs, updates as large as the weights) on every path - one-sequence path,
batched pipeline with per-sequence / MFMA recurrent kernels, per-sequence engine - against the float64 oracle: the float32 noise floor of
such a step (tests/test_gpu_tile_engine.py::test_one_sequence_path_is_the_reference_step uses looser bars at len_max >= 50 for it).
    python tools/one_dbg.py [len_max]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import poi_amd
from oracle import poi_oracle as O
from tests.gpu_util import spatial_params, toy_problem, round_f32
SP_NAMES = ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")
dim, n_dist, len_max = 128, 200, int(sys.argv[1]) if len(sys.argv) > 1 else 50
T = toy_problem(1700 + dim + 50, n_user=14, n_item=60, n_dist=n_dist, dim=dim, len_max=len_max, min_len=1, hot=6)
P0 = spatial_params(1700 + dim, T)
Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
def get(m):
    return {k: (float(getattr(m, k).get_value()) if k == "wd" else getattr(m, k).get_value()) for k in SP_NAMES}
for mode in ("one", "batched", "batched-mfma", "seq"):
    m = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"],
                                     n_dists=[n_dist, 0.2], n_in=dim, n_hidden=dim, init=P0)
    m.ctx.set_engine("seq" if mode == "seq" else "tile"); m.ctx.set_one_sequence_path(mode == "one"); m.ctx.set_small_launch(0 if mode == "batched-mfma" else 1024)
    u = 0
    Pn, out = O.spatial_step(P0, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
    r = m.train(np.int32(u))
    g = get(m)
    print(mode, "losses", r[:3], out[:3])
    for k in SP_NAMES:
        a, b, o = np.asarray(g[k], np.float64), np.asarray(Pn[k], np.float64), np.asarray(P0[k], np.float64)
        print("   %-12s w %.2e   delta %.2e  (|delta| max %.2e)" % (k, np.abs(a - b).max() / max(np.abs(b).max(), 1e-30), np.abs((a - o) - (b - o)).max() / max(np.abs(b - o).max(), 1e-30), np.abs(b - o).max()))
m.ctx.set_one_sequence_path(True); m.ctx.set_small_launch(1800); m.ctx.set_engine("auto")
