#!/usr/bin/env python
"""Drift between the tile engine (f32 MFMA, v_rcp gate activations, regrouped sums) and the per-sequence parity
engine over 20 launches of 500 users on the same data: max-norm relative difference per tensor (one-step parity
is 1e-5, tests/test_gpu_tile_engine.py; measured here: <= 8e-5 after 20 launches)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd
from tests.gpu_util import toy_problem, spatial_params
from tests.test_gpu_tile_engine import _model, _get, SP_NAMES
poi_amd._lib.load()
T = toy_problem(77, n_user=600, n_item=800, n_dist=200, dim=128, len_max=30, hot=100)
P = spatial_params(77, T)
rng = np.random.default_rng(0)
res = {}
for eng in ("tile", "seq"):
    m = _model(poi_amd, T, P); m.ctx.set_engine(eng)
    r = np.random.default_rng(1)
    for it in range(20):
        users = r.permutation(600)[:500].astype(np.int32)
        m.train_batch(users)
    res[eng] = _get(m)
for k in SP_NAMES:
    a, b = np.asarray(res["tile"][k], np.float64), np.asarray(res["seq"][k], np.float64)
    print(k, "max rel diff %.2e" % (np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)))
