#!/usr/bin/env python
"""Sanity: does training through the HIP path learn?  (args: batch_users epochs alpha)  Synthetic check-ins have Zipf POI popularity, so
recall@20 should climb far above the random level 20/N.  Usage: learning_check.py [batch_users] [epochs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from poi_amd import harness  # noqa: E402
from poi_amd.data import make_synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
E = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ALPHA = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
ds = make_synthetic(2000, 3000, 20, seed=4)
p = harness.default_params()
p.update(latent_size=64, epochs=E, gru=2, batch_users=B, seed=2, alpha=ALPHA)
pop = np.bincount(ds.tra_p, minlength=ds.n_item)
top20 = set(np.argsort(-pop)[:20])
print("random recall@20 = %.4f ; popularity-baseline recall@20 = %.4f" % (20 / ds.n_item, np.mean([t in top20 for t in ds.tes_p])))
model, best, hist = harness.train_valid_or_test(ds, p)
print("best recall@20 %.4f  best auc %.4f" % (best.best_recall[-1], best.best_auc))
