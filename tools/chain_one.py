#!/usr/bin/env python
"""The reference schedule's launch chain for rocprofv3 --kernel-trace: 600 one-user steps through model.train_sequence (the one-sequence path),
or, with POI_TE_ONE=0, through the batched pipeline.    rocprofv3 --kernel-trace -f csv -d out -o k -- python tools/chain_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poi_amd
from poi_amd import data as pdata

n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=0.8)
tab = ds.shard(0, n_user)
m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                 n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device=torch.device("cuda", 0), seed=7, coords=ds.coords)
order = np.random.default_rng(5).permutation(n_user)[:600].astype(np.int32)
m.train_sequence(order)
torch.cuda.synchronize()
