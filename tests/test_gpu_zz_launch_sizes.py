"""-m gpu: regression guard on the per-launch time at the sizes the multi-GPU schedules live on.

Round 4's committed record held 1011 us for a 1563-user launch where every other run of the same tree - before and after - measured
676 - 707 us: one 60-launch window of that box, not the code (bisected in round 5: HEAD and the three commits before it all at 676 - 684).
bench.py now takes the median of three windows; this test re-measures the launch sizes of the committed profile (best of three windows: a
transient can only make a window slower) and fails when a size is more than 15 % slower than the newest profiles/rNN_bench.json."""
import glob
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = (256, 1563, 4096)
SLACK = 1.15


def _committed():
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")), reverse=True):
        try:
            sw = json.load(open(path)).get("launch_sweep") or {}
        except Exception:      # noqa: BLE001
            continue
        if all("B=%d" % b in sw for b in SIZES):
            return os.path.basename(path), {b: sw["B=%d" % b]["us_per_launch"] for b in SIZES}      # (the committed MEDIAN; measured here: the best of three windows)
    return None, None


def test_launch_sizes_within_15_percent_of_the_committed_profile():
    import torch
    assert torch.cuda.is_available()
    name, want = _committed()
    if want is None:
        pytest.skip("no committed profile with a launch sweep")
    if name.startswith("r04"):
        want[1563] = min(want[1563], 684.0)      # (the outlier itself: every other record of that tree)
    import poi_amd
    from poi_amd import data as pdata
    n_item, n_user, max_len, D = pdata.SHAPES["gowalla"]
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=0.8)
    tab = ds.shard(0, n_user)
    m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                     n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D, device="cuda:0", seed=7, coords=ds.coords)
    m.ctx.set_batch_cap(64.0)
    lens = np.diff(tab.off.astype(np.int64))
    got = {}
    try:
        for B in SIZES:
            ids = np.random.default_rng(B).permutation(n_user)[:B]
            ids = torch.as_tensor(ids[np.argsort(-lens[ids], kind="stable")].astype(np.int32)).cuda()
            for _ in range(8):
                m.train_batch(ids, sync=False)
            best = 1e30
            for _w in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(40):
                    m.train_batch(ids, sync=False)
                torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 40)
            got[B] = 1e6 * best
    finally:
        m.ctx.set_batch_cap(1.0)
    slow = {B: (round(got[B], 1), round(want[B], 1)) for B in SIZES if got[B] > SLACK * want[B]}
    assert not slow, "launch sizes slower than %s by more than 15 %% (measured us, committed us): %s" % (name, slow)
