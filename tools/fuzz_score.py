#!/usr/bin/env python
"""Randomised check of the fused scoring + top-K entry point (poi_score_topk) against float64 numpy on random shapes: user / item counts
around the 32-wide tile and item-range boundaries, any dim that is a multiple of 4 up to 256, k = 1 .. 32, with / without the dense
distance term, float32 / half item tables, unseeded / seeded with true, random and malformed seed lists.  Ranks must be bit-exact on
rows whose top-K scores are separated by more than the float32 noise; seeded results must equal the unseeded ones exactly.
Round 3: the reference of every comparison is the ONE-STAGE float32 kernel (poi_ctx_set_topk_filter(0)); the default path - the two-stage
f16 filter + exact rescoring wherever it applies (dims 64 / 128 / 256, >= 128 users, self-seeded or seeded) - must reproduce its ids AND
scores bit for bit, also with the distance term from a resident bin matrix (poi_score_topk_ulptai) or computed on the fly (poi_score_topk_geo).
usage: python tools/fuzz_score.py [n_configs] [seed0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402
from oracle import poi_oracle as O  # noqa: E402  (test infrastructure: the checker)


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = poi_amd._lib.context(0)
    checked = 0
    for s in range(seed0, seed0 + n_cfg):
        rng = np.random.default_rng(77_000 + s)
        n = int(rng.choice([1, 5, 31, 32, 33, 100, 127, 128, 129, 300, 1000, 1100, 2500]))
        N = int(rng.choice([33, 64, 100, 255, 256, 257, 1000, 2047, 2048, 5000, 20011, 70000]))
        D = int(rng.choice([4, 8, 20, 32, 36, 64, 64, 100, 128, 128, 128, 132, 200, 256, 256]))
        K = int(min(rng.choice([1, 2, 5, 10, 20, 31, 32]), N))
        mode = str(rng.choice(["plain", "plain", "prob", "geo", "bins"]))
        if mode == "bins" and D > 128:
            mode = "geo"
        with_prob = mode == "prob" and n * N <= 40_000_000
        if mode == "prob" and not with_prob:
            mode = "plain"
        f16 = bool(rng.random() < 0.3)
        users = (rng.standard_normal((n, D)) / np.sqrt(D)).astype(np.float32)
        items = rng.standard_normal((N, D)).astype(np.float16 if f16 else np.float32)
        prob = rng.random((n, N)).astype(np.float32) if with_prob else None
        wd = np.array([rng.uniform(-0.5, 1.0)], np.float32)
        cfg = dict(seed=s, n=n, N=N, D=D, K=K, mode=mode, f16=f16)
        n_dist = int(rng.choice([7, 200, 300]))
        dist = None
        if mode in ("geo", "bins"):
            from poi_amd.data import bin_thresholds, cos_lat
            coords = np.stack([40.0 + rng.random(N) * 0.3, -74.0 + rng.random(N) * 0.3], 1)
            last = rng.integers(0, N, n).astype(np.int32)
            dd = 200.0
            npad = ((n + 31) // 32) * 32
            sts = (rng.random((npad, n_dist + 1)) ** 3).astype(np.float32); sts /= sts.sum(axis=1, keepdims=True); sts[:, n_dist] = 0.0
            dist = dict(thr=torch.as_tensor(bin_thresholds(dd, n_dist)).cuda(), cph=torch.as_tensor(cos_lat(coords)).cuda(), co=torch.as_tensor(coords).cuda(),
                        last=torch.as_tensor(last).cuda(), sts=torch.as_tensor(sts).cuda(), dd=dd)
            wd = np.array([rng.uniform(0.0, 6.0)], np.float32)
            if mode == "bins":
                bb = 1 if n_dist <= 255 else 2
                bins = torch.zeros((npad // 32) * ((N + 31) // 32) * 1024 * bb, dtype=torch.uint8, device="cuda")
                ctx.check(ctx.lib.poi_ulptai_build(ctx.handle, dist["co"].data_ptr(), dist["cph"].data_ptr(), dist["thr"].data_ptr(), dist["last"].data_ptr(), n, N,
                                                   n_dist, dd, bins.data_ptr(), bb, None))
                dist.update(bins=bins, bb=bb)
        if os.environ.get("FUZZ_VERBOSE"):
            print("config", cfg, flush=True)
        du, di = torch.as_tensor(users).cuda(), torch.as_tensor(items).cuda()
        dp = torch.as_tensor(prob).cuda() if with_prob else None
        dwd = torch.as_tensor(wd).cuda()
        if f16:
            ctx.register_f16(di)
        try:
            def run(seed=None, ks=0, two_stage=True):
                idx = torch.empty((n, K), dtype=torch.int32, device="cuda")
                sc = torch.empty((n, K), dtype=torch.float32, device="cuda")
                ctx.set_topk_filter(("items" if s % 2 else "users") if (two_stage and mode == "geo") else two_stage)      # GEO: both filter loop orders
                if seed is not None:
                    ctx.set_topk_seed(seed, ks)
                try:
                    if mode == "geo":
                        ctx.check(ctx.lib.poi_score_topk_geo(ctx.handle, du.data_ptr(), di.data_ptr(), n, N, D, dwd.data_ptr(), dist["sts"].data_ptr(), dist["co"].data_ptr(),
                                                             dist["cph"].data_ptr(), dist["thr"].data_ptr(), dist["last"].data_ptr(), n_dist, dist["dd"], K, idx.data_ptr(),
                                                             sc.data_ptr(), None))
                    elif mode == "bins":
                        ctx.check(ctx.lib.poi_score_topk_ulptai(ctx.handle, du.data_ptr(), di.data_ptr(), n, N, D, dwd.data_ptr(), dist["sts"].data_ptr(), dist["bins"].data_ptr(),
                                                                dist["bb"], n_dist, K, idx.data_ptr(), sc.data_ptr(), None))
                    else:
                        ctx.check(ctx.lib.poi_score_topk(ctx.handle, du.data_ptr(), di.data_ptr(), n, N, D, dwd.data_ptr() if with_prob else None,
                                                         dp.data_ptr() if with_prob else None, K, idx.data_ptr(), sc.data_ptr(), None))
                finally:
                    ctx.set_topk_filter(True)
                return idx.cpu().numpy(), sc.cpu().numpy()
            base_idx, base_sc = run(two_stage=False)                   # the one-stage float32 kernel: reference of everything below
            i1, s1 = run()                                              # default path, unseeded (self-seeded two-stage where it applies)
            assert np.array_equal(i1, base_idx) and np.array_equal(s1.view(np.uint32), base_sc.view(np.uint32)), ("two-stage unseeded", cfg)
            full = users.astype(np.float64) @ items.astype(np.float64).T + (float(wd[0]) * prob.astype(np.float64) if with_prob else 0.0)
            if dist is not None:
                from poi_amd.data import cal_dis_vec
                co = dist["co"].cpu().numpy(); st64 = dist["sts"].cpu().numpy().astype(np.float64); lastn = dist["last"].cpu().numpy()
                for u in range(n):
                    b = cal_dis_vec(co[lastn[u], 0], co[lastn[u], 1], co[:, 0], co[:, 1], dist["dd"], n_dist)
                    full[u] += float(wd[0]) * np.where(b < n_dist, st64[u][np.minimum(b, n_dist)], 0.0)
            exp = O.topk_desc(full, K)
            kk = min(K + 1, N)
            srt = -np.sort(-full, axis=1)[:, :kk]
            gap = np.min(srt[:, :-1] - srt[:, 1:], axis=1) if kk > 1 else np.full(n, 1.0)
            ok = gap > 3e-6 * max(np.abs(srt).max(), 1.0)
            assert np.array_equal(base_idx[ok], exp[ok]), ("ranks", cfg, int(ok.sum()))
            es = np.take_along_axis(full, exp, axis=1)
            assert np.abs(base_sc[ok] - es[ok]).max(initial=0.0) <= 2e-5 * max(np.abs(es).max(), 1.0), ("scores", cfg)
            checked += int(ok.sum())
            # seeded: the true lists, random lists, malformed lists - identical results
            good = torch.as_tensor(base_idx).cuda()
            rnd = torch.as_tensor(rng.integers(-3, N + 3, (n, K)).astype(np.int32)).cuda()
            for name, seed in (("true", good), ("random / malformed", rnd)):
                for two in (False, True):
                    i2, s2 = run(seed, K, two_stage=two)
                    assert np.array_equal(i2, base_idx) and np.array_equal(s2.view(np.uint32), base_sc.view(np.uint32)), ("seeded", name, "two-stage" if two else "one-stage", cfg)
        finally:
            if f16:
                ctx.unregister_f16(di)
        if (s - seed0) % 20 == 19:
            print("config %d ok (%d users x %d items, dim %d, k %d, %s%s); %d rows checked so far" % (s, n, N, D, K, mode, ", f16" if f16 else "", checked), flush=True)
    print("all %d configurations agree (%d rows with unambiguous ranks)" % (n_cfg, checked))


if __name__ == "__main__":
    main()
