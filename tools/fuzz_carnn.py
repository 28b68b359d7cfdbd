#!/usr/bin/env python
"""Randomised check of the CA-RNN training launch (both paths: dims 64 / 128 outer-product path, other dims per-sequence kernel)
against the float64 oracle's batch rule on random small shapes, inside one process (every launch meets an earlier launch's workspace).
usage: python tools/fuzz_carnn.py [n_configs] [seed0]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402
from oracle import poi_oracle as O  # noqa: E402  (the checker)
from tests.gpu_util import assert_close, assert_step_close, batch_mean_update, round_f32, toy_problem  # noqa: E402

NAMES = ("lt", "wd", "M")


def main():
    n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    for s in range(seed0, seed0 + n_cfg):
        rng = np.random.default_rng(31_000 + s)
        dim = int(rng.choice([20, 32, 64, 64, 128, 128]))
        n_dist = int(rng.choice([3, 11, 40, 200, 255, 256, 700]))
        n_item = int(rng.choice([17, 64, 129, 400]))
        n_user = int(rng.integers(1, 60))
        len_max = int(rng.integers(2, 10))
        min_len = int(rng.integers(1, len_max + 1))
        T = toy_problem(9000 + s, n_user=n_user, n_item=n_item, n_dist=n_dist, dim=dim, len_max=len_max, min_len=min_len, hot=max(2, n_item // 3))
        P = round_f32(O.init_carnn_params(np.random.default_rng(s + 3000), n_item, n_dist, dim))
        cfg = dict(seed=s, dim=dim, n_dist=n_dist, n_item=n_item, n_user=n_user, len_max=len_max, min_len=min_len)
        if os.environ.get("FUZZ_VERBOSE"):
            print("config", cfg, flush=True)
        Pm, Qm, DPm, DQm, Mm = T["train"][0], T["train"][2], T["dist"][0], T["dist"][2], T["train"][1]
        k = int(rng.integers(1, n_user + 1))
        users = rng.permutation(n_user)[:k].astype(np.int32)
        news, touched, losses = [], [], []
        for u in users:
            Pn, los = O.carnn_step(P, Pm[u], Qm[u], DPm[u], DQm[u], Mm[u], 0.01, 0.001)
            news.append(Pn); losses.append(los)
            touched.append(dict(lt=np.unique(np.concatenate((Pm[u], Qm[u]))), wd=np.unique(np.concatenate((DPm[u], DQm[u])))))
        exp = batch_mean_update(P, news, touched, ("lt",), ("M",))
        acc = np.zeros_like(P["wd"]); cnt = np.zeros(P["wd"].shape[0])
        for Pn, tch in zip(news, touched):
            acc[tch["wd"]] += Pn["wd"][tch["wd"]] - P["wd"][tch["wd"]]; cnt[tch["wd"]] += 1
        exp["wd"] = P["wd"] + acc / np.maximum(cnt, 1)[:, None, None]
        m = poi_amd.models.OboCARNN(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                    n_dists=[n_dist, 0.2], n_in=dim, n_hidden=dim, init=P)
        got_los = m.train_batch(users)
        assert_close(got_los, losses, "losses %r" % cfg, rtol=2e-5)
        got = {kk: np.asarray(getattr(m, kk).get_value(), np.float64) for kk in NAMES}
        flat = lambda d: {kk: (np.asarray(v).reshape(np.asarray(v).shape[0], -1) if kk == "wd" else np.asarray(v)) for kk, v in d.items() if kk in NAMES}
        assert_step_close(flat(got), flat(exp), flat(P), NAMES, "%r" % cfg)
        if (s - seed0) % 10 == 9:
            print("config %d ok (dim %d, %d bins, %d POIs, %d of %d users, L <= %d)" % (s, dim, n_dist, n_item, k, n_user, len_max), flush=True)
    print("all %d configurations agree" % n_cfg)


if __name__ == "__main__":
    main()
