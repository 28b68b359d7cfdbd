#!/usr/bin/env python
"""Generate tests/golden/*.npz.  Run ONLY in the build container (needs /root/reference).

Part 1 imports the reference's own numpy modules (public/Valuate.py, public/Load_Data_by_length.py
- the half of the hot path that runs under Python 3) and records their outputs on seeded inputs:
these vectors PIN the oracle's numpy half.  Part 2 records the float64 oracle's outputs for the
Theano half on seeded toy inputs, so that a later edit of the oracle cannot drift silently
(those are NOT reference outputs; see oracle/poi_oracle.py header: parity unpinned).

Only arrays are written; no reference source text is stored.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import public.Valuate as V                      # noqa: E402  (reference, read-only)
import public.Load_Data_by_length as LD         # noqa: E402
from oracle import poi_oracle as O              # noqa: E402


def ref_topk(scores, k):
    # public/Valuate.py:132-146 without the Python-2-only np.array(zip(...))
    out = []
    for row in scores:
        idxs = V.fun_idxs_of_max_n_score(row, k)
        out.append(V.fun_sort_idxs_max_to_min((idxs, row)))
    return np.array(out)


def main():
    rng = np.random.default_rng(20260928)

    # ---- top-K (tie-free by construction: distinct float64 draws, min gap checked) -------------
    scores = rng.normal(size=(12, 777))
    srt = np.sort(scores, axis=1)
    assert np.min(np.diff(srt, axis=1)) > 1e-9
    np.savez(os.path.join(HERE, "topk.npz"), scores=scores, k=np.int64(20), ranks=ref_topk(scores, 20),
             ranks5=ref_topk(scores, 5))

    # ---- metrics --------------------------------------------------------------------------------
    n_u, k = 40, 20
    test_lst = rng.integers(0, 60, size=(n_u, 3))
    test_mask = np.array([[1] * m + [0] * (3 - m) for m in rng.integers(1, 4, n_u)])
    recom = np.array([rng.permutation(60)[:k] for _ in range(n_u)])
    zo = np.array([V.fun_hit_zero_one((t, r, m, [0])) for t, r, m in zip(test_lst, recom, test_mask)])
    mp = np.array([V.fun_evaluate_map((t, z, m, [0])) for t, z, m in zip(test_lst, zo, test_mask)])
    nd = np.array([V.fun_evaluate_ndcg((t, z, m, [0])) for t, z, m in zip(test_lst, zo, test_mask)])
    np.savez(os.path.join(HERE, "metrics.npz"), test_lst=test_lst, test_mask=test_mask, recom=recom,
             zero_one=zo, map=mp, ndcg=nd)

    # ---- Haversine bins -------------------------------------------------------------------------
    n = 4000
    lat1 = 40.0 + rng.uniform(0, 0.36, n); lon1 = -74.0 + rng.uniform(0, 0.47, n)
    lat2 = 40.0 + rng.uniform(0, 0.36, n); lon2 = -74.0 + rng.uniform(0, 0.47, n)
    lat2[:10], lon2[:10] = lat1[:10], lon1[:10]                  # identical points -> bin 0
    lat2[10:20] += 3.0                                           # far pairs -> clipped to dist_num
    bins = np.array([LD.cal_dis(a, b, c, d, 200, 200) for a, b, c, d in zip(lat1, lon1, lat2, lon2)])
    bins25 = np.array([LD.cal_dis(a, b, c, d, 25, 1520) for a, b, c, d in zip(lat1, lon1, lat2, lon2)])
    np.savez(os.path.join(HERE, "cal_dis.npz"), lat1=lat1, lon1=lon1, lat2=lat2, lon2=lon2,
             bins_dd200_B200=bins, bins_dd25_B1520=bins25)

    # ---- masks / negatives / negative bins / last-POI bins -------------------------------------
    N, U, B, dd = 60, 9, 200, 200
    coords = np.stack([40.0 + rng.uniform(0, 0.3, N), -74.0 + rng.uniform(0, 0.4, N)], 1)
    lens = [4, 7, 5, 12, 4, 9, 6, 12, 5]
    tra = [list(map(int, rng.integers(0, N, L))) for L in lens]
    tra_dist = []
    for seq in tra:
        d = [B] + [LD.cal_dis(coords[c][0], coords[c][1], coords[p][0], coords[p][1], dd, B)
                   for p, c in zip(seq[:-1], seq[1:])]
        tra_dist.append(d)
    pois_m, dist_m, msks = LD.fun_data_buys_masks(tra, tra_dist, [N], [B])
    random.seed(1234)
    negs = LD.fun_random_neg_masks_tra(N, pois_m)
    cordis = [list(c) for c in coords]
    dist_neg = LD.fun_compute_dist_neg(pois_m, msks, negs, cordis, dd, B)
    ulptai = LD.fun_compute_distance(pois_m, msks, cordis, dd, B)
    ragged = np.array([x for s in tra for x in s]); ragged_d = np.array([x for s in tra_dist for x in s])
    np.savez(os.path.join(HERE, "masks.npz"), n_item=N, n_dist=B, dd=dd, coords=coords, lens=np.array(lens),
             ragged_pois=ragged, ragged_dist=ragged_d, pois_m=np.array(pois_m), dist_m=np.array(dist_m),
             msks=np.array(msks), negs=np.array(negs), dist_neg=np.array(dist_neg), ulptai=np.array(ulptai))

    # ---- load_data on a small sequence file in the ETL's format (poidata/extract_whole_user_buys.py:81-90) ----------------
    # The file is written here (synthetic data; committed as a data fixture), the reference's own load_data reads it.  POI ids are
    # non-contiguous strings, one POI changes its coordinate between check-ins (bins come from the per-check-in coordinates, the table
    # keeps the last one), sequences are ragged; aliases are whatever this process's set order gives - the test compares modulo that.
    import io, contextlib
    from poi_amd import data as pdata
    Nf, Uf = 45, 14
    rng_f = np.random.default_rng(20260929)          # (its own stream: the vectors of Part 2 keep theirs)
    cf = np.stack([40.0 + rng_f.uniform(0, 0.3, Nf), -74.0 + rng_f.uniform(0, 0.4, Nf)], 1)
    names = rng_f.permutation(5000)[:Nf] + 17
    seqs_i = [list(map(int, rng_f.integers(0, Nf, L))) for L in rng_f.integers(5, 19, Uf)]
    per_checkin = [[tuple(cf[i]) for i in s] for s in seqs_i]
    moved = seqs_i[3][1]
    per_checkin[3][1] = (cf[moved][0] + 0.01, cf[moved][1] - 0.02)           # an earlier check-in at another coordinate: the last one wins (:56)
    fpath = os.path.join(HERE, "sequences_small.txt")
    pdata.write_sequence_file(fpath, [[int(names[i]) for i in s] for s in seqs_i], per_checkin, user_ids=list(range(100, 100 + Uf)))
    for split in (-1, -2):
        with contextlib.redirect_stdout(io.StringIO()):
            [(un, inum), pcs, (trp, tep), (trd, ted)] = LD.load_data(fpath, "test" if split == -1 else "valid", split, 200, 200)
        np.savez(os.path.join(HERE, "load_data_split%d.npz" % -split), user_num=un, item_num=inum, pois_cordis=np.array(pcs),
                 lens=np.array([len(x) for x in trp]), tra_pois=np.concatenate(trp), tes_pois=np.array(tep).reshape(-1),
                 tra_dist=np.concatenate(trd), tes_dist=np.array(ted).reshape(-1))

    # ---- Part 2: oracle-generated step vectors (drift guard; NOT reference outputs) -------------
    N, B, D, LM, L = 37, 11, 8, 10, 7
    P = O.init_spatial_params(rng, N, B, D)
    P['bi'] = rng.uniform(-0.2, 0.2, (3, D)); P['bs'] = rng.uniform(-0.2, 0.2, B + 1)
    p = np.full(LM, N); q = np.full(LM, N); dp = np.full(LM, B); dq = np.full(LM, B)
    p[:L] = rng.integers(0, 9, L); q[:L] = rng.integers(5, N, L)
    dp[1:L] = rng.integers(0, B + 1, L - 1); dq[1:L] = rng.integers(0, B + 1, L - 1)
    mask = np.array([1] * L + [0] * (LM - L))
    Pn, out = O.spatial_step(P, p, q, dp, dq, mask, 0.01, 0.001)
    sv = {("in_" + k): np.asarray(v) for k, v in P.items()}
    sv.update({("out_" + k): np.asarray(v) for k, v in Pn.items()})
    np.savez(os.path.join(HERE, "spatial_step.npz"), p=p, q=q, dp=dp, dq=dq, mask=mask, alpha=0.01, lam=0.001,
             los=out[0], sur=out[1], upq=out[2], ls=out[3], **sv)

    G = O.init_gru_params(rng, N, D); G['bi'] = rng.uniform(-0.2, 0.2, (3, D))
    Gn, gl = O.gru_step(G, p, q, mask, 0.01, 0.001)
    sv = {("in_" + k): np.asarray(v) for k, v in G.items()}
    sv.update({("out_" + k): np.asarray(v) for k, v in Gn.items()})
    np.savez(os.path.join(HERE, "gru_step.npz"), p=p, q=q, mask=mask, alpha=0.01, lam=0.001, loss=gl, **sv)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
