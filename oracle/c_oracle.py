"""ctypes access to oracle/_build/libpoi_oracle.so (plain-C float64 restatement).  TEST / BASELINE
INFRASTRUCTURE ONLY - never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libpoi_oracle.so")
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "poi_oracle_c.c")):
        subprocess.run(["make", "-s", "-C", HERE], check=True)
    return ctypes.CDLL(LIB)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def spatial_epoch(P, off, p, q, dp, dq, order, len_max, alpha, lam):
    """Sequential seq_train over `order`; P (dict of float64 arrays) is updated IN PLACE.
    Returns (n, 5) rows [los, sur, upq, ls0, ls1]."""
    lib = load()
    for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "loss_weight"):
        P[k] = np.ascontiguousarray(P[k], np.float64)
    wd = np.array([float(P["wd"])], np.float64)
    off, p, q, dp, dq, order = (np.ascontiguousarray(v, np.int32) for v in (off, p, q, dp, dq, order))
    out = np.zeros((len(order), 5))
    lib.poi_oracle_spatial_epoch(_d(P["lt"]), _d(P["di"]), _d(P["ui"]), _d(P["wh"]), _d(P["bi"]), _d(P["vs"]), _d(P["bs"]),
                                 _d(wd), _d(P["loss_weight"]), ctypes.c_int(P["lt"].shape[0] - 1), ctypes.c_int(P["di"].shape[0] - 1),
                                 ctypes.c_int(P["lt"].shape[1]), _i(off), _i(p), _i(q), _i(dp), _i(dq), _i(order),
                                 ctypes.c_int(len(order)), ctypes.c_int(int(len_max)), ctypes.c_double(alpha), ctypes.c_double(lam), _d(out))
    P["wd"] = float(wd[0])
    return out


def score_topk(users, items, k):
    lib = load()
    users = np.ascontiguousarray(users, np.float64); items = np.ascontiguousarray(items, np.float64)
    n, D = users.shape
    out = np.zeros((n, k), np.int32)
    lib.poi_oracle_score_topk(_d(users), _d(items), ctypes.c_int(n), ctypes.c_int(items.shape[0]), ctypes.c_int(D), ctypes.c_int(k), _i(out))
    return out


def spatial_batch_mean(P, off, p, q, dp, dq, ids, len_max, alpha, lam, threads=None, cap=1.0, absmass=False):
    """Oracle side of the batch rule (include/poi_hip.h): every sequence's reference update evaluated at P, each
    table row moved by the MEAN of the deltas of the sequences touching it, dense tensors by the mean over all
    sequences; with cap > 1, min(k, cap) / k times the SUM of the k touching sequences' deltas
    (poi_ctx_set_batch_cap).  The launch is cut into `threads` slices run concurrently (ctypes releases the GIL); the slices'
    float64 accumulators are added in slice order.  Returns (P_new, out (n, 5), touched) with touched = dict of the
    boolean row masks of lt / di.  P is not modified.  absmass=True: touched also carries "absmass" = {tensor name: the same
    combination of the sequences' |deltas|} - the scale a row's float32 summation noise is proportional to (tests/gpu_util)."""
    import concurrent.futures as cf
    import os as _os
    lib = load()
    A = {k: np.ascontiguousarray(P[k], np.float64) for k in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "loss_weight")}
    wd = np.array([float(P["wd"])], np.float64)
    off, p, q, dp, dq, ids = (np.ascontiguousarray(v, np.int32) for v in (off, p, q, dp, dq, ids))
    n = len(ids)
    n_item, n_dist, D = A["lt"].shape[0] - 1, A["di"].shape[0] - 1, A["lt"].shape[1]
    nd = A["ui"].size + A["wh"].size + A["bi"].size + A["vs"].size + A["bs"].size + 3
    T = max(1, min(threads or (_os.cpu_count() or 1), 32, (n + 63) // 64))
    bounds = [n * i // T for i in range(T + 1)]
    out = np.zeros((n, 5))

    def run(i):
        lo, hi = bounds[i], bounds[i + 1]
        acc_lt = np.zeros_like(A["lt"]); acc_di = np.zeros_like(A["di"])
        c_lt = np.zeros(n_item + 1, np.int32); c_di = np.zeros(n_dist + 1, np.int32)
        acc_d = np.zeros(nd)
        ab_lt = np.zeros_like(A["lt"]) if absmass else None
        ab_di = np.zeros_like(A["di"]) if absmass else None
        ab_d = np.zeros(nd) if absmass else None
        o = np.zeros((hi - lo, 5))
        sl = np.ascontiguousarray(ids[lo:hi])
        lib.poi_oracle_spatial_batch(_d(A["lt"]), _d(A["di"]), _d(A["ui"]), _d(A["wh"]), _d(A["bi"]), _d(A["vs"]), _d(A["bs"]),
                                     _d(wd), _d(A["loss_weight"]), ctypes.c_int(n_item), ctypes.c_int(n_dist), ctypes.c_int(D),
                                     _i(off), _i(p), _i(q), _i(dp), _i(dq), _i(sl), ctypes.c_int(hi - lo), ctypes.c_int(int(len_max)),
                                     ctypes.c_double(alpha), ctypes.c_double(lam), _d(acc_lt), _i(c_lt), _d(acc_di), _i(c_di), _d(acc_d), _d(o),
                                     _d(ab_lt) if absmass else None, _d(ab_di) if absmass else None, _d(ab_d) if absmass else None)
        return acc_lt, c_lt, acc_di, c_di, acc_d, o, ab_lt, ab_di, ab_d

    with cf.ThreadPoolExecutor(T) as ex:
        parts = list(ex.map(run, range(T)))
    acc_lt = sum(x[0] for x in parts); c_lt = sum(x[1].astype(np.int64) for x in parts)
    acc_di = sum(x[2] for x in parts); c_di = sum(x[3].astype(np.int64) for x in parts)
    acc_d = sum(x[4] for x in parts)
    for i, x in enumerate(parts):
        out[bounds[i]:bounds[i + 1]] = x[5]
    N = dict(P)
    sc = lambda c: np.minimum(np.maximum(c, 1), cap) / np.maximum(c, 1)
    N["lt"] = A["lt"] + acc_lt * sc(c_lt)[:, None]
    N["di"] = A["di"] + acc_di * sc(c_di)[:, None]
    o = 0
    for k in ("ui", "wh", "bi", "vs", "bs"):
        N[k] = A[k] + acc_d[o:o + A[k].size].reshape(A[k].shape) * (min(n, cap) / n)
        o += A[k].size
    N["wd"] = float(wd[0] + acc_d[o] * (min(n, cap) / n))
    N["loss_weight"] = A["loss_weight"] + acc_d[o + 1:o + 3] * (min(n, cap) / n)
    touched = dict(lt=c_lt > 0, di=c_di > 0)
    if absmass:
        ab_lt = sum(x[6] for x in parts); ab_di = sum(x[7] for x in parts); ab_d = sum(x[8] for x in parts)
        M = {"lt": ab_lt * sc(c_lt)[:, None], "di": ab_di * sc(c_di)[:, None]}
        o = 0
        for k in ("ui", "wh", "bi", "vs", "bs"):
            M[k] = ab_d[o:o + A[k].size].reshape(A[k].shape) * (min(n, cap) / n)
            o += A[k].size
        M["wd"] = float(ab_d[o] * (min(n, cap) / n))
        M["loss_weight"] = ab_d[o + 1:o + 3] * (min(n, cap) / n)
        touched["absmass"] = M
    return N, out, touched
