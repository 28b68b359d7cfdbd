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
