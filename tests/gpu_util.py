"""Helpers shared by the -m gpu parity tests: seeded toy problems in the reference's padded layout,
model construction from explicit float64 parameters, tolerance checks, and the oracle-side emulation
of the batch ("mean of the touching sequences' reference updates") semantics."""
import numpy as np

from oracle import poi_oracle as O

RTOL = 1e-5     # BASELINE.json north_star: weights within 1e-5 relative after one step (f32 vs f64)


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)) if a.size else 0.0


def assert_close(a, b, name, rtol=RTOL):
    e = rel_err(a, b)
    assert np.asarray(a).shape == np.asarray(b).shape, (name, np.asarray(a).shape, np.asarray(b).shape)
    assert e <= rtol, "%s: max|hip-oracle| / max|oracle| = %.3e > %.1e" % (name, e, rtol)
    return e


def toy_problem(seed, n_user=6, n_item=50, n_dist=11, dim=8, len_max=10, min_len=4, hot=8):
    """Padded tables in the reference layout with repeated POIs (duplicate scatter) and pad rows."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(min_len, len_max + 1, n_user)
    lens[0] = len_max                      # one full-length user: no padding at all
    P = np.full((n_user, len_max), n_item); Q = P.copy()
    DP = np.full((n_user, len_max), n_dist); DQ = DP.copy()
    M = np.zeros((n_user, len_max), int)
    for u, L in enumerate(lens):
        P[u, :L] = rng.integers(0, hot, L)                     # few distinct POIs -> repeats
        Q[u, :L] = rng.integers(hot // 2, n_item, L)
        DP[u, 1:L] = rng.integers(0, n_dist + 1, L - 1)
        DQ[u, 1:L] = rng.integers(0, n_dist + 1, L - 1)
        M[u, :L] = 1
    tes_p = rng.integers(0, n_item, (n_user, 1)); tes_q = rng.integers(0, n_item, (n_user, 1))
    tes_d = rng.integers(0, n_dist + 1, (n_user, 1)); tes_m = np.ones((n_user, 1), int)
    return dict(train=[P, M, Q], test=[tes_p, tes_m, tes_q], dist=[DP, tes_d, DQ], lens=lens,
                n_user=n_user, n_item=n_item, n_dist=n_dist, dim=dim, len_max=len_max)


def spatial_params(seed, T):
    rng = np.random.default_rng(seed + 1000)
    P = O.init_spatial_params(rng, T["n_item"], T["n_dist"], T["dim"])
    P["bi"] = rng.uniform(-0.2, 0.2, P["bi"].shape); P["bs"] = rng.uniform(-0.2, 0.2, P["bs"].shape)
    return round_f32(P)


def gru_params(seed, T):
    rng = np.random.default_rng(seed + 2000)
    P = O.init_gru_params(rng, T["n_item"], T["dim"])
    P["bi"] = rng.uniform(-0.2, 0.2, P["bi"].shape)
    return round_f32(P)


def round_f32(P):
    """The device tables are float32: the oracle runs in float64 FROM the float32-rounded inputs."""
    out = {}
    for k, v in P.items():
        out[k] = float(np.float32(v)) if np.isscalar(v) else np.asarray(v, np.float32).astype(np.float64)
    return out


def batch_mean_update(P, per_seq_new, touched, names_rows, names_dense):
    """Batch semantics (include/poi_hip.h): every row moves by the mean of the updates of the sequences
    that touch it; dense tensors by the mean over all sequences."""
    n = len(per_seq_new)
    N = dict(P)
    for name in names_dense:
        N[name] = np.asarray(P[name], np.float64) + sum(np.asarray(pn[name], np.float64) - np.asarray(P[name], np.float64) for pn in per_seq_new) / n
    for name in names_rows:
        base = P[name]
        acc = np.zeros_like(base); cnt = np.zeros(base.shape[0])
        for pn, tch in zip(per_seq_new, touched):
            rows = tch[name]
            acc[rows] += pn[name][rows] - base[rows]
            cnt[rows] += 1
        nz = cnt > 0
        new = base.copy()
        new[nz] = base[nz] + acc[nz] / cnt[nz, None]
        N[name] = new
    return N
