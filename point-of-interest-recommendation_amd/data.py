"""Host-side input contract of the hot path (the reference's public/Load_Data_by_length.py), vectorised
with numpy, plus the synthetic Foursquare/Gowalla-shaped generator used by bench.py and the tests.

Layouts follow the reference exactly where the model sees them:
  * padding id = n_item for POIs and dist_num for distance bins (Load_Data_by_length.py:115-124);
  * dist[0] = dist_num for every sequence (:77); bin = min(int(km*1000/dd), dist_num) (:38-39);
  * negatives uniform over [0, n_item) rejecting the user's own train POIs (:127-143);
  * negative distance bin t = bin(neg_t, pos_{t-1}) (:165-180).
On the device nothing is padded: sequences are CSR-packed (off / flat arrays); `to_padded()` rebuilds
the reference's nested-list tables for API compatibility.
"""
from __future__ import annotations

import dataclasses

import numpy as np

EARTH_D = 12742                      # Load_Data_by_length.py:30
DEG = 0.017453292519943295           # :31


def cal_dis_vec(lat1, lon1, lat2, lon2, dd, dist_num):
    """Vectorised public/Load_Data_by_length.py:24-42 (same float64 expression order)."""
    lat1, lon1, lat2, lon2 = (np.asarray(v, np.float64) for v in (lat1, lon1, lat2, lon2))
    a = (lat1 - lat2) * DEG
    b = (lon1 - lon2) * DEG
    c = (1.0 - np.cos(a)) / 2 + np.cos(lat1 * DEG) * np.cos(lat2 * DEG) * (1.0 - np.cos(b)) / 2
    dist = EARTH_D * np.arcsin(np.sqrt(c))
    interval = (dist * 1000 / dd).astype(np.int64)
    return np.minimum(interval, dist_num)


def bin_thresholds(dd, dist_num):
    """thr[k-1] = the smallest float64 c with int(12742*asin(sqrt(c))*1000/dd) >= k, k = 1..dist_num,
    found by bisection on the float64 bit pattern with the SAME libm calls as cal_dis
    (Load_Data_by_length.py:36-38).  Then  bin(c) = #{k : c >= thr[k-1]}  reproduces the reference's
    asin/sqrt/int chain exactly for every c, and the device needs no asin/sqrt at all."""
    import math
    import struct

    def f(c):
        return int(EARTH_D * math.asin(math.sqrt(c)) * 1000 / dd)

    def bits(x):
        return struct.unpack("<q", struct.pack("<d", x))[0]

    def val(b):
        return struct.unpack("<d", struct.pack("<q", b))[0]

    out = np.empty(dist_num, np.float64)
    lo_b = 0
    for k in range(1, dist_num + 1):
        hi = math.sin(min(k * dd / 1000.0 / EARTH_D, math.pi / 2)) ** 2
        hi = min(hi * (1 + 1e-6) + 1e-300, 1.0)
        while f(hi) < k and hi < 1.0:
            hi = min(hi * 1.001, 1.0)
        lo, hb = lo_b, bits(hi)            # f(val(lo)) < k <= f(val(hb))  (lo = previous threshold - works since f is monotone)
        if f(val(lo)) >= k:
            out[k - 1] = val(lo)
            continue
        while hb - lo > 1:
            mid = (lo + hb) // 2
            if f(val(mid)) >= k:
                hb = mid
            else:
                lo = mid
        out[k - 1] = val(hb)
        lo_b = lo
    return out


def cos_lat(coords):
    """cos(lat * pi/180) per POI with the SCALAR libm cos the reference calls (cal_dis :35) - always: numpy's vectorised float64 cos
    is a SIMD routine that is not correctly rounded and may differ from libm in the last bit on rare inputs, and one ulp of cos(lat)
    can flip a distance bin at a threshold.  ~0.15 us per POI (1.5 s at 10 M POIs, once per data set)."""
    import math
    lat = np.asarray(coords)[:, 0].astype(np.float64) * DEG
    return np.fromiter(map(math.cos, lat.tolist()), np.float64, count=len(lat))


def padded_to_csr(rows, lens):
    """Nested (U, LM) table + valid lengths -> (off int32 (U+1), flat int32)."""
    rows = np.asarray(rows)
    lens = np.asarray(lens, np.int64)
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    mask = np.arange(rows.shape[1])[None, :] < lens[:, None]
    return off.astype(np.int32), np.ascontiguousarray(rows[mask], dtype=np.int32)


def csr_to_padded(off, flat, pad, len_max=None):
    off = np.asarray(off, np.int64)
    lens = np.diff(off)
    lm = int(lens.max()) if len_max is None else int(len_max)
    out = np.full((len(lens), lm), pad, np.int32)
    mask = np.arange(lm)[None, :] < lens[:, None]
    out[mask] = flat
    return out


def sample_negatives(rng, n_item, off, pos_flat, exclude_off=None, exclude_flat=None):
    """One uniform negative per position, rejecting the user's own items
    (fun_random_neg_masks_tra :127-143; with exclude_* also the test items, _tes :146-162)."""
    off = np.asarray(off, np.int64)
    lens = np.diff(off)
    user_of = np.repeat(np.arange(len(lens)), lens)
    stride = np.int64(n_item) + 1
    own = user_of * stride + np.asarray(pos_flat, np.int64)
    if exclude_flat is not None:
        eoff = np.asarray(exclude_off, np.int64)
        euser = np.repeat(np.arange(len(eoff) - 1), np.diff(eoff))
        own = np.concatenate((own, euser * stride + np.asarray(exclude_flat, np.int64)))
    own = np.unique(own)
    return own, user_of, stride


def random_neg_tra(rng, n_item, off, p_flat):
    own, user_of, stride = sample_negatives(rng, n_item, off, p_flat)
    neg = rng.integers(0, n_item, size=len(p_flat), dtype=np.int64)
    bad = np.isin(user_of * stride + neg, own, assume_unique=False)
    while bad.any():
        neg[bad] = rng.integers(0, n_item, size=int(bad.sum()), dtype=np.int64)
        bad[bad] = np.isin(user_of[bad] * stride + neg[bad], own)
    return neg.astype(np.int32)


def random_neg_tes(rng, n_item, tra_off, tra_flat, tes_off, tes_flat):
    own, _, stride = sample_negatives(rng, n_item, tra_off, tra_flat, tes_off, tes_flat)
    tes_off = np.asarray(tes_off, np.int64)
    user_of = np.repeat(np.arange(len(tes_off) - 1), np.diff(tes_off))
    neg = rng.integers(0, n_item, size=len(tes_flat), dtype=np.int64)
    bad = np.isin(user_of * stride + neg, own)
    while bad.any():
        neg[bad] = rng.integers(0, n_item, size=int(bad.sum()), dtype=np.int64)
        bad[bad] = np.isin(user_of[bad] * stride + neg[bad], own)
    return neg.astype(np.int32)


def dist_neg_bins(off, p_flat, q_flat, coords, dd, dist_num):
    """fun_compute_dist_neg (:165-180): bin(neg_t, pos_{t-1}) for t >= 1, dist_num at t = 0."""
    off = np.asarray(off, np.int64)
    out = np.full(len(p_flat), dist_num, np.int32)
    first = np.zeros(len(p_flat), bool)
    first[off[:-1][np.diff(off) > 0]] = True
    idx = np.nonzero(~first)[0]
    prev = coords[np.asarray(p_flat)[idx - 1]]
    cur = coords[np.asarray(q_flat)[idx]]
    out[idx] = cal_dis_vec(cur[:, 0], cur[:, 1], prev[:, 0], prev[:, 1], dd, dist_num)
    return out


def dist_pos_bins(off, p_flat, coords, dd, dist_num):
    """load_data's train distance sequence (:70-78): bin(pos_t, pos_{t-1}), dist_num at t = 0."""
    off = np.asarray(off, np.int64)
    out = np.full(len(p_flat), dist_num, np.int32)
    first = np.zeros(len(p_flat), bool)
    first[off[:-1][np.diff(off) > 0]] = True
    idx = np.nonzero(~first)[0]
    prev = coords[np.asarray(p_flat)[idx - 1]]
    cur = coords[np.asarray(p_flat)[idx]]
    out[idx] = cal_dis_vec(cur[:, 0], cur[:, 1], prev[:, 0], prev[:, 1], dd, dist_num)
    return out


@dataclasses.dataclass
class CsrTables:
    """Unpadded form of the (train, test, dist) ctor arguments for a user range: what the device holds."""
    off: np.ndarray
    p: np.ndarray
    q: np.ndarray
    dp: np.ndarray
    dq: np.ndarray
    len_max: int               # padded row length the reference would have used (dataset-wide maximum)
    tes_p: np.ndarray          # (n, len_tes)
    tes_q: np.ndarray
    tes_mask: np.ndarray
    tes_dp: np.ndarray

    @property
    def n_user(self):
        return len(self.off) - 1


@dataclasses.dataclass
class PoiDataset:
    """CSR-packed check-in data in the shape Params.__init__ (prog_bpr_gru_spatial.py:49-100) builds."""
    n_user: int
    n_item: int
    dist_num: int
    dd: float
    coords: np.ndarray          # (n_item, 2) float64 lat, lon  (pois_cordis)
    off: np.ndarray             # (n_user+1,) int32
    tra_p: np.ndarray           # flat train POIs
    tra_dp: np.ndarray          # flat train distance bins
    tes_p: np.ndarray           # (n_user,) held-out POI (split = -1)
    tes_dp: np.ndarray          # (n_user,) its distance bin
    tra_q: np.ndarray = None    # flat negatives (resampled per epoch)
    tra_dq: np.ndarray = None   # flat negative distance bins
    tes_q: np.ndarray = None    # (n_user,) test negatives

    @property
    def lens(self):
        return np.diff(np.asarray(self.off, np.int64))

    @property
    def len_max(self):
        return int(self.lens.max())

    def resample_negatives(self, rng):
        """Per-epoch refresh, prog_bpr_gru_spatial.py:221-228."""
        self.tra_q = random_neg_tra(rng, self.n_item, self.off, self.tra_p)
        tes_off = np.arange(self.n_user + 1, dtype=np.int32)
        self.tes_q = random_neg_tes(rng, self.n_item, self.off, self.tra_p, tes_off, self.tes_p)
        self.tra_dq = dist_neg_bins(self.off, self.tra_p, self.tra_q, self.coords, self.dd, self.dist_num)

    def shard(self, lo=0, hi=None):
        """CsrTables of users [lo, hi) (offsets re-based); len_max stays the dataset-wide maximum, as
        in the reference where every user is padded to the longest sequence of the whole file."""
        hi = self.n_user if hi is None else hi
        off = np.asarray(self.off, np.int64)
        a, b = off[lo], off[hi]
        return CsrTables(off=(off[lo:hi + 1] - a).astype(np.int32), p=self.tra_p[a:b], q=self.tra_q[a:b],
                         dp=self.tra_dp[a:b], dq=self.tra_dq[a:b], len_max=self.len_max,
                         tes_p=self.tes_p[lo:hi].reshape(-1, 1), tes_q=self.tes_q[lo:hi].reshape(-1, 1),
                         tes_mask=np.ones((hi - lo, 1), np.int32), tes_dp=self.tes_dp[lo:hi].reshape(-1, 1))

    def last_pois(self):
        return np.asarray(self.tra_p)[np.asarray(self.off, np.int64)[1:] - 1]

    def to_padded(self):
        """The reference's nested tables: (train, test, dist) argument triples of the model ctors."""
        lm = self.len_max
        tra_buys = csr_to_padded(self.off, self.tra_p, self.n_item, lm)
        tra_neg = csr_to_padded(self.off, self.tra_q, self.n_item, lm)
        tra_dist = csr_to_padded(self.off, self.tra_dp, self.dist_num, lm)
        tra_dneg = csr_to_padded(self.off, self.tra_dq, self.dist_num, lm)
        tra_mask = (np.arange(lm)[None, :] < self.lens[:, None]).astype(np.int32)
        tes_buys = self.tes_p.reshape(-1, 1).astype(np.int32)
        tes_neg = self.tes_q.reshape(-1, 1).astype(np.int32)
        tes_dist = self.tes_dp.reshape(-1, 1).astype(np.int32)
        tes_mask = np.ones((self.n_user, 1), np.int32)
        return dict(train=[tra_buys, tra_mask, tra_neg], test=[tes_buys, tes_mask, tes_neg],
                    dist=[tra_dist, tes_dist, tra_dneg])


SHAPES = {
    # name: (n_item, n_user, max_len, dim)   BASELINE.json configs / BASELINE.md section 3
    "tiny": (300, 64, 12, 16),
    "foursquare": (10_000, 5_000, 20, 64),
    "gowalla": (100_000, 50_000, 50, 128),
    # one GPU's slice of BASELINE.json configs[4] (10 M POIs / 1 M users over 8 GPUs, dim 256, fp16 table): all POIs, 1/8 of the users
    "x1": (10_000_000, 125_000, 50, 256),
}


def _local_transitions(rng, coords, w, raw, local, n_nbr):
    """Check-in sequences with a learnable next-POI signal: with probability `local` the next POI is drawn among the
    n_nbr nearest neighbours of the current one (weights = global popularity), otherwise from the global Zipf law.
    Real check-in data is dominated by short hops (the premise of Distance2Pre's distance-interval head); i.i.d. Zipf
    draws carry nothing a sequence model could learn beyond popularity."""
    from scipy.spatial import cKDTree
    n_item, n_user = len(coords), len(raw)
    xy = np.stack([coords[:, 0] * 111.19, coords[:, 1] * 111.19 * np.cos(coords[:, 0].mean() * DEG)], 1)      # km, locally flat
    nbr = cKDTree(xy).query(xy, k=n_nbr + 1)[1][:, 1:]                                   # (n_item, n_nbr), self dropped
    cw = np.cumsum(w[nbr], axis=1); cw /= cw[:, -1:]
    cdf = np.cumsum(w / w.sum())
    lmax = int(raw.max())
    seq = np.empty((n_user, lmax), np.int64)
    seq[:, 0] = np.minimum(np.searchsorted(cdf, rng.random(n_user)), n_item - 1)
    for t in range(1, lmax):
        cur = seq[:, t - 1]
        loc = nbr[cur, np.minimum((cw[cur] < rng.random(n_user)[:, None]).sum(axis=1), n_nbr - 1)]
        glob = np.minimum(np.searchsorted(cdf, rng.random(n_user)), n_item - 1)
        seq[:, t] = np.where(rng.random(n_user) < local, loc, glob)
    return seq[np.arange(lmax)[None, :] < raw[:, None]]                                   # flat, user-major


def make_synthetic(n_user, n_item, max_len, seed, dd=200, ud_km=40, min_len=4, box_km=40.0, zipf=1.0, local=0.0, n_nbr=32):
    """Synthetic Foursquare/Gowalla-shaped data: lognormal sequence lengths clipped to
    [min_len, max_len] (+1 held-out check-in), Zipf POI popularity, POIs uniform in a ~box_km square.
    local = 0: every check-in is an independent Zipf draw (round-1 generator; nothing but popularity to learn);
    local > 0: that fraction of the transitions goes to one of the n_nbr nearest POIs of the current one."""
    rng = np.random.default_rng(seed)
    dist_num = int(ud_km * 1000 / dd)                    # prog_bpr_gru_spatial.py:81
    mu, sigma = np.log(max(max_len / 3.0, min_len)), 0.6
    lens = np.clip(np.rint(rng.lognormal(mu, sigma, n_user)), min_len, max_len).astype(np.int64)
    raw = lens + 1                                        # + the held-out last check-in (split = -1)
    w = 1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), zipf)
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(n_item)
    if local <= 0.0:
        ranks = np.minimum(np.searchsorted(cdf, rng.random(int(raw.sum()))), n_item - 1)
        pois = perm[ranks].astype(np.int32)
    lat = 40.0 + rng.random(n_item) * (box_km / 111.19)
    lon = -74.0 + rng.random(n_item) * (box_km / (111.19 * np.cos(40.0 * DEG)))
    coords = np.stack([lat, lon], 1)
    if local > 0.0:
        wp = np.empty(n_item); wp[perm] = w               # popularity weight of POI id i
        pois = _local_transitions(rng, coords, wp / wp.sum(), raw, float(local), int(n_nbr)).astype(np.int32)
    roff = np.zeros(n_user + 1, np.int64)
    np.cumsum(raw, out=roff[1:])
    rdist = dist_pos_bins(roff, pois, coords, dd, dist_num)
    is_last = np.zeros(len(pois), bool)
    is_last[roff[1:] - 1] = True
    off = np.zeros(n_user + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    ds = PoiDataset(n_user=n_user, n_item=n_item, dist_num=dist_num, dd=float(dd), coords=coords,
                    off=off.astype(np.int32), tra_p=pois[~is_last], tra_dp=rdist[~is_last],
                    tes_p=pois[is_last], tes_dp=rdist[is_last])
    ds.resample_negatives(rng)
    return ds


def load_sequence_file(path, split=-1, dd=200, dist_num=200, seed=0, return_aliases=False):
    """The reference's `load_data` (public/Load_Data_by_length.py:45-112) on its own sequence file - the space-separated table the ETL writes
    (poidata/extract_whole_user_buys.py:81-90: columns check_times pois_different u_id u_pois u_times u_coordinates; `u_pois` = ids joined by
    '/', `u_coordinates` = 'lat,lon' pairs joined by '/') - straight into the CSR PoiDataset the generator returns.  Same rules:
      * train = upois[:split], held-out = upois[split] (split -1: test mode, -2: valid mode; :64-65);
      * distance bin of check-in i = cal_dis(check-in i, check-in i-1) from the PER-CHECK-IN coordinates, dist_num at i = 0 (:68-74);
      * a POI's coordinate in `coords` is the one of its LAST occurrence in file order (dict(zip(...)), :56);
      * n_item = number of distinct ids in the WHOLE file, held-out check-ins included (:57,98).
    Aliases: the reference numbers the POIs in the iteration order of a Python `set` of strings (:98-99) - arbitrary, and different from run to
    run under hash randomisation.  Here: order of first appearance in the file - a relabelling of the same data (tests/test_host_cpu.py checks
    equality with the reference's output modulo that relabelling).  Negatives are drawn as the driver does right after loading
    (prog_bpr_gru_spatial.py:88-91) from `seed`."""
    import pandas as pd
    tab = pd.read_csv(path, sep=" ")                                                      # :52
    seqs = [str(s).split("/") for s in tab["u_pois"]]
    cods = [[tuple(float(v) for v in c.split(",")) for c in str(s).split("/")] for s in tab["u_coordinates"]]
    alias, coord_of = {}, {}
    for upois, ucods in zip(seqs, cods):
        if len(upois) != len(ucods):
            raise ValueError("%s: a user with %d POIs and %d coordinates" % (path, len(upois), len(ucods)))
        if len(upois) < -split:
            raise IndexError("%s: a sequence of %d check-ins cannot be split at %d" % (path, len(upois), split))      # (the reference: IndexError at :65)
        for s_, c_ in zip(upois, ucods):
            if s_ not in alias:
                alias[s_] = len(alias)
            coord_of[s_] = c_
    n_user, n_item = len(seqs), len(alias)
    coords = np.empty((n_item, 2), np.float64)
    for s_, a_ in alias.items():
        coords[a_] = coord_of[s_]
    ids = [np.fromiter((alias[s_] for s_ in upois), np.int32, count=len(upois)) for upois in seqs]
    dist = []
    for ucods in cods:                                                                    # :68-74
        c = np.asarray(ucods, np.float64)
        d = np.full(len(c), dist_num, np.int64)
        if len(c) > 1:
            d[1:] = cal_dis_vec(c[1:, 0], c[1:, 1], c[:-1, 0], c[:-1, 1], dd, dist_num)
        dist.append(d.astype(np.int32))
    lens = np.array([len(x[:split]) for x in ids], np.int64)
    off = np.zeros(n_user + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    ds = PoiDataset(n_user=n_user, n_item=n_item, dist_num=int(dist_num), dd=float(dd), coords=coords, off=off.astype(np.int32),
                    tra_p=np.concatenate([x[:split] for x in ids]).astype(np.int32), tra_dp=np.concatenate([d[:split] for d in dist]).astype(np.int32),
                    tes_p=np.array([x[split] for x in ids], np.int32), tes_dp=np.array([d[split] for d in dist], np.int32))
    ds.resample_negatives(np.random.default_rng(seed))
    return (ds, alias) if return_aliases else ds


def write_sequence_file(path, seqs, coords, user_ids=None):
    """Write check-in sequences in the ETL's format (poidata/extract_whole_user_buys.py:81-90) - the inverse of load_sequence_file, for tests and
    for handing synthetic data to the reference driver.  seqs: list of POI-id lists; coords: (n_item, 2) or a list of per-check-in lists."""
    import pandas as pd
    rows = []
    for k, s in enumerate(seqs):
        cc = coords[k] if isinstance(coords, list) else [coords[i] for i in s]
        rows.append((len(s), "%0.2f" % (1.0 * len(set(s)) / len(s)), user_ids[k] if user_ids is not None else k, "/".join(str(i) for i in s),
                     "/".join(str(t) for t in range(len(s))), "/".join("%r,%r" % (float(c[0]), float(c[1])) for c in cc)))
    cols = ["check_times", "pois_different", "u_id", "u_pois", "u_times", "u_coordinates"]
    pd.DataFrame(rows, columns=cols).to_csv(path, sep=" ", index=False, columns=cols)


def shard_users(n_user, world_size, rank, lens=None):
    """Contiguous user shard [lo, hi) of rank `rank`; with `lens`, boundaries balance the number of
    check-ins (GRU steps) rather than the number of users."""
    if lens is None:
        per = (n_user + world_size - 1) // world_size
        lo = min(rank * per, n_user)
        return lo, min(lo + per, n_user)
    c = np.concatenate(([0], np.cumsum(np.asarray(lens, np.int64))))
    tot = c[-1]
    bounds = [int(np.searchsorted(c, tot * r / world_size, side="left")) for r in range(world_size)] + [n_user]
    bounds[0] = 0
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds[rank], bounds[rank + 1]
