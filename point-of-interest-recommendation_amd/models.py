"""Host-side mirror of the reference's model-object protocol on top of libpoi_hip.so.

Class / method names, argument meaning and return shapes follow the Theano classes the drivers call:
    OboSpatialGru  public/GRU_Spatial.py:42-292      OboGru  public/GRU.py:301-389 (+ GruBasic :32-205)
    OboBpr         public/BPR.py:191-241 (+ MfBasic :28-134)
so that prog_bpr_gru_spatial.py's epoch loop and public/Valuate.py's evaluator run against them
unchanged in shape.  State lives in torch ROCm tensors (device-memory containers only); every piece of
arithmetic is a HIP kernel reached through ctypes.  There is no CPU path: without the library or a GPU
the constructors raise.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .data import CsrTables, bin_thresholds, cos_lat, padded_to_csr


import os

_CHECK_DEVICE_IDS = os.environ.get("POI_CHECK_IDS", "0") == "1"


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class Shared:
    """Stand-in for a Theano shared variable: .get_value() / .set_value() on a device tensor."""

    def __init__(self, tensor, scalar=False, pad=None, unpad=None):
        self.t = tensor
        self.scalar = scalar
        self.pad, self.unpad = pad, unpad      # logical <-> stored layout (models whose dim is padded to a tile-engine dim)

    def get_value(self, borrow=False):
        a = self.t.detach().cpu().numpy()
        if self.unpad is not None:
            a = np.ascontiguousarray(self.unpad(a))
        return a.reshape(()).copy() if self.scalar else a

    def set_value(self, value, borrow=False):
        v = np.asarray(value, dtype=np.float64)
        if self.pad is not None:
            v = self.pad(v)
        v = torch.as_tensor(v, dtype=self.t.dtype).reshape(self.t.shape)
        self.t.copy_(v.to(self.t.device))


class _L2:
    """model.l2 - an object with .eval() (public/GRU_Spatial.py:83-88)."""

    def __init__(self, model, names):
        self.model, self.names = model, names

    def eval(self):
        m = self.model
        acc = torch.zeros(1, dtype=torch.float64, device=m.device)
        for n in self.names:
            t = getattr(m, n).t
            m.ctx.check(m.lib.poi_sumsq(m.ctx.handle, _ptr(t), t.numel(), _ptr(acc), m._stream()))
        return 0.5 * m.alpha_lambda[1] * float(acc.item())


class _Base:
    def _setup(self, device, alpha_lambda):
        if not torch.cuda.is_available():
            raise _lib.PoiError("no ROCm device visible: the next-POI hot path has no CPU implementation")
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.ctx = _lib.context(idx)
        self.lib = self.ctx.lib
        self.alpha_lambda = [float(alpha_lambda[0]), float(alpha_lambda[1])]

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):          # (a device tensor is taken as it is: 10 M-row tables are generated on the device)
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).to(self.device).contiguous()

    def _ids(self, idxs):
        """(int32 device tensor of user ids, lo) - lo is the first id when the ids are a contiguous
        ascending range (then the tensor is a zero-copy view of a resident arange), else None.
        Host ids (scalars, lists, numpy) are range-checked and raise IndexError like the reference's Theano
        gather - the kernels would read off[u] out of bounds.  A DEVICE tensor is trusted (checking it
        costs a host sync per launch); set POI_CHECK_IDS=1 to check those too."""
        if isinstance(idxs, torch.Tensor):
            t = idxs.to(device=self.device, dtype=torch.int32).contiguous()
            if _CHECK_DEVICE_IDS and t.numel():
                lo_, hi_ = int(t.min().item()), int(t.max().item())
                if lo_ < 0 or hi_ >= self.n_user:
                    raise IndexError("user ids must lie in [0, %d) (found %d..%d)" % (self.n_user, lo_, hi_))
            return t, None
        a = np.atleast_1d(np.asarray(idxs)).astype(np.int64)
        if a.size and (a.min() < 0 or a.max() >= self.n_user):
            raise IndexError("user ids must lie in [0, %d) (found %d..%d)" % (self.n_user, int(a.min()), int(a.max())))
        if len(a) and np.all(np.diff(a) == 1):
            lo = int(a[0])
            return self._arange[lo:lo + len(a)], lo
        return torch.as_tensor(a.astype(np.int32)).to(self.device), None

    def _rows(self, table, ids, lo):
        n = ids.numel()
        return table[lo:lo + n] if lo is not None else table.index_select(0, ids.long()).contiguous()

    def _by_length(self, idxs):
        """(ids sorted by descending sequence length, out_row) for the recurrent kernels: a 16-sequence tile runs for
        its longest member, so homogeneous tiles halve the forward pass of a full evaluation.  out_row[k] = position of
        the k-th sorted id in the caller's order: poi_gru_predict writes its result there, so nothing is permuted
        afterwards.  The order is computed on the host from the host-side lengths (once per distinct id list: cached for
        the common "all users" / contiguous-range calls).  Device tensors and small launches are taken as they come."""
        if isinstance(idxs, torch.Tensor):
            return self._ids(idxs)[0], None
        a = np.atleast_1d(np.asarray(idxs)).astype(np.int64)
        ids, lo = self._ids(a)
        if len(a) < 64:
            return ids, None
        key = (int(a[0]), len(a)) if lo is not None else None
        cache = self.__dict__.setdefault("_order_cache", {})
        if key is not None and key in cache:
            return cache[key]
        order = np.argsort(-np.asarray(self._lens)[a], kind="stable").astype(np.int32)
        out = (torch.as_tensor(a[order].astype(np.int32)).to(self.device), torch.as_tensor(order).to(self.device))
        if key is not None and len(cache) < 64:
            cache[key] = out
        return out

    # ---- tables -------------------------------------------------------------------------------
    @staticmethod
    def _check_ids(name, ids, hi):
        """The reference gathers with Theano advanced indexing, which raises IndexError on an id outside the
        table; the kernels would read out of bounds - so ids are checked where they enter."""
        a = np.asarray(ids)
        if a.size and (a.min() < 0 or a.max() > hi):
            raise IndexError("%s: ids must lie in [0, %d] (found %d..%d)" % (name, hi, int(a.min()), int(a.max())))

    def _load_tables(self, train, test):
        self._csr = train if isinstance(train, CsrTables) else None
        if self._csr is not None:
            c = self._csr
            off = np.ascontiguousarray(c.off, np.int32)
            self._check_ids("train POIs", c.p, self.n_item); self._check_ids("train negatives", c.q, self.n_item)
            self._check_ids("test POIs", c.tes_p, self.n_item); self._check_ids("test negatives", c.tes_q, self.n_item)
            self._lens = np.diff(off.astype(np.int64))
            self.len_max, self.max_len = int(c.len_max), int(self._lens.max())
            self._off_host = off
            i32 = lambda v: torch.as_tensor(np.ascontiguousarray(v, dtype=np.int32)).to(self.device)
            self.off, self.p, self.q = i32(off), i32(c.p), i32(c.q)
            self.tes_buys_masks, self.tes_masks, self.tes_buys_neg_masks = i32(c.tes_p), i32(c.tes_mask), i32(c.tes_q)
            self._arange = torch.arange(self.n_user, dtype=torch.int32, device=self.device)
            return
        tra_buys_masks, tra_masks, tra_buys_neg_masks = train
        tes_buys_masks, tes_masks, tes_buys_neg_masks = test
        for nm, t in (("train POIs", tra_buys_masks), ("train negatives", tra_buys_neg_masks), ("test POIs", tes_buys_masks),
                      ("test negatives", tes_buys_neg_masks)):
            self._check_ids(nm, t, self.n_item)
        tra_masks = np.asarray(tra_masks)
        self._lens = tra_masks.sum(axis=1).astype(np.int64)
        self.len_max = int(tra_masks.shape[1])                 # padded length LM of the reference tables
        self.max_len = int(self._lens.max())
        off, p = padded_to_csr(tra_buys_masks, self._lens)
        _, q = padded_to_csr(tra_buys_neg_masks, self._lens)
        self._off_host = off
        self.off, self.p, self.q = (torch.as_tensor(v).to(self.device) for v in (off, p, q))
        self.tes_buys_masks = self._dev(tes_buys_masks, torch.int32)
        self.tes_masks = self._dev(tes_masks, torch.int32)
        self.tes_buys_neg_masks = self._dev(tes_buys_neg_masks, torch.int32)
        self._arange = torch.arange(self.n_user, dtype=torch.int32, device=self.device)

    def update_neg_masks(self, tra_buys_neg_masks, tes_buys_neg_masks):
        """public/GRU.py:79-82 / public/BPR.py:61-64 - new negatives every epoch."""
        self._check_ids("train negatives", tra_buys_neg_masks, self.n_item); self._check_ids("test negatives", tes_buys_neg_masks, self.n_item)
        _, q = padded_to_csr(tra_buys_neg_masks, self._lens)
        self.q = torch.as_tensor(q).to(self.device)
        self.tes_buys_neg_masks = self._dev(tes_buys_neg_masks, torch.int32)

    def set_negatives_csr(self, q_flat, tes_q=None, dq_flat=None):
        """CSR form of update_neg_masks / s_update_neg_masks (no padded tables built)."""
        self.q = torch.as_tensor(np.ascontiguousarray(q_flat, dtype=np.int32)).to(self.device)
        if tes_q is not None:
            self.tes_buys_neg_masks = self._dev(np.asarray(tes_q).reshape(self.n_user, -1), torch.int32)
        if dq_flat is not None:
            self.dq = torch.as_tensor(np.ascontiguousarray(dq_flat, dtype=np.int32)).to(self.device)

    def resample_negatives_device(self, seed):
        """Per-epoch negative refresh entirely on the device (prog_bpr_gru_spatial.py:221-228): new train /
        test negatives (Load_Data_by_length.py:127-162) and, for the spatial model, their distance bins
        (:165-180).  No host work, no upload; reproducible for a given seed."""
        q = torch.empty_like(self.p)
        tq = torch.empty_like(self.tes_buys_masks)
        self.ctx.check(self.lib.poi_sample_negatives(self.ctx.handle, _ptr(self.off), _ptr(self.p), self.n_user, self.n_item,
                                                     _ptr(self.tes_buys_masks), _ptr(self.tes_masks), self.tes_masks.shape[1],
                                                     int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(q), _ptr(tq), self._stream()))
        self.q, self.tes_buys_neg_masks = q, tq
        if getattr(self, "spatial", False):
            if self.coords is None:
                raise _lib.PoiError("resample_negatives_device on the spatial model needs coords= at construction")
            dq = torch.empty_like(self.p)
            self.ctx.check(self.lib.poi_neg_dist_bins(self.ctx.handle, _ptr(self.off), _ptr(self.p), _ptr(self.q), self.n_user,
                                                      _ptr(self.coords), _ptr(self._cphi), _ptr(self._binthr), self.n_dist,
                                                      self.dd * 1000.0, _ptr(dq), self._stream()))
            self.dq = dq

    # ---- snapshots ----------------------------------------------------------------------------
    def update_trained_items(self):
        """public/GRU.py:84-87: eval sees a snapshot of lt, not the live table."""
        self.trained_items.t.copy_(self.lt.t)

    def compute_sub_auc_preference(self, start_end):
        """public/GRU.py:98-110 -> bool ndarray (n, len_tes)."""
        ids, lo = self._ids(start_end)
        n = ids.numel()
        ln = self.tes_masks.shape[1]
        users = self._rows(self.trained_users.t, ids, lo)
        tp, tq, tm = (self._rows(t, ids, lo) for t in (self.tes_buys_masks, self.tes_buys_neg_masks, self.tes_masks))
        out = torch.empty((n, ln), dtype=torch.uint8, device=self.device)
        self.ctx.check(self.lib.poi_auc_preference(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), n, self.kdim,
                                                   _ptr(tp), _ptr(tq), _ptr(tm), ln, _ptr(out), self._stream()))
        return out.cpu().numpy().astype(bool)

    def _users_rows(self, start_end):
        ids, lo = self._ids(start_end)
        return ids, self._rows(self.trained_users.t, ids, lo), lo

    def _prob_rows(self, ids, lo):
        return None, None

    def compute_sub_all_scores(self, start_end):
        """public/GRU.py:93-96 (spatial: public/GRU_Spatial.py:117-125) -> ndarray (n, n_item)."""
        return self.compute_sub_all_scores_device(start_end).cpu().numpy()

    def compute_sub_all_scores_device(self, start_end):
        ids, users, lo = self._users_rows(start_end)
        n = ids.numel()
        wd, prob = self._prob_rows(ids, lo)
        out = torch.empty((n, self.n_item), dtype=torch.float32, device=self.device)
        self.ctx.check(self.lib.poi_score_all(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), n, self.n_item, self.kdim,
                                              _ptr(wd), _ptr(prob), _ptr(out), self._stream()))
        return out

    def compute_sub_topk(self, start_end, k, return_scores=False):
        """Fused a8+a9: (n, k) int32 indices sorted by descending score (public/Valuate.py:132-146)
        without materialising the (n, n_item) score matrix."""
        if k > 32:
            return self._topk_from_scores(start_end, k, return_scores)
        ids, users, lo = self._users_rows(start_end)
        n = ids.numel()
        wd, prob = self._prob_rows(ids, lo)
        idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
        sc = torch.empty((n, k), dtype=torch.float32, device=self.device) if return_scores else None
        seed = self._seed_begin(lo, n, k)
        self.ctx.check(self.lib.poi_score_topk(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), n, self.n_item, self.kdim,
                                               _ptr(wd), _ptr(prob), int(k), _ptr(idx), _ptr(sc), self._stream()))
        self._seed_end(seed, idx)
        return (idx, sc) if return_scores else idx

    # Seeded top-K (include/poi_hip.h, poi_ctx_set_topk_seed): the previous evaluation's top-K ids of a contiguous user range seed the
    # thresholds of the next one - exact whatever they hold, several times fewer candidate insertions.  `topk_seeding = False` disables.
    topk_seeding = True

    def _seed_begin(self, lo, n, k):
        if not self.topk_seeding or lo is None or k > 32:
            return None
        seeds = self.__dict__.setdefault("_topk_seeds", {})
        t = seeds.get(int(k))
        if t is None:
            t = seeds[int(k)] = (torch.full((self.n_user, int(k)), -1, dtype=torch.int32, device=self.device), np.zeros(self.n_user, bool))
        rows = t[0][lo:lo + n]
        if t[1][lo:lo + n].all():                 # seed only with lists a previous evaluation wrote (the first one runs unseeded)
            self.ctx.set_topk_seed(rows, k)
        return rows, t[1], lo, n

    def reset_topk_seeds(self):
        """Forget the lists of earlier evaluations (the next one runs unseeded).  (Measured and not kept, round 4: a static prior - the 20 POIs
        nearest to each user's last check-in - as the seed of the first evaluation.  Under a trained model only 24 % of the final top-20 are
        among them: the seed's K-th best is far below the final one, 8600 survivors per user, every tile overflows - 47 ms against 11 ms unseeded.)"""
        self.__dict__.pop("_topk_seeds", None)

    def _seed_end(self, seed, idx):
        if seed is not None:
            rows, filled, lo, n = seed
            rows.copy_(idx)
            filled[lo:lo + n] = True


    def _topk_from_scores(self, start_end, k, return_scores=False):
        """Cut-offs beyond the fused kernels' k <= 32 (e.g. at_nums = [5, 10, 15, 20, 30, 50], public/Valuate.py:126):
        explicit score rows (compute_sub_all_scores_device, <= 1 GiB at a time) + poi_topk (k <= 64)."""
        if k > 64:
            raise _lib.PoiError("top-K supports k <= 64 (got %d)" % k)
        a = start_end if isinstance(start_end, torch.Tensor) else np.atleast_1d(np.asarray(start_end))
        n = len(a)
        idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
        sc = torch.empty((n, k), dtype=torch.float32, device=self.device) if return_scores else None
        step = max(1, min(n, (1 << 28) // max(self.n_item, 1)))
        for o in range(0, n, step):
            full = self.compute_sub_all_scores_device(a[o:o + step])
            self.ctx.check(self.lib.poi_topk(self.ctx.handle, _ptr(full), full.shape[0], self.n_item, int(k), ctypes.c_void_p(idx.data_ptr() + 4 * o * k),
                                             ctypes.c_void_p(sc.data_ptr() + 4 * o * k) if sc is not None else None, self._stream()))
        return (idx, sc) if return_scores else idx


# =================================================================================================
class GruBasic(_Base):
    """public/GRU.py:32-205."""

    spatial = False

    _pad_ok = True          # dims other than 64 / 128 / 256 may be zero-padded to the next tile-engine dim (not CA-RNN: sigmoid(0) != 0)

    def __init__(self, train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, device="cuda:0", init=None, seed=None, table_dtype="f32",
                 pad_dim=True):
        """pad_dim (default on): a model whose dim is not 64 / 128 / 256 (the reference's own configs use 20 and 32) is STORED with its
        hidden / embedding width zero-padded to the next of those, so that it trains on the MFMA tile engine instead of the
        per-sequence engine (dim 20: 1.5 M -> ~9 M sequences/s).  Exact: a padded hidden unit has zero weights and bias, so its gates
        are sigmoid(0), its candidate tanh(0) = 0 and its state stays 0; every gradient that touches a padded row or column is a
        product with one of those zeros, and the L2 decay of a zero is zero - the padding stays exactly zero and the other entries
        see only additional + 0.0 terms.  get_value / set_value / load_params / predict / checkpoints use the logical shapes.
        table_dtype="f16": the POI table `lt` and its evaluation snapshot are STORED as IEEE half (config X of BASELINE.json:
        "fp16 embeddings"); all arithmetic stays float32 - rows are converted when gathered and rounded to nearest-even when
        written back.  Tile engine only (dim 64 / 128 / 256).  NOTE: an update smaller than half an fp16 ulp of the element
        (2.4e-4 at 0.5) is lost to the rounding; with alpha 0.01 that is most single-sequence updates - use launches with a
        batch cap, whose summed updates are larger."""
        if n_in != n_hidden:
            raise ValueError("the reference drivers always pass n_in == n_hidden (prog_bpr_gru_spatial.py:138-139)")
        if table_dtype not in ("f32", "f16"):
            raise ValueError("table_dtype must be 'f32' or 'f16'")
        self.table_dtype = table_dtype
        self._setup(device, alpha_lambda)
        self.n_user, self.n_item, self.dim = int(n_user), int(n_item), int(n_in)
        D = self.dim
        self.kdim = D           # the dim the kernels see
        if pad_dim and self._pad_ok and D not in (64, 128, 256) and D < 256:
            self.kdim = 64 if D < 64 else 128 if D < 128 else 256
        self._load_tables(train, test)
        rng = np.random.default_rng(seed) if seed is not None else np.random
        u = lambda *s: rng.uniform(-0.5, 0.5, s)
        init = init or {}
        g = lambda k, v: (init[k] if isinstance(init[k], torch.Tensor) else np.asarray(init[k], np.float64)) if k in init else v()
        tdt = torch.float16 if table_dtype == "f16" else torch.float32
        sh = self._shared
        big = (n_item + 1) * D > (1 << 28) and self.kdim == D
        if big:
            # tables of hundreds of millions of elements (config X: 10 M x 256) are drawn ON the device: the host draw is 20 GB of float64
            gen = torch.Generator(device=self.device).manual_seed(0 if seed is None else int(seed))
            u_tab = lambda rows: (torch.rand((rows, D), generator=gen, device=self.device, dtype=torch.float32) - 0.5).to(tdt)
        else:
            u_tab = lambda rows: u(rows, D)
        self.lt = sh(g("lt", lambda: u_tab(n_item + 1)), "cols", tdt)                      # GRU.py:60
        self.ui = sh(g("ui", lambda: u(3, D, self._xw())), "ui")                           # :61 / GRU_Spatial.py:51
        self.wh = sh(g("wh", lambda: u(3, D, D)), "sq")                                    # :62
        self.bi = sh(g("bi", lambda: np.zeros((3, D))), "cols")                            # :64
        self.h0 = Shared(torch.zeros(self.kdim, dtype=torch.float32, device=self.device), unpad=(lambda a: a[:D]) if self.kdim != D else None)   # :63 never trained
        self.trained_items = sh(u_tab(n_item + 1), "cols", tdt)                            # :71
        self.trained_users = sh(u(n_user, D), "cols")                                      # :72
        if table_dtype == "f16":
            self.ctx.register_f16(self.lt.t); self.ctx.register_f16(self.trained_items.t)

    def __del__(self):
        try:
            if getattr(self, "table_dtype", "f32") == "f16":
                self.ctx.unregister_f16(self.lt.t); self.ctx.unregister_f16(self.trained_items.t)
        except Exception:
            pass

    def _xw(self):
        return self.dim

    def _shared(self, value, kind, dtype=torch.float32):
        """Device tensor of a parameter in the STORED layout (width kdim) with logical get / set.  kind: "cols" - last axis D -> kdim;
        "sq" - (3, D, D) -> (3, kdim, kdim); "ui" - (3, D, xw) with xw = D or 2 D column blocks, each block padded on its own."""
        D, K = self.dim, self.kdim
        if K == D:
            return Shared(self._dev(value, dtype))
        if kind == "cols":
            pad = lambda a: np.concatenate((a, np.zeros(a.shape[:-1] + (K - D,))), axis=-1)
            unpad = lambda a: a[..., :D]
        elif kind == "sq":
            def pad(a):
                o = np.zeros((3, K, K)); o[:, :D, :D] = a; return o
            unpad = lambda a: a[:, :D, :D]
        else:
            nb = self._xw() // D
            def pad(a):
                o = np.zeros((3, K, nb * K))
                for b in range(nb):
                    o[:, :D, b * K:b * K + D] = a[:, :, b * D:(b + 1) * D]
                return o
            unpad = lambda a: np.concatenate([a[:, :D, b * K:b * K + D] for b in range(nb)], axis=2)
        return Shared(self._dev(pad(np.asarray(value, np.float64)), dtype), pad=pad, unpad=unpad)

    def _pad_cols(self, t):
        """(n, D) or (n, kdim) device tensor -> (n, kdim)."""
        if t.shape[-1] == self.kdim:
            return t
        o = torch.zeros(t.shape[:-1] + (self.kdim,), dtype=t.dtype, device=t.device)
        o[..., :self.dim] = t
        return o

    def update_trained_users(self, all_hus):
        """public/GRU.py:89-91 ((n_user, D); the padded rows predict_device returns are taken as they are)."""
        t = all_hus if isinstance(all_hus, torch.Tensor) else self._dev(np.asarray(all_hus, np.float64))
        t = t.to(self.device, torch.float32).reshape(self.n_user, -1)
        self.trained_users.t.copy_(self._pad_cols(t))

    # ---- ctypes views ---------------------------------------------------------------------------
    def _params(self, snapshot=False):
        P = _lib.GruParams()
        P.lt = (self.trained_items if snapshot else self.lt).t.data_ptr()
        P.ui, P.wh, P.bi = self.ui.t.data_ptr(), self.wh.t.data_ptr(), self.bi.t.data_ptr()
        P.di = P.vs = P.bs = P.wd = P.lw = None
        P.n_item, P.n_dist, P.dim = self.n_item, 0, self.kdim
        return P

    def _tables(self):
        T = _lib.SeqTables()
        T.off, T.p, T.q = self.off.data_ptr(), self.p.data_ptr(), self.q.data_ptr()
        T.dp = T.dq = None
        T.n_user, T.len_max, T.max_len = self.n_user, self.len_max, self.max_len
        return T

    def predict(self, idxs):
        """public/GRU.py:204-205 -> hts ndarray (n, D)."""
        return np.ascontiguousarray(self.predict_device(idxs)[:, :self.dim].cpu().numpy())

    def predict_device(self, idxs):
        """(n, kdim) device rows (kdim == dim unless the model is stored padded: the padding columns are zero)."""
        ids, out_row = self._by_length(idxs)
        n = ids.numel()
        hts = torch.empty((n, self.kdim), dtype=torch.float32, device=self.device)
        P, T = self._params(snapshot=True), self._tables()
        self.ctx.check(self.lib.poi_gru_predict(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), _ptr(out_row), n, _ptr(hts), None,
                                                self._stream()))
        return hts


class OboGru(GruBasic):
    """public/GRU.py:301-389 - plain GRU + BPR, one SGD step per user sequence."""

    def __init__(self, train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, **kw):
        super().__init__(train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, **kw)
        self.params = [self.ui, self.wh, self.bi]
        self.l2 = _L2(self, ["lt", "ui", "wh", "bi"])                                      # :304-308

    def train(self, idx):
        """seq_train(uidx) -> float (public/GRU.py:387-389)."""
        return float(self.train_batch(np.atleast_1d(idx))[0])

    def train_batch(self, idxs, sync=True):
        """Throughput mode: n sequences per launch, batch semantics of include/poi_hip.h."""
        ids, _ = self._ids(idxs)
        n = ids.numel()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        P, T = self._params(), self._tables()
        self.ctx.check(self.lib.poi_gru_step(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), n,
                                             self.alpha_lambda[0], self.alpha_lambda[1], _ptr(out), self._stream()))
        return out.cpu().numpy() if sync else out


class Gru(OboGru):
    """public/GRU.py:395-498 - the mini-batch GRU: `train(idxs)` takes a LIST of users and makes ONE SGD step on the cost
    -sum(loss) / batch + 0.5 lambda (every gathered row, ui, wh, bi) (:452-459).  Same kernels as OboGru with the launch as the
    mini-batch (poi_ctx_set_batch_cap(0), include/poi_hip.h): loss gradients averaged over the launch, L2 terms summed; predict /
    scoring / AUC are GruBasic's.  Returns -upq, the batch's summed loss (:473), like the reference."""

    def train(self, idxs):
        return float(np.sum(self.train_batch(idxs)))

    def train_batch(self, idxs, sync=True):
        prev = getattr(self.ctx, "batch_cap", 1.0)
        self.ctx.set_batch_cap(0.0)
        try:
            return super().train_batch(idxs, sync=sync)
        finally:
            self.ctx.set_batch_cap(prev)

    def normalize(self):
        """public/GRU.py:476-481: lt rows scaled to unit L2 norm (never called by the reference's drivers)."""
        t = self.lt.t.float()
        self.lt.t.copy_((t / t.pow(2).sum(dim=1, keepdim=True).sqrt()).to(self.lt.t.dtype))


class OboSpatialGru(GruBasic):
    """public/GRU_Spatial.py:42-292 - Distance2Pre."""

    spatial = True

    def __init__(self, train, test, dist, alpha_lambda, n_user, n_item, n_dists, n_in, n_hidden,
                 device="cuda:0", init=None, seed=None, coords=None, table_dtype="f32", pad_dim=True):
        n_dist, dd = n_dists
        self.n_dist, self.dd = int(n_dist), float(dd)                                      # dd in km (ref passes dd/1000)
        super().__init__(train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, device=device, init=init, seed=seed, table_dtype=table_dtype,
                         pad_dim=pad_dim)
        if self._csr is not None:
            dp, dq, tes_dist_masks = (np.ascontiguousarray(v, np.int32) for v in (self._csr.dp, self._csr.dq, self._csr.tes_dp))
        else:
            tra_dist_masks, tes_dist_masks, tra_dist_neg_masks = dist
            _, dp = padded_to_csr(tra_dist_masks, self._lens)
            _, dq = padded_to_csr(tra_dist_neg_masks, self._lens)
        self._check_ids("train distance bins", dp, self.n_dist); self._check_ids("negative distance bins", dq, self.n_dist)
        self._check_ids("test distance bins", tes_dist_masks, self.n_dist)
        self.dp, self.dq = torch.as_tensor(dp).to(self.device), torch.as_tensor(dq).to(self.device)
        self.tes_dist_masks = self._dev(tes_dist_masks, torch.int32)
        rng = np.random.default_rng(None if seed is None else seed + 1) if seed is not None else np.random
        u = lambda *s: rng.uniform(-0.5, 0.5, s)
        D, NB = self.dim, self.n_dist + 1
        init = init or {}
        g = lambda k, v: np.asarray(init[k], np.float64) if k in init else v()
        self.di = self._shared(g("di", lambda: u(NB, D)), "cols")                          # :57
        self.vs = self._shared(g("vs", lambda: u(NB, D)), "cols")                          # :60
        self.bs = Shared(self._dev(g("bs", lambda: np.zeros(NB))))                         # :61
        self.wd = Shared(self._dev(np.reshape(g("wd", lambda: rng.uniform(0, 0.5)), (1,))), scalar=True)   # :66
        self.loss_weight = Shared(self._dev(g("loss_weight", lambda: u(2))))               # :70
        self.trained_dists = self._shared(u(NB, D), "cols")                                # :74
        self.prob = None                       # dense (n_user, n_item) only on request (update_prob)
        self.trained_sus = None                # (n_user, NB) - fused alternative to `prob`
        self.use_bin_matrix = None             # None = auto: resident U x N bin matrix when it is <= 16 GiB, else bins on the fly
        self.coords = None if coords is None else self._dev(np.asarray(coords, np.float64), torch.float64)
        self._cphi = None if coords is None else self._dev(cos_lat(coords), torch.float64)
        self._binthr = None if coords is None else self._dev(bin_thresholds(self.dd * 1000.0, self.n_dist), torch.float64)
        self.params = [self.ui, self.wh, self.bi, self.vs, self.bs, self.wd, self.loss_weight]
        self.l2 = _L2(self, ["lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight"])   # :83-88

    def _xw(self):
        return 2 * self.dim

    def load_params(self, loaded_objects):
        """public/GRU_Spatial.py:92-101: [loss_weight, wd, lt, di, ui, wh, bi, vs, bs]."""
        for sh, v in zip((self.loss_weight, self.wd, self.lt, self.di, self.ui, self.wh, self.bi, self.vs, self.bs), loaded_objects):
            sh.set_value(v)

    def s_update_neg_masks(self, tra_buys_neg_masks, tes_buys_neg_masks, tra_dist_neg_masks):
        """public/GRU_Spatial.py:103-107."""
        self.update_neg_masks(tra_buys_neg_masks, tes_buys_neg_masks)
        _, dq = padded_to_csr(tra_dist_neg_masks, self._lens)
        self.dq = torch.as_tensor(dq).to(self.device)

    def update_trained_dists(self):
        """public/GRU_Spatial.py:109-112."""
        self.trained_dists.t.copy_(self.di.t)

    def update_prob(self, prob):
        """public/GRU_Spatial.py:114-115 - dense (n_user, n_item) matrix (compatibility path)."""
        self.prob = self._dev(np.asarray(prob, np.float64)).reshape(self.n_user, self.n_item)

    def update_trained_sus(self, all_sus):
        """Fused replacement of fun_acquire_prob + update_prob (Load_Data_by_length.py:218-235): keep the
        (n_user, n_dist+1) distance-bin probabilities; prob rows are rebuilt on the device per batch
        from the POI coordinates (needs coords= at construction)."""
        t = all_sus if isinstance(all_sus, torch.Tensor) else self._dev(np.asarray(all_sus, np.float64))
        # rows padded to whole 32-user tiles (poi_score_topk_ulptai reads whole tiles); `trained_sus` is the
        # caller's table, `_sus_masked` the same with column n_dist ("too far") zeroed as the scoring path expects
        pad = ((self.n_user + 31) // 32) * 32
        buf = torch.zeros((pad, self.n_dist + 1), dtype=torch.float32, device=self.device)
        buf[:self.n_user] = t.to(self.device, torch.float32).reshape(self.n_user, self.n_dist + 1)
        self.trained_sus = buf[:self.n_user]
        self._sus_masked = buf.clone()
        self._sus_masked[:, self.n_dist] = 0.0
        self.prob = None
        lens = torch.as_tensor(self._off_host[1:].astype(np.int64) - 1).to(self.device)
        self._last_poi = self.p.index_select(0, lens).contiguous()

    def build_ulptai(self):
        """usrs_last_poi_to_all_intervals (prog_bpr_gru_spatial.py:90): distance bins of (last train POI,
        every POI), built once on the device and kept resident in the scoring kernel's tile order
        (include/poi_hip.h, poi_ulptai_build).  Needs coords= at construction."""
        if self.coords is None:
            raise _lib.PoiError("build_ulptai needs coords= at construction")
        bb = 1 if self.n_dist <= 255 else 2
        nut, nt = (self.n_user + 31) // 32, (self.n_item + 31) // 32
        lens = torch.as_tensor(self._off_host[1:].astype(np.int64) - 1).to(self.device)
        last = self.p.index_select(0, lens).contiguous()
        buf = torch.empty(nut * nt * 1024 * bb, dtype=torch.uint8, device=self.device)
        self.ctx.check(self.lib.poi_ulptai_build(self.ctx.handle, _ptr(self.coords), _ptr(self._cphi), _ptr(self._binthr), _ptr(last),
                                                 self.n_user, self.n_item, self.n_dist, self.dd * 1000.0, _ptr(buf), bb, self._stream()))
        self._ulptai, self._ulptai_bytes, self._ulptai_row = buf, bb, nt * 1024 * bb
        return buf

    def ulptai_host(self):
        """The (n_user, n_item) bin matrix decoded from the device layout (tests / inspection)."""
        buf = getattr(self, "_ulptai", None)
        if buf is None:
            buf = self.build_ulptai()
        bb = self._ulptai_bytes
        nut, nt = (self.n_user + 31) // 32, (self.n_item + 31) // 32
        a = buf.cpu().numpy().view(np.uint8 if bb == 1 else np.uint16).reshape(nut, nt, 64, 16)
        lane, r = np.arange(64)[:, None], np.arange(16)[None, :]
        row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)                 # (64, 16) user row within the tile
        col = np.broadcast_to(lane & 31, (64, 16))
        out = np.empty((nut * 32, nt * 32), np.int32)
        for ut in range(nut):
            blk = np.empty((32, nt, 32), np.int32)
            blk[row, :, col] = a[ut].transpose(1, 2, 0)
            out[ut * 32:(ut + 1) * 32] = blk.reshape(32, nt * 32)
        return out[:self.n_user, :self.n_item]

    def compute_sub_topk(self, start_end, k, return_scores=False):
        """Fused scoring + top-K.  With bin probabilities (update_trained_sus) and coordinates, contiguous
        user ranges starting at a multiple of 32 take the distance term from the resident bin matrix
        (poi_score_topk_ulptai); anything else falls back to the dense prob rows."""
        if k > 32:
            return self._topk_from_scores(start_end, k, return_scores)
        ids, lo = self._ids(start_end)
        ubm = self.use_bin_matrix
        if ubm is None:
            ubm = self.n_user * float(self.n_item) * (1 if self.n_dist <= 255 else 2) <= float(1 << 34)
        if ubm and self.prob is None and self.trained_sus is not None and self.coords is not None and lo is not None and lo % 32 == 0 \
                and self.kdim <= 128:
            if getattr(self, "_ulptai", None) is None:
                self.build_ulptai()
            n = ids.numel()
            users = self._rows(self.trained_users.t, ids, lo)
            st = self._sus_masked[lo:lo + n]
            idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
            sc = torch.empty((n, k), dtype=torch.float32, device=self.device) if return_scores else None
            bins = self._ulptai.data_ptr() + (lo // 32) * self._ulptai_row
            seed = self._seed_begin(lo, n, k)
            self.ctx.check(self.lib.poi_score_topk_ulptai(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), n, self.n_item, self.kdim,
                                                          _ptr(self.wd.t), _ptr(st), bins, self._ulptai_bytes, self.n_dist, int(k),
                                                          _ptr(idx), _ptr(sc), self._stream()))
            self._seed_end(seed, idx)
            return (idx, sc) if return_scores else idx
        if self.prob is None and self.trained_sus is not None and self.coords is not None:
            # anything else (unaligned / arbitrary id lists, dim 256, tables whose U x N bin matrix cannot exist): the bins are
            # computed on the fly inside the scoring kernel - no bin matrix, no dense prob rows
            return self._topk_geo(ids, lo, k, return_scores)
        return super().compute_sub_topk(start_end, k, return_scores)

    def _topk_geo(self, ids, lo, k, return_scores=False):
        n = ids.numel()
        users = self._rows(self.trained_users.t, ids, lo)
        pad = ((n + 31) // 32) * 32
        if lo is not None and lo + pad <= self._sus_masked.shape[0]:
            st = self._sus_masked[lo:lo + pad]
        else:                                   # whole 32-user tiles must be readable
            st = torch.zeros((pad, self.n_dist + 1), dtype=torch.float32, device=self.device)
            st[:n] = self._rows(self._sus_masked, ids, lo)
        lp = self._rows(self._last_poi, ids, lo)
        idx = torch.empty((n, k), dtype=torch.int32, device=self.device)
        sc = torch.empty((n, k), dtype=torch.float32, device=self.device) if return_scores else None
        seed = self._seed_begin(lo, n, k)
        self.ctx.check(self.lib.poi_score_topk_geo(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), n, self.n_item, self.kdim, _ptr(self.wd.t),
                                                   _ptr(st), _ptr(self.coords), _ptr(self._cphi), _ptr(self._binthr), _ptr(lp), self.n_dist,
                                                   self.dd * 1000.0, int(k), _ptr(idx), _ptr(sc), self._stream()))
        self._seed_end(seed, idx)
        return (idx, sc) if return_scores else idx

    def _prob_rows(self, ids, lo):
        if self.prob is not None:
            return self.wd.t, self._rows(self.prob, ids, lo)
        if self.trained_sus is not None:
            if self.coords is None:
                raise _lib.PoiError("update_trained_sus needs coords= at construction")
            n = ids.numel()
            lp, st = self._rows(self._last_poi, ids, lo), self._rows(self.trained_sus, ids, lo)
            need = n * self.n_item                      # persistent grow-only buffer: no allocator churn per batch
            if getattr(self, "_prob_buf", None) is None or self._prob_buf.numel() < need:
                self._prob_buf = torch.empty(need, dtype=torch.float32, device=self.device)
            prob = self._prob_buf[:need].view(n, self.n_item)
            self.ctx.check(self.lib.poi_dist_prob(self.ctx.handle, _ptr(self.coords), _ptr(self._cphi), _ptr(self._binthr), _ptr(lp), _ptr(st), n,
                                                  self.n_item, self.n_dist, self.dd * 1000.0, _ptr(prob), self._stream()))
            return self.wd.t, prob
        return None, None

    def _params(self, snapshot=False):
        P = super()._params(snapshot)
        P.di = (self.trained_dists if snapshot else self.di).t.data_ptr()
        P.vs, P.bs, P.wd, P.lw = self.vs.t.data_ptr(), self.bs.t.data_ptr(), self.wd.t.data_ptr(), self.loss_weight.t.data_ptr()
        P.n_dist = self.n_dist
        return P

    def _tables(self):
        T = super()._tables()
        T.dp, T.dq = self.dp.data_ptr(), self.dq.data_ptr()
        return T

    def train(self, idx):
        """seq_train(uidx) -> [los, sur, upq, ls] (public/GRU_Spatial.py:222,290-292)."""
        o = self.train_batch(np.atleast_1d(idx))[0]
        return [float(o[0]), float(o[1]), float(o[2]), np.array([o[3], o[4]])]

    def train_batch(self, idxs, sync=True):
        """Throughput mode: (n, 5) rows [los, sur, upq, ls0, ls1]."""
        ids, _ = self._ids(idxs)
        n = ids.numel()
        out = torch.empty((n, 5), dtype=torch.float32, device=self.device)
        P, T = self._params(), self._tables()
        self.ctx.check(self.lib.poi_spatial_step(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), n,
                                                 self.alpha_lambda[0], self.alpha_lambda[1], _ptr(out), self._stream()))
        return out.cpu().numpy() if sync else out

    def train_sequence(self, idxs, sync=True):
        """The reference's epoch loop `for uidx in order: model.train(uidx)` (prog_bpr_gru_spatial.py:246-254) without a host round trip
        per step: the ids are staged on the device once and every user is ONE poi_spatial_step launch of one sequence (sequential SGD,
        the reference's semantics - not the batch rule), the (n, 5) loss rows [los, sur, upq, ls0, ls1] are read once at the end."""
        ids, _ = self._ids(idxs)
        n = ids.numel()
        out = torch.empty((n, 5), dtype=torch.float32, device=self.device)
        P, T = self._params(), self._tables()
        pP, pT, st = ctypes.byref(P), ctypes.byref(T), self._stream()
        ip, op = ids.data_ptr(), out.data_ptr()
        step, h, a, l = self.lib.poi_spatial_step, self.ctx.handle, self.alpha_lambda[0], self.alpha_lambda[1]
        for k in range(n):
            rc = step(h, pP, pT, ctypes.c_void_p(ip + 4 * k), 1, a, l, ctypes.c_void_p(op + 20 * k), st)
            if rc:
                self.ctx.check(rc)
        return out.cpu().numpy() if sync else out

    def predict(self, idxs):
        """public/GRU_Spatial.py:282-288 -> [hts (n, D), sts (n, n_dist+1)]."""
        h, s = self.predict_device(idxs)
        return [np.ascontiguousarray(h[:, :self.dim].cpu().numpy()), s.cpu().numpy()]

    def predict_device(self, idxs):
        ids, out_row = self._by_length(idxs)
        n = ids.numel()
        hts = torch.empty((n, self.kdim), dtype=torch.float32, device=self.device)
        sts = torch.empty((n, self.n_dist + 1), dtype=torch.float32, device=self.device)
        P, T = self._params(snapshot=True), self._tables()
        self.ctx.check(self.lib.poi_gru_predict(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), _ptr(out_row), n, _ptr(hts), _ptr(sts),
                                                self._stream()))
        return hts, sts


# =================================================================================================
class OboCARNN(GruBasic):
    """public/CA_RNN.py:46-227 - CA-RNN (flag 3 of prog_bpr_gru_spatial.py:141-151): interval-specific transition
    matrices wd[(n_dist+1), H, D], input matrix M (H, D), sigmoid RNN, BPR.  Same ctor as the reference; `ulptai` (the
    reference's U x N usrs_last_poi_to_all_intervals matrix) is accepted for signature compatibility but never
    uploaded: with coords= the scoring kernel computes those bins on the fly (bit-identical, tested)."""

    spatial = True          # has distance-bin tables (negatives refresh computes dq)
    _pad_ok = False
    sync_names = ("lt", "wd", "M")      # every trainable tensor (dist.model_sync): POI table, interval matrices, input matrix

    def __init__(self, train, test, dist, alpha_lambda, n_user, n_item, n_dists, n_in, n_hidden, ulptai=None,
                 device="cuda:0", init=None, seed=None, coords=None):
        n_dist, dd = n_dists
        self.n_dist, self.dd = int(n_dist), float(dd)
        super().__init__(train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, device=device, init=init, seed=seed)
        if self._csr is not None:
            dp, dq, tes_dist_masks = (np.ascontiguousarray(v, np.int32) for v in (self._csr.dp, self._csr.dq, self._csr.tes_dp))
        else:
            tra_dist_masks, tes_dist_masks, tra_dist_neg_masks = dist
            _, dp = padded_to_csr(tra_dist_masks, self._lens)
            _, dq = padded_to_csr(tra_dist_neg_masks, self._lens)
        self._check_ids("train distance bins", dp, self.n_dist); self._check_ids("negative distance bins", dq, self.n_dist)
        self.dp, self.dq = torch.as_tensor(dp).to(self.device), torch.as_tensor(dq).to(self.device)
        self.tes_dist_masks = self._dev(tes_dist_masks, torch.int32)
        rng = np.random.default_rng(seed + 1) if seed is not None else np.random
        u = lambda *s: rng.uniform(-0.5, 0.5, s)
        D, NB = self.dim, self.n_dist + 1
        init = init or {}
        g = lambda k, v: np.asarray(init[k], np.float64) if k in init else v()
        self.M = Shared(self._dev(g("M", lambda: u(D, D))))                                # CA_RNN.py:55-56
        self.wd = Shared(self._dev(g("wd", lambda: u(NB, D, D))))                          # :61-62
        self.trained_dists = Shared(self._dev(u(NB, D, D)))                                # :66-67
        self.coords = None if coords is None else self._dev(np.asarray(coords, np.float64), torch.float64)
        self._cphi = None if coords is None else self._dev(cos_lat(coords), torch.float64)
        self._binthr = None if coords is None else self._dev(bin_thresholds(self.dd * 1000.0, self.n_dist), torch.float64)
        self.params = [self.M]
        self.l2 = _L2(self, ["lt", "wd", "M"])                                             # :70-76

    def s_update_neg_masks(self, tra_buys_neg_masks, tes_buys_neg_masks, tra_dist_neg_masks):
        """public/CA_RNN.py:80-84."""
        self.update_neg_masks(tra_buys_neg_masks, tes_buys_neg_masks)
        _, dq = padded_to_csr(tra_dist_neg_masks, self._lens)
        self.dq = torch.as_tensor(dq).to(self.device)

    def update_trained_dists(self):
        """public/CA_RNN.py:86-89."""
        self.trained_dists.t.copy_(self.wd.t)

    def _cparams(self, snapshot=False):
        P = _lib.CarnnParams()
        P.lt = (self.trained_items if snapshot else self.lt).t.data_ptr()
        P.wd = (self.trained_dists if snapshot else self.wd).t.data_ptr()
        P.M = self.M.t.data_ptr()
        P.n_item, P.n_dist, P.dim = self.n_item, self.n_dist, self.dim
        return P

    def _tables(self):
        T = super()._tables()
        T.dp, T.dq = self.dp.data_ptr(), self.dq.data_ptr()
        return T

    def train(self, idx):
        """seq_train(uidx) -> los (public/CA_RNN.py:160-170,219-221)."""
        return float(self.train_batch(np.atleast_1d(idx))[0])

    def train_batch(self, idxs, sync=True):
        ids, _ = self._ids(idxs)
        n = ids.numel()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        P, T = self._cparams(), self._tables()
        self.ctx.check(self.lib.poi_carnn_step(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), n,
                                               self.alpha_lambda[0], self.alpha_lambda[1], _ptr(out), self._stream()))
        return out.cpu().numpy() if sync else out

    def predict_device(self, idxs):
        ids, _ = self._ids(idxs)
        n = ids.numel()
        hts = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        P, T = self._cparams(snapshot=True), self._tables()
        self.ctx.check(self.lib.poi_carnn_predict(self.ctx.handle, ctypes.byref(P), ctypes.byref(T), _ptr(ids), n, _ptr(hts), self._stream()))
        return hts

    def compute_sub_all_scores_device(self, start_end):
        """public/CA_RNN.py:91-101 -> (n, n_item) device tensor."""
        if self.coords is None:
            raise _lib.PoiError("OboCARNN scoring needs coords= at construction (the interval of (last train POI, POI) is computed on the device)")
        ids, users, lo = self._users_rows(start_end)
        n = ids.numel()
        if getattr(self, "_last_poi", None) is None:
            lens = torch.as_tensor(self._off_host[1:].astype(np.int64) - 1).to(self.device)
            self._last_poi = self.p.index_select(0, lens).contiguous()
        lp = self._rows(self._last_poi, ids, lo)
        out = torch.empty((n, self.n_item), dtype=torch.float32, device=self.device)
        self.ctx.check(self.lib.poi_carnn_score_all(self.ctx.handle, _ptr(users), _ptr(self.trained_items.t), _ptr(self.M.t), _ptr(self.trained_dists.t),
                                                    _ptr(self.coords), _ptr(self._cphi), _ptr(self._binthr), _ptr(lp), n, self.n_item, self.n_dist,
                                                    self.dim, self.dd * 1000.0, _ptr(out), self._stream()))
        return out

    def compute_sub_topk(self, start_end, k, return_scores=False):
        """Valuate.py:132-146 on the CA-RNN scores: the (n, n_item) rows stay on the device, poi_topk selects."""
        return self._topk_from_scores(start_end, k, return_scores)


# =================================================================================================
class MfBasic(_Base):
    """public/BPR.py:28-134."""

    def __init__(self, train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, device="cuda:0", init=None, seed=None, table_dtype="f32"):
        """table_dtype="f16": the POI table `lt` and its evaluation snapshot are STORED as IEEE half (float32 arithmetic, snapshot-mode steps only)."""
        self._setup(device, alpha_lambda)
        self.n_user, self.n_item, self.dim = int(n_user), int(n_item), int(n_in)
        self.kdim = self.dim
        if table_dtype not in ("f32", "f16"):
            raise ValueError("table_dtype must be 'f32' or 'f16'")
        self.table_dtype = table_dtype
        self._load_tables(train, test)
        rng = np.random.default_rng(seed) if seed is not None else np.random
        u = lambda *s: rng.uniform(-0.5, 0.5, s)
        init = init or {}
        g = lambda k, v: (init[k] if isinstance(init[k], torch.Tensor) else np.asarray(init[k], np.float64)) if k in init else v()
        tab = (lambda a: self._dev(a).to(torch.float16)) if table_dtype == "f16" else self._dev
        if (n_item + 1) * self.dim > (1 << 28):
            # tables of hundreds of millions of elements (config X: 10 M x 256) are drawn ON the device, as in GruBasic: the host draw is 20 GB of float64
            gen = torch.Generator(device=self.device).manual_seed(0 if seed is None else int(seed))
            u_tab = lambda rows: torch.rand((rows, self.dim), generator=gen, device=self.device, dtype=torch.float32) - 0.5
        else:
            u_tab = lambda rows: u(rows, self.dim)
        self.ux = Shared(self._dev(g("ux", lambda: u(n_user, self.dim))))                  # BPR.py:51
        self.lt = Shared(tab(g("lt", lambda: u_tab(n_item + 1))))                          # :52
        self.trained_items = Shared(tab(u_tab(n_item + 1)))
        self.trained_users = Shared(self._dev(u(n_user, self.dim)))
        if table_dtype == "f16":
            self.ctx.register_f16(self.lt.t); self.ctx.register_f16(self.trained_items.t)

    def __del__(self):
        try:
            if getattr(self, "table_dtype", "f32") == "f16":
                self.ctx.unregister_f16(self.lt.t); self.ctx.unregister_f16(self.trained_items.t)
        except Exception:
            pass

    def update_trained_users(self):
        """public/BPR.py:71-74 (no argument: copies ux)."""
        self.trained_users.t.copy_(self.ux.t)


class OboBpr(MfBasic):
    """public/BPR.py:191-241."""

    def __init__(self, train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, **kw):
        super().__init__(train, test, alpha_lambda, n_user, n_item, n_in, n_hidden, **kw)
        self.params = [self.ux, self.lt]
        self.l2 = _L2(self, ["ux", "lt"])                                                  # :194-197

    def train(self, u_idx, pq_idx):
        """bpr_train(uidx, [p, q]) -> -log sigmoid(u)  (public/BPR.py:234-241)."""
        return float(self.train_batch([u_idx], [pq_idx[0]], [pq_idx[1]])[0])

    def train_batch(self, uidx, p, q, mode="snapshot", sync=True):
        conv = lambda v: v.to(self.device, torch.int32).contiguous() if isinstance(v, torch.Tensor) else \
            torch.as_tensor(np.asarray(v, np.int32)).to(self.device)
        u, pp, qq = conv(uidx), conv(p), conv(q)
        n = u.numel()
        loss = torch.empty(n, dtype=torch.float32, device=self.device)
        m = _lib.BPR_HOGWILD if mode == "hogwild" else _lib.BPR_SNAPSHOT
        self.ctx.check(self.lib.poi_bpr_step(self.ctx.handle, _ptr(self.ux.t), _ptr(self.lt.t), self.n_user, self.n_item, self.dim,
                                             _ptr(u), _ptr(pp), _ptr(qq), n, self.alpha_lambda[0], self.alpha_lambda[1],
                                             _ptr(loss), m, self._stream()))
        if sync:
            nb = self.ctx.take_bad_ids(self._stream().value if hasattr(self._stream(), "value") else None)
            if nb:                                     # the reference's gather raises IndexError (public/BPR.py:214-218)
                raise IndexError("%d id(s) outside the user / POI tables in this launch: those triples moved nothing, their losses are NaN" % nb)
        return loss.cpu().numpy() if sync else loss

    def epoch_triples(self):
        """All (user, pos_t, neg_t) triples of the train tables, in the reference's order
        (prog_bpr_gru_spatial.py:240-244) - device int32 tensors."""
        lens = torch.as_tensor(np.diff(self._off_host.astype(np.int64))).to(self.device)
        u = torch.repeat_interleave(self._arange, lens)
        return u, self.p, self.q
