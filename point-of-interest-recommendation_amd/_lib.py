"""ctypes binding of libpoi_hip.so (include/poi_hip.h).  The library is the product: if it is not
built, or a call fails, this module raises - there is no CPU fallback."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpoi_hip.so")
ABI_VERSION = 6

BPR_SNAPSHOT, BPR_HOGWILD = 0, 1


class PoiError(RuntimeError):
    pass


class GruParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "lw")] + \
               [("n_item", c_int32), ("n_dist", c_int32), ("dim", c_int32)]


class CarnnParams(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("lt", "wd", "M")] + [("n_item", c_int32), ("n_dist", c_int32), ("dim", c_int32)]


class SyncSeg(ctypes.Structure):
    _fields_ = [("cur", c_void_p), ("rows", c_int64), ("width", c_int64), ("rule", c_int32), ("dtype", c_int32)]


SYNC_SUM, SYNC_MEAN, SYNC_MEAN_TOUCHED = 0, 1, 2


class SeqTables(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("off", "p", "q", "dp", "dq")] + \
               [("n_user", c_int32), ("len_max", c_int32), ("max_len", c_int32)]


# name -> (restype, argtypes); mirrors include/poi_hip.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    "poi_abi_version": (c_int, []),
    "poi_ctx_create": (c_int, [POINTER(c_void_p), c_int]),
    "poi_ctx_destroy": (c_int, [c_void_p]),
    "poi_last_error": (c_char_p, [c_void_p]),
    "poi_ctx_num_cu": (c_int, [c_void_p]),
    "poi_ctx_set_engine": (c_int, [c_void_p, c_int]),
    "poi_ctx_set_batch_cap": (c_int, [c_void_p, c_float]),
    "poi_ctx_set_graph": (c_int, [c_void_p, c_int, c_int, c_int]),
    "poi_ctx_set_topk_seed": (c_int, [c_void_p, c_void_p, c_int32]),
    "poi_ctx_set_topk_filter": (c_int, [c_void_p, c_int]),
    "poi_ctx_topk_filter_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "poi_ctx_graph_replays": (c_int64, [c_void_p]),
    "poi_ctx_take_bad_ids": (c_int64, [c_void_p, c_void_p]),
    "poi_ctx_register_f16": (c_int, [c_void_p, c_void_p, c_int64]),
    "poi_ctx_unregister_f16": (c_int, [c_void_p, c_void_p]),
    "poi_ctx_set_f16_rounding": (c_int, [c_void_p, c_int, ctypes.c_uint32]),
    "poi_ctx_set_split_products": (c_int, [c_void_p, c_int]),
    "poi_ctx_set_exact_forward": (c_int, [c_void_p, c_int, c_int]),
    "poi_ctx_set_option": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "poi_ctx_set_small_launch": (c_int, [c_void_p, c_int]),
    "poi_ctx_set_one_sequence_path": (c_int, [c_void_p, c_int]),
    "poi_ctx_set_regroup_min": (c_int, [c_void_p, c_int]),
    "poi_bpr_step": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                             c_int32, c_float, c_float, c_void_p, c_int, c_void_p]),
    "poi_spatial_step": (c_int, [c_void_p, POINTER(GruParams), POINTER(SeqTables), c_void_p, c_int32, c_float, c_float,
                                 c_void_p, c_void_p]),
    "poi_gru_step": (c_int, [c_void_p, POINTER(GruParams), POINTER(SeqTables), c_void_p, c_int32, c_float, c_float,
                             c_void_p, c_void_p]),
    "poi_gru_predict": (c_int, [c_void_p, POINTER(GruParams), POINTER(SeqTables), c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                c_void_p]),
    "poi_carnn_step": (c_int, [c_void_p, POINTER(CarnnParams), POINTER(SeqTables), c_void_p, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "poi_carnn_predict": (c_int, [c_void_p, POINTER(CarnnParams), POINTER(SeqTables), c_void_p, c_int32, c_void_p, c_void_p]),
    "poi_carnn_score_all": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                    c_int32, c_int32, c_double, c_void_p, c_void_p]),
    "poi_score_all": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                              c_void_p]),
    "poi_score_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                               c_void_p, c_void_p, c_void_p]),
    "poi_topk": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "poi_auc_preference": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                   c_void_p, c_void_p]),
    "poi_sumsq": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "poi_dist_prob": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double,
                              c_void_p, c_void_p]),
    "poi_ulptai_build": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_double,
                                 c_void_p, c_int32, c_void_p]),
    "poi_score_topk_ulptai": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32,
                                      c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "poi_score_topk_geo": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int32, c_double, c_int32, c_void_p, c_void_p, c_void_p]),
    "poi_rank_metrics": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "poi_sample_negatives": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, ctypes.c_uint64,
                                     c_void_p, c_void_p, c_void_p]),
    "poi_neg_dist_bins": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_double,
                                  c_void_p, c_void_p]),
    "poi_delta_make": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "poi_delta_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "poi_comm_available": (c_int, []),
    "poi_comm_unique_id": (c_int, [c_void_p]),
    "poi_comm_init_rank": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    "poi_comm_destroy": (c_int, [c_void_p]),
    "poi_comm_world": (c_int, [c_void_p]),
    "poi_comm_rank": (c_int, [c_void_p]),
    "poi_allreduce_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "poi_sync_create": (c_int, [c_void_p, c_int, POINTER(SyncSeg), c_int32, POINTER(c_void_p)]),
    "poi_sync_destroy": (c_int, [c_void_p]),
    "poi_sync_begin_epoch": (c_int, [c_void_p, c_void_p]),
    "poi_sync_make_delta": (c_int, [c_void_p, c_void_p]),
    "poi_sync_buffer": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "poi_sync_buffer16": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64)]),
    "poi_sync_apply": (c_int, [c_void_p, c_int32, c_void_p]),
    "poi_sync_end_epoch": (c_int, [c_void_p, c_void_p, c_void_p]),
    "poi_sync_stats": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    "poi_sync_last_error": (c_char_p, []),
    "poi_checksum": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "poi_timing_enable": (c_int, [c_void_p, c_int]),
    "poi_timing_reset": (c_int, [c_void_p]),
    "poi_timing_get": (c_int, [c_void_p, c_char_p, POINTER(c_double), POINTER(c_int64)]),
    "poi_selftest": (c_int, [c_void_p, c_void_p]),
}

_lib = None


def load():
    """dlopen the in-tree library and bind every exported symbol.  Raises PoiError when missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PoiError("%s is not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(no CPU fallback exists for the hot path)" % LIB_PATH)
    # PyTorch is the device-memory container: its HIP runtime (torch/lib/libamdhip64.so) must be the one this
    # library binds to - loading libpoi_hip.so first would pull in a second runtime from /opt/rocm, which
    # knows nothing about torch's allocations and streams ("no HIP device visible").
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    if lib.poi_abi_version() != ABI_VERSION:
        raise PoiError("libpoi_hip.so ABI %d != binding %d" % (lib.poi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


class Context:
    """Owns one poi_ctx (device scratch + error text) bound to a HIP device."""

    def __init__(self, device=0):
        self.lib = load()
        h = c_void_p()
        rc = self.lib.poi_ctx_create(ctypes.byref(h), int(device))
        if rc != 0:
            raise PoiError("poi_ctx_create(%d) failed: %s" % (device, self.lib.poi_last_error(None).decode()))
        self.handle = h
        self.device = int(device)

    def check(self, rc):
        if rc != 0:
            raise PoiError("libpoi_hip error %d: %s" % (rc, self.lib.poi_last_error(self.handle).decode()))

    @property
    def num_cu(self):
        return self.lib.poi_ctx_num_cu(self.handle)

    def set_engine(self, name):
        """'auto' | 'seq' | 'tile' | 'tile32' | 'exact' (float64 arithmetic; see poi_ctx_set_engine)."""
        self.check(self.lib.poi_ctx_set_engine(self.handle, {"auto": 0, "seq": 1, "tile": 2, "tile32": 3, "exact": 4}[name]))

    def set_graph(self, on, min_n=0, max_n=1 << 30):
        """hipGraph replay of the tile engine's training launches (poi_ctx_set_graph)."""
        self.check(self.lib.poi_ctx_set_graph(self.handle, int(bool(on)), int(min_n), int(max_n)))

    def set_topk_seed(self, seed, k_seed):
        """Seed ids of the next fused top-K call (poi_ctx_set_topk_seed): an (n, k_seed) int32 device tensor or None."""
        self.check(self.lib.poi_ctx_set_topk_seed(self.handle, None if seed is None else seed.data_ptr(), int(k_seed)))

    def set_topk_filter(self, on):
        """Two-stage fused top-K (f16 filter + exact float32 rescoring): poi_ctx_set_topk_filter.  on: False / True, or "items" / "users" =
        two-stage with the item-stationary GEO filter forced / forbidden (default: chosen by shape)."""
        mode = {"items": 2, "users": 3}.get(on, int(bool(on))) if isinstance(on, str) else int(bool(on))
        self.check(self.lib.poi_ctx_set_topk_filter(self.handle, mode))

    def topk_filter_stats(self):
        """dict(users, survivors, tiles, tiles_flagged) of the last two-stage fused top-K call (poi_ctx_topk_filter_stats)."""
        v = [c_int64(0) for _ in range(4)]
        self.check(self.lib.poi_ctx_topk_filter_stats(self.handle, *[ctypes.byref(x) for x in v]))
        return dict(zip(("users", "survivors", "tiles", "tiles_flagged"), (x.value for x in v)))

    def take_bad_ids(self, stream=None):
        """Out-of-range ids seen by poi_bpr_step since the last call (synchronises the stream, clears the counter)."""
        v = int(self.lib.poi_ctx_take_bad_ids(self.handle, ctypes.c_void_p(stream or 0)))
        if v < 0:
            self.check(v)
        return v

    def graph_replays(self):
        return int(self.lib.poi_ctx_graph_replays(self.handle))

    def set_batch_cap(self, cap):
        """Batch rule cap (include/poi_hip.h, poi_ctx_set_batch_cap): 1 = mean rule, 0 = the mini-batch rule of public/GRU.py:395-498."""
        self.check(self.lib.poi_ctx_set_batch_cap(self.handle, float(cap)))
        self.batch_cap = float(cap)

    def register_f16(self, tensor):
        """Declare a torch.float16 device tensor as an IEEE-half POI table (poi_ctx_register_f16)."""
        self.check(self.lib.poi_ctx_register_f16(self.handle, tensor.data_ptr(), tensor.numel() * 2))

    def set_f16_rounding(self, mode, seed=0):
        """'nearest' | 'stochastic' write-back of a half POI table (poi_ctx_set_f16_rounding)."""
        self.check(self.lib.poi_ctx_set_f16_rounding(self.handle, {"nearest": 0, "stochastic": 1}[mode], int(seed) & 0xFFFFFFFF))

    def set_split_products(self, on=True):
        """Recurrent kernels of the tile engine on bf16 x 3 split products (default) or float32-input MFMAs (poi_ctx_set_split_products)."""
        self.check(self.lib.poi_ctx_set_split_products(self.handle, 1 if on else 0))

    def set_exact_forward(self, on=True, per_sequence_max=-1):
        """Training launches run the forward pass in fixed point on the int8 matrix cores + float64 gates (poi_ctx_set_exact_forward, default
        on); per_sequence_max: launches of at most this many sequences use the per-sequence float64 kernel (default 512; -1 keeps it)."""
        self.check(self.lib.poi_ctx_set_exact_forward(self.handle, 1 if on else 0, int(per_sequence_max)))

    def set_option(self, name, value):
        """Named tuning switch of the tile engine (poi_ctx_set_option: "forward_table_compact", "forward_table_compact_min", "head_split",
        "early_bins", "hot_bins", "hybrid", "hybrid_min", "hybrid_max", "hybrid_force")."""
        self.check(self.lib.poi_ctx_set_option(self.handle, name.encode(), int(value)))

    def set_small_launch(self, max_sequences=1800):
        """Launches of at most this many sequences use the per-sequence recurrent kernels (poi_ctx_set_small_launch; 0 disables)."""
        self.check(self.lib.poi_ctx_set_small_launch(self.handle, int(max_sequences)))

    def set_regroup_min(self, min_sequences=1280):
        """Launches below this many sequences take the two-table path instead of the regrouped backward pass (poi_ctx_set_regroup_min)."""
        self.check(self.lib.poi_ctx_set_regroup_min(self.handle, int(min_sequences)))

    def set_one_sequence_path(self, on=True):
        """Launches of one Distance2Pre sequence through the five-kernel path (poi_ctx_set_one_sequence_path)."""
        self.check(self.lib.poi_ctx_set_one_sequence_path(self.handle, 1 if on else 0))

    def unregister_f16(self, tensor):
        self.check(self.lib.poi_ctx_unregister_f16(self.handle, tensor.data_ptr()))

    def timing(self, on=True, period=1):
        """Per-kernel HIP-event timing; period N > 1 instruments only every N-th training launch (poi_timing_enable)."""
        self.check(self.lib.poi_timing_reset(self.handle))
        self.check(self.lib.poi_timing_enable(self.handle, (max(int(period), 1) if on else 0)))

    def timing_get(self, kernel):
        """(total milliseconds, launches) recorded for `kernel` since the last timing() call."""
        ms, n = c_double(0), c_int64(0)
        self.check(self.lib.poi_timing_get(self.handle, kernel.encode(), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.poi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def context(device=0):
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]
