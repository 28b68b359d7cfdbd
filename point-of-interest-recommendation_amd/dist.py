"""Multi-GPU path: users are sharded across ranks, every rank holds a full replica of the
parameters and trains its shard with no data-path collective; replicas are reconciled ONCE PER EPOCH:

        theta <- theta_start + combine( sum_r (theta_r - theta_start) )

one all-reduce over xGMI of ONE flat buffer.  The arithmetic (delta, per-row touch flags, combine rule, next
epoch's snapshot) and the collective itself live in libpoi_hip.so (include/poi_hip.h: poi_sync_*, poi_comm_*,
poi_allreduce_tables - the library owns an RCCL communicator); this module only wires them to the model's
tensors and, for the communicator's one-time rendezvous, broadcasts the 128-byte RCCL id over torch.distributed.

Combine rules (per tensor; identity at world size 1), see DESIGN.md "Multi-GPU":
  sum            every replica's epoch counts in full - to first order in alpha the sequential epoch of ONE process
                 over all users (what the reference's loop does, prog_bpr_gru_spatial.py:249-250)
  mean           model averaging (local SGD)
  mean_touched   per table row, the mean over the replicas that moved it (the launch-level batch rule one level up)
DEFAULT_RULES is what bench.py and the harness use.

Tests drive the same protocol on CPU tensors under gloo with a host backend (tests/test_dist_gloo.py)."""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib

RULES = {"sum": _lib.SYNC_SUM, "mean": _lib.SYNC_MEAN, "mean_touched": _lib.SYNC_MEAN_TOUCHED}
# Evidence (tools/quality.py --world 8, N replicas emulated on one GPU, Gowalla shape, cap 64, 120 epochs; DESIGN.md "Multi-GPU"):
# "sum" on everything diverges (a hot row already moved by up to `cap` updates inside every replica's launch moves 8 x that),
# lt "sum" + dense "mean" learns slowest (recall@20 0.27), "mean" on everything 0.45, against 0.55 for one GPU with four launches
# per epoch.  So: the POI table takes the mean over the replicas that MOVED the row (a row only one shard's users visit keeps
# its whole update, hot rows are averaged - the launch-level batch rule one level up), everything every replica moves in
# every launch (distance-bin rows, dense weights) is averaged (model averaging, as in local SGD).
DEFAULT_RULES = {"lt": "mean_touched", "ux": "mean_touched", "wd_table": "mean_touched", "di": "mean", "ui": "mean", "wh": "mean", "bi": "mean",
                 "vs": "mean", "bs": "mean", "wd": "mean", "loss_weight": "mean", "M": "mean"}


class _DevArray:
    """Zero-copy torch view of a device buffer owned by the library (__cuda_array_interface__)."""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class HipSyncBackend:
    """poi_sync_* on the model's device tensors (+ the library's own RCCL communicator when asked for)."""

    def __init__(self, tensors, rules, ctx, device):
        self.ctx, self.lib, self.device = ctx, ctx.lib, device
        self.tensors = list(tensors)              # keep them alive: the library holds raw pointers
        for t in self.tensors:
            if t.dtype not in (torch.float32, torch.float16) or not t.is_contiguous() or t.device != device:
                raise TypeError("ReplicaSync needs contiguous float32 (or float16-stored table) tensors on %s" % device)
        segs = (_lib.SyncSeg * len(self.tensors))()
        for s, t, r in zip(segs, self.tensors, rules):
            width = t.shape[-1] if t.dim() >= 1 and t.shape[-1] > 0 else 1
            if r == "mean_touched" and t.dim() > 2:
                width = t.numel() // t.shape[0]          # a table of matrices (CA-RNN's interval matrices): one matrix = one row
            s.cur, s.rows, s.width, s.rule = t.data_ptr(), t.numel() // width, width, RULES[r]
            s.dtype = 1 if t.dtype == torch.float16 else 0
        h = ctypes.c_void_p()
        self._check(self.lib.poi_sync_create(ctx.handle, device.index, segs, len(self.tensors), ctypes.byref(h)))
        self.handle = h
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._check(self.lib.poi_sync_buffer(h, ctypes.byref(p), ctypes.byref(n)))
        self.n = n.value
        self.flat = torch.as_tensor(_DevArray(p.value, n.value), device=device)
        # tables stored as half: their deltas travel as half in a second flat buffer (poi_sync_buffer16)
        p16, n16 = ctypes.c_void_p(), ctypes.c_int64()
        self._check(self.lib.poi_sync_buffer16(h, ctypes.byref(p16), ctypes.byref(n16)))
        self.n16 = n16.value
        self.flat16 = torch.as_tensor(_DevArray(p16.value, n16.value, "<f2"), device=device) if n16.value else None
        self.comm = None

    def _check(self, rc):
        if rc != 0:
            raise _lib.PoiError("libpoi_hip sync error %d: %s" % (rc, self.lib.poi_sync_last_error().decode()))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def init_comm(self, group=None):
        """The library's own RCCL communicator: rank 0 draws the id, one broadcast hands it to the others.  Returns True when every rank
        holds a communicator, False when the ranks AGREED not to build one.  The collective sequence is the same on every rank whatever
        fails locally: (1) every rank probes that the library can bind librccl - rank 0 by drawing the id (poi_comm_unique_id), the others
        with poi_comm_available, which makes no RCCL call (ncclGetUniqueId on a non-root rank would leave an unused bootstrap listener
        behind), (2) ONE broadcast of the id (zeros if rank 0 could not draw it), (3) ONE all-reduce (MIN) of the probe results, and only
        if all ranks passed (4) the collective poi_comm_init_rank, followed by (5) a second MIN all-reduce of its outcome.
        Caveat: (4) is RCCL's own rendezvous - a rank whose ncclCommInitRank fails FAST leaves the others inside theirs until RCCL's
        bootstrap timeout; the agreement of (5) is only reached after that."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        on_dev = dist.get_backend(group) == "nccl"
        buf = (ctypes.c_char * 128)()
        ok = 1
        try:
            self._check(self.lib.poi_comm_unique_id(buf) if rank == 0 else self.lib.poi_comm_available())
        except Exception:      # noqa: BLE001 - agreed on across ranks below
            ok = 0
            buf = (ctypes.c_char * 128)()
        t = torch.tensor(list(buf.raw), dtype=torch.uint8)
        t = t.to(self.device) if on_dev else t
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        flag = torch.tensor([ok], dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            return False
        raw = bytes(t.cpu().tolist())
        h = ctypes.c_void_p()
        rc = self.lib.poi_comm_init_rank(raw, world, rank, self.device.index, ctypes.byref(h))
        flag = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if rc == 0 and h:
                self.lib.poi_comm_destroy(h)
            return False
        self.comm = h
        return True

    def begin_epoch(self):
        self._check(self.lib.poi_sync_begin_epoch(self.handle, self._stream()))

    def make_delta(self):
        self._check(self.lib.poi_sync_make_delta(self.handle, self._stream()))
        return self.flat

    def apply(self, world):
        self._check(self.lib.poi_sync_apply(self.handle, int(world), self._stream()))

    def end_epoch_rccl(self):
        self._check(self.lib.poi_sync_end_epoch(self.handle, self.comm, self._stream()))

    def stats(self):
        ms, nb = ctypes.c_double(), ctypes.c_int64()
        self._check(self.lib.poi_sync_stats(self.handle, ctypes.byref(ms), ctypes.byref(nb)))
        return ms.value, nb.value

    def checksum(self):
        """64-bit checksum over all synchronised tensors (bit-identical replicas <=> equal checksums)."""
        acc = torch.zeros(1, dtype=torch.int64, device=self.device)
        for t in self.tensors:
            self._check(self.lib.poi_checksum(self.ctx.handle, t.data_ptr(), t.numel() * t.element_size() // 4, acc.data_ptr(), self._stream()))
        return int(acc.item())

    def close(self):
        if getattr(self, "handle", None):
            self.flat = None; self.flat16 = None
            self.lib.poi_sync_destroy(self.handle); self.handle = None
        if getattr(self, "comm", None):
            self.lib.poi_comm_destroy(self.comm); self.comm = None


class ReplicaSync:
    """Per-epoch replica reconciliation of a list of parameter tensors.

    tensors / rules : the tensors (updated in place) and one rule name per tensor ("sum" | "mean" | "mean_touched")
    backend         : object with begin_epoch() / make_delta() -> flat tensor / apply(world); default = the HIP
                      kernels of libpoi_hip.so (needs ctx=); tests pass a host backend
    own_comm        : all-reduce through the library's own RCCL communicator (poi_allreduce_tables) instead of
                      torch.distributed.all_reduce on the same buffer (default: when the backend is nccl)
    force           : run the whole delta / all-reduce / apply path even at world size 1 (self-check)"""

    def __init__(self, tensors, rules=None, group=None, backend=None, ctx=None, force=False, own_comm=None, names=None, force_backend=False):
        self.tensors = list(tensors)
        self.names = list(names) if names is not None else ["t%d" % i for i in range(len(self.tensors))]
        rules = list(rules) if rules is not None else ["sum"] * len(self.tensors)
        if len(rules) != len(self.tensors) or any(r not in RULES for r in rules):
            raise ValueError("one rule of %s per tensor" % sorted(RULES))
        self.rules = rules
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()
        self.active = self.world > 1 or self.force
        self._ctx = ctx
        if backend is None and not self.active and not force_backend:
            backend = None                    # world size 1: no snapshot / delta buffers (2 x the parameter bytes) are allocated
        elif backend is None:
            if ctx is None:
                raise ValueError("ReplicaSync needs ctx= (HIP kernels of libpoi_hip.so) or an explicit backend")
            backend = HipSyncBackend(self.tensors, rules, ctx, self.tensors[0].device)
            if own_comm is None:
                own_comm = self.active and dist.get_backend(group) == "nccl"
            if own_comm and self.active:
                # every rank ends up on the SAME collective: init_comm runs one fixed sequence of collectives and returns the ranks'
                # common verdict; without a communicator all of them use torch.distributed.all_reduce on the same buffers
                if not backend.init_comm(group):
                    import warnings
                    warnings.warn("the library's RCCL communicator could not be built on at least one rank: replica reconciliation falls back to "
                                  "torch.distributed.all_reduce")
                    backend.comm = None
        if isinstance(backend, HipSyncBackend) and own_comm is None:
            own_comm = False
        self.backend = backend
        self.own_comm = bool(own_comm) and getattr(backend, "comm", None) is not None
        self.epochs = 0
        if self.active:
            self.begin_epoch()

    def begin_epoch(self):
        self.backend.begin_epoch()

    def end_epoch(self):
        """All-reduce the deltas and rebuild every replica (the result is the next epoch's snapshot)."""
        if not self.active:
            return
        if self.own_comm:
            self.backend.end_epoch_rccl()
        else:
            flat = self.backend.make_delta()
            self._all_reduce_chunked(flat)
            flat16 = getattr(self.backend, "flat16", None)
            if flat16 is not None:                      # half-stored tables: their deltas, as half
                self._all_reduce_chunked(flat16)
            self.backend.apply(self.world)
        self.epochs += 1

    CHUNK_BYTES = 256 << 20

    def _all_reduce_chunked(self, flat):
        """One all-reduce per 256 MB slice of the flat buffer (config X's half table is 5 GB per replica: a single call would need the
        collective's staging for all of it at once, and a slice that has been reduced can be consumed while the next one is in flight)."""
        n = flat.numel(); step = max(1, self.CHUNK_BYTES // flat.element_size())
        for o in range(0, n, step):
            dist.all_reduce(flat[o:o + step], op=dist.ReduceOp.SUM, group=self.group)

    def report(self):
        """Self-validation block for bench.py: what the collective saw and whether the replicas agree bit for bit."""
        out = {"world_size": self.world, "rules": dict(zip(self.names, self.rules)),
               "collective": "rccl (library communicator, poi_allreduce_tables)" if self.own_comm else
               ("torch.distributed.all_reduce" if self.active else "none (world size 1)"), "epochs_synced": self.epochs}
        if self.backend is None and self._ctx is not None:
            out["allreduce_bytes"] = 0
            acc = torch.zeros(1, dtype=torch.int64, device=self.tensors[0].device)
            for t in self.tensors:
                self._ctx.check(self._ctx.lib.poi_checksum(self._ctx.handle, t.data_ptr(), t.numel() * t.element_size() // 4, acc.data_ptr(),
                                                           ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))
            out["replica_checksums_equal"] = True
            out["replica_checksum"] = "%016x" % (int(acc.item()) & 0xFFFFFFFFFFFFFFFF)
        if isinstance(self.backend, HipSyncBackend):
            if self.own_comm and self.epochs:
                ms, nb = self.backend.stats()
                out["allreduce_ms_last"], out["allreduce_bytes"] = ms, nb
                out["rccl_world_size"] = self.backend.lib.poi_comm_world(self.backend.comm)
            else:
                out["allreduce_bytes"] = self.backend.n * 4 + self.backend.n16 * 2
        if hasattr(self.backend, "checksum"):
            cs = self.backend.checksum()
            if dist.is_initialized():
                t = torch.tensor([cs], dtype=torch.int64, device=self.tensors[0].device)
                lst = [torch.zeros_like(t) for _ in range(self.world)]
                dist.all_gather(lst, t, group=self.group)
                allcs = [int(x.item()) for x in lst]
            else:
                allcs = [cs]
            out["replica_checksums_equal"] = len(set(allcs)) == 1
            out["replica_checksum"] = "%016x" % (allcs[0] & 0xFFFFFFFFFFFFFFFF)
            out["world_size_seen_by_all_gather"] = len(allcs)
        return out

    def close(self):
        if hasattr(self.backend, "close"):
            self.backend.close()


def model_sync(model, names=None, rules=None, **kw):
    """ReplicaSync over EVERY trainable tensor of a model with DEFAULT_RULES (override per name with rules={...}).  The names come
    from the model class (`sync_names`: the tables + model.params - CA-RNN: lt, wd, M) or, failing that, from the attributes the
    model has; a `wd` that is a table of matrices (CA-RNN) takes the "wd_table" rule, the Distance2Pre scalar the "wd" rule."""
    names = names or getattr(model, "sync_names", None) or \
        [n for n in ("lt", "di", "ux", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight", "M") if hasattr(getattr(model, n, None), "t")]
    names = list(names)
    r = dict(DEFAULT_RULES); r.update(rules or {})
    key = lambda n: "wd_table" if (n == "wd" and getattr(model, n).t.dim() >= 3 and "wd" not in (rules or {})) else n
    return ReplicaSync([getattr(model, n).t for n in names], rules=[r[key(n)] for n in names], ctx=model.ctx, names=names, **kw)
