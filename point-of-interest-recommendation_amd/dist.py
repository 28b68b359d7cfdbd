"""Multi-GPU path: users are sharded across ranks, every rank holds a full replica of
the parameters, and replicas are reconciled ONCE PER EPOCH by summing each rank's delta relative to
the epoch-start snapshot (one all-reduce over xGMI per tensor group; RCCL via torch.distributed):

        theta <- theta_start + sum_r (theta_r - theta_start)

With world_size == 1 this is the identity.  No collective touches the per-step data path.
The elementwise delta arithmetic is the HIP kernels poi_delta_make / poi_delta_apply; tests inject
their own arithmetic to exercise the protocol under gloo on CPU tensors."""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist


def _hip_delta_ops(ctx, stream_fn):
    def make(cur, base, out):
        ctx.check(ctx.lib.poi_delta_make(ctx.handle, cur.data_ptr(), base.data_ptr(), out.data_ptr(), cur.numel(), stream_fn()))

    def apply(cur, base, dsum):
        ctx.check(ctx.lib.poi_delta_apply(ctx.handle, cur.data_ptr(), base.data_ptr(), dsum.data_ptr(), cur.numel(), stream_fn()))
    return make, apply


class ReplicaSync:
    """Keeps the epoch-start snapshot of a list of parameter tensors and reconciles replicas."""

    def __init__(self, tensors, group=None, delta_ops=None, ctx=None, force=False):
        self.tensors = list(tensors)
        if delta_ops is None and any(t.dtype != torch.float32 for t in self.tensors):
            raise TypeError("ReplicaSync with the HIP delta kernels needs float32 tensors")
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = bool(force) and dist.is_initialized()      # run the full delta/all-reduce path even at world size 1 (self-check)
        if delta_ops is None:
            if ctx is None:
                raise ValueError("ReplicaSync needs a poi context (HIP delta kernels) or explicit delta_ops")
            dev = self.tensors[0].device
            delta_ops = _hip_delta_ops(ctx, lambda: ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        self.make, self.apply = delta_ops
        # one flat buffer for all tensors: a single large all-reduce instead of many small ones
        n = sum(t.numel() for t in self.tensors)
        self.base = torch.empty(n, dtype=self.tensors[0].dtype, device=self.tensors[0].device)
        self.delta = torch.empty_like(self.base)
        self._views = []
        o = 0
        for t in self.tensors:
            self._views.append((o, t.numel()))
            o += t.numel()
        self.begin_epoch()

    def begin_epoch(self):
        for t, (o, n) in zip(self.tensors, self._views):
            self.base[o:o + n].copy_(t.reshape(-1))

    def end_epoch(self):
        """All-reduce the deltas and rebuild every replica; then start the next epoch's snapshot."""
        if self.world > 1 or self.force:
            for t, (o, n) in zip(self.tensors, self._views):
                self.make(t.reshape(-1), self.base[o:o + n], self.delta[o:o + n])
            dist.all_reduce(self.delta, op=dist.ReduceOp.SUM, group=self.group)
            for t, (o, n) in zip(self.tensors, self._views):
                self.apply(t.reshape(-1), self.base[o:o + n], self.delta[o:o + n])
        self.begin_epoch()
