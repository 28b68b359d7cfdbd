#!/usr/bin/env python
"""Micro-benchmark of the all-POI scoring + top-K kernel (tuning aid): isolates the cost of the
`prob` term and of the top-K filter.  Usage: python tools/bench_score.py [n] [n_item] [dim]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ctx = poi_amd._lib.context(0)
g = torch.Generator(device="cuda").manual_seed(1)
users = (torch.rand((n, D), device="cuda", generator=g) - 0.5).contiguous()
items = (torch.rand((N + 1, D), device="cuda", generator=g) - 0.5).contiguous()
prob = torch.rand((n, N), device="cuda", generator=g).contiguous()
wd = torch.tensor([0.3], device="cuda")
idx = torch.empty((n, 20), dtype=torch.int32, device="cuda")
flops = 2.0 * n * N * D
for label, pw, pp in (("topk, no prob", None, None), ("topk + prob", wd, prob)):
    for it in range(3):
        if it == 1:
            ctx.timing(True)
        ctx.check(ctx.lib.poi_score_topk(ctx.handle, users.data_ptr(), items.data_ptr(), n, N, D,
                                         pw.data_ptr() if pw is not None else None,
                                         pp.data_ptr() if pp is not None else None, 20, idx.data_ptr(), None, None))
    ms, cnt = ctx.timing_get("score_topk")
    ctx.timing(False)
    print("%-16s %8.3f ms/call  %6.1f TFLOP/s  (%.1f%% of f32 MFMA peak)" % (label, ms / cnt, flops / (ms / cnt * 1e-3) / 1e12,
                                                                           100 * flops / (ms / cnt * 1e-3) / 1e12 / 157.3))

full = torch.empty((n, N), dtype=torch.float32, device="cuda")
for it in range(3):
    if it == 1:
        ctx.timing(True)
    ctx.check(ctx.lib.poi_score_all(ctx.handle, users.data_ptr(), items.data_ptr(), n, N, D, None, None, full.data_ptr(), None))
ms, cnt = ctx.timing_get("score_all")
ctx.timing(False)
print("%-16s %8.3f ms/call  %6.1f TFLOP/s  (%.1f%% of f32 MFMA peak)  [writes %.1f GB]" % ("score_all (no topk)", ms / cnt, flops / (ms / cnt * 1e-3) / 1e12,
      100 * flops / (ms / cnt * 1e-3) / 1e12 / 157.3, n * N * 4 / 1e9))
