"""Worker of tests/test_gpu_multi.py::test_half_table_reconciliation_across_two_processes (run under torch.distributed.run, 2 ranks, gloo,
both on cuda:0): a 1 GB half table (4 M x 128) + a float32 tensor, each rank moves its own rows, ReplicaSync reconciles through the
library's delta / combine kernels around REAL cross-process all-reduces of the float32 buffer (touch counts) and of the 1 GB half buffer
(256 MB slices); every rank checks the result against the closed form and rank 0 prints a JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import poi_amd  # noqa: E402

ROWS, W = 4 * 1024 * 1024, 128


def moved_rows(rank):
    r = torch.arange(ROWS, device="cuda")
    return (r % 3 == rank) | (r % 7 == 0)              # rows % 21 in {0, 7, 14} x ... : some rows moved by both ranks, some by one, most by none or one


def delta_of(rank):
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    d = (torch.rand(ROWS, W, device="cuda", generator=g) - 0.5) * 0.02
    return d * moved_rows(rank)[:, None]


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    g = torch.Generator(device="cuda").manual_seed(7)
    tab0 = (torch.rand(ROWS, W, device="cuda", generator=g) - 0.5).half()
    dense0 = torch.rand(256, 128, device="cuda", generator=g)
    cur = [tab0.clone(), dense0.clone()]
    ctx = poi_amd._lib.context(0)
    sync = poi_amd.dist.ReplicaSync(cur, rules=["mean_touched", "mean"], ctx=ctx, own_comm=False)
    assert sync.active and not sync.own_comm and sync.backend.flat16 is not None and sync.backend.flat16.numel() * 2 >= (1 << 30)
    new = [(tab0.float() + delta_of(r)).half() for r in range(2)]
    cur[0].copy_(new[rank]); cur[1].copy_(dense0 + 0.125 * (rank + 1))
    sync.end_epoch()
    torch.cuda.synchronize()
    d16 = [(new[r].float() - tab0.float()).half() for r in range(2)]
    cnt = sum((new[r] != tab0).any(dim=1).float() for r in range(2))
    ssum = (d16[0].float() + d16[1].float()).half().float()           # gloo adds the two half deltas in half
    exp = (tab0.float() + ssum * (1.0 / cnt.clamp(min=1.0))[:, None]).half()
    diff = (cur[0].float() - exp.float()).abs()
    ulp = torch.as_tensor(np.spacing(exp.abs().cpu().numpy())).cuda().float()
    ok_tab = bool((diff <= 1.0001 * ulp).all()) and float((diff == 0).float().mean()) > 0.999
    untouched = cnt == 0
    ok_untouched = torch.equal(cur[0][untouched], tab0[untouched])
    ok_dense = torch.allclose(cur[1], dense0 + 0.1875, rtol=0, atol=2e-7)
    rep = sync.report()
    res = {"rank": rank, "ok_table": ok_tab, "ok_untouched": ok_untouched, "ok_dense": ok_dense, "checksums_equal": rep["replica_checksums_equal"],
           "allreduce_bytes": rep["allreduce_bytes"], "rows_moved_by_both": int((cnt == 2).sum()), "rows_moved_by_one": int((cnt == 1).sum())}
    allres = [None, None]
    dist.all_gather_object(allres, res)
    if rank == 0:
        print(json.dumps(allres), flush=True)
    sync.close()
    dist.destroy_process_group()
    sys.exit(0 if all(r["ok_table"] and r["ok_untouched"] and r["ok_dense"] and r["checksums_equal"] for r in allres) else 4)


if __name__ == "__main__":
    main()
