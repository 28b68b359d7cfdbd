#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: Distance2Pre training epochs (check-in sequences / s) and
all-POI top-20 evaluation (users / s) on synthetic Gowalla-shaped data (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one training epoch over the rank's user shard: ceil(users/B) launches of the batched
training step (B users per launch, batch semantics of include/poi_hip.h) plus, for N > 1, the
per-epoch replica reconciliation (one RCCL all-reduce of the parameter deltas).  Users are sharded
across ranks (total work fixed: strong scaling); every rank holds the full parameter replica.
Inputs are resident in HBM before the timed region.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="gowalla", choices=["tiny", "foursquare", "gowalla"])
    ap.add_argument("--batch-users", type=int, default=12500)
    ap.add_argument("--eval-steps", type=int, default=2)
    ap.add_argument("--eval-chunk", type=int, default=16384,
                    help="users per scoring call: 16384 = 512 user tiles = one workgroup per tile and two per CU with 4 item ranges each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--emulate-world", type=int, default=0, help="tuning aid: train only rank 0's shard of an N-way split on one GPU")
    return ap.parse_args()


def step_flops(D, NB):
    """Algorithmic flops of one GRU step of one sequence, forward + backward:
    54 D^2 for the cell (2D-wide input) + 6 (B+1) D for the distance-softmax head."""
    return 54.0 * D * D + 6.0 * NB * D


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    import poi_amd
    from poi_amd import data as pdata

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:
        dist.init_process_group("nccl", device_id=dev)

    n_item, n_user, max_len, D = pdata.SHAPES[a.shape]
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2)
    lo, hi = pdata.shard_users(n_user, a.emulate_world or world, rank, ds.lens)
    tab = ds.shard(lo, hi)
    n_local = hi - lo
    NB = ds.dist_num + 1
    model = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_local,
                                         n_item=n_item, n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=D, n_hidden=D,
                                         device=dev, seed=7, coords=ds.coords)
    ctx = model.ctx
    sync = poi_amd.dist.model_sync(model, group=None, force=os.environ.get("POI_BENCH_FORCE_SYNC") == "1")

    # shuffled user order (prog_bpr_gru_spatial.py:236-238), cut into launches of B users; inside a launch
    # the ids are sorted by descending length so that 32-sequence tiles are homogeneous.  Resident on device.
    lens_local = np.diff(tab.off.astype(np.int64))
    perm = np.random.default_rng(123).permutation(n_local)
    # launches of equal size, about --batch-users each: a shard of 12600 users is ONE launch, not 12500 + 100
    # (a tiny trailing launch costs the full latency chain of the recurrent kernels)
    n_launch = max(1, int(round(n_local / float(a.batch_users))))
    B = -(-n_local // n_launch)
    batches = []
    for b0 in range(0, n_local, B):
        ids = perm[b0:b0 + B]
        batches.append(ids[np.argsort(-lens_local[ids], kind="stable")])
    order = torch.as_tensor(np.concatenate(batches).astype(np.int32)).to(dev)
    steps_per_epoch = float(np.maximum(lens_local - 1, 0).sum())

    def train_epoch():
        for b0 in range(0, n_local, B):
            model.train_batch(order[b0:b0 + B], sync=False)
        sync.end_epoch()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def rank_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # one untimed pass before the warmup proper: a fresh box pays module loading / first-touch / clock ramp-up
    # on the very first launches (seen as a 6x slower first measurement on small shapes)
    train_epoch()
    torch.cuda.synchronize(dev)
    for _ in range(a.warmup):
        train_epoch()
    ctx.timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        train_epoch()
    barrier()
    dt = rank_max(time.perf_counter() - t0)
    KN = ["seq_train", "te_prep", "te_gather", "te_gemm_ax", "te_rec_fwd", "te_head", "te_rec_bwd", "te_wgrad", "te_gemm_dx",
          "te_finalize", "te_dsum", "te_bin_gemm", "te_scatter", "rows_apply", "dense_apply"]
    kt = {k: ctx.timing_get(k) for k in KN}
    ctx.timing(False)
    seq_per_s = (n_user if not a.emulate_world else n_local) * a.steps / dt

    # ---- evaluation: snapshot -> user vectors -> fused distance term + all-POI score + top-20 -------
    eval_users_per_s = None
    eval_detail = {}
    if not a.no_eval:
        all_ids = np.arange(n_local, dtype=np.int32)
        tes = torch.as_tensor(tab.tes_p.reshape(-1).astype(np.int32)).to(dev)

        def eval_epoch():
            model.update_trained_items(); model.update_trained_dists()
            hts, sts = model.predict_device(all_ids)
            model.update_trained_users(hts); model.update_trained_sus(sts)
            hits = torch.zeros((), dtype=torch.int64, device=dev)
            for c0 in range(0, n_local, a.eval_chunk):
                ids = all_ids[c0:min(c0 + a.eval_chunk, n_local)]
                idx = model.compute_sub_topk(ids, 20)
                hits += (idx == tes[c0:c0 + len(ids), None]).any(dim=1).sum()
            return hits

        ctx.timing(True)
        eval_epoch()                       # warm-up; builds the resident distance-bin matrix (once per data set)
        ms_ulptai = ctx.timing_get("ulptai_build")[0]
        ctx.timing(False)
        ctx.timing(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.eval_steps):
            hits = eval_epoch()
        barrier()
        dte = rank_max(time.perf_counter() - t0)
        ms_score, n_score = ctx.timing_get("score_topk")
        ms_pred = ctx.timing_get("seq_predict")[0] + ctx.timing_get("te_predict")[0]
        ms_dist = ctx.timing_get("dist_prob")[0]
        ctx.timing(False)
        eval_users_per_s = n_user * a.eval_steps / dte
        fl = 2.0 * n_local * n_item * D * a.eval_steps
        eval_detail = {"ms_per_eval": 1e3 * dte / a.eval_steps, "recall_at_20": float(hits.item()) / n_local,
                       "score_topk_tflops": fl / (ms_score * 1e-3) / 1e12 if ms_score > 0 else None,
                       "score_topk_frac_of_f32_mfma_peak": fl / (ms_score * 1e-3) / 1e12 / PEAK_F32_TFLOPS if ms_score > 0 else None,
                       "ms_predict_per_eval": ms_pred / a.eval_steps, "ms_score_topk_per_eval": ms_score / a.eval_steps,
                       "ms_dist_prob_per_eval": ms_dist / a.eval_steps, "ms_ulptai_build_once": ms_ulptai,
                       "eval_chunk_users": a.eval_chunk}

    # ---- roofline of the dominant kernel (live HIP-event timing inside the timed region) -----------
    # algorithmic work per GRU step of one sequence: flops for the contractions, bytes
    # for the gather (3 rows + 4 indices) and the sparse write-back (unique rows per sequence).
    # rows actually written by the sparse write-back: unique table rows per LAUNCH (batch rule: one write per row)
    off64 = tab.off.astype(np.int64)
    order_host = order.cpu().numpy()
    uniq = 0
    for b0 in range(0, n_local, B):
        ids = order_host[b0:b0 + B]
        sel = np.concatenate([np.arange(off64[u], off64[u + 1]) for u in ids])
        uniq += len(np.unique(np.concatenate((tab.p[sel], tab.q[sel])))) + 1 + len(np.unique(np.append(tab.dp[sel], ds.dist_num)))
    D2 = float(D * D)
    # EXECUTED flops per kernel (the roofline fractions are matrix-pipe utilisation).  At D >= 128 the distance-bin
    # half of the step input goes through per-bin tables: te_gemm_ax / te_gemm_dx / the d ui jobs of
    # te_wgrad only multiply the POI half, i.e. 36 D^2 + 6 NB D executed against the 54 D^2 + 6 NB D of the
    # reference formulation (step_flops)
    bintab = D >= 128
    xk = 6 if bintab else 12
    work = {"seq_train": ("flop", step_flops(D, NB) * steps_per_epoch),
            "te_gemm_ax": ("flop", xk * D2 * steps_per_epoch), "te_rec_fwd": ("flop", 6 * D2 * steps_per_epoch),
            "te_head": ("flop", 4.0 * NB * D * steps_per_epoch), "te_rec_bwd": ("flop", 6 * D2 * steps_per_epoch),
            "te_wgrad": ("flop", ((6 + xk) * D2 + 2.0 * NB * D) * steps_per_epoch),     # d ui, d wh and d vs (split-K)
            "te_gemm_dx": ("flop", xk * D2 * steps_per_epoch),
            # te_gather now only builds E = lt[p'] - lt[q'] (two table rows + two indices per step); the gather of the
            # step input [lt[p] | di[dp]] is fused into te_gemm_ax / te_wgrad (rows go straight into their LDS tiles)
            "te_gather": ("byte", (2.0 * D * 4 + 8) * steps_per_epoch),
            "rows_apply": ("byte", 2.0 * uniq * D * 4.0),      # read + write of every touched row
            # sorted scatter: per step dx (2D floats; bintab: D floats + the 3D floats of DA for the per-bin sums)
            # + g*h (D floats) in, every touched row read + written
            "te_scatter": ("byte", (2.0 if bintab else 3.0) * D * 4 * steps_per_epoch + 2.0 * uniq * D * 4.0),
            # per-bin sums of DA (bintab): one read of the 3D-wide DA rows
            "te_dsum": ("byte", 3.0 * D * 4 * steps_per_epoch)}
    kernels = {}
    for k in KN:
        ms, nl = kt[k]
        if nl == 0:
            continue
        # per-step time from the per-launch average: the event recorder keeps at most 8192 regions, so on long
        # runs (--steps > ~180) only the first launches are timed - their average is still the launch duration
        per_step = (ms / nl) * len(batches)
        ent = {"ms_per_step": per_step, "launches": nl, "avg_ms": ms / nl}
        if k in work:
            kind, w = work[k]
            rate = w / (per_step * 1e-3)
            if kind == "flop":
                ent.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=rate / 1e12 / PEAK_F32_TFLOPS)
            else:
                ent.update(bound="hbm", achieved=rate / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=rate / 1e9 / PEAK_HBM_GBS)
        if k == "te_finalize":
            ent["note"] = "runs on the side stream next to te_wgrad / te_gemm_dx: its span overlaps them and is not part of the serial sum"
        kernels[k] = ent
    dom = max((k for k in kernels if "bound" in kernels[k]), key=lambda k: kernels[k]["ms_per_step"])
    # HBM traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    # runs, gfx950 correction applied - profiles/r01f_pmc_traffic.json); only valid for the profiled workload
    traffic = {}
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", "r01f_pmc_traffic.json")))
        if a.shape == "gowalla" and B == 12500 and world == 1:
            traffic = {k: v["hbm_bytes_per_launch"] for k, v in pj["kernels"].items()}
    except Exception:
        pass
    for k in kernels:
        if k in traffic:
            kernels[k]["traffic_bytes_per_launch"] = traffic[k]
    roofline = dict(kernel=dom, traffic=traffic.get(dom), **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_ms")})
    roofline["note"] = "f32 arithmetic on v_mfma_f32_32x32x2_f32 (f32 vector peak == f32-input MFMA peak on gfx950)"
    GS = ("te_gather", "te_dsum", "te_scatter", "rows_apply")
    gs_ms = sum(kernels[k]["ms_per_step"] for k in GS if k in kernels)
    gs_bytes = sum(work[k][1] for k in GS if k in kernels)
    hbm = {"kernels": [k for k in GS if k in kernels], "bound": "hbm",
           "achieved": gs_bytes * a.steps / (gs_ms * a.steps * 1e-3) / 1e9 if gs_ms > 0 else 0.0, "peak": PEAK_HBM_GBS, "unit": "GB/s",
           "traffic": sum(traffic.get(k, 0) for k in GS) or None}
    hbm["frac"] = hbm["achieved"] / PEAK_HBM_GBS
    total_flops = step_flops(D, NB) * steps_per_epoch
    executed_flops = sum(w for k, (kind, w) in work.items() if kind == "flop" and k in kernels and k != "seq_train") or total_flops
    train_kernel_ms = sum(kernels[k]["ms_per_step"] for k in kernels if k != "te_finalize")

    # ---- CPU baseline: plain-C float64 port of the same per-sequence algorithm, 1 thread -------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import c_oracle as C
        from oracle import poi_oracle as O
        rng = np.random.default_rng(7)
        P = O.init_spatial_params(rng, n_item, ds.dist_num, D)
        ordr = perm.astype(np.int32)
        t0 = time.perf_counter()
        C.spatial_epoch(P, tab.off, tab.p, tab.q, tab.dp, tab.dq, ordr[:8], tab.len_max, 0.01, 0.001)
        per = (time.perf_counter() - t0) / 8
        S = int(min(max(a.cpu_seconds / max(per, 1e-6), 16), 4000, n_local))
        t0 = time.perf_counter()
        C.spatial_epoch(P, tab.off, tab.p, tab.q, tab.dp, tab.dq, ordr[:S], tab.len_max, 0.01, 0.001)
        tc = time.perf_counter() - t0
        ne = 16
        t0 = time.perf_counter()
        C.score_topk(rng.uniform(-0.5, 0.5, (ne, D)), P["lt"][:-1], 20)
        te = time.perf_counter() - t0
        # all host cores, user-sharded: one independent replica per core, each running the same
        # sequential per-user SGD over its own slice (ctypes releases the GIL during the C call)
        ncore = os.cpu_count() or 1
        all_cores = None
        if ncore > 1:
            from concurrent.futures import ThreadPoolExecutor
            Sc = max(16, min(S, n_local // ncore))
            reps = [{k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in P.items()} for _ in range(ncore)]

            def work(i):
                C.spatial_epoch(reps[i], tab.off, tab.p, tab.q, tab.dp, tab.dq, ordr[i * Sc:(i + 1) * Sc], tab.len_max, 0.01, 0.001)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(ncore) as ex:
                list(ex.map(work, range(ncore)))
            ta = time.perf_counter() - t0
            all_cores = {"value": ncore * Sc / ta, "unit": "sequences/s", "cores": ncore,
                         "sample": "%d user-sharded replicas x %d sequences" % (ncore, Sc)}
        cpu = {"value": S / tc, "unit": "sequences/s", "cores": 1, "kind": "port", "all_cores": all_cores,
               "sample": "%d sequences of the same shuffled order, sequential per-user SGD (reference semantics), "
                         "plain-C float64 port of public/GRU_Spatial.py:127-229 (Theano cannot be built or shipped)" % S,
               "eval_users_per_s": ne / te, "host_cores_available": os.cpu_count()}

    if rank == 0:
        out = {
            "metric": "check-in sequences/sec training (Distance2Pre) + all-POI top-K eval users/sec",
            "value": seq_per_s, "unit": "sequences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic %s-shape: %d POIs, %d users, seq<=%d, dim=%d, %d distance bins; one step = one "
                                   "Distance2Pre training epoch over all users" % (a.shape, n_item, n_user, max_len, D, ds.dist_num),
                       "batch_users_per_launch": B, "parallelism": "user-shard x%d, per-epoch delta all-reduce" % world,
                       "alpha": 0.01, "lambda": 0.001, "engine": "tile" if "te_rec_fwd" in kernels else "per-sequence"},
            "eval_users_per_s": eval_users_per_s, "eval": eval_detail,
            "roofline": roofline, "roofline_gather_scatter": hbm, "kernels": kernels,
            "train_step_tflops": executed_flops / (train_kernel_ms * 1e-3) / 1e12 if train_kernel_ms > 0 else None,
            "train_step_tflops_reference_formulation": total_flops / (train_kernel_ms * 1e-3) / 1e12 if train_kernel_ms > 0 else None,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
