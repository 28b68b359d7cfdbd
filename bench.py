#!/usr/bin/env python
"""Benchmark of the hot path on MI355X: Distance2Pre training epochs (check-in sequences / s) and all-POI top-20
evaluation (users / s) on synthetic Gowalla-shaped data (BASELINE.json configs[2]).

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one training epoch over the rank's user shard: ceil(users/B) launches of the batched training step
(B users per launch, batch rule of include/poi_hip.h with --batch-cap) plus, for N > 1, the per-epoch replica
reconciliation (one RCCL all-reduce of the parameter deltas through the library's own communicator).  Users are
sharded across ranks (total work fixed: strong scaling); every rank holds the full parameter replica.  Inputs are
resident in HBM before the timed region.  Rank 0 prints ONE compact JSON line (emit(): the contract fields, roofline,
roofline_gather_scatter, cpu_baseline, headline - under 6 KB) as the last line of stdout and writes the FULL record to --full-out
(default gpurun_out/bench_full.json).  Map of main(): setup + timed region -> exact_mode -> evaluation -> per-kernel work model and
roofline blocks -> quality / reference schedule -> secondary shapes (foursquare, dd25, x1 subprocess, BPR-MF) -> launch_sweep ->
multi_gpu.projection -> cpu_baseline -> emit.  Besides the contract fields the full record carries
  roofline / roofline_gather_scatter / kernels   live HIP-event timings of the timed region against gfx950 peaks
  reference_schedule                             the reference's own schedule (one user per step) on the same GPU
  quality                                        recall@20 / AUC after a fixed training wall time: headline mode vs the
                                                 reference schedule, on the same data (which has a next-POI signal)
  multi_gpu                                      what the collective saw + replica checksum equality (self-validating)
  secondary                                      Foursquare-shape numbers (BASELINE.json configs[1])
  cpu_baseline                                   plain-C float64 port of the reference step on the host cores
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
PEAK_BF16_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PEAK_F64_TFLOPS = 78.6      # float64 vector rate = float64 MFMA rate on gfx950 (half the float32 vector peak of MI355X_MICROARCH.md)
PEAK_I8_TOPS = 5000.0       # MI355X_MICROARCH.md / cdna_hip_programming.md: int8 MFMA = 2 x the 2.5 PF bf16 dense peak (measured 3.9 - 4.4 POP/s)
PROFILE_TAG = "r06"         # profiles/<tag>_pmc_traffic.json: committed rocprofv3 PMC passes of this command


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed training epochs (default: a >= 2 s timed window)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shape", default="gowalla", choices=["tiny", "foursquare", "gowalla", "x1"],
                    help="x1 = one GPU's slice of BASELINE.json configs[4]: 10 M POIs, 125 k users (1 M / 8), dim 256, fp16 POI table")
    ap.add_argument("--table-dtype", default=None, choices=["f32", "f16"],
                    help="storage type of the POI table (arithmetic is float32 either way); default f32, x1: f16 as BASELINE.json configs[4] says")
    ap.add_argument("--eval-users", type=int, default=0, help="score only the first N users of the shard in the evaluation passes (0 = all; x1 default 8192)")
    ap.add_argument("--batch-users", type=int, default=12500)
    ap.add_argument("--batch-cap", type=float, default=64.0,
                    help="batch rule cap (poi_ctx_set_batch_cap): a row touched by k sequences of a launch moves by min(k, cap)/k x the sum of "
                         "their reference updates; 1 = mean rule.  64 is the setting whose recall matches the reference schedule (quality block)")
    ap.add_argument("--local", type=float, default=0.8, help="fraction of check-in transitions that go to one of the 32 nearest POIs (0: i.i.d. Zipf draws)")
    ap.add_argument("--dd", type=float, default=200.0, help="distance-bin width in metres (the reference's other configuration: 25 with --ud-km 38 = 1520 bins)")
    ap.add_argument("--ud-km", type=float, default=40.0, help="distance beyond which every POI falls into the last bin")
    ap.add_argument("--eval-steps", type=int, default=5)
    ap.add_argument("--eval-chunk", type=int, default=65536,
                    help="users per scoring call (seeded calls take the two-stage path, which is at its best with all users in one call: 4.4 ms per "
                         "50 k users against 4.7 in 16384-user calls; the one-stage kernel preferred 16384 = 512 user tiles, two workgroups per CU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-refresh", action="store_true",
                    help="replay ONE shuffled order and ONE negative table for every epoch (rounds 1 - 3).  Default: every epoch redraws its negatives on the "
                         "device and takes a new shuffled order, as the reference driver does (prog_bpr_gru_spatial.py:221-238)")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-engine (float64) timing")
    ap.add_argument("--no-x1", action="store_true", help="skip the secondary_x1 block (one GPU's slice of BASELINE.json configs[4], training + one 8192-user evaluation, a subprocess of ~20 s)")
    ap.add_argument("--f16-rounding", default=None, choices=["nearest", "stochastic"],
                    help="write-back rounding of a half POI table (poi_ctx_set_f16_rounding); default: stochastic for --table-dtype f16 - keeps the L2 decay")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--quality-seconds", type=float, default=3.0, help="training wall time of the batched modes in the quality block")
    ap.add_argument("--reference-seconds", type=float, default=15.0, help="training wall time of the reference schedule (one user per step)")
    ap.add_argument("--no-bpr", action="store_true", help="skip the secondary_bpr block (BPR-MF step at the gowalla shape)")
    ap.add_argument("--no-projection", action="store_true", help="skip the multi_gpu.projection block (N = 2 / 4 / 8 shards emulated on this GPU)")
    ap.add_argument("--emulate-world", type=int, default=0, help="tuning aid: train only rank 0's shard of an N-way split on one GPU")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="testing aid: gloo + --same-device runs the whole N > 1 path (sharding, schedules, reconciliation, self-check) with several ranks on ONE GPU")
    ap.add_argument("--same-device", action="store_true", help="testing aid: every rank uses cuda:0 (with --dist-backend gloo)")
    ap.add_argument("--full-out", default=None,
                    help="path of the FULL record (every block; ~25 KB of JSON).  Default gpurun_out/bench_full.json.  stdout's last line is the compact record")
    ap.add_argument("--replica-schedule", default="quality", choices=["quality", "throughput"],
                    help="N > 1: launches per replica and epoch - as many as the one-GPU run (quality, default) or ~--batch-users users each (plan_shard)")
    return ap.parse_args()


def full_out_path(a):
    if a.full_out:
        return os.path.abspath(a.full_out)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, "bench_full.json")


def emit(a, out):
    """The FULL record goes to a file (--full-out); stdout's LAST line is a compact record (< 4 KB) with every contract field, the dominant
    kernel's roofline, the gather / scatter roofline, the CPU baseline and the headline numbers - a driver that keeps only a few KB of the tail
    still parses it (the 23.5 KB single line of round 5 was cut: BENCH_r05.json parsed = null)."""
    path = full_out_path(a)
    with open(path, "w") as f:
        json.dump(out, f)
    r = out["roofline"]; g = out["roofline_gather_scatter"]; c = out.get("cpu_baseline")
    sub = lambda d, ks: None if d is None else {k: d.get(k) for k in ks if k in d}
    cfg = out["config"]
    compact = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    compact["config"] = {"workload": cfg["workload"], "batch_users_per_launch": cfg["batch_users_per_launch"], "batch_rule": "capped sum, cap %g" % cfg["batch_cap"],
                         "parallelism": cfg["parallelism"], "replica_schedule": cfg.get("replica_schedule"), "table_storage": out.get("table_storage")}
    compact["roofline"] = sub(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches", "avg_ms"))
    compact["roofline_gather_scatter"] = {
        "bound": "hbm", "kernels": g["kernels"], "ms_per_epoch": g["ms_per_epoch"], "achieved": g["achieved"], "peak": g["peak"], "unit": g["unit"], "frac": g["frac"],
        "frac_survey_8d": (g.get("survey_8d") or {}).get("frac"), "frac_bytes_moved": (g.get("implementation") or {}).get("frac"),
        "frac_embedding_rows_only": (g.get("embedding_rows_only") or {}).get("frac"), "bytes_per_epoch_survey_8d": (g.get("survey_8d") or {}).get("bytes_per_epoch"),
        "traffic": g.get("traffic"), "sort_ms_hidden": g.get("sort_ms_hidden"), "hot_reduce_ms_hidden": g.get("hot_reduce_ms_hidden"),
        "frac_survey_8d_with_fused_gather_charged": (g.get("fused_gather") or {}).get("frac_survey_8d_with_it"),
        "frac_survey_8d_with_fused_gather_and_hidden_hot_rows_charged": g.get("frac_survey_8d_with_fused_gather_and_hidden_hot_rows_charged"), "x1": g.get("x1")}
    if c:
        compact["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": c["sample"][:160],
                                   "all_cores": sub(c.get("all_cores"), ("value", "cores"))}
    else:
        compact["cpu_baseline"] = None
    compact["headline"] = out["headline"]
    if out.get("multi_gpu"):
        m = out["multi_gpu"]
        compact["multi_gpu"] = {k: v for k, v in m.items() if not isinstance(v, (dict, list))}
        if "throughput_schedule" in m:
            compact["multi_gpu"]["throughput_schedule"] = {k: v for k, v in m["throughput_schedule"].items() if k != "note"}
        if "projection" in m:
            compact["multi_gpu"]["projected_speedup"] = {Nw: {s_: round(e[s_]["projected_speedup"], 3) for s_ in ("quality", "throughput") if s_ in e}
                                                         for Nw, e in m["projection"]["worlds"].items()}
    compact["full_record"] = os.path.relpath(path, ROOT)
    line = json.dumps(compact)
    assert len(line) < 6000, len(line)
    sys.stdout.flush()
    print(line, flush=True)


def step_flops(D, NB):
    """Algorithmic flops of one GRU step of one sequence, forward + backward:
    54 D^2 for the cell (2D-wide input) + 6 (B+1) D for the distance-softmax head."""
    return 54.0 * D * D + 6.0 * NB * D


def make_batches(n_local, lens_local, batch_users, seed=123):
    """Shuffled user order (prog_bpr_gru_spatial.py:236-238) cut into launches of equal size (about batch_users each: a
    shard of 12600 users is ONE launch, not 12500 + 100); inside a launch the ids are sorted by descending length so that
    the 16-sequence recurrent tiles are homogeneous."""
    perm = np.random.default_rng(seed).permutation(n_local)
    n_launch = max(1, int(round(n_local / float(batch_users))))
    B = -(-n_local // n_launch)
    batches = []
    for b0 in range(0, n_local, B):
        ids = perm[b0:b0 + B]
        batches.append(ids[np.argsort(-lens_local[ids], kind="stable")])
    return perm, B, batches


def plan_shard(n_user, lens, world, rank, batch_users, schedule="quality", seed=123):
    """This rank's user range and launch schedule.  Users are sharded by check-ins (data.shard_users).  schedule:
      "quality"     as many launches per replica and epoch as the ONE-GPU run makes (round(n_user / batch_users)): N replicas then learn
                    like one GPU (recall 0.534 vs 0.551, against 0.446 with one launch per replica) - the default;
      "throughput"  launches of about batch_users users: at N = 8 one 6250-user launch per replica (the larger effective batch costs epochs).
    Returns (lo, hi, B, batches): batches are local ids, each sorted by descending length."""
    from poi_amd import data as pdata
    lo, hi = pdata.shard_users(n_user, world, rank, lens)
    n_local = hi - lo
    bu = batch_users
    if schedule == "quality" and world > 1:
        n_launch = max(1, int(round(n_user / float(batch_users))))
        bu = max(1, -(-n_local // n_launch))
    _, B, batches = make_batches(n_local, np.asarray(lens[lo:hi]), bu, seed=seed)
    return lo, hi, B, batches


def evaluate_model(model, tab, n_local, dev, chunk=65536):
    """(recall@20, AUC) through the product's evaluation path (snapshot -> predict -> fused score + top-K; AUC flags)."""
    import torch
    ids = np.arange(n_local, dtype=np.int32)
    model.update_trained_items(); model.update_trained_dists()
    hts, sts = model.predict_device(ids)
    model.update_trained_users(hts); model.update_trained_sus(sts)
    tes = torch.as_tensor(tab.tes_p.reshape(-1).astype(np.int32)).to(dev)
    hits = 0
    auc = 0
    for c0 in range(0, n_local, chunk):
        sub = ids[c0:min(c0 + chunk, n_local)]
        idx = model.compute_sub_topk(sub, 20)
        hits += int((idx == tes[c0:c0 + len(sub), None]).any(dim=1).sum().item())
        auc += int(model.compute_sub_auc_preference(sub).sum())
    return hits / float(n_local), auc / float(n_local)


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
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    n_item, n_user, max_len, D = pdata.SHAPES[a.shape]
    if a.shape == "x1":
        # 10 M POIs: i.i.d. Zipf check-ins (the neighbour structure of --local needs a k-d tree over 10 M points), throughput only
        a.local = 0.0; a.no_quality = True; a.no_secondary = True; a.no_cpu_baseline = True; a.no_exact = True; a.no_x1 = True
        a.eval_users = a.eval_users or 8192
        a.table_dtype = a.table_dtype or "f16"
        if a.steps == 250:
            a.steps = 20
        if a.batch_users == 12500:
            a.batch_users = 16384      # 512 recurrent tiles of 32 sequences: two full rounds of the 256 CUs (12500 -> 391 tiles: 1.5 rounds)
    a.table_dtype = a.table_dtype or "f32"
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=a.local, dd=a.dd, ud_km=a.ud_km)
    lo, hi, B, batches = plan_shard(n_user, ds.lens, a.emulate_world or world, rank, a.batch_users, a.replica_schedule)
    tab = ds.shard(lo, hi)
    n_local = hi - lo
    NB = ds.dist_num + 1

    def new_model(tab_, n_users, dim=D, seed=7):
        return poi_amd.models.OboSpatialGru(train=tab_, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_users,
                                            n_item=len(ds.coords), n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=dim, n_hidden=dim,
                                            device=dev, seed=seed, coords=ds.coords, table_dtype=a.table_dtype)
    model = new_model(tab, n_local)
    ctx = model.ctx
    ctx.set_batch_cap(a.batch_cap)
    a.f16_rounding = a.f16_rounding or ("stochastic" if a.table_dtype == "f16" else "nearest")
    ctx.set_f16_rounding(a.f16_rounding, seed=1)
    sync = poi_amd.dist.model_sync(model, force=os.environ.get("POI_BENCH_FORCE_SYNC") == "1")

    lens_local = np.diff(tab.off.astype(np.int64))
    perm = np.random.default_rng(123).permutation(n_local)
    order = torch.as_tensor(np.concatenate(batches).astype(np.int32)).to(dev)
    steps_per_epoch = float(np.maximum(lens_local - 1, 0).sum())

    # The reference's epoch (prog_bpr_gru_spatial.py:221-238): negatives redrawn, user order reshuffled, every epoch.  Default here too
    # (--no-refresh: one fixed order and negative table, rounds 1 - 3): the redraw runs on the device inside the epoch (0.07 ms, inside the
    # clock), the shuffled orders - cut into launches and sorted by length exactly like `batches` - are drawn before the timed region (64
    # distinct epochs, then cycled: host-side permutations are the reference's `random.shuffle`, not part of the step).
    refresh = not a.no_refresh and a.shape != "x1"
    epoch_orders = [order]
    if refresh:
        for k in range(1, 64):
            _, _, bk = make_batches(n_local, lens_local, B, seed=123 + k)
            epoch_orders.append(torch.as_tensor(np.concatenate(bk).astype(np.int32)).to(dev))
    epoch_no = [0]

    def train_epoch(m=None, order_=None, B_=None, n_=None):
        m = m or model; B_ = B_ or B; n_ = n_ or n_local
        if order_ is None:
            k = epoch_no[0]; epoch_no[0] += 1
            order_ = epoch_orders[k % len(epoch_orders)]
            if refresh and k > 0:
                m.resample_negatives_device(7 * 1000003 + k)
        for b0 in range(0, n_, B_):
            m.train_batch(order_[b0:b0 + B_], sync=False)
        if m is model:
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
    # live HIP-event timing INSIDE the timed region, on every (2 n_launch + 1)-th launch: each launch position of the epoch is sampled
    # in turn and the event pairs (~7 us of stream serialisation per kernel) stay below 1 % of the region
    ctx.timing(True, period=2 * len(batches) + 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        train_epoch()
    barrier()
    dt = rank_max(time.perf_counter() - t0)
    KN = ["seq_train", "te_prep", "te_gather", "te_gemm_ax", "te_rec_fwd", "te_head", "te_rec_bwd", "te_psum", "te_wgrad", "te_gemm_dx",
          "te_finalize", "te_dsum", "te_bin_gemm", "te_scatter", "te_tail", "rows_apply", "dense_apply", "te_sort", "te_hot_early"]
    kt = {k: ctx.timing_get(k) for k in KN}
    ctx.timing(False)
    seq_per_s = (n_user if not a.emulate_world else n_local) * a.steps / dt
    multi = sync.report()
    if world > 1 and not a.emulate_world:
        # a SCALE run validates itself: the collective must have seen every rank and the replicas must agree bit for bit
        bad = []
        if multi.get("rccl_world_size", world) != a.gpus:
            bad.append("RCCL saw world size %s, --gpus %d" % (multi.get("rccl_world_size"), a.gpus))
        if not multi.get("replica_checksums_equal", False):
            bad.append("replica checksums differ after the last reconciliation")
        if bad:
            if rank == 0:
                print(json.dumps({"error": "multi-GPU self-check failed: " + "; ".join(bad), "multi_gpu": multi}), flush=True)
            sys.exit(3)
    # N > 1 under the default (quality) schedule: the throughput schedule - one launch of ~--batch-users users per replica - timed beside it
    alt_schedule = None
    if (world > 1 or a.emulate_world) and a.replica_schedule == "quality":
        _, _, B_t, batches_t = plan_shard(n_user, ds.lens, a.emulate_world or world, rank, a.batch_users, "throughput")
        order_t = torch.as_tensor(np.concatenate(batches_t).astype(np.int32)).to(dev)
        for _ in range(3):
            train_epoch(None, order_t, B_t, n_local)
        n_alt = max(10, min(100, a.steps))
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_alt):
            train_epoch(None, order_t, B_t, n_local)
        barrier()
        dta = rank_max(time.perf_counter() - t0)
        alt_schedule = {"schedule": "throughput", "launches_per_epoch_per_replica": len(batches_t), "batch_users_per_launch": B_t, "epochs": n_alt,
                        "ms_per_epoch": 1e3 * dta / n_alt, "seq_per_s": (n_user if not a.emulate_world else n_local) * n_alt / dta,
                        "note": "one launch per replica and epoch: ~2.5x the throughput of the quality schedule at N = 8, at the recall of ONE launch over all "
                                "users per epoch (DESIGN.md section 7: 0.446 vs 0.534 after 120 epochs)"}
        multi["throughput_schedule"] = alt_schedule
    # steady window: the driver's --steps 20 is a 0.17 s window; when the timed region is shorter than 2 s a second, >= 2 s window
    # of the same epochs is timed and reported beside it (thermally settled clocks; never `value`)
    steady = None
    if dt < 2.0:
        n_st = int(np.ceil(2.2 / max(dt / a.steps, 1e-6)))
        barrier()
        t0 = time.perf_counter()
        for _ in range(n_st):
            train_epoch()
        barrier()
        dts = rank_max(time.perf_counter() - t0)
        steady = {"seconds": dts, "epochs": n_st, "ms_per_epoch": 1e3 * dts / n_st,
                  "seq_per_s": (n_user if not a.emulate_world else n_local) * n_st / dts}
    # exact mode.  The timed engine IS the contract-meeting engine since round 4 (exact forward pass: every row of every tensor inside 1e-5
    # at this shape with no loosening, tests/test_gpu_fullsize.py); beside it: the same epochs with the float32 forward kernels of rounds
    # 1 - 3 (poi_ctx_set_exact_forward(0): what the exact forward costs) and the float64-end-to-end engine (poi_ctx_set_engine(4)).
    exact_mode = None
    if rank == 0 and world == 1 and not a.emulate_world and a.table_dtype == "f32" and not a.no_exact:
        f32_fwd = None
        if D in (64, 128) and os.environ.get("POI_TE_XFWD", "1") != "0":
            ctx.set_exact_forward(False)
            try:
                for _ in range(3):
                    train_epoch()
                n_f = max(10, min(60, a.steps))
                barrier(); t0 = time.perf_counter()
                for _ in range(n_f):
                    train_epoch()
                barrier()
                f32_fwd = n_user * n_f / (time.perf_counter() - t0)
            finally:
                ctx.set_exact_forward(True)
        mx = new_model(tab, n_local, seed=7)
        ctx.set_engine("exact")
        try:
            mx.train_batch(order[:B], sync=True)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            nb = 0
            for b0 in list(range(0, n_local, B))[:2]:
                mx.train_batch(order[b0:b0 + B], sync=False); nb += min(B, n_local - b0)
            torch.cuda.synchronize(dev); tx = time.perf_counter() - t0
        finally:
            ctx.set_engine("auto")
        exact_mode = {"engine": "the timed engine: tile engine with the exact forward pass (te_xfwd.hip: input product + forward recurrence in 40-bit fixed "
                                "point on the int8 matrix cores, float64 gates; float32 head / BPTT / write-back)" if f32_fwd else "exact (float64 arithmetic end to end)",
                      "seq_per_s": seq_per_s if f32_fwd else nb / tx,
                      "float32_forward_seq_per_s": f32_fwd, "slowdown": (f32_fwd / seq_per_s) if f32_fwd else seq_per_s / (nb / tx),
                      "float64_engine": {"engine": "exact_engine.hip: float64 arithmetic end to end, per-sequence GEMVs", "seq_per_s": nb / tx, "ms_per_launch": 1e3 * tx / 2,
                                         "slowdown_vs_timed": seq_per_s / (nb / tx)},
                      "parity": "timed engine: every row of all nine tensors within 1e-5 of the float64 oracle at this shape, every row's update within 1e-4 of its "
                                "absolute mass, 300 sequential steps within 1e-5 (tests/test_gpu_fullsize.py, no per-tensor loosening); with the float32 forward "
                                "kernels: 99.9 % of the POI rows, worst row 6e-6 .. 3e-5 depending on the data"}
        del mx

    # ---- evaluation: snapshot -> user vectors -> fused distance term + all-POI score + top-20 -------
    eval_users_per_s = None
    eval_detail = {}
    if not a.no_eval:
        all_ids = np.arange(n_local, dtype=np.int32)
        tes = torch.as_tensor(tab.tes_p.reshape(-1).astype(np.int32)).to(dev)
        n_eval = min(a.eval_users, n_local) if a.eval_users else n_local

        def eval_epoch():
            model.update_trained_items(); model.update_trained_dists()
            if n_eval < n_local:      # (--eval-users: predict the users that are scored, the other rows of the snapshots stay zero)
                h_e, s_e = model.predict_device(all_ids[:n_eval])
                hts = torch.zeros((n_local, h_e.shape[1]), dtype=h_e.dtype, device=dev); hts[:n_eval] = h_e
                sts = torch.zeros((n_local, s_e.shape[1]), dtype=s_e.dtype, device=dev); sts[:n_eval] = s_e
            else:
                hts, sts = model.predict_device(all_ids)
            model.update_trained_users(hts); model.update_trained_sus(sts)
            hits = torch.zeros((), dtype=torch.int64, device=dev)
            for c0 in range(0, n_eval, a.eval_chunk):
                ids = all_ids[c0:min(c0 + a.eval_chunk, n_eval)]
                idx = model.compute_sub_topk(ids, 20)
                hits += (idx == tes[c0:c0 + len(ids), None]).any(dim=1).sum()
            return hits

        ctx.timing(True)
        eval_epoch()                       # warm-up; builds the resident distance-bin matrix (once per data set)
        ms_ulptai = ctx.timing_get("ulptai_build")[0]
        ctx.timing(False)
        # cold pass: top-K thresholds from -inf (the first evaluation of a run; poi_ctx_set_topk_seed off)
        model.topk_seeding = False
        barrier()
        t0 = time.perf_counter()
        eval_epoch()
        barrier()
        dte_cold = rank_max(time.perf_counter() - t0)
        # the same pass once more under the event recorder: where the first evaluation's time goes (not the timed pass: event pairs serialise the stream)
        ctx.timing(True)
        eval_epoch()
        torch.cuda.synchronize(dev)
        first_breakdown = {k: ctx.timing_get(k)[0] for k in ("seq_predict", "te_predict", "score_maxpass", "score_filter", "score_rescore", "score_topk", "pack_items", "topk_merge")}
        first_breakdown = {k: v for k, v in first_breakdown.items() if v}
        first_breakdown["survivors_per_user"] = ctx.topk_filter_stats()["survivors"] / max(ctx.topk_filter_stats()["users"], 1)
        ctx.timing(False)
        model.topk_seeding = True
        dte_first = dte_cold
        eval_epoch()                       # fills the seeds
        # timed passes at the reference driver's cadence - one training epoch, then one evaluation (prog_bpr_gru_spatial.py:249-290):
        # the top-K lists of the previous evaluation seed the thresholds of this one, under a model that has moved by an epoch.
        # Only the evaluations are timed.
        dte = 0.0
        ctx.timing(True); ctx.lib.poi_timing_enable(ctx.handle, 0)
        for _ in range(a.eval_steps):
            train_epoch()
            ctx.lib.poi_timing_enable(ctx.handle, 1)
            barrier()
            t0 = time.perf_counter()
            hits = eval_epoch()
            barrier()
            dte += rank_max(time.perf_counter() - t0)
            ctx.lib.poi_timing_enable(ctx.handle, 0)
        ms_one, n_score = ctx.timing_get("score_topk")
        ms_filter, ms_rescore, ms_pack = ctx.timing_get("score_filter")[0], ctx.timing_get("score_rescore")[0], ctx.timing_get("pack_items")[0]
        ms_seedk = ctx.timing_get("topk_seed")[0]
        ms_score = ms_one + ms_filter + ms_rescore
        fstats = ctx.topk_filter_stats()
        ms_pred = ctx.timing_get("seq_predict")[0] + ctx.timing_get("te_predict")[0]
        ms_dist = ctx.timing_get("dist_prob")[0]
        ctx.timing(False)
        # (with the per-epoch refresh, the default, the headline evaluation rate is the FIRST evaluation of a run - nothing carried over from an
        # earlier evaluation; the seeded steady state of a training loop that evaluates every epoch and the pass without any seed are beside it)
        eval_users_per_s_seeded = (n_user if n_eval == n_local else n_eval) * a.eval_steps / dte
        eval_users_per_s = ((n_user if n_eval == n_local else n_eval) / dte_first) if refresh else eval_users_per_s_seeded
        fl = 2.0 * n_eval * n_item * D * a.eval_steps
        eval_detail = {"ms_per_eval": 1e3 * dte_first if refresh else 1e3 * dte / a.eval_steps,
                       "headline_is": "the first evaluation of a run, unseeded" if refresh else "the seeded steady state",
                       "ms_per_eval_first": 1e3 * dte_first, "first_eval_kernel_ms": first_breakdown, "recall_at_20_after_timed_training": float(hits.item()) / n_eval,
                       "users_scored_per_eval": n_eval, "users_predicted_per_eval": n_eval,
                       "distance_term": "resident bin matrix" if getattr(model, "_ulptai", None) is not None else "bins on the fly (poi_score_topk_geo)",
                       # 2 U N D over the time of ALL scoring kernels of a timed evaluation (filter + rescoring + pre-pass / fallback): an EQUIVALENT rate - the
                       # two-stage path does most of these flops on the f16 matrix pipe (see two_stage.filter_frac_of_f16_mfma_peak), so it may exceed the f32 peak
                       "score_topk_equivalent_tflops": fl / (ms_score * 1e-3) / 1e12 if ms_score > 0 else None,
                       "score_topk_equivalent_frac_of_f32_mfma_peak": fl / (ms_score * 1e-3) / 1e12 / PEAK_F32_TFLOPS if ms_score > 0 else None,
                       "ms_predict_per_eval": ms_pred / a.eval_steps, "ms_score_topk_per_eval": ms_score / a.eval_steps,
                       "ms_dist_prob_per_eval": ms_dist / a.eval_steps, "ms_ulptai_build_once": ms_ulptai,
                       "eval_chunk_users": a.eval_chunk,
                       "two_stage": {"what": "seeded calls: f16 filter pass (v_mfma_f32_32x32x16_f16 + rigorous error bound) -> exact float32 rescoring of the "
                                             "survivors; bit-identical lists (include/poi_hip.h, poi_ctx_set_topk_filter)",
                                     "ms_filter_per_eval": ms_filter / a.eval_steps, "ms_rescore_per_eval": ms_rescore / a.eval_steps,
                                     "ms_one_stage_fallback_per_eval": ms_one / a.eval_steps, "ms_pack_items_per_eval": ms_pack / a.eval_steps,
                                     "ms_seed_kernel_per_eval": ms_seedk / a.eval_steps,
                                     "survivors_per_user_last_call": fstats["survivors"] / max(fstats["users"], 1),
                                     "tiles_flagged_last_call": fstats["tiles_flagged"], "tiles_last_call": fstats["tiles"],
                                     "filter_frac_of_f16_mfma_peak": (fl / (ms_filter * 1e-3) / 1e12 / 2500.0) if ms_filter > 0 else None},
                       "cadence": "one training epoch between evaluations (untimed), as the reference driver; top-K thresholds seeded from the previous evaluation's lists",
                       "eval_users_per_s_seeded": eval_users_per_s_seeded, "ms_per_eval_seeded": 1e3 * dte / a.eval_steps,
                       "ms_per_eval_unseeded": 1e3 * dte_cold,
                       "eval_users_per_s_unseeded": (n_user if n_eval == n_local else n_eval) / dte_cold}

    # ---- roofline of the dominant kernel (live HIP-event timing inside the timed region) -----------
    off64 = tab.off.astype(np.int64)
    order_host = order.cpu().numpy()
    uniq = 0                       # table rows written per epoch: unique rows per LAUNCH (one write per row and launch)
    uniq_seq = 0                   # n_unique(p U q) + n_unique(dp) per SEQUENCE, summed
    pos = 0
    s_rows = 0                     # per-POI regrouping: distinct step-input POIs per launch (rows of S), summed over the epoch's launches
    for b0 in range(0, n_local, B):
        ids = order_host[b0:b0 + B]
        sel = np.concatenate([np.arange(off64[u], off64[u + 1]) for u in ids])
        s_rows += len(np.unique(np.concatenate([tab.p[off64[u]:off64[u + 1] - 1] for u in ids])))
        uniq += len(np.unique(np.concatenate((tab.p[sel], tab.q[sel])))) + 1 + len(np.unique(np.append(tab.dp[sel], ds.dist_num)))
        pos += len(sel)
    if n_local <= 200000:
        # per-sequence unique counts, vectorised: sort (user, id) pairs
        user_of = np.repeat(np.arange(n_local), lens_local)
        for arrs in ((tab.p, tab.q), (tab.dp,)):
            key = np.concatenate([user_of.astype(np.int64) * (n_item + NB + 2) + np.asarray(x, np.int64) for x in arrs])
            uniq_seq += len(np.unique(key))
    D2 = float(D * D)
    # EXECUTED flops per kernel (the roofline fractions are matrix-pipe utilisation).  At D >= 128 the distance-bin
    # half of the step input goes through per-bin tables: te_gemm_ax / te_gemm_dx / the d ui jobs of
    # te_wgrad only multiply the POI half, i.e. 36 D^2 + 6 NB D executed against the 54 D^2 + 6 NB D of the
    # reference formulation (step_flops)
    bintab = D >= 128 and NB <= 2048     # (te_bintab)
    xk = 6 if bintab else 12
    # per-POI regrouping (bintab): te_gemm_dx and the d ui jobs of te_wgrad contract over the S rows (distinct step-input POIs of a
    # launch) instead of over the steps - rho = S rows / steps
    ppoi = bintab and os.environ.get("POI_TE_PPOI", "1") != "0"
    rho = (s_rows / steps_per_epoch) if ppoi else 1.0
    # forward table (bintab, 16-sequence recurrent tiles): te_gemm_ax multiplies the n_item + 1 table rows once per launch instead of one
    # row per step (abi.hip te_setup: when the table has at most half as many rows as the launch's step capacity)
    n_launches_ = len(batches)
    fwd_tab = (bintab and D < 256 and os.environ.get("POI_TE_FWDTAB", "1") != "0"
               and 2 * (n_item + 1) <= B * max(model.max_len - 1, 1) + 192)
    # exact forward pass (te_xfwd.hip, default for dims 64 / 128 and float32 tables): te_gemm_ax and te_rec_fwd form their products from
    # five int8 digit planes per operand - 15 digit-pair MFMAs per product - and evaluate the gates in float64
    xfwd = D in (64, 128, 256) and os.environ.get("POI_TE_XFWD", "1") != "0"      # (dim 256: te_gemmx<256> + te_rec_fwdd, float64 MFMA recurrence)
    if xfwd:
        fwd_tab = (os.environ.get("POI_TE_FWDTAB", "1") != "0" and 2 * (n_item + 1) <= B * max(model.max_len - 1, 1) + 192)
    # ... over the launch's step-input POIs only (abi.hip: forward_table_compact, launches of >= 1536 sequences on the regrouped path): rho rows per step
    xcomp = xfwd and bintab and B >= 1536 and os.environ.get("POI_TE_XCOMP", "1") != "0"
    if xcomp:
        fwd_tab = True
    ax_rows = rho * steps_per_epoch if xcomp else (n_item + 1.0) * n_launches_ if fwd_tab else steps_per_epoch
    # split products (default): te_rec_bwd and te_wgrad form every float32 product from six bf16 partial products, the training head (<= 256
    # bins: te_head3) from five - priced as EXECUTED bf16 flops against the dense bf16 peak, the float32-equivalent rate beside it
    split = os.environ.get("POI_TE_SPLIT", "1") != "0"
    rec1_max = int(os.environ.get("POI_TE_REC1", "1800"))        # launches of at most this many sequences: per-sequence recurrent kernels (float32 FMAs)
    head3 = split and NB <= 256 and os.environ.get("POI_TE_HEAD3", "1") != "0"
    efuse = head3 and D == 128 and os.environ.get("POI_TE_EFUSE", "1") != "0"      # (abi.hip te_setup: E gathered inside te_head3)
    work = {"seq_train": ("flop", step_flops(D, NB) * steps_per_epoch),
            "te_gemm_ax": ("i8op", 15 * 6 * D2 * ax_rows) if xfwd else ("bf16x6", 6 * xk * D2 * ax_rows) if split and bintab and D >= 256 else ("flop", xk * D2 * ax_rows),
            # (dim 256: te_rec_fwdd - the recurrent products in float64 on the matrix cores, v_mfma_f64_16x16x4_f64)
            # (launches of at most xrec1_max sequences: te_rec_fwd1x - float64 FMAs on the vector ALUs, priced against the float64 vector rate)
            "te_rec_fwd": ("f64flop", 6 * D2 * steps_per_epoch) if xfwd and (D >= 256 or B <= int(os.environ.get("POI_TE_XREC1", "1100"))) else ("i8op", 15 * 6 * D2 * steps_per_epoch) if xfwd else ("flop", 6 * D2 * steps_per_epoch),
            "te_head": ("bf16x5", 5 * 4.0 * NB * D * steps_per_epoch) if head3 else ("flop", 4.0 * NB * D * steps_per_epoch),
            "te_rec_bwd": ("bf16x6", 6 * 6 * D2 * steps_per_epoch) if split and (B > rec1_max or D >= 256) else ("flop", 6 * D2 * steps_per_epoch),
            # d ui (over S rows), d wh and d vs (split-K)
            "te_wgrad": ("bf16x6", 6 * ((6 + xk * rho) * D2 + 2.0 * NB * D) * steps_per_epoch) if split and D in (64, 128, 256) else ("flop", ((6 + xk * rho) * D2 + 2.0 * NB * D) * steps_per_epoch),
            "te_gemm_dx": ("bf16x6", 6 * xk * rho * D2 * steps_per_epoch) if split and D <= 128 else ("flop", xk * rho * D2 * steps_per_epoch),
            # per-POI sums of DA: one read of the 3D-wide DA rows + the S rows written
            "te_psum": ("byte", 3.0 * D * 4 * (1.0 + rho) * steps_per_epoch),
            # implementation bytes of the HBM-bound kernels (what each kernel has to move given the decomposition):
            # te_gather builds E = lt[p'] - lt[q'] (two table rows + two indices in, one packed row out per step); round 6, dim 128: te_head3 gathers
            # the two rows itself (TeArgs.efuse) - te_gather only translates the steps' ids for the compact forward table (three ints per step)
            "te_gather": ("byte", 12.0 * steps_per_epoch) if efuse else ("byte", (3.0 * D * 4 + 8) * steps_per_epoch),
            "rows_apply": ("byte", 2.0 * uniq * D * 4.0),      # read + write of every touched row
            # sorted scatter: h twice per step (the g*h term of the positive and of the negative row) + the dx sums (per-POI regrouping:
            # one D-row per S row; otherwise D floats per step, two-table path 2D) in, every touched row read + written
            "te_scatter": ("byte", ((2.0 + rho) if bintab else 4.0) * D * 4 * steps_per_epoch + 2.0 * uniq * D * 4.0),
            # per-bin sums of DA (bintab): one read of the 3D-wide DA rows
            "te_dsum": ("byte", 3.0 * D * 4 * steps_per_epoch)}
    n_launches = len(batches)
    kernels = {}
    for k in KN:
        ms, nl = kt[k]
        if nl == 0:
            continue
        # per-step time from the per-launch average: the event recorder keeps a bounded number of regions, so on long
        # runs only the first launches are timed - their average is still the launch duration
        per_step = (ms / nl) * n_launches
        ent = {"ms_per_step": per_step, "launches": nl, "avg_ms": ms / nl}
        if k in work:
            kind, w = work[k]
            rate = w / (per_step * 1e-3)
            if kind == "flop":
                ent.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_F32_TFLOPS, unit="TFLOP/s", frac=rate / 1e12 / PEAK_F32_TFLOPS)
            elif kind.startswith("bf16x"):
                ent.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s (bf16, executed partial products)", frac=rate / 1e12 / PEAK_BF16_TFLOPS,
                           float32_equivalent_tflops=rate / float(kind[5:]) / 1e12,
                           note="every float32 product from %s bf16 partial products of three / two planes per operand (v_mfma_f32_*_bf16, float32 accumulation)" % kind[5:])
            elif kind == "f64flop":
                ent.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_F64_TFLOPS, unit="TFLOP/s (float64 MFMA)", frac=rate / 1e12 / PEAK_F64_TFLOPS,
                           note=("exact forward pass at dim 256: the recurrent products on v_mfma_f64_16x16x4_f64 (16 passes: the float64 vector rate), float32 weight "
                                 "fragments streamed from L2, float64 gates in the MFMA's issue shadow") if D >= 256 else
                                "exact forward pass of a small launch: te_rec_fwd1x, one sequence per workgroup, float64 FMAs with register-resident weights")
            elif kind == "i8op":
                ent.update(bound="mfma", achieved=rate / 1e12, peak=PEAK_I8_TOPS, unit="TOP/s (int8)", frac=rate / 1e12 / PEAK_I8_TOPS,
                           float64_equivalent_tflops=rate / 15.0 / 1e12,
                           note="exact forward pass: every product from 15 int8 digit-pair MFMAs (v_mfma_i32_*_i8, exact int32 accumulation), gates in "
                                "float64 on the vector ALUs - the kernel is bound by the float64 gate math and its per-step latency chain, not by the matrix pipe")
            else:
                ent.update(bound="hbm", achieved=rate / 1e9, peak=PEAK_HBM_GBS, unit="GB/s", frac=rate / 1e9 / PEAK_HBM_GBS)
        if k == "te_finalize":
            ent["note"] = "runs on the side stream next to te_wgrad / te_gemm_dx: its span overlaps them and is not part of the serial sum"
        if k in ("te_dsum", "te_bin_gemm", "te_scatter") and kt["te_tail"][1]:
            ent["note"] = ("the distance-bin chain (te_dsum, te_bin_gemm) runs on the side stream from the end of te_wgrad on, next to te_gemm_dx and "
                           "te_scatter (POI rows): the spans overlap and stretch each other (stand-alone: 0.37 / 0.14 / 0.44 ms per epoch, POI_TE_DBG=1) - "
                           "te_tail, te_scatter's start to the join, is the span that counts")
        if k == "te_hot_early":
            ent["note"] = ("side stream, beside te_rec_bwd: the chunk sums of the write-back's hot rows (te_hot_reduce; they need the sorted entries, gcoef and H - nothing later) - part "
                           "of the scatter, off the main stream's chain since round 6 (roofline_gather_scatter.hot_reduce_ms_hidden; same-box A/B at 12500 users: te_tail -13 us, te_rec_bwd +10 us)")
        if k == "te_sort":
            ent["note"] = ("side stream, beside te_rec_fwd: stable radix sort of the table-touch slots + segment bounds + S-row assignment + the bin chain's chunk offsets - part "
                           "of the scatter, hidden from the main stream's chain (roofline_gather_scatter.sort_ms_hidden); the span stretches with what runs beside it")
        if k == "te_tail":
            ent["note"] = ("te_scatter's start to the join with the side stream's distance-bin chain (te_dsum + te_bin_gemm, started behind te_wgrad) "
                           "and the dense write-back that follows it there (dense_apply: its time is inside this span on launches of >= 1024 users); "
                           "the serial sums below use it instead of the three")
        kernels[k] = ent
    dom = max((k for k in kernels if "bound" in kernels[k]), key=lambda k: kernels[k]["ms_per_step"])
    # HBM traffic per launch from the COMMITTED PMC passes of this command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    # separate runs, gfx950 correction applied): static numbers, valid only for the profiled workload - not measured in this run
    traffic = {}
    traffic_src = None
    for tag in (PROFILE_TAG, "r01f"):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", tag + "_pmc_traffic.json")))
            if a.shape == "gowalla" and B == 12500 and world == 1:
                traffic = {k: v["hbm_bytes_per_launch"] for k, v in pj["kernels"].items()}
                traffic_src = "static: profiles/%s_pmc_traffic.json (rocprofv3 PMC passes of this command, not measured in this run)" % tag
            break
        except Exception:
            continue
    for k in kernels:
        if k in traffic:
            kernels[k]["traffic_bytes_per_launch"] = traffic[k]
    roofline = dict(kernel=dom, traffic=traffic.get(dom), traffic_source=traffic_src,
                    **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_ms")})
    roofline["note"] = (kernels[dom].get("note") if kernels[dom].get("note") and kernels[dom].get("bound") == "mfma" and kernels[dom].get("peak") != PEAK_F32_TFLOPS else
                        "f32 arithmetic on v_mfma_f32_32x32x2_f32 (f32 vector peak == f32-input MFMA peak on gfx950)")
    roofline_gs_hook = roofline
    # ---- gather / scatter against the HBM roofline: three accountings over the SAME kernel time --------------------
    GS = ("te_gather", "te_psum", "te_dsum", "te_scatter", "rows_apply")
    forked = "te_tail" in kernels          # te_dsum (+ te_bin_gemm) and te_scatter overlap: their time is the fork-to-join span
    gs_ms = sum(kernels[k]["ms_per_step"] for k in GS if k in kernels and not (forked and k in ("te_dsum", "te_scatter"))) + (kernels["te_tail"]["ms_per_step"] if forked else 0.0)
    gs_impl = sum(work[k][1] for k in GS if k in kernels)
    e = 2.0 if a.table_dtype == "f16" else 4.0      # (config X counts 2-byte elements)
    survey_bytes = 3.0 * pos * D * e + 16.0 * pos + uniq_seq * D * e if uniq_seq else None      # SURVEY.md 8(d) bytes_seq, summed
    rows_only = 2.0 * steps_per_epoch * D * e + 2.0 * uniq * D * e          # E's two table rows per step + touched rows r/w
    acc = lambda b: {"bytes_per_epoch": b, "achieved_GBps": b / (gs_ms * 1e-3) / 1e9, "frac": b / (gs_ms * 1e-3) / 1e9 / PEAK_HBM_GBS} if b and gs_ms > 0 else None
    hbm = {"kernels": [k for k in GS if k in kernels], "bound": "hbm", "peak": PEAK_HBM_GBS, "unit": "GB/s", "ms_per_epoch": gs_ms,
           "survey_8d": acc(survey_bytes), "implementation": acc(gs_impl), "embedding_rows_only": acc(rows_only),
           "traffic": sum(traffic.get(k, 0) for k in GS) * n_launches or None, "traffic_source": traffic_src,
           "definitions": {
               "survey_8d": "SURVEY.md 8(d) / BASELINE.md 4: per sequence 3*L*D*e + 16*L gathered + (n_unique(p U q) + n_unique(dp))*D*e written, summed over "
                            "the epoch's sequences - the contract figure.  NOTE the gather of the step input [lt[p] | di[dp]] is fused into the MFMA-bound "
                            "te_gemm_ax / te_wgrad (rows go straight into their LDS tiles) and its time is NOT in ms_per_epoch",
               "implementation": "bytes the HBM-bound kernels have to move in this decomposition, intermediates included (E rows out, h rows and the per-POI "
                                 "dx sums in, DA rows read once for the per-bin and once for the per-POI sums, S rows out, touched rows read + written)",
               "embedding_rows_only": "table rows only: the two rows of E per step + every touched row read and written once per launch"}}
    hbm["sort_ms_hidden"] = kernels["te_sort"]["ms_per_step"] if "te_sort" in kernels else None
    hbm["hot_reduce_ms_hidden"] = kernels["te_hot_early"]["ms_per_step"] if "te_hot_early" in kernels else None
    if efuse:
        # the two table rows of E are gathered inside te_head3 (an MFMA-bound kernel): A/B on one box (profiles/r06, POI_TE_EFUSE=0 / 1, 12500-user launches):
        # te_gather 71.0 -> 6.7 us, te_head 217.6 -> 228.4 us per launch - the 10.8 us are charged here
        fused_ms = 10.8e-3 * n_launches
        # ... and the hot rows' chunk sums, which run beside te_rec_bwd since round 6 (te_hot_early: a SPAN on the side stream, stretched by the kernel it hides behind),
        # at their stand-alone cost: 19.5 us per 12500-user launch (profiles/r06 notes: te_hot_reduce in the tail, rocprofv3)
        hot_ms = 19.5e-3 * n_launches if "te_hot_early" in kernels else 0.0
        hbm["frac_survey_8d_with_fused_gather_and_hidden_hot_rows_charged"] = (survey_bytes / ((gs_ms + fused_ms + hot_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS) if survey_bytes else None
        hbm["fused_gather"] = {"what": "E = lt[p'] - lt[q'] is gathered inside te_head3 (no E rows in HBM): te_gather 71.0 -> 6.7 us, te_head +10.8 us per 12500-user launch (same-box A/B)",
                               "ms_per_epoch_charged": fused_ms,
                               "frac_survey_8d_with_it": (survey_bytes / ((gs_ms + fused_ms) * 1e-3) / 1e9 / PEAK_HBM_GBS) if survey_bytes else None}
    hbm["achieved"] = (hbm["survey_8d"] or hbm["implementation"])["achieved_GBps"]
    hbm["frac"] = (hbm["survey_8d"] or hbm["implementation"])["frac"]
    roofline_gs_hook["gather_scatter"] = {"bound": "hbm", "kernels": hbm["kernels"], "ms_per_epoch": gs_ms, "frac_survey_8d": (hbm["survey_8d"] or {}).get("frac"),
                                          "frac_bytes_moved": (hbm["implementation"] or {}).get("frac"), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
    total_flops = step_flops(D, NB) * steps_per_epoch
    executed_flops = sum((w if kind == "flop" else w / 15.0 if kind == "i8op" else w / float(kind[5:])) for k, (kind, w) in work.items()
                         if (kind in ("flop", "i8op") or kind.startswith("bf16x")) and k in kernels and k != "seq_train") or total_flops
    train_kernel_ms = sum(kernels[k]["ms_per_step"] for k in kernels if k not in (("te_finalize", "te_dsum", "te_bin_gemm", "te_scatter") if forked else ("te_finalize", "te_tail")))

    solo = rank == 0 and world == 1 and not a.emulate_world
    # ---- the reference's own schedule on the GPU + learning quality at equal wall time --------------------------------
    reference_schedule = None
    quality = None
    if solo and not a.no_quality:
        pop = np.bincount(tab.p, minlength=n_item)
        top = np.argsort(-pop)[:20]

        def timed_training(run_epoch, seconds, users_per_epoch):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < seconds:
                n += run_epoch()
                torch.cuda.synchronize(dev)
            return n, time.perf_counter() - t0

        # (1) reference schedule: model.train(uidx), one SGD step per user in shuffled order (prog_bpr_gru_spatial.py:249-250)
        ctx.set_batch_cap(1.0)
        mref = new_model(tab, n_local, seed=11)
        pr = np.random.default_rng(5).permutation(n_local).astype(np.int32)
        cursor = [0]

        def ref_chunk():
            c0 = cursor[0]
            mref.train_sequence(pr[c0:c0 + 500], sync=False)        # 500 launches of one sequence each, ids staged once
            cursor[0] = (c0 + 500) % max(n_local - 500, 1)
            return 500
        n_ref, t_ref = timed_training(ref_chunk, a.reference_seconds, n_local)
        rec_ref, auc_ref = evaluate_model(mref, tab, n_local, dev)
        reference_schedule = {"seq_per_s": n_ref / t_ref, "ms_per_step": 1e3 * t_ref / n_ref, "users_trained": n_ref, "train_seconds": t_ref,
                              "note": "one SGD step per user in shuffled order, one poi_spatial_step launch per user (model.train_sequence: the reference's "
                                      "`for uidx: model.train(uidx)` loop with the ids staged on the device once): the one-sequence path - five kernels per step"}
        del mref

        # (2) batched modes from the same initial parameters for --quality-seconds of training each.  Every epoch takes a NEW shuffled order, as
        # the reference driver does (32 distinct ones, cycled): with ONE fixed launch composition replayed for hundreds of epochs the 12500-user
        # launches plateau at recall 0.585; reshuffled they pass the reference schedule's 0.60 (tools/quality.py, round 4)
        _orders_cache = {}

        def shuffled_orders(Bq):
            if Bq not in _orders_cache:
                lst = []
                for k in range(32):
                    _, Bq_, bt = make_batches(n_local, lens_local, Bq, seed=321 + k)
                    lst.append(torch.as_tensor(np.concatenate(bt).astype(np.int32)).to(dev))
                _orders_cache[Bq] = (Bq_, lst)
            return _orders_cache[Bq]

        def batched(Bq, cap, seconds):
            ctx.set_batch_cap(cap)
            m = new_model(tab, n_local, seed=11)
            Bq_, ods = shuffled_orders(Bq)
            ep = [0]

            def one():
                if ep[0] > 0:
                    m.resample_negatives_device(99 * 1000003 + ep[0])          # fresh negatives every epoch, as the driver does (:221-228)
                train_epoch(m, ods[ep[0] % len(ods)], Bq_, n_local); ep[0] += 1      # ... and a fresh shuffled order (:236-238)
                return n_local
            n, t = timed_training(one, seconds, n_local)
            rec, auc = evaluate_model(m, tab, n_local, dev)
            return {"batch_users_per_launch": Bq_, "batch_cap": cap, "train_seconds": t, "epochs": ep[0], "seq_per_s": n / t,
                    "recall_at_20": rec, "auc": auc}
        # time to quality: training seconds until recall@20 reaches the reference schedule's (evaluated every 0.25 s of training, 4 s budget)
        def time_to_recall(Bq, cap, target, budget=4.0, slice_s=0.25):
            ctx.set_batch_cap(cap)
            m = new_model(tab, n_local, seed=11)
            Bq_, ods = shuffled_orders(Bq)
            t_train, ep, best = 0.0, 0, 0.0
            while t_train < budget:
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                while time.perf_counter() - t0 < slice_s:
                    if ep:
                        m.resample_negatives_device(99 * 1000003 + ep)
                    train_epoch(m, ods[ep % len(ods)], Bq_, n_local); ep += 1
                    torch.cuda.synchronize(dev)
                t_train += time.perf_counter() - t0
                rec, _ = evaluate_model(m, tab, n_local, dev)
                best = max(best, rec)
                if rec >= target:
                    return {"batch_users_per_launch": Bq_, "batch_cap": cap, "seconds": t_train, "epochs": ep, "recall_at_20": rec}
            return {"batch_users_per_launch": Bq_, "batch_cap": cap, "seconds": None, "epochs": ep, "best_recall_at_20_in_budget": best, "budget_s": budget}
        ttr = {"target_recall_at_20": rec_ref, "what": "training seconds until recall@20 reaches what the reference schedule (one user per step) has after %.1f s" % t_ref,
               "B=1 (reference schedule)": {"seconds": t_ref, "recall_at_20": rec_ref},
               "B=256": time_to_recall(256, 16.0, rec_ref), "B=1563": time_to_recall(1563, 32.0, rec_ref), "B=12500": time_to_recall(a.batch_users, a.batch_cap, rec_ref, budget=6.0)}
        modes = {"headline": batched(a.batch_users, a.batch_cap, a.quality_seconds),
                 "headline_mean_rule": batched(a.batch_users, 1.0, a.quality_seconds),
                 "small_launches": batched(256, 16.0, a.quality_seconds)}
        ctx.set_batch_cap(a.batch_cap)
        quality = {"data": "synthetic with a next-POI signal: %.0f %% of the transitions go to one of the 32 nearest POIs" % (100 * a.local),
                   "random_recall_at_20": 20.0 / n_item, "popularity_recall_at_20": float(np.isin(tab.tes_p.reshape(-1), top).mean()),
                   "reference_schedule": {"train_seconds": t_ref, "users_trained": n_ref, "epochs": n_ref / float(n_local),
                                          "recall_at_20": rec_ref, "auc": auc_ref},
                   "modes": modes, "time_to_recall": ttr,
                   "headline_vs_reference": {"recall_ratio": modes["headline"]["recall_at_20"] / max(rec_ref, 1e-9),
                                             "wall_time_ratio": modes["headline"]["train_seconds"] / t_ref,
                                             "statement": "headline mode (B = %d, cap = %g) after %.1f s of training vs the reference schedule after %.1f s"
                                                          % (B, a.batch_cap, modes["headline"]["train_seconds"], t_ref)}}

    # ---- secondary shape: BASELINE.json configs[1] (Foursquare shape, D = 64, the two-table path) ------------------------
    secondary = None
    if solo and not a.no_secondary and a.shape != "foursquare":
        ni2, nu2, ml2, D2_ = pdata.SHAPES["foursquare"]
        ds2 = pdata.make_synthetic(nu2, ni2, ml2, seed=20260928 + 1, local=a.local)
        tab2 = ds2.shard(0, nu2)
        m2 = poi_amd.models.OboSpatialGru(train=tab2, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=nu2, n_item=ni2,
                                          n_dists=[ds2.dist_num, ds2.dd / 1000.0], n_in=D2_, n_hidden=D2_, device=dev, seed=7, coords=ds2.coords)
        l2 = np.diff(tab2.off.astype(np.int64))
        _, B2, bt2 = make_batches(nu2, l2, nu2)
        od2 = torch.as_tensor(np.concatenate(bt2).astype(np.int32)).to(dev)
        for _ in range(5):
            train_epoch(m2, od2, B2, nu2)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(400):
            train_epoch(m2, od2, B2, nu2)
        torch.cuda.synchronize(dev); t2 = time.perf_counter() - t0
        evaluate_model(m2, tab2, nu2, dev)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(20):
            rec2, auc2 = evaluate_model(m2, tab2, nu2, dev)
        torch.cuda.synchronize(dev); te2 = time.perf_counter() - t0
        secondary = {"workload": "synthetic foursquare-shape: %d POIs, %d users, seq<=%d, dim=%d (BASELINE.json configs[1]); one %d-user launch per epoch"
                                 % (ni2, nu2, ml2, D2_, B2),
                     "train_seq_per_s": nu2 * 400 / t2, "ms_per_epoch": 1e3 * t2 / 400, "eval_users_per_s": nu2 * 20 / te2,
                     "recall_at_20_after_405_epochs": rec2, "auc": auc2}
        del m2
        # the reference's in-source config (prog_bpr_gru_spatial.py:54-78): latent_size 20 - stored zero-padded to dim 64 (tile engine)
        m3 = poi_amd.models.OboSpatialGru(train=tab2, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=nu2, n_item=ni2,
                                          n_dists=[ds2.dist_num, ds2.dd / 1000.0], n_in=20, n_hidden=20, device=dev, seed=7, coords=ds2.coords)
        for _ in range(5):
            train_epoch(m3, od2, B2, nu2)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(400):
            train_epoch(m3, od2, B2, nu2)
        torch.cuda.synchronize(dev); t3 = time.perf_counter() - t0
        secondary["reference_latent_size_20"] = {"train_seq_per_s": nu2 * 400 / t3, "ms_per_epoch": 1e3 * t3 / 400, "kernel_dim": m3.kdim,
                                                 "note": "same data, dim = 20 (the reference's default): stored zero-padded to the tile engine's dim 64, exact"}
        del m3

    # ---- launch_sweep: sequences/s against the launch size (what the small-launch paths of DESIGN.md section 5 buy) -------------------------
    launch_sweep = None
    if solo and not a.no_quality and a.shape == "gowalla":
        ctx.set_batch_cap(a.batch_cap)
        msw = new_model(tab, n_local, seed=7)
        launch_sweep = {}
        for Bs in (1, 16, 256, 1024, 1563, 4096):
            ids_s = np.random.default_rng(Bs).permutation(n_local)[:Bs]
            ids_s = torch.as_tensor(ids_s[np.argsort(-lens_local[ids_s], kind="stable")].astype(np.int32)).to(dev)
            for _ in range(5):
                msw.train_batch(ids_s, sync=False)
            reps = 200 if Bs <= 256 else 60
            win = []
            for _w in range(3):      # three windows, the median counts: one window of round 4's record was 48 % off (1011 us at B = 1563 against 676 - 684 in
                torch.cuda.synchronize(dev); t0 = time.perf_counter()      # every other run of the same tree: a transient of that box)
                for _ in range(reps):
                    msw.train_batch(ids_s, sync=False)
                torch.cuda.synchronize(dev); win.append((time.perf_counter() - t0) / reps)
            ts = sorted(win)[1]
            launch_sweep["B=%d" % Bs] = {"us_per_launch": 1e6 * ts, "seq_per_s": Bs / ts, "us_min": 1e6 * min(win), "us_max": 1e6 * max(win)}
        xrec1_max = int(os.environ.get("POI_TE_XREC1", "1100"))
        launch_sweep["note"] = ("median of three windows; one fixed launch repeated (host loop through models.train_batch): B = 1 takes the one-sequence path; the forward recurrence runs per "
                                "sequence in float64 (te_rec_fwd1x) up to %d sequences and the backward one per sequence (te_rec_bwd1) up to %d, above that 16-sequence tiles "
                                "(int8 digits forward, bf16 split products backward)" % (xrec1_max, rec1_max))
        del msw

    # ---- multi_gpu.projection: rank 0's shard of an N-way user split trained on THIS GPU under both replica schedules (what --emulate-world N times),
    # + the reconciliation's local kernels timed here and its all-reduce priced at one xGMI link - no hardware curve, a projection
    if solo and not a.no_quality and a.shape == "gowalla" and not a.no_projection:
        proj = {"one_gpu_ms_per_epoch": 1e3 * dt / a.steps, "worlds": {}}
        ctx.set_batch_cap(a.batch_cap)
        # local part of the reconciliation (delta + touch counts, combine, next snapshot) on the real parameter set
        rs = poi_amd.dist.model_sync(model, force_backend=True)
        recon_local_ms, recon_bytes = None, 0
        if rs.backend is not None:
            rs.backend.begin_epoch()
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for _ in range(20):
                rs.backend.make_delta(); rs.backend.apply(8)
            torch.cuda.synchronize(dev); recon_local_ms = 1e3 * (time.perf_counter() - t0) / 20
            recon_bytes = rs.backend.n * 4 + rs.backend.n16 * 2
            rs.backend.begin_epoch()
        rs.close()
        for Nw in (2, 4, 8):
            ent = {}
            ar_ms = 1e3 * 2.0 * (Nw - 1) / Nw * recon_bytes / 153e9      # ring all-reduce bound by ONE xGMI link per direction
            for sched in ("quality", "throughput"):
                lo_p, hi_p, B_p, batches_p = plan_shard(n_user, ds.lens, Nw, 0, a.batch_users, sched)
                tab_p = ds.shard(lo_p, hi_p)
                mp = new_model(tab_p, hi_p - lo_p, seed=7)
                order_p = torch.as_tensor(np.concatenate(batches_p).astype(np.int32)).to(dev)
                for _ in range(3):
                    train_epoch(mp, order_p, B_p, hi_p - lo_p)
                torch.cuda.synchronize(dev); t0 = time.perf_counter()
                n_p = 30
                for _ in range(n_p):
                    train_epoch(mp, order_p, B_p, hi_p - lo_p)
                torch.cuda.synchronize(dev); tp = 1e3 * (time.perf_counter() - t0) / n_p
                tot = tp + (recon_local_ms or 0.0) + ar_ms
                ent[sched] = {"launches_per_epoch_per_replica": len(batches_p), "batch_users_per_launch": B_p, "train_ms_per_epoch_rank0": tp,
                              "ms_per_epoch_with_reconciliation": tot, "projected_speedup": (1e3 * dt / a.steps) / tot}
                del mp, tab_p
            ent["allreduce_ms_estimate"] = ar_ms
            proj["worlds"]["N=%d" % Nw] = ent
        proj["reconciliation_local_ms"] = recon_local_ms
        proj["reconciliation_bytes"] = recon_bytes
        proj["note"] = ("PROJECTION from one GPU, not a measured curve: rank 0's shard (users balanced by check-ins) trained alone, + the reconciliation's local kernels "
                        "(measured) + a ring all-reduce of the flat delta buffer at one xGMI link (153 GB/s, estimate); quality = as many launches per replica "
                        "and epoch as the one-GPU run (learns like it, DESIGN.md section 7), throughput = launches of ~--batch-users users")
        multi["projection"] = proj

    # ---- secondary_bpr: the BPR-MF step (flag 0: OboBpr.bpr_train, public/BPR.py:201-241) - the one piece of the path that IS a pure gather / scatter:
    # every (user, positive, negative) triple of an epoch (prog_bpr_gru_spatial.py:240-244), snapshot mode (sorted, no float atomics: bitwise reproducible)
    secondary_bpr = None
    if solo and not a.no_bpr and a.shape in ("gowalla", "x1"):
        mb = poi_amd.models.OboBpr(train=tab, test=None, alpha_lambda=[0.01, 0.001], n_user=n_local, n_item=n_item, n_in=D, n_hidden=D, device=dev, seed=7,
                                   table_dtype=a.table_dtype)
        ctx.set_batch_cap(a.batch_cap)
        bu, bp, bq = mb.epoch_triples()
        nt = int(bu.numel())
        e_lt = 2.0 if a.table_dtype == "f16" else 4.0
        bytes_triple = 2.0 * D * 4 + 4.0 * D * e_lt + 12.0      # three rows read, three rows written (user row float32, POI rows in the table's type), three int32
        table_mb = ((n_item + 1) * D * e_lt + n_local * D * 4.0) / 1e6
        resident = table_mb < 256.0
        secondary_bpr = {"workload": "BPR-MF step over the %d (user, positive, negative) triples of one epoch of the %s shape, dim %d, %s POI table (%.0f MB of tables: %s)"
                                     % (nt, a.shape, D, a.table_dtype, table_mb, "inside the 256 MB cache behind L2 - the rates below are NOT HBM traffic" if resident else "streamed from HBM"),
                         "bytes_per_triple_survey_8d": bytes_triple, "tables_MB": table_mb, "tables_cache_resident": resident, "launches": {}}
        hu, hp, hq = bu.cpu().numpy(), bp.cpu().numpy(), bq.cpu().numpy()
        for Bt, mode in ((nt, "snapshot"), (262144, "snapshot")) + (((nt, "hogwild"),) if a.table_dtype == "f32" else ()):
            def bpr_epoch():
                for b0 in range(0, nt, Bt):
                    mb.train_batch(bu[b0:b0 + Bt], bp[b0:b0 + Bt], bq[b0:b0 + Bt], mode=mode, sync=False)
            for _ in range(3):
                bpr_epoch()
            n_rep = 20 if a.shape == "gowalla" else 5
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for _ in range(n_rep):
                bpr_epoch()
            torch.cuda.synchronize(dev); tb = (time.perf_counter() - t0) / n_rep
            ctx.timing(True)
            for _ in range(2):
                bpr_epoch()
            ktb = {k: ctx.timing_get(k) for k in ("bpr_sort", "bpr_users", "bpr_items", "bpr_hogwild")}
            ctx.timing(False)
            ent = {"ms_per_epoch": 1e3 * tb, "triples_per_s": nt / tb,
                   # the contract count (what the per-triple reference formulation moves) over the measured time: an EQUIVALENT rate - the snapshot kernels
                   # move fewer rows than that (sorted touches: a run's row is read once), and tables below 256 MB never leave the cache
                   "reference_formulation_equivalent_GBs": nt * bytes_triple / tb / 1e9,
                   "reference_formulation_equivalent_frac_of_hbm_peak": nt * bytes_triple / tb / 1e9 / PEAK_HBM_GBS,
                   "regions_us_per_launch": {k: 1e3 * v[0] / max(v[1], 1) for k, v in ktb.items() if v[1]}}
            if mode == "snapshot":
                # bytes the sorted kernels move (csrc/bpr.hip), from the launch compositions themselves: per launch, users pass = one ux row per distinct user +
                # two POI rows per triple in, one shadow row per distinct user out; items pass = one ux row per POI touch in, every distinct POI row read + written;
                # commit = shadow -> ux per distinct user; sort = three 9-bit passes over 3 n (key, value) pairs, read + written
                moved = 0.0
                for b0 in range(0, nt, Bt):
                    n_b = min(Bt, nt - b0)
                    du = len(np.unique(hu[b0:b0 + n_b])); di_ = len(np.unique(np.concatenate((hp[b0:b0 + n_b], hq[b0:b0 + n_b]))))
                    moved += (du * D * 4.0 + 2.0 * n_b * D * e_lt + du * D * 4.0) + (2.0 * n_b * D * 4.0 + 2.0 * di_ * D * e_lt) + 2.0 * du * D * 4.0 + 3.0 * 3 * n_b * 8.0 * 2
                ent["moved_bytes_model_per_epoch"] = moved
                ent["roofline"] = {"bound": "hbm", "achieved": moved / tb / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": moved / tb / 1e9 / PEAK_HBM_GBS,
                                   "note": "bytes the sorted kernels move (model from the launch's distinct rows) over the measured time" +
                                           ("; the tables sit in the 256 MB cache: an L2 / cache rate, not HBM" if resident else "; the POI table streams from HBM")}
            secondary_bpr["launches"]["%s, %d triples per launch" % (mode, Bt)] = ent
        secondary_bpr["note"] = ("snapshot = every triple at the launch-entry values, rows combined by the capped-sum rule, 3 n table touches sorted by row and summed in a fixed "
                                 "order (csrc/bpr.hip: no float atomics, bitwise reproducible - tests/test_gpu_bpr.py); hogwild = the racy in-place kernel, for scale")
        secondary_bpr["finite"] = bool(torch.isfinite(mb.lt.t[:1 << 20].float()).all() and torch.isfinite(mb.ux.t).all())
        del mb

    # ---- secondary_dd25: the reference's other spatial configuration (dd = 25 m: 1520 bins, public/GRU_Spatial.py:247), training only --------
    secondary_dd25 = None
    if solo and not a.no_secondary and a.shape == "gowalla" and a.dd == 200.0:
        ds5 = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928 + 2, local=a.local, dd=25.0, ud_km=38.0)
        tab5 = ds5.shard(0, n_user)
        m5 = poi_amd.models.OboSpatialGru(train=tab5, test=None, dist=None, alpha_lambda=[0.01, 0.001], n_user=n_user, n_item=n_item,
                                          n_dists=[ds5.dist_num, ds5.dd / 1000.0], n_in=D, n_hidden=D, device=dev, seed=7, coords=ds5.coords)
        for _ in range(3):
            train_epoch(m5, order, B, n_local)
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        for _ in range(40):
            train_epoch(m5, order, B, n_local)
        torch.cuda.synchronize(dev); t5 = time.perf_counter() - t0
        secondary_dd25 = {"workload": "gowalla shape with dd = 25 m, UD = 38 km: %d distance bins (chunked head, per-bin tables beyond 256 bins)" % ds5.dist_num,
                          "train_seq_per_s": n_user * 40 / t5, "ms_per_epoch": 1e3 * t5 / 40}
        del m5, tab5, ds5

    # ---- secondary_x1: one GPU's slice of BASELINE.json configs[4] (10 M POIs, 125 k users, dim 256, half POI table), training only -----
    secondary_x1 = None
    if solo and not a.no_x1 and a.shape == "gowalla":
        import subprocess
        try:
            t0 = time.perf_counter()
            x1_path = os.path.join(os.path.dirname(full_out_path(a)), "bench_full_x1.json")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--shape", "x1", "--steps", "8", "--warmup", "2", "--eval-steps", "2", "--full-out", x1_path],
                               capture_output=True, text=True, timeout=300)
            j = json.load(open(x1_path))
            secondary_x1 = {"workload": j["config"]["workload"], "table_storage": j["table_storage"], "f16_rounding": j["config"].get("f16_rounding"),
                            "train_seq_per_s": j["value"], "ms_per_epoch": j["ms_per_step"], "batch_users_per_launch": j["config"]["batch_users_per_launch"],
                            "dominant_kernel": j["roofline"]["kernel"], "dominant_frac": j["roofline"]["frac"], "wall_s_incl_data_generation": time.perf_counter() - t0,
                            "gather_scatter": {k: (j.get("roofline_gather_scatter") or {}).get(k) for k in ("kernels", "ms_per_epoch", "frac", "traffic")},
                            "bpr": (lambda L: L and {"workload": j["secondary_bpr"]["workload"], "triples_per_s": L["triples_per_s"], "ms_per_epoch": L["ms_per_epoch"],
                                                     "frac_survey_8d": L["reference_formulation_equivalent_frac_of_hbm_peak"], "frac_bytes_moved": (L.get("roofline") or {}).get("frac"),
                                                     "regions_us_per_launch": L["regions_us_per_launch"]})(
                                next((v for k, v in ((j.get("secondary_bpr") or {}).get("launches") or {}).items() if k.startswith("snapshot")), None)),
                            "gather_scatter_frac_bytes_moved": ((j.get("roofline_gather_scatter") or {}).get("implementation") or {}).get("frac"),
                            "eval_users_per_s": j.get("eval_users_per_s"), "eval_ms_per_8192_users_x_10M_pois": (j.get("eval") or {}).get("ms_per_eval"),
                            "eval_filter_ms": ((j.get("eval") or {}).get("two_stage") or {}).get("ms_filter_per_eval"),
                            "tests": "tests/test_gpu_configx.py: the 10 M x 256 half table at full size (touched rows vs the float64 oracle, > 2^31-element indexing, stochastic rounding, GEO top-K over 10 M POIs)"}
        except Exception as e:          # (the block is informational: a failure must not cost the headline line)
            secondary_x1 = {"error": repr(e)[:300]}

    # ---- CPU baseline: plain-C float64 port of the same per-sequence algorithm, 1 thread -------------
    cpu = None
    if solo and not a.no_cpu_baseline:
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

            def work_fn(i):
                C.spatial_epoch(reps[i], tab.off, tab.p, tab.q, tab.dp, tab.dq, ordr[i * Sc:(i + 1) * Sc], tab.len_max, 0.01, 0.001)
            t0 = time.perf_counter()
            with ThreadPoolExecutor(ncore) as ex:
                list(ex.map(work_fn, range(ncore)))
            ta = time.perf_counter() - t0
            all_cores = {"value": ncore * Sc / ta, "unit": "sequences/s", "cores": ncore,
                         "sample": "%d user-sharded replicas x %d sequences" % (ncore, Sc)}
        cpu = {"value": S / tc, "unit": "sequences/s", "cores": 1, "kind": "port", "all_cores": all_cores,
               "sample": "%d sequences of the same shuffled order, sequential per-user SGD (reference semantics), "
                         "plain-C float64 port of public/GRU_Spatial.py:127-229 (Theano cannot be built or shipped)" % S,
               "eval_users_per_s": ne / te, "host_cores_available": os.cpu_count()}

    if secondary_x1 and secondary_x1.get("gather_scatter"):
        # config X's slice: the one shape whose POI table (5.1 GB of half rows) streams from HBM instead of sitting in the 256 MB cache
        hbm["x1"] = {"frac_survey_8d": secondary_x1["gather_scatter"].get("frac"), "frac_bytes_moved": secondary_x1.get("gather_scatter_frac_bytes_moved"),
                     "ms_per_epoch": secondary_x1["gather_scatter"].get("ms_per_epoch")}
    if rank == 0:
        out = {
            "metric": "check-in sequences/sec training (Distance2Pre) + all-POI top-K eval users/sec",
            "value": seq_per_s, "unit": "sequences/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (forward pass: int8 x 5 fixed point + f64 gates)" if xfwd else "f32", "table_storage": a.table_dtype, "data": "synthetic",
            "config": {"workload": "synthetic %s-shape: %d POIs, %d users, seq<=%d, dim=%d, %d distance bins; one step = one "
                                   "Distance2Pre training epoch over all users" % (a.shape, n_item, n_user, max_len, D, ds.dist_num),
                       "batch_users_per_launch": B, "batch_rule": "capped sum: a row touched by k sequences of a launch moves by min(k, %g)/k x the sum of "
                                                                  "their reference updates (include/poi_hip.h); see `quality` for what it learns" % a.batch_cap,
                       "batch_cap": a.batch_cap, "local_transition_fraction": a.local,
                       "parallelism": "user-shard x%d, per-epoch delta all-reduce" % world,
                       "replica_schedule": a.replica_schedule if (world > 1 or a.emulate_world) else None, "launches_per_epoch_per_replica": len(batches),
                       "f16_rounding": a.f16_rounding if a.table_dtype == "f16" else None,
                       "alpha": 0.01, "lambda": 0.001, "engine": "tile" if "te_rec_fwd" in kernels else "per-sequence",
                       "epoch_refresh": ("every epoch redraws its negatives + negative distance bins on the device (inside the clock) and takes a new shuffled "
                                         "user order (64 distinct orders drawn before the timed region, cycled): prog_bpr_gru_spatial.py:221-238") if refresh else
                                        "none: one fixed order and negative table (--no-refresh)",
                       "arithmetic": ("forward pass (input product, recurrence, gates): ~40-bit fixed point on the int8 matrix cores + float64 gate math, results "
                                      "rounded to f32 for the head / BPTT / gradient / write-back kernels, which compute in f32 (te_head3, te_rec_bwd, te_wgrad: every product from five / six bf16 partial products, f32 accumulation)") if xfwd else
                                     "f32 results throughout; the recurrent kernels of launches above the small-launch bound and the forward table "
                                     "(te_gemm_ax of large launches: te_ptab_s3) form their f32 products from three bf16 planes per operand (six MFMA "
                                     "partial products, f32 accumulate; <=5.1e-6 of the f64 oracle, same bar as the f32 MFMA path -- "
                                     "tests/test_gpu_tile_engine.py): their `frac` entries under `kernels` are f32-equivalent flops over the f32 "
                                     "matrix peak; te_wgrad / te_head / te_gemm_dx use the f32 MFMA",
                       "s_rows_per_step": rho},
            "timed_window_s": dt,
            "kernels": kernels, "quality": quality, "multi_gpu": multi, "secondary": secondary, "secondary_dd25": secondary_dd25, "launch_sweep": launch_sweep, "secondary_x1": secondary_x1, "secondary_bpr": secondary_bpr,
            "train_step_tflops": executed_flops / (train_kernel_ms * 1e-3) / 1e12 if train_kernel_ms > 0 else None,
            "train_step_tflops_reference_formulation": total_flops / (train_kernel_ms * 1e-3) / 1e12 if train_kernel_ms > 0 else None,
            "reference_schedule": reference_schedule, "eval": eval_detail, "roofline_gather_scatter": hbm,
            "roofline": roofline, "cpu_baseline": cpu,
            "steady_window": steady, "exact_mode": exact_mode, "eval_users_per_s": eval_users_per_s,
        }
        # compact recap LAST: a record that keeps only the tail of this line still holds every headline number
        out["headline"] = {
            "train_seq_per_s": seq_per_s, "ms_per_epoch": 1e3 * dt / a.steps, "steady_seq_per_s": steady and steady["seq_per_s"],
            "eval_users_per_s": eval_users_per_s, "ms_per_eval": eval_detail.get("ms_per_eval"), "eval_users_per_s_seeded": eval_detail.get("eval_users_per_s_seeded"),
            "filter_frac_of_f16_mfma_peak": (eval_detail.get("two_stage") or {}).get("filter_frac_of_f16_mfma_peak"),
            "dominant_kernel": roofline["kernel"], "dominant_frac": roofline["frac"],
            "gather_scatter_frac_survey_8d": (hbm.get("survey_8d") or {}).get("frac"), "gather_scatter_ms_per_epoch": hbm["ms_per_epoch"],
            "exact_seq_per_s": exact_mode and exact_mode["seq_per_s"], "exact_slowdown": exact_mode and exact_mode["slowdown"],
            "float32_forward_seq_per_s": exact_mode and exact_mode.get("float32_forward_seq_per_s"),
            "float64_engine_seq_per_s": exact_mode and (exact_mode.get("float64_engine") or {}).get("seq_per_s"),
            "reference_schedule_steps_per_s": reference_schedule and reference_schedule["seq_per_s"],
            "recall_headline_vs_reference": quality and quality["headline_vs_reference"]["recall_ratio"],
            "time_to_reference_recall_s": quality and {k: (v.get("seconds") if isinstance(v, dict) else v) for k, v in quality["time_to_recall"].items() if k.startswith("B=")},
            "x1_train_seq_per_s": secondary_x1 and secondary_x1.get("train_seq_per_s"), "x1_eval_users_per_s": secondary_x1 and secondary_x1.get("eval_users_per_s"),
            "dd25_1520_bins_train_seq_per_s": secondary_dd25 and secondary_dd25["train_seq_per_s"],
            "bpr_triples_per_s": secondary_bpr and max(v["triples_per_s"] for k, v in secondary_bpr["launches"].items() if k.startswith("snapshot")),
            "bpr_reference_formulation_equivalent_frac": secondary_bpr and max(v["reference_formulation_equivalent_frac_of_hbm_peak"] for k, v in secondary_bpr["launches"].items() if k.startswith("snapshot")),
            "bpr_x1_triples_per_s": secondary_x1 and (secondary_x1.get("bpr") or {}).get("triples_per_s"),
            "bpr_x1_frac_of_hbm_on_bytes_moved": secondary_x1 and (secondary_x1.get("bpr") or {}).get("frac_bytes_moved"),
            "bpr_x1_frac_of_hbm_survey_8d": secondary_x1 and (secondary_x1.get("bpr") or {}).get("frac_survey_8d"),
            "cpu_1core_seq_per_s": cpu and cpu["value"], "cpu_allcores_seq_per_s": cpu and cpu["all_cores"] and cpu["all_cores"]["value"],
        }
        emit(a, out)
    sync.close()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
