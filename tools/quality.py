#!/usr/bin/env python
"""Learning-quality experiments on synthetic data WITH a next-POI signal (data.make_synthetic(local=0.8)).

    python tools/quality.py --shape foursquare --batch 1 --epochs 4
    python tools/quality.py --shape foursquare --batch 256 --alpha 0.01 --cap 64 --epochs 20
    python tools/quality.py --shape foursquare --batch 5000 --world 8 --rules default      # N replicas emulated on one GPU

Prints one JSON line per epoch: wall time spent TRAINING so far, recall@20, AUC; --seconds stops on a time budget."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poi_amd
from poi_amd import data as pdata, harness
from poi_amd.evaluate import GlobalBest, fun_predict_auc_recall_map_ndcg


def evaluate(model, ds, p, ses_tes, ses_auc):
    model.update_trained_items(); model.update_trained_dists()
    hs, ss = zip(*[model.predict_device(se) for se in ses_tes])
    model.update_trained_users(torch.cat(hs)); model.update_trained_sus(torch.cat(ss))
    m = fun_predict_auc_recall_map_ndcg(p, model, GlobalBest(p["at_nums"]), 0, ses_auc, ses_tes, ds.tes_p.reshape(-1, 1),
                                        np.ones((ds.n_user, 1), np.int32))
    return m["at"][20]["recall"], m["auc"]


def main(argv=None):
    """Runs the experiment; returns the list of per-evaluation records (also printed as JSON lines)."""
    records = []
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="foursquare")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=0.01)
    ap.add_argument("--cap", type=float, default=1.0, help="batch rule cap (poi_ctx_set_batch_cap): 1 = mean rule")
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--rules", default="default", help="default | sum | mean | mean_touched (all tensors)")
    ap.add_argument("--shard-batch", type=int, default=0, help="users per launch inside a replica (default: --batch)")
    ap.add_argument("--local", type=float, default=0.8)
    ap.add_argument("--users", type=int, default=0)
    ap.add_argument("--dim", type=int, default=0)
    ap.add_argument("--eval-every", type=int, default=1)
    ap.add_argument("--limit", type=int, default=0, help="train only the first LIMIT users of each epoch's shuffled order (time-boxed B=1 runs)")
    a = ap.parse_args(argv)
    n_item, n_user, max_len, D = pdata.SHAPES[a.shape]
    n_user = a.users or n_user; D = a.dim or D
    ds = pdata.make_synthetic(n_user, n_item, max_len, seed=20260928, local=a.local)
    p = harness.default_params(); p.update(latent_size=D, alpha=a.alpha, gru=2)
    model = harness.build_model(ds, p, seed=2)
    model.ctx.set_batch_cap(a.cap)
    pop = np.bincount(ds.tra_p, minlength=n_item); top = np.argsort(-pop)[:20]
    print(json.dumps({"random_recall@20": 20.0 / n_item, "popularity_recall@20": float(np.isin(ds.tes_p, top).mean()),
                      "last_poi_neighbourhood": "transitions go to one of the 32 nearest POIs with p=%.2f" % a.local}), flush=True)
    ses_tes = harness.compute_start_end(n_user, 16384); ses_auc = ses_tes
    lens = ds.lens
    names = [n for n in ("lt", "di", "ui", "wh", "bi", "vs", "bs", "wd", "loss_weight")]
    sync = None
    if a.world > 1:
        rules = None if a.rules == "default" else {n: a.rules for n in names}
        sync = poi_amd.dist.model_sync(model, rules=rules, force_backend=True)
        bounds = [pdata.shard_users(n_user, a.world, r, lens) for r in range(a.world)]
    t_train = 0.0
    for epoch in range(a.epochs):
        if epoch > 0:
            model.resample_negatives_device(7 * 1000003 + epoch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        order = np.random.default_rng(123 + epoch).permutation(n_user).astype(np.int32)
        if a.limit:
            order = order[:a.limit]

        def run(ids):
            if a.batch <= 1:
                for u in ids:
                    model.train(np.int32(u))
            else:
                nl = max(1, int(round(len(ids) / float(a.batch)))); Be = -(-len(ids) // nl)
                for b0 in range(0, len(ids), Be):
                    sub = ids[b0:b0 + Be]
                    model.train_batch(torch.as_tensor(sub[np.argsort(-lens[sub], kind="stable")]).cuda(), sync=False)
        if sync is None:
            run(order)
        else:
            start = [getattr(model, n).t.clone() for n in names]
            sync.backend.begin_epoch()
            total = torch.zeros_like(sync.backend.flat)
            f16 = sync.backend.flat16                           # deltas of a half-stored table (None with float32 tables)
            total16 = torch.zeros(f16.numel(), device=f16.device) if f16 is not None else None
            for lo, hi in bounds:
                for n, s in zip(names, start):
                    getattr(model, n).t.copy_(s)
                run(order[(order >= lo) & (order < hi)])
                total += sync.backend.make_delta()
                if f16 is not None:
                    total16 += f16.float()
            for n, s in zip(names, start):
                getattr(model, n).t.copy_(s)
            sync.backend.flat.copy_(total)
            if f16 is not None:
                f16.copy_(total16.half())
            sync.backend.apply(a.world)
        torch.cuda.synchronize(); t_train += time.perf_counter() - t0
        last = epoch == a.epochs - 1 or (a.seconds and t_train >= a.seconds)
        if (epoch + 1) % a.eval_every and not last:
            continue
        rec, auc = evaluate(model, ds, p, ses_tes, ses_auc)
        records.append({"epoch": epoch, "recall": rec, "auc": auc, "train_s": t_train})
        print(json.dumps({"epoch": epoch, "train_s": round(t_train, 3), "seq_per_s": round((a.limit or n_user) * (epoch + 1) / t_train, 1),
                          "recall@20": round(rec, 4), "auc": round(auc, 4), "batch": a.batch, "cap": a.cap, "alpha": a.alpha, "world": a.world}), flush=True)
        if a.seconds and t_train >= a.seconds:
            break
    model.ctx.set_batch_cap(1.0)
    if sync is not None:
        sync.close()
    return records


if __name__ == "__main__":
    main()
