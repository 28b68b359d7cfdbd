"""Evaluator in the shape of public/Valuate.py: AUC from model.compute_sub_auc_preference, top-K from
the fused device kernel (model.compute_sub_topk) instead of (n, N) score matrices + per-row numpy
argpartition, then hit / recall / precision / F1 / MAP / NDCG @ at_nums (Valuate.py:23-88,149-172),
vectorised over users."""
from __future__ import annotations

import numpy as np


class GlobalBest:
    """Best-so-far bookkeeping (public/Global_Best.py:21-82, without the printing)."""

    def __init__(self, at_nums):
        self.at_nums = list(at_nums)
        z = lambda: np.zeros(len(at_nums))
        self.best_auc, self.best_epoch_auc = 0.0, 0
        for n in ("recall", "precis", "f1scor", "map", "ndcg"):
            setattr(self, "best_" + n, z())
            setattr(self, "best_epoch_" + n, np.zeros(len(at_nums), int))


def rank_metrics(all_ranks, tes_buys_masks, tes_masks, at_nums):
    """Valuate.py:149-172 on an (U, Kmax) rank matrix.  Returns {k: dict(hits, recall, precision, f1, map, ndcg)}."""
    ranks = np.asarray(all_ranks)
    tes = np.asarray(tes_buys_masks)
    msk = np.asarray(tes_masks).astype(bool)
    n_test = msk.sum(axis=1)                                       # len(test_lst) per user
    denom = float(msk.sum())
    out = {}
    for k in at_nums:
        rec = ranks[:, :k]
        # zero_one[u, r] = 1 if rec[u, r] is one of the user's valid test items (fun_hit_zero_one)
        zo = ((rec[:, :, None] == tes[:, None, :]) & msk[:, None, :]).any(axis=2).astype(np.int64)
        hits = float(zo.sum())
        recall = hits / denom
        precis = hits / (k * len(zo))
        f1 = 2.0 * recall * precis / (recall + precis) if recall + precis > 0 else 0.0
        cum = zo.cumsum(axis=1) * zo                               # fun_evaluate_map
        ap = (cum / np.arange(1, k + 1)[None, :]).sum(axis=1) / np.maximum(n_test, 1)
        disc = 1.0 / np.log2(np.arange(k) + 2.0)                   # fun_evaluate_ndcg
        dcg = (zo * disc[None, :]).sum(axis=1)
        ideal = np.array([disc[:min(int(t), k)].sum() for t in n_test])
        ndcg = np.where(zo.sum(axis=1) > 0, dcg / np.maximum(ideal, 1e-300), 0.0)
        out[k] = dict(hits=hits, recall=recall, precision=precis, f1=f1, map=float(ap.mean()), ndcg=float(ndcg.mean()))
    return out


def coalesce_ranges(starts_ends, target=65536):
    """Merge consecutive contiguous id ranges into calls of up to `target` users.  The reference evaluates in
    batch_size_test users per call (Params.compute_start_end); the metrics are sums over users, so the grouping is
    free - and the two-stage scoring path is at its best with as many users per call as there are (call sizes;
    the cap bounds the per-call survivor lists: 32 KB per user)."""
    out, cur = [], None
    for se in starts_ends:
        se = np.asarray(se)
        contiguous = len(se) > 0 and np.all(np.diff(se) == 1)
        if cur is not None and contiguous and len(cur) and cur[-1] + 1 == se[0] and len(cur) + len(se) <= target:
            cur = np.concatenate((cur, se))
        else:
            if cur is not None:
                out.append(cur)
            cur = se
    if cur is not None:
        out.append(cur)
    return out


def device_rank_metrics(model, starts_ends_tes, at_nums):
    """Fused top-K + metric accumulation on the device: only (len(at_nums), 3) doubles reach the host."""
    import ctypes
    import torch
    at_nums = list(at_nums)
    if not at_nums or len(at_nums) > 8 or any(b <= a for a, b in zip(at_nums, at_nums[1:])) or at_nums[0] <= 0 or at_nums[-1] > 64:
        raise ValueError("at_nums must be 1..8 strictly ascending cut-offs <= 64 (got %r)" % (at_nums,))
    kmax = at_nums[-1]
    acc = torch.zeros((len(at_nums), 3), dtype=torch.float64, device=model.device)
    at = torch.as_tensor(np.asarray(at_nums, np.int32)).to(model.device)
    for se in coalesce_ranges(starts_ends_tes):
        ids, lo = model._ids(se)
        idx = model.compute_sub_topk(se, kmax)
        tp, tm = model._rows(model.tes_buys_masks, ids, lo), model._rows(model.tes_masks, ids, lo)
        model.ctx.check(model.lib.poi_rank_metrics(model.ctx.handle, idx.data_ptr(), idx.shape[0], kmax, tp.data_ptr(), tm.data_ptr(),
                                                   tm.shape[1], at.data_ptr(), len(at_nums), acc.data_ptr(), model._stream()))
    a = acc.cpu().numpy()
    n_user = model.n_user
    denom = float(model.tes_masks.sum().item())
    out = {}
    for i, k in enumerate(at_nums):
        hits = float(a[i, 0]); rec = hits / denom; pre = hits / (k * n_user)
        out[k] = dict(hits=hits, recall=rec, precision=pre, f1=2.0 * rec * pre / (rec + pre) if rec + pre > 0 else 0.0,
                      map=float(a[i, 1]) / n_user, ndcg=float(a[i, 2]) / n_user)
    return out


def fun_predict_auc_recall_map_ndcg(p, model, best, epoch, starts_ends_auc, starts_ends_tes, tes_buys_masks, tes_masks, on_device=True):
    """Same signature and side effects on `best` as public/Valuate.py:103-191; returns the metrics too.
    With on_device (default) the ranks never leave the GPU; on_device=False downloads them and uses the
    vectorised numpy restatement (rank_metrics) - both are tested against the reference's helpers."""
    at_nums = p["at_nums"]
    upqs = np.concatenate([model.compute_sub_auc_preference(se) for se in starts_ends_auc])
    auc = float(upqs.sum()) / float(np.sum(tes_masks))             # Valuate.py:113-118
    if auc > best.best_auc:
        best.best_auc, best.best_epoch_auc = auc, epoch
    kmax = at_nums[-1]
    if on_device:
        all_ranks = None
        m = device_rank_metrics(model, starts_ends_tes, at_nums)
    else:
        all_ranks = np.concatenate([model.compute_sub_topk(se, kmax).cpu().numpy() for se in starts_ends_tes])
        m = rank_metrics(all_ranks, tes_buys_masks, tes_masks, at_nums)
    for i, k in enumerate(at_nums):
        for name, key in (("recall", "recall"), ("precis", "precision"), ("f1scor", "f1"), ("map", "map"), ("ndcg", "ndcg")):
            cur = getattr(best, "best_" + name)
            if m[k][key] > cur[i]:
                cur[i] = m[k][key]
                getattr(best, "best_epoch_" + name)[i] = epoch
    return dict(auc=auc, at=m, ranks=all_ranks)
