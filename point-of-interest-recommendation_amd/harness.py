"""Epoch loop in the shape of prog_bpr_gru_spatial.py:182-334 (train_valid_or_test) on a PoiDataset:
per-epoch negative refresh -> shuffled user order -> train (one user per step like the reference, or
`batch_users` per launch) -> snapshots -> predict -> evaluate.  The reference's three-way timing split
(train / user vectors / test, :232,264,299,305-310) is kept."""
from __future__ import annotations

import os
import pickle
import time

import numpy as np
import torch

from . import models
from .evaluate import GlobalBest, fun_predict_auc_recall_map_ndcg


def compute_start_end(user_num, size):
    """Params.compute_start_end (prog_bpr_gru_spatial.py:156-179): contiguous np.arange batches."""
    return [np.arange(s, min(s + size, user_num), dtype=np.int32) for s in range(0, user_num, size)]


def default_params():
    """The in-source config of prog_bpr_gru_spatial.py:54-78 (flag 2 = Distance2Pre)."""
    return dict(at_nums=[5, 10, 15, 20], epochs=3, latent_size=20, alpha=0.01, **{"lambda": 0.001}, gru=2,
                batch_size_test=32, batch_users=1, seed=123,
                dataset="synthetic", UD=40, dd=200, load_epoch=0, save_per_epoch=0)       # :54-78 (checkpoint naming / cadence)


def load_dataset(p, root="."):
    """Params.__init__ of the driver (prog_bpr_gru_spatial.py:81-91): p['dataset'] names a sequence file in the ETL's format
    ('Foursquare.txt' / 'Gowalla.txt' under `root`, or any path) -> data.load_sequence_file with p['split'] (-1 test / -2 valid), p['dd'] and
    dist_num = int(UD * 1000 / dd) (:81).  'synthetic' / 'synthetic:<shape>' -> the generator (data.SHAPES)."""
    from . import data as pdata
    name = str(p.get("dataset", "synthetic"))
    dist_num = int(p["UD"] * 1000 / p["dd"])
    if name.startswith("synthetic"):
        shape = name.split(":", 1)[1] if ":" in name else "tiny"
        n_item, n_user, max_len, _ = pdata.SHAPES[shape]
        return pdata.make_synthetic(n_user, n_item, max_len, seed=p.get("seed", 0), dd=p["dd"], ud_km=p["UD"], local=p.get("local", 0.8))
    path = name if os.path.exists(name) else os.path.join(root, name)
    if not os.path.exists(path):
        raise FileNotFoundError("dataset %r: no such sequence file (looked at %s)" % (name, path))
    return pdata.load_sequence_file(path, split=p.get("split", -1), dd=p["dd"], dist_num=dist_num, seed=p.get("seed", 0))


def build_model(ds, p, device="cuda:0", seed=None):
    tab = ds.shard()
    size = p["latent_size"]
    al = [p["alpha"], p["lambda"]]
    if p["gru"] == 0:
        return models.OboBpr(train=tab, test=None, alpha_lambda=al, n_user=ds.n_user, n_item=ds.n_item, n_in=size, n_hidden=size,
                             device=device, seed=seed)
    if p["gru"] == 1:
        return models.OboGru(train=tab, test=None, alpha_lambda=al, n_user=ds.n_user, n_item=ds.n_item, n_in=size, n_hidden=size,
                             device=device, seed=seed)
    if p["gru"] == 3:                                                   # prog_bpr_gru_spatial.py:141-151
        return models.OboCARNN(train=tab, test=None, dist=None, alpha_lambda=al, n_user=ds.n_user, n_item=ds.n_item,
                               n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=size, n_hidden=size, device=device, seed=seed, coords=ds.coords)
    return models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=al, n_user=ds.n_user, n_item=ds.n_item,
                                n_dists=[ds.dist_num, ds.dd / 1000.0], n_in=size, n_hidden=size, device=device, seed=seed,
                                coords=ds.coords)


CKPT_ORDER = ("loss_weight", "wd", "lt", "di", "ui", "wh", "bi", "vs", "bs")      # prog_bpr_gru_spatial.py:325-327


def checkpoint_path(p, model_name, epoch, root="./model"):
    """File name of prog_bpr_gru_spatial.py:207-209,321-322."""
    return os.path.join(root, os.path.basename(str(p["dataset"])), "%s_size%s_UD%s_dd%s_epoch%s" % (model_name, p["latent_size"], p["UD"], p["dd"], epoch))


def _py2_compatible(stream):
    """Rewrite the GLOBAL opcodes of a protocol-2 pickle so that numpy >= 2's private module path
    (numpy._core.multiarray) becomes the public one every numpy since 1.x exports (numpy.core.multiarray): the
    reference unpickles these files with Python 2 + an old numpy, which has no numpy._core.  GLOBAL arguments are
    newline-terminated text ("c<module>\\n<name>\\n"), so the opcode is rewritten in place, opcode by opcode
    (pickletools.genops), never by a blind byte replace that could hit array payload."""
    import pickletools
    ops = list(pickletools.genops(stream))
    out, prev = bytearray(), 0
    for i, (op, arg, pos) in enumerate(ops):
        if op.name == "GLOBAL" and arg.startswith("numpy._core."):
            end = ops[i + 1][2] if i + 1 < len(ops) else len(stream)
            mod, name = arg.split(" ")
            out += stream[prev:pos] + b"c" + mod.replace("numpy._core.", "numpy.core.").encode() + b"\n" + name.encode() + b"\n"
            prev = end
    out += stream[prev:]
    return bytes(out)


def dump_checkpoint(values, path):
    """The reference's checkpoint file (prog_bpr_gru_spatial.py:323-330): a pickled list of the nine parameter
    arrays [loss_weight, wd, lt, di, ui, wh, bi, vs, bs] as float64 (Theano's floatX there), protocol 2 =
    cPickle.HIGHEST_PROTOCOL of Python 2, with module paths an old numpy can import (_py2_compatible) - readable
    by the reference's cPickle.load and by load_checkpoint."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(_py2_compatible(pickle.dumps([np.asarray(v, np.float64) for v in values], protocol=2)))


def read_checkpoint(path):
    """List of nine arrays from a checkpoint written by dump_checkpoint or by the reference (Python 2 pickle)."""
    with open(path, "rb") as f:
        objs = pickle.load(f, encoding="latin1")
    if len(objs) != len(CKPT_ORDER):
        raise ValueError("checkpoint %s holds %d objects, expected %d" % (path, len(objs), len(CKPT_ORDER)))
    return [np.asarray(o) for o in objs]


def save_checkpoint(model, path):
    dump_checkpoint([getattr(model, k).get_value() for k in CKPT_ORDER], path)


def load_checkpoint(model, path):
    """model.load_params(cPickle.load(f)) of prog_bpr_gru_spatial.py:210-213."""
    model.load_params(read_checkpoint(path))


def train_valid_or_test(ds, p, device="cuda:0", log=print):
    if ds is None or isinstance(ds, str):                          # a sequence file: BASELINE.json configs[0] through the product
        if isinstance(ds, str):
            p = dict(p, dataset=ds)
        ds = load_dataset(p)
    model = build_model(ds, p, device, seed=p.get("seed"))
    ini_epoch = 0
    if p["gru"] == 2 and p.get("load_epoch", 0):                  # prog_bpr_gru_spatial.py:204-214
        load_checkpoint(model, checkpoint_path(p, model.__class__.__name__, p["load_epoch"], p.get("model_root", "./model")))
        ini_epoch = p["load_epoch"] + 1
    best = GlobalBest(p["at_nums"])
    U = ds.n_user
    ses_tes = compute_start_end(U, p["batch_size_test"])
    ses_auc = compute_start_end(U, p["batch_size_test"] * 10)
    # the reference predicts in batch_size_test-user calls (32 by default: 1563 launches of 0.2 ms per Gowalla epoch = 0.32 s, 30x the
    # training time); the result rows are the same for any chunking, so the predict passes use chunks of >= 16384 users
    ses_pred = compute_start_end(U, max(int(p["batch_size_test"]), 16384))
    tes_p, tes_m = ds.tes_p.reshape(-1, 1), np.ones((U, 1), np.int32)
    lens = ds.lens
    history = []
    rng_neg = np.random.default_rng(p.get("seed", 0) + 1000)
    B = int(p.get("batch_users", 1))
    for epoch in range(ini_epoch, p["epochs"]):
        if epoch > 0:                                               # :221-228 (every epoch after the first, also after a resume)
            if p.get("device_negatives", True):                     # on the GPU: ~0.1 ms instead of ~0.4 s of numpy
                model.resample_negatives_device(p.get("seed", 0) * 1000003 + epoch)
            else:
                ds.resample_negatives(rng_neg)
                model.set_negatives_csr(ds.tra_q, ds.tes_q, ds.tra_dq if p["gru"] in (2, 3) else None)
        t0 = time.time()
        order = np.random.default_rng(123 + epoch).permutation(U).astype(np.int32)      # :236-238
        loss = 0.0
        if p["gru"] == 0:                                           # :240-244 - one triple per valid position
            u, pp, qq = model.epoch_triples()
            if B <= 1:
                off = ds.off.astype(np.int64)
                hp, hq = model.p.cpu().numpy(), model.q.cpu().numpy()      # current (possibly device-refreshed) tables
                for uidx in order:
                    for i in range(off[uidx], off[uidx + 1]):
                        loss += model.train(int(uidx), [int(hp[i]), int(hq[i])])
            else:
                loss = float(model.train_batch(u, pp, qq, mode="snapshot").sum())
        elif B <= 1 and p["gru"] == 2:                              # :246-254, one launch per user without a host round trip per step
            loss += float(model.train_sequence(order)[:, 0].sum())
        elif B <= 1:
            for uidx in order:
                loss += model.train(np.int32(uidx))
        else:
            # launches of equal size (about B users each): no tiny trailing launch
            Be = -(-U // max(1, int(round(U / float(B)))))
            for b0 in range(0, U, Be):
                ids = order[b0:b0 + Be]
                ids = ids[np.argsort(-lens[ids], kind="stable")]
                out = model.train_batch(ids)
                loss += float(out[:, 0].sum()) if p["gru"] == 2 else float(out.sum())
        l2 = model.l2.eval()                                        # :255
        t1 = time.time()
        model.update_trained_items()                                # :266-298
        if p["gru"] == 0:
            model.update_trained_users()
        elif p["gru"] == 1:
            model.update_trained_users(torch.cat([model.predict_device(se) for se in ses_pred]))
        elif p["gru"] == 3:                                         # :293-300
            model.update_trained_dists()
            model.update_trained_users(torch.cat([model.predict_device(se) for se in ses_pred]))
        else:
            model.update_trained_dists()
            hs, ss = zip(*[model.predict_device(se) for se in ses_pred])
            model.update_trained_users(torch.cat(hs))
            model.update_trained_sus(torch.cat(ss))
        t2 = time.time()
        m = fun_predict_auc_recall_map_ndcg(p, model, best, epoch, ses_auc, ses_tes, tes_p, tes_m)
        t3 = time.time()
        history.append(dict(epoch=epoch, loss=loss, l2=l2, auc=m["auc"], recall=[m["at"][k]["recall"] for k in p["at_nums"]],
                            times=(t1 - t0, t2 - t1, t3 - t2)))
        log("epoch %d  sum_loss = %.3f = %.3f + %.3f  auc %.4f  recall@%d %.4f  time (train, user, test) %.2fs %.2fs %.2fs"
            % (epoch, loss + l2, loss, l2, m["auc"], p["at_nums"][-1], m["at"][p["at_nums"][-1]]["recall"], t1 - t0, t2 - t1, t3 - t2))
        if p["gru"] == 2 and p.get("save_per_epoch", 0) and epoch % p["save_per_epoch"] == 0 and epoch != 0:      # :320-330
            save_checkpoint(model, checkpoint_path(p, model.__class__.__name__, epoch, p.get("model_root", "./model")))
    return model, best, history


def cal_s(ds, p, device="cuda:0", out_root="./Lmdd", log=print):
    """Mode 's' of the reference driver (prog_bpr_gru_spatial.py:337-362): build the Distance2Pre model, load the checkpoint of
    p['load_epoch'], snapshot the tables, predict every user and save the (n_user, n_dist + 1) bin probabilities `sts` with np.save under
    ./Lmdd/<dataset>_size<D>_UD<UD>_dd<dd>_epoch<e>last1(.npy).  Returns (path, sts)."""
    if p["gru"] != 2:
        raise ValueError("cal_s is the Distance2Pre (gru = 2) mode of the reference driver")
    model = build_model(ds, p, device, seed=p.get("seed"))
    path = checkpoint_path(p, model.__class__.__name__, p["load_epoch"], p.get("model_root", "./model"))
    log("Loading model ...")
    load_checkpoint(model, path)
    log("\tPredicting ...")
    model.update_trained_items(); model.update_trained_dists()
    all_sus = []
    for se in compute_start_end(ds.n_user, max(int(p["batch_size_test"]), 16384)):          # (:354-356; rows are the same for any chunking)
        all_sus.append(model.predict_device(se)[1])
    sts = torch.cat(all_sus).cpu().numpy()
    os.makedirs(out_root, exist_ok=True)
    out = os.path.join(out_root, "%s_size%s_UD%s_dd%s_epoch%slast1" % (p["dataset"], p["latent_size"], p["UD"], p["dd"], p["load_epoch"]))
    np.save(out, sts)
    return out + ".npy", sts
