import sys; sys.path.insert(0,'.')
import numpy as np
from oracle import poi_oracle as O
from tests.gpu_util import *
import poi_amd
for dim in (128,256):
    T = toy_problem(80 + dim, n_user=75, n_item=200, n_dist=200, dim=dim, len_max=13)
    P = spatial_params(80 + dim, T)
    ids = np.arange(2, 73, dtype=np.int32)
    eh, es = O.spatial_predict(P, P["lt"], P["di"], T["train"][0][ids], T["dist"][0][ids], T["train"][1][ids])
    for eng in ("tile","seq"):
        model = poi_amd.models.OboSpatialGru(train=T["train"], test=T["test"], dist=T["dist"], alpha_lambda=[0.01, 0.001], n_user=T["n_user"], n_item=T["n_item"], n_dists=[T["n_dist"], 0.2], n_in=dim, n_hidden=dim, init=P)
        model.ctx.set_engine(eng)
        model.update_trained_items(); model.update_trained_dists()
        hts, sts = model.predict(ids)
        err = np.abs(hts-eh).max(axis=1)
        print(dim, eng, rel_err(hts,eh), rel_err(sts,es), "rows >1e-5:", int((err>1e-5).sum()), "of", len(err), "median", np.median(err))
