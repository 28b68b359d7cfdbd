python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/t.log 2>&1; grep -E "passed|failed|Error|assert|FAILED" gpurun_out/t.log | tail -20
python - <<EOP
import sys; sys.path.insert(0,'.')
import numpy as np, time, torch
import poi_amd
from poi_amd import data as pdata
# D = 256 throughput: gowalla-size user set, 100k POIs
ds = pdata.make_synthetic(50000, 100000, 50, seed=5, local=0.8)
tab = ds.shard(0, 50000)
for D in (128, 256):
    m = poi_amd.models.OboSpatialGru(train=tab, test=None, dist=None, alpha_lambda=[0.01,0.001], n_user=50000, n_item=100000, n_dists=[ds.dist_num, ds.dd/1000.0], n_in=D, n_hidden=D, seed=7, coords=ds.coords)
    m.ctx.set_batch_cap(64.0)
    lens = ds.lens
    ids = np.arange(12500); ids = ids[np.argsort(-lens[ids], kind="stable")]
    t = torch.as_tensor(ids.astype(np.int32)).cuda()
    for _ in range(3): m.train_batch(t, sync=False)
    m.ctx.timing(True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): m.train_batch(t, sync=False)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print("D", D, "ms/launch", dt*1e3, "seq/s", 12500/dt, {k: round(m.ctx.timing_get(k)[0]/10,3) for k in ("te_gemm_ax","te_rec_fwd","te_head","te_rec_bwd","te_wgrad","te_gemm_dx","te_scatter","te_dsum","te_bin_gemm")})
    m.ctx.timing(False)
EOP
