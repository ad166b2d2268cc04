import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import cases, bench
from infercnvpy_amd import _engine
from infercnvpy_amd._plan import GenePlan
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
plan = GenePlan(v["chromosome"], v["start"], window_size=250, step=10)
ip, ix, dv = bench.synth_csr_on_device(torch, 20000, 20000, 0.07, 3)
dm = _engine.DeviceMatrix(indptr=ip, indices=ix, data=dv, shape=(20000, 20000))
ref = (_engine.column_sums(dm)[0] / 20000).float()
raw = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=None)
res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=5000)
torch.cuda.synchronize()
thr = res.thr.cpu().numpy(); print("thr", thr)
for k in range(4):
    a = raw.out[k*5000:(k+1)*5000].abs()
    tf = np.float32(thr[k])
    print(k, "ties", int((a == float(tf)).sum()), "zeros", int((a == 0).sum()), "n", a.numel())
import time
for fmt in ("csr",):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): res = _engine.run_hot_path(plan, dm, ref, dynamic_threshold=1.5, chunksize=5000, profile=True)
    torch.cuda.synchronize(); print("run ms", (time.perf_counter() - t0) / 3 * 1e3, res.profile.smooth_ms, res.profile.apply_ms)
