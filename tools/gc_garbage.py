import gc, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, pandas as pd, torch, scipy.sparse as sp
import bench, cases
import infercnvpy_amd as cnv
from infercnvpy_amd._compat import SimpleAnnData
bench.quiet_repeated_warnings()
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
H = bench.synth_rows(torch, 0, 50_000, bench.G).cpu().numpy()
ref = H[:2000].mean(axis=0)
def run(kind):
    gc.collect(); gc.set_debug(gc.DEBUG_SAVEALL); gc.garbage.clear()
    if kind == "host dense":
        ad = SimpleAnnData(H, var=var); cnv.tl.infercnv(ad, reference=ref, devices=[0]); del ad
    elif kind == "host dense, reference=None":
        ad = SimpleAnnData(H, var=var); cnv.tl.infercnv(ad); del ad
    elif kind == "host csr":
        ad = SimpleAnnData(sp.csr_matrix(np.where(H > 1.0, H, 0)), var=var); cnv.tl.infercnv(ad, window_size=250); del ad
    else:
        ad = SimpleAnnData(torch.from_numpy(H[:20000]).cuda(), var=var); cnv.tl.infercnv(ad); del ad
    n = gc.collect()
    hist = collections.Counter(type(o).__name__ for o in gc.garbage)
    big = [(type(o).__name__, getattr(o, "nbytes", None) or (o.numel() * o.element_size() if hasattr(o, "numel") else 0)) for o in gc.garbage
           if isinstance(o, np.ndarray) or hasattr(o, "numel")]
    big = sorted([b for b in big if b[1] > 1 << 20], key=lambda b: -b[1])[:8]
    print(kind, "-> cyclic garbage objects:", n, hist.most_common(12), "large buffers:", big, flush=True)
    fr = [o for o in gc.garbage if type(o).__name__ in ("function", "cell", "frame")][:6]
    for o in fr:
        print("     ", type(o).__name__, getattr(o, "__qualname__", ""), getattr(getattr(o, "f_code", None), "co_name", ""))
    gc.set_debug(0); gc.garbage.clear()
for k in ("resident", "host dense", "host dense, reference=None", "host csr"):
    run(k); run(k)
