#!/usr/bin/env python
"""The GPU timeline of one cnv.tl.infercnv(adata, calculate_gene_values=True) call on resident config-2 data: run under
rocprofv3 --kernel-trace (tools/trace_gene_call.sh), then list kernels and the idle gaps between them."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
    import csv

    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    # the last call: kernels after the last k_colchain
    idx = [i for i, r in enumerate(rows) if "k_colchain<" in r["Kernel_Name"]]
    last = rows[idx[-1]:]
    t_prev = None
    t0 = int(last[0]["Start_Timestamp"])
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if t_prev is None else (s - t_prev) / 1e3
        print(f"  +{(s - t0) / 1e6:7.3f} ms  gap {gap:8.1f} us  {(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:70]}")
        t_prev = e
    print(f"  call on the GPU: {(t_prev - t0) / 1e6:.3f} ms")
    sys.exit(0)
import pandas as pd  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import cases  # noqa: E402
import infercnvpy_amd as cnv  # noqa: E402
from infercnvpy_amd._compat import SimpleAnnData  # noqa: E402

bench.quiet_repeated_warnings()
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
ad = SimpleAnnData(bench.synth_rows(torch, 0, 100_000, bench.G), var=var)
for _ in range(4):
    ad.layers.clear()
    cnv.tl.infercnv(ad, calculate_gene_values=True)
torch.cuda.synchronize()
