#!/usr/bin/env python
"""Why a resident-input call costs more after the host-input (e2e) legs of bench.py ran in the same process
(profiles/r06_bench_n1.json: config2_position_ordered_var 4.05 ms per step, host bound, against 3.24 alone): time the
resident call (random and genome-ordered var) before / after bench.py's own e2e legs, then after each suspected remedy.
    BENCH_E2E_LEGS=1m python tools/host_issue_after_e2e.py"""
import ctypes
import gc
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import pandas as pd  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import cases  # noqa: E402
import infercnvpy_amd as cnv  # noqa: E402
from infercnvpy_amd._compat import SimpleAnnData  # noqa: E402

bench.quiet_repeated_warnings()
v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
vp, _ = cases.position_ordered(v)
var_pos = pd.DataFrame({"chromosome": vp["chromosome"], "start": vp["start"], "end": vp["end"]}, index=vp["names"])
cells = 100_000
X = bench.synth_rows(torch, 0, cells, bench.G)
ad = SimpleAnnData(X, var=var)
ad_pos = SimpleAnnData(X, var=var_pos)
small = SimpleAnnData(X[:2000].contiguous(), var=var)


def timed(a, n=50):
    for _ in range(3):
        cnv.tl.infercnv(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    each = []
    for _ in range(n):
        t1 = time.perf_counter()
        cnv.tl.infercnv(a)
        each.append((time.perf_counter() - t1) * 1e3)
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    slow = [(i, round(x, 1)) for i, x in enumerate(each) if x > 3 * sorted(each)[n // 2]]
    if slow:
        print("      calls slower than 3 x the median (index, ms):", slow[:12], "reserved GB",
              round(torch.cuda.memory_reserved() / 1e9, 1), flush=True)
    return ti / n * 1e3, (time.perf_counter() - t0) / n * 1e3


def show(label):
    a, b = timed(ad)
    e, f = timed(ad_pos)
    c, d = timed(small, 200)
    print(f"{label:52s} random var: issue {a:6.3f} total {b:6.3f} | ordered var: issue {e:6.3f} total {f:6.3f} | "
          f"2000 cells: {d:6.3f} ms | threads {threading.active_count()}", flush=True)


_gc_t = [0.0]


def _gc_cb(phase, info):
    if phase == "start":
        _gc_t[0] = time.perf_counter()
    else:
        dt = (time.perf_counter() - _gc_t[0]) * 1e3
        if dt > 5:
            print(f"      gc generation {info['generation']}: {dt:.1f} ms, collected {info['collected']}, "
                  f"tracked objects {len(gc.get_objects())}", flush=True)


gc.callbacks.append(_gc_cb)
show("fresh process")
legs = bench.e2e_legs(torch)
print("   e2e legs run:", {k[:40]: round(x.get("seconds", 0), 3) for k, x in legs.items()})
show("after bench.e2e_legs")
show("again")
hip = ctypes.CDLL("libamdhip64.so")
pool = ctypes.c_void_p()
hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0)
torch.cuda.synchronize()
hip.hipMemPoolTrimTo(pool, ctypes.c_size_t(0))
show("after hipMemPoolTrimTo(default pool, 0)")
torch.cuda.empty_cache()
show("after torch.cuda.empty_cache()")
gc.collect()
show("after gc.collect()")
