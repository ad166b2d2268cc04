#!/usr/bin/env python
"""Host -> HBM copy rates for a large pageable numpy matrix: torch's pageable copy against a pinned staging ring
filled by a thread pool (numpy copies release the GIL).  Decides how tl.infercnv stages its input."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
rows = int(gb * 1e9 / 80000)
X = np.ones((rows, 20000), dtype=np.float32)
X[::1000] = 2.0
dev = torch.empty((rows, 20000), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()


def pageable():
    t0 = time.perf_counter()
    dev.copy_(torch.from_numpy(X))
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def staged(n_threads, piece_rows, n_slots=3):
    pool = ThreadPoolExecutor(n_threads)
    slots = [torch.empty((piece_rows, 20000), dtype=torch.float32).pin_memory() for _ in range(n_slots)]
    views = [s.numpy() for s in slots]
    events = [None] * n_slots
    st = torch.cuda.Stream()

    def fill(slot, r0, r1):
        n = r1 - r0
        sub = max(1, n // n_threads)
        futs = []
        for a in range(0, n, sub):
            b = min(n, a + sub)
            futs.append(pool.submit(np.copyto, views[slot][a:b], X[r0 + a:r0 + b]))
        for f in futs:
            f.result()

    def run():
        t0 = time.perf_counter()
        k = 0
        for r0 in range(0, rows, piece_rows):
            r1 = min(rows, r0 + piece_rows)
            s = k % n_slots
            if events[s] is not None:
                events[s].synchronize()
            fill(s, r0, r1)
            with torch.cuda.stream(st):
                dev[r0:r1].copy_(slots[s][: r1 - r0], non_blocking=True)
                events[s] = torch.cuda.Event()
                events[s].record(st)
            k += 1
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run()
    best = min(run() for _ in range(2))
    pool.shutdown()
    return best


pageable()
t = min(pageable() for _ in range(2))
print(f"pageable torch copy           {X.nbytes / t / 1e9:6.1f} GB/s")
tp = torch.from_numpy(X[: rows // 4]).pin_memory()
torch.cuda.synchronize()
t0 = time.perf_counter()
dev[: rows // 4].copy_(tp, non_blocking=True)
torch.cuda.synchronize()
print(f"pinned source (upper bound)   {tp.numel() * 4 / (time.perf_counter() - t0) / 1e9:6.1f} GB/s")
del tp
for n_threads in (4, 8, 16, 32):
    for piece_mb in (64, 256):
        pr = piece_mb * (1 << 20) // 80000
        t = staged(n_threads, pr)
        print(f"staged ring {n_threads:2d} threads {piece_mb:4d} MB pieces  {X.nbytes / t / 1e9:6.1f} GB/s")
ok = bool((dev[::1000, 0] == 2.0).all().item())
print("check", ok)
