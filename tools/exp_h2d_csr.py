#!/usr/bin/env python
"""Why does the CSR upload of tl.infercnv (two arrays per piece) reach 41 GB/s where the dense one reaches 50?
Pageable host -> HBM copies of the config-4 arrays (indices int32 + values float32, 2.8 GB each) in different orders
and piece sizes, alone and with a device -> host copy running beside them (what the CSR drain does)."""
import sys
import threading
import time

import numpy as np
import torch

n = 700_000_000
idx = np.arange(n, dtype=np.int32)
val = np.ones(n, dtype=np.float32)
d_idx = torch.empty(n, dtype=torch.int32, device="cuda")
d_val = torch.empty(n, dtype=torch.float32, device="cuda")
out_d = torch.ones(150_000_000, dtype=torch.float64, device="cuda")  # 1.2 GB: the size of config 4's X_cnv values
out_h = np.empty(150_000_000, dtype=np.float64)
torch.cuda.synchronize()


def timed(fn, label, bytes_):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:70s} {dt * 1e3:8.1f} ms  {bytes_ / dt / 1e9:6.1f} GB/s", flush=True)


def whole():
    d_idx.copy_(torch.from_numpy(idx))
    d_val.copy_(torch.from_numpy(val))


def pieces(piece, streams=1):
    sts = [torch.cuda.Stream() for _ in range(streams)]

    def run():
        k = 0
        for a in range(0, n, piece):
            b = min(n, a + piece)
            with torch.cuda.stream(sts[k % streams]):
                d_idx[a:b].copy_(torch.from_numpy(idx[a:b]))
            k += 1
            with torch.cuda.stream(sts[k % streams]):
                d_val[a:b].copy_(torch.from_numpy(val[a:b]))
            k += 1
    return run


def two_threads():
    def up(dst, src):
        with torch.cuda.stream(torch.cuda.Stream()):
            piece = 91_000_000
            for a in range(0, n, piece):
                b = min(n, a + piece)
                dst[a:b].copy_(torch.from_numpy(src[a:b]))
            torch.cuda.current_stream().synchronize()
    ths = [threading.Thread(target=up, args=(d_idx, idx)), threading.Thread(target=up, args=(d_val, val))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


def with_d2h(fn):
    def run():
        def back():
            with torch.cuda.stream(torch.cuda.Stream()):
                for a in range(0, out_d.numel(), 30_000_000):
                    torch.from_numpy(out_h[a:a + 30_000_000]).copy_(out_d[a:a + 30_000_000])
        th = threading.Thread(target=back)
        th.start()
        fn()
        th.join()
    return run


B = 2 * 4 * n
for rep in range(2):
    timed(whole, "two whole arrays, one after the other", B)
    timed(pieces(91_000_000), "pieces of 364 MB, alternating arrays (what SlabStream does)", B)
    timed(pieces(500_000_000 // 4 * 4), "pieces of 2 GB, alternating arrays", B)
    timed(pieces(91_000_000, streams=2), "pieces of 364 MB, alternating arrays on two streams", B)
    timed(two_threads, "two uploader threads, one per array", B)
    timed(with_d2h(pieces(91_000_000)), "pieces of 364 MB + a 1.2 GB device -> host copy beside them", B)
    timed(with_d2h(two_threads), "two uploader threads + the device -> host copy", B)
