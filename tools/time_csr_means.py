#!/usr/bin/env python
"""The CSR reference means (k_csr_tile_bounds16 + k_colchain_csrq) over the density of the matrix: time of the stage and
the statistics that explain it -- stored entries per (row, column tile), the share of rows whose entries of a tile exceed
the producers' 16 slots (those rows take the guarded per-entry loads), LDS rows per round of 64 input rows.
    python tools/time_csr_means.py [cells]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from infercnvpy_amd import _engine  # noqa: E402

G = 20000
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
n_lines, n_tiles = G * 4 // 128, 256
line_tile = torch.empty(n_lines, dtype=torch.int64)
for t in range(n_tiles):
    line_tile[t * n_lines // n_tiles:(t + 1) * n_lines // n_tiles] = t
line_tile = line_tile.cuda()
for dens in (0.02, 0.035, 0.07, 0.105, 0.14, 0.21):
    ip, ix, dv = bench.synth_csr_on_device(torch, cells, G, dens, seed=3)
    dm = _engine.DeviceMatrix(indptr=ip, indices=ix, data=dv, shape=(cells, G))
    for _ in range(2):
        _engine.column_chain(dm, None, None, cells)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 5
    ev[0].record()
    for _ in range(reps):
        _engine.column_chain(dm, None, None, cells)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    # statistics on the first 64 000 rows
    sub = 64_000
    e1 = int(ip[sub].item())
    rows = torch.repeat_interleave(torch.arange(sub, device="cuda"), (ip[1:sub + 1] - ip[:sub]))
    tile = line_tile[(ix[:e1].long() * 4) >> 7]
    cnt = torch.zeros(sub * n_tiles, dtype=torch.int32, device="cuda")
    cnt.index_add_(0, rows * n_tiles + tile, torch.ones(e1, dtype=torch.int32, device="cuda"))
    cnt = cnt.view(sub, n_tiles)
    over16 = (cnt > 16).float().mean().item()
    over8 = (cnt > 8).float().mean().item()
    # rounds (64 rows) with at least one row over 16 entries in a tile
    rounds_slow = (cnt.view(sub // 64, 64, n_tiles) > 16).any(dim=1).float().mean().item()
    nnz = dv.numel()
    print(f"density {dens:5.3f}  entries/row {nnz / cells:7.1f}  means stage {ms:7.3f} ms  ({nnz * 8 / ms / 1e6:6.0f} GB/s of "
          f"the entries)  ms per 1e9 entries {ms / (nnz / 1e9):6.3f}  | entries per (row, tile) mean {cnt.float().mean().item():5.2f}"
          f"  rows>8 {over8:7.4f}  rows>16 {over16:7.4f}  rounds with a >16 row {rounds_slow:6.3f}", flush=True)
    del ip, ix, dv, dm, rows, tile, cnt
    torch.cuda.empty_cache()
