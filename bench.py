#!/usr/bin/env python
"""Benchmark of the tl.infercnv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script re-executes itself as
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` (one rank per
GPU, backend nccl = RCCL over xGMI); it exits non-zero if the box has fewer than N GPUs.  Started by a launcher
(WORLD_SIZE set) it checks that WORLD_SIZE == --gpus.

One "step" = one pass of the whole hot path over one batch of synthetic cells that is already resident in HBM:
reference mean (float64 column sums, + ONE RCCL all-reduce of [G + 1] float64 when N > 1) -> fused centre / clip /
pyramid-smooth / median kernel -> per-chunk std -> threshold.
  N = 1   BASELINE config 2: dense fp32 100 000 cells x 20 000 genes (chr1..22, random var order), window 100,
          step 10, chunksize 5000.
  N > 1   BASELINE config 3: 1 000 000 cells x 20 000 genes in total, row shards aligned to the 5000-cell chunks
          (dist.shard_bounds; 125 000 cells per GPU at N = 8): strong scaling.  Every 5000-cell chunk is generated
          from its own seed, so the data do not depend on N.

Prints ONE JSON line on rank 0: value = cells/s of the whole job (HBM-resident input), plus
  roofline     - the smoothing kernel: algorithmic bytes (4*G + 4*W per cell) / its average HIP-event duration
                 (events recorded by the library on the launch stream), against the 8 TB/s HBM peak
  cpu_baseline - the numpy oracle (a port of the reference algorithm, oracle/) on this box's host cores on a
                 bounded sample of the same workload (rank 0, N = 1 only)
  e2e          - the public cnv.tl.infercnv(adata) from HOST memory to a host CSR X_cnv (PCIe-inclusive; rank 0,
                 N = 1 only): dense 200 000 x 20 000 and the config-4 CSR, with the stage times.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHUNK = 5000
CONFIG3_CELLS = 1_000_000


def synth_chunk(torch, n_rows, n_genes, seed):
    """gamma(0.3, 1) with entries < 0.5 zeroed (~19 % nnz), SURVEY §8(d) config 2 -- generated on the GPU."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    g = torch._standard_gamma(torch.full((n_rows, n_genes), 0.3, device="cuda"), generator=gen)
    return torch.where(g < 0.5, torch.zeros_like(g), g)


def synth_rows(torch, row0, row1, n_genes, chunk=CHUNK, seed0=2):
    """Rows [row0, row1) of the synthetic matrix; chunk k (rows k*chunk ...) has seed seed0 + k."""
    out = torch.empty((row1 - row0, n_genes), dtype=torch.float32, device="cuda")
    r = row0
    while r < row1:
        k = r // chunk
        blk = synth_chunk(torch, chunk, n_genes, seed0 + k)
        a, b = r - k * chunk, min(row1, (k + 1) * chunk) - k * chunk
        out[r - row0: r - row0 + (b - a)] = blk[a:b]
        r += b - a
    return out


def synth_csr_on_device(torch, n_cells, n_genes, density, seed):
    """10x-like CSR: Bernoulli(density) mask x log1p(1 + poisson-ish counts)."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    indptr = [torch.zeros(1, dtype=torch.int64, device="cuda")]
    indices, data = [], []
    rows, nnz = 10_000, 0
    for r in range(0, n_cells, rows):
        k = min(rows, n_cells - r)
        mask = torch.rand((k, n_genes), device="cuda", generator=gen) < density
        vals = torch.log1p(1.0 + torch.floor(torch.rand((k, n_genes), device="cuda", generator=gen) ** 3 * 8.0))
        counts = mask.sum(dim=1)
        indptr.append(nnz + torch.cumsum(counts, 0))
        nnz += int(counts.sum())
        idx = mask.nonzero(as_tuple=False)
        indices.append(idx[:, 1].to(torch.int32))
        data.append(vals[mask].float())
    return torch.cat(indptr), torch.cat(indices), torch.cat(data)


# ----------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle's chunk kernel, one worker per host core; every worker generates its own chunk, so
# the number is compute only (no pickling of the matrix through the pool feeder)
# ----------------------------------------------------------------------------------------------------------
def _cpu_worker(args):
    import numpy as np

    import cases
    from oracle import infercnv_oracle as O

    seed, n_cells, window, step, reps = args
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    X = cases.synthetic_expr(n_cells, 20000, seed=seed)
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)[None, :]
    t0 = time.perf_counter()
    for _ in range(reps):
        O.infercnv_chunk(X, v["chromosome"], v["start"], ref, 3, window, step, 1.5)
    return time.perf_counter() - t0


def cpu_baseline(window=100, step=10, cells_per_worker=400, reps=3):
    from concurrent.futures import ProcessPoolExecutor

    cores = os.cpu_count() or 1
    tasks = [(100 + i, cells_per_worker, window, step, reps) for i in range(cores)]
    t0 = time.perf_counter()
    with ProcessPoolExecutor(max_workers=cores) as pool:
        busy = list(pool.map(_cpu_worker, tasks))
    wall = time.perf_counter() - t0
    n = cells_per_worker * cores * reps
    # all workers run concurrently: throughput = cells / the slowest worker's compute time
    return {
        "value": n / max(busy), "unit": "cells/s", "cores": cores, "kind": "port",
        "sample": f"{cores} processes x {reps} x {cells_per_worker}-cell chunks ({n} cells x 20000 genes dense fp32, "
                  f"window {window} step {step}), oracle chunk kernel (per-row np.convolve as the reference) on "
                  f"worker-local data; slowest worker {max(busy):.1f} s compute, {wall:.1f} s wall incl. start-up",
    }


# ----------------------------------------------------------------------------------------------------------
# end to end through the public API, from host memory
# ----------------------------------------------------------------------------------------------------------
def e2e_legs(torch, window_dense=100, dense_cells=200_000, csr_cells=500_000):
    import numpy as np
    import pandas as pd
    import scipy.sparse as sp

    import cases
    import infercnvpy_amd as cnv
    from infercnvpy_amd._compat import SimpleAnnData

    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    var = pd.DataFrame({"chromosome": v["chromosome"], "start": v["start"], "end": v["end"]}, index=v["names"])
    legs = {}

    def run(name, X, window, **kw):
        import gc

        ref = np.asarray(X[:2000].mean(axis=0), dtype=np.float64).ravel().astype(np.float32)
        best, ad = None, None
        for _ in range(2):  # first call pays one-time costs (pinned staging buffers, plan tables)
            ad = None  # a fresh AnnData per call: releasing the previous X_cnv (> 1 GB) is not part of the call
            gc.collect()
            ad = SimpleAnnData(X, var=var)
            tm = {}
            t0 = time.perf_counter()
            cnv.tl.infercnv(ad, reference=ref, window_size=window, step=10, _timings=tm, **kw)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, tm)
        dt, tm = best
        in_bytes = X.data.nbytes + X.indices.nbytes + X.indptr.nbytes if sp.issparse(X) else X.nbytes
        legs[name] = {
            "cells": int(X.shape[0]), "seconds": dt, "cells_per_s": X.shape[0] / dt,
            "h2d_GBps": in_bytes / max(tm.get("h2d", dt), 1e-9) / 1e9, "input_GB": in_bytes / 1e9,
            "x_cnv_nnz": int(ad.obsm["X_cnv"].nnz), "stages_s": {k: round(float(x), 4) for k, x in tm.items()},
        }

    Xd = synth_rows(torch, 0, dense_cells, 20000).cpu().numpy()
    run(f"dense fp32 {dense_cells} x 20000, window {window_dense}", Xd, window_dense)
    del Xd
    ip, ix, dv = synth_csr_on_device(torch, csr_cells, 20000, 0.07, seed=3)
    Xs = sp.csr_matrix((dv.cpu().numpy(), ix.cpu().numpy(), ip.cpu().numpy()), shape=(csr_cells, 20000))
    del ip, ix, dv
    torch.cuda.empty_cache()
    run(f"CSR fp32 {csr_cells} x 20000 density 0.07, window 250 (BASELINE config 4)", Xs, 250)
    return legs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # default: 300 steps = 1 s of timed GPU work (a 20-step region is 65 ms: too short for an outside sampler to see,
    # and short enough to sit inside the boost window of the clocks)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=None, help="total cells (default: 100 000 at N=1, 1 000 000 at N>1)")
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--step", type=int, default=10)
    ap.add_argument("--chunksize", type=int, default=CHUNK)
    ap.add_argument("--format", choices=["dense", "csr"], default="dense",
                    help="csr = BASELINE config 4 style input (not the default bench line)")
    ap.add_argument("--density", type=float, default=0.07)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-refmean", action="store_true", help="exclude the reference-mean pass from the step")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # no launcher: spawn one rank per GPU ourselves (fail loudly rather than report a fake n_gpus)
        import torch

        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} requested but this box has {have} visible GPU(s)")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}")

    import torch

    import cases
    from infercnvpy_amd import _engine
    from infercnvpy_amd import dist as icd
    from infercnvpy_amd._plan import GenePlan

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        if torch.cuda.device_count() <= local_rank:
            sys.exit(f"bench.py: rank {rank} has no GPU (device_count {torch.cuda.device_count()})")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world
    else:
        torch.cuda.set_device(0)
    n_gpus = world

    G = 20000
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=args.window, step=args.step)
    W = plan.n_windows
    n_total = args.cells if args.cells is not None else (100_000 if n_gpus == 1 else CONFIG3_CELLS)
    bounds = icd.shard_bounds(n_total, n_gpus, args.chunksize)
    row0, row1 = bounds[rank]
    n_local = row1 - row0
    if args.format == "dense":
        X = synth_rows(torch, row0, row1, G, chunk=args.chunksize)
        dm = _engine.DeviceMatrix(dense=X)
        nnz_row = G
    else:
        ip, ix, dv = synth_csr_on_device(torch, n_local, G, args.density, seed=3 + rank)
        dm = _engine.DeviceMatrix(indptr=ip, indices=ix, data=dv, shape=(n_local, G))
        nnz_row = dv.numel() / max(n_local, 1)
    out = _engine.alloc_out(n_local, W)
    sums = torch.zeros((1, G), dtype=torch.float64, device="cuda")
    fixed_ref = None
    if args.no_refmean:
        fixed_ref = (_engine.column_sums(dm)[0] / n_local).float()

    def one_step():
        if fixed_ref is None:
            sums.zero_()
            _engine.column_sums(dm, None, 1, sums)
            # the only collective of the path: [G] float64 sums + the row count, over RCCL / xGMI
            ref = icd.reference_means(sums, [n_local], "float32", device_out=True)[0] if dist is not None \
                else (sums[0] / n_local).float()
        else:
            ref = fixed_ref
        # no host synchronisation inside a step: the library records HIP events around the smoothing kernel on
        # the launch stream (icv_profile_begin) and the times are read after the timed region
        return icd.run_shard(plan, dm, ref, global_row0=row0, n_obs_global=n_total, lfc_clip=3.0,
                             dynamic_threshold=1.5, chunksize=args.chunksize, all_bounds=bounds, out=out)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    _engine.profile_begin(plan)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    dt = time.perf_counter() - t0
    smooth_ms = [r.smooth_ms for r in _engine.profile_collect(plan)]
    assert len(smooth_ms) == args.steps
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps)

    # dense 4*G + 4*W = 87 208 B/cell at window 100 / step 10; CSR 8*nnz_row + 8 + 4*W
    bytes_per_cell = (4 * G + 4 * W) if args.format == "dense" else (8 * nnz_row + 8 + 4 * W)
    avg_smooth_ms = sum(smooth_ms) / max(len(smooth_ms), 1)
    achieved = bytes_per_cell * n_local / (avg_smooth_ms * 1e-3) / 1e9
    x16 = args.format == "dense" and args.window == 100 and args.step == 10
    if x16:
        kernel_name = "k_smooth_x16<10,10,chunk moments> (dense fp32, window 100 / step 10)"
    elif args.format == "dense" and args.window == 250 and args.step == 10:
        kernel_name = "k_smooth_x16<5,50,chunk moments> (dense fp32, window 250 / step 10)"
    elif (args.format == "csr" and args.window % 2 == 0
          and args.window // math.gcd(args.step, args.window // 2) > 10):  # long windows (icv_api.hip: sd_fraction_bits)
        kernel_name = ("k_sd_table + k_sd_base + k_smooth_sd (CSR, long windows: stored entries only, differences to "
                       "the zero row in fixed-point block bins)")
    elif args.format == "csr":
        kernel_name = "k_csr_prepare + k_smooth_ws<..., CSR> (prepared entries on a zero row in LDS)"
    else:
        kernel_name = "k_smooth_ws (variant for this window; generic k_smooth if the plan does not fit)"
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    # HBM bytes from the PMC counters are collected by tools/r02_round_end.sh (separate rocprofv3 --pmc passes) for
    # the default workload and committed; they scale with the cells of a launch.  Other workloads: no counter data.
    if os.path.exists(pmc_path) and x16:
        try:
            rec = json.load(open(pmc_path))
            if "x16" in rec.get("kernel", ""):
                traffic = rec.get("k_smooth_hbm_bytes_per_launch") * (n_local / 100_000.0)
        except Exception:
            traffic = None

    result = {
        "metric": f"cells/sec through the tl.infercnv hot path (window={args.window}), input resident in HBM",
        "value": value,
        "unit": "cells/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if n_gpus > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f64",  # arithmetic type from the block sums on (np.convolve is float64); I/O is float32
        "data": "synthetic",
        "config": {
            "workload": (("BASELINE config 2: dense fp32" if n_gpus == 1 else "BASELINE config 3: dense fp32")
                         if args.format == "dense" else f"BASELINE config 4 style: CSR fp32 density {args.density}") +
                        f" {n_total} cells x {G} genes (chr1..22, random var order), "
                        f"window {args.window}, step {args.step}, chunksize {args.chunksize}, lfc_clip 3, "
                        f"dynamic_threshold 1.5, reference = all-cell mean"
                        + (" (precomputed, excluded from the step)" if args.no_refmean else " (in the step)"),
            "io_dtype": "f32 matrix in, f32 x_res out",
            "cells_total": n_total,
            "cells_per_gpu": [b - a for a, b in bounds],
            "n_windows": W,
            "parallelism": f"{n_gpus} rank(s) (torch.distributed world size "
                           f"{dist.get_world_size() if dist is not None else 1}, backend "
                           f"{'nccl/RCCL' if dist is not None else 'none'}), row shards aligned to the chunks, one "
                           f"all-reduce of the [G+1] float64 reference sums per step, no other collective",
        },
        "roofline": {
            "kernel": kernel_name,
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_cell": bytes_per_cell,
            "kernel_ms": avg_smooth_ms,
            "kernel_ms_min_max": [min(smooth_ms), max(smooth_ms)],
            "cells_per_launch": n_local,
        },
    }
    if rank == 0 and n_gpus == 1:
        if not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(window=args.window, step=args.step)
            except Exception as e:  # the GPU number must still be reported
                result["cpu_baseline"] = {"error": repr(e)}
        if not args.no_e2e and args.format == "dense":
            del X, dm, out
            torch.cuda.empty_cache()
            try:
                result["e2e"] = e2e_legs(torch)
            except Exception as e:
                result["e2e"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
