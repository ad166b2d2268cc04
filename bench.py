#!/usr/bin/env python
"""Benchmark of the tl.infercnv hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one batch of synthetic cells that is already
resident in HBM: reference mean (float64 column sums, + one RCCL all-reduce when N > 1) ->
fused centre/clip/pyramid-smooth/median kernel -> per-chunk std -> threshold.  Workload at N=1 =
BASELINE config 2 (dense fp32 100 000 cells x 20 000 genes on chr1..22, window 100, step 10,
chunksize 5000); for N > 1 every rank owns 100 000 cells (weak scaling, chunk-aligned shards, the
only collective is the reference-mean all-reduce).

Prints ONE JSON line on rank 0: metric cells/s (whole job), plus
  roofline     - the smoothing kernel: algorithmic bytes (4*G + 4*W per cell) / its average
                 HIP-event duration, against the 8 TB/s HBM peak
  cpu_baseline - the numpy oracle (a port of the reference algorithm, oracle/) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_on_device(torch, n_cells, n_genes, seed):
    """gamma(0.3, 1) with entries < 0.5 zeroed (~19 % nnz), SURVEY §8(d) config 2 -- on the GPU."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    out = torch.empty((n_cells, n_genes), dtype=torch.float32, device="cuda")
    rows = 10_000
    for r in range(0, n_cells, rows):
        k = min(rows, n_cells - r)
        g = torch._standard_gamma(torch.full((k, n_genes), 0.3, device="cuda"), generator=gen)
        out[r:r + k] = torch.where(g < 0.5, torch.zeros_like(g), g)
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


def cpu_baseline(cells_per_worker=250, window=100, step=10):
    """Oracle (numpy port of the reference algorithm) on the host cores, reference-style fan-out."""
    import numpy as np

    import cases
    from oracle import infercnv_oracle as O

    cores = os.cpu_count() or 1
    n = cells_per_worker * cores
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    X = cases.synthetic_expr(n, 20000, seed=2)
    ref = X.mean(axis=0, dtype=np.float64).astype(np.float32)
    t0 = time.perf_counter()
    O.infercnv(X, v["chromosome"], v["start"], reference=ref, window_size=window, step=step,
               chunksize=cells_per_worker, n_jobs=cores)
    dt = time.perf_counter() - t0
    return {
        "value": n / dt, "unit": "cells/s", "cores": cores, "kind": "port",
        "sample": f"{n} cells x 20000 genes dense fp32, window {window} step {step}, {cores} processes x "
                  f"{cells_per_worker}-cell chunks (ProcessPoolExecutor, per-row np.convolve as the reference), "
                  f"{dt:.1f} s wall",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=100_000, help="cells per GPU")
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--step", type=int, default=10)
    ap.add_argument("--chunksize", type=int, default=5000)
    ap.add_argument("--format", choices=["dense", "csr"], default="dense",
                    help="csr = BASELINE config 4 style input (not the default bench line)")
    ap.add_argument("--density", type=float, default=0.07)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-refmean", action="store_true", help="exclude the reference-mean pass from the step")
    args = ap.parse_args()

    import torch

    import cases
    from infercnvpy_amd import _engine
    from infercnvpy_amd._plan import GenePlan

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    n_gpus = world

    G = 20000
    v = cases.synthetic_var(cases.GENES_PER_CHROM_20K)
    plan = GenePlan(v["chromosome"], v["start"], window_size=args.window, step=args.step)
    W = plan.n_windows
    n_local = args.cells
    if args.format == "dense":
        X = synth_on_device(torch, n_local, G, seed=2 + rank)
        dm = _engine.DeviceMatrix(dense=X)
        nnz_row = G
    else:
        ip, ix, dv = synth_csr_on_device(torch, n_local, G, args.density, seed=3 + rank)
        dm = _engine.DeviceMatrix(indptr=ip, indices=ix, data=dv, shape=(n_local, G))
        nnz_row = dv.numel() / n_local
    out = _engine.alloc_out(n_local, W)
    sums = torch.zeros((1, G), dtype=torch.float64, device="cuda")
    fixed_ref = None
    if args.no_refmean:
        fixed_ref = (_engine.column_sums(dm)[0] / n_local).float()

    def one_step():
        if fixed_ref is None:
            sums.zero_()
            _engine.column_sums(dm, None, 1, sums)
            if dist is not None:
                dist.all_reduce(sums)  # the only collective of the path: [G] float64 over RCCL/xGMI
            ref = (sums[0] / (n_local * n_gpus)).float()
        else:
            ref = fixed_ref
        # no host synchronisation inside a step: the library records HIP events around the smoothing kernel on
        # the launch stream (icv_profile_begin) and the times are read after the timed region
        return _engine.run_hot_path(plan, dm, ref, lfc_clip=3.0, dynamic_threshold=1.5, chunksize=args.chunksize,
                                    out=out)

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
    cells_total = n_local * n_gpus
    value = cells_total / (dt / args.steps)

    # dense 4*G + 4*W = 87 208 B/cell at window 100 / step 10; CSR 8*nnz_row + 8 + 4*W
    bytes_per_cell = (4 * G + 4 * W) if args.format == "dense" else (8 * nnz_row + 8 + 4 * W)
    avg_smooth_ms = sum(smooth_ms) / max(len(smooth_ms), 1)
    achieved = bytes_per_cell * n_local / (avg_smooth_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    # counters were collected for the default workload (dense, window 100, 100 000 cells per launch); traffic is
    # proportional to the cells of a launch, other workloads have no counter data
    if os.path.exists(pmc_path) and args.format == "dense" and args.window == 100 and args.step == 10:
        try:
            traffic = json.load(open(pmc_path)).get("k_smooth_hbm_bytes_per_launch")
            if traffic is not None:
                traffic = traffic * (n_local / 100_000.0)
        except Exception:
            traffic = None

    result = {
        "metric": "cells/sec through tl.infercnv (window=100)" if args.window == 100 else
                  f"cells/sec through tl.infercnv (window={args.window})",
        "value": value,
        "unit": "cells/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",  # arithmetic type from the block sums on (np.convolve is float64); I/O is float32
        "data": "synthetic",
        "config": {
            "workload": ("BASELINE config 2: dense fp32" if args.format == "dense" else
                         f"BASELINE config 4 style: CSR fp32 density {args.density}") +
                        f" {n_local} cells/GPU x {G} genes (chr1..22, random var order), "
                        f"window {args.window}, step {args.step}, chunksize {args.chunksize}, lfc_clip 3, "
                        f"dynamic_threshold 1.5, reference = all-cell mean"
                        + (" (precomputed, excluded from the step)" if args.no_refmean else " (in the step)"),
            "io_dtype": "f32 matrix in, f32 x_res out",
            "cells_total": cells_total,
            "n_windows": W,
            "parallelism": f"row shards x{n_gpus}, all-reduce of the [G] float64 reference sums only",
        },
        "roofline": {
            "kernel": ("k_smooth_ws<10,4,4,10,10> (dense fp32 fast path)" if args.format == "dense" and args.window == 100
                       else "k_smooth_ws (variant for this window / format; generic k_smooth if the plan does not fit)"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "bytes_per_cell": bytes_per_cell,
            "kernel_ms": avg_smooth_ms,
        },
    }
    if rank == 0:
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(window=args.window, step=args.step)
            except Exception as e:  # the GPU number must still be reported
                result["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
