"""First-contact insurance for the driver's multi-GPU bench: the N-rank command line of ``bench.py``
runs on every GPU test run -- all ranks on cuda:0, collectives through gloo (``--dry-run-one-gpu``: prints
``"dry_run": true``, never a measurement) -- and its ONE JSON line is parsed: the keys a scaling record is built from are
there, the work was sharded as ``dist.shard_bounds`` says, both forms of the reference means were timed.  No hardware
curve is asked of this test."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().splitlines() if ln.strip()]
    assert lines, p.stderr[-2000:]
    # the JSON line is the LAST line of stdout, and the only one that parses as the result object
    rec = json.loads(lines[-1])
    return rec, lines, p.stderr


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_n_rank_command_line_dry_run(n_ranks):
    rec, lines, err = _run_bench("--gpus", str(n_ranks), "--dry-run-one-gpu", "--steps", "2", "--warmup", "1", "--no-e2e")
    assert rec["dry_run"] is True
    assert rec["n_gpus"] == n_ranks and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["unit"] == "cells/s" and rec["higher_is_better"] is True
    assert rec["value"] > 0 and rec["value_allreduce_means"] > 0 and rec["value_chained_means"] > 0
    assert rec["value_blocks_means"] > 0
    assert set(rec["forms"]) >= {"value", "value_allreduce_means", "value_chained_means", "value_blocks_means"}
    assert "integer blocks" in rec["forms"]["value"]
    # `value` is the faster of the two exact forms of the means
    assert rec["value"] == pytest.approx(max(rec["value_blocks_means"], rec["value_chained_means"]), rel=1e-9)
    per_gpu = rec["config"]["cells_per_gpu"]
    assert len(per_gpu) == n_ranks and sum(per_gpu) == rec["config"]["cells_total"]
    assert all(c % 5000 == 0 for c in per_gpu[:-1])  # chunk-aligned shards
    assert rec["roofline"]["bound"] == "hbm" and 0 < rec["roofline"]["frac"] < 1.5
    assert "gloo" in rec["config"]["parallelism"]
    # the digest is the last key of the line (what survives a truncated record) and repeats the headline
    assert list(rec)[-1] == "summary"
    assert rec["summary"]["value_cells_per_s"] == round(rec["value"], 0)


def test_one_gpu_line_ends_with_json_and_warns_once():
    """N = 1 at a reduced size: the JSON line is the last line of stdout, the digest is its last key, and the
    reference's "Using mean of all cells" warning appears once in the process output, not once per call."""
    rec, lines, err = _run_bench("--steps", "3", "--warmup", "1", "--cells", "20000", "--no-cpu-baseline", "--no-e2e",
                                 "--no-extra")
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["value"] > 0
    assert rec["roofline"]["kernel_ms"] > 0 and "stages" in rec
    assert list(rec)[-1] == "summary"
    assert err.count("Using mean of all cells as reference") <= 1
    assert "bench-summary " in err
