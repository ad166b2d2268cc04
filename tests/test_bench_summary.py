"""CPU tests of bench.py's host logic: the digest that survives a truncated record is built from
whatever legs the line holds -- errored or skipped legs must not cost the headline -- and from the committed evidence
line it repeats the numbers the judge reads."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_summary_of_a_minimal_and_of_a_broken_line():
    s = bench._summary({"value": 3.1e7, "ms_per_step": 3.2, "roofline": {"frac": 0.7, "kernel_ms": 1.55}})
    assert s == {"value_cells_per_s": 31000000.0, "ms_per_step": 3.2, "roofline_frac": 0.7, "kernel_ms": 1.55}
    s = bench._summary({"value": 1.0, "ms_per_step": None, "roofline": None, "e2e": "failed", "cpu_baseline": None,
                        "extra": {"config4_csr_w250": {"error": "boom"}, "scores_and_gene_values": {"error": "x" * 500},
                                  "config3_cells_on_one_gpu": {"ms_per_step": 30.0, "roofline": {}},
                                  "config5": {"error": "oom"}}})
    assert s["value_cells_per_s"] == 1.0 and s["ms_per_step"] is None and s["roofline_frac"] is None
    assert len(s["scores_and_gene_values_error"]) == 200 and "config4_ms_per_step" not in s
    assert s["one_million_cells_ms_per_step"] == 30.0 and s["one_million_cells_roofline_frac"] is None
    json.dumps(s)  # serialisable whatever went wrong


def test_summary_of_an_n_rank_line():
    s = bench._summary({"value": 2.0e8, "ms_per_step": 5.0, "roofline": {"frac": 0.7, "kernel_ms": 1.9},
                        "value_allreduce_means": 2.4e8, "value_chained_means": 1.0e8, "value_blocks_means": 2.0e8})
    assert s["value_allreduce_means"] == 2.4e8 and s["value_chained_means"] == 1.0e8 and s["value_blocks_means"] == 2.0e8


def test_summary_repeats_the_committed_evidence_line():
    path = os.path.join(ROOT, "profiles", "r06_bench_n1.json")
    rec = json.loads(open(path).read().strip().splitlines()[-1])
    assert list(rec)[-1] == "summary"
    s = bench._summary(rec)
    for k, v in rec["summary"].items():  # (keys added to the digest after the run may be missing from the record)
        assert s[k] == v, k
    assert s["value_cells_per_s"] == round(rec["value"], 0)
    assert s["config4_ms_per_step"] == round(rec["extra"]["config4_csr_w250"]["ms_per_step"], 3)
    assert set(s["e2e_cells_per_s"]) == {k[:70] for k in rec["e2e"]}
