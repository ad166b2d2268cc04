#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection.csv: per kernel name, mean counter value per dispatch."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")[:60]
    acc[k][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", 0)))
for k, cs in acc.items():
    if not any(w in k for w in ("smooth", "apply", "colsum", "colchain", "thr_", "csr_", "row_offsets", "gram", "ward", "gene", "chain_", "key_hist")):
        continue
    print(k)
    for c, vals in sorted(cs.items()):
        print(f"   {c:28s} n={len(vals):3d} mean={sum(vals) / len(vals):.6g}")
