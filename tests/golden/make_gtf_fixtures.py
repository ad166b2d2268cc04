"""Reduce the reference's GTF test data (tests/data/*.gtf, data files -- no code) to small fixtures.

Keeps every ``gene`` record plus the first 40 non-gene records (so that feature filtering is exercised).
Run in the build container (needs /root/reference):   python tests/golden/make_gtf_fixtures.py
"""
import os

SRC = "/root/reference/tests/data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "data")

for name in ("chr21_gencode.gtf", "chr1_ensembl.gtf"):
    kept, other = [], 0
    with open(os.path.join(SRC, name)) as fh:
        for line in fh:
            if line.startswith("#"):
                kept.append(line)
                continue
            f = line.split("\t")
            if len(f) > 2 and f[2] == "gene":
                kept.append(line)
            elif other < 40:
                kept.append(line)
                other += 1
    os.makedirs(DST, exist_ok=True)
    with open(os.path.join(DST, name.replace(".gtf", "_genes.gtf")), "w") as out:
        out.writelines(kept)
    print(name, len(kept), "lines")
